#!/bin/bash
# does bench.py's in-step launch time agree with rocprofv3's average over the same command?
cd $GRAFT_REPO_ROOT
bash scripts/gpu_bench_summary.sh r3_check --no-cpu-baseline --no-alt | head -8
bash scripts/gpu_prof.sh r3_check_prof python bench.py --no-alt --no-cpu-baseline | head -6
python - <<'PY'
import csv, glob, json
b = json.loads([l for l in open('gpurun_out/r3_check_prof/stdout.log') if l.startswith('{')][-1])
r = b['roofline']
f = glob.glob('gpurun_out/r3_check_prof/**/*kernel_stats.csv', recursive=True)[0]
for row in csv.DictReader(open(f)):
  if 'gemm_tn_kernel<128' in row['Name']:
    avg = float(row['AverageNs']) / 1e6
    print('rocprof gemm_tn<128>: calls %s avg %.4f ms -> frac %.4f ; bench (same run) in-step avg %.4f ms frac %.4f isolated %.4f ms ; profiled step %.3f vs %.3f ms' % (
        row['Calls'], avg, r['algorithmic_gflop_per_launch'] / avg / 157.3 / 1e3 * 1e3 / 1e3 * 1e3 if False else r['algorithmic_gflop_per_launch'] / (avg * 1e-3) / 1e3 / 157.3,
        r['avg_launch_ms'], r['frac'], r['isolated']['avg_launch_ms'], b['profiled_ms_per_step'], b['ms_per_step']))
PY
