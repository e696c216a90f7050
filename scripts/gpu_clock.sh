#!/bin/bash
# Effective shader clock per kernel: GRBM_GUI_ACTIVE (GPU-busy cycles) / kernel duration.  usage: gpu_clock.sh <conv-mode>
MODE="${1:-bf16}"
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=$ROOT/gpurun_out/clock_$MODE
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o c -- python $ROOT/bench.py --steps 6 --warmup 2 --no-alt --no-cpu-baseline --conv-mode $MODE > $OUT/run.log 2>&1
cd $ROOT
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
cc = glob.glob(os.path.join(out, '**/*counter_collection.csv'), recursive=True)[0]
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(cc)):
    if r['Counter_Name'] != 'GRBM_GUI_ACTIVE': continue
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
    k = r['Kernel_Name'][:70]
    agg[k][0] += float(r['Counter_Value']); agg[k][1] += d; agg[k][2] += 1
for k, (cyc, sec, n) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print('%-72s n=%4d  %.3f GHz  avg %.1f us' % (k, n, cyc / sec / 1e9, sec / n * 1e6))
PY
