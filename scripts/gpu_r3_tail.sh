#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
timeout 900 python -m pytest tests/test_gpu_fft_conv.py tests/test_gpu_bf16.py tests/test_gpu_fullsize_grads.py -q -m gpu -x 2>&1 | tail -12 > gpurun_out/r3d/pytest.log
tail -5 gpurun_out/r3d/pytest.log
for T in 0 1; do
  echo "== no_tail_split=$T"
  timeout 300 python bench.py --steps-only --steps 30 --warmup 5 --tune no_tail_split=$T 2>/dev/null | cut -c1-260
done
echo "== bf16 (XCD rectangles)"
timeout 300 python scripts/bench_conv_bf16.py --layers 8,9 2>&1 | tail -4
timeout 300 python bench.py --conv-mode bf16 --steps-only --steps 20 --warmup 5 2>/dev/null | cut -c1-260
timeout 300 python bench.py --conv-mode bf16 --steps-only --steps 20 --warmup 5 --tune xcd_gm=1 2>/dev/null | cut -c1-260
timeout 600 python bench.py --no-cpu-baseline --no-alt > gpurun_out/r3d/bench.json 2> gpurun_out/r3d/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r3d/bench.json') if l.startswith('{')][-1])
r = d['roofline']
print('bench', d['ms_per_step'], d['ms_per_step_median'], r['kernel'], 'in-step frac', r['frac'], 'isolated', (r.get('isolated') or {}).get('frac'),
      'hw', d.get('step_executed_gflop'), d.get('step_hw_frac'), 'profiled ms', d.get('profiled_ms_per_step'))
for k, v in r['per_shape'].items():
  print('   ', k, v)
for g in r['by_kernel']:
  print('  ', g['kernel'], g['launches_per_step'], g['ms_per_step'], g.get('frac'), (g.get('isolated') or {}).get('frac'))
PY
