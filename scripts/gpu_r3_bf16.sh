#!/bin/bash
# round 3: the one-wave-per-SIMD schedule of the 256 x 256 bf16 kernel (bf16_sched: 1 = ping-pong of rounds 1-2, 2 = new with a
# ring of 4, 3 = new with a ring of 3) -- parity first, then the per-layer micro-benchmark and the bf16 step under each
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r3b/pytest_bf16.log
cat gpurun_out/r3b/pytest_bf16.log
for S in 1 2 3; do
  echo "== bf16_sched=$S"
  timeout 300 python scripts/bench_conv_bf16.py --layers 8,9 --tune bf16_sched=$S 2>&1 | tail -5 | tee gpurun_out/r3b/conv_bf16_sched$S.txt
  timeout 300 python bench.py --conv-mode bf16 --steps-only --steps 20 --warmup 5 --tune bf16_sched=$S 2>/dev/null | tee gpurun_out/r3b/bench_bf16_sched$S.json | cut -c1-300
done
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3b/bench.json 2> gpurun_out/r3b/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r3b/bench.json') if l.startswith('{')][-1])
r = d['roofline']
print('bench', d['ms_per_step'], d['ms_per_step_median'], r['kernel'], 'in-step frac', r['frac'], 'isolated', (r.get('isolated') or {}).get('frac'),
      'hw', d.get('step_executed_gflop'), d.get('step_hw_frac'), 'profiled ms', d.get('profiled_ms_per_step'), 'alt_bf16', d.get('alt_bf16', {}).get('ms_per_step'))
for k, v in r['per_shape'].items():
  print('   ', k, v)
for g in r['by_kernel']:
  print('  ', g['kernel'], g['launches_per_step'], g['ms_per_step'], g.get('frac'), (g.get('isolated') or {}).get('frac'))
PY
