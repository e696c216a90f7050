#!/bin/bash
# round 3, first GPU call: whole GPU suite (new tests included), default bench line, config-3 and config-5 benches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 1500 python -m pytest tests -q -m gpu -s 2>&1 | tail -120 > gpurun_out/r3a/pytest_gpu.log
tail -25 gpurun_out/r3a/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
tail -3 gpurun_out/r3a/bench.err
python - <<'PY'
import json
try:
  d = json.loads([l for l in open('gpurun_out/r3a/bench.json') if l.startswith('{')][-1])
  r = d['roofline']
  print('bench', d['ms_per_step'], d['ms_per_step_median'], d['value'], r['kernel'], 'in-step frac', r['frac'], 'isolated', (r.get('isolated') or {}).get('frac'),
        'hw', d.get('step_executed_gflop'), d.get('step_hw_frac'), 'profiled ms', d.get('profiled_ms_per_step'), 'alt_bf16', d.get('alt_bf16', {}).get('ms_per_step'))
  for g in r['by_kernel']:
    print('  ', g['kernel'], g['launches_per_step'], g['ms_per_step'], g.get('frac'), (g.get('isolated') or {}).get('frac'))
except Exception as e:
  print('bench line unreadable', e)
PY
timeout 600 python scripts/bench_inference.py > gpurun_out/r3a/inference_fp32.json 2> gpurun_out/r3a/inference_fp32.err
tail -2 gpurun_out/r3a/inference_fp32.err; cat gpurun_out/r3a/inference_fp32.json | cut -c1-1500
timeout 300 python scripts/bench_decode.py > gpurun_out/r3a/decode.json 2> gpurun_out/r3a/decode.err
cat gpurun_out/r3a/decode.json
