#!/bin/bash
# round 4: the default bench line (parity incl. the absolute CTC bound, roofline by kernel, world-1 comm probe) + dp self-launch tests
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4g
mkdir -p $OUT
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'EOF'
import json
d = json.load(open('gpurun_out/r4g/bench.json'))
print({k: d[k] for k in ('value', 'ms_per_step', 'ms_per_step_median', 'ctc_loss_delta', 'max_logit_err', 'step_hw_frac', 'step_executed_gflop')})
print(d['parity']['ctc_loss_delta_fp32_output'], d['parity']['passed'])
r = d['roofline']
print('dominant', r['kernel'], r['frac'], r['ms_per_step'], r.get('traffic'))
for g in r['by_kernel']:
  print('  %-40s launches %5.1f ms %.3f frac %s iso %s' % (g['kernel'], g['launches_per_step'], g['ms_per_step'], g.get('frac'), (g.get('isolated') or {}).get('frac')))
print(d.get('comm_probe_world1'))
print(d['alt_bf16']['ms_per_step'], d['cpu_baseline']['value'])
EOF
tail -3 $OUT/bench.err
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_dp4.py -q -m gpu -x 2>&1 | grep -v '^  File "/usr' | tail -8
