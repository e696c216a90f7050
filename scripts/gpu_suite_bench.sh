#!/bin/bash
# the whole GPU suite, then the default bench line (what the driver runs at round end)
cd $GRAFT_REPO_ROOT
O=gpurun_out/suite; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12 | tee $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench.err | grep '^{' > $O/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/suite/bench.json'))
r=d['roofline']
print('ms', d['ms_per_step'], 'value', d['value'], 'hw_frac', d.get('step_hw_frac'), 'gflop', d.get('step_executed_gflop'))
print('dom', r['kernel'], r['frac'], r['ms_per_step'])
for g in r['by_kernel'][:12]: print('  ', g['kernel'], g.get('launches_per_step'), g['ms_per_step'], g.get('frac'), g.get('hbm_frac'))
print('bf16', d['alt_bf16']['ms_per_step'], 'x6', d['alt_bf16x6']['ms_per_step'])
print('parity', d['max_logit_err'], d['ctc_loss_delta'])
PY
tail -5 $O/bench.err
