#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -40 > $O/pytest_gpu.log; tail -12 $O/pytest_gpu.log
S0=$SECONDS; timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall $((SECONDS - S0)) s"; tail -5 $O/bench.err
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r5d/bench.json') if x.startswith('{')]
if l:
  b=json.loads(l[-1])
  print('ms', b['ms_per_step'], 'alt_bf16', b.get('alt_bf16',{}).get('ms_per_step'), 'x6', b.get('alt_bf16x6',{}).get('ms_per_step'))
  print('bf16 roofline', json.dumps(b.get('alt_bf16',{}).get('roofline'))[:1500])
  print('comm fp32', json.dumps(b.get('comm_model_8gpu'))[:1200])
  print('comm bf16', json.dumps(b.get('alt_bf16',{}).get('comm_model_8gpu'))[:1200])
  print('c2', json.dumps(b.get('configs2_inference')))
  print('c4', json.dumps(b.get('configs4_decode')))
PY
