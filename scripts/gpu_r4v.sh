#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scripts/gpu_prof.sh r4v_default python bench.py --no-cpu-baseline | grep -E "gemm_nn_bf16|split|transpose" | cut -c1-60,108-175
echo ---
bash scripts/gpu_prof.sh r4v_x6 python bench.py --steps-only --steps 25 --warmup 5 --conv-mode bf16x6 | grep -E "gemm_nn_bf16|split|transpose" | cut -c1-60,108-175
grep '^{' gpurun_out/r4v_default/stdout.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:v.get('ms_per_step') for k,v in d.items() if k.startswith('alt_')})"
rm -rf gpurun_out/r4v_default gpurun_out/r4v_x6
