#!/bin/bash
cd $GRAFT_REPO_ROOT
ms() { grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; }
timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_configs.py -q -m gpu -x -k "bf16 or config3" 2>&1 | tail -3
for rep in 1 2 3; do echo "bf16: $(timeout 200 python bench.py --steps-only --steps 100 --conv-mode bf16 2>/dev/null | ms)"; done
bash scripts/gpu_prof.sh r5l_prof_bf16 python bench.py --steps-only --steps 20 --warmup 5 --conv-mode bf16 | grep "finish\|colsum\|wgrad_tr"
rm -rf gpurun_out/r5l_prof_bf16
