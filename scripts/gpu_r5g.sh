#!/bin/bash
cd $GRAFT_REPO_ROOT
ms() { grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; }
timeout 600 python -m pytest tests/test_gpu_fft_conv.py -q -m gpu -x -k "stream_k or lost" 2>&1 | tail -3
for rep in 1 2; do for M in fp32 bf16 bf16x6; do echo "$M: $(timeout 200 python bench.py --steps-only --steps 100 --conv-mode $M 2>/dev/null | ms)"; done; done
