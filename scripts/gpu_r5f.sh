#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f; mkdir -p $O
ms() { grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; }
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -40 > $O/pytest_gpu.log; tail -12 $O/pytest_gpu.log
for B in 32 512; do python scripts/bench_mel.py 80 $B 2>/dev/null | tail -1; done | tee $O/mel_timing.txt
for M in fp32 bf16 bf16x6; do echo "$M: $(timeout 200 python bench.py --steps-only --steps 60 --conv-mode $M 2>/dev/null | ms)"; done | tee $O/steps.txt
