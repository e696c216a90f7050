#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_configs.py tests/test_gpu_api.py -q -m gpu -x 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -8
b() { timeout 200 python bench.py --steps-only --steps 60 --warmup 10 --conv-mode bf16 "$@" 2>/dev/null | grep '^{' | sed 's/.*"ms_per_step": \([0-9.]*\).*/  step \1 ms/'; }

