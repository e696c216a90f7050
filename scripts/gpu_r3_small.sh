#!/bin/bash
# the classification layer's filter gradient beside its back-prop to the input (ST_WGRAD_SIDE_TOP), both arithmetics
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3s
mkdir -p $OUT
for V in 1 0 1 0; do
  echo "== ST_WGRAD_SIDE_TOP=$V"
  ST_WGRAD_SIDE_TOP=$V timeout 200 python bench.py --steps-only --steps 40 --warmup 5 2>/dev/null | cut -c150-260
  ST_WGRAD_SIDE_TOP=$V timeout 200 python bench.py --conv-mode bf16 --steps-only --steps 40 --warmup 5 2>/dev/null | cut -c150-260
done
timeout 900 python -m pytest tests/test_gpu_fullsize_grads.py tests/test_gpu_bf16.py tests/test_gpu_configs.py tests/test_gpu_api.py tests/test_gpu_dp4.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5 | tee $OUT/pytest_top.log
