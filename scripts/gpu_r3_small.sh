#!/bin/bash
# bf16 activations: filter copies after the update with an event per layer (engine._refresh_wb_after_update)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3s
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_configs.py tests/test_gpu_api.py tests/test_gpu_dp4.py -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5 | tee $OUT/pytest_wb.log
