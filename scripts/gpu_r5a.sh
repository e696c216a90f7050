#!/bin/bash
# round 5, call 1: the suite on the round's first changes, the LDS transpose-read probe, host enqueue cost per step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
python -c "from speecht_amd.build import build_library; build_library(force=True, verbose=False)" 2>&1 | tail -3
./scripts/ubench/tr16_probe > gpurun_out/r5a/tr16_probe.txt 2>&1
for M in fp32 bf16; do timeout 300 python scripts/bench_host_cost.py --conv-mode $M 2>/dev/null | grep '^{' > gpurun_out/r5a/host_cost_$M.json; cat gpurun_out/r5a/host_cost_$M.json; done
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -30 > gpurun_out/r5a/pytest_gpu.log
tail -15 gpurun_out/r5a/pytest_gpu.log
