#!/bin/bash
# second GPU pass of the round: new tests (RCCL C-ABI, beam search), decode bench, RCCL-transport bench
set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q -k "beam or rccl" 2>&1 | tail -15
python scripts/bench_decode.py 2>&1 | tail -3 | tee gpurun_out/decode.log
python bench.py --steps 5 --warmup 2 --force-allreduce --allreduce rccl --no-alt --no-cpu-baseline 2>&1 | grep '^{' | tee gpurun_out/bench_rccl.log
