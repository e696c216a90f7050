#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r3c/pytest_bf16.log
cat gpurun_out/r3c/pytest_bf16.log
for S in 1 2 4; do
  echo "== bf16_sched=$S"
  timeout 300 python scripts/bench_conv_bf16.py --layers 8,9 --tune bf16_sched=$S 2>&1 | tail -4 | tee gpurun_out/r3c/conv_bf16_sched$S.txt
  timeout 300 python bench.py --conv-mode bf16 --steps-only --steps 20 --warmup 5 --tune bf16_sched=$S 2>/dev/null | tee gpurun_out/r3c/bench_bf16_sched$S.json | cut -c1-300
done
bash scripts/gpu_pmc_shapes.sh bf16_w4k64 --conv-mode bf16 --tune bf16_sched=4 2>&1 | grep "gemm_nn_bf16_kernel<256" | tee gpurun_out/r3c/pmc_w4k64.txt
