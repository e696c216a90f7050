#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4i
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_fft_conv.py tests/test_gpu_bf16.py -q -m gpu -x -k "bf16" 2>&1 | grep -v '^  File "/usr' | tail -6
for E in 1 0 1 0; do
  echo "== ST_FFT_BF16=$E"
  ST_FFT_BF16=$E timeout 200 python bench.py --steps-only --steps 40 --warmup 5 --conv-mode bf16 2>/dev/null | cut -c1-260 | tee -a $OUT/bench_bf16.txt
done
bash scripts/gpu_prof.sh r4i_prof python bench.py --steps-only --steps 12 --warmup 4 --conv-mode bf16 | head -24 | cut -c1-190
rm -rf gpurun_out/r4i_prof
