#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4l
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fft_conv.py tests/test_gpu_fullsize_grads.py -q -m gpu -x 2>&1 | grep -v '^  File "/usr' | tail -8
for i in 1 2; do
  timeout 200 python bench.py --steps-only --steps 40 --warmup 5 2>/dev/null | cut -c1-260 | tee -a $OUT/bench.txt
done
bash scripts/gpu_prof.sh r4l_prof python bench.py --steps-only --steps 12 --warmup 4 | head -14 | cut -c1-190
rm -rf gpurun_out/r4l_prof
