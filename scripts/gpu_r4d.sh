#!/bin/bash
# round 4: transposed-operand back-prop (no gbwd, no flips)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4d
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_fft_conv.py tests/test_gpu_fullsize_grads.py tests/test_gpu_parity.py tests/test_gpu_api.py -q -m gpu -x 2>&1 | grep -v '^  File "/usr' | tail -30 > $OUT/pytest.log
tail -12 $OUT/pytest.log
for i in 1 2; do
  timeout 200 python bench.py --steps-only --steps 40 --warmup 5 2>/dev/null | cut -c1-260 | tee -a $OUT/bench.txt
done
bash scripts/gpu_prof.sh r4d_prof python bench.py --steps-only --steps 12 --warmup 4 | head -30 > $OUT/kernel_top.txt
python scripts/step_timeline.py $(find gpurun_out/r4d_prof -name '*kernel_trace.csv' | head -1) > $OUT/step_timeline.txt 2>/dev/null
rm -rf gpurun_out/r4d_prof
cut -c1-180 $OUT/kernel_top.txt | head -12; sed -n '/gemm_nn_kernel<128, 32/,/gemm_tn_kernel<128, 2, 2>/p' $OUT/step_timeline.txt | head -40
