#!/bin/bash
# round 4: back-prop's inverse transform as two overlap-add passes over whole windows
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4c
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fft_conv.py tests/test_gpu_fullsize_grads.py -q -m gpu -x 2>&1 | grep -v '^  File "/usr' | tail -30 > $OUT/pytest.log
tail -8 $OUT/pytest.log
for i in 1 2; do
  timeout 200 python bench.py --steps-only --steps 40 --warmup 5 2>/dev/null | cut -c1-260 | tee -a $OUT/bench.txt
done
bash scripts/gpu_prof.sh r4c_prof python bench.py --steps-only --steps 12 --warmup 4 | head -30 > $OUT/kernel_top.txt
python scripts/step_timeline.py $(find gpurun_out/r4c_prof -name '*kernel_trace.csv' | head -1) > $OUT/step_timeline.txt 2>/dev/null
rm -rf gpurun_out/r4c_prof
cat $OUT/kernel_top.txt | cut -c1-200; tail -75 $OUT/step_timeline.txt
