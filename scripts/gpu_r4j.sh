#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4j
mkdir -p $OUT
for E in 1 0 1 0; do
  echo "== ST_FFT_X3=$E"
  ST_FFT_X3=$E timeout 200 python bench.py --steps-only --steps 40 --warmup 5 2>/dev/null | cut -c1-260 | tee -a $OUT/bench_x3.txt
done
ST_FFT_X3=1 timeout 900 python -m pytest tests/test_gpu_fullsize_grads.py -q -m gpu -x -k "frequency_domain" -s 2>&1 | grep -v '^  File "/usr' | tail -12
ST_FFT_X3=1 bash scripts/gpu_prof.sh r4j_prof python bench.py --steps-only --steps 12 --warmup 4 | head -16 | cut -c1-190
rm -rf gpurun_out/r4j_prof
