#!/bin/bash
# round 4: the top 1-tap layers per half-batch (CTC under the other half's GEMMs)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4e
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_grads.py tests/test_gpu_fullsize.py tests/test_gpu_api.py tests/test_gpu_dp4.py -q -m gpu -x 2>&1 | grep -v '^  File "/usr' | tail -30 > $OUT/pytest.log
tail -12 $OUT/pytest.log
for E in 1 0 1 0; do
  echo "== ST_PIPELINE_CTC=$E"
  ST_PIPELINE_CTC=$E timeout 200 python bench.py --steps-only --steps 40 --warmup 5 2>/dev/null | cut -c1-260 | tee -a $OUT/bench.txt
done
bash scripts/gpu_prof.sh r4e_prof python bench.py --steps-only --steps 12 --warmup 4 | head -30 > $OUT/kernel_top.txt
python scripts/step_timeline.py $(find gpurun_out/r4e_prof -name '*kernel_trace.csv' | head -1) > $OUT/step_timeline.txt 2>/dev/null
rm -rf gpurun_out/r4e_prof
sed -n '/idft_rows_kernel<1, 24>/,/dft_rows_kernel<3>/p' $OUT/step_timeline.txt | head -50
