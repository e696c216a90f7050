#!/bin/bash
# round 4, second GPU call: stream-K v2 (static map, epoch flags, 2 or 3 workgroups per CU), double log-softmax in CTC
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4b
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fft_conv.py tests/test_gpu_parity.py tests/test_gpu_api.py -q -m gpu -x 2>&1 | grep -v '^  File "/usr' | tail -40 > $OUT/pytest.log
tail -8 $OUT/pytest.log
timeout 120 python scripts/diag_ctc_loss.py 2>&1 | grep -v amdgpu.ids | tee $OUT/diag_ctc.txt
timeout 120 python scripts/bench_ctc.py 2>&1 | tail -1 | tee $OUT/ctc.txt
for S in 64 96; do
echo "== batched products, stream-K forced, slots $S"; timeout 200 python scripts/bench_gemm_batched.py --tune streamk=1 --tune streamk_slots=$S 2>&1 | grep -v amdgpu.ids | grep -v "^tn" | tee $OUT/gemm_sk$S.txt
done
for T in "streamk=0" "streamk_slots=96" "streamk=2" "streamk=0" "streamk_slots=96" "streamk=2"; do
  echo "== bench $T"
  timeout 200 python bench.py --steps-only --steps 40 --warmup 5 --tune $T 2>/dev/null | cut -c1-260 | tee -a $OUT/bench_ab.txt
done
