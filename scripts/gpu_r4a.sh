#!/bin/bash
# round 4, first GPU call: the persistent stream-K per-bin products (parity + micro-benchmark + step A/B), the CTC loss pair
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4a
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_fft_conv.py tests/test_gpu_parity.py tests/test_gpu_api.py -q -m gpu -x 2>&1 | tail -25 > $OUT/pytest_a.log
tail -5 $OUT/pytest_a.log
timeout 120 python scripts/diag_ctc_loss.py 2>&1 | grep -v amdgpu.ids | tee $OUT/diag_ctc.txt
timeout 120 python scripts/bench_ctc.py 2>&1 | tail -1 | tee $OUT/ctc.txt
echo "== batched products, stream-K policy"; timeout 200 python scripts/bench_gemm_batched.py 2>&1 | grep -v amdgpu.ids | tee $OUT/gemm_sk.txt
echo "== batched products, streamk=2 (off)"; timeout 200 python scripts/bench_gemm_batched.py --tune streamk=2 2>&1 | grep -v amdgpu.ids | tee $OUT/gemm_plain.txt
for T in 0 2 0 2; do
  echo "== bench streamk=$T"
  timeout 200 python bench.py --steps-only --steps 40 --warmup 5 --tune streamk=$T 2>/dev/null | tee $OUT/bench_sk$T.json | cut -c1-260
done
