# session 2, call 9: input hand-over ahead of the step -- tests, A/B step time (same box), API loop, inference
mkdir -p gpurun_out/s2c9
timeout 2400 python -m pytest tests/test_gpu_fft_conv.py tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_fullsize.py tests/test_gpu_config2.py tests/test_gpu_dp4.py -q -m gpu -x 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -12 > gpurun_out/s2c9/pytest.log
cat gpurun_out/s2c9/pytest.log
for M in fp32 bf16; do
  for V in 1 0 1 0 1 0; do
    echo "mode $M prefetch=$V: $(ST_PREFETCH_INPUT=$V python bench.py --steps-only --steps 40 --warmup 8 --conv-mode $M 2>/dev/null | tail -1 | cut -c150-260)" | tee -a gpurun_out/s2c9/ab.txt
  done
done
for V in 1 0; do
  ST_PREFETCH_INPUT=$V timeout 300 python scripts/bench_api_train.py --conv-mode fp32 2>/dev/null | grep '^{' | cut -c1-260 | tee -a gpurun_out/s2c9/ab.txt
  ST_PREFETCH_INPUT=$V timeout 300 python scripts/bench_api_train.py --conv-mode bf16 2>/dev/null | grep '^{' | cut -c1-260 | tee -a gpurun_out/s2c9/ab.txt
  ST_PREFETCH_INPUT=$V timeout 600 python scripts/bench_inference.py 2>/dev/null | cut -c1-400 | tee -a gpurun_out/s2c9/ab.txt
done
bash scripts/gpu_timeline.sh s2c9_tl > /dev/null 2>&1
cp gpurun_out/s2c9_tl/step_timeline.txt gpurun_out/s2c9/step_timeline_fp32.txt
head -12 gpurun_out/s2c9/step_timeline_fp32.txt | cut -c1-100; tail -3 gpurun_out/s2c9/step_timeline_fp32.txt
