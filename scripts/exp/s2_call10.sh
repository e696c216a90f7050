mkdir -p gpurun_out/s2c10
timeout 900 python -m pytest tests/test_gpu_fft_conv.py -q -m gpu -x -k "handed_over or shape_switching" 2>&1 | tail -2
for M in fp32 bf16; do
  for V in 1 0 1 0 1 0; do
    echo "mode $M prefetch=$V: $(ST_PREFETCH_INPUT=$V python bench.py --steps-only --steps 40 --warmup 8 --conv-mode $M 2>/dev/null | tail -1 | cut -c150-260)" | tee -a gpurun_out/s2c10/ab.txt
  done
done
for V in 1 0 1 0; do
  ST_PREFETCH_INPUT=$V timeout 300 python scripts/bench_api_train.py --conv-mode fp32 2>/dev/null | grep '^{' | cut -c100-200 | tee -a gpurun_out/s2c10/ab.txt
  ST_PREFETCH_INPUT=$V timeout 300 python scripts/bench_api_train.py --conv-mode bf16 2>/dev/null | grep '^{' | cut -c100-200 | tee -a gpurun_out/s2c10/ab.txt
done
bash scripts/gpu_timeline.sh s2c10_tl > /dev/null 2>&1
cp gpurun_out/s2c10_tl/step_timeline.txt gpurun_out/s2c10/step_timeline_fp32.txt
head -14 gpurun_out/s2c10/step_timeline_fp32.txt | cut -c1-100; tail -8 gpurun_out/s2c10/step_timeline_fp32.txt | cut -c1-100
