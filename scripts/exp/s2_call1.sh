# session 2, call 1: split update + pipelined filters_idft -- parity tests, A/B step times, timeline
mkdir -p gpurun_out/s2c1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "update_in_two_pieces or fifty_shapes or conv_fwd_bwd" 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -8 > gpurun_out/s2c1/pytest_a.log
cat gpurun_out/s2c1/pytest_a.log
timeout 1200 python -m pytest tests/test_gpu_fft_conv.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_grads.py tests/test_gpu_api.py -q -m gpu -x 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -8 > gpurun_out/s2c1/pytest_b.log
cat gpurun_out/s2c1/pytest_b.log
for M in fp32 bf16; do
  for S in 0 1 0 1; do
    echo "mode $M split $S: $(ST_SPLIT_UPDATE=$S python bench.py --steps-only --steps 40 --warmup 8 --conv-mode $M 2>/dev/null | tail -1)" | tee -a gpurun_out/s2c1/ab.txt
  done
done
bash scripts/gpu_timeline.sh s2c1_tl > /dev/null 2>&1
cp gpurun_out/s2c1_tl/step_timeline.txt gpurun_out/s2c1/step_timeline_fp32.txt
bash scripts/gpu_timeline.sh s2c1_tlb --conv-mode bf16 > /dev/null 2>&1
cp gpurun_out/s2c1_tlb/step_timeline.txt gpurun_out/s2c1/step_timeline_bf16.txt
head -30 gpurun_out/s2c1/step_timeline_fp32.txt | cut -c1-100
tail -22 gpurun_out/s2c1/step_timeline_fp32.txt | cut -c1-100
