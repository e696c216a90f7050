# session 2, call 2: MFMA filters_idft + low-priority side streams
mkdir -p gpurun_out/s2c2
timeout 1500 python -m pytest tests/test_gpu_fft_conv.py tests/test_gpu_fullsize_grads.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -8 > gpurun_out/s2c2/pytest.log
cat gpurun_out/s2c2/pytest.log
run() { echo "$1: $(env $2 python bench.py --steps-only --steps 40 --warmup 8 --conv-mode $3 2>/dev/null | tail -1 | cut -c100-260)" | tee -a gpurun_out/s2c2/ab.txt; }
for M in fp32 bf16; do
  run "$M base(split,mfma-idft)" "ST_X=0" $M
  run "$M nosplit" "ST_SPLIT_UPDATE=0" $M
  run "$M lowprio" "ST_SIDE_PRIORITY=low" $M
  run "$M lowprio+nosplit" "ST_SIDE_PRIORITY=low ST_SPLIT_UPDATE=0" $M
  run "$M base again" "ST_X=0" $M
done
bash scripts/gpu_timeline.sh s2c2_tl > /dev/null 2>&1
cp gpurun_out/s2c2_tl/step_timeline.txt gpurun_out/s2c2/step_timeline_fp32.txt
ST_SIDE_PRIORITY=low bash scripts/gpu_timeline.sh s2c2_tlp > /dev/null 2>&1
cp gpurun_out/s2c2_tlp/step_timeline.txt gpurun_out/s2c2/step_timeline_fp32_lowprio.txt
grep -n "filters_idft\|step " gpurun_out/s2c2/step_timeline_fp32.txt | cut -c1-100
