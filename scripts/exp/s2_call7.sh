# session 2, call 7: cached shape descriptions -- tests, then A/B of variable-length training (same box)
mkdir -p gpurun_out/s2c7
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fft_conv.py tests/test_gpu_api.py tests/test_gpu_bf16.py tests/test_gpu_config2.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -12 > gpurun_out/s2c7/pytest.log
cat gpurun_out/s2c7/pytest.log
for M in bf16 fp32; do
  for V in 1 0 1 0; do
    echo "mode $M shape_cache=$V" | tee -a gpurun_out/s2c7/ab.txt
    ST_SHAPE_CACHE=$V python scripts/bench_varlen_train.py --batch 32 --mels 80 --orders arrival bucketed --conv-mode $M --out gpurun_out/s2c7/varlen_${M}_$V.json 2>/dev/null | tail -2 | cut -c1-420 | tee -a gpurun_out/s2c7/ab.txt
  done
done
