mkdir -p gpurun_out/c6
timeout 1500 python -m pytest tests/test_gpu_fft_conv.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_grads.py tests/test_gpu_config2.py tests/test_gpu_dp4.py -q -m gpu -x 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -15 > gpurun_out/c6/pytest.log
cat gpurun_out/c6/pytest.log
python scripts/bench_varlen_train.py --sweep 401 601 801 1001 1101 1301 --out gpurun_out/c6/sweep_fp32.json 2>/dev/null | tail -8 | cut -c1-160
python scripts/bench_varlen_train.py --batch 32 --mels 80 --out gpurun_out/c6/varlen_fp32_b32_m80.json 2>/dev/null | tail -3 | cut -c1-400
