# session 2, call 4: half-tile row split of the batched products, L10 slices on short batches; head bubble by length; sweep
mkdir -p gpurun_out/s2c4
timeout 1500 python -m pytest tests/test_gpu_fft_conv.py tests/test_gpu_parity.py tests/test_gpu_config2.py -q -m gpu -x 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -8 > gpurun_out/s2c4/pytest.log
cat gpurun_out/s2c4/pytest.log
for F in 201 501 1001 1101; do python scripts/exp/head_bubble.py --frames $F 2>/dev/null | grep '^{' | cut -c1-200 | tee -a gpurun_out/s2c4/head_bubble.txt; done
python scripts/bench_varlen_train.py --sweep 201 301 501 601 1001 1101 1201 1501 --out gpurun_out/s2c4/sweep_fp32.json 2>/dev/null | tail -10 | cut -c1-160
python scripts/bench_varlen_train.py --batch 32 --mels 80 --out gpurun_out/s2c4/varlen_fp32_b32_m80.json 2>/dev/null | tail -3 | cut -c1-300
timeout 600 python scripts/bench_inference.py 2>/dev/null | cut -c1-600 > gpurun_out/s2c4/inference_fp32.json; cut -c1-400 gpurun_out/s2c4/inference_fp32.json
