mkdir -p gpurun_out/c1
python bench.py --steps-only --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/c1/steps_only_fp32.json
python scripts/bench_varlen_train.py --batch 32 --mels 80 --out gpurun_out/c1/varlen_fp32_b32_m80.json 2>gpurun_out/c1/err1.log | tail -3
python scripts/bench_varlen_train.py --batch 64 --mels 128 --rate 22050 --steps 40 --out gpurun_out/c1/varlen_fp32_b64_m128_22k.json 2>gpurun_out/c1/err2.log | tail -3
python scripts/bench_varlen_train.py --batch 32 --mels 80 --conv-mode bf16 --out gpurun_out/c1/varlen_bf16_b32_m80.json 2>gpurun_out/c1/err3.log | tail -3
cat gpurun_out/c1/steps_only_fp32.json
tail -3 gpurun_out/c1/err1.log
