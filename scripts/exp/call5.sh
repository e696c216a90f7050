mkdir -p gpurun_out/c5
python scripts/bench_varlen_train.py --sweep --out gpurun_out/c5/sweep_fp32.json 2>gpurun_out/c5/sweep.err | tail -20
tail -3 gpurun_out/c5/sweep.err
python scripts/bench_varlen_train.py --sweep --conv-mode bf16 --out gpurun_out/c5/sweep_bf16.json 2>/dev/null | tail -20
