mkdir -p gpurun_out/c4
for M in fp32 bf16; do python scripts/exp/head_bubble.py --conv-mode $M 2>/dev/null | grep '^{' | tee gpurun_out/c4/head_bubble_$M.json; done
python scripts/exp/ctc_mask_probe.py 2>gpurun_out/c4/ctc_mask.err | grep '^{' | tee gpurun_out/c4/ctc_mask_probe.json
tail -3 gpurun_out/c4/ctc_mask.err
python scripts/bench_varlen_train.py --sweep --out gpurun_out/c4/sweep_fp32.json 2>gpurun_out/c4/sweep.err | tail -20
tail -3 gpurun_out/c4/sweep.err
python scripts/bench_varlen_train.py --sweep --conv-mode bf16 --out gpurun_out/c4/sweep_bf16.json 2>/dev/null | tail -20
