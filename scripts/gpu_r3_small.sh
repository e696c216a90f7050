#!/bin/bash
# bf16 activations: reduction splits of the filter gradients (policy target = (tile, split) pairs per launch), step level
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3s
mkdir -p $OUT
for T in 512 224 256 192; do
  echo "== bf16_wgrad_target=$T"
  timeout 200 python scripts/bench_conv_bf16.py --layers 0,1,10 --reps 20 --tune bf16_wgrad_target=$T 2>&1 | tail -4 | tee $OUT/conv_target$T.txt
  timeout 200 python bench.py --conv-mode bf16 --steps-only --steps 40 --warmup 5 --tune bf16_wgrad_target=$T 2>/dev/null | tee $OUT/bench_target$T.json | cut -c1-300
done
