#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5k; mkdir -p $O
bash scripts/gpu_prof.sh r5k_prof_bf16 python bench.py --steps-only --steps 20 --warmup 5 --conv-mode bf16 | head -14 > $O/kernel_top_bf16.txt
python scripts/step_timeline.py $(find gpurun_out/r5k_prof_bf16 -name '*kernel_trace.csv' | head -1) > $O/step_timeline_bf16.txt 2>/dev/null
rm -rf gpurun_out/r5k_prof_bf16
