#!/bin/bash
# usage: gpu_timeline.sh <tag> [bench args]: one training step as the GPU ran it (rocprofv3 kernel trace) -> gpurun_out/<tag>/step_timeline.txt
cd $GRAFT_REPO_ROOT
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
bash scripts/gpu_prof.sh ${TAG}_prof python bench.py --steps-only --steps 12 --warmup 4 "$@" | head -30 > $OUT/kernel_top.txt
python scripts/step_timeline.py $(find gpurun_out/${TAG}_prof -name '*kernel_trace.csv' | head -1) > $OUT/step_timeline.txt 2>/dev/null
rm -rf gpurun_out/${TAG}_prof
