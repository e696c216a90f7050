#!/bin/bash
# usage: gpu_quick.sh <tag> [pytest targets...]: the named GPU tests, then the fp32 step (two runs) and its timeline
cd $GRAFT_REPO_ROOT
TAG=$1; shift
mkdir -p gpurun_out/$TAG
if [ $# -gt 0 ]; then python -m pytest "$@" -m gpu -x -q 2>&1 | tail -4; fi
for i in 1 2; do python bench.py --steps-only --steps 40 2>/dev/null | tail -1 | cut -c100-260; done
bash scripts/gpu_timeline.sh $TAG > /dev/null
grep -E "idft_rows|dft_rows" gpurun_out/$TAG/kernel_top.txt | cut -c1-60,110-170
tail -1 gpurun_out/$TAG/step_timeline.txt
