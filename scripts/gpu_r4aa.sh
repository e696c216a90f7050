#!/bin/bash
cd $GRAFT_REPO_ROOT
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
python -m pytest tests/test_gpu_fft_conv.py -m gpu -x -q -k "hands_its_frames or matches_oracle or shape_switching" 2>&1 | grep -E "passed|failed|Error|assert|^E " | tail -12
for t in 0 1 0 1; do echo "no_fused=$t: fp32 $(python bench.py --steps-only --steps 100 --tune no_fused_transforms=$t 2>/dev/null | ms)"; done
bash scripts/gpu_timeline.sh r4aa > /dev/null; sed -n 1,40p gpurun_out/r4aa/step_timeline.txt | cut -c1-100
