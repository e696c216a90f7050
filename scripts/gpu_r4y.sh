#!/bin/bash
cd $GRAFT_REPO_ROOT
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
python -m pytest tests/test_gpu_fft_conv.py -m gpu -x -q -k "persistent" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for t in 64 128 64 128; do echo "tile $t: fp32 $(python bench.py --steps-only --steps 100 --tune streamk_tile=$t 2>/dev/null | ms)"; done
bash scripts/gpu_timeline.sh r4y --tune streamk_tile=128 > /dev/null; grep -E "gemm_nn_bins" gpurun_out/r4y/kernel_top.txt | cut -c1-70,108-175
