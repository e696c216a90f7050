#!/bin/bash
cd $GRAFT_REPO_ROOT
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
python -m pytest tests/test_gpu_fft_conv.py -m gpu -x -q -k "reduction_major or matches_oracle or hands_its" 2>&1 | grep -E "passed|failed|Error|assert|^E " | tail -8
for t in 0 2 0 2; do echo "streamk knob $t: fp32 $(python bench.py --steps-only --steps 100 --tune streamk=$t 2>/dev/null | ms)"; done
bash scripts/gpu_timeline.sh r4ac > /dev/null; grep -E "gemm_tn|gemm_nn_bins" gpurun_out/r4ac/kernel_top.txt | cut -c1-78,108-175
sed -n '/idft_rows_kernel<3, 24/,$p' gpurun_out/r4ac/step_timeline.txt | cut -c1-100 | head -22
