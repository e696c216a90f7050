#!/bin/bash
cd $GRAFT_REPO_ROOT
ms() { python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
python -m pytest tests/test_gpu_fft_conv.py tests/test_gpu_configs.py tests/test_gpu_bf16.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for i in 1 2; do echo "bf16: $(python bench.py --steps-only --steps 40 --conv-mode bf16 2>/dev/null | ms)"; done
bash scripts/gpu_timeline.sh r4s --conv-mode bf16 > /dev/null
sed -n '/dft_rows_kernel<3, true, 1>/,/idft_rows_kernel<3, 24, true>/p' gpurun_out/r4s/step_timeline.txt | cut -c1-100
