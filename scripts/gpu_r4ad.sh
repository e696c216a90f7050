#!/bin/bash
cd $GRAFT_REPO_ROOT
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for t in 1 0 1 0; do echo "two sides $t: fp32 $(ST_WGRAD_TWO_SIDES=$t python bench.py --steps-only --steps 100 2>/dev/null | ms)  x6 $(ST_WGRAD_TWO_SIDES=$t python bench.py --steps-only --steps 100 --conv-mode bf16x6 2>/dev/null | ms)"; done
for t in 2; do echo "two sides, TN kernel for lag products: $(python bench.py --steps-only --steps 100 --tune streamk=2 2>/dev/null | ms)"; done
bash scripts/gpu_timeline.sh r4ad > /dev/null
sed -n '/idft_rows_kernel<3, 24/,$p' gpurun_out/r4ad/step_timeline.txt | cut -c1-100 | head -40
