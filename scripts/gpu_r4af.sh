#!/bin/bash
cd $GRAFT_REPO_ROOT
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
python bench.py --steps-only --steps 40 > /dev/null 2>&1
for g in 0 1 2 4 8 0; do echo "xcd_gm=$g: fp32 $(python bench.py --steps-only --steps 100 --tune xcd_gm=$g 2>/dev/null | ms)"; done
