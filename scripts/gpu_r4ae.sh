#!/bin/bash
cd $GRAFT_REPO_ROOT
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
python bench.py --steps-only --steps 40 > /dev/null 2>&1
for t in 1 0 1 0 1 0; do echo "two side streams for the narrow filter-gradient chains $t: fp32 $(ST_WGRAD_TWO_SIDES=$t python bench.py --steps-only --steps 150 2>/dev/null | ms)"; done
