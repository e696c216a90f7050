#!/bin/bash
cd $GRAFT_REPO_ROOT
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('ms_per_step_median'))"; }
echo "bf16 steps-only 400: $(python bench.py --steps-only --steps 400 --conv-mode bf16 2>/dev/null | ms)"
echo "bf16 steps-only 4000: $(python bench.py --steps-only --steps 4000 --conv-mode bf16 2>/dev/null | ms)"
echo "bf16 steps-only 400 again: $(python bench.py --steps-only --steps 400 --conv-mode bf16 2>/dev/null | ms)"
echo "x6 steps-only 2000: $(python bench.py --steps-only --steps 2000 --conv-mode bf16x6 2>/dev/null | ms)"
rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -v "^$" | head -30
