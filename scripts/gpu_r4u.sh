#!/bin/bash
cd $GRAFT_REPO_ROOT
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('ms_per_step_median'), {k:v.get('ms_per_step') for k,v in d.items() if k.startswith('alt_')})"; }
echo "fp32 steps-only: $(python bench.py --steps-only --steps 100 2>/dev/null | ms)"
echo "x6 steps-only: $(python bench.py --steps-only --steps 100 --conv-mode bf16x6 2>/dev/null | ms)"
echo "bf16 steps-only: $(python bench.py --steps-only --steps 100 --conv-mode bf16 2>/dev/null | ms)"
echo "default: $(python bench.py --no-cpu-baseline 2>/dev/null | ms)"
python scripts/exp/alt_timing_probe.py 2>/dev/null | tail -1
echo "fp32 steps-only: $(python bench.py --steps-only --steps 100 2>/dev/null | ms)"
