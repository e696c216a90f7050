#!/bin/bash
cd $GRAFT_REPO_ROOT
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
python bench.py --steps-only --steps 40 > /dev/null 2>&1
for o in "h2d,side,side2,upload,collective" "h2d,-,-,side,side2,upload" "h2d,-,-,side,-,-,-,side2,upload" "h2d,side,-,-,side2,upload" "side,side2,upload,-,h2d"; do
  echo "$o: fp32 $(ST_STREAM_ORDER=$o python bench.py --steps-only --steps 100 2>/dev/null | ms) bf16 $(ST_STREAM_ORDER=$o python bench.py --steps-only --steps 100 --conv-mode bf16 2>/dev/null | ms) x6 $(ST_STREAM_ORDER=$o python bench.py --steps-only --steps 100 --conv-mode bf16x6 2>/dev/null | ms)"
done
