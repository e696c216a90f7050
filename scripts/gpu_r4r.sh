#!/bin/bash
cd $GRAFT_REPO_ROOT
ms() { python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for w in 64 128 256 512; do for m in fp32 bf16; do echo "tail wgs $w $m: $(ST_UPDATE_TAIL_WGS=$w python bench.py --steps-only --steps 40 --conv-mode $m 2>/dev/null | ms)"; done; done
for m in fp32 bf16; do echo "unsplit $m: $(ST_SPLIT_UPDATE=0 python bench.py --steps-only --steps 40 --conv-mode $m 2>/dev/null | ms)"; done
