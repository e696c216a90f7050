#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --no-alt --no-cpu-baseline 2>/dev/null | cut -c1-300
ST_SHARE_GPU=1 ST_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 --batch 4 --seconds 2 --no-alt --no-cpu-baseline 2>/dev/null | cut -c1-400
