#!/bin/bash
cd $GRAFT_REPO_ROOT
for V in 1 0 1 0; do
  echo "== ST_BWD_OPERANDS_EARLY=$V"
  ST_BWD_OPERANDS_EARLY=$V timeout 200 python bench.py --steps-only --steps 40 --warmup 5 2>/dev/null | cut -c150-260
done
