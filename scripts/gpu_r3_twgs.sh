#!/bin/bash
cd $GRAFT_REPO_ROOT
for W in 0 256 384 768 1024; do
  echo "== transform_wgs=$W"
  timeout 300 python bench.py --steps-only --steps 30 --warmup 5 --tune transform_wgs=$W 2>/dev/null | cut -c100-260
done
