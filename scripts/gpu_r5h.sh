#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_api.py -q -m gpu -x -k "graph or protocol" 2>&1 | tail -15
for M in bf16 fp32; do for G in "" "--graph"; do timeout 300 python scripts/bench_api_train.py --conv-mode $M $G 2>/dev/null | grep '^{'; done; done
