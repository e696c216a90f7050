#!/bin/bash
# round 4: pipelined beam search (configs[4])
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4f
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "beam or bucketed" 2>&1 | grep -v '^  File "/usr' | tail -15 | tee $OUT/pytest.log
timeout 400 python scripts/bench_decode.py 2>/dev/null | tee $OUT/decode_config5.json | cut -c1-1500
