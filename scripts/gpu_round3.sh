#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k beam 2>&1 | tail -5
python scripts/bench_decode.py 2>&1 | grep '^{' | tee gpurun_out/decode.json
