#!/bin/bash
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q -k "mfcc or cli or beam" 2>&1 | tail -15
