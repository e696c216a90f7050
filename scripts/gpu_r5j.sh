#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fft_conv.py tests/test_gpu_configs.py tests/test_gpu_bf16.py tests/test_gpu_api.py -q -m gpu -x -k "bf16 or config3 or graph" 2>&1 | tail -4
