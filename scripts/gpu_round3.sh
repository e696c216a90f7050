#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-alt --no-cpu-baseline 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'])"
bash scripts/gpu_traffic.sh 2>&1 | tail -2
