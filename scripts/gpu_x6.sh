#!/bin/bash
mkdir -p gpurun_out
for env in "ST_BF16_CFG256=642" "ST_BF16_CFG256=3249"; do
  echo "$env: $(env $env timeout 300 python bench.py --steps 10 --warmup 3 --conv-mode bf16 --no-alt --no-cpu-baseline 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_avg_loss'])")"
done
ST_BF16_CFG256=3249 timeout 600 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q 2>&1 | tail -2
ST_CONV_MODE=bf16x6 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -2
