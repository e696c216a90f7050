#!/bin/bash
mkdir -p gpurun_out
for cfg in 642 6424 3244 324 323; do
  echo "cfg256 $cfg: $(ST_BF16_CFG256=$cfg python bench.py --steps 10 --warmup 3 --conv-mode bf16 --no-alt --no-cpu-baseline 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_avg_loss'])")"
done
ST_BF16_CFG256=6424 timeout 600 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q 2>&1 | tail -2
