#!/bin/bash
# usage: gpu_check.sh "<pytest -k expression or empty>" [bench args...]: GPU tests, then a short bench line
mkdir -p gpurun_out
K="$1"; shift
if [ -n "$K" ]; then timeout 1200 python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | grep -E "passed|failed|rror" | tail -3
else timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3; fi
python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>&1 | grep '^{' | tee gpurun_out/bench_check.json | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('headline', d['ms_per_step'], d['value'], d['dtype'])
for k in ('alt_bf16x6','alt_bf16'):
    if k in d: print(k, d[k]['ms_per_step'], d[k]['value'])"
