#!/bin/bash
set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q 2>&1 | tail -8
ST_CONV_MODE=bf16x6 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
ST_CONV_MODE=bf16 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bf16 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-alt --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_bf16_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' | tee gpurun_out/bench_alt.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['alt_bf16x6'], d['alt_bf16'])"
