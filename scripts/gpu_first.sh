#!/bin/bash
# first GPU pass: parity tests, smoke, short bench, rocprof kernel stats
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -30 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
tail -5 gpurun_out/smoke.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
tail -5 gpurun_out/bench.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o w2l -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof | head; find gpurun_out/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} head -30 {}; tail -3 gpurun_out/rocprof.log
