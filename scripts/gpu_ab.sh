#!/bin/bash
# A/B of an environment switch on the default bench: gpu_ab.sh VAR v1 v2 ... (three rounds, interleaved)
cd $GRAFT_REPO_ROOT
VAR=$1; shift
for rep in 1 2 3; do
  for v in "$@"; do
    env $VAR=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt $BENCH_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$VAR=$v', d['ms_per_step'], d.get('ms_per_step_median'), d['value'], d.get('ms_each_step', [])[:4], d.get('host_ms_each_step', [])[:4])"
  done
done
