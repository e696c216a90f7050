#!/bin/bash
# bf16-activation step: the three ways bench.py times it, on one box
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('default fp32', d['ms_per_step'], 'alt_bf16', d['alt_bf16']['ms_per_step'])"
  timeout 200 python bench.py --conv-mode bf16 --no-cpu-baseline --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('main loop bf16', d['ms_per_step'], d.get('ms_per_step_median'), d.get('host_enqueue_ms_per_step'))"
  timeout 200 python bench.py --conv-mode bf16 --steps-only --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('steps-only bf16', d['ms_per_step'])"
done
