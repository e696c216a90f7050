#!/bin/bash
cd $GRAFT_REPO_ROOT
ms() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('ms_per_step_median'), {k:v.get('ms_per_step') for k,v in d.items() if k.startswith('alt_')}, d.get('comm_probe_world1',{}).get('with_collective_ms'))"; }
for q in 4 8 16; do echo "GPU_MAX_HW_QUEUES=$q: $(GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline 2>/dev/null | ms)"; done
