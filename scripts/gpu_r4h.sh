#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4h
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_dp4.py tests/test_gpu_api.py -q -m gpu -x 2>&1 | grep -v '^  File "/usr' | tail -8
timeout 600 python bench.py --no-cpu-baseline --no-alt > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; wc -l $OUT/bench.json
python - <<'EOF'
import json
d = json.loads(open('gpurun_out/r4h/bench.json').read().splitlines()[0])
print({k: d[k] for k in ('value', 'ms_per_step', 'ms_per_step_median', 'ctc_loss_delta')})
print(d.get('comm_probe_world1'))
EOF
