#!/bin/bash
cd $GRAFT_REPO_ROOT
for V in 1 0 1 0; do
  echo "== ST_BWD_OPERANDS_WIDE_FIRST=$V"
  ST_BWD_OPERANDS_WIDE_FIRST=$V timeout 200 python bench.py --steps-only --steps 40 --warmup 5 2>/dev/null | cut -c150-260
done
ST_BWD_OPERANDS_WIDE_FIRST=1 bash scripts/gpu_prof.sh r3_wf1 python bench.py --steps-only --steps 20 --warmup 5 > /dev/null 2>&1
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r3_wf1/r3_wf1_kernel_stats.csv')):
    if 'ctc' in r['Name'] or 'filters_dft_bwd_kernel<32' in r['Name']: print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3, 1))
PY
timeout 600 python -m pytest tests/test_gpu_fullsize_grads.py tests/test_gpu_api.py tests/test_gpu_dp4.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
