#!/bin/bash
cd $GRAFT_REPO_ROOT
python scripts/bench_ctc.py 2>&1 | grep ctc_loss_grad | cut -c1-150
python scripts/bench_ctc.py --frames 1501 --labels 400 --batch 16 2>&1 | grep ctc_loss_grad | cut -c1-150
bash scripts/gpu_prof.sh r3_ctc_old python scripts/bench_ctc.py > /dev/null 2>&1
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r3_ctc_old/r3_ctc_old_kernel_stats.csv')):
    if 'ctc' in r['Name']: print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3, 1))
PY
