#!/bin/bash
# PMC pass (counters only; separate from --stats runs): usage gpu_pmc.sh "<counters>" <cmd...>
export TMPDIR=/tmp
CNT="$1"; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT -o pmc -- "$@" > $OUT/run.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, collections, glob, os
f = glob.glob(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/pmc/*counter_collection.csv'))
if not f:
    print('no counter csv'); print(open(os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out/pmc/run.log')).read()[-2000:]); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'gemm' not in k and 'ctc' not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print('   %-32s n=%d mean=%.4g min=%.4g max=%.4g' % (c, len(v), sum(v)/len(v), min(v), max(v)))
PY
