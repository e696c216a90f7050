#!/bin/bash
# Fabric traffic of the mel feature kernels (FETCH_SIZE / WRITE_SIZE, one rocprofv3 pass each, gfx950 read correction
# as in gpu_traffic.sh) for B x 10 s clips (B = $1, default 32) -> gpurun_out/mel_traffic_b$B/mel_traffic.json
export TMPDIR=/tmp
B=${1:-32}
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=$ROOT/gpurun_out/mel_traffic_b$B
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o t -- python $ROOT/scripts/bench_mel.py 80 $B > $OUT/$C.log 2>&1
done
grep -h "^{" $OUT/WRITE_SIZE.log | tail -1 > $OUT/timing_under_rocprof.txt
cd $ROOT
BATCH=$B python - "$OUT" <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob(os.path.join(out, c, '**', '*counter_collection.csv'), recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'mel_' in r['Kernel_Name'] and r['Counter_Name'] == c:
            import re
            name = re.search(r'(mel_\w+)', r['Kernel_Name']).group(1)
            agg[name].append(float(r['Counter_Value']) * 1024 * (2 if c == 'FETCH_SIZE' else 1))
    for k, v in agg.items():
        res[k][c.lower() + '_bytes_per_launch'] = sum(v) / len(v)
        res[k]['launches'] = len(v)
batch = int(os.environ['BATCH'])
alg = batch * (160000 * 4 + 1001 * 80 * 4)
res['_algorithmic_bytes_per_batch'] = alg
res['_fabric_bytes_per_batch'] = sum(v.get('fetch_size_bytes_per_launch', 0) + v.get('write_size_bytes_per_launch', 0) for k, v in res.items() if isinstance(v, dict))
res['_fabric_over_algorithmic'] = res['_fabric_bytes_per_batch'] / alg
sys.path.insert(0, os.getcwd())
from speecht_amd.build import source_digest
res['_source_digest'] = source_digest()
res['_note'] = 'FETCH_SIZE x 1024 x 2 (gfx950 wide-read correction) and WRITE_SIZE x 1024, averaged per launch; %d clips of 10 s, 80 mels' % batch
json.dump(res, open(os.path.join(out, 'mel_traffic.json'), 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
