#!/bin/bash
# Fabric traffic of the mel feature kernels (FETCH_SIZE / WRITE_SIZE, one rocprofv3 pass each, gfx950 read correction
# as in gpu_traffic.sh) for 32 x 10 s clips -> gpurun_out/mel_traffic/mel_traffic.json
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=$ROOT/gpurun_out/mel_traffic
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o t -- python $ROOT/scripts/bench_mel.py 80 > $OUT/$C.log 2>&1
done
cd $ROOT
python - "$OUT" <<'PY'
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
alg = 32 * (160000 * 4 + 1001 * 80 * 4)
res['_algorithmic_bytes_per_batch'] = alg
res['_note'] = 'FETCH_SIZE x 1024 x 2 (gfx950 wide-read correction) and WRITE_SIZE x 1024, averaged per launch; 32 clips of 10 s, 80 mels'
json.dump(res, open(os.path.join(out, 'mel_traffic.json'), 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
