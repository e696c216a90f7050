#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters, as MI355X_MICROARCH.md prescribes:
# separate --pmc passes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2), kernel-trace only, no
# sys/hip tracing.  FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes
# of wide (16 B/lane) coalesced reads, so the read side is doubled.  Writes profiles/traffic.json.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic
mkdir -p $OUT
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$C.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json, os
root = os.environ['GRAFT_REPO_ROOT']
vals = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob(os.path.join(root, 'gpurun_out/traffic', c, '**', '*counter_collection.csv'), recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if 'gemm_nn_kernel<128, 128, 2, 2, 0, true>' in r['Kernel_Name'] and r['Counter_Name'] == c]
    vals[c] = (sum(float(r['Counter_Value']) for r in rows) / len(rows), len(rows))
fetch = vals['FETCH_SIZE'][0] * 1024 * 2      # gfx950 correction (MI355X_MICROARCH.md, HBM section)
write = vals['WRITE_SIZE'][0] * 1024
out = dict(kernel='gemm_nn_kernel<128,128,2,2,0,true>', launches=vals['FETCH_SIZE'][1],
           fetch_bytes_per_launch=fetch, write_bytes_per_launch=write, bytes_per_launch=fetch + write,
           note='FETCH_SIZE x 1024 x 2 (gfx950 wide-read correction) + WRITE_SIZE x 1024, averaged over all launches of the symbol')
json.dump(out, open(os.path.join(root, 'gpurun_out/traffic/traffic.json'), 'w'), indent=1)
print(json.dumps(out))
PY
