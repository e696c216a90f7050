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
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-alt --no-cpu-baseline > $OUT/$C.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import collections, csv, glob, json, os, re
root = os.environ['GRAFT_REPO_ROOT']


def bench_key(name):
    """rocprof symbol -> the key bench.py's roofline uses (the library's launch-trace name)."""
    m = re.search(r'gemm_nn_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (true|false)>', name)
    if m:
        return 'gemm_nn<%s,%s,%s,%s,%s> epi=%s' % (m.group(1), m.group(2), m.group(3), m.group(4),
                                                     'fast' if m.group(6) == 'true' else 'clamped', m.group(5))
    m = re.search(r'gemm_tn_kernel<(\d+),', name)
    return 'gemm_tn<%s>' % m.group(1) if m else None


vals = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob(os.path.join(root, 'gpurun_out/traffic', c, '**', '*counter_collection.csv'), recursive=True)
    for r in csv.DictReader(open(f[0])):
        k = bench_key(r['Kernel_Name'])
        if k and r['Counter_Name'] == c:
            vals[k][c][0] += float(r['Counter_Value'])
            vals[k][c][1] += 1
by = {}
for k, d in vals.items():
    fetch = d['FETCH_SIZE'][0] / max(d['FETCH_SIZE'][1], 1) * 1024 * 2      # gfx950 correction (MI355X_MICROARCH.md, HBM section)
    write = d['WRITE_SIZE'][0] / max(d['WRITE_SIZE'][1], 1) * 1024
    by[k] = dict(launches=d['FETCH_SIZE'][1], fetch_bytes_per_launch=fetch, write_bytes_per_launch=write, bytes_per_launch=fetch + write)
out = dict(by_kernel=by, command='bench.py --steps 2 --warmup 1 --no-alt --no-cpu-baseline',
           note='FETCH_SIZE x 1024 x 2 (gfx950 wide-read correction) + WRITE_SIZE x 1024, averaged over all launches of the symbol '
                '(the training steps and the roofline-measurement launches of the same command: the same launch mix)')
json.dump(out, open(os.path.join(root, 'gpurun_out/traffic/traffic.json'), 'w'), indent=1)
print(json.dumps(out))
PY
