#!/bin/bash
# Fabric traffic of the bf16-activation step (configs[3] arithmetic): FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc passes
# (counters + kernel trace only) over `bench.py --steps-only --conv-mode bf16`, read side doubled (gfx950: FETCH_SIZE tallies
# 128-byte requests at 64 B, MI355X_MICROARCH.md) -> gpurun_out/traffic_bf16/traffic_bf16.json, stamped with the source digest
# (bench.py quotes it in alt_bf16.roofline only when the digest matches the build it runs).
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=$ROOT/gpurun_out/traffic_bf16
rm -rf $OUT; mkdir -p $OUT
STEPS=6; WARM=2
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o t -- python $ROOT/bench.py --steps-only --steps $STEPS --warmup $WARM --conv-mode bf16 > $OUT/pmc_$C.log 2>&1
done
cd $ROOT
OUT=$OUT STEPS=$((STEPS + WARM)) python - <<'PY'
import collections, csv, glob, json, os, re, sys
root, out_dir, steps = os.getcwd(), os.environ['OUT'], int(os.environ['STEPS'])
sys.path.insert(0, root)
from speecht_amd.build import source_digest
vals = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob(os.path.join(out_dir, 'pmc_' + c, '**', '*counter_collection.csv'), recursive=True)
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name'] == c:
            m = re.search(r'(\w+_kernel(<[^>]*>)?)', r['Kernel_Name'])
            k = m.group(1) if m else r['Kernel_Name'][:60]
            vals[k][c][0] += float(r['Counter_Value'])
            vals[k][c][1] += 1
by, step_fetch, step_write, startup = {}, 0.0, 0.0, 0.0
for k, d in vals.items():
    n = max(d['FETCH_SIZE'][1], d['WRITE_SIZE'][1], 1)
    fetch, write = d['FETCH_SIZE'][0] * 1024 * 2, d['WRITE_SIZE'][0] * 1024
    if 'at::native' in k or n < steps:
        startup += fetch + write
    else:
        step_fetch += fetch / steps
        step_write += write / steps
    by[k] = dict(launches_per_step=round(n / steps, 2), bytes_per_launch=(fetch + write) / n, bytes_per_step=(fetch + write) / steps)
res = dict(source_digest=source_digest(), conv_mode='bf16', step_bytes=step_fetch + step_write, step_fetch_bytes=step_fetch,
           step_write_bytes=step_write, steps_profiled=steps, startup_bytes_not_counted=startup, by_kernel=by,
           note='FETCH_SIZE x 1024 x 2 (gfx950 wide-read correction) + WRITE_SIZE x 1024; separate rocprofv3 --pmc passes with --kernel-trace only')
json.dump(res, open(os.path.join(out_dir, 'traffic_bf16.json'), 'w'), indent=1)
print('bf16 step fabric bytes: %.2f GB (fetch %.2f, write %.2f)' % (res['step_bytes'] / 1e9, step_fetch / 1e9, step_write / 1e9))
PY
cp $OUT/traffic_bf16.json $ROOT/profiles/
find $OUT -name '*.csv' -delete
