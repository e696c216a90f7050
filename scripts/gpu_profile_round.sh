#!/bin/bash
# The measured artefacts profiles/ holds for a round, all from ONE build (stamped with build.source_digest()):
#   (order of execution: 3, 4, then 1, 2 -- the bench line quotes the PMC files of ITS build)
#   1. the default bench line (what the driver runs)                       -> gpurun_out/round/bench.json
#   2. rocprofv3 --kernel-trace --stats of `bench.py --no-alt --no-cpu-baseline` -> kernel_stats.csv, step_timeline.txt
#   3. PMC passes over `bench.py --steps-only` (every launch belongs to a training step), counters + kernel-trace only,
#      FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md: 3 + 2 of the 4 TCC slots), read side doubled
#      (gfx950: FETCH_SIZE tallies 128-B requests at 64 B)                  -> traffic.json (per kernel and per step)
#   4. SQ pass (matrix-pipe busy, wave-time split)                          -> mfma_util.json
# usage: gpu_profile_round.sh [tag]   (outputs under gpurun_out/round<tag>/; copy to profiles/ with the round prefix)
TAG="$1"
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=$ROOT/gpurun_out/round$TAG
mkdir -p $OUT
cd $ROOT
STEPS=6; WARM=2
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o t -- python $ROOT/bench.py --steps-only --steps $STEPS --warmup $WARM > $OUT/pmc_$C.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_SQ -o m -- python $ROOT/bench.py --steps-only --steps $STEPS --warmup $WARM > $OUT/pmc_SQ.log 2>&1
cd $ROOT
OUT=$OUT STEPS=$((STEPS + WARM)) python - <<'PY'
import collections, csv, glob, json, os, re, sys
root, out_dir, steps = os.getcwd(), os.environ['OUT'], int(os.environ['STEPS'])
sys.path.insert(0, root)
from speecht_amd.build import source_digest
digest = source_digest()


def bench_key(name):
    """rocprof symbol -> the key bench.py's roofline uses (the library's launch-trace name)."""
    m = re.search(r'gemm_nn_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (true|false), (true|false)>', name)
    if m:
        return 'gemm_nn<%s,%s,%s,%s,%s> epi=%s' % (m.group(1), m.group(2), m.group(3), m.group(4),
                                                     ('fast-bt' if m.group(7) == 'true' else 'fast') if m.group(6) == 'true' else 'clamped',
                                                     m.group(5))
    m = re.search(r'gemm_nn_bins_kernel<(\d+), (\d+), (\d+), (\d+), \d+, (true|false)>', name)
    if m:
        return 'gemm_nn_bins<%s,%s,%s,%s%s>' % (m.group(1), m.group(2), m.group(3), m.group(4), ',bt' if m.group(5) == 'true' else '')
    m = re.search(r'gemm_tn_kernel<(\d+),', name)
    if m:
        return 'gemm_tn<%s>' % m.group(1)
    m = re.search(r'(\w+_kernel(<[^>]*>)?)', name)
    return m.group(1) if m else name[:60]


vals = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob(os.path.join(out_dir, 'pmc_' + c, '**', '*counter_collection.csv'), recursive=True)
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name'] == c:
            k = bench_key(r['Kernel_Name'])
            vals[k][c][0] += float(r['Counter_Value'])
            vals[k][c][1] += 1
by, step_fetch, step_write, startup = {}, 0.0, 0.0, 0.0
for k, d in vals.items():
    n = max(d['FETCH_SIZE'][1], d['WRITE_SIZE'][1], 1)
    fetch_total = d['FETCH_SIZE'][0] * 1024 * 2          # gfx950 correction (MI355X_MICROARCH.md, HBM section)
    write_total = d['WRITE_SIZE'][0] * 1024
    if 'at::native' in k or n < steps:                   # torch fills / casts of buffer creation, one-off packing: not a step's
        startup += fetch_total + write_total
    else:
        step_fetch += fetch_total / steps
        step_write += write_total / steps
    by[k] = dict(launches=n, launches_per_step=round(n / steps, 2), fetch_bytes_per_launch=fetch_total / n,
                 write_bytes_per_launch=write_total / n, bytes_per_launch=(fetch_total + write_total) / n,
                 bytes_per_step=(fetch_total + write_total) / steps)
traffic = dict(source_digest=digest, by_kernel=by, step_bytes=step_fetch + step_write, step_fetch_bytes=step_fetch,
               step_write_bytes=step_write, steps_profiled=steps, startup_bytes_not_counted=startup,
               command='bench.py --steps-only --steps 6 --warmup 2 (kernels launched at least once per step belong to the 8 training '
                       'steps; torch fill kernels of buffer creation and the one-off weight packing are summed apart)',
               note='FETCH_SIZE x 1024 x 2 (gfx950 wide-read correction) + WRITE_SIZE x 1024; separate rocprofv3 --pmc passes with '
                    '--kernel-trace only; per launch = total of the symbol / its launches, per step = total / steps')
json.dump(traffic, open(os.path.join(out_dir, 'traffic.json'), 'w'), indent=1)
print('step bytes: %.2f GB (fetch %.2f, write %.2f)' % ((step_fetch + step_write) / 1e9, step_fetch / 1e9, step_write / 1e9))

cc = glob.glob(os.path.join(out_dir, 'pmc_SQ', '**', '*counter_collection.csv'), recursive=True)[0]
kt = glob.glob(os.path.join(out_dir, 'pmc_SQ', '**', '*kernel_trace.csv'), recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(cc)):
    name = r['Kernel_Name']
    m = re.search(r'((gemm_\w+|dft_rows_kernel|idft_rows_kernel|ctc_\w+)(<[^>]*>)?)', name)
    if not m:
        continue
    key = m.group(1)
    agg[key][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVE_CYCLES':
        agg[key]['seconds'] += dur.get(r['Dispatch_Id'], 0.0)
        agg[key]['launches'] += 1
util = dict(source_digest=digest)
for k, d in agg.items():
    if d['seconds'] <= 0:
        continue
    simd_cycles = d['seconds'] * 2.4e9 * 1024
    wc = d['SQ_WAVE_CYCLES'] or 1.0
    util[k] = dict(launches=int(d['launches']), ms=round(d['seconds'] * 1e3, 3),
                   mfma_busy_frac_at_2p4ghz=round(d['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles, 4),
                   wave_time_active=round(d['SQ_ACTIVE_INST_ANY'] / wc, 3), wave_time_issue_stall=round(d['SQ_WAIT_INST_ANY'] / wc, 3),
                   wave_time_parked=round(d['SQ_WAIT_ANY'] / wc, 3))
json.dump(util, open(os.path.join(out_dir, 'mfma_util.json'), 'w'), indent=1)
PY
# the bench quotes profiles/traffic.json / mfma_util.json only when their digest is the digest of the sources it runs: put the
# files just collected where it looks (on this box's copy; scripts/copy_round_profiles.sh does the same for the repository)
cp $OUT/traffic.json $OUT/mfma_util.json $ROOT/profiles/
S0=$SECONDS; python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench.py default run: $((SECONDS - S0)) s wall" | tee $OUT/bench_wall.txt
# rocprofv3 --kernel-trace --stats twice: (a) over `bench.py --steps-only` -- every launch under a symbol is a training step's,
# the file the in-step roofline of the bench line is checked against (kernel_stats_steps_only.csv: average duration of the
# dominant symbol = roofline.avg_launch_ms); (b) over the bench command itself (isolated-measurement launches included)
bash scripts/gpu_prof.sh round${TAG}_prof_steps python bench.py --steps-only --steps 40 --warmup 5 | head -40 > $OUT/kernel_top_steps_only.txt
python scripts/step_timeline.py $(find gpurun_out/round${TAG}_prof_steps -name '*kernel_trace.csv' | head -1) > $OUT/step_timeline.txt 2>/dev/null
cp $(find gpurun_out/round${TAG}_prof_steps -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_steps_only.csv
grep '^{' gpurun_out/round${TAG}_prof_steps/stdout.log > $OUT/bench_steps_only_under_rocprof.json
bash scripts/gpu_prof.sh round${TAG}_prof python bench.py --no-alt --no-cpu-baseline | head -40 > $OUT/kernel_top.txt
cp $(find gpurun_out/round${TAG}_prof -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
grep '^{' gpurun_out/round${TAG}_prof/stdout.log > $OUT/bench_under_rocprof.json
python - <<PY
import json
b = json.loads([l for l in open('$OUT/bench.json') if l.startswith('{')][-1])
r = b['roofline']
print('bench: %.3f ms/step (median %.3f), %s in-step frac %.4f (isolated %s), step_hw_frac %s, traffic %s' % (
    b['ms_per_step'], b['ms_per_step_median'], r['kernel'], r['frac'], (r.get('isolated') or {}).get('frac'), b.get('step_hw_frac'),
    'quoted' if r.get('traffic') else 'NOT quoted: ' + str(r.get('traffic_stale'))))
PY
find gpurun_out/round${TAG}_prof gpurun_out/round${TAG}_prof_steps $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
find $OUT -name '*.csv' -size +5M -delete
