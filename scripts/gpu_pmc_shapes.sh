#!/bin/bash
# Per-SHAPE counters of the GEMM launches of one mode (rocprofv3 lists dispatches; the same kernel symbol runs several layer
# shapes per step, told apart here by their grid size): duration, fabric bytes (FETCH_SIZE x 2 per MI355X_MICROARCH.md,
# WRITE_SIZE), L2 hit rate, matrix-pipe busy, wave-time split, shader clock.  Separate --pmc passes, kernel-trace only.
# usage: gpu_pmc_shapes.sh <tag> <bench.py args...>      -> gpurun_out/pmc_shapes_<tag>/summary.json
TAG="$1"; shift
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=$ROOT/gpurun_out/pmc_shapes_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
CMD="python $ROOT/bench.py --steps-only --steps 3 --warmup 1 $*"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -o f -- $CMD > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/w -o w -- $CMD > $OUT/w.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/s -o s -- $CMD > $OUT/s.log 2>&1
cd $ROOT
python - "$OUT" <<'PY'
import collections, csv, glob, json, os, re, sys
out_dir = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for tag in ('f', 'w', 's'):
    cc = glob.glob(os.path.join(out_dir, tag, '**/*counter_collection.csv'), recursive=True)
    kt = glob.glob(os.path.join(out_dir, tag, '**/*kernel_trace.csv'), recursive=True)
    if not cc or not kt:
        print('pass', tag, 'produced no counters:', open(os.path.join(out_dir, tag + '.log')).read()[-1200:]); continue
    dur = {r['Dispatch_Id']: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9 for r in csv.DictReader(open(kt[0]))}
    seen = set()
    for r in csv.DictReader(open(cc[0])):
        m = re.search(os.environ.get('KERNEL_FILTER', r'(gemm_\w+<[^>]*>)'), r['Kernel_Name'])
        if not m:
            continue
        key = '%s grid=%s' % (m.group(1), r.get('Grid_Size', r.get('Grid_Size_X', '?')))
        agg[key][r['Counter_Name']] += float(r['Counter_Value'])
        if (tag, r['Dispatch_Id']) not in seen:
            seen.add((tag, r['Dispatch_Id']))
            agg[key]['seconds_' + tag] += dur.get(r['Dispatch_Id'], 0.0)
            agg[key]['launches_' + tag] += 1
res = {}
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get('seconds_s', 0.0)):
    n = max(d['launches_s'], 1.0)
    sec = d['seconds_s'] / n
    if sec <= 0:
        continue
    wc = d['SQ_WAVE_CYCLES'] or 1.0
    fetch = d['FETCH_SIZE'] * 1024 * 2 / max(d['launches_f'], 1.0)
    write = d['WRITE_SIZE'] * 1024 / max(d['launches_w'], 1.0)
    res[k] = dict(launches=int(n), avg_us=round(sec * 1e6, 1),
                  fabric_read_mb=round(fetch / 1e6, 1), fabric_write_mb=round(write / 1e6, 1),
                  fabric_tbs=round((fetch + write) / sec / 1e12, 2),
                  l2_hit_rate=round(d['TCC_HIT_sum'] / max(d['TCC_HIT_sum'] + d['TCC_MISS_sum'], 1.0), 4),
                  l2_requests_m=round((d['TCC_HIT_sum'] + d['TCC_MISS_sum']) / max(d['launches_w'], 1.0) / 1e6, 2),
                  mfma_busy_frac_at_2p4ghz=round(d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['seconds_s'] * 2.4e9 * 1024), 4),
                  clock_ghz=round(d['GRBM_GUI_ACTIVE'] / 8.0 / d['seconds_s'] / 1e9, 3) if d['GRBM_GUI_ACTIVE'] else None,   # (the counter sums the 8 XCDs)
                  wave_time_active=round(d['SQ_ACTIVE_INST_ANY'] / wc, 3), wave_time_issue_stall=round(d['SQ_WAIT_INST_ANY'] / wc, 3),
                  wave_time_parked=round(d['SQ_WAIT_ANY'] / wc, 3), wave_time_lds_issue_stall=round(d['SQ_WAIT_INST_LDS'] / wc, 3))
json.dump(res, open(os.path.join(out_dir, 'summary.json'), 'w'), indent=1)
for k, v in res.items():
    print(k, json.dumps(v))
PY
find $OUT -name '*.csv' -delete
