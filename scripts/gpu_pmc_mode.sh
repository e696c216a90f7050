#!/bin/bash
# Matrix-pipe / issue / LDS counters of the GEMM kernels for one arithmetic mode (fp32 | bf16 | bf16x6), each
# counter set in its own rocprofv3 pass (counters + kernel-trace only).  Output: gpurun_out/pmc_<mode>/summary.json
# MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x 2.4 GHz x 1024 SIMDs)   (MI355X_MICROARCH.md)
MODE="${1:-fp32}"
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=$ROOT/gpurun_out/pmc_$MODE
rm -rf $OUT; mkdir -p $OUT
cd /tmp
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-alt --no-cpu-baseline --conv-mode $MODE"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/a -o a -- $CMD > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $OUT/b -o b -- $CMD > $OUT/b.log 2>&1
cd $ROOT
python - "$OUT" "$MODE" <<'PY'
import csv, glob, json, os, sys, collections
out_dir, mode = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
for tag in ('a', 'b'):
    cc = glob.glob(os.path.join(out_dir, tag, '**/*counter_collection.csv'), recursive=True)
    kt = glob.glob(os.path.join(out_dir, tag, '**/*kernel_trace.csv'), recursive=True)
    if not cc or not kt:
        print('pass', tag, 'produced no counters:', open(os.path.join(out_dir, tag + '.log')).read()[-1500:]); continue
    dur = {r['Dispatch_Id']: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9 for r in csv.DictReader(open(kt[0]))}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(cc[0])):
        name = r['Kernel_Name']
        import re as _re
        m = _re.search(os.environ.get('KERNEL_FILTER', r'gemm_\w+<[^>]*>'), name)
        if not m: continue
        key = m.group(0)
        agg[key][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES':
            agg[key]['seconds'] += dur.get(r['Dispatch_Id'], 0.0); agg[key]['launches'] += 1
    for k, d in agg.items():
        if d['seconds'] <= 0: continue
        simd = d['seconds'] * 2.4e9 * 1024
        wc = d['SQ_WAVE_CYCLES'] or 1.0
        o = res[k]
        o.setdefault('launches', int(d['launches'])); o.setdefault('ms_total', round(d['seconds'] * 1e3, 3))
        if tag == 'a':
            o.update(mfma_busy_frac_at_2p4ghz=round(d['SQ_VALU_MFMA_BUSY_CYCLES'] / simd, 4),
                     wave_time_active=round(d['SQ_ACTIVE_INST_ANY'] / wc, 3), wave_time_issue_stall=round(d['SQ_WAIT_INST_ANY'] / wc, 3),
                     wave_time_parked=round(d['SQ_WAIT_ANY'] / wc, 3))
        else:
            o.update(wave_time_lds_issue_stall=round(d['SQ_WAIT_INST_LDS'] / wc, 3),
                     lds_array_busy_frac=round(d['SQ_LDS_IDX_ACTIVE'] / (d['seconds'] * 2.4e9 * 256), 4),
                     lds_bank_conflict_frac_of_lds_cycles=round(d['SQ_LDS_BANK_CONFLICT'] / max(d['SQ_LDS_IDX_ACTIVE'], 1.0), 4),
                     lds_insts_per_launch=round(d['SQ_INSTS_LDS'] / d['launches'], 1),
                     bf16_mfma_mops_per_launch=round(d['SQ_INSTS_VALU_MFMA_MOPS_BF16'] / d['launches'], 1))
json.dump({'mode': mode, 'kernels': res}, open(os.path.join(out_dir, 'summary.json'), 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
find $OUT -name '*.csv' -size +8M -delete
