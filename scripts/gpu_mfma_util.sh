#!/bin/bash
# Matrix-pipe utilisation of the big kernels from PMC counters (its own pass: counters + kernel-trace only).
# SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs (MI355X_MICROARCH.md: = 64 x N_mfma for
# v_mfma_f32_32x32x2_f32); utilisation = busy / (kernel duration x 2.4 GHz x 1024 SIMDs).
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/mfma
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT -o m -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-alt --no-cpu-baseline > $OUT/run.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json, os, collections
root = os.environ['GRAFT_REPO_ROOT']
cc = glob.glob(os.path.join(root, 'gpurun_out/mfma/**/*counter_collection.csv'), recursive=True)[0]
kt = glob.glob(os.path.join(root, 'gpurun_out/mfma/**/*kernel_trace.csv'), recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(cc)):
    name = r['Kernel_Name']
    if 'gemm_' not in name: continue
    key = name[name.index('gemm_'):name.index('>') + 1]
    agg[key][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVE_CYCLES':
        agg[key]['seconds'] += dur.get(r['Dispatch_Id'], 0.0)
        agg[key]['launches'] += 1
out = {}
for k, d in agg.items():
    if d['seconds'] <= 0: continue
    simd_cycles = d['seconds'] * 2.4e9 * 1024
    wc = d['SQ_WAVE_CYCLES'] or 1.0
    out[k] = dict(launches=int(d['launches']), ms=round(d['seconds'] * 1e3, 3),
                  mfma_busy_frac_at_2p4ghz=round(d['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles, 4),
                  wave_time_active=round(d['SQ_ACTIVE_INST_ANY'] / wc, 3), wave_time_issue_stall=round(d['SQ_WAIT_INST_ANY'] / wc, 3),
                  wave_time_parked=round(d['SQ_WAIT_ANY'] / wc, 3))
json.dump(out, open(os.path.join(root, 'gpurun_out/mfma/mfma_util.json'), 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
