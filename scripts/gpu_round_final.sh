#!/bin/bash
# the artefacts profiles/ holds for the round: default bench line, kernel stats + step timeline of the fp32 step, PMC
# traffic and matrix-pipe utilisation of the gemm kernels, the other configs' benches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
bash scripts/gpu_prof.sh final_prof python bench.py --no-alt --no-cpu-baseline | head -40 > gpurun_out/final/kernel_top.txt
python scripts/step_timeline.py $(find gpurun_out/final_prof -name '*kernel_trace.csv' | head -1) > gpurun_out/final/step_timeline.txt
cp $(find gpurun_out/final_prof -name '*kernel_stats.csv' | head -1) gpurun_out/final/kernel_stats.csv
grep '^{' gpurun_out/final_prof/stdout.log > gpurun_out/final/bench_under_rocprof.json
bash scripts/gpu_traffic.sh > gpurun_out/final/traffic.log 2>&1
bash scripts/gpu_mfma_util.sh > gpurun_out/final/mfma.log 2>&1
cp gpurun_out/traffic/traffic.json gpurun_out/mfma/mfma_util.json gpurun_out/final/ 2>/dev/null
python scripts/bench_inference.py > gpurun_out/final/inference_fp32.json 2> gpurun_out/final/inference_fp32.err
python scripts/bench_inference.py --conv-mode bf16 > gpurun_out/final/inference_bf16.json 2> gpurun_out/final/inference_bf16.err
python scripts/bench_decode.py > gpurun_out/final/decode_config5.json 2> gpurun_out/final/decode.err
bash scripts/gpu_prof.sh final_prof_x6 python bench.py --no-alt --no-cpu-baseline --conv-mode bf16x6 | head -30 > gpurun_out/final/kernel_top_bf16x6.txt
cp $(find gpurun_out/final_prof_x6 -name '*kernel_stats.csv' | head -1) gpurun_out/final/kernel_stats_bf16x6.csv
rm -rf gpurun_out/final_prof/*kernel_trace.csv gpurun_out/final_prof_x6/*kernel_trace.csv
python -c "
import json; d=json.loads([l for l in open('gpurun_out/final/bench.json') if l.startswith('{')][-1])
print(d['ms_per_step'], d['ms_per_step_median'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('all_matrix_launches'), d['cpu_baseline']['value'], d.get('alt_bf16x6',{}).get('ms_per_step'), d.get('alt_bf16',{}).get('ms_per_step'))"
tail -2 gpurun_out/final/traffic.log | cut -c1-600; tail -30 gpurun_out/final/mfma.log | head -40
