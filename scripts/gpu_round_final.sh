#!/bin/bash
# usage: gpu_round_final.sh <round>: everything profiles/ holds for a round, from one build -- GPU suite, default bench + rocprof
# stats + PMC passes (scripts/gpu_profile_round.sh), configs[2] / configs[4] benches, bf16- and bf16x6-mode kernel stats and
# timelines, per-shape PMC.  Outputs under gpurun_out/round<round>/ and gpurun_out/r<round>final/; scripts/copy_round_profiles.sh
# <round> puts them into profiles/ under their round names.
R=${1:?round number}
cd $GRAFT_REPO_ROOT
F=gpurun_out/r${R}final
mkdir -p $F
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 > $F/pytest_gpu.log
cat $F/pytest_gpu.log
bash scripts/gpu_profile_round.sh $R 2>&1 | tail -6
timeout 600 python scripts/bench_inference.py > $F/inference_config3_fp32.json 2>/dev/null
timeout 600 python scripts/bench_inference.py --conv-mode bf16 > $F/inference_config3_bf16.json 2>/dev/null
timeout 300 python scripts/bench_decode.py > $F/decode_config5.json 2>/dev/null
cut -c1-300 $F/inference_config3_fp32.json; cut -c1-700 $F/decode_config5.json
for MODE in bf16 bf16x6; do
  bash scripts/gpu_prof.sh r${R}_prof_$MODE python bench.py --steps-only --steps 20 --warmup 5 --conv-mode $MODE | head -14 > $F/kernel_top_$MODE.txt
  cp $(find gpurun_out/r${R}_prof_$MODE -name '*kernel_stats.csv' | head -1) $F/kernel_stats_${MODE}_mode.csv
  python scripts/step_timeline.py $(find gpurun_out/r${R}_prof_$MODE -name '*kernel_trace.csv' | head -1) > $F/step_timeline_${MODE}_mode.txt 2>/dev/null
  grep '^{' gpurun_out/r${R}_prof_$MODE/stdout.log > $F/bench_steps_only_${MODE}_under_rocprof.json
  rm -rf gpurun_out/r${R}_prof_$MODE
done
bash scripts/gpu_pmc_shapes.sh r${R}_bf16 --conv-mode bf16 > $F/pmc_shapes_bf16.txt 2>&1
bash scripts/gpu_pmc_shapes.sh r${R}_fp32 > $F/pmc_shapes_fp32.txt 2>&1
find gpurun_out -name '*.csv' -size +4M -delete
du -sh gpurun_out | tail -1
