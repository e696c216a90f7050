#!/bin/bash
# usage: copy_round_profiles.sh <round>: gpurun_out/ (scratch) -> profiles/ (tracked), the artefacts of scripts/gpu_round_final.sh
# under their round names
set -e
N=${1:?round number}
cd "$(dirname "$0")/.."
R=gpurun_out/round$N F=gpurun_out/r${N}final P=profiles/r${N}
cp $R/bench.json ${P}_bench.json
cp $R/bench_under_rocprof.json ${P}_bench_under_rocprof.json
cp $R/bench_steps_only_under_rocprof.json ${P}_bench_steps_only_under_rocprof.json
cp $R/kernel_stats.csv ${P}_kernel_stats.csv
cp $R/kernel_stats_steps_only.csv ${P}_kernel_stats_steps_only.csv
cp $R/step_timeline.txt ${P}_step_timeline.txt
cp $R/traffic.json profiles/traffic.json
cp $R/mfma_util.json profiles/mfma_util.json
for MODE in bf16 bf16x6; do
  cp $F/kernel_stats_${MODE}_mode.csv ${P}_${MODE}_mode_kernel_stats.csv
  cp $F/step_timeline_${MODE}_mode.txt ${P}_${MODE}_mode_step_timeline.txt
done
cp $F/decode_config5.json ${P}_decode_config5.json
cp $F/inference_config3_fp32.json ${P}_inference_config3_fp32.json
cp $F/inference_config3_bf16.json ${P}_inference_config3_bf16.json
cp $F/pmc_shapes_bf16.txt ${P}_pmc_shapes_bf16.txt
cp $F/pmc_shapes_fp32.txt ${P}_pmc_shapes_fp32.txt
[ -f $F/pmc_shapes_bf16_round5_kernels.txt ] && cp $F/pmc_shapes_bf16_round5_kernels.txt ${P}_pmc_shapes_bf16_round5_kernels.txt
cp $F/pytest_gpu.log ${P}_pytest_gpu.log
cp $F/traffic_bf16.json profiles/traffic_bf16.json
cp $F/mel_traffic_b32.json ${P}_mel_traffic_b32.json
cp $F/mel_traffic_b512.json ${P}_mel_traffic_b512.json
cp $F/mel_timing.txt ${P}_mel_timing.txt
for M in fp32 bf16; do cp $F/host_cost_$M.json ${P}_host_cost_$M.json; cp $F/api_train_$M.json ${P}_api_train_$M.json; done
cp $F/forced_build.txt ${P}_forced_build.txt
for f in varlen_train_fp32_b32_m80 varlen_train_fp32_b64_m128_22050hz varlen_train_bf16_b32_m80 step_by_length_fp32 step_by_length_bf16 \
         api_train_fp32_world2_shared_gpu api_train_bf16_world2_shared_gpu bench_fp32_world2_shared_gpu bench_bf16_world2_shared_gpu \
         head_bubble_fp32 head_bubble_bf16 ctc_mask_probe; do
  [ -f $F/$f.json ] && cp $F/$f.json ${P}_$f.json
done
python - <<'PY'
import json
from speecht_amd import build
d = build.source_digest()
for f in ('profiles/traffic.json', 'profiles/mfma_util.json', 'profiles/traffic_bf16.json'):
  got = json.load(open(f)).get('source_digest')
  print(f, 'digest', got[:12], 'matches sources' if got == d else 'STALE against ' + d[:12])
PY
