#!/bin/bash
# usage: gpu_round_final.sh <round>: everything profiles/ holds for a round, from one build -- GPU suite, default bench + rocprof
# stats + PMC passes (scripts/gpu_profile_round.sh), configs[2] / configs[4] benches, bf16- and bf16x6-mode kernel stats and
# timelines, per-shape PMC.  Outputs under gpurun_out/round<round>/ and gpurun_out/r<round>final/; scripts/copy_round_profiles.sh
# <round> puts them into profiles/ under their round names.
R=${1:?round number}
cd $GRAFT_REPO_ROOT
F=gpurun_out/r${R}final
mkdir -p $F
# gfx950 compilation on the box itself, once per round (the shipped .so is otherwise reused when the source digest matches)
python -c "from speecht_amd.build import build_library; build_library(force=True, verbose=False)" 2>&1 | tail -3; echo "forced build rc=$?" | tee $F/forced_build.txt
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 > $F/pytest_gpu.log
cat $F/pytest_gpu.log
# (the bench line quotes profiles/traffic_bf16.json for its alt_bf16.roofline: collected BEFORE the default bench run below)
bash scripts/gpu_traffic_bf16.sh 2>&1 | tail -2
cp gpurun_out/traffic_bf16/traffic_bf16.json $F/
bash scripts/gpu_profile_round.sh $R 2>&1 | tail -6
for B in 32 512; do bash scripts/gpu_mel_traffic.sh $B > /dev/null 2>&1; cp gpurun_out/mel_traffic_b$B/mel_traffic.json $F/mel_traffic_b$B.json; cp gpurun_out/mel_traffic_b$B/timing_under_rocprof.txt $F/mel_timing_under_rocprof_b$B.txt; done
for B in 32 512; do python scripts/bench_mel.py 80 $B 2>/dev/null | tail -1; done > $F/mel_timing.txt; cat $F/mel_timing.txt
for M in fp32 bf16; do timeout 300 python scripts/bench_host_cost.py --conv-mode $M 2>/dev/null | grep '^{' > $F/host_cost_$M.json; timeout 300 python scripts/bench_api_train.py --conv-mode $M 2>/dev/null | grep '^{' > $F/api_train_$M.json; done
cat $F/host_cost_bf16.json $F/api_train_bf16.json
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
# the bf16 step's own kernels (filter gradients through transposing reads, the 7-tap panel kernel): same counters
KERNEL_FILTER='((?:conv_taps_bf16_kernel|wgrad_tr_bf16_kernel<[^>]*>))' bash scripts/gpu_pmc_shapes.sh r${R}_bf16_new --conv-mode bf16 > $F/pmc_shapes_bf16_round5_kernels.txt 2>&1
# round 6: training in the reference's real regime (a new (B, max_T) per step), the per-length step sweep, the data-parallel API
# loop on a shared GPU, the step-head and CU-masked-CTC probes
timeout 600 python scripts/bench_varlen_train.py --batch 32 --mels 80 --out $F/varlen_train_fp32_b32_m80.json > /dev/null 2>&1
timeout 600 python scripts/bench_varlen_train.py --batch 64 --mels 128 --rate 22050 --steps 40 --out $F/varlen_train_fp32_b64_m128_22050hz.json > /dev/null 2>&1
timeout 600 python scripts/bench_varlen_train.py --batch 32 --mels 80 --conv-mode bf16 --out $F/varlen_train_bf16_b32_m80.json > /dev/null 2>&1
timeout 600 python scripts/bench_varlen_train.py --sweep --out $F/step_by_length_fp32.json > /dev/null 2>&1
timeout 600 python scripts/bench_varlen_train.py --sweep --conv-mode bf16 --out $F/step_by_length_bf16.json > /dev/null 2>&1
for M in fp32 bf16; do
  timeout 300 python scripts/bench_api_train.py --conv-mode $M --world 2 2>/dev/null | grep '^{' > $F/api_train_${M}_world2_shared_gpu.json
  ST_SHARE_GPU=1 ST_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 3 --steps-only --conv-mode $M 2>/dev/null | grep '^{' > $F/bench_${M}_world2_shared_gpu.json
  python scripts/exp/head_bubble.py --conv-mode $M 2>/dev/null | grep '^{' > $F/head_bubble_$M.json
done
python scripts/exp/ctc_mask_probe.py 2>/dev/null | grep '^{' > $F/ctc_mask_probe.json
cat $F/varlen_train_fp32_b32_m80.json | cut -c1-1200
find gpurun_out -name '*.csv' -size +4M -delete
du -sh gpurun_out | tail -1
