#!/bin/bash
# everything profiles/ holds for round 3, from one build: GPU suite, default bench + rocprof stats + PMC passes
# (scripts/gpu_profile_round.sh), configs[2] / configs[4] benches, bf16-mode kernel stats and per-shape PMC
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3final
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r3final/pytest_gpu.log
cat gpurun_out/r3final/pytest_gpu.log
bash scripts/gpu_profile_round.sh 3 2>&1 | tail -6
timeout 600 python scripts/bench_inference.py > gpurun_out/r3final/inference_config3_fp32.json 2>/dev/null
timeout 600 python scripts/bench_inference.py --conv-mode bf16 > gpurun_out/r3final/inference_config3_bf16.json 2>/dev/null
timeout 300 python scripts/bench_decode.py > gpurun_out/r3final/decode_config5.json 2>/dev/null
cut -c1-300 gpurun_out/r3final/inference_config3_fp32.json; cut -c1-400 gpurun_out/r3final/decode_config5.json
bash scripts/gpu_prof.sh r3_prof_bf16 python bench.py --steps-only --steps 20 --warmup 5 --conv-mode bf16 | head -14 > gpurun_out/r3final/kernel_top_bf16.txt
cp $(find gpurun_out/r3_prof_bf16 -name '*kernel_stats.csv' | head -1) gpurun_out/r3final/kernel_stats_bf16_mode.csv
bash scripts/gpu_pmc_shapes.sh r3_bf16 --conv-mode bf16 > gpurun_out/r3final/pmc_shapes_bf16.txt 2>&1
bash scripts/gpu_pmc_shapes.sh r3_fp32 > gpurun_out/r3final/pmc_shapes_fp32.txt 2>&1
cd scripts/ubench && timeout 120 ./gemm_issue > $GRAFT_REPO_ROOT/gpurun_out/r3final/ubench_gemm_issue.txt 2>&1; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r3_prof_bf16/*/*kernel_trace.csv
find gpurun_out -name '*.csv' -size +4M -delete
du -sh gpurun_out | tail -1
