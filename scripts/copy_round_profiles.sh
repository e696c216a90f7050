#!/bin/bash
# gpurun_out/ (scratch) -> profiles/ (tracked): the artefacts of scripts/gpu_round3_final.sh under their round-3 names
set -e
cd "$(dirname "$0")/.."
R=gpurun_out/round3 F=gpurun_out/r3final
cp $R/bench.json profiles/r3_bench.json
cp $R/bench_under_rocprof.json profiles/r3_bench_under_rocprof.json
cp $R/kernel_stats.csv profiles/r3_kernel_stats.csv
cp $R/step_timeline.txt profiles/r3_step_timeline.txt
cp $R/traffic.json profiles/traffic.json
cp $R/mfma_util.json profiles/mfma_util.json
cp $F/kernel_stats_bf16_mode.csv profiles/r3_bf16_mode_kernel_stats.csv
cp $F/decode_config5.json profiles/r3_decode_config5.json
cp $F/inference_config3_fp32.json profiles/r3_inference_config3_fp32.json
cp $F/inference_config3_bf16.json profiles/r3_inference_config3_bf16.json
cp $F/pmc_shapes_bf16.txt profiles/r3_pmc_shapes_bf16.txt
cp $F/pmc_shapes_fp32.txt profiles/r3_pmc_shapes_fp32.txt
cp $F/pytest_gpu.log profiles/r3_pytest_gpu.log
cp $F/ubench_gemm_issue.txt profiles/r3_ubench_gemm_issue.txt
python - <<'PY'
import json
from speecht_amd import build
d = build.source_digest()
for f in ('profiles/traffic.json', 'profiles/mfma_util.json'):
  got = json.load(open(f)).get('source_digest')
  print(f, 'digest', got[:12], 'matches sources' if got == d else 'STALE against ' + d[:12])
PY
