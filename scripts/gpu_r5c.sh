#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c; mkdir -p $O
ms() { grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; }
timeout 120 python scripts/exp/graph_debug.py fp32 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL\|amdgpu.ids" | tee $O/graph_debug_fp32.txt
timeout 300 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_configs.py -q -m gpu -x -k "bf16 or config3" 2>&1 | tail -5 | tee $O/pytest_bf16.log
for rep in 1 2; do
  for T in 512 256 768 1024; do echo "bf16 wgrad target $T: $(timeout 200 python bench.py --steps-only --steps 100 --conv-mode bf16 --tune bf16_wgrad_target=$T 2>/dev/null | ms)"; done
  echo "bf16 tr=0: $(ST_BF16_WGRAD_TR=0 timeout 200 python bench.py --steps-only --steps 100 --conv-mode bf16 2>/dev/null | ms)"
done 2>&1 | tee $O/ab.txt
bash scripts/gpu_prof.sh r5c_prof_bf16 python bench.py --steps-only --steps 20 --warmup 5 --conv-mode bf16 | head -12 > $O/kernel_top_bf16.txt; cat $O/kernel_top_bf16.txt
python scripts/step_timeline.py $(find gpurun_out/r5c_prof_bf16 -name '*kernel_trace.csv' | head -1) > $O/step_timeline_bf16.txt 2>/dev/null
rm -rf gpurun_out/r5c_prof_bf16
