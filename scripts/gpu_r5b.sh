#!/bin/bash
# round 5, call 2: the transposing-read filter gradient (bf16 activations), the whole-step graph, wide beams
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b; mkdir -p $O
ms() { grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; }
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_parity.py -q -m gpu -x -k "bf16 or beam" 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -25 > $O/pytest_a.log; tail -8 $O/pytest_a.log
timeout 600 python -m pytest tests/test_gpu_api.py -q -m gpu -x -k "whole_step_graph" 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -40 > $O/pytest_graph.log; tail -25 $O/pytest_graph.log
for rep in 1 2; do
  echo "bf16 tr=1: $(timeout 200 python bench.py --steps-only --steps 100 --conv-mode bf16 2>/dev/null | ms)  tr=0: $(ST_BF16_WGRAD_TR=0 timeout 200 python bench.py --steps-only --steps 100 --conv-mode bf16 2>/dev/null | ms)  tr=1+graph: $(timeout 200 python bench.py --steps-only --steps 100 --conv-mode bf16 --graph 2>$O/graph_bf16.err | ms)"
  echo "fp32 eager: $(timeout 200 python bench.py --steps-only --steps 60 2>/dev/null | ms)  graph: $(timeout 200 python bench.py --steps-only --steps 60 --graph 2>$O/graph_fp32.err | ms)"
done 2>&1 | tee $O/ab.txt
tail -5 $O/graph_bf16.err $O/graph_fp32.err
timeout 400 python -m pytest tests/test_gpu_configs.py -q -m gpu -x -k "config3" -s 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -40 > $O/pytest_config3.log; tail -12 $O/pytest_config3.log
bash scripts/gpu_prof.sh r5b_prof_bf16 python bench.py --steps-only --steps 20 --warmup 5 --conv-mode bf16 | head -16 > $O/kernel_top_bf16.txt; cat $O/kernel_top_bf16.txt
python scripts/step_timeline.py $(find gpurun_out/r5b_prof_bf16 -name '*kernel_trace.csv' | head -1) > $O/step_timeline_bf16.txt 2>/dev/null
cp $(find gpurun_out/r5b_prof_bf16 -name '*kernel_stats.csv' | head -1) $O/kernel_stats_bf16_mode.csv
rm -rf gpurun_out/r5b_prof_bf16
