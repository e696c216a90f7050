#!/bin/bash
# round 6: the three-product form of the per-bin products -- parity tests, then the step in its variants (ST_TUNE knobs)
cd $GRAFT_REPO_ROOT
O=gpurun_out/g3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_g3.py tests/test_gpu_fft_conv.py -x -q 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15 | tee $O/pytest.log
for k in ${VARIANTS:-no_g3=0 no_g3=1 no_g3=0 no_g3=1}; do
  ST_TUNE=$k timeout 300 python bench.py --steps-only --steps 30 --warmup 8 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$k', d['ms_per_step'], d.get('ms_per_step_median'))" | tee -a $O/ab.txt
done
bash scripts/gpu_prof.sh g3_prof python bench.py --steps-only --steps 20 --warmup 5 | head -${TOP:-24} | tee $O/kernel_top.txt
python scripts/step_timeline.py $(find gpurun_out/g3_prof -name '*kernel_trace.csv' | head -1) > $O/step_timeline.txt 2>/dev/null
