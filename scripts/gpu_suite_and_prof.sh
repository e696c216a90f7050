#!/bin/bash
# the whole GPU suite, then a profiled bench of the default step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
bash scripts/gpu_prof.sh exp_new python bench.py --steps 10 --warmup 3 --no-alt --no-cpu-baseline "$@" | head -12
grep '^{' gpurun_out/exp_new/stdout.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('all_matrix_launches'), d.get('max_logit_err'), d.get('ctc_loss_delta_rel'))"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_plain.json 2> gpurun_out/bench_plain.err
python -c "
import json; d=json.loads([l for l in open('gpurun_out/bench_plain.json') if l.startswith('{')][-1])
print('plain', d['ms_per_step'], d.get('ms_per_step_median'), d['value'], d['roofline']['frac'], d.get('alt_bf16x6',{}).get('ms_per_step'), d.get('alt_bf16',{}).get('ms_per_step'))"
