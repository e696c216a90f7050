#!/bin/bash
# experiment: frequency-domain transforms / first layer on its polyphase view -- parity tests, then a profiled bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fft_conv.py -x -q 2>&1 | tail -15 > gpurun_out/exp_fft_tests.log
timeout 1200 python -m pytest tests/test_gpu_fullsize_grads.py -x -q -s 2>&1 | tail -40 >> gpurun_out/exp_fft_tests.log
cat gpurun_out/exp_fft_tests.log
bash scripts/gpu_prof.sh exp_new python bench.py --steps 10 --warmup 3 --no-alt --no-cpu-baseline "$@"
grep '^{' gpurun_out/exp_new/stdout.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('all_matrix_launches'), d.get('max_logit_err'), d.get('ctc_loss_delta_rel'))"
tail -5 gpurun_out/exp_new/stdout.log | cut -c1-400
