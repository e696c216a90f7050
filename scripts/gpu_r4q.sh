#!/bin/bash
# split update + two decoder streams: parity tests, fp32 / bf16 step, timeline, config-5 decode
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4q
python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_bf16.py tests/test_gpu_dp4.py -m gpu -x -q 2>&1 | tail -4
for m in fp32 bf16; do for i in 1 2; do python bench.py --steps-only --steps 40 --conv-mode $m 2>/dev/null | tail -1 | cut -c100-260; done; done
ST_SPLIT_UPDATE=0 python bench.py --steps-only --steps 40 2>/dev/null | tail -1 | cut -c100-260
ST_SPLIT_UPDATE=0 python bench.py --steps-only --steps 40 --conv-mode bf16 2>/dev/null | tail -1 | cut -c100-260
bash scripts/gpu_timeline.sh r4q > /dev/null
head -40 gpurun_out/r4q/step_timeline.txt | cut -c1-100
python scripts/bench_decode.py > gpurun_out/r4q/decode.json 2>gpurun_out/r4q/decode.err; cat gpurun_out/r4q/decode.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if 'overlap' in k or k in ('forward_ms','beam_ms_device_only')}); print(d['transcribe_beam'])"
