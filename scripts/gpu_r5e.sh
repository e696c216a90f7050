#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "mel or mfcc" 2>&1 | tail -4
for B in 32 512; do python scripts/bench_mel.py 80 $B 2>/dev/null | tail -1; done | tee $O/mel_timing.txt
bash scripts/gpu_mel_traffic.sh 32 > $O/mel_traffic_b32.log 2>&1; cp gpurun_out/mel_traffic_b32/mel_traffic.json $O/mel_traffic_b32.json; python -c "import json; d=json.load(open('$O/mel_traffic_b32.json')); print({k:v for k,v in d.items() if k.startswith('_') and k!='_note'}); print({k:v for k,v in d.items() if not k.startswith('_')})"
bash scripts/gpu_mel_traffic.sh 512 > $O/mel_traffic_b512.log 2>&1; cp gpurun_out/mel_traffic_b512/mel_traffic.json $O/mel_traffic_b512.json; python -c "import json; d=json.load(open('$O/mel_traffic_b512.json')); print({k:v for k,v in d.items() if k.startswith('_') and k!='_note'})"
bash scripts/gpu_prof.sh r5e_mel python scripts/bench_mel.py 80 512 | grep mel_ | head; rm -rf gpurun_out/r5e_mel gpurun_out/mel_traffic_b32/*SIZE gpurun_out/mel_traffic_b512/*SIZE
for M in fp32 bf16; do timeout 300 python scripts/bench_api_train.py --conv-mode $M 2>/dev/null | grep '^{' | tee $O/api_train_$M.json; done
