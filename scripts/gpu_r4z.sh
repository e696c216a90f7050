#!/bin/bash
cd $GRAFT_REPO_ROOT
for b in 16 32 64; do python scripts/bench_decode.py --beam $b --samples 3 --pipeline-batches 2 --resident-batches 4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('beam', $b, 'device ms', d['beam_ms_device_only'], 'incl d2h', d['beam_ms_incl_d2h'])"; done
