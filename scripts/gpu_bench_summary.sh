#!/bin/bash
# usage: gpu_bench_summary.sh <tag> [bench.py args]: one bench.py run -> gpurun_out/<tag>.json + a short summary of its roofline
TAG="$1"; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py "$@" > gpurun_out/$TAG.json 2> gpurun_out/$TAG.err
tail -3 gpurun_out/$TAG.err | grep -v amdgpu.ids
python - "$TAG" <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/%s.json' % sys.argv[1]) if l.startswith('{')][-1])
r = d['roofline']
print('bench', d['ms_per_step'], d['ms_per_step_median'], d['value'], r['kernel'], 'in-step frac', r['frac'], 'isolated', (r.get('isolated') or {}).get('frac'),
      'hw', d.get('step_executed_gflop'), d.get('step_hw_frac'), 'profiled ms', d.get('profiled_ms_per_step'), 'alt_bf16', d.get('alt_bf16', {}).get('ms_per_step'))
for k, v in (r.get('per_shape') or {}).items():
  print('   ', k, v)
for g in r.get('by_kernel', []):
  print('  ', g['kernel'], g['launches_per_step'], g['ms_per_step'], g.get('frac'), (g.get('isolated') or {}).get('frac'))
PY
