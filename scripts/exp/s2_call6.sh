# session 2, call 6: full GPU suite on the current build + default bench line
mkdir -p gpurun_out/s2c6
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -12 > gpurun_out/s2c6/pytest_gpu.log
cat gpurun_out/s2c6/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/s2c6/bench.json 2> gpurun_out/s2c6/bench.err; echo rc=$?
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/s2c6/bench.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_median','step_hw_frac')})
r=d['roofline']; print(r['kernel'], r['frac'], r['avg_launch_ms'])
for g in r['by_kernel']:
  print(g['kernel'][:60], g['launches_per_step'], round(g['ms_per_step'],3), g.get('frac'), g.get('hbm_frac'), g.get('mfma_frac'))
print('alt', d['alt_bf16']['ms_per_step'], d['alt_bf16x6']['ms_per_step'], d['configs2_inference']['utterances_per_s'], d['configs4_decode']['utterances_per_s'])
PY
