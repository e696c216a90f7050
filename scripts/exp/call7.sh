mkdir -p gpurun_out/c7
timeout 900 python bench.py > gpurun_out/c7/bench.json 2> gpurun_out/c7/bench.err
echo rc=$?
tail -3 gpurun_out/c7/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/c7/bench.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_median','step_hw_frac')})
r=d['roofline']; print(r['kernel'], r['frac'], r['avg_launch_ms'])
for g in r['by_kernel']:
  print(g['kernel'], g['launches_per_step'], g['ms_per_step'], g.get('frac'), g.get('hbm_frac'), g.get('mfma_frac'), g.get('algorithmic_mb_per_launch'))
print('parity', d['parity']['max_logit_err'], d['parity']['max_logit_err_rel'], d['parity_nonzero_bias']['max_logit_err'], d['parity_nonzero_bias']['max_logit_err_rel'], d['parity_nonzero_bias']['max_abs_logit'])
print('alt', d['alt_bf16']['ms_per_step'], d['alt_bf16x6']['ms_per_step'], d['configs2_inference']['utterances_per_s'], d['configs4_decode']['utterances_per_s'])
PY
