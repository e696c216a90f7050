"""Per-GPU shards of BASELINE configs[3] and configs[4] at their real sizes (what one of the 8 ranks runs).

configs[3]: 8 x MI355X data-parallel training, global batch 256 -> one rank: B = 32 x 10 s, bf16 activations /
            fp32 CTC.  Checked against the oracle run with the same storage model (oracle.bf16_round where the
            device writes bf16) on the WHOLE shard: logits, per-utterance losses, all 22 gradient tensors.
configs[4]: 8 x MI355X, 30 s utterances, batch 128 -> one rank: B = 16, T = 3001 (T' = 1501), forward + greedy +
            prefix beam search (beam 16).  Logits against the float64 oracle on two rows; decoders against the
            oracle decoders fed the device's own logits (so that ties cannot flip), all 16 rows greedy, 2 rows beam.
The 8-rank exchange itself (RCCL) cannot run on a 1-GPU box; its control flow is covered by
test_gpu_api.py::test_data_parallel_two_ranks_on_one_gpu_match_single_process and bench.py --gpus 2 (ST_SHARE_GPU).
"""
import time

import numpy as np
import pytest

from oracle import w2l_oracle as O
from tests import workloads as WL

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

ULP = 2.0 ** -8


def scaled_err(a, b):
  s = float(np.max(np.abs(b))) + 1e-30
  d = np.abs(np.asarray(a, np.float64) - b) / s
  return float(d.max()), float(d.mean())


def plane(t3, buf):
  """[B,T,C] float64 copy of the valid region of a bf16 (or fp32) buffer laid out like the padded NWC tensor t3."""
  v = buf.view(t3.batch, t3.t_pitch, t3.c_pitch)[:, t3.halo:t3.halo + t3.frames, :t3.channels]
  return v.float().cpu().numpy().astype(np.float64)


def test_config3_rank_shard_bf16_training_step_vs_bf16_storage_oracle():
  """One rank's shard of configs[3] (B = 32 x 10 s, bf16 activations / fp32 CTC) at full size, two ways:

  * end to end from the inputs against the oracle with the bf16 storage model: logits and losses (tight), the 22
    gradient tensors (loose: in bf16 storage a 1e-5 difference upstream -- fp32 vs float64 accumulation, the CTC
    kernel's 1.5e-5 -- turns into 1-ulp flips of stored values, i.e. noise of ~2e-4 of a tensor's scale, which the
    ill-conditioned lower layers amplify ~75x on the way to L0);
  * kernel by kernel on the device's OWN stored operands, where nothing is amplified: every layer's stored output
    against round_bf16(conv(stored input)), every stored activation gradient against round_bf16(mask * back-prop of
    the stored gradient above), every filter / bias gradient against the float64 product of the stored operands.
    Here a mismatch can only be a value on a rounding boundary: max <= 1 bf16 ulp of the tensor scale, mean <= 0.01."""
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  from speecht_amd._lib import launch_trace
  from speecht_amd.engine import Wav2LetterEngine
  layers = WL.w2l_layers(80)
  params = WL.xavier_params(layers, seed=42, dtype=np.float32)
  B, L = 32, len(layers)
  x, seq, labels = WL.make_batch([1001] * B, 80, seed=103)            # rank 3's shard of bench.py's global batch
  x = x.astype(np.float32)
  eng = Wav2LetterEngine(layers, device='cuda:0', conv_mode='bf16')
  eng.set_weights(params)
  eng.load_batch(x, seq)
  eng.set_labels(labels)
  with launch_trace() as tr:
    eng.forward()
    eng.ctc_loss_grad(1.0 / (8 * B))                                  # d avg_loss over the GLOBAL batch of 256
    eng.backward()
  torch.cuda.synchronize()
  eng.check_ctc_status()
  assert sum(1 for l in tr.lines if l.startswith('gemm_nn_bf16<256,NP=1>')) >= 4, '\n'.join(tr.lines)   # L9's passes + L8's forward per-bin products
  p64 = [(F.astype(np.float64), b.astype(np.float64)) for F, b in params]
  grads = eng.get_grads()

  # ---- end to end from the inputs ----
  t0 = time.time()
  spectral = set(eng.fftb)            # the 32-tap layer: block DFTs + per-bin products on bf16 spectra (its own storage model)
  batched = lambda bins, kernel='gemm_nn_bf16<': sum(1 for l in tr.lines if l.startswith(kernel) and ' batched bins=%d ' % bins in l)
  # (two per-bin products of 48 bins; the lag products as 96 real / imaginary ones, since round 5 straight from the spectra planes
  # through the transposing-read kernel; the stride-1 layers' filter gradients on that kernel too: L1-L7, L9, L10)
  assert spectral == {8} and batched(48) == 2 and batched(96, 'wgrad_tr_bf16<128,128,32,lag>') == 1, '\n'.join(tr.lines)
  assert sum(1 for l in tr.lines if l.startswith('wgrad_tr_bf16<128,128,32> ')) == 9, '\n'.join(tr.lines)
  # (the 7-tap layers whose two tensors share a frame pitch on the panel kernel: forward L1-L6, back-prop to the input L2-L7)
  assert sum(1 for l in tr.lines if l.startswith('conv_taps_bf16<128,128,64,panel> ')) == 12, '\n'.join(tr.lines)
  logits, acts = O.wav2letter_forward(x.astype(np.float64), p64, layers, keep=True, store=O.bf16_round, spectral=spectral)
  loss, g_logits = O.ctc_loss_and_grad(logits, labels, seq // 2)
  ref_grads = O.wav2letter_backward(acts, p64, layers, g_logits / (8 * B), store=O.bf16_round, spectral=spectral)
  print('bf16-storage oracle on the whole shard: %.1f s' % (time.time() - t0))
  got = eng.logits_time_major().cpu().numpy()
  mx, mean = scaled_err(got, logits)
  print('logits: max %.2f ulp, mean %.3f ulp (bf16 ulp of the tensor scale)' % (mx / ULP, mean / ULP))
  assert mx < 16 * ULP and mean < 0.5 * ULP, (mx, mean)
  np.testing.assert_allclose(eng.loss.cpu().numpy(), loss, rtol=2e-2)
  for i, ((gF, gb), (rF, rb)) in enumerate(zip(grads, ref_grads)):
    mxF, meanF = scaled_err(gF, rF)
    mxb, _ = scaled_err(gb, rb)
    print('end to end L%d: filters max %.2f ulp mean %.3f ulp, bias max %.2f ulp' % (i, mxF / ULP, meanF / ULP, mxb / ULP))
    assert mxF < 64 * ULP and meanF < 8 * ULP and mxb < 64 * ULP, (i, mxF, meanF, mxb)
  del acts, ref_grads

  # ---- kernel by kernel on the device's stored operands ----
  # (a stored value is the fp32 accumulation rounded to bf16, the expectation the float64 one: where the exact value sits
  # within ~1e-7 of a rounding boundary the two may land on neighbouring bf16 values -- one spacing, which in the top
  # binade of the tensor is up to 2 ULP of the tensor's maximum; the mean bound says it is a handful of elements)
  t0 = time.time()
  Xs = [plane(eng.X[i], eng.Xb[i]) for i in range(L)]                 # stored bf16 inputs of every layer
  dZs = [plane(eng.dZ[i], eng.dZb[i]) for i in range(L)]              # stored bf16 gradients wrt every layer's output
  for i, ((F, b), (W, s, cin, cout, relu)) in enumerate(zip(p64, layers)):
    if i in spectral:
      # the frequency-domain layer against ITS storage model on the stored operands: spectra of the stored input, of the stored
      # gradient and of the fp32 master filters rounded to bf16, everything else exact (an element of a spectrum within fp32
      # rounding of a bf16 boundary may land on the other side than in float64: slightly looser means than the W-tap layers')
      y, dx, dF, db = O.block_dft_conv(Xs[i], F, b, relu, dz=dZs[i], store=O.bf16_round)
    else:
      y = O.conv1d_same_fwd(Xs[i], O.bf16_round(F), b, s, relu)
      dx, dF, db = O.conv1d_same_bwd(Xs[i], O.bf16_round(F), None, dZs[i], s, relu=False, need_dx=(i > 0))
    mean_tol = 0.05 if i in spectral else 0.01
    if i in spectral:
      # ... and against the DIRECT form on the same stored operands (the storage model above must not define the
      # accuracy loss of bf16 spectra away).  What bf16 spectra cost relative to the W-tap bf16 kernel, which rounds only its
      # output: the oracle's two forms differ by max 1.5 / mean 0.06 ulp (forward), 1.5 / 0.12 ulp (back-prop) and 2.5e-3 of
      # the maximum (filter gradient, 0.6 ulp) on random data of this layer's shape class; the bounds leave a factor ~3.
      yd = O.conv1d_same_fwd(Xs[i], O.bf16_round(F), b, s, relu)
      dxd, dFd, dbd = O.conv1d_same_bwd(Xs[i], O.bf16_round(F), None, dZs[i], s, relu=False, need_dx=True)
      mx, mean = scaled_err(Xs[i + 1], O.bf16_round(yd))
      print('spectral L%d vs DIRECT form: forward max %.2f ulp mean %.3f ulp' % (i, mx / ULP, mean / ULP))
      assert mx <= 4 * ULP and mean < 0.25 * ULP, ('forward vs direct form', i, mx / ULP, mean / ULP)
      dxm = dxd * (Xs[i] > 0) if layers[i - 1][4] else dxd
      mx, mean = scaled_err(dZs[i - 1], O.bf16_round(dxm))
      print('spectral L%d vs DIRECT form: back-prop max %.2f ulp mean %.3f ulp' % (i, mx / ULP, mean / ULP))
      assert mx <= 4 * ULP and mean < 0.4 * ULP, ('back-prop vs direct form', i, mx / ULP, mean / ULP)
      mxF, _ = scaled_err(grads[i][0], dFd)
      mxb, _ = scaled_err(grads[i][1], dbd)
      print('spectral L%d vs DIRECT form: filter gradient %.1e, bias gradient %.1e of max' % (i, mxF, mxb))
      assert mxF < 8e-3 and mxb < 2e-4, ('filter/bias gradient vs direct form', i, mxF, mxb)
      del yd, dxd, dFd, dxm
    if i + 1 < L:
      mx, mean = scaled_err(Xs[i + 1], O.bf16_round(y))
      assert mx <= 2.01 * ULP and mean < mean_tol * ULP, ('forward', i, mx / ULP, mean / ULP)
    else:
      mx, _ = scaled_err(plane(eng.X[L], eng.X[L].buf), y)            # logits stay fp32
      assert mx < 1e-5, ('logits from stored X10', mx)
    # filter / bias gradient of layer i from its stored operands
    mxF, _ = scaled_err(grads[i][0], dF)
    mxb, _ = scaled_err(grads[i][1], db)
    assert mxF < (1e-3 if i in spectral else 2e-4) and mxb < 2e-4, ('filter/bias gradient', i, mxF, mxb)
    # gradient handed to the layer below: mask of the stored activation, rounded to bf16 when written
    if i > 0:
      if layers[i - 1][4]:
        dx = dx * (Xs[i] > 0)
      mx, mean = scaled_err(dZs[i - 1], O.bf16_round(dx))
      assert mx <= 2.01 * ULP and mean < mean_tol * ULP, ('back-prop to the input', i, mx / ULP, mean / ULP)
    print('kernel-level L%d ok (filters %.1e, bias %.1e of max)' % (i, mxF, mxb))
  # the bf16 copy of d loss / d logits and the fp32 CTC gradient it was rounded from
  mx, _ = scaled_err(dZs[L - 1], O.bf16_round(plane(eng.dZ[L - 1], eng.dZ[L - 1].buf)))
  assert mx == 0.0
  dl_ref = np.transpose(O.ctc_loss_and_grad(got.astype(np.float64), labels, seq // 2)[1], (1, 0, 2)) / (8 * B)
  mx, _ = scaled_err(plane(eng.dZ[L - 1], eng.dZ[L - 1].buf), dl_ref)
  assert mx < 2e-4, mx
  print('kernel-by-kernel checks on the stored operands: %.1f s' % (time.time() - t0))


def test_config4_rank_shard_long_form_forward_and_decoders():
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  from speecht_amd.engine import Wav2LetterEngine
  layers = WL.w2l_layers(80)
  params = WL.xavier_params(layers, seed=42, dtype=np.float32)
  B, T = 16, 3001
  frames = [T] * 14 + [2500, 1777]
  x, seq, _ = WL.make_batch(frames, 80, seed=11)
  x = x.astype(np.float32)
  eng = Wav2LetterEngine(layers, device='cuda:0')
  eng.set_weights(params)
  eng.load_batch(x, seq)
  eng.forward()
  torch.cuda.synchronize()
  got = eng.logits_time_major().cpu().numpy()
  assert got.shape == (1501, B, 29)
  rows = [0, 15]
  p64 = [(F.astype(np.float64), b.astype(np.float64)) for F, b in params]
  ref = O.wav2letter_forward(x[rows].astype(np.float64), p64, layers)
  err = float(np.max(np.abs(got[:, rows] - ref)))
  print('config 4 shard: max|logit err| on rows %s = %.2e (max |logit| %.3f)' % (rows, err, float(np.max(np.abs(ref)))))
  assert err < 1e-4 and err < 2e-5 * float(np.max(np.abs(ref)))
  ids, score = eng.greedy_decode()
  ref_ids, ref_score = O.ctc_greedy_decode(got, seq // 2)
  assert ids == ref_ids
  np.testing.assert_allclose(score, ref_score, rtol=1e-5)
  b_ids, b_score = eng.beam_search_decode(beam_width=16)
  r_ids, r_score = O.ctc_beam_search_decode(got[:, rows], (seq // 2)[rows], beam_width=16)
  assert [b_ids[r] for r in rows] == r_ids
  np.testing.assert_allclose(b_score[rows], r_score, rtol=1e-4, atol=1e-3)
  # the beam's best labelling is at least as probable as the greedy path's labelling
  assert all(len(i) <= 1501 for i in b_ids)


def test_bench_self_launch_two_ranks_share_one_gpu():
  """`python bench.py --gpus 2` exactly as the driver invokes it (no torch.distributed environment): bench.py
  re-launches itself under torch.distributed.run.  On a 1-GPU box the two ranks share cuda:0 and use gloo
  (test knobs ST_SHARE_GPU / ST_DIST_BACKEND); the control flow, the bucketed exchange hooks, the max-over-ranks
  timing and the one JSON line are the production ones."""
  import json
  import os
  import subprocess
  import sys
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, ST_SHARE_GPU='1', ST_DIST_BACKEND='gloo')
  for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
    env.pop(k, None)
  r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                      '--batch', '4', '--seconds', '2', '--no-alt', '--no-cpu-baseline'],
                     env=env, cwd=root, capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, r.stdout
  out = json.loads(lines[0])
  assert out['n_gpus'] == 2 and out['config']['global_batch'] == 8 and out['scaling'] == 'weak'
  assert out['replicas_identical'] is True
  assert len(out['per_rank_ms_per_step']) == 2 and out['comm']['allreduce_alone_ms'] > 0
  assert out['max_logit_err'] < 1e-4 and out['parity']['max_logit_err_rel'] < 2e-5 and out['parity']['ctc_loss_delta_rel'] < 1e-4
  assert out['parity_nonzero_bias']['passed'] is True
