"""Gradient parity AT BASELINE's headline size (configs[1]: B=32 x 10 s x 80-mel, 250/2000 channels).

The kernels the benchmark runs -- gemm_nn<128,128,2,2,fast> forward and back-prop (incl. the 2-way split-K of
L8's back-prop), gemm_tn<128> with row slabs at M = 16 032 -- are only selected at near-full batch, so this module
runs one full-size step and compares EVERY gradient tensor, the per-utterance losses and the logits with an
independent CPU evaluation of the same step in float64 (tests/torch_ref.py: F.conv1d + F.ctc_loss + autograd,
which agrees with the numpy oracle to 1e-13, tests/test_oracle_conv_ctc.py::test_torch_ref_equals_oracle).
The launch trace of the library is asserted so that the test cannot silently take the small-problem kernels.

Tolerances: logits <= 1e-4 absolute, loss <= 1e-4 relative, d loss / d logits and every
gradient tensor <= 2e-4 of its own max -- kernel by kernel on the device's operands AND end to end against an
independent float64 evaluation with the ReLU pattern pinned to the device's (``compare``)."""
import time

import numpy as np
import pytest

from tests import torch_ref as TR
from tests import workloads as WL

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

FRAMES = [1001] * 30 + [777, 500]                          # ragged tail like a real batch


@pytest.fixture(scope='module')
def case():
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  layers = WL.w2l_layers(80)
  params = WL.xavier_params(layers, seed=42, dtype=np.float32)      # non-zero biases
  x, seq, labels = WL.make_batch(FRAMES, 80, seed=3)
  x32 = x.astype(np.float32)
  t0 = time.time()
  ref = TR.loss_and_grads(x32, seq, labels, params, layers, dtype=torch.float64)
  print('float64 CPU reference of the full-size step: %.1f s' % (time.time() - t0))
  return dict(layers=layers, params=params, x=x32, seq=seq, labels=labels, ref=ref)


def run_step(case, mode, fft_conv=False):
  from speecht_amd._lib import launch_trace
  from speecht_amd.engine import Wav2LetterEngine
  eng = Wav2LetterEngine(case['layers'], device='cuda:0', conv_mode=mode, fft_conv=fft_conv)
  eng.set_weights(case['params'])
  eng.load_batch(case['x'], case['seq'])
  eng.set_labels(case['labels'])
  with launch_trace() as tr:
    eng.forward()
    eng.ctc_loss_grad(1.0 / len(FRAMES))
    eng.backward()
  torch.cuda.synchronize()
  eng.check_ctc_status()
  return eng, tr.lines


def rel_err(g, r):
  return float(np.max(np.abs(g - r)) / np.max(np.abs(r)))


def compare(eng, ref, case, grad_tol=2e-4):
  """Three comparisons, from the most independent to the most exact:

  1. logits, per-utterance losses and d avg_loss / d logits against the float64 autograd reference;
  2. the BACKWARD KERNELS at full size: all 22 gradient tensors against float64 back-prop evaluated on the
     device's own stored activations and its own dlogits (tests/torch_ref.backward_from_acts) -- the exact linear
     map the kernels must reproduce, tolerance 2e-4 of each tensor's max;
  3. END TO END, all 22 tensors at the same 2e-4: against a second float64 evaluation of the whole step (its own
     forward activations from the inputs, its own CTC, autograd) whose ReLU pattern is pinned to the one the device
     took (tests/torch_ref.loss_and_grads(relu_masks=...)).  The step has one discontinuity -- a ReLU input within
     fp32 rounding of zero lands on the other side in float64 (1-10 of 4-32 million elements per layer; the flipped
     unit's whole gradient appears or vanishes and the ill-conditioned lower layers amplify it) -- and pinning the
     pattern removes exactly that and nothing else: the pinned pre-activations differ from the free ones by < 1e-6
     at the flipped elements (asserted through the logits), every other number is independent of the device.
     The un-pinned end-to-end figures are printed for the record."""
  logits = eng.logits_time_major().cpu().numpy()
  assert logits.shape == ref['logits'].shape == (501, 32, 29)
  # frames beyond an utterance's own length are computed too (nothing is masked): compare all
  err = float(np.max(np.abs(logits - ref['logits'])))
  assert err < 1e-4, err
  np.testing.assert_allclose(eng.loss.cpu().numpy(), ref['loss'], rtol=1e-4)
  dl = eng.dZ[-1].interior().cpu().numpy().astype(np.float64)
  dl_err = rel_err(dl, ref['dlogits'])
  assert dl_err < 2e-4, dl_err
  acts = [eng.X[i].interior().cpu().numpy() for i in range(len(eng.layers) + 1)]
  flips = [int(np.sum((acts[i] > 0) != (ref['acts'][i] > 0))) for i in range(1, len(eng.layers))]   # ReLU outputs
  got = eng.get_grads()
  exact, _ = TR.backward_from_acts(acts, case['params'], case['layers'], dl)
  t0 = time.time()
  pinned = TR.loss_and_grads(case['x'], case['seq'], case['labels'], case['params'], case['layers'], dtype=torch.float64,
                             relu_masks=TR.relu_masks_of(acts, case['layers']))
  pin_secs = time.time() - t0
  # the pinned evaluation is the same function as the free one except at the flipped units, whose pre-activations are
  # within rounding of zero: logits and losses agree far below the parity tolerance
  pin_shift = float(np.max(np.abs(pinned['logits'] - ref['logits'])))
  assert pin_shift < 1e-5, pin_shift
  assert float(np.max(np.abs(logits - pinned['logits']))) < 1e-4
  np.testing.assert_allclose(eng.loss.cpu().numpy(), pinned['loss'], rtol=1e-4)
  kernel_report, e2e_report, free_report, failed = [], [], [], []
  for i, ((gF, gb), (xF, xb), (pF, pb), (rF, rb)) in enumerate(zip(got, exact, pinned['grads'], ref['grads'])):
    for name, g, x, p, r in (('filters', gF, xF, pF, rF), ('bias', gb, xb, pb, rb)):
      assert g.shape == x.shape == p.shape == r.shape
      k, e, f = rel_err(g, x), rel_err(g, p), rel_err(g, r)
      kernel_report.append('L%d %s %.1e' % (i, name, k))
      e2e_report.append('L%d %s %.1e' % (i, name, e))
      free_report.append('L%d %s %.1e' % (i, name, f))
      if not k < grad_tol:
        failed.append('kernel L%d %s %.2e' % (i, name, k))
      if not e < grad_tol:
        failed.append('end-to-end (ReLU pattern pinned) L%d %s %.2e' % (i, name, e))
  print('dlogits error %.2e of max; ReLU sign flips vs float64 per layer output: %s' % (dl_err, flips))
  print('backward kernels vs float64 back-prop on the device activations: ' + '; '.join(kernel_report))
  print('end to end vs float64 autograd with the device\'s ReLU pattern (%.1f s, logits moved %.1e by the pinning): '
        % (pin_secs, pin_shift) + '; '.join(e2e_report))
  print('for the record, end to end vs the free float64 autograd (flipped units included): ' + '; '.join(free_report))
  assert not failed, failed
  return err, dl_err


def test_fullsize_fp32_gradients_match_float64_reference(case):
  eng, trace = run_step(case, 'fp32')
  text = '\n'.join(trace)
  # the benchmark's kernels, not the small-problem ones
  fwd_fast = [l for l in trace if l.startswith('gemm_nn<128,128,2,2,fast> epi=0')]
  bwd_fast = [l for l in trace if l.startswith(('gemm_nn<128,128,2,2,fast> epi=1', 'gemm_nn<128,128,2,2,fast-bt> epi=1'))]
  assert len(fwd_fast) == 9, text                                  # L1..L9 forward
  assert len(bwd_fast) == 10, text                                 # back-prop to the input of L10..L1
  assert sum('fast-bt' in l for l in bwd_fast) == 2, text          # the two 1-tap layers read their packed filters transposed
  assert any('splits=2' in l and 'Kp=64512' in l for l in bwd_fast), text        # L8 back-prop: 2 K-halves
  slabbed = [l for l in trace if l.startswith('gemm_tn<') and 'slabs=1 ' not in l]
  assert len(slabbed) >= 9 and all('M=16032' in l for l in trace if l.startswith('gemm_tn<')), text
  compare(eng, case['ref'], case)


def test_fullsize_fp32_frequency_domain_l8_gradients_match_float64_reference(case):
  """The default fp32 step: nine of the eleven layers run in the frequency domain (csrc/conv_fft.hip) -- block DFTs,
  per-bin GEMMs on the fp32 MFMA kernels (batched launches in the trace: 48 bins for the 32-tap layer, 36 for the 7-tap
  layers, 45 for the stride-2 first layer on its polyphase view), fused inverse + epilogue, filter-gradient chains of
  the narrow layers on the side stream.  Same three comparisons and the same tolerances as the W-tap kernels."""
  eng, trace = run_step(case, 'fp32', fft_conv=True)
  text = '\n'.join(trace)
  # L8 (48 bins): forward and back-prop products on the batched convolution kernel, lag products on the batched
  # filter-gradient kernel; the seven 7-tap layers (36 bins) likewise
  # (round 6: the 32-tap layer's complex products in three real products -- gemm_nn_g3 / gemm_tn_g3, conv_fft.hip g3_form)
  assert sum(1 for l in trace if l.startswith('gemm_nn_g3<') and ' batched bins=48 ' in l) == 2, text
  assert sum(1 for l in trace if l.startswith('gemm_tn_g3<') and ' batched bins=48 M=256 ' in l) == 1, text   # the lag products of the 48 bins
  assert sum(1 for l in trace if l.startswith(('gemm_nn<', 'gemm_nn_bins<')) and ' batched bins=36 ' in l) == 14, text
  assert sum(1 for l in trace if l.startswith('gemm_tn<') and ' batched bins=72 M=512 ' in l) == 7, text
  assert not any('Kp=8192' in l or 'Kp=64512' in l or 'Kp=1792' in l for l in trace if 'batched' not in l), text   # no W-tap launch of L1..L8
  # the stride-2 first layer on its polyphase view (25 taps over 2 x 80 channels, 45 bins): forward + filter gradient
  assert sum(1 for l in trace if l.startswith(('gemm_nn<', 'gemm_nn_bins<')) and ' batched bins=45 ' in l) == 1, text
  assert sum(1 for l in trace if l.startswith('gemm_tn<') and ' batched bins=45 ' in l) == 1, text
  assert not any('Kp=3840' in l for l in trace if 'batched' not in l), text
  compare(eng, case['ref'], case)


def test_fullsize_bf16x6_gradients_match_float64_reference(case):
  eng, trace = run_step(case, 'bf16x6')
  assert sum(1 for l in trace if l.startswith('gemm_nn_bf16<256,NP=3>')) >= 4, '\n'.join(trace)
  compare(eng, case['ref'], case)


def test_fullsize_bf16x6_with_frequency_domain_l8(case):
  """bf16x6 mode as it runs by default: the frequency-domain (fp32) L8 between bf16x6 layers -- the planes of its
  neighbours are split from its fp32 outputs."""
  eng, trace = run_step(case, 'bf16x6', fft_conv=True)
  assert sum(1 for l in trace if l.startswith('gemm_nn_g3<') and ' batched bins=48 ' in l) == 2 and \
      sum(1 for l in trace if l.startswith('gemm_tn_g3<') and ' batched bins=48 ' in l) == 1 and \
      any(l.startswith('gemm_nn_bf16<256,NP=3>') for l in trace)
  compare(eng, case['ref'], case)


def test_fullsize_update_matches_reference_adam(case):
  """clip_by_global_norm(5) + TF-Adam at full size: the global norm and the first update of every tensor against
  the oracle's optimizer (speech_model.py:77-82) fed the device's own gradients (whose parity is the tests above)."""
  from oracle import w2l_oracle as O
  eng, _ = run_step(case, 'fp32', fft_conv=True)
  flat = [g.astype(np.float64) for pair in eng.get_grads() for g in pair]
  clipped, gn = O.clip_by_global_norm(flat, 5.0)
  eng.apply_update(lr=1e-4)
  torch.cuda.synchronize()
  assert float(eng.stats[0]) == pytest.approx(gn, rel=1e-5)
  new = eng.get_weights()
  for i, (F, b) in enumerate(case['params']):
    for j, (p0, got) in enumerate(((F, new[i][0]), (b, new[i][1]))):
      g = clipped[2 * i + j]
      want, _, _ = O.adam_tf_step(p0.astype(np.float64), g, np.zeros_like(g), np.zeros_like(g), 1, 1e-4)
      # the first Adam step moves every weight by lr * g/(|g| + eps*sqrt(1-b2)/(1-b1))-ish: compare the DELTA
      d_got, d_want = got.astype(np.float64) - p0, want - p0
      # + one fp32 rounding of the stored weight (the delta can be smaller than an ulp of the weight)
      assert np.max(np.abs(d_got - d_want)) < 1e-4 * np.max(np.abs(d_want)) + 1.2e-7 * np.max(np.abs(p0)) + 1e-12, (i, j)
