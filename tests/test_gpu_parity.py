"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on seeded inputs.

Tolerances (fp32 path vs float64 oracle; BASELINE.json north_star): logits |err| <= 1e-4
absolute, CTC loss <= 1e-4 relative, gradients <= 2e-4 of the tensor's max magnitude, greedy
decode strings bit-identical.
"""
import math

import numpy as np
import pytest

from oracle import w2l_oracle as O
from tests import workloads as WL

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  return 'cuda:0'


def make_engine(layers, dev, conv_mode=None):
  from speecht_amd.engine import Wav2LetterEngine
  return Wav2LetterEngine(layers, device=dev, conv_mode=conv_mode)


def rel_err(a, b):
  return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


CONV_CASES = [
    # B, T, W, s, cin, cout, relu
    (3, 41, 48, 2, 16, 24, True),      # L0-like, odd T -> pad (23,24)
    (2, 40, 48, 2, 80, 250, True),     # L0 real channels, even T -> pad (23,23)
    (2, 37, 7, 1, 250, 250, True),     # L1-7 real channels (250 -> pitch 256)
    (2, 45, 32, 1, 24, 40, True),      # L8-like asymmetric pad (15,16)
    (1, 33, 32, 1, 250, 2000, True),   # L8 real channels, n_pad 2048 > pitch 2000
    (2, 29, 1, 1, 2000, 2000, True),   # L9 real, K=2000 -> k_pad 2016
    (3, 21, 1, 1, 2000, 29, False),    # L10 real, n_pad 32
    (8, 150, 1, 1, 2000, 29, False),   # L10 on 1 200 rows: the reduction in slices (round 6: short batches too; 10 tiles -> 8 slices)
    (5, 131, 7, 1, 40, 40, True),      # several 128-row tiles, n_pad 64
    (1, 3, 7, 1, 16, 16, True),        # fewer frames than the filter width
]


@pytest.mark.parametrize('B,T,W,s,cin,cout,relu', CONV_CASES)
def test_conv_fwd_bwd(dev, B, T, W, s, cin, cout, relu):
  rng = np.random.default_rng(B * 1000 + T)
  x = rng.standard_normal((B, T, cin))
  F = rng.standard_normal((W, cin, cout)) * (1.0 / math.sqrt(W * cin))
  b = rng.standard_normal(cout) * 0.1
  eng = make_engine([(W, s, cin, cout, relu)], dev)
  eng.set_weights([(F, b)])
  eng.load_batch(x, [T] * B)
  eng.forward()
  y = eng.X[1].interior().cpu().numpy()
  yref = O.conv1d_same_fwd(x, F, b, s, relu)
  assert y.shape == yref.shape
  assert np.max(np.abs(y - yref)) < 2e-5 * max(1.0, np.max(np.abs(yref)))
  # halos and pad channels must stay zero
  full = eng.X[1].buf.view(B, eng.X[1].t_pitch, eng.X[1].c_pitch).cpu().numpy()
  assert np.all(full[:, :, cout:] == 0)
  # backward: feed dz (gradient wrt the pre-activation) directly
  dy = rng.standard_normal(yref.shape)
  dz = dy * (yref > 0) if relu else dy
  eng.dZ[0].interior().copy_(torch.as_tensor(dz, dtype=torch.float32))
  eng.backward()
  (gF, gb), = eng.get_grads()
  _, dF, db = O.conv1d_same_bwd(x, F, yref, dy, s, relu, need_dx=False)
  assert rel_err(gF, dF) < 2e-5
  assert rel_err(gb, db) < 2e-5
  # padded part of the flat gradient must be exactly zero (it enters the global norm)
  gsum = float(eng.grads.double().pow(2).sum())
  assert gsum == pytest.approx(float((gF.astype(np.float64) ** 2).sum() + (gb.astype(np.float64) ** 2).sum()), rel=1e-6)


@pytest.mark.parametrize('B,T,W,s,cin,cout', [(1, 101, 32, 1, 250, 2000), (1, 201, 48, 2, 80, 250),
                                              (3, 90, 7, 1, 250, 250), (1, 5, 1, 1, 2000, 2000)])
def test_conv_fwd_split_reduction_for_few_rows(dev, B, T, W, s, cin, cout):
  """Single-utterance shapes leave most CUs without an output tile; the forward then splits the reduction
  (st_conv1d_nwc_fwd_ws_f32).  Same result as the unsplit launch up to fp32 summation order, same oracle bound,
  deterministic, and the workspace-less entry point still works."""
  from speecht_amd import _lib
  rng = np.random.default_rng(T)
  x = rng.standard_normal((B, T, cin))
  F = rng.standard_normal((W, cin, cout)) * (1.0 / math.sqrt(W * cin))
  b = rng.standard_normal(cout) * 0.1
  eng = make_engine([(W, s, cin, cout, True)], dev)
  eng.set_weights([(F, b)])
  eng.load_batch(x, [T] * B)
  assert _lib.load().st_conv1d_fwd_ws(eng.X[0].ref, eng.X[1].ref, W) > 0
  eng.forward()
  y_split = eng.X[1].buf.clone()
  eng.forward()
  assert torch.equal(y_split, eng.X[1].buf)
  l = eng.layers[0]
  pf, pb = eng._slice(eng.params, 0)
  _lib.call('st_conv1d_nwc_fwd_f32', eng.X[0].ref, eng._ptr(pf), eng._ptr(pb), W, s, eng.geo[0][2], 1, eng.X[1].ref,
            eng.stream_ptr)
  y_one = eng.X[1].buf.clone()
  yref = O.conv1d_same_fwd(x, F, b, s, True)
  scale = max(1.0, float(np.max(np.abs(yref))))
  assert float((y_split - y_one).abs().max()) < 1e-5 * scale
  eng.X[1].buf.copy_(y_split)
  assert np.max(np.abs(eng.X[1].interior().cpu().numpy() - yref)) < 2e-5 * scale
  full = y_split.view(B, eng.X[1].t_pitch, eng.X[1].c_pitch).cpu().numpy()
  assert np.all(full[:, :, cout:] == 0)


def test_conv_bwd_data_with_relu_mask(dev):
  rng = np.random.default_rng(3)
  layers = [(7, 1, 16, 250, True), (32, 1, 250, 40, True), (1, 1, 40, 29, False)]
  params = WL.xavier_params(layers, seed=5)
  B, T = 2, 57
  x = rng.standard_normal((B, T, 16))
  eng = make_engine(layers, dev)
  eng.set_weights(params)
  eng.load_batch(x, [T] * B)
  eng.forward()
  logits, acts = O.wav2letter_forward(x, params, layers, keep=True)
  assert np.max(np.abs(eng.logits_time_major().cpu().numpy() - logits)) < 1e-5
  dl = rng.standard_normal(logits.shape)
  eng.dZ[-1].interior().copy_(torch.as_tensor(np.transpose(dl, (1, 0, 2)), dtype=torch.float32))
  eng.backward()
  ref = O.wav2letter_backward(acts, params, layers, dl)
  for (gF, gb), (rF, rb) in zip(eng.get_grads(), ref):
    assert rel_err(gF, rF) < 2e-5 and rel_err(gb, rb) < 2e-5


def _ctc_case(rng, B, T, C, lengths, lens=None):
  logits = (rng.standard_normal((T, B, C)) * 2.0).astype(np.float32).astype(np.float64)   # exactly what the GPU gets
  labels = [rng.integers(0, C - 1, L).tolist() for L in lengths]
  lens = np.full(B, T) if lens is None else np.asarray(lens)
  return logits, labels, lens


def run_ctc(dev, logits_tm, labels, lens, scale=1.0):
  T, B, C = logits_tm.shape
  eng = make_engine([(1, 1, 16, C, False)], dev)
  eng.load_batch(np.zeros((B, T, 16)), [T] * B)
  eng.X[-1].interior().copy_(torch.as_tensor(np.transpose(logits_tm, (1, 0, 2)), dtype=torch.float32))
  eng.ctc_lens = torch.as_tensor(np.asarray(lens, dtype=np.int32)).to(dev)
  eng.set_labels(labels)
  eng.ctc_loss_grad(scale)
  torch.cuda.synchronize()
  grad = np.transpose(eng.dZ[-1].interior().cpu().numpy(), (1, 0, 2))
  return eng, eng.loss.cpu().numpy(), grad


@pytest.mark.parametrize('T,lengths', [
    (50, [0, 1, 7, 20]),             # KPL 1 (U <= 64): empty label, single label
    (120, [31, 32, 45, 3]),          # KPL 1/2 boundary (U = 63, 65)
    (300, [150, 95, 140]),           # KPL 5: the bench shape (L = 150 -> U = 301)
    (260, [130, 127, 128]),          # exact lane boundaries
    (700, [330, 200]),               # KPL 12
])
def test_ctc_loss_grad(dev, T, lengths):
  rng = np.random.default_rng(T)
  B = len(lengths)
  lens = [T - 3 * i for i in range(B)]           # ragged
  logits, labels, lens = _ctc_case(rng, B, T, 29, lengths, lens)
  if lengths[0] >= 5:
    labels[0][1] = labels[0][0]                  # force repeats
    labels[0][3] = labels[0][2]
  ref_loss, ref_grad = O.ctc_loss_and_grad(logits, labels, lens)
  eng, loss, grad = run_ctc(dev, logits, labels, lens, scale=0.5)
  assert not eng.ctc_status.cpu().numpy().any()
  np.testing.assert_allclose(loss, ref_loss, rtol=1e-5)
  assert np.max(np.abs(grad - 0.5 * ref_grad)) < 5e-5
  for b in range(B):
    assert np.all(grad[lens[b]:, b] == 0)


def test_ctc_with_masked_classes_and_the_loss_as_a_float_pair(dev):
  """Logits of -inf (a masked class): a state killed by a dead emission must go back to the zero state, not keep
  a zero mantissa under a live-scale exponent that wins the next alignment and flushes live neighbours.  Whole classes
  masked for long runs, from the first frame, on states far from the start (unreachable AND dead for dozens of frames), on
  the blank; loss and gradient against the float64 oracle.  And the loss pair: hi is the fp32 loss, hi + lo agrees with the
  oracle's -log p far below one fp32 ulp."""
  rng = np.random.default_rng(77)
  T, C = 240, 29
  lengths = [60, 90, 30, 100]
  B = len(lengths)
  logits, labels, lens = _ctc_case(rng, B, T, C, lengths)
  labels = [[int(v) % 20 for v in l] for l in labels]          # classes 20..27 never occur in a label
  logits[:, 0, 20:28] = -np.inf                                # unused classes masked for the whole utterance
  logits[:40, 1, 25] = -np.inf                                 # ... from the first frame on for a while
  logits[100:180, 2, 22] = -np.inf
  logits[5:25, 3, labels[3][50]] = -np.inf                     # a class the label USES, dead while its states are still unreachable
  logits[0:3, 1, 28] = -np.inf                                 # the blank itself at the start: the path must open with the first label
  ref_loss, ref_grad = O.ctc_loss_and_grad(logits, labels, lens)
  assert np.all(np.isfinite(ref_loss))
  eng, loss, grad = run_ctc(dev, logits, labels, lens, scale=1.0)
  assert not eng.ctc_status.cpu().numpy().any()
  np.testing.assert_allclose(loss, ref_loss, rtol=1e-5)
  assert np.all(np.isfinite(grad)) and np.max(np.abs(grad - ref_grad)) < 5e-5
  precise = eng.losses_precise()
  assert np.max(np.abs(precise - ref_loss)) < 2e-5, (precise - ref_loss)
  assert np.array_equal(loss, precise.astype(np.float32))       # hi is the rounded double, lo the remainder


def test_ctc_closed_forms_and_errors(dev):
  C = 29
  T = 7
  _, loss, grad = run_ctc(dev, np.zeros((T, 1, C)), [[]], [T])
  assert loss[0] == pytest.approx(T * math.log(C), rel=1e-5)
  _, loss, _ = run_ctc(dev, np.zeros((6, 1, C)), [[2]], [6])
  assert loss[0] == pytest.approx(-math.log(6 * 7 / 2 / C ** 6), rel=1e-5)
  _, loss, _ = run_ctc(dev, np.zeros((3, 1, C)), [[1, 1]], [3])
  assert loss[0] == pytest.approx(3 * math.log(C), rel=1e-5)
  # "aa" cannot be emitted in 2 frames: TF raises InvalidArgument; we flag + raise in the host layer
  eng, loss, grad = run_ctc(dev, np.zeros((2, 2, C)), [[1, 1], [3]], [2, 2])
  assert eng.ctc_status.cpu().numpy().tolist() == [1, 0]
  assert math.isinf(loss[0]) and np.all(grad[:, 0] == 0) and np.isfinite(loss[1])
  with pytest.raises(ValueError):
    eng.check_ctc_status()


def test_greedy_decode_bit_identical(dev):
  rng = np.random.default_rng(9)
  T, B, C = 301, 6, 29
  logits = rng.standard_normal((T, B, C)).astype(np.float32)
  logits[:, 0, :] = np.round(logits[:, 0, :])          # many exact ties -> lowest index must win
  logits[:, 1, 28] += 3.0                              # mostly blanks
  logits[:, 2, :] = 0.0                                # all ties: 'a' every frame -> one 'a'
  logits[10:200, 3, 5] = 9.0                           # long run merges
  lens = np.array([301, 300, 17, 250, 1, 150])
  eng = make_engine([(1, 1, 16, C, False)], dev)
  eng.load_batch(np.zeros((B, T, 16)), [T] * B)
  eng.X[-1].interior().copy_(torch.as_tensor(np.transpose(logits, (1, 0, 2))))
  eng.ctc_lens = torch.as_tensor(lens.astype(np.int32)).to(dev)
  for merge in (True, False):
    ids, score = eng.greedy_decode(merge)
    ref_ids, ref_score = O.ctc_greedy_decode(logits.astype(np.float64), lens, merge)
    assert ids == ref_ids
    np.testing.assert_allclose(score, ref_score, rtol=1e-5)


@pytest.mark.parametrize('beam', [1, 4, 16, 24, 40, 64])
def test_beam_search_matches_oracle(dev, beam):
  """st_ctc_beam_search_decode vs the float64 prefix beam search of the oracle: identical label
  sequences, log-probabilities within 1e-4 relative (fp32 log-sum-exp over up to 401 frames)."""
  rng = np.random.default_rng(40 + beam)
  T, B, C = 401, 8, 29
  logits = (rng.standard_normal((T, B, C)) * 2.0).astype(np.float32)
  logits[:, 1, 28] += 4.0                                # mostly blanks
  logits[:, 2, :] *= 4.0                                 # sharply peaked: beam collapses onto greedy
  logits[50:300, 3, 7] += 6.0                            # one long run of a single label
  logits[:, 4, :] *= 0.1
  logits[::2, 4, 11] += 9.0; logits[1::2, 4, 28] += 9.0  # l, blank, l, blank ... -> repeated labels
  lens = np.array([401, 400, 333, 301, 200, 1, 0, 57])
  eng = make_engine([(1, 1, 16, C, False)], dev)
  eng.load_batch(np.zeros((B, T, 16)), [T] * B)
  eng.X[-1].interior().copy_(torch.as_tensor(np.transpose(logits, (1, 0, 2))))
  eng.ctc_lens = torch.as_tensor(lens.astype(np.int32)).to(dev)
  ids, logp = eng.beam_search_decode(beam)
  ref_ids, ref_logp = O.ctc_beam_search_decode(logits.astype(np.float64), lens, beam)
  assert ids == ref_ids
  np.testing.assert_allclose(logp, ref_logp, rtol=1e-4, atol=1e-4)
  assert ids[6] == [] and logp[6, 0] == 0.0              # zero frames: empty prefix with probability 1
  if beam >= 16:
    assert ids[4] == [11] * 100                          # repeats separated by blanks survive (200 frames)


def test_beam_search_long_form(dev):
  """Config 5 length (30 s -> T' = 1501), beam 16, labels on almost every frame: absolute log-probabilities
  reach -3000, where fp32 could no longer order near-ties -- the kernel keeps scores relative to the best
  entry (double offset), so the label sequences still match the float64 oracle exactly."""
  rng = np.random.default_rng(77)
  T, B, C = 1501, 2, 29
  logits = (rng.standard_normal((T, B, C)) * 3.0).astype(np.float32)
  lens = np.array([1501, 1203])
  eng = make_engine([(1, 1, 16, C, False)], dev)
  eng.load_batch(np.zeros((B, T, 16)), [T] * B)
  eng.X[-1].interior().copy_(torch.as_tensor(np.transpose(logits, (1, 0, 2))))
  eng.ctc_lens = torch.as_tensor(lens.astype(np.int32)).to(dev)
  ids, logp = eng.beam_search_decode(16)
  ref_ids, ref_logp = O.ctc_beam_search_decode(logits.astype(np.float64), lens, 16)
  assert ids == ref_ids
  np.testing.assert_allclose(logp, ref_logp, rtol=1e-5)


def test_beam_search_massive_ties(dev):
  """All-equal logits: hundreds of candidates tie exactly, more than the 64 survivors the wave-parallel
  selection holds, so the sequential rounds take over.  Exact ties are ordered by candidate index, which makes
  the result well defined: deterministic across runs and equal to the oracle's (whose float64 sums tie too)."""
  T, B, C = 40, 2, 29
  logits = np.zeros((T, B, C), dtype=np.float32)
  lens = np.array([40, 17])
  eng = make_engine([(1, 1, 16, C, False)], dev)
  eng.load_batch(np.zeros((B, T, 16)), [T] * B)
  eng.X[-1].interior().copy_(torch.as_tensor(np.transpose(logits, (1, 0, 2))))
  eng.ctc_lens = torch.as_tensor(lens.astype(np.int32)).to(dev)
  ids1, logp1 = eng.beam_search_decode(16)
  ids2, logp2 = eng.beam_search_decode(16)
  assert ids1 == ids2 and np.array_equal(logp1, logp2)
  assert all(0 <= v < 28 for seq in ids1 for v in seq) and len(ids1[0]) <= 40 and len(ids1[1]) <= 17
  ref_ids, ref_logp = O.ctc_beam_search_decode(logits.astype(np.float64), lens, 16)
  np.testing.assert_allclose(logp1, ref_logp, rtol=1e-5)


def test_beam_search_rejects_bad_arguments(dev):
  eng = make_engine([(1, 1, 16, 29, False)], dev)
  eng.load_batch(np.zeros((2, 9, 16)), [9, 9])
  eng.ctc_lens = torch.as_tensor(np.array([9, 9], dtype=np.int32)).to(dev)
  from speecht_amd._lib import SpeechtHipError
  with pytest.raises(SpeechtHipError, match='beam width'):
    eng.beam_search_decode(129)
  with pytest.raises(ValueError, match='input_transform'):
    eng.beam_search_decode(16, input_transform='softmax')


def _beam_case(dev, logits, lens):
  T, B, C = logits.shape
  eng = make_engine([(1, 1, 16, C, False)], dev)
  eng.load_batch(np.zeros((B, T, 16)), [T] * B)
  eng.X[-1].interior().copy_(torch.as_tensor(np.transpose(logits, (1, 0, 2))))
  eng.ctc_lens = torch.as_tensor(np.asarray(lens).astype(np.int32)).to(dev)
  return eng


@pytest.mark.parametrize('beam,B', [(65, 2), (100, 4), (128, 3)])
def test_wide_beam_search_matches_oracle(dev, beam, B):
  """Beams wider than the wavefront (the reference runs beam_width=100, speech_model.py:109): two beam entries per lane, the
  classes on a pitch of 32 and the exact W-th largest score as the selection bound (ctc_beam_kernel<64, 128>) -- identical
  label sequences and log-probabilities within 1e-4 of the float64 oracle search, on plain logits and on the reference's
  decoder input log10(softmax + 1e-8); the launch trace says which instantiation ran."""
  from speecht_amd._lib import launch_trace
  rng = np.random.default_rng(400 + beam)
  T, C = 121, 29
  logits = (rng.standard_normal((T, B, C)) * 2.0).astype(np.float32)
  logits[:, 0, 28] += 3.0                                # mostly blanks
  if B > 2:
    logits[:, 2, :] *= 0.1
    logits[::2, 2, 11] += 9.0; logits[1::2, 2, 28] += 9.0  # l, blank, l, blank ... -> repeated labels
  lens = np.array([121, 97, 120, 33][:B])
  eng = _beam_case(dev, logits, lens)
  for transform in (None, 'log10_softmax'):
    with launch_trace() as tr:
      ids, logp = eng.beam_search_decode(beam, input_transform=transform)
    assert any(l.startswith('ctc_beam<wide> beam=%d transform=%d' % (beam, 1 if transform else 0)) for l in tr.lines), tr.lines
    ref_ids, ref_logp = O.ctc_beam_search_decode(logits.astype(np.float64), lens, beam, input_transform=transform)
    assert ids == ref_ids, transform
    np.testing.assert_allclose(logp, ref_logp, rtol=1e-4, atol=1e-4)
  if B > 2:
    assert len(ids[2]) > 50 and set(ids[2]) == {11}      # repeats separated by blanks survive; merge_repeated collapses them
    merged, _ = eng.beam_search_decode(beam, input_transform='log10_softmax', merge_repeated=True)
    assert merged[2] == [11] and all(a != b for seq in merged for a, b in zip(seq, seq[1:]))


def test_wide_beam_search_massive_ties_take_the_sequential_rounds(dev):
  """All-equal logits at beam 100: thousands of candidates tie exactly, more than the 256 survivors the wave-parallel selection
  of the wide instantiation holds -- the sequential rounds order them by candidate index; deterministic and equal to the
  oracle's probabilities."""
  T, B, C = 24, 2, 29
  logits = np.zeros((T, B, C), dtype=np.float32)
  lens = np.array([24, 9])
  eng = _beam_case(dev, logits, lens)
  ids1, logp1 = eng.beam_search_decode(100)
  ids2, logp2 = eng.beam_search_decode(100)
  assert ids1 == ids2 and np.array_equal(logp1, logp2)
  ref_ids, ref_logp = O.ctc_beam_search_decode(logits.astype(np.float64), lens, 100)
  np.testing.assert_allclose(logp1, ref_logp, rtol=1e-5)
  assert ids1 == ref_ids


def test_reference_operating_point_beam_100_on_long_form(dev):
  """The reference's decoder call without its KenLM scorer (speech_model.py:101-111): beam_width=100, merge_repeated=False,
  input log10(softmax(logits) + 1e-8), on a 30 s utterance (T' = 1501): label sequence identical to the float64 oracle search,
  log-probability within 1e-5 relative (scores kept relative to the best entry with a double offset)."""
  rng = np.random.default_rng(78)
  T, B, C = 1501, 1, 29
  logits = (rng.standard_normal((T, B, C)) * 3.0).astype(np.float32)
  lens = np.array([1501])
  eng = _beam_case(dev, logits, lens)
  ids, logp = eng.beam_search_decode(100, input_transform='log10_softmax')
  ref_ids, ref_logp = O.ctc_beam_search_decode(logits.astype(np.float64), lens, 100, input_transform='log10_softmax')
  assert ids == ref_ids and len(ids[0]) > 1000
  np.testing.assert_allclose(logp, ref_logp, rtol=1e-5)


@pytest.mark.parametrize('n,clip', [(1000, 5.0), (1 << 20, 5.0), (4099, 0.01)])
def test_clip_adam(dev, n, clip):
  from speecht_amd import _lib
  import ctypes
  rng = np.random.default_rng(n)
  p = rng.standard_normal(n); g = rng.standard_normal(n) * (0.001 if clip == 5.0 and n == 1000 else 1.0)
  m = rng.standard_normal(n) * 0.1; v = rng.random(n) * 0.01
  t = [torch.as_tensor(a, dtype=torch.float32).to(dev) for a in (p, g, m, v)]
  stats = torch.zeros(2, device=dev)
  ws = torch.zeros(2048, device=dev)
  step, lr = 3, 1e-3
  lr_t = lr * math.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
  P = lambda x: ctypes.c_void_p(x.data_ptr())
  _lib.call('st_global_norm_clip_adam_f32', P(t[0]), P(t[1]), P(t[2]), P(t[3]), n, clip, lr_t, 0.9, 0.999, 1e-3,
            P(stats), P(ws), ws.numel() * 4, None)
  torch.cuda.synchronize()
  g32, p32, m32, v32 = (a.astype(np.float32).astype(np.float64) for a in (g, p, m, v))
  clipped, gn = O.clip_by_global_norm([g32], clip)
  pr, mr, vr = O.adam_tf_step(p32, clipped[0], m32, v32, step, lr)
  assert float(stats[0]) == pytest.approx(gn, rel=1e-5)
  assert float(stats[1]) == pytest.approx(clip / max(gn, clip), rel=1e-5)
  np.testing.assert_allclose(t[0].cpu().numpy(), pr, rtol=2e-5, atol=1e-6)
  np.testing.assert_allclose(t[2].cpu().numpy(), mr, rtol=2e-5, atol=1e-7)
  np.testing.assert_allclose(t[3].cpu().numpy(), vr, rtol=2e-5, atol=1e-9)


def test_small_train_step_vs_oracle_and_golden(dev, golden_dir):
  import os
  case = WL.small_train_case()
  gold = np.load(os.path.join(golden_dir, 'w2l_small_golden.npz'))
  eng = make_engine(case['layers'], dev)
  eng.set_weights(case['params'])
  eng.load_batch(case['x'], case['seq_lens'])
  eng.set_labels(case['labels'])
  eng.forward()
  eng.ctc_loss_grad(1.0 / 3)
  eng.backward()
  grads = eng.get_grads()
  eng.apply_update(lr=1e-4)
  dec, score = eng.greedy_decode()
  torch.cuda.synchronize()
  ref = O.train_step(case['x'], case['seq_lens'], case['labels'], case['params'], case['layers'],
                     O.zero_opt_state(case['params']), lr=1e-4)
  logits = eng.logits_time_major().cpu().numpy()
  assert np.max(np.abs(logits - ref['logits'])) < 1e-4
  assert np.max(np.abs(logits - gold['logits'])) < 1e-4
  loss = eng.loss.cpu().numpy()
  np.testing.assert_allclose(loss, ref['loss'], rtol=1e-4)
  assert float(loss.mean()) == pytest.approx(float(gold['avg_loss']), rel=1e-4)
  assert float(eng.stats[0]) == pytest.approx(ref['grad_norm'], rel=1e-4)
  for i, ((gF, gb), (rF, rb)) in enumerate(zip(grads, ref['grads'])):
    assert rel_err(gF, rF) < 2e-4, i
    assert rel_err(gb, rb) < 2e-4, i
    np.testing.assert_allclose(gb, gold['gb%d' % i], atol=2e-4 * np.max(np.abs(gold['gb%d' % i])))
  for (pF, pb), (rF, rb) in zip(eng.get_weights(), ref['params']):
    assert np.max(np.abs(pF - rF)) < 2e-6 and np.max(np.abs(pb - rb)) < 2e-6
  ref_dec, _ = O.ctc_greedy_decode(ref['logits'], case['seq_lens'] // 2)
  assert dec == ref_dec
  gold_dec = [[int(v) for v in row if v >= 0] for row in gold['decoded']]
  assert dec == gold_dec


def test_full_width_forward_logits(dev):
  """Real Wav2Letter widths (80-mel, 250/2000 channels), B=2 ragged, ~2 s clips."""
  layers = WL.w2l_layers(80)
  params = WL.xavier_params(layers, seed=42)
  x, seq_lens, labels = WL.make_batch([201, 160], 80, seed=2)
  eng = make_engine(layers, dev)
  eng.set_weights(params)
  eng.load_batch(x, seq_lens)
  eng.forward()
  logits = eng.logits_time_major().cpu().numpy()
  ref = O.wav2letter_forward(x, params, layers)
  assert logits.shape == ref.shape == (101, 2, 29)
  assert np.max(np.abs(logits - ref)) < 1e-4
  ids, _ = eng.greedy_decode()
  ref_ids, _ = O.ctc_greedy_decode(ref, seq_lens // 2)
  assert ids == ref_ids


def test_full_width_forward_logits_128_mel(dev):
  """The reference's default feature width (n_mels = 128, speecht-cli:53): L0 then has a 128-channel input,
  whose k-tiles are all whole (the unclamped-address kernel variant), unlike the 80-mel benchmark model."""
  layers = WL.w2l_layers(128)
  params = WL.xavier_params(layers, seed=11)
  x, seq_lens, _ = WL.make_batch([150, 97], 128, seed=5)
  eng = make_engine(layers, dev)
  eng.set_weights(params)
  eng.load_batch(x, seq_lens)
  eng.forward()
  logits = eng.logits_time_major().cpu().numpy()
  ref = O.wav2letter_forward(x, params, layers)
  assert logits.shape == ref.shape == (75, 2, 29)
  assert np.max(np.abs(logits - ref)) < 1e-4


@pytest.mark.parametrize('n_mels,sr', [(80, 16000), (128, 22050)])
def test_melspec_vs_oracle(dev, golden_dir, n_mels, sr):
  """calc_power_spectrogram (preprocessing.py:36-58): ragged batch, odd lengths; the features are
  z-normalised dB values (std 1), tolerance 1e-3 absolute (fp32 FFT + log10 vs float64)."""
  import os
  from speecht_amd.preprocessing import calc_power_spectrogram, calc_power_spectrogram_batch
  # noise clips (no element reaches power_to_db's 80 dB floor: mean / std come in closed form from the sums the FFT kernel
  # kept, the statistics pass exits at once), a pure tone and a clip that ends in digital silence (most elements ON the floor:
  # the statistics pass runs) -- both branches of mel_stats_kernel / mel_finish_kernel in one batch
  audio = [O.synthetic_audio(7, 16000 + 77), O.synthetic_audio(8, 32000), O.synthetic_audio(9, 5003),
           np.sin(2 * np.pi * 440.0 * np.arange(12345) / sr).astype(np.float32),
           np.concatenate([O.synthetic_audio(10, 8000), np.zeros(8000, np.float32)])]
  dyn = [O.calc_power_spectrogram(a, sr, n_mels=n_mels) for a in audio]
  feats = calc_power_spectrogram_batch(audio, sr, n_mels=n_mels)
  for a, f, ref in zip(audio, feats, dyn):
    assert f.shape == ref.shape == (1 + len(a) // 160, n_mels)
    assert np.max(np.abs(f - ref)) < 1e-3
  single = calc_power_spectrogram(audio[0], sr, n_mels=n_mels)
  np.testing.assert_array_equal(single, feats[0])
  if n_mels == 80 and sr == 16000:
    gold = np.load(os.path.join(golden_dir, 'w2l_small_golden.npz'))['mel80']
    assert np.max(np.abs(feats[0] - gold)) < 1e-3


@pytest.mark.parametrize('n_mels', [13, 40, 256])
def test_melspec_filterbank_extremes_and_planned_form(dev, n_mels):
  """Few wide filters (13 mels: up to 60 bins = 4 plan pieces each, summed in plan order), many narrow ones
  (256 mels: 9+ projection rounds), an odd frame count (the last frame pair is half empty), and the planned entry
  points (st_melspec_plan_f32 once + st_melspec_planned_f32) bit-identical to the one-call form, run to run."""
  import ctypes
  from speecht_amd import _lib
  from speecht_amd.preprocessing import calc_power_spectrogram_batch, mel_filterbank
  sr = 16000
  audio = [O.synthetic_audio(21, 16000), O.synthetic_audio(22, 8000 + 159), O.synthetic_audio(23, 4000)]   # 101, 51, 26 frames
  feats = calc_power_spectrogram_batch(audio, sr, n_mels=n_mels)
  for a, f in zip(audio, feats):
    ref = O.calc_power_spectrogram(a, sr, n_mels=n_mels)
    assert f.shape == ref.shape and np.max(np.abs(f - ref)) < 1e-3
  lens = np.array([len(a) for a in audio], dtype=np.int64)
  s_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
  f_off = np.concatenate([[0], np.cumsum(1 + lens // 160)]).astype(np.int64)
  total = int(f_off[-1])
  d = lambda a: torch.as_tensor(a).to(dev)
  x, so, fo = d(np.concatenate(audio)), d(s_off), d(f_off)
  basis = d(mel_filterbank(float(sr), 512, n_mels).astype(np.float32)).contiguous()
  lib = _lib.load()
  plan = torch.empty(lib.st_melspec_plan_bytes() // 4, dtype=torch.float32, device=dev)
  ws = torch.empty(lib.st_melspec_ws(len(audio), total, n_mels) // 4 + 64, dtype=torch.float32, device=dev)
  P = lambda t: ctypes.c_void_p(t.data_ptr())
  _lib.call('st_melspec_plan_f32', P(basis), n_mels, 512, P(plan), plan.numel() * 4, None)
  outs = []
  for _ in range(2):
    out = torch.empty(total * n_mels, dtype=torch.float32, device=dev)
    _lib.call('st_melspec_planned_f32', P(x), P(so), len(audio), int(lens.max()), P(plan), n_mels, 512, 160, P(fo),
              total, P(out), P(ws), ws.numel() * 4, None)
    outs.append(out.view(total, n_mels).cpu().numpy())
  np.testing.assert_array_equal(outs[0], outs[1])
  np.testing.assert_array_equal(outs[0], np.concatenate(feats))


def test_mfcc_vs_oracle(dev):
  """calc_mfccs (preprocessing.py:61-84): 13 MFCC + delta + delta-delta, each block z-normalised; ragged
  batch.  Tolerance 2e-3 absolute on unit-variance features (fp32 FFT/log10/DCT vs float64; the delta-delta
  block divides small numbers by a small std)."""
  from speecht_amd.preprocessing import calc_mfccs, calc_mfccs_batch
  sr = 16000
  audio = [O.synthetic_audio(7, 16000 + 77), O.synthetic_audio(8, 32000), O.synthetic_audio(9, 5003),
           (0.3 * np.sin(2 * np.pi * 440.0 * np.arange(12345) / sr) + 0.01 * O.synthetic_audio(3, 12345)).astype(np.float32)]
  feats = calc_mfccs_batch(audio, sr)
  for a, f in zip(audio, feats):
    ref = O.calc_mfccs(a, sr)
    assert f.shape == ref.shape == (1 + len(a) // 160, 39)
    assert np.max(np.abs(f - ref)) < 2e-3
  np.testing.assert_array_equal(calc_mfccs(audio[0], sr), feats[0])


def test_shape_switching_reuses_buffers_exactly(dev):
  """Real batches change (B, max_T) every step: buffers are re-described, halos re-zeroed.  Results
  after switching shapes must be bit-identical to a fresh engine's."""
  layers = WL.w2l_layers(16, width=40, fc=72)
  params = WL.xavier_params(layers, seed=9)
  shapes = [[97, 80, 61], [33, 20], [140, 139, 101, 50], [33, 20], [97, 80, 61]]
  eng = make_engine(layers, dev)
  eng.set_weights(params)
  for k, frames in enumerate(shapes):
    x, seq, labels = WL.make_batch(frames, 16, seed=20 + len(frames))
    outs = []
    for e in (eng, make_engine(layers, dev)):
      if e is not eng:
        e.set_weights(params)
      e.load_batch(x, seq)
      e.set_labels(labels)
      e.forward()
      e.ctc_loss_grad(1.0 / len(frames))
      e.backward()
      torch.cuda.synchronize()
      outs.append((e.X[-1].interior().clone(), e.grads.clone(), e.loss.clone()))
    assert torch.equal(outs[0][0], outs[1][0]), k
    assert torch.equal(outs[0][1], outs[1][1]), k
    assert torch.equal(outs[0][2], outs[1][2]), k


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_fifty_shapes_of_training_stay_bit_identical_to_fresh_engines_and_allocate_nothing(dev, mode):
  """The reference's regime (speech_input.py:37-45,169-179; training.py:57-65): a new (B, max_T) nearly every step.  Fifty
  training steps -- forward, CTC, backward, clip + Adam -- over ragged batches of changing batch size and length on ONE engine
  (buffers re-described per shape, halos re-zeroed, derived operands rebuilt after every update); every step's logits, losses,
  gradients and updated weights equal, bit for bit, those of an engine built fresh for that step from the same state.  Then
  the same fifty shapes again on the warm engine: no buffer of its storage is re-allocated and the process makes no device
  allocation at all (`reserve` + one pass is the warm-up)."""
  layers = WL.w2l_layers(16, width=40, fc=72)
  params = WL.xavier_params(layers, seed=9)
  rng = np.random.default_rng(50)
  shapes = [sorted(rng.integers(24, 200, int(rng.integers(2, 6))).tolist(), reverse=True) for _ in range(32)]
  # ... eighteen of them come round again, in another order: a shape seen before is re-entered from the engine's cached description
  # (descriptors put back, halos / planes re-zeroed, freshness re-evaluated against the shape left behind) -- same bit-for-bit bar
  shapes += [shapes[k] for k in rng.permutation(32)[:18]]
  eng = make_engine(layers, dev, conv_mode=mode)
  eng.set_weights(params)
  eng.reserve(5, 200, min_frames=24, step=16)

  def one_step(e, k, frames):
    x, seq, labels = WL.make_batch(frames, 16, seed=300 + k)
    e.load_batch(x, seq)
    e.set_labels(labels)
    e.forward()
    e.ctc_loss_grad(1.0 / len(frames))
    e.backward()
    logits, grads, loss = e.X[-1].interior().clone(), e.grads.clone(), e.loss.clone()
    e.apply_update(1e-3)
    torch.cuda.synchronize()
    return logits, grads, loss, e.params.clone()

  reentered = 0
  for k, frames in enumerate(shapes):
    seen = (len(frames), max(frames)) in (eng.__dict__.get('_shape_cache') or {})
    reentered += int(seen and eng._shape != (len(frames), max(frames)))
    fresh = make_engine(layers, dev, conv_mode=mode)
    for name in ('params', 'adam_m', 'adam_v'):
      getattr(fresh, name).copy_(getattr(eng, name))
    fresh.step_count = eng.step_count
    fresh.mark_weights_changed()
    a, b = one_step(eng, k, frames), one_step(fresh, k, frames)
    for what, u, v in zip(('logits', 'gradients', 'losses', 'weights'), a, b):
      assert torch.equal(u, v), (k, frames, what, float((u - v).abs().max()))
    del fresh
  assert reentered >= 10, reentered
  torch.cuda.synchronize()
  torch.cuda.empty_cache()
  generation = eng._storage.generation
  stats = torch.cuda.memory_stats(dev)
  mallocs, reserved = stats.get('num_device_alloc'), torch.cuda.memory_reserved(dev)
  for k, frames in enumerate(shapes):
    one_step(eng, k, frames)
  assert eng._storage.generation == generation, 'the engine re-allocated a buffer on a shape it had seen'
  # (one_step's clones come out of the caching allocator's pool; the engine itself must not have asked the device for memory)
  after = torch.cuda.memory_stats(dev).get('num_device_alloc')
  assert torch.cuda.memory_reserved(dev) <= reserved + (64 << 20) and (mallocs is None or after - mallocs <= 4), (mallocs, after)


def test_ctc_long_form_kpl16(dev):
  """BASELINE config 5 shape: 30 s utterances -> T' = 1500 frames, ~450 labels (U = 901, 16 lattice
  states per lane).  Loss 1e-6 relative (measured 8e-8), gradient 2e-4 absolute (measured 9e-5).
  Without the per-frame occupancy normalisation of ctc_grad_kernel the gradient error is 6e-4: the
  accumulated 1-ULP bias of v_exp_f32/v_log_f32 over 1500 dependent steps."""
  rng = np.random.default_rng(77)
  T, lengths = 1500, [450, 400]
  logits, labels, lens = _ctc_case(rng, 2, T, 29, lengths, [1500, 1377])
  ref_loss, ref_grad = O.ctc_loss_and_grad(logits, labels, lens)
  eng, loss, grad = run_ctc(dev, logits, labels, lens)
  np.testing.assert_allclose(loss, ref_loss, rtol=1e-6)
  assert np.max(np.abs(grad - ref_grad)) < 2e-4
  live = np.arange(T)[:, None] < np.asarray(lens)[None, :]
  assert np.max(np.abs(grad.sum(axis=2)[live])) < 2e-6          # occupancies sum to one


def test_bucketed_inference_matches_oracle_per_bucket(dev):
  """BASELINE config 3 (reduced widths for the oracle's sake): variable-length utterances, bucketed
  by length; every bucket must decode exactly like the oracle on that same padded batch."""
  from speecht_amd.inference import make_buckets, padding_overhead, transcribe
  layers = WL.w2l_layers(16, width=40, fc=72)
  params = WL.xavier_params(layers, seed=13)
  rng = np.random.default_rng(5)
  lengths = rng.integers(40, 300, 23).tolist()                 # ~2-15 s at 50 output frames/s scale
  feats = [WL.synthetic_features(100 + i, t, 16).astype(np.float32) for i, t in enumerate(lengths)]
  eng = make_engine(layers, dev)
  eng.set_weights(params)
  ids, text = transcribe(eng, feats, batch_size=8)
  buckets = make_buckets(lengths, 8)
  assert sorted(i for b in buckets for i in b) == list(range(23))
  assert padding_overhead(lengths, buckets) < padding_overhead(lengths, [list(range(i, min(i + 8, 23))) for i in range(0, 23, 8)])
  for idx in buckets:
    x, seq, _ = O.pad_batch([feats[i].astype(np.float64) for i in idx], 16)
    ref_ids, _ = O.ctc_greedy_decode(O.wav2letter_forward(x, params, layers), seq // 2)
    assert [ids[i] for i in idx] == ref_ids
  assert text[0] == O.ids_to_sentence(ids[0])
  # the pipelined default (stager thread, asynchronous read-back) and the serial loop run the same launches
  for bucket in (True, False):
    a, _ = transcribe(eng, feats, batch_size=8, bucket=bucket, pipeline=True)
    b, _ = transcribe(eng, feats, batch_size=8, bucket=bucket, pipeline=False)
    assert a == b
  assert transcribe(eng, [], batch_size=8) == ([], [])
  one, _ = transcribe(eng, feats[:1], batch_size=8)
  assert one == transcribe(eng, feats[:1], batch_size=8, pipeline=False)[0]
  # an error in the stager thread (ragged feature width) surfaces in the caller
  bad = feats[:3] + [np.zeros((50, 17), np.float32)]
  with pytest.raises(Exception):
    transcribe(eng, bad, batch_size=2, bucket=False)
  assert len(transcribe(eng, feats[:4], batch_size=2)[0]) == 4          # and the engine is still usable


def test_pipelined_beam_search_matches_the_serial_loop_and_the_oracle(dev):
  """inference.transcribe(beam_width=...): the prefix beam search of batch k on a decoder stream of its own (CU-masked where
  the runtime has hipExtStreamCreateWithCUMask) under the forward pass of batch k + 1, logits double-buffered -- the ids of
  every utterance equal the serial loop's and, batch by batch, the oracle's search on the device's own logits; the async
  entry point alone returns what the synchronous one returns (ids and log-probabilities)."""
  from speecht_amd.inference import transcribe
  layers = WL.w2l_layers(16, width=40, fc=72)
  params = WL.xavier_params(layers, seed=21, bias_range=0.3)
  # sharper logits than a fresh network's (near-flat rows make every candidate a near tie): scale up the top layer
  params[-1] = (params[-1][0] * 12.0, params[-1][1] * 4.0)
  rng = np.random.default_rng(8)
  lengths = rng.integers(60, 260, 21).tolist()
  feats = [WL.synthetic_features(300 + i, t, 16).astype(np.float32) for i, t in enumerate(lengths)]
  eng = make_engine(layers, dev)
  eng.set_weights(params)
  for bucket in (True, False):
    a, _ = transcribe(eng, feats, batch_size=4, bucket=bucket, pipeline=True, beam_width=8)
    b, _ = transcribe(eng, feats, batch_size=4, bucket=bucket, pipeline=False, beam_width=8)
    assert a == b and any(len(s) > 0 for s in a)
  # against the oracle's search on the device's own logits, one batch; and async == sync
  idx = list(range(4))
  x, seq, _ = O.pad_batch([feats[i].astype(np.float64) for i in idx], 16)
  eng.load_batch(x, seq)
  eng.forward()
  sync_ids, sync_lp = eng.beam_search_decode(8)
  async_ids, async_lp = eng.beam_search_decode_async(8).result()
  assert async_ids == sync_ids and np.array_equal(async_lp, sync_lp)
  # a handle whose slot a later call re-used says so instead of returning that batch's transcripts (ring of streams + 1 = 2 slots)
  stale = eng.beam_search_decode_async(8)
  kept = eng.beam_search_decode_async(8)
  eng.beam_search_decode_async(8)
  with pytest.raises(RuntimeError, match='read too late'):
    stale.result()
  assert kept.result()[0] == sync_ids
  torch.cuda.synchronize()
  logits = eng.logits_time_major().cpu().numpy().astype(np.float64)
  ref_ids, ref_lp = O.ctc_beam_search_decode(logits, seq // 2, 8)
  assert sync_ids == ref_ids
  np.testing.assert_allclose(sync_lp, ref_lp, rtol=1e-4, atol=1e-4)
  serial, _ = transcribe(eng, [feats[i] for i in idx], batch_size=4, bucket=False, pipeline=False, beam_width=8)
  assert serial == ref_ids


@pytest.mark.parametrize('mode', ['fp32', 'bf16x6'])
@pytest.mark.parametrize('frames', [[1], [7, 3, 5], [48, 47, 2, 31, 96]])
def test_odd_shapes_mfcc_width(dev, mode, frames):
  """39-feature (MFCC) input, channel counts that are not multiples of 16, utterances shorter than the
  filters (1 frame: everything but one tap is padding), B = 1: logits, loss and every gradient against the
  oracle.  Labels are kept short enough for T' = ceil(T/2) // ... frames (empty where nothing fits)."""
  from speecht_amd.engine import Wav2LetterEngine
  layers = WL.w2l_layers(39, width=24, fc=40)
  params = WL.xavier_params(layers, seed=3)
  rng = np.random.default_rng(len(frames))
  T = max(frames)
  x = np.zeros((len(frames), T, 39))
  for b, t in enumerate(frames):
    x[b, :t] = rng.standard_normal((t, 39))
  labels = [list(rng.integers(0, 28, max(0, (t // 2) // 3))) for t in frames]
  eng = Wav2LetterEngine(layers, device=dev, conv_mode=mode)
  eng.set_weights(params)
  eng.load_batch(x, frames)
  eng.set_labels(labels)
  eng.forward()
  eng.ctc_loss_grad(1.0 / len(frames))
  eng.backward()
  eng.check_ctc_status()
  ref = O.train_step(x, np.asarray(frames), labels, params, layers, None, update=False)
  logits = eng.logits_time_major().cpu().numpy()
  assert logits.shape == ref['logits'].shape
  assert np.max(np.abs(logits - ref['logits'])) < 1e-4
  np.testing.assert_allclose(eng.loss.cpu().numpy(), ref['loss'], rtol=1e-4, atol=1e-5)
  for i, ((gF, gb), (rF, rb)) in enumerate(zip(eng.get_grads(), ref['grads'])):
    assert rel_err(gF, rF) < 3e-4 or np.max(np.abs(rF)) < 1e-12, i
    assert rel_err(gb, rb) < 3e-4 or np.max(np.abs(rb)) < 1e-12, i
