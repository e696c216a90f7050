"""Pins the numpy oracle against an independent torch-CPU formulation and analytical cases.

The reference's own tests hold no numeric vector for this path, so these are the
oracle's pins: F.conv1d + autograd, F.ctc_loss, closed-form CTC values.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import w2l_oracle as O


def torch_conv_same(x, filt, bias, stride, relu):
  # x [B,T,Cin] -> NCW, filters [W,Cin,Cout] -> [Cout,Cin,W]
  t_out, pl, pr = O.same_padding(x.shape[1], filt.shape[0], stride)
  xt = F.pad(x.permute(0, 2, 1), (pl, pr))
  y = F.conv1d(xt, filt.permute(2, 1, 0), bias, stride=stride).permute(0, 2, 1)
  return torch.relu(y) if relu else y


@pytest.mark.parametrize('T,W,s,cin,cout,relu', [
    (21, 48, 2, 5, 7, True), (20, 48, 2, 5, 7, True), (17, 7, 1, 6, 6, True),
    (33, 32, 1, 4, 9, True), (9, 1, 1, 8, 3, False), (50, 5, 3, 3, 4, True)])
def test_conv_fwd_bwd_vs_torch(T, W, s, cin, cout, relu):
  rng = np.random.default_rng(0)
  x = rng.standard_normal((3, T, cin))
  filt = rng.standard_normal((W, cin, cout)) * 0.2
  bias = rng.standard_normal(cout) * 0.1
  y = O.conv1d_same_fwd(x, filt, bias, s, relu)
  xt = torch.tensor(x, requires_grad=True)
  ft = torch.tensor(filt, requires_grad=True)
  bt = torch.tensor(bias, requires_grad=True)
  yt = torch_conv_same(xt, ft, bt, s, relu)
  assert y.shape == tuple(yt.shape)
  np.testing.assert_allclose(y, yt.detach().numpy(), rtol=1e-12, atol=1e-12)
  dy = rng.standard_normal(y.shape)
  yt.backward(torch.tensor(dy))
  dx, dF, db = O.conv1d_same_bwd(x, filt, y, dy, s, relu)
  np.testing.assert_allclose(dx, xt.grad.numpy(), rtol=1e-11, atol=1e-11)
  np.testing.assert_allclose(dF, ft.grad.numpy(), rtol=1e-11, atol=1e-11)
  np.testing.assert_allclose(db, bt.grad.numpy(), rtol=1e-11, atol=1e-11)


def test_same_padding_asymmetry():
  # Appendix A1: L0 pads (23,23) for even T and (23,24) for odd T; L8 (15,16); W7 (3,3)
  assert O.same_padding(1000, 48, 2) == (500, 23, 23)
  assert O.same_padding(1001, 48, 2) == (501, 23, 24)
  assert O.same_padding(501, 32, 1) == (501, 15, 16)
  assert O.same_padding(501, 7, 1) == (501, 3, 3)
  # delta filter exposes which tap lines up with the output frame
  x = np.arange(1.0, 11.0).reshape(1, 10, 1)
  for W, s in [(4, 1), (48, 2), (32, 1)]:
    t_out, pl, _ = O.same_padding(10, W, s)
    filt = np.zeros((W, 1, 1))
    filt[pl, 0, 0] = 1.0
    y = O.conv1d_same_fwd(x, filt, np.zeros(1), s, relu=False)
    np.testing.assert_allclose(y[0, :, 0], x[0, ::s, 0])


def _rand_ctc_case(rng, T, B, C, Lmax):
  logits = rng.standard_normal((T, B, C)) * 2.0
  lens = rng.integers(max(2, T // 2), T + 1, B)
  labels = []
  for b in range(B):
    L = int(rng.integers(0, min(Lmax, lens[b] // 2) + 1))
    labels.append(rng.integers(0, C - 1, L).tolist())
  return logits, labels, lens


@pytest.mark.parametrize('seed', range(4))
def test_ctc_vs_torch(seed):
  rng = np.random.default_rng(seed)
  T, B, C = 40, 5, 29
  logits, labels, lens = _rand_ctc_case(rng, T, B, C, 12)
  if seed == 0:
    labels[0] = [3, 3, 3, 7, 7]          # repeats need separating blanks
    labels[1] = []                       # empty label (U = 1)
  loss, grad = O.ctc_loss_and_grad(logits, labels, lens)
  lt = torch.tensor(logits, requires_grad=True)
  lp = torch.log_softmax(lt, dim=-1)
  flat = torch.tensor([v for l in labels for v in l], dtype=torch.long)
  tl = F.ctc_loss(lp, flat, torch.tensor(lens), torch.tensor([len(l) for l in labels]),
                  blank=C - 1, reduction='none', zero_infinity=False)
  np.testing.assert_allclose(loss, tl.detach().numpy(), rtol=1e-10, atol=1e-10)
  tl.sum().backward()
  np.testing.assert_allclose(grad, lt.grad.numpy(), rtol=1e-8, atol=1e-10)
  for b in range(B):                     # no gradient beyond the utterance length
    assert np.all(grad[lens[b]:, b] == 0)


def test_ctc_closed_forms():
  C = 29
  # uniform logits, empty label: only the all-blank path, p = 29^-T
  T = 7
  loss, grad = O.ctc_loss_and_grad(np.zeros((T, 1, C)), [[]], [T])
  assert loss[0] == pytest.approx(T * math.log(C), rel=1e-12)
  # T = 1, L = 1: p = y(label)
  lg = np.random.default_rng(1).standard_normal((1, 1, C))
  loss, grad = O.ctc_loss_and_grad(lg, [[4]], [1])
  y = np.exp(O.log_softmax(lg[0, 0]))
  assert loss[0] == pytest.approx(-math.log(y[4]), rel=1e-12)
  expect = y.copy()
  expect[4] -= 1.0
  np.testing.assert_allclose(grad[0, 0], expect, atol=1e-12)
  # uniform logits, one label, T frames: #paths = T(T+1)/2 (choose start and end of the run)
  T = 6
  loss, _ = O.ctc_loss_and_grad(np.zeros((T, 1, C)), [[2]], [T])
  assert loss[0] == pytest.approx(-math.log(T * (T + 1) / 2 / C ** T), rel=1e-12)
  # a repeated label needs a blank in between: "aa" in 2 frames is impossible
  with pytest.raises(ValueError):
    O.ctc_loss_and_grad(np.zeros((2, 1, C)), [[1, 1]], [2])
  # ... and in 3 frames exactly one path a,blank,a
  loss, _ = O.ctc_loss_and_grad(np.zeros((3, 1, C)), [[1, 1]], [3])
  assert loss[0] == pytest.approx(3 * math.log(C), rel=1e-12)


def test_ctc_grad_finite_difference():
  rng = np.random.default_rng(5)
  logits, labels, lens = _rand_ctc_case(rng, 12, 2, 6, 4)
  loss, grad = O.ctc_loss_and_grad(logits, labels, lens)
  eps = 1e-6
  for (t, b, c) in [(0, 0, 1), (3, 1, 5), (7, 0, 2), (11, 1, 0)]:
    lp = logits.copy(); lp[t, b, c] += eps
    lm = logits.copy(); lm[t, b, c] -= eps
    fd = (O.ctc_loss_and_grad(lp, labels, lens)[0][b] - O.ctc_loss_and_grad(lm, labels, lens)[0][b]) / (2 * eps)
    assert grad[t, b, c] == pytest.approx(fd, abs=1e-6)


def test_greedy_decode_semantics():
  C = 29
  blank = C - 1
  seq = [0, blank, 0, 0, 5, blank, blank, 5, 7]   # "a, blank, a" keeps both a's; "a a" merges
  logits = np.full((len(seq), 1, C), -1.0)
  for t, k in enumerate(seq):
    logits[t, 0, k] = 2.0
  ids, score = O.ctc_greedy_decode(logits, [len(seq)])
  assert ids == [[0, 0, 5, 5, 7]]
  assert score[0, 0] == pytest.approx(-2.0 * len(seq))
  ids, _ = O.ctc_greedy_decode(logits, [4])       # length cuts the tail
  assert ids == [[0, 0]]
  tie = np.zeros((1, 1, C))                       # ties -> lowest index
  assert O.ctc_greedy_decode(tie, [1])[0] == [[0]]
  ids, _ = O.ctc_greedy_decode(logits, [len(seq)], merge_repeated=False)
  assert ids == [[0, 0, 0, 5, 5, 7]]


def test_adam_tf_vs_torch_adam_differs_and_closed_form():
  p = np.array([1.0, -2.0]); g = np.array([0.5, -0.25])
  p1, m1, v1 = O.adam_tf_step(p, g, np.zeros(2), np.zeros(2), 1, lr=1e-3)
  m = 0.1 * g; v = 0.001 * g * g
  lr_t = 1e-3 * math.sqrt(1 - 0.999) / (1 - 0.9)
  np.testing.assert_allclose(p1, p - lr_t * m / (np.sqrt(v) + 1e-3), rtol=1e-15)
  clipped, gn = O.clip_by_global_norm([np.array([3.0, 4.0]), np.array([12.0])], 5.0)
  assert gn == pytest.approx(13.0)
  np.testing.assert_allclose(clipped[0], np.array([3.0, 4.0]) * 5 / 13)
  same, gn = O.clip_by_global_norm([np.array([0.3, 0.4])], 5.0)
  np.testing.assert_allclose(same[0], [0.3, 0.4])


def test_train_step_grads_vs_torch_autograd():
  """Whole-path gradient (11 layers + CTC mean) against torch autograd on a tiny batch."""
  rng = np.random.default_rng(11)
  layers = [(48, 2, 6, 10, True), (7, 1, 10, 10, True), (32, 1, 10, 16, True),
            (1, 1, 16, 16, True), (1, 1, 16, 29, False)]
  params = O.xavier_init(layers, seed=3, bias_range=0.05)
  B, T = 3, 41
  x = rng.standard_normal((B, T, 6))
  x[1, 30:] = 0; x[2, 25:] = 0
  seq = np.array([41, 30, 25])
  labels = [[1, 2, 2, 3], [5], [7, 8, 9]]
  out = O.train_step(x, seq, labels, params, layers, O.zero_opt_state(params), update=False)
  tp = [(torch.tensor(F_, requires_grad=True), torch.tensor(b_, requires_grad=True)) for F_, b_ in params]
  h = torch.tensor(x)
  for (F_, b_), (W, s, cin, cout, relu) in zip(tp, layers):
    h = torch_conv_same(h, F_, b_, s, relu)
  lp = torch.log_softmax(h.permute(1, 0, 2), dim=-1)
  flat = torch.tensor([v for l in labels for v in l])
  tl = F.ctc_loss(lp, flat, torch.tensor(seq // 2), torch.tensor([len(l) for l in labels]),
                  blank=28, reduction='none').mean()
  assert out['avg_loss'] == pytest.approx(float(tl), rel=1e-10)
  tl.backward()
  for (gF, gb), (F_, b_) in zip(out['grads'], tp):
    np.testing.assert_allclose(gF, F_.grad.numpy(), rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(gb, b_.grad.numpy(), rtol=1e-7, atol=1e-10)


def test_levenshtein():
  assert O.levenshtein('kitten', 'sitting') == 3
  assert O.levenshtein('', 'abc') == 3
  assert O.levenshtein('a b c'.split(), 'a c'.split()) == 1


def test_bf16_three_way_split_is_exact_and_six_terms_suffice():
  """Numerical model of the experimental bf16x6 path (csrc/conv_bf16.hip): every fp32 value is the
  exact sum of three bf16 pieces, and the six largest cross terms reproduce an fp32 dot product at
  least as accurately as an fp32 FMA chain."""
  def bf16_rn(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)

  def split3(x):
    h = bf16_rn(x); r = (x - h).astype(np.float32); m = bf16_rn(r); lo = bf16_rn((r - m).astype(np.float32))
    return h, m, lo
  rng = np.random.default_rng(0)
  a = (rng.standard_normal(200000) * np.exp(rng.uniform(-20, 20, 200000))).astype(np.float32)
  h, m, lo = split3(a)
  assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + lo.astype(np.float64), a.astype(np.float64))
  K = 4096
  A = np.maximum(rng.standard_normal((8, K)), 0).astype(np.float32)
  B = (rng.uniform(-1, 1, (K, 8)) * 0.02).astype(np.float32)
  ref = A.astype(np.float64) @ B.astype(np.float64)
  chain = np.zeros((8, 8), np.float32)
  for k in range(K):
    chain = (chain + A[:, k:k + 1] * B[k:k + 1, :]).astype(np.float32)
  Ap, Bp = split3(A), split3(B)
  acc = np.zeros((8, 8), np.float32)
  for k in range(0, K, 16):
    for ia, ib in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)):
      acc = (acc + Ap[ia][:, k:k + 16].astype(np.float64) @ Bp[ib][k:k + 16].astype(np.float64)).astype(np.float32)
  assert np.max(np.abs(acc - ref)) <= np.max(np.abs(chain - ref))


def test_beam_search_oracle_equals_exhaustive_enumeration():
  """With a beam wide enough to hold every prefix the search is exact: its top path is the most probable
  labelling and its score that labelling's total probability (all 4^5 alignments enumerated)."""
  import itertools
  rng = np.random.default_rng(0)
  T, C = 5, 4
  for _ in range(10):
    x = rng.standard_normal((T, 1, C)) * 2
    lp = x[:, 0] - np.log(np.exp(x[:, 0]).sum(1, keepdims=True))
    mass = {}
    for path in itertools.product(range(C), repeat=T):
      lab, prev = [], -1
      for k in path:
        if k != C - 1 and k != prev:
          lab.append(k)
        prev = k
      mass[tuple(lab)] = np.logaddexp(mass.get(tuple(lab), -np.inf), sum(lp[t, k] for t, k in enumerate(path)))
    best = max(mass, key=mass.get)
    ids, score = O.ctc_beam_search_decode(x, [T], beam_width=1000)
    assert tuple(ids[0]) == best
    assert abs(score[0, 0] - mass[best]) < 1e-9


def test_beam_search_oracle_beam1_and_empty():
  rng = np.random.default_rng(1)
  x = rng.standard_normal((30, 2, 29)) * 3
  ids, score = O.ctc_beam_search_decode(x, [30, 0], beam_width=1)
  assert ids[1] == [] and score[1, 0] == 0.0
  assert len(ids[0]) > 0 and score[0, 0] < 0.0


def test_torch_ref_equals_oracle():
  """tests/torch_ref.py (F.conv1d + F.ctc_loss + autograd, float64) and the numpy oracle are two independent
  formulations of the step; the full-size GPU gradient tests lean on the former, so pin them against each other
  on a full-depth network with ragged lengths, repeats and non-zero biases."""
  from tests import torch_ref as TR
  from tests import workloads as WL
  layers = WL.w2l_layers(16, width=24, fc=40)
  params = WL.xavier_params(layers, seed=5)
  x, seq, labels = WL.make_batch([75, 61, 40], 16, seed=4)
  labels[1] = [1, 1, 1, 2]
  ref = O.train_step(x, seq, labels, params, layers, O.zero_opt_state(params), update=False)
  got = TR.loss_and_grads(x, seq, labels, params, layers, dtype=torch.float64)
  np.testing.assert_allclose(got['logits'], ref['logits'], rtol=0, atol=1e-12)
  np.testing.assert_allclose(got['loss'], ref['loss'], rtol=1e-11)
  for (gF, gb), (rF, rb) in zip(got['grads'], ref['grads']):
    np.testing.assert_allclose(gF, rF, rtol=0, atol=1e-11 * np.max(np.abs(rF)) + 1e-300)
    np.testing.assert_allclose(gb, rb, rtol=0, atol=1e-11 * np.max(np.abs(rb)) + 1e-300)


def test_torch_ref_with_pinned_relu_pattern():
  """``loss_and_grads(relu_masks=...)`` (the end-to-end reference of tests/test_gpu_fullsize_grads.py): with the masks of
  its own forward pass it IS the free evaluation; with one unit's mask flipped it is the same network with that unit
  forced -- the gradient of a forced-off unit's incoming filter column vanishes, everything stays finite."""
  from tests import torch_ref as TR
  from tests import workloads as WL
  layers = WL.w2l_layers(16, width=24, fc=40)
  params = WL.xavier_params(layers, seed=5)
  x, seq, labels = WL.make_batch([75, 61, 40], 16, seed=4)
  free = TR.loss_and_grads(x, seq, labels, params, layers, dtype=torch.float64)
  masks = TR.relu_masks_of(free['acts'], layers)
  assert masks[-1] is None and all(m is not None and m.dtype == bool for m in masks[:-1])
  pinned = TR.loss_and_grads(x, seq, labels, params, layers, dtype=torch.float64, relu_masks=masks)
  np.testing.assert_allclose(pinned['logits'], free['logits'], rtol=0, atol=1e-13)
  np.testing.assert_allclose(pinned['loss'], free['loss'], rtol=1e-13)
  for (gF, gb), (rF, rb) in zip(pinned['grads'], free['grads']):
    np.testing.assert_allclose(gF, rF, rtol=0, atol=1e-12 * np.max(np.abs(rF)) + 1e-300)
    np.testing.assert_allclose(gb, rb, rtol=0, atol=1e-12 * np.max(np.abs(rb)) + 1e-300)
  # channel 3 of layer 2 forced off everywhere: no gradient reaches its filter column or its bias
  forced = [None if m is None else m.copy() for m in masks]
  forced[2][:, :, 3] = False
  off = TR.loss_and_grads(x, seq, labels, params, layers, dtype=torch.float64, relu_masks=forced)
  assert np.all(off['grads'][2][0][:, :, 3] == 0.0) and off['grads'][2][1][3] == 0.0
  assert np.max(np.abs(off['grads'][2][0][:, :, 4])) > 0.0 and np.isfinite(off['avg_loss'])


@pytest.mark.parametrize('T,W,cin,cout', [(37, 48, 5, 4), (40, 48, 3, 6), (21, 6, 4, 3), (22, 7, 2, 5)])
def test_stride2_conv_is_a_stride1_conv_on_frame_pairs(T, W, cin, cout):
  """The identity behind the engine's polyphase first layer (engine._polyphase): a stride-2 'SAME'
  convolution equals a stride-1 convolution over frame pairs with pl2 = ceil(pl / 2), shift = 2 pl2 - pl,
  W2 = ceil((W + shift) / 2) taps F2[j][p] = F[2j + p - shift] (zero outside) -- checked here in float64 with the
  oracle's own convolution on both sides (speech_model.py:155,279)."""
  rng = np.random.default_rng(T * W)
  x = rng.standard_normal((2, T, cin))
  F = rng.standard_normal((W, cin, cout))
  bias = rng.standard_normal(cout)
  y_ref = O.conv1d_same_fwd(x, F, bias, 2, False)
  t_out, pl, _ = O.same_padding(T, W, 2)
  pl2 = (pl + 1) // 2
  shift = 2 * pl2 - pl
  W2 = (W + shift + 1) // 2
  F2 = np.zeros((W2, 2, cin, cout))
  for w in range(W):
    F2[(w + shift) // 2, (w + shift) % 2] = F[w]
  # frame pairs as channels, with enough zero frames on both sides for the taps to reach
  lo, hi = 2 * pl2, 2 * (W2 + 1)
  xp = np.zeros((2, lo + 2 * t_out + hi, cin))
  xp[:, lo:lo + T] = x
  X2 = xp.reshape(2, -1, 2 * cin)
  y = np.zeros_like(y_ref)
  for t in range(t_out):
    for j in range(W2):
      y[:, t] += X2[:, t + j] @ F2[j].reshape(2 * cin, cout)      # X2 index t + j - pl2, shifted by the pl2 pad pairs
  np.testing.assert_allclose(y + bias, y_ref, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('T,W,cin,cout', [(150, 7, 5, 6), (201, 32, 4, 3), (64, 12, 3, 4), (70, 25, 6, 2)])
def test_block_dft_formulation_of_the_convolution_and_its_gradients(T, W, cin, cout):
  """The mathematics csrc/conv_fft.hip implements, in float64 numpy against the oracle's convolution (speech_model.py:155,
  173, 177, 78): blocks of V = 64 output frames, N = V + W - 1 point DFTs over real input (bins 0 .. N/2),
    forward   Y[k] = S[k] conj(G[k]) per bin, S from x[jV - pl + n] (overlap-save), y = first V points of the inverse;
    to input  X[k] = Z[k] G[k], Z from the V frames of dz zero-padded; frame jV + t' collects block j at m = t' + pl and
              its two neighbours at m +- V (overlap-add);
    filters   Q[k] = sum_rows S[k]^T conj(Z[k]),  dF[w] = inverse at lag w;   bias = sum_rows Re Z[0];
  with the complex products written as the real [re | im] x [[Gr, -Gi], [Gi, Gr]] embedding the GEMM kernels run."""
  V = 64
  N = V + W - 1
  bins = N // 2 + 1
  rng = np.random.default_rng(T + W)
  x = rng.standard_normal((2, T, cin))
  F = rng.standard_normal((W, cin, cout)) / np.sqrt(W * cin)
  bias = rng.standard_normal(cout)
  y_ref = O.conv1d_same_fwd(x, F, bias, 1, False)
  dz = rng.standard_normal(y_ref.shape)
  dx_ref, dF_ref, db_ref = O.conv1d_same_bwd(x, F, y_ref, dz, 1, relu=False)
  _, pl, _ = O.same_padding(T, W, 1)
  blocks = -(-T // V)
  xp = np.zeros((2, pl + blocks * V + N, cin))
  xp[:, pl:pl + T] = x
  dzp = np.zeros((2, blocks * V, cout))
  dzp[:, :T] = dz
  k = np.arange(bins)[:, None]
  wk = np.where((k[:, 0] == 0) | (2 * k[:, 0] == N), 1.0, 2.0) / N            # hermitian halves count twice

  def dft(seg):                                                                # [rows, n <= N, C] -> [bins, rows, C] complex
    n = np.arange(seg.shape[1])[None, :]
    return np.einsum('kn,rnc->krc', np.exp(-2j * np.pi * k * n / N), seg)

  def idft_real(spec, m):                                                      # [bins, rows, C] -> points m: [rows, len(m), C]
    ph = np.exp(2j * np.pi * k * np.asarray(m)[None, :] / N)                   # [bins, len(m)]
    return np.einsum('k,km,krc->rmc', wk, ph, spec).real

  def embed_conj(Gc):        # [re | im] x this = [re | im] of (row vector) . conj(Gc): the forward operand gfwd of the kernels
    return np.block([[Gc.real, -Gc.imag], [Gc.imag, Gc.real]])

  def embed(Gc):             # [re | im] x this = [re | im] of (row vector) . Gc: the back-prop operand gbwd (on Gc = G^T)
    return np.block([[Gc.real, Gc.imag], [-Gc.imag, Gc.real]])

  rows = [(b, j) for b in range(2) for j in range(blocks)]
  S = dft(np.stack([xp[b, j * V:j * V + N] for b, j in rows]))                 # x[jV - pl + n]: xp is shifted by pl
  Z = dft(np.stack([dzp[b, j * V:j * V + V] for b, j in rows]))
  G = np.einsum('kw,wco->kco', np.exp(-2j * np.pi * k * np.arange(W)[None, :] / N), F)

  # forward: per bin [Sr | Si] x [[Gr, -Gi], [Gi, Gr]] = S conj(G)
  Y = np.empty((bins, len(rows), cout), complex)
  for q in range(bins):
    out = np.hstack([S[q].real, S[q].imag]) @ embed_conj(G[q])
    Y[q] = out[:, :cout] + 1j * out[:, cout:]
  yb = idft_real(Y, np.arange(V)) + bias
  y = np.zeros_like(y_ref)
  for r, (b, j) in enumerate(rows):
    n = min(V, T - j * V)
    y[b, j * V:j * V + n] = yb[r, :n]
  np.testing.assert_allclose(y, y_ref, rtol=1e-10, atol=1e-10)

  # back-prop to the input: per bin [Zr | Zi] x [[Gr^T, Gi^T], [-Gi^T, Gr^T]] = Z G^T, three inverse terms
  X = np.empty((bins, len(rows), cin), complex)
  for q in range(bins):
    out = np.hstack([Z[q].real, Z[q].imag]) @ embed(G[q].T)
    X[q] = out[:, :cin] + 1j * out[:, cin:]
  tp = np.arange(V)
  own, below, above = idft_real(X, tp + pl), idft_real(X, tp + V + pl), idft_real(X, tp - V + pl)
  ok_below, ok_above = (tp + V + pl < N), (tp - V + pl >= 0)
  dx = np.zeros_like(dx_ref)
  for r, (b, j) in enumerate(rows):
    v = own[r].copy()
    if j > 0:
      v[ok_below] += below[r - 1][ok_below]        # block j - 1 reaches W - 1 - pl frames into this one
    if j + 1 < blocks:
      v[ok_above] += above[r + 1][ok_above]        # block j + 1 reaches pl frames back into this one
    n = min(V, T - j * V)
    dx[b, j * V:j * V + n] = v[:n]
  np.testing.assert_allclose(dx, dx_ref, rtol=1e-10, atol=1e-10)

  # filter gradient: lag products per bin as [Sr | Si]^T [Zr | Zi] (four real blocks), then the W lags; bias from bin 0
  dF = np.zeros_like(dF_ref)
  for q in range(bins):
    P = np.hstack([S[q].real, S[q].imag]).T @ np.hstack([Z[q].real, Z[q].imag])
    re = P[:cin, :cout] + P[cin:, cout:]
    im = P[cin:, :cout] - P[:cin, cout:]
    ang = 2 * np.pi * q * np.arange(W) / N
    dF += wk[q] * (re[None] * np.cos(ang)[:, None, None] - im[None] * np.sin(ang)[:, None, None])
  np.testing.assert_allclose(dF, dF_ref, rtol=1e-10, atol=1e-10)
  np.testing.assert_allclose(Z[0].real.sum(axis=0), db_ref, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize('T,W,cin,cout', [(150, 7, 5, 6), (201, 32, 4, 3), (64, 12, 3, 4)])
def test_oracle_block_dft_conv_equals_the_direct_form_and_its_storage_model_is_mild(T, W, cin, cout):
  """oracle.block_dft_conv -- the storage-model restatement of the frequency-domain layer that the bf16-plane kernels are
  checked against -- equals conv1d_same_fwd / conv1d_same_bwd to 1e-12 without a storage hook; with spectra rounded to bf16 the
  results move by well under one bf16 ulp of the tensor scale, and wav2letter_forward / _backward(spectral=...) route a layer
  through it."""
  rng = np.random.default_rng(T)
  x = rng.standard_normal((2, T, cin))
  F = rng.standard_normal((W, cin, cout)) / np.sqrt(W * cin)
  b = rng.standard_normal(cout)
  y_ref = O.conv1d_same_fwd(x, F, b, 1, True)
  dy = rng.standard_normal(y_ref.shape)
  pa = rng.standard_normal(x.shape)
  dx_ref, dF_ref, db_ref = O.conv1d_same_bwd(x, F, y_ref, dy, 1, True)
  dz = dy * (y_ref > 0)
  y, dx, dF, db = O.block_dft_conv(x, F, b, True, dz=dz, prev_act=pa)
  np.testing.assert_allclose(y, y_ref, atol=1e-12)
  np.testing.assert_allclose(dx, dx_ref * (pa > 0), atol=1e-12)
  np.testing.assert_allclose(dF, dF_ref, atol=1e-11)
  np.testing.assert_allclose(db, db_ref, atol=1e-12)
  y2, dx2, dF2, _ = O.block_dft_conv(x, F, b, True, dz=dz, prev_act=pa, store=O.bf16_round)
  for got, ref in ((y2, y_ref), (dx2, dx_ref * (pa > 0)), (dF2, dF_ref)):
    assert 0 < np.max(np.abs(got - ref)) < 2.0 ** -8 * np.max(np.abs(ref))
  layers = [(W, 1, cin, cout, True), (1, 1, cout, 5, False)]
  params = [(F, b), (rng.standard_normal((1, cout, 5)), np.zeros(5))]
  la, acts_a = O.wav2letter_forward(x, params, layers, keep=True)
  lb, acts_b = O.wav2letter_forward(x, params, layers, keep=True, spectral={0})
  np.testing.assert_allclose(la, lb, atol=1e-11)
  dl = rng.standard_normal(la.shape)
  for (fa, ba), (fb, bb) in zip(O.wav2letter_backward(acts_a, params, layers, dl), O.wav2letter_backward(acts_b, params, layers, dl, spectral={0})):
    np.testing.assert_allclose(fa, fb, atol=1e-10)
    np.testing.assert_allclose(ba, bb, atol=1e-10)
