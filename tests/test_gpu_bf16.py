"""BASELINE config 4 ("bf16 activations / fp32 CTC"): the bf16-activation entry points of
include/speecht_hip.h against the oracle run with its bf16 *storage model* (oracle.bf16_round applied
exactly where the device path writes bf16: input, filter copies, layer outputs, activation gradients).

Tolerances: both sides round the same real numbers to bf16, but the device accumulates in fp32 and the
oracle in float64, so a value that sits on a rounding boundary may land one bf16 ulp (2^-8 relative) apart;
such flips are rare and stay local, hence "max error <= a few ulp of the tensor's scale, mean error far
below one ulp".  Against the unrounded fp32 oracle the gap is the expected bf16 quantisation noise."""
import numpy as np
import pytest
import torch

from oracle import w2l_oracle as O
from tests import workloads as WL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
  assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
  return torch.device('cuda:0')


def engine(layers, dev, mode='bf16'):
  from speecht_amd.engine import Wav2LetterEngine
  return Wav2LetterEngine(layers, device=dev, conv_mode=mode)


def scaled_err(a, b):
  s = float(np.max(np.abs(b))) + 1e-30
  d = np.abs(np.asarray(a, np.float64) - b) / s
  return float(d.max()), float(d.mean())


ULP = 2.0 ** -8

CASES = [
    # B, T, W, s, cin, cout, relu
    (3, 41, 48, 2, 16, 24, True),       # L0-like: stride 2 (two phase planes in the filter gradient), odd T
    (2, 40, 48, 2, 80, 250, True),      # L0 real channels
    (2, 37, 7, 1, 250, 250, True),      # L1-L7
    (2, 33, 32, 1, 40, 200, True),      # L8-like (pad 15/16)
    (2, 29, 1, 1, 200, 29, False),      # L10: 29 outputs on a 128-wide tile, fp32 logits
    (1, 300, 7, 1, 32, 48, True),       # long reduction -> split filter gradient
]


@pytest.mark.parametrize('B,T,W,s,cin,cout,relu', CASES)
def test_conv_layer_pair_bf16(dev, B, T, W, s, cin, cout, relu):
  """Two stacked layers [probe (W,s,cin->cout), 1x1 head -> 29] so that forward, back-prop to the input of
  the head, both filter gradients and the ReLU mask of the probe layer are all exercised."""
  layers = [(W, s, cin, cout, relu), (1, 1, cout, 29, False)]
  rng = np.random.default_rng(B * 1000 + T + W)
  params = WL.xavier_params(layers, seed=T, bias_range=0.05)
  x = rng.standard_normal((B, T, cin))
  eng = engine(layers, dev)
  eng.set_weights(params)
  eng.load_batch(x, [T] * B)
  eng.forward()
  logits = eng.logits_time_major().cpu().numpy()
  ref_logits, acts = O.wav2letter_forward(x, params, layers, keep=True, store=O.bf16_round)
  mx, mean = scaled_err(logits, ref_logits)
  assert mx < 4 * ULP and mean < 0.1 * ULP, (mx, mean)
  # the stored bf16 activation of the probe layer
  got = eng.Xb[1].view(eng.X[1].batch, eng.X[1].t_pitch, eng.X[1].c_pitch)[
      :, eng.X[1].halo:eng.X[1].halo + eng.X[1].frames, :cout].float().cpu().numpy()
  mx, mean = scaled_err(got, acts[1])
  assert mx < 2 * ULP and mean < 0.05 * ULP, (mx, mean)
  # back-prop of a random upstream gradient
  t_out = ref_logits.shape[0]
  dl = rng.standard_normal((t_out, B, 29)) / (B * t_out)
  eng.dZ[-1].interior().copy_(torch.as_tensor(np.transpose(dl, (1, 0, 2)), dtype=torch.float32))
  eng.backward()
  grads = eng.get_grads()
  ref = O.wav2letter_backward(acts, params, layers, dl, store=O.bf16_round)
  for i, ((gF, gb), (rF, rb)) in enumerate(zip(grads, ref)):
    mx, mean = scaled_err(gF, rF)
    assert mx < 4 * ULP and mean < 0.2 * ULP, (i, mx, mean)
    mx, _ = scaled_err(gb, rb)
    assert mx < 4 * ULP, (i, mx)


def test_split_backprop_bf16(dev):
  """Back-prop through a wide layer whose tile grid is small (the L8 situation): the reduction is split over
  channel chunks into fp32 slabs and a second pass sums, masks and rounds (st_conv1d_bwd_data_bf16_ws > 0)."""
  from speecht_amd import _lib
  layers = [(1, 1, 16, 48, True), (16, 1, 48, 512, True), (1, 1, 512, 29, False)]
  rng = np.random.default_rng(5)
  params = WL.xavier_params(layers, seed=5, bias_range=0.05)
  B, T = 2, 50
  x = rng.standard_normal((B, T, 16))
  eng = engine(layers, dev)
  eng.set_weights(params)
  eng.load_batch(x, [T] * B)
  assert _lib.load().st_conv1d_bwd_data_bf16_ws(eng.dZ[1].ref, eng.dZ[0].ref, 16) > 0
  eng.forward()
  _, acts = O.wav2letter_forward(x, params, layers, keep=True, store=O.bf16_round)
  dl = rng.standard_normal((T, B, 29)) / (B * T)
  eng.dZ[-1].interior().copy_(torch.as_tensor(np.transpose(dl, (1, 0, 2)), dtype=torch.float32))
  eng.backward()
  ref = O.wav2letter_backward(acts, params, layers, dl, store=O.bf16_round)
  for i, ((gF, gb), (rF, rb)) in enumerate(zip(eng.get_grads(), ref)):
    mx, mean = scaled_err(gF, rF)
    assert mx < 8 * ULP and mean < 0.5 * ULP, (i, mx, mean)
    assert scaled_err(gb, rb)[0] < 8 * ULP, i


def test_small_train_step_bf16(dev):
  case = WL.small_train_case()
  eng = engine(case['layers'], dev)
  eng.set_weights(case['params'])
  eng.load_batch(case['x'], case['seq_lens'])
  eng.set_labels(case['labels'])
  eng.forward()
  eng.ctc_loss_grad(1.0 / 3)
  eng.backward()
  grads = eng.get_grads()
  eng.apply_update(lr=1e-4)
  torch.cuda.synchronize()
  args = (case['x'], case['seq_lens'], case['labels'], case['params'], case['layers'], O.zero_opt_state(case['params']))
  ref = O.train_step(*args, lr=1e-4, store=O.bf16_round)
  ref32 = O.train_step(*args, lr=1e-4)
  logits = eng.logits_time_major().cpu().numpy()
  mx, mean = scaled_err(logits, ref['logits'])
  assert mx < 8 * ULP and mean < 0.5 * ULP, (mx, mean)
  loss = eng.loss.cpu().numpy()
  np.testing.assert_allclose(loss, ref['loss'], rtol=2e-3)         # same bf16 model
  np.testing.assert_allclose(loss, ref32['loss'], rtol=2e-2)       # vs full precision: quantisation noise only
  assert float(eng.stats[0]) == pytest.approx(ref['grad_norm'], rel=2e-2)
  for i, ((gF, gb), (rF, rb)) in enumerate(zip(grads, ref['grads'])):
    mx, mean = scaled_err(gF, rF)
    assert mx < 16 * ULP and mean < 1.0 * ULP, (i, mx, mean)
    assert scaled_err(gb, rb)[0] < 16 * ULP, i
  # masters stay fp32: the Adam step moves every weight by at most lr
  for (pF, pb), (rF, rb), (oF, ob) in zip(eng.get_weights(), ref['params'], case['params']):
    assert np.max(np.abs(pF - oF)) <= 1.01e-4 and np.max(np.abs(pF - rF)) < 2.1e-4


def test_full_width_forward_bf16_vs_fp32_oracle(dev):
  """Real widths, B=2 ragged: bf16 storage costs < 2 % of the logit scale against the fp32 reference math."""
  layers = WL.w2l_layers(80)
  params = WL.xavier_params(layers, seed=42)
  x, seq_lens, _ = WL.make_batch([201, 160], 80, seed=2)
  eng = engine(layers, dev)
  eng.set_weights(params)
  eng.load_batch(x, seq_lens)
  # 202 output rows leave most CUs without a tile: the wide layers split their reduction (fp32 slabs)
  from speecht_amd import _lib
  assert _lib.load().st_conv1d_fwd_bf16_ws(eng.X[8].ref, eng.X[9].ref, 32) > 0
  eng.forward()
  logits = eng.logits_time_major().cpu().numpy()
  ref_b = O.wav2letter_forward(x, params, layers, store=O.bf16_round)
  ref = O.wav2letter_forward(x, params, layers)
  mx, mean = scaled_err(logits, ref_b)
  assert mx < 8 * ULP and mean < 0.5 * ULP, (mx, mean)
  assert scaled_err(logits, ref)[0] < 2e-2
  dec, _ = eng.greedy_decode()
  ref_dec, _ = O.ctc_greedy_decode(logits.astype(np.float64), np.asarray(seq_lens) // 2)
  assert dec == ref_dec


def test_bf16_entry_points_reject_bad_arguments(dev):
  from speecht_amd import _lib
  eng = engine([(7, 1, 16, 16, True), (1, 1, 16, 29, False)], dev)
  eng.load_batch(np.zeros((1, 20, 16)), [20])
  with pytest.raises(_lib.SpeechtHipError, match='null argument'):
    _lib.call('st_conv1d_nwc_fwd_bf16', eng.X[0].ref, None, None, None, 7, 1, 3, 1, eng.X[1].ref, None, None, None)
  with pytest.raises(_lib.SpeechtHipError, match='workspace'):
    _lib.call('st_conv1d_nwc_bwd_filter_bf16', eng.X[0].ref, eng._ptr(eng.Xb[0]), eng.dZ[0].ref, eng._ptr(eng.dZb[0]),
              7, 1, 3, eng._ptr(eng.grads), eng._ptr(eng.grads), None, 0, None)


@pytest.mark.parametrize('mode', ['fp32', 'bf16x6', 'bf16'])
def test_second_step_uses_updated_weights(dev, mode):
  """Every derived copy of the filters (flipped operand for back-prop, bf16 copies, split planes) must be
  rebuilt after an Adam step: step 2 of a running engine equals step 1 of a fresh engine that was handed the
  updated weights."""
  case = WL.small_train_case()

  def one_step(eng):
    eng.load_batch(case['x'], case['seq_lens'])
    eng.set_labels(case['labels'])
    eng.forward()
    eng.ctc_loss_grad(1.0 / 3)
    eng.backward()
    return eng.loss.cpu().numpy().copy(), eng.get_grads()

  a = engine(case['layers'], dev, mode)
  a.set_weights(case['params'])
  one_step(a)
  a.apply_update(lr=1e-2)                      # a large step so that stale operands would show
  loss_a, grads_a = one_step(a)
  b = engine(case['layers'], dev, mode)
  b.set_weights(a.get_weights())
  loss_b, grads_b = one_step(b)
  np.testing.assert_array_equal(loss_a, loss_b)
  for (fa, ba), (fb, bb) in zip(grads_a, grads_b):
    np.testing.assert_array_equal(fa, fb)
    np.testing.assert_array_equal(ba, bb)


@pytest.mark.parametrize('mode', ['bf16', 'bf16x6'])
def test_training_reduces_loss_like_fp32(dev, mode):
  """40 Adam steps on one small batch: the loss curve of the alternative arithmetic tracks the fp32 path
  (same start, same end within 3 % for bf16 activations, 0.1 % for the fp32-accurate split)."""
  case = WL.small_train_case()
  curves = {}
  for m in ('fp32', mode):
    eng = engine(case['layers'], dev, m)
    eng.set_weights(case['params'])
    eng.load_batch(case['x'], case['seq_lens'])
    eng.set_labels(case['labels'])
    losses = []
    for _ in range(40):
      eng.forward()
      eng.ctc_loss_grad(1.0 / 3)
      eng.backward()
      eng.apply_update(lr=1e-3)
      losses.append(float(eng.loss.mean()))
    curves[m] = np.array(losses)
  ref, got = curves['fp32'], curves[mode]
  assert ref[-1] < 0.7 * ref[0] and got[-1] < 0.7 * got[0]
  tol = 3e-2 if mode == 'bf16' else 1e-3
  assert abs(got[0] - ref[0]) <= tol * ref[0] and abs(got[-1] - ref[-1]) <= tol * ref[-1], (ref[[0, -1]], got[[0, -1]])


@pytest.mark.parametrize('frames', [[1], [7, 3, 5], [48, 47, 2, 31, 96]])
def test_odd_shapes_bf16(dev, frames):
  """39-feature input, channel counts that are not multiples of 16, utterances shorter than the filters, B = 1
  in the bf16-activation mode, against the oracle's bf16 storage model."""
  layers = WL.w2l_layers(39, width=24, fc=40)
  params = WL.xavier_params(layers, seed=3)
  rng = np.random.default_rng(len(frames))
  x = np.zeros((len(frames), max(frames), 39))
  for b, t in enumerate(frames):
    x[b, :t] = rng.standard_normal((t, 39))
  labels = [list(rng.integers(0, 28, max(0, (t // 2) // 3))) for t in frames]
  eng = engine(layers, dev)
  eng.set_weights(params)
  eng.load_batch(x, frames)
  eng.set_labels(labels)
  eng.forward()
  eng.ctc_loss_grad(1.0 / len(frames))
  eng.backward()
  eng.check_ctc_status()
  ref = O.train_step(x, np.asarray(frames), labels, params, layers, None, update=False, store=O.bf16_round)
  mx, mean = scaled_err(eng.logits_time_major().cpu().numpy(), ref['logits'])
  assert mx < 8 * ULP and mean < 0.5 * ULP, (mx, mean)
  np.testing.assert_allclose(eng.loss.cpu().numpy(), ref['loss'], rtol=5e-3, atol=1e-4)
  for i, ((gF, gb), (rF, rb)) in enumerate(zip(eng.get_grads(), ref['grads'])):
    if np.max(np.abs(rF)) > 1e-12:
      assert scaled_err(gF, rF)[0] < 16 * ULP, i
      assert scaled_err(gb, rb)[0] < 16 * ULP, i


def test_panel_kernel_7tap_layers_bf16(dev):
  """The 7-tap 250 -> 250 layers at a batch that fills the chip run on conv_taps_bf16.hip (input panel resident in LDS) in the
  forward pass and in back-prop to the input.  Ragged batch, three such layers + head:
    * the launch trace names the kernel for forward L0, L1 and back-prop through L1 (the layers whose two tensors share a frame
      pitch);
    * on the device's OWN stored operands the stored results are round_bf16 of the float64 oracle's (<= 1 bf16 spacing on a
      rounding boundary, a handful of elements);
    * the rows between utterances stay zero, and the general kernel (st_set_tuning bf16_taps_panel = 1) stores the same first
      layer output up to such boundary cases."""
  from speecht_amd._lib import launch_trace, set_tuning
  layers = [(7, 1, 250, 250, True)] * 3 + [(1, 1, 250, 29, False)]
  frames = [1001, 1000, 777, 640, 1001, 333, 901, 5, 999, 1001]
  B, T = len(frames), max(frames)
  rng = np.random.default_rng(77)
  params = WL.xavier_params(layers, seed=5, bias_range=0.05)
  x = np.zeros((B, T, 250))
  for b, t in enumerate(frames):
    x[b, :t] = rng.standard_normal((t, 250))
  eng = engine(layers, dev)
  eng.set_weights(params)
  dl = rng.standard_normal((B, T, 29)) / (B * T)

  def run():
    eng.load_batch(x, frames)
    eng.forward()
    eng.dZ[-1].interior().copy_(torch.as_tensor(dl, dtype=torch.float32))
    eng.backward()
    torch.cuda.synchronize()
    return [eng.Xb[i].clone() for i in range(4)], [eng.dZb[i].clone() for i in range(3)]

  with launch_trace() as tr:
    Xp, dZp = run()
  taps = [l for l in tr.lines if l.startswith('conv_taps_bf16<128,128,64,panel>')]
  assert len(taps) >= 3, '\n'.join(tr.lines)          # (a fourth when the head's input tensor happens to share the frame pitch)
  try:
    set_tuning('bf16_taps_panel', 1)
    with launch_trace() as tr:
      Xg, dZg = run()
    assert not any(l.startswith('conv_taps_bf16') for l in tr.lines)
  finally:
    set_tuning('bf16_taps_panel', 0)

  def frames_of(t3, plane):
    return plane.view(t3.batch, t3.t_pitch, t3.c_pitch)[:, t3.halo:t3.halo + t3.frames].float().cpu().numpy().astype(np.float64)

  def same(a, b, what):                      # the two kernels sum in a different order: values on a rounding boundary may differ
    a, b = a.float(), b.float()
    d = (a - b).abs()
    scale = float(b.abs().max())
    assert float(d.max()) <= 2 * ULP * scale and int((d > 0).sum()) < 1e-3 * d.numel(), (what, float(d.max()) / scale)

  same(Xp[1], Xg[1], 'X1')                   # (same stored input; further up the two runs' inputs already differ in such elements)
  for i in (1, 2):
    t3 = eng.X[i]
    v = Xp[i].view(t3.batch, t3.t_pitch, t3.c_pitch).float()
    assert float(v[:, :t3.halo].abs().max()) == 0 and float(v[:, t3.halo + t3.frames:].abs().max()) == 0, 'halo rows of X%d' % i
    assert float(v[:, :, 250:].abs().max()) == 0, 'padding channels of X%d' % i
  v = dZp[0].view(eng.dZ[0].batch, eng.dZ[0].t_pitch, eng.dZ[0].c_pitch).float()
  assert float(v[:, :eng.dZ[0].halo].abs().max()) == 0 and float(v[:, eng.dZ[0].halo + eng.dZ[0].frames:].abs().max()) == 0

  # against the oracle on the stored operands
  F64 = [(O.bf16_round(F.astype(np.float64)), b.astype(np.float64)) for F, b in params]
  for i in (0, 1):
    y = O.conv1d_same_fwd(frames_of(eng.X[i], Xp[i])[:, :, :250], F64[i][0], F64[i][1], 1, True)
    mx, mean = scaled_err(frames_of(eng.X[i + 1], Xp[i + 1])[:, :, :250], O.bf16_round(y))
    assert mx <= 2.01 * ULP and mean < 0.01 * ULP, ('forward', i, mx / ULP, mean / ULP)
  xs1 = frames_of(eng.X[1], Xp[1])[:, :, :250]
  dx, _, _ = O.conv1d_same_bwd(xs1, F64[1][0], None, frames_of(eng.dZ[1], dZp[1])[:, :, :250], 1, relu=False, need_dx=True)
  mx, mean = scaled_err(frames_of(eng.dZ[0], dZp[0])[:, :, :250], O.bf16_round(dx * (xs1 > 0)))
  assert mx <= 2.01 * ULP and mean < 0.01 * ULP, ('back-prop', mx / ULP, mean / ULP)
