"""Frequency-domain convolution entry points (csrc/conv_fft.hip: st_conv1d_nwc_{fwd,bwd_data,bwd_filter}_fft_f32)
against the float64 oracle's tf.nn.conv1d('SAME') + bias + relu and its gradients (speech_model.py:155,173,177,78).
Tolerance: 2e-5 of the tensor's max (fp32 direct DFTs of N <= 128 points around exact-fp32 GEMMs)."""
import ctypes
import math

import numpy as np
import pytest

from oracle import w2l_oracle as O

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  return torch.device('cuda:0')


def dev_tensor(dev, batch, frames, channels, halo_l, halo_r, data=None):
  from speecht_amd.engine import DevTensor3
  storage = torch.zeros(DevTensor3.numel(batch, frames, channels, halo_l, halo_r), dtype=torch.float32, device=dev)
  t = DevTensor3(storage, batch, frames, channels, halo_l, halo_r)
  if data is not None:
    t.interior().copy_(torch.as_tensor(data, dtype=torch.float32))
  return t


# (the last case: 5 x 32 blocks = 160 rows, padded to 192 per bin -- the per-bin products run as a launch of whole 128-row tiles
#  and one of the last 64 rows, st::gemm_nn_batched)
@pytest.mark.parametrize('W,B,T,cin,cout,relu', [(32, 3, 77, 130, 200, True), (32, 2, 200, 250, 300, False), (32, 5, 63, 250, 129, True),
                                                 (7, 4, 150, 250, 250, True), (7, 2, 64, 130, 129, False), (12, 3, 100, 200, 250, True),
                                                 (32, 5, 2000, 130, 1000, True)])
@pytest.mark.parametrize('idft_valu', [0, 1])
def test_fft_conv_matches_oracle(dev, W, B, T, cin, cout, relu, idft_valu):
  # idft_valu: the filter gradient's inverse transform on the matrix pipe (filters_idft_mfma_kernel, round 6) or as one thread
  # per (c, o) on the vector ALU (st_set_tuning("filters_idft_valu", 1)); both the split and the 2 x 2 block form of the lag
  # products occur in the shapes above (spectra halves of 256 / 192 columns)
  from speecht_amd._lib import set_tuning
  set_tuning('filters_idft_valu', idft_valu)
  try:
    _fft_conv_matches_oracle(dev, W, B, T, cin, cout, relu)
  finally:
    set_tuning('filters_idft_valu', 0)


def _fft_conv_matches_oracle(dev, W, B, T, cin, cout, relu):
  from speecht_amd import _lib
  from speecht_amd._lib import call
  from speecht_amd.engine import channel_pitch
  lib = _lib.load()
  rng = np.random.default_rng(B * 100 + T)
  x = rng.standard_normal((B, T, cin))
  F = rng.standard_normal((W, cin, cout)) / np.sqrt(W * cin)
  bias = rng.standard_normal(cout) * 0.1
  y_ref = O.conv1d_same_fwd(x, F, bias, 1, relu)
  dy = rng.standard_normal(y_ref.shape)
  prev_act = rng.standard_normal(x.shape)                               # ReLU output of the layer below (mask source)
  dx_ref, dF_ref, _ = O.conv1d_same_bwd(x, F, y_ref, dy, 1, relu)
  dx_ref = dx_ref * (prev_act > 0)
  dz = dy * (y_ref > 0) if relu else dy

  _, pl, pr = O.same_padding(T, W, 1)
  P = lambda t: ctypes.c_void_p(t.data_ptr())
  cpi, cpo = channel_pitch(cin), channel_pitch(cout)
  kv, kp, npad = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
  call('st_packed_dims', W, cpi, cout, ctypes.byref(kv), ctypes.byref(kp), ctypes.byref(npad))
  packed = torch.zeros(kp.value * npad.value, device=dev)
  Fd = torch.as_tensor(F, dtype=torch.float32).to(dev).contiguous()
  call('st_pack_filters_f32', P(Fd), W, cin, cout, cpi, P(packed), None)
  call('st_packed_dims', W, cpo, cin, ctypes.byref(kv), ctypes.byref(kp), ctypes.byref(npad))
  packed_t = torch.zeros(kp.value * npad.value, device=dev)
  call('st_filters_flip_transpose_f32', P(packed), W, cin, cout, cpi, cpo, P(packed_t), None)
  bias_d = torch.zeros(2048, device=dev)
  bias_d[:cout] = torch.as_tensor(bias, dtype=torch.float32)

  xt = dev_tensor(dev, B, T, cin, pl, pr, x)
  yt = dev_tensor(dev, B, T, cout, W - 1 - pl, pl)
  act = dev_tensor(dev, B, T, cin, pl, pr, prev_act)
  dzt = dev_tensor(dev, B, T, cout, W - 1 - pl, pl, dz)
  dxt = dev_tensor(dev, B, T, cin, 3, 3)

  tables = torch.zeros(lib.st_conv1d_fft_table_floats(), device=dev)
  call('st_conv1d_fft_tables_f32', W, pl, P(tables), tables.numel(), None)
  gfwd = torch.empty(lib.st_conv1d_fft_filter_floats(W, cpi, cout), device=dev)
  call('st_conv1d_fft_filters_f32', P(packed), W, cin, cout, cpi, P(tables), P(gfwd), None)
  sf = torch.empty(lib.st_conv1d_fft_sf_floats(xt.ref, yt.ref, W), device=dev)
  zf = torch.empty(lib.st_conv1d_fft_zf_floats(dzt.ref, W), device=dev)
  ws = torch.empty(lib.st_conv1d_fft_ws(xt.ref, yt.ref, W) // 4 + 64, device=dev)

  call('st_conv1d_nwc_fwd_fft_f32', xt.ref, P(gfwd), P(bias_d), W, pl, int(relu), yt.ref, P(tables), P(sf), P(ws),
       ws.numel() * 4, None)
  y = yt.interior().cpu().numpy()
  assert np.max(np.abs(y - y_ref)) < 2e-5 * np.max(np.abs(y_ref))
  # pad channels and halo rows of the output stay zero
  whole = yt.buf.view(B, yt.t_pitch, yt.c_pitch)
  assert float(whole[:, :, cout:].abs().max()) == 0.0 and float(whole[:, :yt.halo].abs().max()) == 0.0

  call('st_conv1d_fft_dz_spectra_f32', dzt.ref, W, P(tables), P(zf), None)
  call('st_conv1d_nwc_bwd_data_fft_f32', dzt.ref, P(zf), P(gfwd), W, pl, act.ref, dxt.ref, P(tables), P(ws), ws.numel() * 4, None)
  dx = dxt.interior().cpu().numpy()
  assert np.max(np.abs(dx - dx_ref)) < 2e-5 * np.max(np.abs(dx_ref))

  call('st_packed_dims', W, cpi, cout, ctypes.byref(kv), ctypes.byref(kp), ctypes.byref(npad))
  dpacked = torch.full((kp.value * npad.value,), 7.0, device=dev)
  call('st_conv1d_nwc_bwd_filter_fft_f32', xt.ref, dzt.ref, P(sf), P(zf), W, P(tables), P(dpacked), P(ws), ws.numel() * 4, None)
  dFd = torch.empty(W * cin * cout, device=dev)
  call('st_unpack_filters_f32', P(dpacked), W, cin, cout, cpi, P(dFd), None)
  dF = dFd.view(W, cin, cout).cpu().numpy()
  assert np.max(np.abs(dF - dF_ref)) < 2e-5 * np.max(np.abs(dF_ref))
  # bias gradient read off bin 0 of the same spectra (pads written as zeros)
  db = torch.full((npad.value,), 7.0, device=dev)
  call('st_conv1d_fft_bias_grad_f32', dzt.ref, W, P(zf), P(db), None)
  db_ref = dz.reshape(-1, cout).sum(axis=0)
  assert np.max(np.abs(db[:cout].cpu().numpy() - db_ref)) < 2e-5 * np.max(np.abs(db_ref))
  assert float(db[cout:].abs().max()) == 0.0
  # padding of the packed gradient is exactly zero (it is part of the flat gradient's global norm)
  G = dpacked.view(kp.value, npad.value)
  assert float(G[:, cout:].abs().max()) == 0.0
  V = G[:W * cpi].view(W, cpi, npad.value)
  if cpi > cin:
    assert float(V[:, cin:, :].abs().max()) == 0.0


@pytest.mark.parametrize('planes', [3, 1])
@pytest.mark.parametrize('W,B,T,cin,cout,relu', [(32, 3, 77, 130, 200, True), (32, 2, 200, 250, 300, False), (7, 4, 150, 250, 250, True),
                                                 (12, 3, 100, 200, 250, True)])
def test_fft_conv_on_the_bf16_matrix_pipe(dev, planes, W, B, T, cin, cout, relu):
  """The frequency-domain layer with its per-bin products on the bf16 matrix pipe (st_conv1d_*_fft_planes, round 4).
  planes = 3: fp32 tensors, every spectrum value split exactly into three bf16 planes, six product terms -- against the
  float64 oracle at the fp32 path's own tolerance (2e-5 of the tensor maximum).
  planes = 1: bf16 tensors (BASELINE configs[3] arithmetic), ONE bf16 plane per spectrum -- against the oracle's block-DFT
  form with the same storage model (oracle.block_dft_conv(store=bf16_round): S, Z and G rounded to bf16, everything else
  exact): a stored bf16 result may sit one spacing from the model's where the value is on a rounding boundary (<= 2 bf16 ulp of
  the tensor maximum), the mean far below one; the fp32 gradients of filters and bias to 1e-3 / 2e-5."""
  from speecht_amd import _lib
  from speecht_amd._lib import call
  from speecht_amd.engine import channel_pitch
  lib = _lib.load()
  ULP = 2.0 ** -8
  rng = np.random.default_rng(B * 100 + T + planes)
  bf = planes == 1
  q = O.bf16_round if bf else (lambda a: a)
  x = q(rng.standard_normal((B, T, cin)))
  F = (rng.standard_normal((W, cin, cout)) / np.sqrt(W * cin)).astype(np.float32).astype(np.float64)
  bias = rng.standard_normal(cout) * 0.1
  prev_act = q(rng.standard_normal(x.shape))
  y_exact = O.conv1d_same_fwd(x, F, bias, 1, relu)
  dy = rng.standard_normal(y_exact.shape)
  dz = q(dy * (y_exact > 0) if relu else dy)
  y_ref, dx_ref, dF_ref, db_ref = O.block_dft_conv(x, F, bias, relu, dz=dz, prev_act=prev_act, store=O.bf16_round if bf else None)

  _, pl, pr = O.same_padding(T, W, 1)
  P = lambda t: ctypes.c_void_p(t.data_ptr())
  cpi = channel_pitch(cin)
  kv, kp, npad = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
  call('st_packed_dims', W, cpi, cout, ctypes.byref(kv), ctypes.byref(kp), ctypes.byref(npad))
  packed = torch.zeros(kp.value * npad.value, device=dev)
  call('st_pack_filters_f32', P(torch.as_tensor(F, dtype=torch.float32).to(dev).contiguous()), W, cin, cout, cpi, P(packed), None)
  bias_d = torch.zeros(2048, device=dev)
  bias_d[:cout] = torch.as_tensor(bias, dtype=torch.float32)
  xt = dev_tensor(dev, B, T, cin, pl, pr, x)
  yt = dev_tensor(dev, B, T, cout, W - 1 - pl, pl)
  act = dev_tensor(dev, B, T, cin, pl, pr, prev_act)
  dzt = dev_tensor(dev, B, T, cout, W - 1 - pl, pl, dz)
  dxt = dev_tensor(dev, B, T, cin, 3, 3)
  b16 = lambda t: t.buf.to(torch.bfloat16) if bf else None
  xb, actb, dzb = b16(xt), b16(act), b16(dzt)
  yb = torch.zeros_like(yt.buf, dtype=torch.bfloat16) if bf else None
  dxb = torch.zeros_like(dxt.buf, dtype=torch.bfloat16) if bf else None
  PB = lambda t: P(t) if t is not None else None

  tables = torch.zeros(lib.st_conv1d_fft_table_floats(), device=dev)
  call('st_conv1d_fft_tables_f32', W, pl, P(tables), tables.numel(), None)
  ge = lib.st_conv1d_fft_filter_plane_elems(W, cpi, cout)
  g = torch.empty(planes * ge, dtype=torch.bfloat16, device=dev)
  gt = torch.empty(planes * ge, dtype=torch.bfloat16, device=dev)
  call('st_conv1d_fft_filters_planes', P(packed), W, cin, cout, cpi, P(tables), P(g), P(gt), planes, None)
  sf = torch.empty(planes * lib.st_conv1d_fft_sf_floats(xt.ref, yt.ref, W), dtype=torch.bfloat16, device=dev)
  zf = torch.empty(planes * lib.st_conv1d_fft_zf_floats(dzt.ref, W), dtype=torch.bfloat16, device=dev)
  ws = torch.empty(lib.st_conv1d_fft_planes_ws(xt.ref, yt.ref, W, planes) // 4 + 64, device=dev)
  rows_pad = ctypes.c_int()
  blocks = ctypes.c_int()
  call('st_conv1d_fft_plan', W, T, B, None, None, ctypes.byref(blocks), None, ctypes.byref(rows_pad))
  dc = torch.full((rows_pad.value * npad.value,), 7.0, device=dev)

  def stored(t3, buf_b):
    v = (buf_b if bf else t3.buf).view(t3.batch, t3.t_pitch, t3.c_pitch)
    return v[:, t3.halo:t3.halo + t3.frames, :t3.channels].float().cpu().numpy().astype(np.float64), v

  call('st_conv1d_nwc_fwd_fft_planes', xt.ref, PB(xb), P(gt), P(bias_d), W, pl, int(relu), yt.ref, PB(yb), P(tables), P(sf), planes,
       P(ws), ws.numel() * 4, None)
  y, whole = stored(yt, yb)
  scale = np.max(np.abs(y_ref))
  if bf:
    d = np.abs(y - O.bf16_round(y_ref)) / scale
    assert d.max() <= 2.01 * ULP and d.mean() < 0.05 * ULP, ('forward', d.max() / ULP, d.mean() / ULP)
    assert np.max(np.abs(y - y_exact)) < 4 * ULP * np.max(np.abs(y_exact))        # and near the exact convolution: quantisation noise only
  else:
    assert np.max(np.abs(y - y_ref)) < 2e-5 * scale
  assert float(whole[:, :, cout:].float().abs().max()) == 0.0 and float(whole[:, :yt.halo].float().abs().max()) == 0.0

  call('st_conv1d_fft_dz_spectra_planes', dzt.ref, PB(dzb), W, P(tables), P(zf), planes, P(dc), None)
  call('st_conv1d_nwc_bwd_data_fft_planes', dzt.ref, P(zf), P(g), W, pl, act.ref, PB(actb), dxt.ref, PB(dxb), P(tables), planes, P(ws),
       ws.numel() * 4, None)
  dx, _ = stored(dxt, dxb)
  if bf:
    d = np.abs(dx - O.bf16_round(dx_ref)) / np.max(np.abs(dx_ref))
    assert d.max() <= 2.01 * ULP and d.mean() < 0.05 * ULP, ('back-prop to the input', d.max() / ULP, d.mean() / ULP)
  else:
    assert np.max(np.abs(dx - dx_ref)) < 2e-5 * np.max(np.abs(dx_ref))

  dpacked = torch.full((kp.value * npad.value,), 7.0, device=dev)
  call('st_conv1d_nwc_bwd_filter_fft_planes', xt.ref, dzt.ref, P(sf), P(zf), W, P(tables), P(dpacked), planes, P(ws), ws.numel() * 4, None)
  dFd = torch.empty(W * cin * cout, device=dev)
  call('st_unpack_filters_f32', P(dpacked), W, cin, cout, cpi, P(dFd), None)
  dF = dFd.view(W, cin, cout).cpu().numpy()
  # (bf16 spectra: an element within fp32 rounding of a bf16 boundary lands on the other side than in the float64 model)
  assert np.max(np.abs(dF - dF_ref)) < (1e-3 if bf else 2e-5) * np.max(np.abs(dF_ref))
  if bf:
    # round 5: the lag products read the spectra planes as they lie (transposing LDS reads, the rotated operand a register
    # shuffle) where half-spectra and outputs tile the kernel; the round-4 form -- reduction-minor copies first -- is the same sum
    # of the same bf16 products in another order
    from speecht_amd._lib import launch_trace, set_tuning
    with launch_trace() as tr:
      call('st_conv1d_nwc_bwd_filter_fft_planes', xt.ref, dzt.ref, P(sf), P(zf), W, P(tables), P(dpacked), planes, P(ws), ws.numel() * 4, None)
    direct = any(l.startswith('wgrad_tr_bf16<128,128,32,lag>') for l in tr.lines)
    assert direct == ((cpi + 63) // 64 * 64 % 128 == 0 and npad.value % 128 == 0), tr.lines       # (half-spectrum columns, output columns)
    if direct:
      copies = torch.full((kp.value * npad.value,), 7.0, device=dev)
      set_tuning('bf16_lag_copies', 1)
      try:
        with launch_trace() as tr2:
          call('st_conv1d_nwc_bwd_filter_fft_planes', xt.ref, dzt.ref, P(sf), P(zf), W, P(tables), P(copies), planes, P(ws), ws.numel() * 4, None)
      finally:
        set_tuning('bf16_lag_copies', 0)
      assert not any('lag' in l for l in tr2.lines) and any(l.startswith('gemm_nn_bf16<') for l in tr2.lines), tr2.lines
      assert float((copies - dpacked).abs().max()) < 1e-5 * float(dpacked.abs().max())
  db = torch.full((npad.value,), 7.0, device=dev)
  call('st_conv1d_fft_bias_grad_dc_f32', P(dc), B * blocks.value, cout, npad.value, P(db), None)
  assert np.max(np.abs(db[:cout].cpu().numpy() - db_ref)) < 2e-5 * np.max(np.abs(db_ref))
  assert float(db[cout:].abs().max()) == 0.0
  G = dpacked.view(kp.value, npad.value)
  assert float(G[:, cout:].abs().max()) == 0.0


@pytest.mark.parametrize('B,T,cin,cout', [(4, 1001, 80, 250), (3, 400, 128, 250), (2, 333, 40, 130)])
def test_stride2_layer_on_its_polyphase_view(dev, B, T, cin, cout):
  """The model's first layer (48 taps, stride 2; speech_model.py:279) through the stride-1 entry points: the input read
  as [B][T/2][2 * c_pitch] (frame pairs as channels), y[t] = sum_w F[w] x[2t + w - pl] = sum_{j,p} F[2j + p - shift]
  X2[t + j - pl2][p], i.e. ceil((W + shift) / 2) taps whose packed filters are the layer's own rows moved down by
  `shift` channel blocks.  Forward and filter gradient against the oracle's strided convolution."""
  from speecht_amd import _lib
  from speecht_amd._lib import call, Tensor3
  from speecht_amd.engine import channel_pitch
  lib = _lib.load()
  W, relu = 48, True
  rng = np.random.default_rng(B + T)
  x = rng.standard_normal((B, T, cin))
  F = rng.standard_normal((W, cin, cout)) / np.sqrt(W * cin)
  bias = rng.standard_normal(cout) * 0.1
  y_ref = O.conv1d_same_fwd(x, F, bias, 2, relu)
  dy = rng.standard_normal(y_ref.shape)
  _, dF_ref, _ = O.conv1d_same_bwd(x, F, y_ref, dy, 2, relu)
  dz = dy * (y_ref > 0)
  t_out, pl, pr = O.same_padding(T, W, 2)
  pl2 = (pl + 1) // 2
  shift = 2 * pl2 - pl
  W2 = (W + shift + 1) // 2
  P = lambda t: ctypes.c_void_p(t.data_ptr())
  cp, cpo = channel_pitch(cin), channel_pitch(cout)
  halo_l = pl + (pl & 1)
  halo_r = pr + ((halo_l + T + pr) & 1)
  xt = dev_tensor(dev, B, T, cin, halo_l, halo_r, x)
  x2 = Tensor3(xt.buf.data_ptr(), B, t_out, 2 * cp, xt.halo // 2, xt.t_pitch // 2, 2 * cp)
  x2ref = ctypes.byref(x2)
  yt = dev_tensor(dev, B, t_out, cout, 3, 2)
  dzt = dev_tensor(dev, B, t_out, cout, 3, 2, dz)
  kv, kp, npad = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
  call('st_packed_dims', W, cp, cout, ctypes.byref(kv), ctypes.byref(kp), ctypes.byref(npad))
  packed = torch.zeros(kp.value * npad.value, device=dev)
  Fd = torch.as_tensor(F, dtype=torch.float32).to(dev).contiguous()
  call('st_pack_filters_f32', P(Fd), W, cin, cout, cp, P(packed), None)
  n, o = W * cp * npad.value, shift * cp * npad.value
  packed2 = torch.zeros(2 * W2 * cp * npad.value, device=dev)
  packed2[o:o + n] = packed[:n]
  bias_d = torch.zeros(2048, device=dev)
  bias_d[:cout] = torch.as_tensor(bias, dtype=torch.float32)
  tables = torch.zeros(lib.st_conv1d_fft_table_floats(), device=dev)
  call('st_conv1d_fft_tables_f32', W2, pl2, P(tables), tables.numel(), None)
  gfwd = torch.empty(lib.st_conv1d_fft_filter_floats(W2, 2 * cp, cout), device=dev)
  call('st_conv1d_fft_filters_f32', P(packed2), W2, 2 * cp, cout, 2 * cp, P(tables), P(gfwd), None)
  sf = torch.empty(lib.st_conv1d_fft_sf_floats(x2ref, yt.ref, W2), device=dev)
  zf = torch.empty(lib.st_conv1d_fft_zf_floats(dzt.ref, W2), device=dev)
  ws = torch.empty(lib.st_conv1d_fft_ws(x2ref, yt.ref, W2) // 4 + 64, device=dev)
  call('st_conv1d_nwc_fwd_fft_f32', x2ref, P(gfwd), P(bias_d), W2, pl2, int(relu), yt.ref, P(tables), P(sf), P(ws),
       ws.numel() * 4, None)
  y = yt.interior().cpu().numpy()
  assert np.max(np.abs(y - y_ref)) < 2e-5 * np.max(np.abs(y_ref))
  call('st_conv1d_fft_dz_spectra_f32', dzt.ref, W2, P(tables), P(zf), None)
  dpacked2 = torch.full((2 * W2 * cp * npad.value,), 7.0, device=dev)
  call('st_conv1d_nwc_bwd_filter_fft_f32', x2ref, dzt.ref, P(sf), P(zf), W2, P(tables), P(dpacked2), P(ws), ws.numel() * 4, None)
  dpacked = torch.zeros(kp.value * npad.value, device=dev)
  dpacked[:n] = dpacked2[o:o + n]
  dFd = torch.empty(W * cin * cout, device=dev)
  call('st_unpack_filters_f32', P(dpacked), W, cin, cout, cp, P(dFd), None)
  dF = dFd.view(W, cin, cout).cpu().numpy()
  assert np.max(np.abs(dF - dF_ref)) < 2e-5 * np.max(np.abs(dF_ref))
  # the pad channels of the packed gradient are exactly zero
  G = dpacked[:n].view(W, cp, npad.value)
  assert float(G[:, :, cout:].abs().max()) == 0.0
  if cp > cin:
    assert float(G[:, cin:, :].abs().max()) == 0.0


def test_frequency_domain_layer_survives_shape_switching_and_weight_updates(dev):
  """Real training batches change (B, max_T) every step and Adam changes the weights every step: the engine keeps the
  filter spectra across shapes (they depend on the layer only), rebuilds them on a side stream after every update and
  re-describes the spectra buffers per shape.  After a walk through shapes and updates the engine must be
  bit-identical to a fresh engine that is handed the same weights and runs the last step only."""
  from speecht_amd.engine import Wav2LetterEngine
  from tests import workloads as WL
  layers = WL.w2l_layers(80)
  params = WL.xavier_params(layers, seed=42, dtype=np.float32)
  shapes = [[601] * 8, [1001] * 5, [333] * 2, [601] * 8]            # the third is too small for the frequency path
  eng = Wav2LetterEngine(layers, device=dev)
  eng.set_weights(params)
  batches = [WL.make_batch(f, 80, seed=20 + k) for k, f in enumerate(shapes)]
  used = []
  for k, (x, seq, labels) in enumerate(batches):
    if k == len(batches) - 1:
      before = eng.get_weights()
    eng.load_batch(x.astype(np.float32), seq)
    eng.set_labels(labels)
    eng.forward()
    used.append(bool(eng.fft))
    eng.ctc_loss_grad(1.0 / len(labels))
    eng.backward()
    if k < len(batches) - 1:
      eng.apply_update(1e-3)
  torch.cuda.synchronize()
  assert used == [True, True, False, True]
  fresh = Wav2LetterEngine(layers, device=dev)
  fresh.set_weights(before)
  x, seq, labels = batches[-1]
  fresh.load_batch(x.astype(np.float32), seq)
  fresh.set_labels(labels)
  fresh.forward()
  fresh.ctc_loss_grad(1.0 / len(labels))
  fresh.backward()
  torch.cuda.synchronize()
  assert torch.equal(eng.X[-1].buf, fresh.X[-1].buf)
  assert torch.equal(eng.grads, fresh.grads)


def test_training_steps_agree_between_frequency_and_w_tap_kernels(dev):
  """Three clip + Adam steps with the frequency-domain layers (polyphase first layer, 7-tap and 32-tap layers, bias
  gradients from bin 0, filter-gradient chains on the side stream) against the same three steps on the W-tap kernels
  only: losses to 1e-5, every weight tensor to 2e-4 of its max -- one ReLU sign flip of a near-zero pre-activation
  between the two formulations moves individual filter taps by about that much after three steps (the full-size
  gradient tests take this discontinuity apart); a wrong operand, tap shift or stream order would be off by O(1)."""
  from speecht_amd.engine import Wav2LetterEngine
  from tests import workloads as WL
  layers = WL.w2l_layers(80)
  params = WL.xavier_params(layers, seed=7, dtype=np.float32)
  x, seq, labels = WL.make_batch([601] * 12, 80, seed=11)          # 3 612 output rows: every layer class takes the path
  runs = {}
  for fft in (True, False):
    eng = Wav2LetterEngine(layers, device=dev, fft_conv=fft)
    eng.set_weights(params)
    losses = []
    for _ in range(3):
      eng.load_batch(x.astype(np.float32), seq)
      eng.set_labels(labels)
      eng.forward()
      eng.ctc_loss_grad(1.0 / len(labels))
      eng.backward()
      eng.apply_update(1e-4)
      losses.append(eng.fetch_losses().copy())
    assert bool(eng.fft) == fft and (not fft or 0 in eng.fft)
    runs[fft] = (np.array(losses), eng.get_weights())
  np.testing.assert_allclose(runs[True][0], runs[False][0], rtol=1e-5)
  for (Fa, ba), (Fb, bb) in zip(runs[True][1], runs[False][1]):
    assert np.max(np.abs(Fa - Fb)) < 2e-4 * np.max(np.abs(Fb))
    assert np.max(np.abs(ba - bb)) < 2e-4 * max(np.max(np.abs(bb)), 1e-3)


@pytest.mark.parametrize('bt', [False, True])
@pytest.mark.parametrize('slots', [64, 96])
@pytest.mark.parametrize('bins,M,K,N', [(36, 256, 512, 512), (45, 256, 384, 512), (36, 128, 512, 512), (3, 256, 512, 512),
                                       (48, 256, 4096, 512), (48, 256, 512, 4096), (33, 256, 512, 512), (5, 64, 32, 128),
                                       (37, 192, 96, 256), (20, 256, 512, 512)])
def test_batched_products_as_one_persistent_stream_k_launch(dev, bins, M, K, N, slots, bt):
  """st_gemm_nn_batched_ws_f32: a launch whose 64 x 128 tiles would load the CUs unevenly (36 bins of the 7-tap layers: 576
  tiles, three on a quarter of the CUs and two on the rest) runs as ONE persistent launch that deals the (bin, tile, k-tile)
  list in equal runs to 8 x 64 or 8 x 96 workgroups (csrc/streamk_map.h); a tile cut into pieces is summed head + next + ...
  by the workgroup holding its start.  Against float64 matmul per bin; bit-identical across repetitions -- with scratch of any
  content, while another stream keeps part of the chip busy (uneven load: the hand-off must not depend on who runs when) and
  with the reading CUs' caches warm from the previous repetition; flags back at zero and no poll timed out; within fp32
  rounding of the plain launch.  Other shapes are forced through the kernel (st_set_tuning("streamk", 1)): whole tiles only,
  several pieces per tile (K = 4096 on 96 workgroups per XCD: 3 pieces), XCDs without tiles.  bt: the second operand given
  transposed ([n][k], what back-prop to the input reads the forward filter spectra as)."""
  from speecht_amd import _lib
  from speecht_amd._lib import call, launch_trace, set_tuning
  lib = _lib.load()
  rng = np.random.default_rng(bins * 1000 + K)
  A = torch.as_tensor(rng.standard_normal((bins, M, K)), dtype=torch.float32).to(dev)
  B = torch.as_tensor(rng.standard_normal((bins, K, N)) / np.sqrt(K), dtype=torch.float32).to(dev)
  Bop = B.transpose(1, 2).contiguous() if bt else B               # [bins][N][K] when transposed
  entry = 'st_gemm_nn_batched_bt_ws_f32' if bt else 'st_gemm_nn_batched_ws_f32'
  P = lambda t: ctypes.c_void_p(t.data_ptr())
  ws_bytes, ctrl_words = lib.st_gemm_nn_batched_ws_bytes(), lib.st_gemm_nn_batched_ctrl_bytes() // 4
  ws = torch.full((ws_bytes // 4,), float('nan'), dtype=torch.float32, device=dev)            # scratch: any content
  tiles = bins * (M // 64) * (N // 128)
  uneven = 256 < tiles <= 768 and math.ceil(tiles / 256) / (tiles / 256) > 1.2
  whole_rounds = bins == 48 and N == 4096                       # the 32-tap layer's forward products: the plain launch
  if not uneven and not whole_rounds:
    set_tuning('streamk', 1)
  set_tuning('streamk_slots', slots)
  side = torch.cuda.Stream(dev)
  busy = torch.randn(4096, 4096, device=dev)
  outs = []
  try:
    for rep in range(4):
      C = torch.full((bins, M, N), float('nan'), dtype=torch.float32, device=dev)
      if rep == 1:
        ws[:ctrl_words].zero_()                        # from here on the control words can be checked: they must come back to zero
      if rep >= 2:                                     # uneven load: a matmul of torch's on another stream takes CUs away
        with torch.cuda.stream(side):
          busy @ busy
      with launch_trace() as tr:
        call(entry, P(A), K, M * K, P(Bop), K * N, P(C), N, M * N, M, K, N, bins, P(ws), ws_bytes, None)
      torch.cuda.synchronize()
      outs.append(C)
  finally:
    set_tuning('streamk', 0)
    set_tuning('streamk_slots', 0)
  line = tr.lines[0]
  if whole_rounds:
    assert line.startswith('gemm_nn<') and ('fast-bt' in line) == bt, line
  else:
    assert line.startswith('gemm_nn_bins<64,128,2,2,bt> batched' if bt else 'gemm_nn_bins<64,128,2,2> batched') and ' streamk ' in line, line
    if bins == 36 and M == 256 and N == 512 and K == 512:
      assert ('wgs=512 upw=18' if slots == 64 else 'wgs=768 upw=12') in line, line
    assert int(ws[:ctrl_words].view(torch.int32).abs().sum()) == 0     # flags taken back, the timeout count still zero
  for o in outs[1:]:
    assert torch.equal(outs[0], o)
  ref = torch.matmul(A.double(), B.double())
  tol = 2e-6 * max(1.0, math.sqrt(K / 512.0))                    # fp32 chains: the rounding grows with the reduction length
  err = float((outs[0].double() - ref).abs().max() / ref.abs().max())
  assert err < tol, err
  plain = torch.empty_like(outs[0])
  if bt:                                                       # without scratch: the plain transposed-operand launch
    call(entry, P(A), K, M * K, P(Bop), K * N, P(plain), N, M * N, M, K, N, bins, None, 0, None)
  else:
    call('st_gemm_nn_batched_f32', P(A), K, M * K, P(B), K * N, P(plain), N, M * N, M, K, N, bins, None)
  torch.cuda.synchronize()
  assert float((plain.double() - ref).abs().max() / ref.abs().max()) < tol


def test_a_lost_stream_k_producer_poisons_its_tile_and_is_counted(dev):
  """a reader whose bounded poll runs out used to add whatever the producer's slot held and only bump a
  word nobody read.  With the test hook that makes every producer publish a wrong epoch (`streamk_test_drop`): the tiles that
  needed a hand-off come out NaN (never a silent sum with an unpublished partial), the others are right, the library's lost-count
  rises (st_streamk_lost_count / _fetch_async, what engine.fetch_losses raises on), and the next launch -- knob off, same
  scratch -- is correct again."""
  from speecht_amd import _lib
  from speecht_amd._lib import call, set_tuning
  lib = _lib.load()
  bins, M, K, N = 36, 256, 512, 512
  rng = np.random.default_rng(5)
  A = torch.as_tensor(rng.standard_normal((bins, M, K)), dtype=torch.float32).to(dev)
  B = torch.as_tensor(rng.standard_normal((bins, K, N)) / np.sqrt(K), dtype=torch.float32).to(dev)
  P = lambda t: ctypes.c_void_p(t.data_ptr())
  ws_bytes = lib.st_gemm_nn_batched_ws_bytes()
  ws = torch.zeros(ws_bytes // 4, dtype=torch.float32, device=dev)
  ref = torch.matmul(A.double(), B.double())

  def lost():
    n = ctypes.c_uint32(0)
    call('st_streamk_lost_count', ctypes.byref(n))
    return n.value

  before = lost()
  C = torch.zeros((bins, M, N), dtype=torch.float32, device=dev)
  set_tuning('streamk_test_drop', 1)
  try:
    call('st_gemm_nn_batched_ws_f32', P(A), K, M * K, P(B), K * N, P(C), N, M * N, M, K, N, bins, P(ws), ws_bytes, None)
    torch.cuda.synchronize()
  finally:
    set_tuning('streamk_test_drop', 0)
  dropped = lost() - before
  assert dropped > 0
  tiles = C.view(bins, M // 64, 64, N // 128, 128).permute(0, 1, 3, 2, 4).reshape(-1, 64 * 128)
  nan_tiles = torch.isnan(tiles).all(dim=1)
  mixed = torch.isnan(tiles).any(dim=1) & ~nan_tiles
  assert int(mixed.sum()) == 0 and 0 < int(nan_tiles.sum()) < tiles.shape[0], (int(mixed.sum()), int(nan_tiles.sum()))
  good = ~torch.isnan(C)
  assert float((C.double() - ref)[good].abs().max() / ref.abs().max()) < 2e-6
  # the pinned-word form the engine uses, ordered behind the stream's work
  host = torch.zeros(1, dtype=torch.int32, pin_memory=True)
  call('st_streamk_lost_fetch_async', ctypes.c_void_p(host.data_ptr()), None)
  torch.cuda.synchronize()
  assert int(host[0]) == before + dropped
  # same scratch, producers publishing again: exact, and nothing new is counted
  C2 = torch.zeros_like(C)
  call('st_gemm_nn_batched_ws_f32', P(A), K, M * K, P(B), K * N, P(C2), N, M * N, M, K, N, bins, P(ws), ws_bytes, None)
  torch.cuda.synchronize()
  assert not torch.isnan(C2).any() and float((C2.double() - ref).abs().max() / ref.abs().max()) < 2e-6
  assert lost() == before + dropped


@pytest.mark.parametrize('B,T,cin,cout,relu_below', [(3, 70, 2000, 2000, True), (2, 333, 2000, 29, True), (4, 40, 250, 2000, False),
                                                     (1, 5, 130, 64, True)])
def test_one_tap_back_prop_reads_the_forward_filters_transposed(dev, B, T, cin, cout, relu_below):
  """st_conv1d_1tap_bwd_data_bias_f32 (the two 1 x 1 layers on top, speech_model.py:288,292): dx = mask * (dz W^T) and the
  column sums of dx (the bias gradient of the layer below) straight from the layer's packed filters -- against the oracle
  and against the flipped / transposed-copy form it replaces (same arithmetic per element: equal to fp32 rounding of the
  reduction order, which is the same k order -> bit-identical dx)."""
  from speecht_amd import _lib
  from speecht_amd._lib import call
  from speecht_amd.engine import channel_pitch
  lib = _lib.load()
  rng = np.random.default_rng(B * 1000 + cout)
  x = rng.standard_normal((B, T, cin))
  F = rng.standard_normal((1, cin, cout)) / np.sqrt(cin)
  dz = rng.standard_normal((B, T, cout))
  prev_act = rng.standard_normal(x.shape)
  y = O.conv1d_same_fwd(x, F, np.zeros(cout), 1, False)
  dx_ref, _, _ = O.conv1d_same_bwd(x, F, y, dz, 1, False)
  if relu_below:
    dx_ref = dx_ref * (prev_act > 0)
  P = lambda t: ctypes.c_void_p(t.data_ptr())
  cpi, cpo = channel_pitch(cin), channel_pitch(cout)
  kv, kp, npad = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
  call('st_packed_dims', 1, cpi, cout, ctypes.byref(kv), ctypes.byref(kp), ctypes.byref(npad))
  packed = torch.zeros(kp.value * npad.value, device=dev)
  call('st_pack_filters_f32', P(torch.as_tensor(F, dtype=torch.float32).to(dev).contiguous()), 1, cin, cout, cpi, P(packed), None)
  call('st_packed_dims', 1, cpo, cin, ctypes.byref(kv), ctypes.byref(kp), ctypes.byref(npad))
  packed_t = torch.zeros(kp.value * npad.value, device=dev)
  call('st_filters_flip_transpose_f32', P(packed), 1, cin, cout, cpi, cpo, P(packed_t), None)
  npi = npad.value
  dzt = dev_tensor(dev, B, T, cout, 0, 0, dz)
  act = dev_tensor(dev, B, T, cin, 0, 0, prev_act)
  outs = []
  for transposed in (True, False):
    dxt = dev_tensor(dev, B, T, cin, 2, 1)
    dxt.buf.fill_(float('nan'))
    call('st_zero_halos_f32', dxt.ref, None)
    db = torch.full((npi,), 7.0, device=dev)
    ws = torch.empty(lib.st_conv1d_bwd_data_bias_ws(dzt.ref, dxt.ref, 1) // 4 + 64, device=dev)
    if transposed:
      call('st_conv1d_1tap_bwd_data_bias_f32', dzt.ref, P(packed), act.ref if relu_below else None, dxt.ref, P(db), P(ws),
           ws.numel() * 4, None)
    else:
      call('st_conv1d_nwc_bwd_data_bias_f32', dzt.ref, P(packed_t), 1, 0, act.ref if relu_below else None, dxt.ref, P(db), P(ws),
           ws.numel() * 4, None)
    torch.cuda.synchronize()
    dx = dxt.interior().cpu().numpy()
    assert np.max(np.abs(dx - dx_ref)) < 2e-5 * np.max(np.abs(dx_ref))
    assert np.max(np.abs(db[:cin].cpu().numpy() - dx_ref.reshape(-1, cin).sum(axis=0))) < 2e-5 * np.max(np.abs(dx_ref)) * np.sqrt(B * T)
    whole = dxt.buf.view(B, dxt.t_pitch, dxt.c_pitch)
    assert float(whole[:, :, cin:].nan_to_num(nan=0.0).abs().max()) == 0.0 if dxt.c_pitch > cin else True
    outs.append(dxt.interior().clone())
  assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('B,T,taps', [(16, 500, (7, 7, 5)), (32, 250, (7, 3, 7)), (2, 4096 // 8, (7, 7, 7))])
def test_inverse_transform_hands_its_frames_to_the_next_layers_forward_transform(dev, B, T, taps):
  """st_conv1d_nwc_fwd_fft_chain_f32: in a chain of frequency-domain layers the inverse transform of layer i (bias, ReLU) feeds the
  forward transform of layer i + 1 in registers (idft_dft_rows_kernel: the accumulator layout is a k-step order of the next DFT
  once its matrix has the columns permuted; halo frames of the neighbour blocks through LDS), and in back-prop ONE inverse
  transform per block over its whole window replaces the three-term overlap-add (spills through LDS), masks and transforms
  the frames again for the layer below (idft_ola_dft_rows_kernel) -- when the shapes allow -- at most 8
  blocks per utterance, batch x blocks a multiple of 128, a next window reaching <= 4 frames into a neighbour.  Against the
  separate kernels (st_set_tuning("no_fused_transforms", 1)): logits, every stored activation and every gradient (the filter
  gradients read the handed-over spectra) to fp32 rounding; against the float64 oracle at the usual bound; ragged last blocks
  (T not a multiple of 64), different taps per layer (other halos), pad channels."""
  from speecht_amd.engine import Wav2LetterEngine
  from speecht_amd._lib import launch_trace, set_tuning
  from tests import workloads as WL
  layers = [(5, 1, 40, 120, True), (taps[0], 1, 120, 128, True), (taps[1], 1, 128, 250, True), (taps[2], 1, 250, 128, True),
            (1, 1, 128, 29, False)]                          # W-tap bottom layer, three frequency-domain layers, the 1-tap classifier
  params = WL.xavier_params(layers, seed=9, bias_range=0.05, dtype=np.float32)
  x, seq, labels = WL.make_batch([T] * (B - 1) + [T - 37], 40, seed=77)
  runs = []
  for off in (0, 1):
    set_tuning('no_fused_transforms', off)
    try:
      eng = Wav2LetterEngine(layers, device=dev)
      eng.fft_min_rows_narrow, eng.fft_min_width = 0, 2
      eng.set_weights(params)
      eng.load_batch(x.astype(np.float32), seq)
      eng.set_labels(labels)
      with launch_trace() as tr:
        eng.forward()
        eng.ctc_loss_grad(1.0 / B)
        eng.backward()
      torch.cuda.synchronize()
      runs.append(([t.buf.clone() for t in eng.X], eng.grads.clone(), tr.lines, eng.get_grads(), eng.logits_time_major().cpu().numpy()))
    finally:
      set_tuning('no_fused_transforms', 0)
  fused = [l for l in runs[0][2] if l.startswith('idft_dft_rows<')]
  fits = (-(-T // 64)) <= 8 and (B * (-(-T // 64))) % 128 == 0
  assert set(eng.fft) == {1, 2, 3}
  assert len(fused) == (2 if fits else 0), '\n'.join(runs[0][2])             # layers 1 -> 2 and 2 -> 3
  assert not any(l.startswith('idft_dft_rows<') for l in runs[1][2])
  # back-prop: the window-form inverse with the overlap-add through LDS for all three layers, the dz spectra of the layer
  # below riding along where that layer is a frequency-domain one (3 -> 2, 2 -> 1; layer 0 is a W-tap layer)
  ola = [l for l in runs[0][2] if l.startswith('idft_ola_dft_rows<')]
  assert len(ola) == (3 if fits else 0) and sum(1 for l in ola if ',dz-spectra>' in l) == (2 if fits else 0), '\n'.join(runs[0][2])
  assert not any(l.startswith('idft_ola_dft_rows<') for l in runs[1][2])
  if fits:
    # only the bottom layer of the chain transforms its input itself, only the top one its dz
    assert sum(1 for l in runs[0][2] if l.startswith('dft_rows<')) == 2
    assert sum(1 for l in runs[1][2] if l.startswith('dft_rows<')) == 6
  for a, b in zip(runs[0][0], runs[1][0]):
    assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max()))
  assert float((runs[0][1] - runs[1][1]).abs().max()) <= 5e-6 * float(runs[1][1].abs().max())
  p64 = [(F.astype(np.float64), b.astype(np.float64)) for F, b in params]
  logits, acts = O.wav2letter_forward(x.astype(np.float64), p64, layers, keep=True)
  assert np.max(np.abs(runs[0][4] - logits)) < 1e-4
  loss, g_logits = O.ctc_loss_and_grad(logits, labels, seq // 2)         # (the engine hands CTC sequence_lengths // 2, like the reference)
  ref = O.wav2letter_backward(acts, p64, layers, g_logits / B)
  for (gF, gb), (rF, rb) in zip(runs[0][3], ref):
    assert np.max(np.abs(gF - rF)) <= 2e-4 * np.max(np.abs(rF)) and np.max(np.abs(gb - rb)) <= 2e-4 * max(np.max(np.abs(rb)), 1e-30)
