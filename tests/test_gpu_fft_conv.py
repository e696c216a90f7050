"""Frequency-domain convolution entry points (csrc/conv_fft.hip: st_conv1d_nwc_{fwd,bwd_data,bwd_filter}_fft_f32)
against the float64 oracle's tf.nn.conv1d('SAME') + bias + relu and its gradients (speech_model.py:155,173,177,78).
Tolerance: 2e-5 of the tensor's max (fp32 direct DFTs of N <= 128 points around exact-fp32 GEMMs)."""
import ctypes

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


@pytest.mark.parametrize('B,T,cin,cout,relu', [(3, 77, 130, 200, True), (2, 200, 250, 300, False), (5, 63, 250, 129, True)])
def test_fft_conv_matches_oracle(dev, B, T, cin, cout, relu):
  from speecht_amd import _lib
  from speecht_amd._lib import call
  from speecht_amd.engine import channel_pitch
  lib = _lib.load()
  W = 32
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
  gfwd = torch.empty(lib.st_conv1d_fft_filter_floats(W, cpi, cin, cout, 0), device=dev)
  gbwd = torch.empty(lib.st_conv1d_fft_filter_floats(W, cpi, cin, cout, 1), device=dev)
  call('st_conv1d_fft_filters_f32', P(packed), P(packed_t), W, cin, cout, cpi, cpo, P(tables), P(gfwd), P(gbwd), None)
  sf = torch.empty(lib.st_conv1d_fft_sf_floats(xt.ref, yt.ref, W), device=dev)
  sft = torch.empty_like(sf)
  zf = torch.empty(lib.st_conv1d_fft_zf_floats(dzt.ref, W), device=dev)
  ws = torch.empty(lib.st_conv1d_fft_ws(xt.ref, yt.ref, W) // 4 + 64, device=dev)

  call('st_conv1d_nwc_fwd_fft_f32', xt.ref, P(gfwd), P(bias_d), W, pl, int(relu), yt.ref, P(tables), P(sf), P(sft), P(ws),
       ws.numel() * 4, None)
  y = yt.interior().cpu().numpy()
  assert np.max(np.abs(y - y_ref)) < 2e-5 * np.max(np.abs(y_ref))
  # pad channels and halo rows of the output stay zero
  whole = yt.buf.view(B, yt.t_pitch, yt.c_pitch)
  assert float(whole[:, :, cout:].abs().max()) == 0.0 and float(whole[:, :yt.halo].abs().max()) == 0.0

  call('st_conv1d_fft_dz_spectra_f32', dzt.ref, W, P(tables), P(zf), None)
  call('st_conv1d_nwc_bwd_data_fft_f32', dzt.ref, P(zf), P(gbwd), W, pl, act.ref, dxt.ref, P(tables), P(ws), ws.numel() * 4, None)
  dx = dxt.interior().cpu().numpy()
  assert np.max(np.abs(dx - dx_ref)) < 2e-5 * np.max(np.abs(dx_ref))

  call('st_packed_dims', W, cpi, cout, ctypes.byref(kv), ctypes.byref(kp), ctypes.byref(npad))
  dpacked = torch.full((kp.value * npad.value,), 7.0, device=dev)
  call('st_conv1d_nwc_bwd_filter_fft_f32', xt.ref, dzt.ref, P(sf), P(sft), P(zf), W, P(tables), P(dpacked), P(ws), ws.numel() * 4, None)
  dFd = torch.empty(W * cin * cout, device=dev)
  call('st_unpack_filters_f32', P(dpacked), W, cin, cout, cpi, P(dFd), None)
  dF = dFd.view(W, cin, cout).cpu().numpy()
  assert np.max(np.abs(dF - dF_ref)) < 2e-5 * np.max(np.abs(dF_ref))
  # padding of the packed gradient is exactly zero (it is part of the flat gradient's global norm)
  G = dpacked.view(kp.value, npad.value)
  assert float(G[:, cout:].abs().max()) == 0.0
  V = G[:W * cpi].view(W, cpi, npad.value)
  if cpi > cin:
    assert float(V[:, cin:, :].abs().max()) == 0.0


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
