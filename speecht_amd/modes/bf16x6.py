"""bf16x6 mode (experimental, opt-in: ST_CONV_MODE=bf16x6): the layers that stay GEMMs -- at the training shapes the 2000 x 2000
layer -- as fp32-accurate products of three exact bf16 pieces per operand, six terms on the bf16 matrix pipe with fp32
accumulation (csrc/conv_bf16.hip, NP = 3); everything else as in the fp32 mode it derives from."""
import torch

from .._lib import call
from ..engine_buffers import _round_up
from .fp32 import Fp32Mode


class Bf16x6Mode(Fp32Mode):
  _one_tap_in_place = False
  _flip_every_layer = True
  _top_gradient_beside = False
  shape_attrs = ()                  # (its weight planes are re-chosen per shape in `_alloc_planes`: described anew every time)

  def _workspace_bytes(self, lib):
    ws = super()._workspace_bytes(lib)
    return max([ws] + [lib.st_exp_conv1d_bwd_data_bf16x6_ws(self.e.dZ[i].ref, self.e.dZ[i - 1].ref, l.width)
                       for i, l in enumerate(self.e.layers) if i > 0])

  def _alloc_mode_planes(self):
    self._alloc_planes()

  def prepare_forward_graph(self):
    if not self.e._wplanes_fresh:
      self._refresh_wplanes()
    super().prepare_forward_graph()

  # ---- which layers take the bf16x6 kernels ------------------------------------------------------------------------
  def _in_fft(self, i):
    return self.e.fft_conv and i in getattr(self.e, '_fft_layers', ())

  def _x6_fwd(self, i):
    return self.e.layers[i].n_pad % 128 == 0 and not self._in_fft(i)

  def _x6_bwd(self, i):
    l = self.e.layers[i]
    return (i > 0 and l.nt_pad % 128 == 0 and l.width * l.cout_pitch >= 256 and
            not self._in_fft(i))

  def _x6_wgrad(self, i):
    l = self.e.layers[i]
    tiles = -(-(l.width * l.cin_pitch) // 128) * (l.n_pad // 128)
    return (i > 0 and l.stride == 1 and l.n_pad % 128 == 0 and tiles >= 192 and
            not self._in_fft(i))

  def _alloc_planes(self):
    self.e.Xp = {i: self.e._planes('Xp%d' % i, self.e.X[i].buf.numel()) for i in range(len(self.e.layers)) if self._x6_fwd(i)}
    self.e.dZp = {i: self.e._planes('dZp%d' % i, self.e.dZ[i].buf.numel()) for i in range(len(self.e.layers)) if self._x6_bwd(i)}
    # filter gradient: transposed (reduction-major) planes of the layer input and of dz
    self.e.tq, self.e.XTp, self.e.dZTp = {}, {}, {}
    for i, l in enumerate(self.e.layers):
      if self._x6_wgrad(i):
        tq = _round_up(max(self.e.X[i].t_pitch, self.e.dZ[i].frames), 32)
        red = self.e.X[i].batch * tq
        self.e.tq[i] = tq
        self.e.XTp[i] = self.e._planes('XTp%d' % i, l.cin_pitch * red + 4096)
        self.e.dZTp[i] = self.e._planes('dZTp%d' % i, l.n_pad * red)
    # weight planes of exactly the layers that run on this path for the current shape (the frequency-domain set
    # depends on the shape); buffers are kept across shapes
    if not hasattr(self.e, '_wp_store'):
      self.e._wp_store, self.e._wtp_store = {}, {}
    def kept(store, i, numel):
      if i not in store:
        store[i] = torch.zeros(numel, dtype=torch.bfloat16, device=self.e.device)
      return store[i]
    self.e.Wp = {i: kept(self.e._wp_store, i, 3 * l.k_pad * l.n_pad) for i, l in enumerate(self.e.layers) if self._x6_fwd(i)}
    self.e.WTp = {i: kept(self.e._wtp_store, i, 3 * l.kt_pad * l.nt_pad) for i, l in enumerate(self.e.layers) if self._x6_bwd(i)}
    self.e._wplanes_fresh = False
    self.e._wtplanes_fresh = False

  def _refresh_wplanes(self):
    for i, wp in self.e.Wp.items():
      l = self.e.layers[i]
      pf, _ = self.e._slice(self.e.params, i)
      call('st_exp_split3_transpose_bf16', self.e._ptr(pf), l.k_pad, l.n_pad, self.e._ptr(wp), self.e.stream_ptr)
    self.e._wplanes_fresh = True

  def _refresh_wtplanes(self):
    for i, wp in self.e.WTp.items():
      l = self.e.layers[i]
      call('st_exp_split3_transpose_bf16', self.e._ptr(self.e.packed_t[i]), l.kt_pad, l.nt_pad, self.e._ptr(wp), self.e.stream_ptr)
    self.e._wtplanes_fresh = True

  # ---- the per-layer hooks of Fp32Mode.forward / backward ------------------------------------------------------------
  def _forward_prologue(self):
    if not self.e._wplanes_fresh:
      self._refresh_wplanes()
    if self._x6_fwd(0):
      call('st_exp_split3_bf16', self.e._ptr(self.e.X[0].buf), self.e.X[0].buf.numel(), self.e._ptr(self.e.Xp[0]), self.e.stream_ptr)

  def _x6_forward_layer(self, i, pb):
    l, s = self.e.layers[i], self.e.stream_ptr
    if i > 0 and not self._x6_fwd(i - 1):
      call('st_exp_split3_bf16', self.e._ptr(self.e.X[i].buf), self.e.X[i].buf.numel(), self.e._ptr(self.e.Xp[i]), s)
    yp = self.e._ptr(self.e.Xp[i + 1]) if (i + 1 < len(self.e.layers) and self._x6_fwd(i + 1)) else None
    call('st_exp_conv1d_fwd_bf16x6', self.e.X[i].ref, self.e._ptr(self.e.Xp[i]), self.e._ptr(self.e.Wp[i]), self.e._ptr(pb),
         l.width, l.stride, self.e.geo[i][2], int(l.relu), self.e.X[i + 1].ref, yp, s)

  def _backward_prologue(self):
    self._wait_bwd_operands()                   # the split planes are derived from all transposed copies at once

  def _x6_filter_gradient(self, i, gf, gb, need_bias):
    l, s = self.e.layers[i], self.e.stream_ptr
    tq, red = self.e.tq[i], self.e.X[i].batch * self.e.tq[i]
    call('st_exp_transpose_split3_bf16', self.e.X[i].ref, 0, self.e.X[i].t_pitch, tq, l.cin_pitch * red + 4096,
         self.e._ptr(self.e.XTp[i]), s)
    call('st_exp_transpose_split3_bf16', self.e.dZ[i].ref, self.e.dZ[i].halo, self.e.dZ[i].frames, tq, l.n_pad * red,
         self.e._ptr(self.e.dZTp[i]), s)
    call('st_exp_conv1d_bwd_filter_bf16x6', self.e._ptr(self.e.XTp[i]), self.e._ptr(self.e.dZTp[i]), self.e.X[i].batch, tq,
         l.width, l.cin_pitch, self.e.X[i].halo - self.e.geo[i][2], l.cout, self.e._ptr(gf), s)
    if need_bias:
      call('st_bias_grad_f32', self.e.dZ[i].ref, self.e._ptr(gb), self.e._ptr(self.e.wgrad_ws), self.e.wgrad_ws.numel() * 4, s)

  def _x6_back_prop(self, i):
    l, s = self.e.layers[i], self.e.stream_ptr
    act = self.e.X[i].ref if self.e.layers[i - 1].relu else None
    if not self.e._wtplanes_fresh:
      self._refresh_wtplanes()
    if not (i + 1 < len(self.e.layers) and self._x6_bwd(i + 1)):       # producer was not on this path
      call('st_exp_split3_bf16', self.e._ptr(self.e.dZ[i].buf), self.e.dZ[i].buf.numel(), self.e._ptr(self.e.dZp[i]), s)
    dxp = self.e._ptr(self.e.dZp[i - 1]) if self._x6_bwd(i - 1) else None
    call('st_exp_conv1d_bwd_data_bf16x6', self.e.dZ[i].ref, self.e._ptr(self.e.dZp[i]), self.e._ptr(self.e.WTp[i]), l.width,
         self.e.geo[i][2], act, self.e.dZ[i - 1].ref, dxp, self.e._ptr(self.e.wgrad_ws), self.e.wgrad_ws.numel() * 4, s)
