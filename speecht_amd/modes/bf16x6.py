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

  def _workspace_bytes(self, lib):
    ws = super()._workspace_bytes(lib)
    return max([ws] + [lib.st_exp_conv1d_bwd_data_bf16x6_ws(self.dZ[i].ref, self.dZ[i - 1].ref, l.width)
                       for i, l in enumerate(self.layers) if i > 0])

  def _alloc_mode_planes(self):
    self._alloc_planes()

  def prepare_forward_graph(self):
    if not self._wplanes_fresh:
      self._refresh_wplanes()
    super().prepare_forward_graph()

  # ---- which layers take the bf16x6 kernels ------------------------------------------------------------------------
  def _in_fft(self, i):
    return self.fft_conv and i in getattr(self, '_fft_layers', ())

  def _x6_fwd(self, i):
    return self.layers[i].n_pad % 128 == 0 and not self._in_fft(i)

  def _x6_bwd(self, i):
    l = self.layers[i]
    return (i > 0 and l.nt_pad % 128 == 0 and l.width * l.cout_pitch >= 256 and
            not self._in_fft(i))

  def _x6_wgrad(self, i):
    l = self.layers[i]
    tiles = -(-(l.width * l.cin_pitch) // 128) * (l.n_pad // 128)
    return (i > 0 and l.stride == 1 and l.n_pad % 128 == 0 and tiles >= 192 and
            not self._in_fft(i))

  def _alloc_planes(self):
    self.Xp = {i: self._planes('Xp%d' % i, self.X[i].buf.numel()) for i in range(len(self.layers)) if self._x6_fwd(i)}
    self.dZp = {i: self._planes('dZp%d' % i, self.dZ[i].buf.numel()) for i in range(len(self.layers)) if self._x6_bwd(i)}
    # filter gradient: transposed (reduction-major) planes of the layer input and of dz
    self.tq, self.XTp, self.dZTp = {}, {}, {}
    for i, l in enumerate(self.layers):
      if self._x6_wgrad(i):
        tq = _round_up(max(self.X[i].t_pitch, self.dZ[i].frames), 32)
        red = self.X[i].batch * tq
        self.tq[i] = tq
        self.XTp[i] = self._planes('XTp%d' % i, l.cin_pitch * red + 4096)
        self.dZTp[i] = self._planes('dZTp%d' % i, l.n_pad * red)
    # weight planes of exactly the layers that run on this path for the current shape (the frequency-domain set
    # depends on the shape); buffers are kept across shapes
    if not hasattr(self, '_wp_store'):
      self._wp_store, self._wtp_store = {}, {}
    def kept(store, i, numel):
      if i not in store:
        store[i] = torch.zeros(numel, dtype=torch.bfloat16, device=self.device)
      return store[i]
    self.Wp = {i: kept(self._wp_store, i, 3 * l.k_pad * l.n_pad) for i, l in enumerate(self.layers) if self._x6_fwd(i)}
    self.WTp = {i: kept(self._wtp_store, i, 3 * l.kt_pad * l.nt_pad) for i, l in enumerate(self.layers) if self._x6_bwd(i)}
    self._wplanes_fresh = False
    self._wtplanes_fresh = False

  def _refresh_wplanes(self):
    for i, wp in self.Wp.items():
      l = self.layers[i]
      pf, _ = self._slice(self.params, i)
      call('st_exp_split3_transpose_bf16', self._ptr(pf), l.k_pad, l.n_pad, self._ptr(wp), self.stream_ptr)
    self._wplanes_fresh = True

  def _refresh_wtplanes(self):
    for i, wp in self.WTp.items():
      l = self.layers[i]
      call('st_exp_split3_transpose_bf16', self._ptr(self.packed_t[i]), l.kt_pad, l.nt_pad, self._ptr(wp), self.stream_ptr)
    self._wtplanes_fresh = True

  # ---- the per-layer hooks of Fp32Mode.forward / backward ------------------------------------------------------------
  def _forward_prologue(self):
    if not self._wplanes_fresh:
      self._refresh_wplanes()
    if self._x6_fwd(0):
      call('st_exp_split3_bf16', self._ptr(self.X[0].buf), self.X[0].buf.numel(), self._ptr(self.Xp[0]), self.stream_ptr)

  def _x6_forward_layer(self, i, pb):
    l, s = self.layers[i], self.stream_ptr
    if i > 0 and not self._x6_fwd(i - 1):
      call('st_exp_split3_bf16', self._ptr(self.X[i].buf), self.X[i].buf.numel(), self._ptr(self.Xp[i]), s)
    yp = self._ptr(self.Xp[i + 1]) if (i + 1 < len(self.layers) and self._x6_fwd(i + 1)) else None
    call('st_exp_conv1d_fwd_bf16x6', self.X[i].ref, self._ptr(self.Xp[i]), self._ptr(self.Wp[i]), self._ptr(pb),
         l.width, l.stride, self.geo[i][2], int(l.relu), self.X[i + 1].ref, yp, s)

  def _backward_prologue(self):
    self._wait_bwd_operands()                   # the split planes are derived from all transposed copies at once

  def _x6_filter_gradient(self, i, gf, gb, need_bias):
    l, s = self.layers[i], self.stream_ptr
    tq, red = self.tq[i], self.X[i].batch * self.tq[i]
    call('st_exp_transpose_split3_bf16', self.X[i].ref, 0, self.X[i].t_pitch, tq, l.cin_pitch * red + 4096,
         self._ptr(self.XTp[i]), s)
    call('st_exp_transpose_split3_bf16', self.dZ[i].ref, self.dZ[i].halo, self.dZ[i].frames, tq, l.n_pad * red,
         self._ptr(self.dZTp[i]), s)
    call('st_exp_conv1d_bwd_filter_bf16x6', self._ptr(self.XTp[i]), self._ptr(self.dZTp[i]), self.X[i].batch, tq,
         l.width, l.cin_pitch, self.X[i].halo - self.geo[i][2], l.cout, self._ptr(gf), s)
    if need_bias:
      call('st_bias_grad_f32', self.dZ[i].ref, self._ptr(gb), self._ptr(self.wgrad_ws), self.wgrad_ws.numel() * 4, s)

  def _x6_back_prop(self, i):
    l, s = self.layers[i], self.stream_ptr
    act = self.X[i].ref if self.layers[i - 1].relu else None
    if not self._wtplanes_fresh:
      self._refresh_wtplanes()
    if not (i + 1 < len(self.layers) and self._x6_bwd(i + 1)):       # producer was not on this path
      call('st_exp_split3_bf16', self._ptr(self.dZ[i].buf), self.dZ[i].buf.numel(), self._ptr(self.dZp[i]), s)
    dxp = self._ptr(self.dZp[i - 1]) if self._x6_bwd(i - 1) else None
    call('st_exp_conv1d_bwd_data_bf16x6', self.dZ[i].ref, self._ptr(self.dZp[i]), self._ptr(self.WTp[i]), l.width,
         self.geo[i][2], act, self.dZ[i - 1].ref, dxp, self._ptr(self.wgrad_ws), self.wgrad_ws.numel() * 4, s)
