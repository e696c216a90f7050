"""bf16 activations (BASELINE configs[3] arithmetic; csrc/conv_bf16.hip NP = 1, csrc/wgrad_tr_bf16.hip, the 32-tap layer through
st_conv1d_*_fft_planes): activations and activation gradients in HBM as bf16, fp32 masters / accumulation / logits / CTC / Adam."""
import ctypes
import os

import torch

from .. import _lib
from .._lib import call
from .base import ModeBase


class Bf16Mode(ModeBase):

  def alloc(self, batch):
    # the wide long-filter layer in the frequency domain with its per-bin products on the bf16 matrix pipe
    self._fftb_layers = {i for i in range(len(self.layers)) if self._use_fft_bf16(i, batch, self.geo[i][1])}
    self._alloc_bf16()
    self._alloc_fft_bf16(batch)

  def forward(self):
    return self._forward_bf16()

  def backward(self, on_layer_done, wanted):
    self._join_side_stream()
    return self._backward_bf16(on_layer_done, wanted)

  def refresh_under_ctc(self):
    if not self._wtplanes_fresh and hasattr(self, 'WTb'):
      self._on_side_stream(lambda: self._refresh_bf16_filters(True))

  def refresh_after_update(self):
    if hasattr(self, 'Wb'):
      self._refresh_wb_after_update()              # the bf16 copies the next forward pass reads

  def prepare_forward_graph(self):
    if not self._wplanes_fresh:
      self._refresh_bf16_filters(False)            # derived operands are rebuilt outside the graph
    self._join_side_stream()
    if getattr(self, '_wb_ready', None):
      self._wb_ready.clear()                       # (covered by the join above)

  def _use_fft_bf16(self, i, batch, t_out):
    """bf16 activations (configs[3]): the 32-tap 250 -> 2000 layer runs as block DFTs + per-bin products on the bf16 matrix pipe
    (st_conv1d_*_fft_planes, one bf16 plane): 51.5 GFLOP per pass instead of the W-tap kernel's 513.  Only the wide
    long-filter layer: the narrow layers' W-tap bf16 kernels are launch-bound (~30 us), nothing to gain there."""
    l = self.layers[i]
    return (self.conv_mode == 'bf16' and self.fft_conv and os.environ.get('ST_FFT_BF16', '1') != '0' and i > 0 and
            l.stride == 1 and 16 <= l.width <= 33 and l.n_pad % 128 == 0 and batch * t_out >= self.fft_min_rows)

  def _alloc_fft_bf16(self, batch):
    lib = _lib.load()
    self.fftb = {}
    for i in sorted(self._fftb_layers):
      l = self.layers[i]
      t_in, t_out, pl, pr = self.geo[i]
      view = lambda name, numel, dtype=None: self._storage.view('fftb%d_%s' % (i, name), numel, dtype)
      bf = torch.bfloat16
      tables, fresh_tables = view('tables', lib.st_conv1d_fft_table_floats())
      if getattr(self, '_fftb_table_key', {}).get(i) != (l.width, pl):
        fresh_tables = True
      if not hasattr(self, '_fftb_table_key'):
        self._fftb_table_key = {}
      self._fftb_table_key[i] = (l.width, pl)
      ge = lib.st_conv1d_fft_filter_plane_elems(l.width, l.cin_pitch, l.cout)
      g, fresh_g = view('g', ge, bf)
      rows_pad, blocks = ctypes.c_int(), ctypes.c_int()
      call('st_conv1d_fft_plan', l.width, t_out, batch, None, None, ctypes.byref(blocks), None, ctypes.byref(rows_pad))
      f = dict(tables=tables, g=g, gt=view('gt', ge, bf)[0],
               sf=view('sf', lib.st_conv1d_fft_sf_floats(self.X[i].ref, self.X[i + 1].ref, l.width), bf)[0],
               zf=view('zf', lib.st_conv1d_fft_zf_floats(self.dZ[i].ref, l.width), bf)[0],
               dc=view('dc', rows_pad.value * l.n_pad)[0], rows=batch * blocks.value,
               ws=view('ws', lib.st_conv1d_fft_planes_ws(self.X[i].ref, self.X[i + 1].ref, l.width, 1) // 4 + 64)[0], pl=pl)
      if fresh_tables:
        call('st_conv1d_fft_tables_f32', l.width, pl, self._ptr(tables), tables.numel(), self.stream_ptr)
      if fresh_g:
        self._wplanes_fresh = False
      self.fftb[i] = f
    if set(self.fftb) != getattr(self, '_fftb_prev', None):
      self._wplanes_fresh = False
      self._wtplanes_fresh = False
    self._fftb_prev = set(self.fftb)

  def _alloc_bf16(self):
    L = len(self.layers)
    lib = _lib.load()
    # the filter gradients of the stride-1 layers read both planes as they lie (LDS transpose reads, csrc/wgrad_tr_bf16.hip) and
    # run up to `slack` rows past the last one: zeros behind every plane
    slack = lib.st_conv1d_bwd_filter_tr_bf16_slack_rows()
    self.Xb = [self._planes('Xb%d' % i, self.X[i].buf.numel(), 1, slack * self.X[i].c_pitch) for i in range(L)]
    self.dZb = [self._planes('dZb%d' % i, self.dZ[i].buf.numel(), 1, slack * self.dZ[i].c_pitch) for i in range(L)]
    self._wgrad_tr = [os.environ.get('ST_BF16_WGRAD_TR', '1') != '0' and
                      lib.st_conv1d_bwd_filter_tr_bf16_ws(self.X[i].ref, self.dZ[i].ref, l.width, l.stride, self.geo[i][2]) > 0
                      for i, l in enumerate(self.layers)]
    wgrad_ws = lambda i: (lib.st_conv1d_bwd_filter_tr_bf16_ws if self._wgrad_tr[i] else lib.st_conv1d_bwd_filter_bf16_ws)(
        self.X[i].ref, self.dZ[i].ref, self.layers[i].width, self.layers[i].stride, self.geo[i][2])
    ws = max(wgrad_ws(i) for i in range(L))
    ws = max([ws] + [lib.st_conv1d_bwd_data_bf16_ws(self.dZ[i].ref, self.dZ[i - 1].ref, l.width)
                     for i, l in enumerate(self.layers) if i > 0])
    ws = max([ws] + [lib.st_conv1d_fwd_bf16_ws(self.X[i].ref, self.X[i + 1].ref, l.width)
                     for i, l in enumerate(self.layers)])
    self.wgrad_ws_b, _ = self._storage.view('wgrad_ws_b', ws // 4 + 64)
    # the narrow layers' filter gradients run beside back-prop to the input on the side stream: their own scratch
    # (the classification layer beside its back-prop, as in fp32: measured, no gain here -- 3.15 ms either way)
    self._side_wgrad_bf16 = [i for i, l in enumerate(self.layers) if self.side_filter_gradient and i > 0 and l.cout <= 512 and l.cin <= 512]
    ws2 = max([0] + [wgrad_ws(i) for i in self._side_wgrad_bf16])
    self.wgrad_ws_b2 = self._storage.view('wgrad_ws_b2', ws2 // 4 + 64)[0] if ws2 else None
    self.wgrad_ws_b3 = self._storage.view('wgrad_ws_b3', ws2 // 4 + 64)[0] if ws2 else None    # second side stream
    if not hasattr(self, 'Wb'):
      z = lambda n: torch.zeros(n, dtype=torch.bfloat16, device=self.device)
      self.Wb = [z(l.k_pad * l.n_pad) for l in self.layers]
      self.WTb = [None] + [z(l.kt_pad * l.nt_pad) for l in self.layers[1:]]

  def _refresh_bf16_filters(self, transposed, layers=None):
    fftb = getattr(self, 'fftb', {})
    for i, l in enumerate(self.layers):
      if layers is not None and i not in layers:
        continue
      if i in fftb:
        # a frequency-domain layer: its filter spectra (one bf16 plane, both operand layouts) instead of the two bf16 copies
        if not transposed:
          f = fftb[i]
          call('st_conv1d_fft_filters_planes', self._ptr(self._slice(self.params, i)[0]), l.width, l.cin, l.cout, l.cin_pitch,
               self._ptr(f['tables']), self._ptr(f['g']), self._ptr(f['gt']), 1, self.stream_ptr)
        continue
      if transposed and i > 0:
        call('st_filters_bwd_bf16', self._ptr(self._slice(self.params, i)[0]), l.width, l.cin, l.cout, l.cin_pitch,
             l.cout_pitch, self._ptr(self.WTb[i]), self.stream_ptr)
      elif not transposed:
        call('st_filters_bf16', self._ptr(self._slice(self.params, i)[0]), l.k_pad, l.n_pad, self._ptr(self.Wb[i]),
             self.stream_ptr)
    if layers is not None:
      return
    if transposed:
      self._wtplanes_fresh = True
    else:
      self._wplanes_fresh = True

  def _refresh_wb_after_update(self):
    """After an update: the bottom layer's bf16 filter copy on the compute stream (the next forward pass needs it at
    once), the others on the side stream, bottom layer first, an event per layer -- the forward pass waits layer by
    layer instead of for the whole list (eleven small kernels, ~130 us end to end, during which the chip was idle)."""
    L = len(self.layers)
    self._wb_ready = {}
    self._refresh_bf16_filters(False, layers=[0])

    def rest():
      for i in range(1, L):
        self._refresh_bf16_filters(False, layers=[i])
        ev = torch.cuda.Event()
        ev.record(self._stream)
        self._wb_ready[i] = ev
    self._on_side_stream(rest)
    self._wplanes_fresh = True

  def _forward_bf16(self):
    s, L = self.stream_ptr, len(self.layers)
    main = self._stream if self._stream is not None else torch.cuda.current_stream(self.device)
    ready = getattr(self, '_wb_ready', None) or {}
    if not self._wplanes_fresh:
      self._join_side_stream()                     # (a rebuild still running there writes the same buffers)
      ready.clear()
      self._refresh_bf16_filters(False)
    call('st_cast_bf16', self._ptr(self.X[0].buf), self.X[0].buf.numel(), self._ptr(self.Xb[0]), s)
    for i, l in enumerate(self.layers):
      last = i + 1 == L
      if ready:
        # the side stream works bottom layer first: the first layers wait for their own copy, the fourth for all that
        # remain (by then the side stream is through; every wait costs the compute stream a few microseconds)
        if i >= 3:
          main.wait_event(ready[L - 1])
          ready.clear()
        elif i in ready:
          main.wait_event(ready.pop(i))
      if i in self.fftb and not last:
        f = self.fftb[i]
        call('st_conv1d_nwc_fwd_fft_planes', self.X[i].ref, self._ptr(self.Xb[i]), self._ptr(f['gt']), self._ptr(self._slice(self.params, i)[1]),
             l.width, f['pl'], int(l.relu), self.X[i + 1].ref, self._ptr(self.Xb[i + 1]), self._ptr(f['tables']), self._ptr(f['sf']), 1,
             self._ptr(f['ws']), f['ws'].numel() * 4, s)
        continue
      call('st_conv1d_nwc_fwd_ws_bf16', self.X[i].ref, self._ptr(self.Xb[i]), self._ptr(self.Wb[i]),
           self._ptr(self._slice(self.params, i)[1]), l.width, l.stride, self.geo[i][2], int(l.relu), self.X[i + 1].ref,
           None if last else self._ptr(self.Xb[i + 1]), self._ptr(self.X[i + 1].buf) if last else None,
           self._ptr(self.wgrad_ws_b), self.wgrad_ws_b.numel() * 4 if self.split_small_batches else 0, s)

  def _backward_bf16(self, on_layer_done, wanted=lambda i: True):
    s, L = self.stream_ptr, len(self.layers)
    if not self._wtplanes_fresh:
      self._refresh_bf16_filters(True)
    call('st_cast_bf16', self._ptr(self.dZ[L - 1].buf), self.dZ[L - 1].buf.numel(), self._ptr(self.dZb[L - 1]), s)
    side = False
    for i in reversed(range(L)):
      l = self.layers[i]
      gf, gb = self._slice(self.grads, i)
      beside = i in self._side_wgrad_bf16      # this layer's filter gradient runs beside its back-prop to the input

      if i in self.fftb:
        # frequency-domain layer: ONE transform of dz (bf16 spectra + the fp32 block sums) serves the filter gradient, the bias
        # gradient and back-prop to the input
        f = self.fftb[i]
        call('st_conv1d_fft_dz_spectra_planes', self.dZ[i].ref, self._ptr(self.dZb[i]), l.width, self._ptr(f['tables']), self._ptr(f['zf']), 1,
             self._ptr(f['dc']), s)
        call('st_conv1d_nwc_bwd_filter_fft_planes', self.X[i].ref, self.dZ[i].ref, self._ptr(f['sf']), self._ptr(f['zf']), l.width,
             self._ptr(f['tables']), self._ptr(gf), 1, self._ptr(f['ws']), f['ws'].numel() * 4, s)
        call('st_conv1d_fft_bias_grad_dc_f32', self._ptr(f['dc']), f['rows'], l.cout, l.n_pad, self._ptr(gb), s)
        if on_layer_done is not None and wanted(i):
          if side:                       # (filter gradients of layers above still on the side streams: same bucket, see below)
            self._join_side_stream()
            side = False
          on_layer_done(i)
        relu_in = self.layers[i - 1].relu
        call('st_conv1d_nwc_bwd_data_fft_planes', self.dZ[i].ref, self._ptr(f['zf']), self._ptr(f['g']), l.width, f['pl'],
             self.X[i].ref if relu_in else None, self._ptr(self.Xb[i]) if relu_in else None, self.dZ[i - 1].ref, self._ptr(self.dZb[i - 1]),
             self._ptr(f['tables']), 1, self._ptr(f['ws']), f['ws'].numel() * 4, s)
        continue

      def filter_gradient(i=i, l=l, gf=gf, gb=gb, ws=(self.wgrad_ws_b3 if (i % 2 == 1 and self.wgrad_ws_b3 is not None)
                                                         else self.wgrad_ws_b2) if beside else self.wgrad_ws_b):
        call('st_conv1d_nwc_bwd_filter_tr_bf16' if self._wgrad_tr[i] else 'st_conv1d_nwc_bwd_filter_bf16', self.X[i].ref,
             self._ptr(self.Xb[i]), self.dZ[i].ref, self._ptr(self.dZb[i]), l.width, l.stride, self.geo[i][2], self._ptr(gf),
             self._ptr(gb), self._ptr(ws), ws.numel() * 4, self.stream_ptr)
      if beside:
        # two side streams take the chains in turn (each needs only its own layer's tensors): with all seven on one
        # stream that stream, not back-prop to the input, set the length of the backward pass of the narrow layers
        self._on_side_stream(filter_gradient, second=(i % 2 == 1 and self.wgrad_ws_b3 is not None))
        side = True
      else:
        filter_gradient()
        if on_layer_done is not None and wanted(i):
          if side:
            # the bucket this layer completes also holds layers whose filter gradients are still in flight on the side
            # streams (bottom bucket L0..L3: L1-L3 run beside back-prop, L0 does not); the exchange is ordered behind the
            # compute stream only
            self._join_side_stream()
            side = False
          on_layer_done(i)
      if i > 0:
        relu_in = self.layers[i - 1].relu
        call('st_conv1d_nwc_bwd_data_bf16', self.dZ[i].ref, self._ptr(self.dZb[i]), self._ptr(self.WTb[i]), l.width,
             self.geo[i][2], self.X[i].ref if relu_in else None, self._ptr(self.Xb[i]) if relu_in else None,
             self.dZ[i - 1].ref, self._ptr(self.dZb[i - 1]), self._ptr(self.wgrad_ws_b), self.wgrad_ws_b.numel() * 4, s)
      if beside and on_layer_done is not None and wanted(i):
        self._join_side_stream()
        side = False
        on_layer_done(i)
    if side:
      self._join_side_stream()
