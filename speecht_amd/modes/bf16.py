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
    self.e._fftb_layers = {i for i in range(len(self.e.layers)) if self._use_fft_bf16(i, batch, self.e.geo[i][1])}
    self._alloc_bf16()
    self._alloc_fft_bf16(batch)

  def forward(self):
    return self._forward_bf16()

  def backward(self, on_layer_done, wanted):
    self.e._join_side_stream()
    return self._backward_bf16(on_layer_done, wanted)

  def refresh_under_ctc(self):
    if not self.e._wtplanes_fresh and hasattr(self.e, 'WTb'):
      self.e._on_side_stream(lambda: self._refresh_bf16_filters(True))

  def refresh_after_update(self):
    if hasattr(self.e, 'Wb'):
      self._refresh_wb_after_update()              # the bf16 copies the next forward pass reads

  def prepare_forward_graph(self):
    if not self.e._wplanes_fresh:
      self._refresh_bf16_filters(False)            # derived operands are rebuilt outside the graph
    self.e._join_side_stream()
    if getattr(self.e, '_wb_ready', None):
      self.e._wb_ready.clear()                       # (covered by the join above)

  def _use_fft_bf16(self, i, batch, t_out):
    """bf16 activations (configs[3]): the 32-tap 250 -> 2000 layer runs as block DFTs + per-bin products on the bf16 matrix pipe
    (st_conv1d_*_fft_planes, one bf16 plane): 51.5 GFLOP per pass instead of the W-tap kernel's 513.  Only the wide
    long-filter layer: the narrow layers' W-tap bf16 kernels are launch-bound (~30 us), nothing to gain there."""
    l = self.e.layers[i]
    return (self.e.conv_mode == 'bf16' and self.e.fft_conv and os.environ.get('ST_FFT_BF16', '1') != '0' and i > 0 and
            l.stride == 1 and 16 <= l.width <= 33 and l.n_pad % 128 == 0 and batch * t_out >= self.e.fft_min_rows)

  def _alloc_fft_bf16(self, batch):
    lib = _lib.load()
    self.e.fftb = {}
    for i in sorted(self.e._fftb_layers):
      l = self.e.layers[i]
      t_in, t_out, pl, pr = self.e.geo[i]
      view = lambda name, numel, dtype=None: self.e._storage.view('fftb%d_%s' % (i, name), numel, dtype)
      bf = torch.bfloat16
      tables, fresh_tables = view('tables', lib.st_conv1d_fft_table_floats())
      if getattr(self.e, '_fftb_table_key', {}).get(i) != (l.width, pl):
        fresh_tables = True
      if not hasattr(self.e, '_fftb_table_key'):
        self.e._fftb_table_key = {}
      self.e._fftb_table_key[i] = (l.width, pl)
      ge = lib.st_conv1d_fft_filter_plane_elems(l.width, l.cin_pitch, l.cout)
      g, fresh_g = view('g', ge, bf)
      rows_pad, blocks = ctypes.c_int(), ctypes.c_int()
      call('st_conv1d_fft_plan', l.width, t_out, batch, None, None, ctypes.byref(blocks), None, ctypes.byref(rows_pad))
      f = dict(tables=tables, g=g, gt=view('gt', ge, bf)[0],
               sf=view('sf', lib.st_conv1d_fft_sf_floats(self.e.X[i].ref, self.e.X[i + 1].ref, l.width), bf)[0],
               zf=view('zf', lib.st_conv1d_fft_zf_floats(self.e.dZ[i].ref, l.width), bf)[0],
               dc=view('dc', rows_pad.value * l.n_pad)[0], rows=batch * blocks.value,
               ws=view('ws', lib.st_conv1d_fft_planes_ws(self.e.X[i].ref, self.e.X[i + 1].ref, l.width, 1) // 4 + 64)[0], pl=pl)
      if fresh_tables:
        call('st_conv1d_fft_tables_f32', l.width, pl, self.e._ptr(tables), tables.numel(), self.e.stream_ptr)
      if fresh_g:
        self.e._wplanes_fresh = False
      self.e.fftb[i] = f
    self._fftb_transition(None)

  # ---- shapes seen before (Wav2LetterEngine._reenter_shape) --------------------------------------------------------------
  shape_attrs = ('_fftb_layers', 'fftb', 'Xb', 'dZb', '_wgrad_tr', 'wgrad_ws_b', '_side_wgrad_bf16', 'wgrad_ws_b2', 'wgrad_ws_b3')

  def shape_token(self):
    return {i: (self.e.layers[i].width, f['pl']) for i, f in self.e.fftb.items()}

  def reenter(self, token):
    self._fftb_transition(token)

  def _fftb_transition(self, table_keys):
    """What entering a shape does that depends on the shape left behind (see SpectralLayers._fft_transition)."""
    for i, key in (table_keys or {}).items():
      if self.e._fftb_table_key.get(i) != key:
        f = self.e.fftb[i]
        call('st_conv1d_fft_tables_f32', key[0], key[1], self.e._ptr(f['tables']), f['tables'].numel(), self.e.stream_ptr)
        self.e._fftb_table_key[i] = key
    if set(self.e.fftb) != getattr(self.e, '_fftb_prev', None):
      self.e._wplanes_fresh = False
      self.e._wtplanes_fresh = False
    self.e._fftb_prev = set(self.e.fftb)

  def _alloc_bf16(self):
    L = len(self.e.layers)
    lib = _lib.load()
    # the filter gradients of the stride-1 layers read both planes as they lie (LDS transpose reads, csrc/wgrad_tr_bf16.hip) and
    # run up to `slack` rows past the last one: zeros behind every plane
    slack = lib.st_conv1d_bwd_filter_tr_bf16_slack_rows()
    self.e.Xb = [self.e._planes('Xb%d' % i, self.e.X[i].buf.numel(), 1, slack * self.e.X[i].c_pitch) for i in range(L)]
    self.e.dZb = [self.e._planes('dZb%d' % i, self.e.dZ[i].buf.numel(), 1, slack * self.e.dZ[i].c_pitch) for i in range(L)]
    self.e._wgrad_tr = [os.environ.get('ST_BF16_WGRAD_TR', '1') != '0' and
                      lib.st_conv1d_bwd_filter_tr_bf16_ws(self.e.X[i].ref, self.e.dZ[i].ref, l.width, l.stride, self.e.geo[i][2]) > 0
                      for i, l in enumerate(self.e.layers)]
    wgrad_ws = lambda i: (lib.st_conv1d_bwd_filter_tr_bf16_ws if self.e._wgrad_tr[i] else lib.st_conv1d_bwd_filter_bf16_ws)(
        self.e.X[i].ref, self.e.dZ[i].ref, self.e.layers[i].width, self.e.layers[i].stride, self.e.geo[i][2])
    ws = max(wgrad_ws(i) for i in range(L))
    ws = max([ws] + [lib.st_conv1d_bwd_data_bf16_ws(self.e.dZ[i].ref, self.e.dZ[i - 1].ref, l.width)
                     for i, l in enumerate(self.e.layers) if i > 0])
    ws = max([ws] + [lib.st_conv1d_fwd_bf16_ws(self.e.X[i].ref, self.e.X[i + 1].ref, l.width)
                     for i, l in enumerate(self.e.layers)])
    self.e.wgrad_ws_b, _ = self.e._storage.view('wgrad_ws_b', ws // 4 + 64)
    # the narrow layers' filter gradients run beside back-prop to the input on the side stream: their own scratch
    # (the classification layer beside its back-prop, as in fp32: measured, no gain here -- 3.15 ms either way)
    self.e._side_wgrad_bf16 = [i for i, l in enumerate(self.e.layers) if self.e.side_filter_gradient and i > 0 and l.cout <= 512 and l.cin <= 512]
    ws2 = max([0] + [wgrad_ws(i) for i in self.e._side_wgrad_bf16])
    self.e.wgrad_ws_b2 = self.e._storage.view('wgrad_ws_b2', ws2 // 4 + 64)[0] if ws2 else None
    self.e.wgrad_ws_b3 = self.e._storage.view('wgrad_ws_b3', ws2 // 4 + 64)[0] if ws2 else None    # second side stream
    if not hasattr(self.e, 'Wb'):
      z = lambda n: torch.zeros(n, dtype=torch.bfloat16, device=self.e.device)
      self.e.Wb = [z(l.k_pad * l.n_pad) for l in self.e.layers]
      self.e.WTb = [None] + [z(l.kt_pad * l.nt_pad) for l in self.e.layers[1:]]

  def _refresh_bf16_filters(self, transposed, layers=None):
    fftb = getattr(self.e, 'fftb', {})
    for i, l in enumerate(self.e.layers):
      if layers is not None and i not in layers:
        continue
      if i in fftb:
        # a frequency-domain layer: its filter spectra (one bf16 plane, both operand layouts) instead of the two bf16 copies
        if not transposed:
          f = fftb[i]
          call('st_conv1d_fft_filters_planes', self.e._ptr(self.e._slice(self.e.params, i)[0]), l.width, l.cin, l.cout, l.cin_pitch,
               self.e._ptr(f['tables']), self.e._ptr(f['g']), self.e._ptr(f['gt']), 1, self.e.stream_ptr)
        continue
      if transposed and i > 0:
        call('st_filters_bwd_bf16', self.e._ptr(self.e._slice(self.e.params, i)[0]), l.width, l.cin, l.cout, l.cin_pitch,
             l.cout_pitch, self.e._ptr(self.e.WTb[i]), self.e.stream_ptr)
      elif not transposed:
        call('st_filters_bf16', self.e._ptr(self.e._slice(self.e.params, i)[0]), l.k_pad, l.n_pad, self.e._ptr(self.e.Wb[i]),
             self.e.stream_ptr)
    if layers is not None:
      return
    if transposed:
      self.e._wtplanes_fresh = True
    else:
      self.e._wplanes_fresh = True

  def _refresh_wb_after_update(self):
    """After an update: the bottom layer's bf16 filter copy on the compute stream (the next forward pass needs it at
    once), the others on the side stream -- the small copies bottom layer first, the frequency-domain layers' filter spectra
    (L9: a 16 M-weight transform and its second operand layout, 0.16 ms) LAST, an event per layer: the forward pass waits for
    what a layer reads, not for the whole list, and the spectra are built beside the eight layers below them."""
    L = len(self.e.layers)
    self.e._wb_ready = {}
    self._refresh_bf16_filters(False, layers=[0])
    fftb = getattr(self.e, 'fftb', {})
    order = [i for i in range(1, L) if i not in fftb] + [i for i in range(1, L) if i in fftb]

    def rest():
      for i in order:
        self._refresh_bf16_filters(False, layers=[i])
        ev = torch.cuda.Event()
        ev.record(self.e._stream)
        self.e._wb_ready[i] = ev
    self.e._on_side_stream(rest)
    self.e._wb_order = order
    self.e._wplanes_fresh = True

  def _forward_bf16(self):
    s, L = self.e.stream_ptr, len(self.e.layers)
    main = self.e._stream if self.e._stream is not None else torch.cuda.current_stream(self.e.device)
    ready = getattr(self.e, '_wb_ready', None) or {}
    if not self.e._wplanes_fresh:
      self.e._join_side_stream()                     # (a rebuild still running there writes the same buffers)
      ready.clear()
      self._refresh_bf16_filters(False)
    call('st_cast_bf16', self.e._ptr(self.e.X[0].buf), self.e.X[0].buf.numel(), self.e._ptr(self.e.Xb[0]), s)
    for i, l in enumerate(self.e.layers):
      last = i + 1 == L
      if i in ready:
        # the side stream works in self.e._wb_order: the first layers wait for their own copy, the fourth for every small copy
        # (by then they are through; every wait costs the compute stream a few microseconds), a frequency-domain layer for its
        # own spectra
        if i in self.e.fftb or i < 3:
          main.wait_event(ready.pop(i))
        else:
          small = [j for j in self.e._wb_order if j not in self.e.fftb]
          main.wait_event(ready[small[-1]])
          for j in small:
            ready.pop(j, None)
      if i in self.e.fftb and not last:
        f = self.e.fftb[i]
        call('st_conv1d_nwc_fwd_fft_planes', self.e.X[i].ref, self.e._ptr(self.e.Xb[i]), self.e._ptr(f['gt']), self.e._ptr(self.e._slice(self.e.params, i)[1]),
             l.width, f['pl'], int(l.relu), self.e.X[i + 1].ref, self.e._ptr(self.e.Xb[i + 1]), self.e._ptr(f['tables']), self.e._ptr(f['sf']), 1,
             self.e._ptr(f['ws']), f['ws'].numel() * 4, s)
        continue
      call('st_conv1d_nwc_fwd_ws_bf16', self.e.X[i].ref, self.e._ptr(self.e.Xb[i]), self.e._ptr(self.e.Wb[i]),
           self.e._ptr(self.e._slice(self.e.params, i)[1]), l.width, l.stride, self.e.geo[i][2], int(l.relu), self.e.X[i + 1].ref,
           None if last else self.e._ptr(self.e.Xb[i + 1]), self.e._ptr(self.e.X[i + 1].buf) if last else None,
           self.e._ptr(self.e.wgrad_ws_b), self.e.wgrad_ws_b.numel() * 4 if self.e.split_small_batches else 0, s)

  def _backward_bf16(self, on_layer_done, wanted=lambda i: True):
    s, L = self.e.stream_ptr, len(self.e.layers)
    if not self.e._wtplanes_fresh:
      self._refresh_bf16_filters(True)
    call('st_cast_bf16', self.e._ptr(self.e.dZ[L - 1].buf), self.e.dZ[L - 1].buf.numel(), self.e._ptr(self.e.dZb[L - 1]), s)
    side = False
    for i in reversed(range(L)):
      l = self.e.layers[i]
      gf, gb = self.e._slice(self.e.grads, i)
      beside = i in self.e._side_wgrad_bf16      # this layer's filter gradient runs beside its back-prop to the input

      if i in self.e.fftb:
        # frequency-domain layer: ONE transform of dz (bf16 spectra + the fp32 block sums) serves the filter gradient, the bias
        # gradient and back-prop to the input
        f = self.e.fftb[i]
        call('st_conv1d_fft_dz_spectra_planes', self.e.dZ[i].ref, self.e._ptr(self.e.dZb[i]), l.width, self.e._ptr(f['tables']), self.e._ptr(f['zf']), 1,
             self.e._ptr(f['dc']), s)
        call('st_conv1d_nwc_bwd_filter_fft_planes', self.e.X[i].ref, self.e.dZ[i].ref, self.e._ptr(f['sf']), self.e._ptr(f['zf']), l.width,
             self.e._ptr(f['tables']), self.e._ptr(gf), 1, self.e._ptr(f['ws']), f['ws'].numel() * 4, s)
        call('st_conv1d_fft_bias_grad_dc_f32', self.e._ptr(f['dc']), f['rows'], l.cout, l.n_pad, self.e._ptr(gb), s)
        if on_layer_done is not None and wanted(i):
          if side:                       # (filter gradients of layers above still on the side streams: same bucket, see below)
            self.e._join_side_stream()
            side = False
          on_layer_done(i)
        relu_in = self.e.layers[i - 1].relu
        call('st_conv1d_nwc_bwd_data_fft_planes', self.e.dZ[i].ref, self.e._ptr(f['zf']), self.e._ptr(f['g']), l.width, f['pl'],
             self.e.X[i].ref if relu_in else None, self.e._ptr(self.e.Xb[i]) if relu_in else None, self.e.dZ[i - 1].ref, self.e._ptr(self.e.dZb[i - 1]),
             self.e._ptr(f['tables']), 1, self.e._ptr(f['ws']), f['ws'].numel() * 4, s)
        continue

      def filter_gradient(i=i, l=l, gf=gf, gb=gb, ws=(self.e.wgrad_ws_b3 if (i % 2 == 1 and self.e.wgrad_ws_b3 is not None)
                                                         else self.e.wgrad_ws_b2) if beside else self.e.wgrad_ws_b):
        call('st_conv1d_nwc_bwd_filter_tr_bf16' if self.e._wgrad_tr[i] else 'st_conv1d_nwc_bwd_filter_bf16', self.e.X[i].ref,
             self.e._ptr(self.e.Xb[i]), self.e.dZ[i].ref, self.e._ptr(self.e.dZb[i]), l.width, l.stride, self.e.geo[i][2], self.e._ptr(gf),
             self.e._ptr(gb), self.e._ptr(ws), ws.numel() * 4, self.e.stream_ptr)
      if beside:
        # two side streams take the chains in turn (each needs only its own layer's tensors): with all seven on one
        # stream that stream, not back-prop to the input, set the length of the backward pass of the narrow layers
        self.e._on_side_stream(filter_gradient, second=(i % 2 == 1 and self.e.wgrad_ws_b3 is not None))
        side = True
      else:
        filter_gradient()
        if on_layer_done is not None and wanted(i):
          if side:
            # the bucket this layer completes also holds layers whose filter gradients are still in flight on the side
            # streams (bottom bucket L0..L3: L1-L3 run beside back-prop, L0 does not); the exchange is ordered behind the
            # compute stream only
            self.e._join_side_stream()
            side = False
          on_layer_done(i)
      if i > 0:
        relu_in = self.e.layers[i - 1].relu
        call('st_conv1d_nwc_bwd_data_bf16', self.e.dZ[i].ref, self.e._ptr(self.e.dZb[i]), self.e._ptr(self.e.WTb[i]), l.width,
             self.e.geo[i][2], self.e.X[i].ref if relu_in else None, self.e._ptr(self.e.Xb[i]) if relu_in else None,
             self.e.dZ[i - 1].ref, self.e._ptr(self.e.dZb[i - 1]), self.e._ptr(self.e.wgrad_ws_b), self.e.wgrad_ws_b.numel() * 4, s)
      if beside and on_layer_done is not None and wanted(i):
        self.e._join_side_stream()
        side = False
        on_layer_done(i)
    if side:
      self.e._join_side_stream()
