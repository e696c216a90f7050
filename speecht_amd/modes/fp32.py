"""fp32 mode (BASELINE configs[1], the headline): exact-fp32 MFMA kernels (csrc/conv_gemm.hip) with nine of the eleven
layers in the frequency domain (csrc/conv_fft.hip, `SpectralLayers`).  The bf16x6 mode derives from this class and
overrides the per-layer hooks `_x6_*`."""
import ctypes
import os

import torch

from .. import _lib
from .._lib import call
from .base import ModeBase
from .spectral import SpectralLayers


class Fp32Mode(SpectralLayers, ModeBase):
  _one_tap_in_place = True        # 1-tap layers back-propagate through their own packed filters read transposed
  _flip_every_layer = False       # (bf16x6: every layer's operand planes are split from a flipped / transposed copy)

  # ---- buffers ---------------------------------------------------------------------------------------------------
  def alloc(self, batch):
    lib = _lib.load()
    ws = self._workspace_bytes(lib)
    self.e.wgrad_ws, _ = self.e._storage.view('wgrad_ws', ws // 4 + 64)
    # The classification layer (2000 -> 29): its filter gradient streams the activations, its back-prop to the input streams
    # the mask and writes dZ of the layer below -- two HBM-bound launches of ~90 us each that do not depend on each other.
    # Side by side on two streams (own scratch for the one on the side stream).
    top = len(self.e.layers) - 1
    self.e._side_wgrad_top = (self._top_gradient_beside and self.e.side_filter_gradient and os.environ.get('ST_WGRAD_SIDE_TOP', '1') != '0' and
                            top > 0 and self.e.layers[top].cout <= 64 and self.e.layers[top].width == 1)
    if self.e._side_wgrad_top:
      ws_top = lib.st_conv1d_bwd_filter_ws(self.e.X[top].ref, self.e.dZ[top].ref, self.e.layers[top].width)
      self.e.wgrad_ws_top, _ = self.e._storage.view('wgrad_ws_top', ws_top // 4 + 64)
    # which layers run in the frequency domain is decided first: they leave the bf16x6 plane plumbing alone
    self.e._fft_layers = {i for i in range(len(self.e.layers)) if self._use_fft(i, batch, self.e.geo[i][1])}
    self._alloc_mode_planes()
    self._alloc_fft(batch)

  _top_gradient_beside = True

  # ---- shapes seen before (Wav2LetterEngine._reenter_shape) --------------------------------------------------------------
  shape_attrs = ('wgrad_ws', '_side_wgrad_top', 'wgrad_ws_top', '_fft_layers', 'fft')

  def shape_token(self):
    return {i: (f['width'], f['pl']) for i, f in self.e.fft.items()}

  def reenter(self, token):
    self._fft_transition(token)

  def _workspace_bytes(self, lib):
    ws = max(lib.st_conv1d_bwd_filter_ws(self.e.X[i].ref, self.e.dZ[i].ref, l.width) for i, l in enumerate(self.e.layers))
    ws = max([ws] + [lib.st_conv1d_bwd_data_bias_ws(self.e.dZ[i].ref, self.e.dZ[i - 1].ref, l.width)
                     for i, l in enumerate(self.e.layers) if i > 0])
    return max([ws] + [lib.st_conv1d_fwd_ws(self.e.X[i].ref, self.e.X[i + 1].ref, l.width) for i, l in enumerate(self.e.layers)])

  def _alloc_mode_planes(self):
    pass

  # ---- hooks of the bf16x6 mode (nothing here) -----------------------------------------------------------------------
  def _x6_fwd(self, i):
    return False

  def _x6_bwd(self, i):
    return False

  def _x6_wgrad(self, i):
    return False

  def _forward_prologue(self):
    pass

  def _backward_prologue(self):
    pass

  def _x6_forward_layer(self, i, pb):              # (reached only where `_x6_fwd(i)` holds: Bf16x6Mode)
    raise NotImplementedError

  def _x6_filter_gradient(self, i, gf, gb, need_bias):
    raise NotImplementedError

  def _x6_back_prop(self, i):
    raise NotImplementedError

  # ---- refreshes -----------------------------------------------------------------------------------------------------
  def refresh_under_ctc(self):
    # the filter operands of back-prop (flipped / transposed copies of the weights Adam just updated) are rebuilt
    # on the side stream while the CTC recursion runs
    if not self._packed_t_ok() and self._flip_layers():
      self.e._on_side_stream(self._refresh_backward_operands)

  def refresh_after_update(self):
    self._refresh_gfwd()          # the forward filter spectra of the frequency-domain layers for the next pass

  def prepare_forward_graph(self):
    self.e._join_side_stream()
    if self.e.fft and self.e.fft_conv and not self.e._gfwd_fresh:
      self._refresh_fft_filters()
    self._wait_gfwd()                              # no waits on outside events inside a capture

  def _transposed_in_place(self, i):
    """Back-prop to the input of layer i reads the layer's own packed filters as a transposed operand
    (st_conv1d_1tap_bwd_data_bias_f32): one tap, whole 32-deep k-tiles over the output channels."""
    l = self.e.layers[i]
    return (self._one_tap_in_place and i > 0 and l.width == 1 and l.stride == 1 and l.cout_pitch % 32 == 0 and
            l.n_pad >= l.cout_pitch and l.nt_pad % 128 == 0)

  def _flip_layers(self):
    """Layers whose back-prop to the input still needs the flipped / transposed copy of the weights: W-tap layers of
    more than one tap (and everything on the bf16x6 path, whose operand planes are split from those copies).  The
    frequency-domain layers read their forward spectra transposed, 1-tap layers their packed filters (round 4): at the
    model's training shapes NO copy is rebuilt any more (rounds 1-3: ~0.33 ms of HBM-bound launches per step)."""
    return [i for i in range(1, len(self.e.layers))
            if self._flip_every_layer or not ((i in self.e.fft and self.e.fft_conv) or self._transposed_in_place(i))]

  def _refresh_backward_operands(self):
    """The flipped / transposed weight copies of the layers that still need one (`_flip_layers`), top layer first, the order
    back-prop consumes them in; an event after each lets the compute stream wait for what it is about to use only."""
    s = self.e.stream_ptr
    stream = self.e._stream if self.e._stream is not None else torch.cuda.current_stream(self.e.device)
    self.e._bwd_ready = {}
    for i in reversed(self._flip_layers()):
      l = self.e.layers[i]
      call('st_filters_flip_transpose_f32', self.e._ptr(self.e._slice(self.e.params, i)[0]), l.width, l.cin, l.cout, l.cin_pitch,
           l.cout_pitch, self.e._ptr(self.e.packed_t[i]), s)
      ev = torch.cuda.Event()
      ev.record(stream)
      self.e._bwd_ready[i] = ev
    self.e._packed_t_fresh = True
    self.e._packed_t_layers = frozenset(self._flip_layers())
    self.e._wtplanes_fresh = False

  def _packed_t_ok(self):
    """The flipped / transposed copies are current for every layer that needs one NOW: which layers do depends on state that
    can change between steps (the shape's frequency-domain set, `fft_conv`), so a refresh remembers the set it rebuilt."""
    return self.e._packed_t_fresh and frozenset(self._flip_layers()) <= getattr(self.e, '_packed_t_layers', frozenset())

  def _wait_bwd_operands(self, i=None):
    """The compute stream waits for the back-prop operands of layer i (None: of every layer) if they were rebuilt on the
    side stream.  The side stream works top layer first, so a lower layer's event covers the ones above it."""
    ready = getattr(self.e, '_bwd_ready', None)
    if not ready:
      return
    keys = [k for k in ready if i is None or k >= i]
    if keys:
      (self.e._stream if self.e._stream is not None else torch.cuda.current_stream(self.e.device)).wait_event(ready[min(keys)])
      for k in keys:
        del ready[k]

  def forward(self):
    """X[0] -> logits X[-1] through the eleven layers (speech_model.py:279-295): per layer the frequency-domain entry
    point, the W-tap kernel or -- in the bf16x6 mode -- that mode's kernel, as decided per shape by `_use_fft` / `_x6_fwd`."""
    s = self.e.stream_ptr
    self._forward_prologue()
    sf_ready = False                 # the previous layer's call left this layer's input spectra behind
    for i, l in enumerate(self.e.layers):
      pf, pb = self.e._slice(self.e.params, i)
      if not (i in self.e.fft and self.e.fft_conv):
        sf_ready = False
      if self._x6_fwd(i):
        self._x6_forward_layer(i, pb)
      elif i in self.e.fft and self.e.fft_conv:
        f = self.e.fft[i]
        if not self.e._gfwd_fresh:
          self.e._join_side_stream()
          self._refresh_fft_filters()
        self._wait_gfwd(i)                               # the filter spectra may still be on their way (side stream)
        # a chain of frequency-domain layers: where the shapes allow, this layer's inverse transform hands its frames to the
        # next layer's forward transform in registers and leaves that layer's input spectra behind (`sf_ready` for its call)
        nxt = self.e.fft.get(i + 1) if self.e.fft_conv else None
        if nxt is not None and nxt['shift'] is not None:
          nxt = None
        written = ctypes.c_int(0)
        call('st_conv1d_nwc_fwd_fft_chain_f32', f['xref'], self.e._ptr(f['gfwd']), self.e._ptr(pb), f['width'], f['pl'],
             int(l.relu), self.e.X[i + 1].ref, self.e._ptr(f['tables']), self.e._ptr(f['sf']), int(sf_ready),
             self.e._ptr(nxt['tables']) if nxt else None, self.e._ptr(nxt['sf']) if nxt else None, nxt['width'] if nxt else 0,
             nxt['pl'] if nxt else 0, ctypes.byref(written), self.e._ptr(f['ws']), f['ws'].numel() * 4, s)
        sf_ready = written.value == 1
        continue
      else:
        call('st_conv1d_nwc_fwd_ws_f32', self.e.X[i].ref, self.e._ptr(pf), self.e._ptr(pb), l.width, l.stride,
             self.e.geo[i][2], int(l.relu), self.e.X[i + 1].ref, self.e._ptr(self.e.wgrad_ws),
             self.e.wgrad_ws.numel() * 4 if self.e.split_small_batches else 0, s)

  def backward(self, on_layer_done, wanted):
    """Back-prop from dZ[-1] through the frequency-domain / W-tap (/ bf16x6) kernels; hooks as `Wav2LetterEngine.backward`
    describes them."""
    s = self.e.stream_ptr
    if not self._packed_t_ok():
      self._refresh_backward_operands()           # (normally done on the side stream by ctc_loss_grad; nothing at the model's shapes)
    if self.e.fft and self.e.fft_conv and not self.e._gfwd_fresh:
      self.e._join_side_stream()                    # (weights written after the forward pass: back-prop reads the same spectra)
      self._refresh_fft_filters()
    self._wait_gfwd()
    self._backward_prologue()
    # waits per layer; after the two layers on top one wait covers everything below (by then the side stream is through)
    wait_all_below = len(self.e.layers) - 3
    side_wgrad, deferred = False, None     # a filter gradient is in flight on the side stream; its layer's hook is due
    top_pending = None                     # the classification layer's filter gradient is in flight on the second side stream
    hook = on_layer_done
    if hook is not None:
      def on_layer_done(j):
        nonlocal top_pending
        if top_pending is not None and top_pending != j:
          self.e._join_side_stream(second_only=True)
          hook(top_pending)
          top_pending = None
        if top_pending != j:
          hook(j)
    bias_from_above = False      # layer i's bias gradient already written by the back-prop kernel of layer i + 1
    zf_ready = False             # layer i's dz spectra already written by the back-prop kernel of layer i + 1
    for i in reversed(range(len(self.e.layers))):
      l = self.e.layers[i]
      gf, gb = self.e._slice(self.e.grads, i)
      need_bias, bias_from_above = not bias_from_above, False
      if self._x6_wgrad(i):
        self._x6_filter_gradient(i, gf, gb, need_bias)
      elif i in self.e.fft and self.e.fft_conv:
        f = self.e.fft[i]
        # the spectra of dz serve the filter gradient here and back-prop to the input below (the layer above may have left
        # them behind already: its back-prop kernel transformed the frames it had just produced, `zf_ready`)
        if not zf_ready:
          call('st_conv1d_fft_dz_spectra_f32', self.e.dZ[i].ref, f['width'], self.e._ptr(f['tables']), self.e._ptr(f['zf']), s)
        zf_ready = False
        polyphase = f['shift'] is not None       # the gradient comes out in the shifted layout of the polyphase taps

        def filter_gradient(f=f, l=l, i=i, gf=gf, gb=gb, need_bias=need_bias, polyphase=polyphase, ws=f.get('ws2', f['ws'])):
          call('st_conv1d_nwc_bwd_filter_fft_f32', f['xref'], self.e.dZ[i].ref, self.e._ptr(f['sf']), self.e._ptr(f['zf']), f['width'],
               self.e._ptr(f['tables']), self.e._ptr(f['dpacked2'] if polyphase else gf), self.e._ptr(ws), ws.numel() * 4, self.e.stream_ptr)
          if polyphase:
            cp = self.e.X[i].c_pitch
            n, o = l.width * cp * l.n_pad, f['shift'] * cp * l.n_pad
            with torch.cuda.stream(self.e._stream if self.e._stream is not None else torch.cuda.current_stream(self.e.device)):
              gf[:n].copy_(f['dpacked2'][o:o + n], non_blocking=True)
          if need_bias:      # bin 0 of the spectra is the sum over the frames
            call('st_conv1d_fft_bias_grad_f32', self.e.dZ[i].ref, f['width'], self.e._ptr(f['zf']), self.e._ptr(gb), self.e.stream_ptr)
        if 'ws2' in f:
          # The filter gradient (lag products, inverse transform of the filters, bias sum) and back-prop to the input
          # (products, inverse transform) both hang off the spectra of dz and are independent: on two streams.  The
          # narrow layers' products (36 bins x 16 tiles) leave a quarter of the CU slots of their last round empty --
          # side by side they fill each other's gaps -- and the HBM-bound transforms of one chain run under the
          # matrix-pipe-bound products of the other (measured: 7.84 -> 7.43 ms per step).
          # (round 4: with back-prop's transforms fused the compute stream needs ~80 us per narrow layer, ONE side stream's
          # chain -- 73 + 42 + 9 us, in order -- had become the pace of the backward pass: the chains take the two side streams in turn)
          self.e._on_side_stream(filter_gradient, second=(i % 2 == 1))
          side_wgrad, deferred = True, i
        else:
          filter_gradient()
      elif i + 1 == len(self.e.layers) and self.e._side_wgrad_top:
        def top_gradient(i=i, l=l, gf=gf, gb=gb, need_bias=need_bias, ws=self.e.wgrad_ws_top):
          call('st_conv1d_nwc_bwd_filter_f32', self.e.X[i].ref, self.e.dZ[i].ref, l.width, l.stride, self.e.geo[i][2],
               self.e._ptr(gf), self.e._ptr(gb) if need_bias else None, self.e._ptr(ws), ws.numel() * 4, self.e.stream_ptr)
        # on the SECOND side stream (the first is still rebuilding back-prop operands when CTC ends); its hook is due with
        # the next layer's -- the two share a reduce bucket, and nothing waits for this launch until then
        self.e._on_side_stream(top_gradient, second=True)
        top_pending = i
      else:
        call('st_conv1d_nwc_bwd_filter_f32', self.e.X[i].ref, self.e.dZ[i].ref, l.width, l.stride, self.e.geo[i][2],
             self.e._ptr(gf), self.e._ptr(gb) if need_bias else None, self.e._ptr(self.e.wgrad_ws), self.e.wgrad_ws.numel() * 4, s)
      if on_layer_done is not None and deferred != i and wanted(i):
        if side_wgrad:
          # this layer's own gradient ran on the compute stream, but its bucket also holds the layers above whose filter
          # gradients are still in flight on the side streams (the bottom bucket L0..L3: L1-L3 went there, L0 did not): the
          # exchange is ordered behind the compute stream only, so the compute stream waits for them first
          self.e._join_side_stream()
          side_wgrad = False
        on_layer_done(i)
      if i > 0 and self._x6_bwd(i):
        self._x6_back_prop(i)
      elif i > 0 and i in self.e.fft and self.e.fft_conv:
        f = self.e.fft[i]
        act = self.e.X[i].ref if self.e.layers[i - 1].relu else None
        below = self.e.fft.get(i - 1)                # a frequency-domain layer below: its dz spectra can ride along
        written = ctypes.c_int(0)
        call('st_conv1d_nwc_bwd_data_fft_chain_f32', self.e.dZ[i].ref, self.e._ptr(f['zf']), self.e._ptr(f['gfwd']), l.width,
             self.e.geo[i][2], act, self.e.dZ[i - 1].ref, self.e._ptr(f['tables']), self.e._ptr(below['tables']) if below else None,
             self.e._ptr(below['zf']) if below else None, below['width'] if below else 0, ctypes.byref(written), self.e._ptr(f['ws']),
             f['ws'].numel() * 4, s)
        zf_ready = written.value == 1
      elif i > 0:
        # X[i] is the ReLU output of layer i-1: its sign is the mask of tf.nn.relu's gradient
        # the kernel that writes dZ[i-1] also sums its columns: the bias gradient of layer i - 1
        act = self.e.X[i].ref if self.e.layers[i - 1].relu else None
        if self._transposed_in_place(i):           # dx = dz W^T straight from the layer's packed filters
          call('st_conv1d_1tap_bwd_data_bias_f32', self.e.dZ[i].ref, self.e._ptr(self.e._slice(self.e.params, i)[0]), act, self.e.dZ[i - 1].ref,
               self.e._ptr(self.e._slice(self.e.grads, i - 1)[1]), self.e._ptr(self.e.wgrad_ws), self.e.wgrad_ws.numel() * 4, s)
        else:
          self._wait_bwd_operands(i if i > wait_all_below else None)
          call('st_conv1d_nwc_bwd_data_bias_f32', self.e.dZ[i].ref, self.e._ptr(self.e.packed_t[i]), l.width, self.e.geo[i][2],
               act, self.e.dZ[i - 1].ref, self.e._ptr(self.e._slice(self.e.grads, i - 1)[1]), self.e._ptr(self.e.wgrad_ws),
               self.e.wgrad_ws.numel() * 4, s)
        bias_from_above = True
      if deferred == i and on_layer_done is not None and wanted(i):
        # the gradient of this layer is complete when the side stream is (and with it those of the layers above that went
        # the same way: the stream runs in order): hand the bucket to the all-reduce only now, with back-prop to the input
        # already enqueued beside it
        self.e._join_side_stream()
        side_wgrad = False
        on_layer_done(i)
      if deferred == i:
        deferred = None
    if side_wgrad or top_pending is not None:
      self.e._join_side_stream()
    if top_pending is not None and hook is not None:
      hook(top_pending)
