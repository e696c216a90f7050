"""Frequency-domain layers (csrc/conv_fft.hip) of the fp32 and bf16x6 modes: which layers take the path for a shape, their
tables / filter spectra / scratch, and the refresh of the filter spectra after an update."""
import ctypes
import os

import torch

from .. import _lib
from .._lib import Tensor3, call


class SpectralLayers:
  """Mixin of `Fp32Mode`: `fft[i]` describes layer i on the frequency path (tables, the ONE set of filter spectra `gfwd`,
  input / gradient spectra, scratch per stream)."""

  def _polyphase(self, i):
    """A stride-2 layer as a stride-1 layer on the polyphase view of its input: x read as [B][T/2][2 * c_pitch] (frame
    pairs as channels), y[t] = sum_w F[w] x[2t + w - pl] = sum_{j,p} F[2j + p - shift] X2[t + j - pl2][p] with
    pl2 = ceil(pl / 2), shift = 2 pl2 - pl: width2 = ceil((W + shift) / 2) taps whose packed filters are the layer's
    own rows moved down by `shift` channel blocks (zeros around them).  Returns (width2, pl2, shift) or None."""
    l = self.e.layers[i]
    if l.stride != 2:
      return None
    pl = self.e.geo[i][2]
    pl2 = (pl + 1) // 2
    shift = 2 * pl2 - pl
    return (l.width + shift + 1) // 2, pl2, shift

  def _use_fft(self, i, batch, t_out):
    l = self.e.layers[i]
    wide = l.stride == 1 and l.width >= 16
    if not (self.e.fft_conv and self.e.conv_mode in ('fp32', 'bf16x6') and l.n_pad % 128 == 0 and
            batch * t_out >= (self.e.fft_min_rows if wide else self.e.fft_min_rows_narrow)):
      return False
    if l.stride == 2:        # first layer of the model (48 taps, stride 2): 25 polyphase taps over 2 x 80 channels
      width2 = self._polyphase(i)[0]
      return i == 0 and self.e.fft_first_layer and self.e.fft_min_width <= width2 <= 33
    return i > 0 and l.stride == 1 and self.e.fft_min_width <= l.width <= 33 and l.nt_pad % 128 == 0

  def _alloc_fft(self, batch):
    """Per frequency-domain layer: the transform tables and the filter spectra in both operand layouts (functions of
    the layer only: kept across shapes), the input / gradient spectra and one scratch area (sized by the shape)."""
    lib = _lib.load()
    self.e.fft = {}
    for i, l in enumerate(self.e.layers):
      t_in, t_out, pl, pr = self.e.geo[i]
      if i not in self.e._fft_layers:
        continue
      view = lambda name, numel: self.e._storage.view('fft%d_%s' % (i, name), numel)
      f = dict(x=self.e.X[i].desc, width=l.width, pl=pl, cin=l.cin, cin_pitch=l.cin_pitch, shift=None)
      if l.stride == 2:
        width2, pl2, shift = self._polyphase(i)
        x = self.e.X[i]
        assert x.halo % 2 == 0 and x.t_pitch % 2 == 0
        cp2 = 2 * x.c_pitch
        f.update(x=Tensor3(x.buf.data_ptr(), batch, t_out, cp2, x.halo // 2, x.t_pitch // 2, cp2), width=width2, pl=pl2,
                 cin=cp2, cin_pitch=cp2, shift=shift)
        # the layer's packed filters between zero blocks, and the gradient in the same layout
        rows = 2 * width2 * x.c_pitch * l.n_pad
        f['packed2'], fresh_p = view('packed2', rows)
        f['dpacked2'] = view('dpacked2', rows)[0]
        if fresh_p:
          f['packed2'].zero_()
          self.e._gfwd_fresh = False
      f['xref'] = ctypes.byref(f['x'])
      tables, fresh_tables = view('tables', lib.st_conv1d_fft_table_floats())
      # the tables are functions of (taps, left padding): a new shape or another model may change either for the same
      # layer index, so the pair is kept with them
      if getattr(self.e, '_fft_table_key', {}).get(i) != (f['width'], f['pl']):
        fresh_tables = True
      if not hasattr(self.e, '_fft_table_key'):
        self.e._fft_table_key = {}
      self.e._fft_table_key[i] = (f['width'], f['pl'])
      # ONE set of filter spectra: back-prop to the input reads it as a transposed operand (csrc/conv_fft.hip)
      gfwd, fresh_f = view('gfwd', lib.st_conv1d_fft_filter_floats(f['width'], f['cin_pitch'], l.cout))
      f.update(tables=tables, gfwd=gfwd,
               sf=view('sf', lib.st_conv1d_fft_sf_floats(f['xref'], self.e.X[i + 1].ref, f['width']))[0],
               zf=view('zf', lib.st_conv1d_fft_zf_floats(self.e.dZ[i].ref, f['width']))[0],
               ws=view('ws', lib.st_conv1d_fft_ws(f['xref'], self.e.X[i + 1].ref, f['width']) // 4 + 64)[0])
      # (the wide 32-tap layer stays on one stream: its chain side by side, or only its HBM-bound inverse transform of the
      # lag products beside back-prop's products, both measured slower: 7.37 -> 7.43 ms)
      if self.e.side_filter_gradient and i > 0 and l.cout <= 512:
        f['ws2'] = view('ws2', lib.st_conv1d_fft_ws(f['xref'], self.e.X[i + 1].ref, f['width']) // 4 + 64)[0]
      if fresh_tables:
        call('st_conv1d_fft_tables_f32', f['width'], f['pl'], self.e._ptr(tables), tables.numel(), self.e.stream_ptr)
      if fresh_f:
        self.e._gfwd_fresh = False
      self.e.fft[i] = f
    self._fft_transition(None)

  def _fft_transition(self, table_keys):
    """What entering a shape does that depends on the shape left behind.  ``table_keys``: {layer: (taps, left padding)} of a
    cached description being put back -- its tables are rebuilt if another shape has left different ones in the layer's buffer
    (a fresh description has just done that itself)."""
    for i, key in (table_keys or {}).items():
      if self.e._fft_table_key.get(i) != key:
        f = self.e.fft[i]
        call('st_conv1d_fft_tables_f32', f['width'], f['pl'], self.e._ptr(f['tables']), f['tables'].numel(), self.e.stream_ptr)
        self.e._fft_table_key[i] = key
    if set(self.e.fft) != getattr(self.e, '_fft_prev', None):     # a layer (re)joined the path: its spectra may be stale
      self.e._gfwd_fresh = False
      self.e._packed_t_fresh = False                             # (and a layer that left it needs its flipped copy again)
    self.e._fft_prev = set(self.e.fft)

  def _refresh_fft_filters(self, layers=None):
    """Filter spectra of the frequency-domain layers (all, or the given ones) from the current weights, in layer
    order; on a side stream an event is recorded after each layer so that the forward pass waits for the layer it is
    about to run, not for all.  (Back-prop to the input reads the same spectra, transposed.)"""
    stream = self.e._stream if self.e._stream is not None else torch.cuda.current_stream(self.e.device)
    if layers is None:
      self.e._gfwd_ready = {}
    for i, f in self.e.fft.items():
      if layers is not None and i not in layers:
        continue
      l = self.e.layers[i]
      pf = self.e._slice(self.e.params, i)[0]
      if f['shift'] is not None:
        cp = self.e.X[i].c_pitch
        n = l.width * cp * l.n_pad
        with torch.cuda.stream(stream):
          f['packed2'][f['shift'] * cp * l.n_pad:f['shift'] * cp * l.n_pad + n].copy_(pf[:n], non_blocking=True)
        pf = f['packed2']
      call('st_conv1d_fft_filters_f32', self.e._ptr(pf), f['width'], f['cin'], l.cout, f['cin_pitch'], self.e._ptr(f['tables']),
           self.e._ptr(f['gfwd']), self.e.stream_ptr)
      if stream is getattr(self.e, '_side', None):
        ev = torch.cuda.Event()
        ev.record(stream)
        self.e._gfwd_ready[i] = ev
    self.e._gfwd_fresh = True

  def _refresh_gfwd(self):
    """After an update: the bottom layer's spectra on the compute stream (the next step needs them at once; a
    cross-stream wait there costs more than the 25 us of work), the others on the side stream, bottom layer first."""
    if self.e.fft and self.e._shape is not None:
      first = min(self.e.fft)
      self.e._gfwd_ready = {}
      self._refresh_fft_filters(layers=[first])
      rest = [i for i in self.e.fft if i != first]
      if rest:
        self.e._on_side_stream(lambda: self._refresh_fft_filters(layers=rest))

  def _wait_gfwd(self, i=None):
    """The compute stream waits for the forward filter spectra of layer i (None: of every layer) if they were rebuilt on
    the side stream after the update.  The side stream works bottom layer first: the first three frequency-domain layers
    wait for their own spectra, the fourth for all that remain (by then the side stream is through, and every wait
    costs the compute stream a few microseconds)."""
    ready = getattr(self.e, '_gfwd_ready', None)
    if not ready:
      return
    order = sorted(self.e.fft)
    if i is not None and i in order and order.index(i) >= 3:
      i = None
    keys = [k for k in ready if i is None or k <= i]
    if keys:
      (self.e._stream if self.e._stream is not None else torch.cuda.current_stream(self.e.device)).wait_event(ready[max(keys)])
      for k in keys:
        del ready[k]
