"""Device-side execution of the Wav2Letter step: buffers in HBM + the launch sequence.

torch is used for device memory, streams and (in data_parallel.py) torch.distributed only; every
arithmetic op of the path is a call into libspeecht_hip.so through ``_lib`` (no fallback).

The engine is the CORE shared by the three arithmetic modes: weights / gradients / Adam state in four flat fp32 buffers, the
padded NWC activation tensors (engine_buffers.py), the role streams and the side-stream fork / join (engine_streams.py), the
input side (H2D staging, label uploads), CTC, clip + Adam, loss read-back and the decoders
(engine_decode.py).  What differs between fp32, bf16x6 and bf16 -- derived operands and the forward / backward launch
sequences -- lives in speecht_amd/modes/ behind `self.mode`.
"""
import ctypes
import math
import os

import numpy as np
import torch

from . import _lib
from ._lib import Tensor3, call
from .engine_buffers import DevTensor3, LayerSpec, _round_up, _StagedHostBatch, _Storage, channel_pitch, same_padding   # noqa: F401
from .engine_decode import DecodeMixin, _PendingBeamDecode, _PendingDecode, beam_input_transform, merge_repeated_labels   # noqa: F401
from .engine_streams import decoder_stream_pair, decoder_streams, role_stream   # noqa: F401
from .modes import make_mode


class Wav2LetterEngine(DecodeMixin):
  """Owns weights/optimizer state and runs forward / loss / backward / update on one GPU."""

  def __init__(self, layers, device='cuda:0', stream=None, conv_mode=None, split_small_batches=True, fft_conv=None):
    _lib.load()
    # With few output rows (single-utterance / live inference) the forward GEMMs split their reduction over
    # the otherwise idle CUs (3x lower latency for one 2 s utterance).  The summation order then depends on the
    # batch size, so an utterance's logits are equal to ~1e-6 rather than bit-identical across batch sizes;
    # split_small_batches=False keeps the single-pass kernels for every shape (bit-exact batch invariance).
    self.split_small_batches = bool(split_small_batches)
    # 'fp32': exact-f32 MFMA kernels (default).  'bf16x6': EXPERIMENTAL fp32-accurate split-bf16 path
    # (csrc/conv_bf16.hip, NP = 3) for forward and back-prop-to-input of the wide layers.
    # 'bf16': BASELINE config 4 -- bf16 activations and activation gradients, fp32 masters / accumulation /
    # logits / CTC / Adam (csrc/conv_bf16.hip, NP = 1).
    self.conv_mode = conv_mode or os.environ.get('ST_CONV_MODE', 'fp32')
    assert self.conv_mode in ('fp32', 'bf16x6', 'bf16'), self.conv_mode
    # fp32 mode: long, wide filters (the 32-tap 250 -> 2000 layer) run in the frequency domain (csrc/conv_fft.hip:
    # ~10x fewer multiplications, same fp32 arithmetic class; results differ from the W-tap kernels by rounding,
    # ~1e-6 of the tensor scale).  fft_conv=False / ST_FFT_CONV=0 keeps the W-tap kernels everywhere.
    self.fft_conv = (os.environ.get('ST_FFT_CONV', '1') != '0') if fft_conv is None else bool(fft_conv)
    self.fft_min_width = int(os.environ.get('ST_FFT_MIN_WIDTH', '7'))
    # output rows (B * T') from which the path pays (measured on training steps of 1-16 x 10 s utterances): the 32-tap
    # layer from ~1 000 rows (B = 2: 3.05 -> 2.94 ms, B = 4: 3.71 -> 3.17), the 7-tap layers and the first layer from ~3 000
    # (B = 4, 2 004 rows: 3.17 -> 3.27 with them; B = 8, 4 008 rows: 3.86 -> 3.66); below that the W-tap kernels with their
    # split reductions are faster (B = 1: 2.64 vs 2.88)
    self.fft_min_rows = int(os.environ.get('ST_FFT_MIN_ROWS', '1000'))
    self.fft_min_rows_narrow = int(os.environ.get('ST_FFT_MIN_ROWS_NARROW', '3000'))
    # the stride-2 first layer (48 taps over 80 mel channels) on its polyphase view: 25 taps over 160 channels
    self.fft_first_layer = os.environ.get('ST_FFT_FIRST_LAYER', '1') != '0'
    self.side_filter_gradient = os.environ.get('ST_WGRAD_SIDE', '1') != '0'
    self.device = torch.device(device)
    if self.device.type != 'cuda':
      raise _lib.SpeechtHipError('Wav2LetterEngine needs a GPU device (no CPU path exists)')
    role_stream(self.device, 'side')               # the role -> hardware-queue map is fixed before anything else makes streams
    self.layers = [LayerSpec(*l) for l in layers]
    self.num_classes = self.layers[-1].cout
    self._stream = stream
    # flat parameter layout
    self.offsets = []
    off = 0
    for l in self.layers:
      self.offsets.append((off, off + l.k_pad * l.n_pad))
      off += l.k_pad * l.n_pad + l.n_pad
    self.n_flat = off
    z = lambda: torch.zeros(self.n_flat, dtype=torch.float32, device=self.device)
    self.params, self.adam_m, self.adam_v = z(), z(), z()
    # the gradient buffer carries one extra slot behind the last layer: the update gate (number of utterances of
    # this step that CTC could not align).  It sits inside the last all-reduce bucket so that in data-parallel
    # training every rank sees the global count and skips the update together (apply_update / fetch_losses).
    self.reduce_buffer = torch.zeros(self.n_flat + 16, dtype=torch.float32, device=self.device)
    self.grads = self.reduce_buffer[:self.n_flat]
    self.gate = self.reduce_buffer[self.n_flat:self.n_flat + 1]
    # ... and next to it this rank's share of the global mean loss (st_ctc_status_gate_loss_f32): SUM-reduced with the gradients,
    # slot 1 holds the mean loss of the GLOBAL batch on every rank -- no scalar exchange of its own (speech_model.py:75)
    self.gate_slots = self.reduce_buffer[self.n_flat:self.n_flat + 2]
    self.packed_t = [None] + [torch.zeros(l.kt_pad * l.nt_pad, dtype=torch.float32, device=self.device)
                              for l in self.layers[1:]]
    self._packed_t_fresh = False
    self.stats = torch.zeros(2, dtype=torch.float32, device=self.device)
    self.norm_ws = torch.zeros(_lib.load().st_global_norm_ws(self.n_flat) // 4, dtype=torch.float32,
                               device=self.device)
    self.step_count = 0
    self._shape = None
    self._storage = _Storage(self.device)
    self.ctc_ws = None
    self._wplanes_fresh = False
    self._wtplanes_fresh = False
    self._gfwd_fresh = False
    self.fft = {}
    self.fftb = {}
    self.mode = make_mode(self)                    # the arithmetic mode's buffers and launch sequences (speecht_amd/modes/)
    # lost stream-K hand-offs are counted per process by the library; this engine reports the ones after its creation
    seen = ctypes.c_uint32(0)
    call('st_streamk_lost_count', ctypes.byref(seen))
    self._sk_lost = [torch.zeros(1, dtype=torch.int32, pin_memory=True), int(seen.value)]

  def __getattr__(self, name):
    """Mode-level helpers (`_use_fft`, `_transposed_in_place`, `_refresh_fft_filters`, ...) stay reachable on the engine: tests,
    bench.py and the profiling scripts call a few of them.  Only reached when the engine itself has no such attribute."""
    mode = self.__dict__.get('mode')
    if mode is not None and not name.startswith('__') and hasattr(type(mode), name):
      return getattr(mode, name)
    raise AttributeError("'{}' object has no attribute '{}'".format(type(self).__name__, name))

  # ---- plumbing --------------------------------------------------------------------------

  @property
  def stream_ptr(self):
    s = self._stream if self._stream is not None else torch.cuda.current_stream(self.device)
    return ctypes.c_void_p(s.cuda_stream)

  def _slice(self, flat, i):
    fo, bo = self.offsets[i]
    l = self.layers[i]
    return flat[fo:bo], flat[bo:bo + l.n_pad]

  @property
  def layer_ranges(self):
    """(start, end) of each layer's filters+bias inside the flat buffers."""
    return [(fo, bo + l.n_pad) for (fo, bo), l in zip(self.offsets, self.layers)]

  @property
  def reduce_ranges(self):
    """``layer_ranges`` over ``reduce_buffer``: the last layer's slice also carries the update gate."""
    r = self.layer_ranges
    r[-1] = (r[-1][0], self.n_flat + 16)
    return r

  def _ptr(self, t):
    return ctypes.c_void_p(t.data_ptr())

  # ---- weights in the reference's layout (exporting.py:30-40: [W, Cin, Cout] + [Cout]) ----------

  def _pack(self, flat, params):
    """Fill one of the flat buffers (weights, Adam m or v) from per-layer arrays in the reference's layout."""
    flat.zero_()
    for i, ((F, b), l) in enumerate(zip(params, self.layers)):
      assert F.shape == (l.width, l.cin, l.cout) and b.shape == (l.cout,), (i, F.shape, b.shape)
      Fd = torch.as_tensor(np.ascontiguousarray(F), dtype=torch.float32).to(self.device).contiguous()
      pf, pb = self._slice(flat, i)
      call('st_pack_filters_f32', self._ptr(Fd), l.width, l.cin, l.cout, l.cin_pitch, self._ptr(pf), self.stream_ptr)
      pb[:l.cout] = torch.as_tensor(np.asarray(b), dtype=torch.float32).to(self.device)
    torch.cuda.synchronize(self.device)

  def set_weights(self, params):
    """params: list of (filters [W,Cin,Cout], bias [Cout]) numpy arrays."""
    self._pack(self.params, params)
    self.mark_weights_changed()

  def set_adam_state(self, m, v, step):
    """Adam moments in the reference's layout (lists like ``set_weights``) and the number of updates applied."""
    self._pack(self.adam_m, m)
    self._pack(self.adam_v, v)
    self.step_count = int(step)

  def get_adam_state(self):
    return self._unpack(self.adam_m), self._unpack(self.adam_v)

  def mark_weights_changed(self):
    """Call after writing ``self.params`` directly: derived operand copies are rebuilt on next use."""
    self._packed_t_fresh = False
    self._wplanes_fresh = False
    self._wtplanes_fresh = False
    self._gfwd_fresh = False

  def _unpack(self, flat):
    out = []
    for i, l in enumerate(self.layers):
      pf, pb = self._slice(flat, i)
      Fd = torch.empty(l.width * l.cin * l.cout, dtype=torch.float32, device=self.device)
      call('st_unpack_filters_f32', self._ptr(pf), l.width, l.cin, l.cout, l.cin_pitch, self._ptr(Fd), self.stream_ptr)
      out.append((Fd.view(l.width, l.cin, l.cout).cpu().numpy(), pb[:l.cout].cpu().numpy()))
    return out

  def get_weights(self):
    return self._unpack(self.params)

  def get_grads(self):
    return self._unpack(self.grads)

  def init_xavier(self, seed=None):
    """tf.contrib.layers.xavier_initializer + zero biases (speech_model.py:150-152)."""
    rng = np.random.default_rng(seed)
    params = []
    for l in self.layers:
      limit = math.sqrt(6.0 / (l.width * l.cin + l.width * l.cout))
      params.append((rng.uniform(-limit, limit, (l.width, l.cin, l.cout)).astype(np.float32),
                     np.zeros(l.cout, np.float32)))
    self.set_weights(params)

  # ---- activation buffers ------------------------------------------------------------------

  def _tensor(self, name, batch, frames, channels, halo_l, halo_r, clear=False):
    storage, fresh = self._storage.view(name, DevTensor3.numel(batch, frames, channels, halo_l, halo_r))
    t = DevTensor3(storage, batch, frames, channels, halo_l, halo_r)
    # what has to happen whenever the shape is (re-)entered on storage another shape has written: the other shape's interiors
    # lie where this shape's halo rows are
    if not fresh:
      if clear:
        t.buf.zero_()
      else:
        call('st_zero_halos_f32', t.ref, self.stream_ptr)
    log = self.__dict__.get('_describe_log')
    if log is not None:
      # (byte ranges: the whole view, or the halo rows -- the rows behind utterance b and in front of b + 1 are one range)
      base, row = t.buf.data_ptr(), t.c_pitch * 4
      if clear:
        log.append((base, t.buf.numel() * 4))
      else:
        tail = t.t_pitch - t.halo - t.frames
        for b in range(t.batch + 1):
          lo = b * t.t_pitch - (tail if b > 0 else 0)
          hi = b * t.t_pitch + (t.halo if b < t.batch else 0)
          if hi > lo:
            log.append((base + lo * row, (hi - lo) * row))
    return t

  # ---- shapes seen before -----------------------------------------------------------------------------------------
  # The reference pads every batch to its own longest member (speech_input.py:37-45): (B, max_T) changes nearly every step, and a
  # training run walks the same few hundred shapes over and over.  Describing a shape is ~60 C calls (geometry, workspace sizes,
  # plans) and as many Python objects -- 0.55 ms of host time in fp32, 1.0 ms with bf16 activations, where it is what bounds the
  # step (2.4 ms of kernels, enqueued in 1.0 ms).  A description is therefore kept per (B, T): re-entering a shape puts the cached
  # descriptors back, re-zeroes what another shape's interiors may have overwritten (halo rows, the input tensor, the bf16
  # planes: ONE launch over a device table of byte ranges, st_zero_regions) and re-evaluates only what depends on the shape left behind (the modes' `reenter`: stale filter spectra when the set
  # of frequency-domain layers changed, transform tables).  Entries die with the storage generation they were described on.
  _SHAPE_ATTRS = ('X', 'dZ', 'geo', 't_out', 'loss_pair', 'loss', 'loss_lo', 'ctc_status', 'dec_ids', 'dec_lens', 'dec_score')

  def _shape_knobs(self):
    env = os.environ.get
    return (self.fft_conv, self.fft_min_width, self.fft_min_rows, self.fft_min_rows_narrow, self.fft_first_layer,
            self.side_filter_gradient, self.split_small_batches, _lib.TUNING_EPOCH[0], env('ST_WGRAD_SIDE_TOP'), env('ST_FFT_BF16'),
            env('ST_BF16_WGRAD_TR'))

  def _reenter_shape(self, batch, frames):
    """Put a cached description of (batch, frames) back; False when there is none (or it is stale)."""
    cache = self.__dict__.get('_shape_cache')
    entry = cache.get((batch, frames)) if cache else None
    if entry is None or entry['generation'] != self._storage.generation or entry['knobs'] != self._shape_knobs():
      return False
    self.__dict__.update(entry['state'])
    table, count = entry['zero']
    if count:
      call('st_zero_regions', self._ptr(table), count, self.stream_ptr)
    self.mode.reenter(entry['mode'])
    self._shape = (batch, frames)
    return True

  def _remember_shape(self, batch, frames, ranges):
    if not self.mode.shape_attrs or os.environ.get('ST_SHAPE_CACHE', '1') == '0':
      return
    cache = self.__dict__.setdefault('_shape_cache', {})
    generation = self._storage.generation
    if any(e['generation'] != generation for e in cache.values()) or len(cache) >= 4096:
      cache.clear()                                  # (their views may point into buffers that have been replaced since)
    state = {k: self.__dict__[k] for k in self._SHAPE_ATTRS + tuple(self.mode.shape_attrs) if k in self.__dict__}
    # the byte ranges to zero on re-entry as a device table, long ranges in pieces of 1 MB (one workgroup each: st_zero_regions)
    pieces = []
    for address, nbytes in ranges:
      assert address % 16 == 0 and nbytes % 16 == 0, (address, nbytes)
      for off in range(0, nbytes, 1 << 20):
        pieces.append((address + off, min(1 << 20, nbytes - off)))
    table = torch.from_numpy(np.asarray(pieces, dtype=np.uint64).reshape(-1, 2).view(np.int64)).to(self.device) if pieces else None
    cache[(batch, frames)] = dict(state=state, zero=(table, len(pieces)), generation=generation, knobs=self._shape_knobs(),
                                  mode=self.mode.shape_token())

  def _ensure_shape(self, batch, frames):
    if self._shape == (batch, frames):
      return
    if self._reenter_shape(batch, frames):
      return
    self._describe_log = ranges = []
    try:
      self._describe_shape(batch, frames)
    finally:
      self._describe_log = None
    self._remember_shape(batch, frames, ranges)

  def _describe_shape(self, batch, frames):
    dev = self.device
    self.X, self.dZ = [], []
    t = frames
    geo = []
    for l in self.layers:
      t_out, pl, pr = same_padding(t, l.width, l.stride)
      geo.append((t, t_out, pl, pr))
      t = t_out
    self.geo = geo
    for i, l in enumerate(self.layers):
      t_in, t_out, pl, pr = geo[i]
      halo_l, halo_r = pl, max(pr, (t_out - 1) * l.stride + l.width - pl - t_in)
      if l.stride == 2:
        # a stride-2 layer may run on its polyphase view (frame pairs as channels, `_polyphase`): that view needs an
        # even left halo and an even frame pitch
        halo_l += halo_l & 1
        halo_r += (halo_l + t_in + halo_r) & 1
      # X[0]'s pad channels are not written by any kernel: clear the whole view on re-use
      self.X.append(self._tensor('X%d' % i, batch, t_in, l.cin, halo_l, halo_r, clear=(i == 0)))
      # gradient wrt this layer's pre-activation output; halo for its own back-prop-to-input conv
      self.dZ.append(self._tensor('dZ%d' % i, batch, t_out, l.cout, l.width - 1 - pl, pl))
    last = self.layers[-1]
    self.X.append(self._tensor('X%d' % len(self.layers), batch, geo[-1][1], last.cout, 0, 0))   # logits [B, T', C]
    self.t_out = geo[-1][1]
    # per-utterance CTC losses as (hi, lo) float pairs: `loss` is the fp32 value (what tf.nn.ctc_loss returns), `loss_lo` what
    # fp32 cannot hold of -log p at that magnitude (st_ctc_loss_grad_hilo_f32); one buffer, so one copy brings both back
    loss_buf = self._storage.view('loss', 2 * batch)[0]
    self.loss_pair = loss_buf[:2 * batch]
    self.loss, self.loss_lo = loss_buf[:batch], loss_buf[batch:2 * batch]
    self.ctc_status = self._storage.view('ctc_status', batch, torch.int32)[0][:batch]
    self.dec_ids = self._storage.view('dec_ids', batch * self.t_out, torch.int32)[0][:batch * self.t_out]
    self.dec_lens = self._storage.view('dec_lens', batch, torch.int32)[0][:batch]
    self.dec_score = self._storage.view('dec_score', batch)[0][:batch]
    self.mode.alloc(batch)                         # what this arithmetic needs beyond the shared buffers
    self._shape = (batch, frames)

  def reserve(self, batch, max_frames, min_frames=None, step=64):
    """Size every named device buffer for training batches of up to ``batch`` x ``max_frames`` BEFORE the first step: the
    reference pads every batch to its own longest member (speech_input.py:37-45), so (B, max_T) changes nearly every step, and
    the grow-only storage would otherwise re-allocate whenever a longer batch arrives.  Buffer sizes are not monotone in the
    length (short batches split their reductions into slabs, the frequency-domain layers switch on above a row count), so a
    ladder of lengths is described once -- host work and a few halo-clearing launches each -- largest first."""
    lo = int(min_frames) if min_frames else int(step)
    ladder = sorted({int(max_frames)} | set(range(lo, int(max_frames), int(step))), reverse=True)
    for frames in ladder:
      self._ensure_shape(int(batch), frames)

  def _planes(self, name, numel, n=3, slack=0):
    """n zeroed bf16 planes of `numel` elements each (whole buffer cleared when re-used).  ``slack``: that many further zero
    elements stay allocated behind the (single) plane -- readable zeros for kernels that run past the last row
    (st_conv1d_nwc_bwd_filter_tr_bf16); the returned view does not include them."""
    buf, fresh = self._storage.view(name, n * numel + slack, torch.bfloat16)
    v = buf[:n * numel + slack]
    if not fresh:
      v.zero_()
    log = self.__dict__.get('_describe_log')
    if log is not None:
      log.append((v.data_ptr(), -(-v.numel() * 2 // 16) * 16))      # (whole 16-byte units: the storage is allocated in larger ones)
    return v[:n * numel]

  # ---- the path ----------------------------------------------------------------------------

  def load_batch(self, inputs, seq_lens):
    """inputs: [B, T, input_size] (numpy or torch, any float dtype, or a ``speech_input.StagedBatch`` that the
    input pipeline already copied to the device); seq_lens: [B] unpadded frames."""
    if hasattr(inputs, 'event') and hasattr(inputs, 'tensor'):
      stream = self._stream if self._stream is not None else torch.cuda.current_stream(self.device)
      stream.wait_event(inputs.event)              # H2D ran on the pipeline's copy stream
      inputs.tensor.record_stream(stream)          # keep the allocator from recycling it under the copy below
      staged, inputs = inputs, inputs.tensor
    else:
      staged = None
    x = torch.as_tensor(inputs)
    B, T, C = x.shape
    assert C == self.layers[0].cin, 'input_size mismatch'
    self._ensure_shape(B, T)
    self.X[0].interior().copy_(x.to(torch.float32), non_blocking=True)
    if staged is not None and hasattr(staged, 'consumed'):
      staged.taken = True
      staged.consumed.record(self._stream if self._stream is not None else torch.cuda.current_stream(self.device))
    self.seq_lens_host = np.asarray(seq_lens, dtype=np.int64)
    # the reference feeds sequence_lengths // 2 to CTC and the decoder (speech_model.py:74,114)
    self.ctc_lens = self._upload_i32((self.seq_lens_host // 2).astype(np.int32))

  def stage_host_batch(self, x_host):
    """Asynchronous H2D copy of a padded feature batch [B, T, C] (float32; a pinned torch tensor copies without
    an intermediate host copy) on the engine's copy stream into one of two staging buffers in HBM.  Returns a
    handle for ``load_batch``; the copy of batch k+1 overlaps the kernels of batch k.  A staging buffer is
    re-used only after the compute stream has consumed it (event recorded by ``load_batch``)."""
    if not hasattr(self, '_h2d'):
      self._h2d = dict(stream=role_stream(self.device, 'h2d'), slots=[None, None], turn=0)
    h = self._h2d
    x = torch.as_tensor(x_host)
    if x.dtype != torch.float32:
      x = x.to(torch.float32)
    # the slot whose turn it is, else the other one if that is free: the turn only moves once a slot is really taken, so a
    # refused call (both busy) leaves the state as it was and the next call succeeds as soon as either batch is consumed
    turn = h['turn'] ^ 1
    for cand in (turn, turn ^ 1):
      slot = h['slots'][cand]
      if slot is None or slot.taken:
        turn = cand
        break
    else:
      # two staging buffers: a third batch staged before either was handed to load_batch would overwrite one
      raise RuntimeError('stage_host_batch: both staging buffers hold batches that load_batch has not consumed yet '
                         '(discard_staged_batch(handle) releases one that will not be used)')
    h['turn'] = turn
    if slot is None or slot.tensor.shape != x.shape:
      if slot is not None:
        slot.consumed.synchronize()
      slot = _StagedHostBatch(torch.empty(x.shape, dtype=torch.float32, device=self.device))
      h['slots'][turn] = slot
    slot.taken = False
    with torch.cuda.stream(h['stream']):
      h['stream'].wait_event(slot.consumed)            # the compute stream is done reading this buffer
      slot.tensor.copy_(x, non_blocking=True)
      slot.event.record(h['stream'])
    return slot

  def discard_staged_batch(self, staged):
    """Release a staged batch that will not be handed to ``load_batch`` (end of an epoch, an exception in the feeder): its
    staging buffer becomes free for the next ``stage_host_batch`` once the copy into it has finished."""
    if not staged.taken:
      staged.taken = True
      staged.consumed.record(self._h2d['stream'])      # "consumed" right behind the copy on the copy stream

  def _upload_i32(self, values):
    """Small int32 host array -> device through a ring of pinned slots.  A hipMemcpyAsync from pageable memory
    only returns once the stream's earlier kernels have finished, which would stall the thread that enqueues
    the next batch behind the previous batch's forward; from pinned memory the copy is a stream operation."""
    if not hasattr(self, '_pin_ring'):
      self._pin_ring, self._pin_turn = [[None, None] for _ in range(8)], 0
    slot = self._pin_ring[self._pin_turn % len(self._pin_ring)]
    self._pin_turn += 1
    n = int(values.shape[0])
    if slot[1] is not None:
      slot[1].synchronize()                        # the copy that last read this slot (8 uploads ago)
    if slot[0] is None or slot[0].numel() < n:
      slot[0] = torch.empty(max(n, 1024), dtype=torch.int32, pin_memory=True)
    slot[0][:n].copy_(torch.from_numpy(np.ascontiguousarray(values, dtype=np.int32)))
    # on the copy stream: the consumers (CTC, the decoders) come a whole forward pass later and wait for the event
    # there (`_wait_uploads`); on the compute stream three such copies cost the start of every step ~40 us
    main = self._stream if self._stream is not None else torch.cuda.current_stream(self.device)
    if not hasattr(self, '_up_stream'):
      self._up_stream, self._uploads = role_stream(self.device, 'upload'), []
    with torch.cuda.stream(self._up_stream):
      dev = torch.empty(n, dtype=torch.int32, device=self.device)
      dev.copy_(slot[0][:n], non_blocking=True)
      if slot[1] is None:
        slot[1] = torch.cuda.Event()
      slot[1].record(self._up_stream)
    dev.record_stream(main)
    self._uploads.append(slot[1])
    return dev

  def _wait_uploads(self):
    """The compute stream waits for the int32 uploads (lengths, labels) issued since the last call."""
    ups = getattr(self, '_uploads', None)
    if ups:
      (self._stream if self._stream is not None else torch.cuda.current_stream(self.device)).wait_event(ups[-1])
      del ups[:]

  def forward(self):
    """X[0] -> logits X[-1] through the eleven layers (speech_model.py:279-295), by the arithmetic mode's kernels."""
    return self.mode.forward()

  def forward_graph(self):
    """``forward()`` replayed from a HIP graph: the launch sequence of the current (batch, frames) shape is
    captured once and replayed with a single launch afterwards -- for small batches (live / single-utterance
    inference) the eleven kernels are launch-bound and the CPU cost of enqueueing them is the latency.
    Buffers, weights and the batch are read through the same device pointers on replay, so new inputs
    (``load_batch`` with the same shape) and in-place weight updates are picked up; a new shape captures anew."""
    if not hasattr(self, '_graphs'):
      self._graphs, self._graph_seen = {}, set()
    self.mode.prepare_forward_graph()              # derived operands are rebuilt, outside events waited for, outside the graph
    key = (self._shape, self._storage.generation)
    graph = self._graphs.get(key)
    if graph is None:
      # a shape is captured the second time it shows up: the first pass runs eagerly (it also is the warm-up
      # -- lazy allocations, env lookups), so a stream of all-different shapes pays nothing for graphs
      if key not in self._graph_seen:
        self._graph_seen = {k for k in self._graph_seen if k[1] == self._storage.generation} | {key}
        return self.forward()
      self._graphs = {k: g for k, g in self._graphs.items() if k[1] == self._storage.generation}   # drop stale captures
      torch.cuda.synchronize(self.device)
      graph = torch.cuda.CUDAGraph()
      own_stream, self._stream = self._stream, None
      try:
        with torch.cuda.graph(graph):              # our C ABI launches on torch's current (capturing) stream
          self.forward()
      finally:
        self._stream = own_stream
      self._graphs[key] = graph
    graph.replay()

  def logits_time_major(self):
    """[T', B, C] like tf.transpose(outputs, (1, 0, 2)) (speech_model.py:295)."""
    return self.X[-1].interior().permute(1, 0, 2)

  def set_labels(self, label_list):
    lens = [len(l) for l in label_list]
    offs = np.zeros(len(label_list) + 1, dtype=np.int32)
    offs[1:] = np.cumsum(lens)
    self.max_label_len = int(max(lens + [0]))
    # CSR ids (+1 pad entry so that the buffer is never empty); array-per-utterance inputs stay in numpy
    ids = np.concatenate([np.asarray(l, dtype=np.int32).reshape(-1) for l in label_list] + [np.zeros(1, np.int32)])
    # tf.nn.ctc_loss raises InvalidArgument for a label outside [0, num_classes - 1); the kernels index LDS with
    # the id, so it must never reach them (vocabulary.letter_to_id maps e.g. a digit to a negative id)
    self._rejected_labels = []
    if ids.size > 1 and (int(ids.min()) < 0 or int(ids.max()) >= self.num_classes - 1):
      bad = [b for b, l in enumerate(label_list) if len(l) and (min(l) < 0 or max(l) >= self.num_classes - 1)]
      if not getattr(self, 'defer_label_errors', False):
        raise ValueError('label ids must lie in [0, {}) (blank = {}); offending utterances: {}'.format(
            self.num_classes - 1, self.num_classes - 1, bad))
      # Data-parallel training (set by SpeechModel.enable_data_parallel): raising here, on this rank only, would leave
      # the other ranks waiting in the gradient all-reduce.  The offending utterances get an empty label (nothing
      # out of range reaches a kernel) and a status word of their own after the CTC call; the status count travels
      # with the gradients, every rank's update is gated off together and every rank raises in fetch_losses.
      self._rejected_labels = bad
      label_list = [[] if b in bad else l for b, l in enumerate(label_list)]
      lens = [len(l) for l in label_list]
      offs[1:] = np.cumsum(lens)
      self.max_label_len = int(max(lens + [0]))
      ids = np.concatenate([np.asarray(l, dtype=np.int32).reshape(-1) for l in label_list] + [np.zeros(1, np.int32)])
    self.label_ids = self._upload_i32(ids)
    self.label_offs = self._upload_i32(offs)

  def _on_side_stream(self, fn, second=False):
    """Run ``fn`` (which enqueues kernels through ``self.stream_ptr``) on the engine's side stream (``second``: on a
    second one), ordered after everything enqueued so far on the compute stream; ``_join_side_stream`` makes the
    compute stream wait for both.  Used to put small HBM-bound operand preparation next to the CTC recursion, which is
    a latency chain of 500 dependent steps on 64 wavefronts and leaves the rest of the chip idle, and independent
    chains of the backward pass next to each other."""
    main = self._stream if self._stream is not None else torch.cuda.current_stream(self.device)
    if getattr(self, '_side', None) is None:
      self._side = role_stream(self.device, 'side')
    if second and getattr(self, '_side2', None) is None:
      self._side2 = role_stream(self.device, 'side2')
    stream = self._side2 if second else self._side
    fork = torch.cuda.Event()
    fork.record(main)
    stream.wait_event(fork)
    saved, self._stream = self._stream, stream
    try:
      fn()
    finally:
      self._stream = saved
    done = torch.cuda.Event()
    done.record(stream)
    if second:
      self._side2_done = done
    else:
      self._side_done = done

  def _join_side_stream(self, second_only=False):
    main = self._stream if self._stream is not None else torch.cuda.current_stream(self.device)
    for name in (('_side2_done',) if second_only else ('_side_done', '_side2_done')):
      done = getattr(self, name, None)
      if done is not None:
        main.wait_event(done)
        setattr(self, name, None)

  def ctc_loss_grad(self, grad_scale):
    B, T = self.X[-1].batch, self.X[-1].frames
    lib = _lib.load()
    need = lib.st_ctc_ws(B, T, self.max_label_len)
    if need == 0:
      raise ValueError('label of length {} is too long for the CTC kernel (max 511)'.format(self.max_label_len))
    if self.ctc_ws is None or self.ctc_ws.numel() * 4 < need:
      self.ctc_ws = torch.empty(need // 4 + 64, dtype=torch.float32, device=self.device)
    # the filter operands of back-prop that derive from the weights Adam just updated are rebuilt on the side stream while
    # the CTC recursion runs (a latency chain of 500 dependent steps on 64 wavefronts: the rest of the chip is idle)
    self.mode.refresh_under_ctc()
    self._wait_uploads()
    call('st_ctc_loss_grad_hilo_f32', self.X[-1].ref, self._ptr(self.label_ids), self._ptr(self.label_offs),
         self._ptr(self.ctc_lens), self.max_label_len, float(grad_scale), self._ptr(self.loss), self._ptr(self.loss_lo), self.dZ[-1].ref,
         self._ptr(self.ctc_status), self._ptr(self.ctc_ws), self.ctc_ws.numel() * 4, self.stream_ptr)
    if getattr(self, '_rejected_labels', None):            # labels refused on the host (deferred mode): status 2
      stream = self._stream if self._stream is not None else torch.cuda.current_stream(self.device)
      with torch.cuda.stream(stream):
        self.ctc_status.index_fill_(0, torch.as_tensor(self._rejected_labels, dtype=torch.int64).to(self.device, non_blocking=True), 2)
    call('st_ctc_status_gate_loss_f32', self._ptr(self.ctc_status), B, self._ptr(self.loss), self._ptr(self.loss_lo), float(grad_scale),
         self._ptr(self.gate), self.stream_ptr)

  def backward(self, on_layer_done=None, hook_layers=None):
    """Back-prop from dZ[-1] (already holding d avg_loss / d logits).  ``on_layer_done(i)`` is
    called after layer i's filter/bias gradients have been enqueued (for bucketed all-reduce) -- for every layer, or only
    for those in ``hook_layers`` (the layers that complete a reduce bucket, `GradientAllReducer.hook_layers`): a hook on a
    layer whose filter gradient runs on the side stream makes the compute stream WAIT for that stream first, so hooks
    nobody needs cost the overlap of the two chains (round 4: a forced world-1 exchange cost the step 0.24 ms, most of it
    six such waits)."""
    wanted = (lambda i: True) if hook_layers is None else (lambda i: i in hook_layers)
    return self.mode.backward(on_layer_done, wanted)

  def _adam_rate(self, lr, beta1, beta2):
    """The bias-corrected rate of the NEXT update (tf.train.AdamOptimizer: lr * sqrt(1 - beta2^t) / (1 - beta1^t)); counts it."""
    self.step_count += 1
    t = self.step_count
    return lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)

  def _refresh_after_update(self):
    """The operands the NEXT forward pass derives from the weights: filter spectra of the frequency-domain layers / the bf16 filter
    copies -- the bottom layer's on the compute stream, the others on the side stream with an event each."""
    if self._shape is not None:
      self.mode.refresh_after_update()

  def apply_update(self, lr, max_grad_norm=5.0, beta1=0.9, beta2=0.999, eps=1e-3):
    """clip_by_global_norm + tf.train.AdamOptimizer(epsilon=1e-3) (speech_model.py:77-82)."""
    lr_t = self._adam_rate(lr, beta1, beta2)
    # gated on the device: a step whose batch CTC rejected (on any rank) leaves params / m / v untouched
    call('st_global_norm_clip_adam_gated_f32', self._ptr(self.params), self._ptr(self.grads), self._ptr(self.adam_m),
         self._ptr(self.adam_v), self.n_flat, float(max_grad_norm), float(lr_t), beta1, beta2, eps,
         self._ptr(self.stats), self._ptr(self.gate), self._ptr(self.norm_ws), self.norm_ws.numel() * 4,
         self.stream_ptr)
    self._updates_in_flight = getattr(self, '_updates_in_flight', 0) + 1
    self.mark_weights_changed()
    self._refresh_after_update()

  def fetch_losses(self, precise=False):
    """Per-utterance CTC losses [B] on the host, after checking the status words: both arrays come back in one
    pinned, asynchronous copy each and a single event wait (a step's only host synchronisation).  float32 like
    tf.nn.ctc_loss; ``precise=True`` returns float64 = hi + lo of the kernel's (hi, lo) pairs (-log p to ~1e-6 where one
    fp32 ulp of a 10 s utterance's loss is 1.2e-4)."""
    return self.fetch_losses_end(self.fetch_losses_begin(), precise)

  def fetch_losses_begin(self, stream=None):
    """The read-back of `fetch_losses` enqueued where the stream stands NOW: losses, CTC status words, the update gate, the mean-loss
    slot and the library's lost-hand-off count go to pinned host memory behind everything enqueued so far, an event marks the
    copies.  Called right behind `ctc_loss_grad`, with back-prop and the update enqueued after it, `fetch_losses_end` returns as
    soon as CTC is through -- the host hands the loss back to its caller and prepares the next batch while the GPU is still in the
    backward pass.  ``stream``: read back on that stream instead of the compute stream (data parallelism: the gate and the mean
    loss are final once the FIRST gradient bucket is reduced -- the caller makes ``stream`` wait for that bucket and nothing else,
    `GradientAllReducer.after_first_bucket`)."""
    B = self.loss.numel()
    if not hasattr(self, '_loss_host') or self._loss_host[0].numel() < 2 * B:
      self._loss_host = (torch.empty(max(2 * B, 128), dtype=torch.float32, pin_memory=True),
                         torch.empty(max(B, 64), dtype=torch.int32, pin_memory=True), torch.cuda.Event(),
                         torch.empty(16, dtype=torch.float32, pin_memory=True))
    loss_h, status_h, event, gate_h = self._loss_host
    if stream is None:
      stream = self._stream if self._stream is not None else torch.cuda.current_stream(self.device)
    lost_h = self._streamk_lost_async(stream)
    with torch.cuda.stream(stream):
      loss_h[:2 * B].copy_(self.loss_pair, non_blocking=True)
      status_h[:B].copy_(self.ctc_status, non_blocking=True)
      gate_h[:2].copy_(self.gate_slots, non_blocking=True)
      event.record(stream)
    return (B, lost_h)

  def fetch_losses_end(self, handle, precise=False):
    """Wait for the copies of `fetch_losses_begin`, check the status words (raises like `fetch_losses`), return the losses."""
    B, lost_h = handle
    loss_h, status_h, event, gate_h = self._loss_host
    event.synchronize()
    self._check_streamk_lost(lost_h)
    st = status_h[:B].numpy()
    skipped = getattr(self, '_updates_in_flight', 0)
    self._updates_in_flight = 0
    if st.any() or float(gate_h[0]) != 0.0:
      # the gated Adam kernel enqueued behind this CTC evaluation was a no-op: take its step count back so that
      # the bias correction stays in step with the updates that really happened
      self.step_count -= min(skipped, 1)
      if (st == 2).any():
        raise ValueError('label ids must lie in [0, {}) (blank = {}); offending utterances: {}'.format(
            self.num_classes - 1, self.num_classes - 1, np.nonzero(st == 2)[0].tolist()))
      if st.any():
        raise ValueError('Not enough time for target transition sequence (utterances {})'.format(np.nonzero(st)[0].tolist()))
      raise ValueError('batch rejected: {:g} utterance(s) on other ranks had no valid CTC alignment or out-of-range label ids'
                       .format(float(gate_h[0])))
    pair = loss_h[:2 * B].numpy()
    self.mean_loss_reduced = float(gate_h[1])      # sum over ranks of (sum_b loss_b) * grad_scale: the global mean under data parallelism
    if precise:
      return pair[:B].astype(np.float64) + pair[B:].astype(np.float64)
    return pair[:B].copy()

  def _streamk_lost_async(self, stream):
    """The library's count of lost stream-K hand-offs (st_streamk_lost_ptr: a reader's bounded poll ran out and its tile became
    NaN) on its way to pinned host memory behind everything enqueued on ``stream``; `_check_streamk_lost` reads it after the
    caller's own synchronisation."""
    host = self._sk_lost[0]
    call('st_streamk_lost_fetch_async', ctypes.c_void_p(host.data_ptr()), ctypes.c_void_p(stream.cuda_stream))
    return host

  def _check_streamk_lost(self, host):
    seen = self._sk_lost[1]
    now = int(host[0])
    if now != seen:
      self._sk_lost[1] = now
      raise _lib.SpeechtHipError('{} hand-off(s) of the persistent per-bin products timed out (a producer workgroup never '
                                 'published its partial tile); the tiles concerned were set to NaN'.format(now - seen))

  def losses_precise(self):
    """float64 losses (hi + lo) straight from the device buffers, no status check (tests, bench parity)."""
    pair = self.loss_pair.cpu().numpy().astype(np.float64)
    B = self.loss.numel()
    return pair[:B] + pair[B:]

  def check_ctc_status(self):
    st = self.ctc_status.cpu().numpy()
    if st.any():
      raise ValueError('Not enough time for target transition sequence (utterances {})'.format(np.nonzero(st)[0].tolist()))
