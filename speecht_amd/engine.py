"""Device-side execution of the Wav2Letter step: buffers in HBM + the launch sequence.

torch is used for device memory, streams and (in data_parallel.py) torch.distributed only; every
arithmetic op of the path is a call into libspeecht_hip.so through ``_lib`` (no fallback).

HBM layout (DESIGN.md "Data layout"):
  * activations X[i] / gradients dZ[i]: padded NWC ``st_tensor3`` buffers, zero halos sized for
    the consuming convolution, channel pitch rounded to 16 floats (32 for wide tensors, ``channel_pitch``);
  * parameters, gradients, Adam m/v: four flat fp32 buffers with identical layout
    [F0 | b0 | F1 | b1 | ...], filters in the packed GEMM layout [k_pad][n_pad] -- so the
    gradient all-reduce and clip+Adam each see one contiguous buffer.
"""
import ctypes
import math
import os

import numpy as np
import torch

from . import _lib
from ._lib import Tensor3, call


def _round_up(a, b):
  return (a + b - 1) // b * b


def channel_pitch(channels):
  """Channel pitch of a padded NWC tensor: a multiple of 16 floats (the kernels' requirement); wide tensors
  round to 32 so that every 32-deep k-tile of the convolutions is whole (2000 -> 2016: the GEMM kernels then
  take their unclamped-address variant), narrow ones (80-mel input, 29 logits) keep the cheaper multiple of 16."""
  return _round_up(channels, 32 if channels > 128 else 16)


def same_padding(t_in, width, stride):
  """tf.nn.conv1d 'SAME' (speech_model.py:155): extra zero goes to the right."""
  t_out = -(-t_in // stride)
  pad_total = max((t_out - 1) * stride + width - t_in, 0)
  return t_out, pad_total // 2, pad_total - pad_total // 2


class DevTensor3:
  """A padded NWC view (st_tensor3 descriptor) over a slice of persistent device storage."""

  def __init__(self, storage, batch, frames, channels, halo_l, halo_r):
    self.batch, self.frames, self.channels = batch, frames, channels
    self.halo = halo_l
    self.c_pitch = channel_pitch(channels)
    self.t_pitch = halo_l + frames + halo_r
    self.buf = storage[:batch * self.t_pitch * self.c_pitch]
    self.desc = Tensor3(self.buf.data_ptr(), batch, frames, channels, halo_l, self.t_pitch, self.c_pitch)

  @staticmethod
  def numel(batch, frames, channels, halo_l, halo_r):
    return batch * (halo_l + frames + halo_r) * channel_pitch(channels)

  @property
  def ref(self):
    return ctypes.byref(self.desc)

  def interior(self):
    """[B, T, C] strided view of the valid region."""
    v = self.buf.view(self.batch, self.t_pitch, self.c_pitch)
    return v[:, self.halo:self.halo + self.frames, :self.channels]


class _Storage:
  """Grow-only named device buffers: real training batches change (B, max_T) every step, so the
  activation buffers are re-described per shape instead of re-allocated; only the halo rows have to
  be re-zeroed (interiors are fully overwritten by the producing kernel)."""

  def __init__(self, device):
    self.device = device
    self.bufs = {}
    self.generation = 0          # bumped on every (re)allocation: captured graphs hold the old pointers

  def view(self, name, numel, dtype=None):
    import torch as _t
    dtype = dtype or _t.float32
    cur = self.bufs.get(name)
    fresh = cur is None or cur.numel() < numel
    if fresh:
      cur = _t.zeros(max(numel, 1), dtype=dtype, device=self.device)
      self.bufs[name] = cur
      self.generation += 1
    return cur, fresh


class LayerSpec:
  def __init__(self, width, stride, cin, cout, relu):
    self.width, self.stride, self.cin, self.cout, self.relu = width, stride, cin, cout, relu
    self.cin_pitch = channel_pitch(cin)
    self.cout_pitch = channel_pitch(cout)
    kv, kp, npad = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    call('st_packed_dims', width, self.cin_pitch, cout, ctypes.byref(kv), ctypes.byref(kp), ctypes.byref(npad))
    self.k_valid, self.k_pad, self.n_pad = kv.value, kp.value, npad.value
    # transposed operand for back-prop to the input: [ru32(W*cout_pitch)][n_pad(cin)]
    call('st_packed_dims', width, self.cout_pitch, cin, ctypes.byref(kv), ctypes.byref(kp), ctypes.byref(npad))
    self.kt_pad, self.nt_pad = kp.value, npad.value


class _PendingDecode:
  """Decoder outputs on their way to pinned host memory (``Wav2LetterEngine.greedy_decode_async``).  The engine
  alternates between two host slots: read a handle before issuing the second decode after it."""

  def __init__(self, slot, batch, t_out):
    self._slot, self._batch, self._t_out = slot, batch, t_out

  def result(self):
    ids_host, lens_host, event = self._slot
    event.synchronize()
    lens = lens_host[:self._batch].numpy()
    ids = ids_host[:self._batch * self._t_out].numpy().reshape(self._batch, self._t_out)
    return [ids[b, :lens[b]].tolist() for b in range(self._batch)]


class _PendingBeamDecode:
  """Prefix-beam-search outputs on their way to pinned host memory (``Wav2LetterEngine.beam_search_decode_async``):
  ``result()`` waits for that batch only and returns (list of id lists, log_prob [B, 1])."""

  def __init__(self, slot, batch, t_out):
    self._slot, self._batch, self._t_out = slot, batch, t_out
    self._generation = slot['generation']

  def result(self):
    s = self._slot
    if s['generation'] != self._generation:
      # the slots form a ring of (decoder streams + 1): this handle's pinned buffers and event now belong to a later batch
      raise RuntimeError('beam-search handle read too late: {} further beam_search_decode_async call(s) re-used its slot; read a '
                         'handle before issuing more than len(decode streams) further calls'.format(s['generation'] - self._generation))
    s['event'].synchronize()
    lens = s['lens_h'][:self._batch].numpy()
    ids = s['ids_h'][:self._batch * self._t_out].numpy().reshape(self._batch, self._t_out)
    return ([ids[b, :lens[b]].tolist() for b in range(self._batch)],
            s['score_h'][:self._batch].numpy().reshape(-1, 1).copy())


def beam_input_transform(name):
  """The `input_transform` code of st_ctc_beam_search_decode_ex: None / 'logits' -> 0, 'log10_softmax' -> 1 (the reference's
  decoder input, tf.log(tf.nn.softmax(logits) + 1e-8) / log(10), speech_model.py:102)."""
  if name in (None, 'logits', 0):
    return 0
  if name in ('log10_softmax', 1):
    return 1
  raise ValueError("input_transform must be None, 'logits' or 'log10_softmax', got {!r}".format(name))


def merge_repeated_labels(seq):
  """tf.nn.ctc_beam_search_decoder(merge_repeated=True) on an output prefix: consecutive equal labels collapse (TF's LabelSeq walk;
  it also collapses genuine double letters, which is why the reference passes False, speech_model.py:110)."""
  return [v for i, v in enumerate(seq) if i == 0 or v != seq[i - 1]]


def decoder_streams(device, decoders=2):
  """(compute stream, [decoder streams]) for overlapping a one-wave-per-utterance decoder with the NEXT batches' forward passes.

  fp32 MFMAs execute on the VALU datapath: a VALU / LDS chain that shares its SIMD with the waves of an fp32 GEMM gets an
  issue slot every ~25 cycles instead of every ~4 (measured round 4: the CTC recursion under a GEMM 122 -> 560 us, the beam
  search beside the next forward pass 3.9 -> 6.5 ms per batch).  So the two run on DISJOINT compute units: streams created
  with hipExtStreamCreateWithCUMask, the decoders on 16 CUs (mask bits 0..15 -- observed on MI355X: two CUs of every XCD; any
  other layout tried costs the GEMMs 10-70 %), the forward pass on the other 240 (+11 % on the forward pass alone, 3.6 -> 4.0
  ms at configs[4]).  A search is one wavefront per utterance -- 16 of the decoder CUs' 64 SIMDs at configs[4] -- and takes a
  little longer than the forward pass: with ONE decoder stream the search sets the pace (4.18 ms per batch against 7.4 serial),
  with two the searches of consecutive batches run side by side on the same 16 CUs and the forward pass does (4.02 ms).
  Placement is a matter of speed only.  Falls back to plain streams where the runtime lacks the call."""
  import ctypes as C
  import glob
  key = (str(device), int(decoders))
  if key in _DECODER_STREAMS:
    return _DECODER_STREAMS[key]
  made = None
  if os.environ.get('ST_DECODER_CU_MASK', '1') != '0':
    try:
      hip = C.CDLL(glob.glob(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64*'))[0])
      made = []
      with torch.cuda.device(device):
        for words in [[0xffff0000] + [0xffffffff] * 7] + [[0x0000ffff] + [0] * 7] * int(decoders):
          arr = (C.c_uint32 * 8)(*words)
          handle = C.c_void_p()
          if hip.hipExtStreamCreateWithCUMask(C.byref(handle), 8, arr) != 0 or not handle.value:
            raise OSError('hipExtStreamCreateWithCUMask failed')
          made.append(torch.cuda.ExternalStream(handle.value, device=device))
    except (OSError, IndexError, AttributeError):
      made = None
  if made is None:
    made = [torch.cuda.Stream(device) for _ in range(1 + int(decoders))]
  _DECODER_STREAMS[key] = (made[0], made[1:])
  return _DECODER_STREAMS[key]


def decoder_stream_pair(device):
  """(compute stream, decoder stream): `decoder_streams` with one decoder."""
  compute, decoders = decoder_streams(device, 1)
  return compute, decoders[0]


_DECODER_STREAMS = {}
_ROLE_STREAMS = {}


def role_stream(device, role):
  """The process-wide stream of a role ('h2d', 'upload', 'side', 'side2', 'collective') on a device.

  Which HARDWARE queue a HIP stream lands on is decided when it is first used, round-robin over the runtime's four queues,
  and two streams on one queue do not overlap: the queue serialises them.  With a stream per engine the same engine ran
  its step at different speeds depending on how many streams the process had used before it was built (measured round 4:
  the bf16x6 step 6.4 ms in a process of its own, 7.0 ms as the bench's side measurement behind an fp32 engine and a
  collective stream; the bf16 step 2.66 / 2.86 ms; GPU_MAX_HW_QUEUES=8 instead: fp32 step 7.0 -> 9.2 ms).  So the streams
  of all roles are created -- and used once, in a fixed order -- when the first of them is asked for, every engine of the
  process shares them (ordering between engines is by the events each engine records anyway), and the role -> queue map is
  the same in every process: compute stream q0, h2d q1, side q2, side2 q3, upload q0, collective q1 -- the two side streams
  on queues of their own (with side2 on the compute stream's queue, as creation order had it before: bf16 step 2.68 instead
  of 2.57 ms, fp32 and bf16x6 within 0.5 %; DESIGN 4.8 item 5 has the orders that were measured)."""
  key = str(device)
  pool = _ROLE_STREAMS.get(key)
  if pool is None:
    pool = {}
    with torch.cuda.device(device):
      torch.zeros(1, device=device)                       # the compute (current) stream has its queue first
      for r in ('h2d', 'side', 'side2', 'upload', 'collective'):
        pool[r] = torch.cuda.Stream(device)
        with torch.cuda.stream(pool[r]):
          torch.zeros(1, device=device)
      torch.cuda.synchronize(device)
    _ROLE_STREAMS[key] = pool
  if role not in pool:                                   # (a role the order left out)
    pool[role] = torch.cuda.Stream(device)
  return pool[role]


class _StagedHostBatch:
  """One of the engine's two H2D staging buffers: ``event`` = copy finished, ``consumed`` = the compute stream has
  read it (``Wav2LetterEngine.stage_host_batch`` / ``load_batch``)."""

  def __init__(self, tensor):
    self.tensor = tensor
    self.event = torch.cuda.Event()
    self.consumed = torch.cuda.Event()
    self.consumed.record()
    self.taken = True             # handed to load_batch (host-side state; `consumed` is the device-side one)


class Wav2LetterEngine:
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
    # lost stream-K hand-offs are counted per process by the library; this engine reports the ones after its creation
    seen = ctypes.c_uint32(0)
    call('st_streamk_lost_count', ctypes.byref(seen))
    self._sk_lost = [torch.zeros(1, dtype=torch.int32, pin_memory=True), int(seen.value)]

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
    if not fresh:
      if clear:
        t.buf.zero_()
      else:
        call('st_zero_halos_f32', t.ref, self.stream_ptr)
    return t

  def _ensure_shape(self, batch, frames):
    if self._shape == (batch, frames):
      return
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
    lib = _lib.load()
    ws = max(lib.st_conv1d_bwd_filter_ws(self.X[i].ref, self.dZ[i].ref, l.width) for i, l in enumerate(self.layers))
    ws = max([ws] + [lib.st_conv1d_bwd_data_bias_ws(self.dZ[i].ref, self.dZ[i - 1].ref, l.width)
                     for i, l in enumerate(self.layers) if i > 0])
    ws = max([ws] + [lib.st_conv1d_fwd_ws(self.X[i].ref, self.X[i + 1].ref, l.width) for i, l in enumerate(self.layers)])
    if self.conv_mode == 'bf16x6':
      ws = max([ws] + [lib.st_exp_conv1d_bwd_data_bf16x6_ws(self.dZ[i].ref, self.dZ[i - 1].ref, l.width)
                       for i, l in enumerate(self.layers) if i > 0])
    self.wgrad_ws, _ = self._storage.view('wgrad_ws', ws // 4 + 64)
    # The classification layer (2000 -> 29): its filter gradient streams the activations, its back-prop to the input streams
    # the mask and writes dZ of the layer below -- two HBM-bound launches of ~90 us each that do not depend on each other.
    # Side by side on two streams (own scratch for the one on the side stream).
    top = len(self.layers) - 1
    self._side_wgrad_top = self.side_filter_gradient and os.environ.get('ST_WGRAD_SIDE_TOP', '1') != '0' and top > 0 and self.layers[top].cout <= 64 and self.layers[top].width == 1
    if self._side_wgrad_top:
      ws_top = lib.st_conv1d_bwd_filter_ws(self.X[top].ref, self.dZ[top].ref, self.layers[top].width)
      self.wgrad_ws_top, _ = self._storage.view('wgrad_ws_top', ws_top // 4 + 64)
    # per-utterance CTC losses as (hi, lo) float pairs: `loss` is the fp32 value (what tf.nn.ctc_loss returns), `loss_lo` what
    # fp32 cannot hold of -log p at that magnitude (st_ctc_loss_grad_hilo_f32); one buffer, so one copy brings both back
    loss_buf = self._storage.view('loss', 2 * batch)[0]
    self.loss_pair = loss_buf[:2 * batch]
    self.loss, self.loss_lo = loss_buf[:batch], loss_buf[batch:2 * batch]
    self.ctc_status = self._storage.view('ctc_status', batch, torch.int32)[0][:batch]
    self.dec_ids = self._storage.view('dec_ids', batch * self.t_out, torch.int32)[0][:batch * self.t_out]
    self.dec_lens = self._storage.view('dec_lens', batch, torch.int32)[0][:batch]
    self.dec_score = self._storage.view('dec_score', batch)[0][:batch]
    # which layers run in the frequency domain is decided first: they leave the bf16x6 plane plumbing alone
    self._fft_layers = {i for i in range(len(self.layers)) if self._use_fft(i, batch, geo[i][1])}
    # bf16 activations: the wide long-filter layer in the frequency domain with its per-bin products on the bf16 matrix pipe
    self._fftb_layers = {i for i in range(len(self.layers)) if self._use_fft_bf16(i, batch, geo[i][1])}
    if self.conv_mode == 'bf16x6':
      self._alloc_planes()
    if self.conv_mode == 'bf16':
      self._alloc_bf16()
      self._alloc_fft_bf16(batch)
    self._alloc_fft(batch)
    self._shape = (batch, frames)

  # ---- frequency-domain layers (csrc/conv_fft.hip) ---------------------------------------------------
  def _polyphase(self, i):
    """A stride-2 layer as a stride-1 layer on the polyphase view of its input: x read as [B][T/2][2 * c_pitch] (frame
    pairs as channels), y[t] = sum_w F[w] x[2t + w - pl] = sum_{j,p} F[2j + p - shift] X2[t + j - pl2][p] with
    pl2 = ceil(pl / 2), shift = 2 pl2 - pl: width2 = ceil((W + shift) / 2) taps whose packed filters are the layer's
    own rows moved down by `shift` channel blocks (zeros around them).  Returns (width2, pl2, shift) or None."""
    l = self.layers[i]
    if l.stride != 2:
      return None
    pl = self.geo[i][2]
    pl2 = (pl + 1) // 2
    shift = 2 * pl2 - pl
    return (l.width + shift + 1) // 2, pl2, shift

  def _use_fft(self, i, batch, t_out):
    l = self.layers[i]
    wide = l.stride == 1 and l.width >= 16
    if not (self.fft_conv and self.conv_mode in ('fp32', 'bf16x6') and l.n_pad % 128 == 0 and
            batch * t_out >= (self.fft_min_rows if wide else self.fft_min_rows_narrow)):
      return False
    if l.stride == 2:        # first layer of the model (48 taps, stride 2): 25 polyphase taps over 2 x 80 channels
      width2 = self._polyphase(i)[0]
      return i == 0 and self.fft_first_layer and self.fft_min_width <= width2 <= 33
    return i > 0 and l.stride == 1 and self.fft_min_width <= l.width <= 33 and l.nt_pad % 128 == 0

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
      self.__dict__.setdefault('_fftb_table_key', {})[i] = (l.width, pl)
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

  def _alloc_fft(self, batch):
    """Per frequency-domain layer: the transform tables and the filter spectra in both operand layouts (functions of
    the layer only: kept across shapes), the input / gradient spectra and one scratch area (sized by the shape)."""
    lib = _lib.load()
    self.fft = {}
    for i, l in enumerate(self.layers):
      t_in, t_out, pl, pr = self.geo[i]
      if i not in self._fft_layers:
        continue
      view = lambda name, numel: self._storage.view('fft%d_%s' % (i, name), numel)
      f = dict(x=self.X[i].desc, width=l.width, pl=pl, cin=l.cin, cin_pitch=l.cin_pitch, shift=None)
      if l.stride == 2:
        width2, pl2, shift = self._polyphase(i)
        x = self.X[i]
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
          self._gfwd_fresh = False
      f['xref'] = ctypes.byref(f['x'])
      tables, fresh_tables = view('tables', lib.st_conv1d_fft_table_floats())
      # the tables are functions of (taps, left padding): a new shape or another model may change either for the same
      # layer index, so the pair is kept with them
      if getattr(self, '_fft_table_key', {}).get(i) != (f['width'], f['pl']):
        fresh_tables = True
      self.__dict__.setdefault('_fft_table_key', {})[i] = (f['width'], f['pl'])
      # ONE set of filter spectra: back-prop to the input reads it as a transposed operand (csrc/conv_fft.hip)
      gfwd, fresh_f = view('gfwd', lib.st_conv1d_fft_filter_floats(f['width'], f['cin_pitch'], l.cout))
      f.update(tables=tables, gfwd=gfwd,
               sf=view('sf', lib.st_conv1d_fft_sf_floats(f['xref'], self.X[i + 1].ref, f['width']))[0],
               zf=view('zf', lib.st_conv1d_fft_zf_floats(self.dZ[i].ref, f['width']))[0],
               ws=view('ws', lib.st_conv1d_fft_ws(f['xref'], self.X[i + 1].ref, f['width']) // 4 + 64)[0])
      # (the wide 32-tap layer stays on one stream: its chain side by side, or only its HBM-bound inverse transform of the
      # lag products beside back-prop's products, both measured slower: 7.37 -> 7.43 ms)
      if self.side_filter_gradient and i > 0 and l.cout <= 512:
        f['ws2'] = view('ws2', lib.st_conv1d_fft_ws(f['xref'], self.X[i + 1].ref, f['width']) // 4 + 64)[0]
      if fresh_tables:
        call('st_conv1d_fft_tables_f32', f['width'], f['pl'], self._ptr(tables), tables.numel(), self.stream_ptr)
      if fresh_f:
        self._gfwd_fresh = False
      self.fft[i] = f
    if set(self.fft) != getattr(self, '_fft_prev', None):     # a layer (re)joined the path: its spectra may be stale
      self._gfwd_fresh = False
      self._packed_t_fresh = False                             # (and a layer that left it needs its flipped copy again)
    self._fft_prev = set(self.fft)

  def _refresh_fft_filters(self, layers=None):
    """Filter spectra of the frequency-domain layers (all, or the given ones) from the current weights, in layer
    order; on a side stream an event is recorded after each layer so that the forward pass waits for the layer it is
    about to run, not for all.  (Back-prop to the input reads the same spectra, transposed.)"""
    stream = self._stream if self._stream is not None else torch.cuda.current_stream(self.device)
    if layers is None:
      self._gfwd_ready = {}
    for i, f in self.fft.items():
      if layers is not None and i not in layers:
        continue
      l = self.layers[i]
      pf = self._slice(self.params, i)[0]
      if f['shift'] is not None:
        cp = self.X[i].c_pitch
        n = l.width * cp * l.n_pad
        with torch.cuda.stream(stream):
          f['packed2'][f['shift'] * cp * l.n_pad:f['shift'] * cp * l.n_pad + n].copy_(pf[:n], non_blocking=True)
        pf = f['packed2']
      call('st_conv1d_fft_filters_f32', self._ptr(pf), f['width'], f['cin'], l.cout, f['cin_pitch'], self._ptr(f['tables']),
           self._ptr(f['gfwd']), self.stream_ptr)
      if stream is getattr(self, '_side', None):
        ev = torch.cuda.Event()
        ev.record(stream)
        self._gfwd_ready[i] = ev
    self._gfwd_fresh = True

  def _refresh_gfwd(self):
    """After an update: the bottom layer's spectra on the compute stream (the next step needs them at once; a
    cross-stream wait there costs more than the 25 us of work), the others on the side stream, bottom layer first."""
    if self.fft and self._shape is not None:
      first = min(self.fft)
      self._gfwd_ready = {}
      self._refresh_fft_filters(layers=[first])
      rest = [i for i in self.fft if i != first]
      if rest:
        self._on_side_stream(lambda: self._refresh_fft_filters(layers=rest))

  def _wait_gfwd(self, i=None):
    """The compute stream waits for the forward filter spectra of layer i (None: of every layer) if they were rebuilt on
    the side stream after the update.  The side stream works bottom layer first: the first three frequency-domain layers
    wait for their own spectra, the fourth for all that remain (by then the side stream is through, and every wait
    costs the compute stream a few microseconds)."""
    ready = getattr(self, '_gfwd_ready', None)
    if not ready:
      return
    order = sorted(self.fft)
    if i is not None and i in order and order.index(i) >= 3:
      i = None
    keys = [k for k in ready if i is None or k <= i]
    if keys:
      (self._stream if self._stream is not None else torch.cuda.current_stream(self.device)).wait_event(ready[max(keys)])
      for k in keys:
        del ready[k]

  # ---- bf16 activations (config 4) ------------------------------------------------------------------
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

  # ---- bf16x6 (experimental) ------------------------------------------------------------------
  def _in_fft(self, i):
    return self.fft_conv and i in getattr(self, '_fft_layers', ())

  def _x6_fwd(self, i):
    return self.conv_mode == 'bf16x6' and self.layers[i].n_pad % 128 == 0 and not self._in_fft(i)

  def _x6_bwd(self, i):
    l = self.layers[i]
    return (self.conv_mode == 'bf16x6' and i > 0 and l.nt_pad % 128 == 0 and l.width * l.cout_pitch >= 256 and
            not self._in_fft(i))

  def _x6_wgrad(self, i):
    l = self.layers[i]
    tiles = -(-(l.width * l.cin_pitch) // 128) * (l.n_pad // 128)
    return (self.conv_mode == 'bf16x6' and i > 0 and l.stride == 1 and l.n_pad % 128 == 0 and tiles >= 192 and
            not self._in_fft(i))

  def _planes(self, name, numel, n=3, slack=0):
    """n zeroed bf16 planes of `numel` elements each (whole buffer cleared when re-used).  ``slack``: that many further zero
    elements stay allocated behind the (single) plane -- readable zeros for kernels that run past the last row
    (st_conv1d_nwc_bwd_filter_tr_bf16); the returned view does not include them."""
    buf, fresh = self._storage.view(name, n * numel + slack, torch.bfloat16)
    v = buf[:n * numel + slack]
    if not fresh:
      v.zero_()
    return v[:n * numel]

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
    self.ctc_lens = self._upload_i32((self.seq_lens_host // 2).astype(np.int32), fixed='ctc_lens')

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

  def _upload_i32(self, values, fixed=None):
    """Small int32 host array -> device through a ring of pinned slots.  A hipMemcpyAsync from pageable memory
    only returns once the stream's earlier kernels have finished, which would stall the thread that enqueues
    the next batch behind the previous batch's forward; from pinned memory the copy is a stream operation.
    ``fixed``: with whole-step graphs on (`enable_step_graph`) the array goes to a persistent device buffer of that name --
    one per step parity, so that the upload of step k + 1 does not wait for step k's kernels -- whose address a captured
    launch can hold."""
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
      if fixed is not None and getattr(self, '_step_graph_on', False):
        par = self._step_parity
        dev = self._storage.view('%s_par%d' % (fixed, par), _round_up(max(n, 1), 4096), torch.int32)[0][:n]
        if fixed in self._parity_written[par]:
          # written before and not consumed by a graph step since (eager passes in between: evaluation, a decode): whatever is
          # enqueued on the compute stream so far may still read the buffer
          busy = torch.cuda.Event()
          busy.record(main)
          self._up_stream.wait_event(busy)
        elif self._parity_consumed[par] is not None:
          self._up_stream.wait_event(self._parity_consumed[par])      # the step that last read this parity's buffers is through
        self._parity_written[par].add(fixed)
      else:
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
    """X[0] -> logits X[-1] through the eleven layers (speech_model.py:279-295): per layer the frequency-domain entry
    point, the bf16x6 kernel or the W-tap kernel, as decided per shape by `_use_fft` / `_x6_fwd`."""
    if self.conv_mode == 'bf16':
      return self._forward_bf16()
    s = self.stream_ptr
    x6 = self.conv_mode == 'bf16x6'
    if x6 and not self._wplanes_fresh:
      self._refresh_wplanes()
    if x6 and self._x6_fwd(0):
      call('st_exp_split3_bf16', self._ptr(self.X[0].buf), self.X[0].buf.numel(), self._ptr(self.Xp[0]), s)
    sf_ready = False                 # the previous layer's call left this layer's input spectra behind
    for i, l in enumerate(self.layers):
      pf, pb = self._slice(self.params, i)
      if not (i in self.fft and self.fft_conv):
        sf_ready = False
      if x6 and self._x6_fwd(i):
        if i > 0 and not self._x6_fwd(i - 1):
          call('st_exp_split3_bf16', self._ptr(self.X[i].buf), self.X[i].buf.numel(), self._ptr(self.Xp[i]), s)
        yp = self._ptr(self.Xp[i + 1]) if (i + 1 < len(self.layers) and self._x6_fwd(i + 1)) else None
        call('st_exp_conv1d_fwd_bf16x6', self.X[i].ref, self._ptr(self.Xp[i]), self._ptr(self.Wp[i]), self._ptr(pb),
             l.width, l.stride, self.geo[i][2], int(l.relu), self.X[i + 1].ref, yp, s)
      elif i in self.fft and self.fft_conv:
        f = self.fft[i]
        if not self._gfwd_fresh:
          self._join_side_stream()
          self._refresh_fft_filters()
        self._wait_gfwd(i)                               # the filter spectra may still be on their way (side stream)
        # a chain of frequency-domain layers: where the shapes allow, this layer's inverse transform hands its frames to the
        # next layer's forward transform in registers and leaves that layer's input spectra behind (`sf_ready` for its call)
        nxt = self.fft.get(i + 1) if self.fft_conv else None
        if nxt is not None and nxt['shift'] is not None:
          nxt = None
        written = ctypes.c_int(0)
        call('st_conv1d_nwc_fwd_fft_chain_f32', f['xref'], self._ptr(f['gfwd']), self._ptr(pb), f['width'], f['pl'],
             int(l.relu), self.X[i + 1].ref, self._ptr(f['tables']), self._ptr(f['sf']), int(sf_ready),
             self._ptr(nxt['tables']) if nxt else None, self._ptr(nxt['sf']) if nxt else None, nxt['width'] if nxt else 0,
             nxt['pl'] if nxt else 0, ctypes.byref(written), self._ptr(f['ws']), f['ws'].numel() * 4, s)
        sf_ready = written.value == 1
        continue
      else:
        call('st_conv1d_nwc_fwd_ws_f32', self.X[i].ref, self._ptr(pf), self._ptr(pb), l.width, l.stride,
             self.geo[i][2], int(l.relu), self.X[i + 1].ref, self._ptr(self.wgrad_ws),
             self.wgrad_ws.numel() * 4 if self.split_small_batches else 0, s)

  def forward_graph(self):
    """``forward()`` replayed from a HIP graph: the launch sequence of the current (batch, frames) shape is
    captured once and replayed with a single launch afterwards -- for small batches (live / single-utterance
    inference) the eleven kernels are launch-bound and the CPU cost of enqueueing them is the latency.
    Buffers, weights and the batch are read through the same device pointers on replay, so new inputs
    (``load_batch`` with the same shape) and in-place weight updates are picked up; a new shape captures anew."""
    if not hasattr(self, '_graphs'):
      self._graphs, self._graph_seen = {}, set()
    if self.conv_mode == 'bf16' and not self._wplanes_fresh:
      self._refresh_bf16_filters(False)            # derived operands are rebuilt outside the graph
    if self.conv_mode == 'bf16x6' and not self._wplanes_fresh:
      self._refresh_wplanes()
    self._join_side_stream()
    if getattr(self, '_wb_ready', None):
      self._wb_ready.clear()                       # (covered by the join above)
    if self.fft and self.fft_conv and not self._gfwd_fresh:
      self._refresh_fft_filters()
    self._wait_gfwd()                              # no waits on outside events inside a capture
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
    self.label_ids = self._upload_i32(ids, fixed='label_ids')
    self.label_offs = self._upload_i32(offs, fixed='label_offs')

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
    # the filter operands of back-prop (flipped / transposed copies of the weights Adam just updated) are rebuilt
    # on the side stream while the CTC recursion runs
    if self.conv_mode == 'bf16':
      if not self._wtplanes_fresh and hasattr(self, 'WTb'):
        self._on_side_stream(lambda: self._refresh_bf16_filters(True))
    elif not self._packed_t_ok() and self._flip_layers():
      self._on_side_stream(self._refresh_backward_operands)
    self._wait_uploads()
    call('st_ctc_loss_grad_hilo_f32', self.X[-1].ref, self._ptr(self.label_ids), self._ptr(self.label_offs),
         self._ptr(self.ctc_lens), self.max_label_len, float(grad_scale), self._ptr(self.loss), self._ptr(self.loss_lo), self.dZ[-1].ref,
         self._ptr(self.ctc_status), self._ptr(self.ctc_ws), self.ctc_ws.numel() * 4, self.stream_ptr)
    if getattr(self, '_rejected_labels', None):            # labels refused on the host (deferred mode): status 2
      stream = self._stream if self._stream is not None else torch.cuda.current_stream(self.device)
      with torch.cuda.stream(stream):
        self.ctc_status.index_fill_(0, torch.as_tensor(self._rejected_labels, dtype=torch.int64).to(self.device, non_blocking=True), 2)
    call('st_ctc_status_gate_f32', self._ptr(self.ctc_status), B, self._ptr(self.gate), self.stream_ptr)

  def _transposed_in_place(self, i):
    """Back-prop to the input of layer i reads the layer's own packed filters as a transposed operand
    (st_conv1d_1tap_bwd_data_bias_f32): one tap, whole 32-deep k-tiles over the output channels."""
    l = self.layers[i]
    return (self.conv_mode == 'fp32' and i > 0 and l.width == 1 and l.stride == 1 and l.cout_pitch % 32 == 0 and
            l.n_pad >= l.cout_pitch and l.nt_pad % 128 == 0)

  def _flip_layers(self):
    """Layers whose back-prop to the input still needs the flipped / transposed copy of the weights: W-tap layers of
    more than one tap (and everything on the bf16x6 path, whose operand planes are split from those copies).  The
    frequency-domain layers read their forward spectra transposed, 1-tap layers their packed filters (round 4): at the
    model's training shapes NO copy is rebuilt any more (rounds 1-3: ~0.33 ms of HBM-bound launches per step)."""
    if self.conv_mode == 'bf16':
      return []
    return [i for i in range(1, len(self.layers))
            if self.conv_mode == 'bf16x6' or not ((i in self.fft and self.fft_conv) or self._transposed_in_place(i))]

  def _refresh_backward_operands(self):
    """The flipped / transposed weight copies of the layers that still need one (`_flip_layers`), top layer first, the order
    back-prop consumes them in; an event after each lets the compute stream wait for what it is about to use only."""
    s = self.stream_ptr
    stream = self._stream if self._stream is not None else torch.cuda.current_stream(self.device)
    self._bwd_ready = {}
    for i in reversed(self._flip_layers()):
      l = self.layers[i]
      call('st_filters_flip_transpose_f32', self._ptr(self._slice(self.params, i)[0]), l.width, l.cin, l.cout, l.cin_pitch,
           l.cout_pitch, self._ptr(self.packed_t[i]), s)
      ev = torch.cuda.Event()
      ev.record(stream)
      self._bwd_ready[i] = ev
    self._packed_t_fresh = True
    self._packed_t_layers = frozenset(self._flip_layers())
    self._wtplanes_fresh = False

  def _packed_t_ok(self):
    """The flipped / transposed copies are current for every layer that needs one NOW: which layers do depends on state that
    can change between steps (the shape's frequency-domain set, `fft_conv`), so a refresh remembers the set it rebuilt."""
    return self._packed_t_fresh and frozenset(self._flip_layers()) <= getattr(self, '_packed_t_layers', frozenset())

  def _wait_bwd_operands(self, i=None):
    """The compute stream waits for the back-prop operands of layer i (None: of every layer) if they were rebuilt on the
    side stream.  The side stream works top layer first, so a lower layer's event covers the ones above it."""
    ready = getattr(self, '_bwd_ready', None)
    if not ready:
      return
    keys = [k for k in ready if i is None or k >= i]
    if keys:
      (self._stream if self._stream is not None else torch.cuda.current_stream(self.device)).wait_event(ready[min(keys)])
      for k in keys:
        del ready[k]

  def backward(self, on_layer_done=None, hook_layers=None):
    """Back-prop from dZ[-1] (already holding d avg_loss / d logits).  ``on_layer_done(i)`` is
    called after layer i's filter/bias gradients have been enqueued (for bucketed all-reduce) -- for every layer, or only
    for those in ``hook_layers`` (the layers that complete a reduce bucket, `GradientAllReducer.hook_layers`): a hook on a
    layer whose filter gradient runs on the side stream makes the compute stream WAIT for that stream first, so hooks
    nobody needs cost the overlap of the two chains (round 4: a forced world-1 exchange cost the step 0.24 ms, most of it
    six such waits)."""
    s = self.stream_ptr
    wanted = (lambda i: True) if hook_layers is None else (lambda i: i in hook_layers)
    if self.conv_mode == 'bf16':
      self._join_side_stream()
      return self._backward_bf16(on_layer_done, wanted)
    if not self._packed_t_ok():
      self._refresh_backward_operands()           # (normally done on the side stream by ctc_loss_grad; nothing at the model's shapes)
    if self.fft and self.fft_conv and not self._gfwd_fresh:
      self._join_side_stream()                    # (weights written after the forward pass: back-prop reads the same spectra)
      self._refresh_fft_filters()
    self._wait_gfwd()
    if self.conv_mode == 'bf16x6':
      self._wait_bwd_operands()                   # the split planes are derived from all transposed copies at once
    # waits per layer; after the two layers on top one wait covers everything below (by then the side stream is through)
    wait_all_below = len(self.layers) - 3
    side_wgrad, deferred = False, None     # a filter gradient is in flight on the side stream; its layer's hook is due
    top_pending = None                     # the classification layer's filter gradient is in flight on the second side stream
    hook = on_layer_done
    if hook is not None:
      def on_layer_done(j):
        nonlocal top_pending
        if top_pending is not None and top_pending != j:
          self._join_side_stream(second_only=True)
          hook(top_pending)
          top_pending = None
        if top_pending != j:
          hook(j)
    bias_from_above = False      # layer i's bias gradient already written by the back-prop kernel of layer i + 1
    zf_ready = False             # layer i's dz spectra already written by the back-prop kernel of layer i + 1
    for i in reversed(range(len(self.layers))):
      l = self.layers[i]
      gf, gb = self._slice(self.grads, i)
      need_bias, bias_from_above = not bias_from_above, False
      if self.conv_mode == 'bf16x6' and self._x6_wgrad(i):
        tq, red = self.tq[i], self.X[i].batch * self.tq[i]
        call('st_exp_transpose_split3_bf16', self.X[i].ref, 0, self.X[i].t_pitch, tq, l.cin_pitch * red + 4096,
             self._ptr(self.XTp[i]), s)
        call('st_exp_transpose_split3_bf16', self.dZ[i].ref, self.dZ[i].halo, self.dZ[i].frames, tq, l.n_pad * red,
             self._ptr(self.dZTp[i]), s)
        call('st_exp_conv1d_bwd_filter_bf16x6', self._ptr(self.XTp[i]), self._ptr(self.dZTp[i]), self.X[i].batch, tq,
             l.width, l.cin_pitch, self.X[i].halo - self.geo[i][2], l.cout, self._ptr(gf), s)
        if need_bias:
          call('st_bias_grad_f32', self.dZ[i].ref, self._ptr(gb), self._ptr(self.wgrad_ws), self.wgrad_ws.numel() * 4, s)
      elif i in self.fft and self.fft_conv:
        f = self.fft[i]
        # the spectra of dz serve the filter gradient here and back-prop to the input below (the layer above may have left
        # them behind already: its back-prop kernel transformed the frames it had just produced, `zf_ready`)
        if not zf_ready:
          call('st_conv1d_fft_dz_spectra_f32', self.dZ[i].ref, f['width'], self._ptr(f['tables']), self._ptr(f['zf']), s)
        zf_ready = False
        polyphase = f['shift'] is not None       # the gradient comes out in the shifted layout of the polyphase taps

        def filter_gradient(f=f, l=l, i=i, gf=gf, gb=gb, need_bias=need_bias, polyphase=polyphase, ws=f.get('ws2', f['ws'])):
          call('st_conv1d_nwc_bwd_filter_fft_f32', f['xref'], self.dZ[i].ref, self._ptr(f['sf']), self._ptr(f['zf']), f['width'],
               self._ptr(f['tables']), self._ptr(f['dpacked2'] if polyphase else gf), self._ptr(ws), ws.numel() * 4, self.stream_ptr)
          if polyphase:
            cp = self.X[i].c_pitch
            n, o = l.width * cp * l.n_pad, f['shift'] * cp * l.n_pad
            with torch.cuda.stream(self._stream if self._stream is not None else torch.cuda.current_stream(self.device)):
              gf[:n].copy_(f['dpacked2'][o:o + n], non_blocking=True)
          if need_bias:      # bin 0 of the spectra is the sum over the frames
            call('st_conv1d_fft_bias_grad_f32', self.dZ[i].ref, f['width'], self._ptr(f['zf']), self._ptr(gb), self.stream_ptr)
        if 'ws2' in f:
          # The filter gradient (lag products, inverse transform of the filters, bias sum) and back-prop to the input
          # (products, inverse transform) both hang off the spectra of dz and are independent: on two streams.  The
          # narrow layers' products (36 bins x 16 tiles) leave a quarter of the CU slots of their last round empty --
          # side by side they fill each other's gaps -- and the HBM-bound transforms of one chain run under the
          # matrix-pipe-bound products of the other (measured: 7.84 -> 7.43 ms per step).
          # (round 4: with back-prop's transforms fused the compute stream needs ~80 us per narrow layer, ONE side stream's
          # chain -- 73 + 42 + 9 us, in order -- had become the pace of the backward pass: the chains take the two side streams in turn)
          self._on_side_stream(filter_gradient, second=(i % 2 == 1))
          side_wgrad, deferred = True, i
        else:
          filter_gradient()
      elif i + 1 == len(self.layers) and self._side_wgrad_top and self.conv_mode == 'fp32':
        def top_gradient(i=i, l=l, gf=gf, gb=gb, need_bias=need_bias, ws=self.wgrad_ws_top):
          call('st_conv1d_nwc_bwd_filter_f32', self.X[i].ref, self.dZ[i].ref, l.width, l.stride, self.geo[i][2],
               self._ptr(gf), self._ptr(gb) if need_bias else None, self._ptr(ws), ws.numel() * 4, self.stream_ptr)
        # on the SECOND side stream (the first is still rebuilding back-prop operands when CTC ends); its hook is due with
        # the next layer's -- the two share a reduce bucket, and nothing waits for this launch until then
        self._on_side_stream(top_gradient, second=True)
        top_pending = i
      else:
        call('st_conv1d_nwc_bwd_filter_f32', self.X[i].ref, self.dZ[i].ref, l.width, l.stride, self.geo[i][2],
             self._ptr(gf), self._ptr(gb) if need_bias else None, self._ptr(self.wgrad_ws), self.wgrad_ws.numel() * 4, s)
      if on_layer_done is not None and deferred != i and wanted(i):
        if side_wgrad:
          # this layer's own gradient ran on the compute stream, but its bucket also holds the layers above whose filter
          # gradients are still in flight on the side streams (the bottom bucket L0..L3: L1-L3 went there, L0 did not): the
          # exchange is ordered behind the compute stream only, so the compute stream waits for them first
          self._join_side_stream()
          side_wgrad = False
        on_layer_done(i)
      if i > 0 and self._x6_bwd(i):
        act = self.X[i].ref if self.layers[i - 1].relu else None
        if not self._wtplanes_fresh:
          self._refresh_wtplanes()
        if not (i + 1 < len(self.layers) and self._x6_bwd(i + 1)):       # producer was not on this path
          call('st_exp_split3_bf16', self._ptr(self.dZ[i].buf), self.dZ[i].buf.numel(), self._ptr(self.dZp[i]), s)
        dxp = self._ptr(self.dZp[i - 1]) if self._x6_bwd(i - 1) else None
        call('st_exp_conv1d_bwd_data_bf16x6', self.dZ[i].ref, self._ptr(self.dZp[i]), self._ptr(self.WTp[i]), l.width,
             self.geo[i][2], act, self.dZ[i - 1].ref, dxp, self._ptr(self.wgrad_ws), self.wgrad_ws.numel() * 4, s)
      elif i > 0 and i in self.fft and self.fft_conv:
        f = self.fft[i]
        act = self.X[i].ref if self.layers[i - 1].relu else None
        below = self.fft.get(i - 1)                # a frequency-domain layer below: its dz spectra can ride along
        written = ctypes.c_int(0)
        call('st_conv1d_nwc_bwd_data_fft_chain_f32', self.dZ[i].ref, self._ptr(f['zf']), self._ptr(f['gfwd']), l.width,
             self.geo[i][2], act, self.dZ[i - 1].ref, self._ptr(f['tables']), self._ptr(below['tables']) if below else None,
             self._ptr(below['zf']) if below else None, below['width'] if below else 0, ctypes.byref(written), self._ptr(f['ws']),
             f['ws'].numel() * 4, s)
        zf_ready = written.value == 1
      elif i > 0:
        # X[i] is the ReLU output of layer i-1: its sign is the mask of tf.nn.relu's gradient
        # the kernel that writes dZ[i-1] also sums its columns: the bias gradient of layer i - 1
        act = self.X[i].ref if self.layers[i - 1].relu else None
        if self._transposed_in_place(i):           # dx = dz W^T straight from the layer's packed filters
          call('st_conv1d_1tap_bwd_data_bias_f32', self.dZ[i].ref, self._ptr(self._slice(self.params, i)[0]), act, self.dZ[i - 1].ref,
               self._ptr(self._slice(self.grads, i - 1)[1]), self._ptr(self.wgrad_ws), self.wgrad_ws.numel() * 4, s)
        else:
          self._wait_bwd_operands(i if i > wait_all_below else None)
          call('st_conv1d_nwc_bwd_data_bias_f32', self.dZ[i].ref, self._ptr(self.packed_t[i]), l.width, self.geo[i][2],
               act, self.dZ[i - 1].ref, self._ptr(self._slice(self.grads, i - 1)[1]), self._ptr(self.wgrad_ws),
               self.wgrad_ws.numel() * 4, s)
        bias_from_above = True
      if deferred == i and on_layer_done is not None and wanted(i):
        # the gradient of this layer is complete when the side stream is (and with it those of the layers above that went
        # the same way: the stream runs in order): hand the bucket to the all-reduce only now, with back-prop to the input
        # already enqueued beside it
        self._join_side_stream()
        side_wgrad = False
        on_layer_done(i)
      if deferred == i:
        deferred = None
    if side_wgrad or top_pending is not None:
      self._join_side_stream()
    if top_pending is not None and hook is not None:
      hook(top_pending)

  def _adam_rate(self, lr, beta1, beta2):
    """The bias-corrected rate of the NEXT update (tf.train.AdamOptimizer: lr * sqrt(1 - beta2^t) / (1 - beta1^t)); counts it."""
    self.step_count += 1
    t = self.step_count
    return lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)

  def _refresh_after_update(self):
    """The operands the NEXT forward pass derives from the weights: filter spectra of the frequency-domain layers / the bf16 filter
    copies -- the bottom layer's on the compute stream, the others on the side stream with an event each."""
    if self.fft:
      self._refresh_gfwd()
    elif self.conv_mode == 'bf16' and hasattr(self, 'Wb'):
      self._refresh_wb_after_update()

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

  # ---- the whole training step from a HIP graph ------------------------------------------------------------------
  def enable_step_graph(self, on=True):
    """From the next `load_batch` on, lengths and labels are uploaded into persistent per-parity device buffers so that
    `train_step_graph` can replay captured launches that hold their addresses."""
    self._step_graph_on = bool(on)
    if not hasattr(self, '_step_parity'):
      self._step_parity, self._parity_consumed, self._parity_written = 0, [None, None], [set(), set()]
      self._step_graphs, self._step_graph_seen = {}, set()
      self._rate_host = torch.zeros(16, dtype=torch.float32, pin_memory=True)
      self._rate_turn = 0
      # the device scalars of the Adam rate exist (and are zero-filled) before any stream copies into them: created lazily, the
      # fill kernel on a busy compute stream ran AFTER the copy on the idle upload stream and wiped the step's rate
      for par in (0, 1):
        self._storage.view('adam_rate_par%d' % par, 4)
      torch.cuda.synchronize(self.device)

  def _step_body(self, grad_scale, max_grad_norm, beta1, beta2, eps, rate_dev):
    """The launch sequence of one training step in the order a captured graph holds it: the operands derived from the weights
    FIRST (what `apply_update` does at its end for the next step: inside a graph nothing may outlive the capture, and at the
    head of the step the rebuild still overlaps the first layers), forward, CTC, backward, clip + Adam with the rate read from
    device memory; every side stream joined at the end."""
    self.mark_weights_changed()
    self._refresh_after_update()
    self.forward()
    self.ctc_loss_grad(grad_scale)
    self.backward()
    call('st_global_norm_clip_adam_gated_dev_f32', self._ptr(self.params), self._ptr(self.grads), self._ptr(self.adam_m),
         self._ptr(self.adam_v), self.n_flat, float(max_grad_norm), 0.0, self._ptr(rate_dev), beta1, beta2, eps,
         self._ptr(self.stats), self._ptr(self.gate), self._ptr(self.norm_ws), self.norm_ws.numel() * 4, self.stream_ptr)
    self._join_side_stream()
    for name in ('_gfwd_ready', '_wb_ready', '_bwd_ready'):      # events of this sequence: consumed inside it
      d = getattr(self, name, None)
      if d:
        d.clear()

  def train_step_graph(self, grad_scale, lr, max_grad_norm=5.0, beta1=0.9, beta2=0.999, eps=1e-3):
    """forward + ctc_loss_grad + backward + apply_update of the batch `load_batch` / `set_labels` staged, as ONE graph launch.

    The step's ~100 kernel launches and ~40 cross-stream events are captured once per (shape, label-length class, parity) -- the
    second time a key shows up; the first time runs the same sequence eagerly -- and replayed afterwards: the launch sequence,
    its side-stream forks and joins included, is identical, so the result is bit-identical to the eager step
    (tests/test_gpu_api.py).  What changes from step to step lives in device memory the captured launches point at: the input
    batch (X[0]), lengths and labels (per-parity buffers, `enable_step_graph`), the Adam rate (a device scalar per parity,
    uploaded with them).  Single-process training only: the data-parallel exchange stays on the eager path."""
    if not getattr(self, '_step_graph_on', False):
      raise RuntimeError('train_step_graph: call enable_step_graph() before load_batch / set_labels')
    lib = _lib.load()
    B, T = self.X[-1].batch, self.X[-1].frames
    kpl_len = next((k * 32 - 1 for k in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16) if k * 64 >= 2 * self.max_label_len + 1), None)
    if kpl_len is None:
      raise ValueError('label of length {} is too long for the CTC kernel (max 511)'.format(self.max_label_len))
    par = self._step_parity
    main = self._stream if self._stream is not None else torch.cuda.current_stream(self.device)
    # the rate of this update: pinned slot -> this parity's device scalar, with the step's other uploads
    lr_t = self._adam_rate(lr, beta1, beta2)
    self._rate_turn = (self._rate_turn + 1) % 16
    self._rate_host[self._rate_turn] = lr_t
    rate_dev = self._storage.view('adam_rate_par%d' % par, 4)[0]
    with torch.cuda.stream(self._up_stream):
      if self._parity_consumed[par] is not None:
        self._up_stream.wait_event(self._parity_consumed[par])
      rate_dev[:1].copy_(self._rate_host[self._rate_turn:self._rate_turn + 1], non_blocking=True)
      ev = torch.cuda.Event()
      ev.record(self._up_stream)
      self._uploads.append(ev)
    need = lib.st_ctc_ws(B, T, kpl_len)
    if self.ctc_ws is None or self.ctc_ws.numel() * 4 < need:
      self.ctc_ws = torch.empty(need // 4 + 64, dtype=torch.float32, device=self.device)
    key = (self._shape, self._storage.generation, kpl_len, par, float(grad_scale), float(max_grad_norm), beta1, beta2, eps,
           self.fft_conv, self.ctc_ws.data_ptr())
    # everything enqueued outside the graph that it depends on: uploads, side-stream work of an eager step before this one
    self._wait_uploads()
    self._join_side_stream()
    saved_len, self.max_label_len = self.max_label_len, kpl_len      # (the CTC launch depends on the length class only)
    try:
      graph = self._step_graphs.get(key)
      if getattr(self, '_rejected_labels', None):
        graph = None                                                  # (deferred label errors: the eager sequence marks them)
        self._step_body(grad_scale, max_grad_norm, beta1, beta2, eps, rate_dev)
      elif graph is None and key not in self._step_graph_seen:
        self._step_graph_seen = {k for k in self._step_graph_seen if k[1] == self._storage.generation} | {key}
        self._step_body(grad_scale, max_grad_norm, beta1, beta2, eps, rate_dev)
      else:
        if graph is None:
          self._step_graphs = {k: g for k, g in self._step_graphs.items() if k[1] == self._storage.generation}
          torch.cuda.synchronize(self.device)
          graph = torch.cuda.CUDAGraph()
          own_stream, self._stream = self._stream, None
          try:
            with torch.cuda.graph(graph, capture_error_mode='thread_local'):
              self._step_body(grad_scale, max_grad_norm, beta1, beta2, eps, rate_dev)
          finally:
            self._stream = own_stream
          self._step_graphs[key] = graph
        graph.replay()
    finally:
      self.max_label_len = saved_len
    self._updates_in_flight = getattr(self, '_updates_in_flight', 0) + 1
    self.mark_weights_changed()                  # an eager pass after this one rebuilds its operands itself
    done = torch.cuda.Event()
    done.record(main)
    self._parity_consumed[par] = done
    self._parity_written[par].clear()
    self._step_parity = par ^ 1

  def greedy_decode(self, merge_repeated=True):
    """tf.nn.ctc_greedy_decoder (speech_model.py:113-115) -> (list of id lists, neg_sum_logits [B,1])."""
    self._wait_uploads()
    call('st_ctc_greedy_decode', self.X[-1].ref, self._ptr(self.ctc_lens), int(merge_repeated),
         self._ptr(self.dec_ids), self.t_out, self._ptr(self.dec_lens), self._ptr(self.dec_score), self.stream_ptr)
    lens = self.dec_lens.cpu().numpy()
    ids = self.dec_ids.view(-1, self.t_out).cpu().numpy()
    return [ids[b, :lens[b]].tolist() for b in range(len(lens))], self.dec_score.cpu().numpy().reshape(-1, 1)

  def greedy_decode_async(self, merge_repeated=True):
    """``greedy_decode`` without the host synchronisation: launches the decoder and the D2H copies of its
    outputs into pinned host buffers and returns a handle; ``handle.result()`` waits for that batch only.  Lets
    a caller enqueue the next batch's forward before it reads this batch's transcripts (inference.transcribe)."""
    self._wait_uploads()
    call('st_ctc_greedy_decode', self.X[-1].ref, self._ptr(self.ctc_lens), int(merge_repeated),
         self._ptr(self.dec_ids), self.t_out, self._ptr(self.dec_lens), self._ptr(self.dec_score), self.stream_ptr)
    B, n = self.dec_lens.numel(), self.dec_ids.numel()
    if not hasattr(self, '_dec_host'):
      self._dec_host, self._dec_turn = [None, None], 0
    self._dec_turn ^= 1
    slot = self._dec_host[self._dec_turn]
    if slot is None or slot[0].numel() < n or slot[1].numel() < B:
      if slot is not None:
        slot[2].synchronize()                                      # a copy into the old buffers may be in flight
      slot = [torch.empty(max(n, 1), dtype=torch.int32, pin_memory=True),
              torch.empty(max(B, 1), dtype=torch.int32, pin_memory=True), torch.cuda.Event()]
      self._dec_host[self._dec_turn] = slot
    stream = self._stream if self._stream is not None else torch.cuda.current_stream(self.device)
    with torch.cuda.stream(stream):
      slot[0][:n].copy_(self.dec_ids, non_blocking=True)
      slot[1][:B].copy_(self.dec_lens, non_blocking=True)
      slot[2].record(stream)
    return _PendingDecode(slot, B, self.t_out)

  def beam_search_decode(self, beam_width=16, input_transform=None, merge_repeated=False):
    """LM-free CTC prefix beam search, top path (stock tf.nn.ctc_beam_search_decoder semantics; the
    reference's own beam search needs its KenLM fork, speech_model.py:101-111)
    -> (list of id lists, log_prob [B,1]).  Beams up to 128 (the reference runs 100).  ``input_transform='log10_softmax'``
    searches on log10(softmax(logits) + 1e-8), the reference's decoder input (speech_model.py:102); ``merge_repeated``
    (reference: False, speech_model.py:110) collapses repeated labels of the returned prefix the way TF's decoder does."""
    lib = _lib.load()
    B = self.dec_lens.numel()
    need = lib.st_ctc_beam_ws(B, self.t_out, int(beam_width))
    ws = self._storage.view('beam_ws', need // 4 + 16, torch.int32)[0]
    self._wait_uploads()
    call('st_ctc_beam_search_decode_ex', self.X[-1].ref, self._ptr(self.ctc_lens), int(beam_width), beam_input_transform(input_transform),
         self._ptr(self.dec_ids), self.t_out, self._ptr(self.dec_lens), self._ptr(self.dec_score),
         self._ptr(ws), ws.numel() * 4, self.stream_ptr)
    lens = self.dec_lens.cpu().numpy()
    ids = self.dec_ids.view(-1, self.t_out).cpu().numpy()
    out = [ids[b, :lens[b]].tolist() for b in range(len(lens))]
    if merge_repeated:
      out = [merge_repeated_labels(seq) for seq in out]
    return out, self.dec_score.cpu().numpy().reshape(-1, 1)

  def beam_search_decode_async(self, beam_width=16, decode_stream=None, input_transform=None):
    """``beam_search_decode`` without the host synchronisation and OFF the compute stream: the logits and lengths of this
    batch are copied into a decoder slot, the search runs on ``decode_stream`` (default: a stream of the engine's own;
    `decoder_streams` gives CU-masked ones -- a list of streams is used in turn, consecutive batches' searches then run side by
    side) and its outputs go to pinned host memory; returns a handle whose
    ``result()`` waits for this batch only.  The caller enqueues the next batch's forward pass meanwhile -- the search is ONE
    wavefront per utterance (3.9 ms for 16 x 30 s, beam 16: as long as the forward pass) and leaves the chip to it."""
    lib = _lib.load()
    B, T = self.dec_lens.numel(), self.t_out
    xl = self.X[-1]
    need = lib.st_ctc_beam_ws(B, T, int(beam_width))
    streams = list(decode_stream) if isinstance(decode_stream, (list, tuple)) else [decode_stream]
    # one slot more than decoder streams: the forward pass fills a slot while every stream searches one
    if not hasattr(self, '_beam_slots') or len(self._beam_slots) != len(streams) + 1:
      for old in getattr(self, '_beam_slots', []):
        if old is not None:
          old['event'].synchronize()
      self._beam_slots, self._beam_turn = [None] * (len(streams) + 1), 0
    self._beam_turn += 1
    which = self._beam_turn % len(self._beam_slots)
    slot = self._beam_slots[which]
    decode_stream = streams[self._beam_turn % len(streams)]
    main = self._stream if self._stream is not None else torch.cuda.current_stream(self.device)
    if decode_stream is None:
      if getattr(self, '_decode_stream', None) is None:
        self._decode_stream = torch.cuda.Stream(self.device)
      decode_stream = self._decode_stream
    if slot is None or slot['logits'].numel() < xl.buf.numel() or slot['ids'].numel() < B * T or slot['ws'].numel() * 4 < need or \
        slot['lens'].numel() < B:
      if slot is not None:
        slot['event'].synchronize()                               # the old buffers may still be in use
      i32 = lambda n, **kw: torch.empty(max(n, 1), dtype=torch.int32, **kw)
      slot = dict(logits=torch.empty(xl.buf.numel(), dtype=torch.float32, device=self.device), lens=i32(B, device=self.device),
                  ids=i32(B * T, device=self.device), out_lens=i32(B, device=self.device),
                  score=torch.empty(max(B, 1), dtype=torch.float32, device=self.device), ws=i32(need // 4 + 16, device=self.device),
                  ids_h=i32(B * T, pin_memory=True), lens_h=i32(B, pin_memory=True),
                  score_h=torch.empty(max(B, 1), dtype=torch.float32, pin_memory=True), event=torch.cuda.Event(), generation=0)
      slot['event'].record(decode_stream)
      self._beam_slots[which] = slot
    slot['generation'] += 1                                       # handles of the batch that last used this slot are stale from here on
    self._wait_uploads()
    main.wait_event(slot['event'])                               # the search that last read this slot is through
    with torch.cuda.stream(main):
      slot['logits'][:xl.buf.numel()].copy_(xl.buf, non_blocking=True)
      slot['lens'][:B].copy_(self.ctc_lens, non_blocking=True)
      ready = torch.cuda.Event()
      ready.record(main)
    desc = Tensor3(slot['logits'].data_ptr(), xl.batch, xl.frames, xl.channels, xl.halo, xl.t_pitch, xl.c_pitch)
    decode_stream.wait_event(ready)
    call('st_ctc_beam_search_decode_ex', ctypes.byref(desc), self._ptr(slot['lens']), int(beam_width), beam_input_transform(input_transform),
         self._ptr(slot['ids']), T,
         self._ptr(slot['out_lens']), self._ptr(slot['score']), self._ptr(slot['ws']), slot['ws'].numel() * 4,
         ctypes.c_void_p(decode_stream.cuda_stream))
    with torch.cuda.stream(decode_stream):
      slot['ids_h'][:B * T].copy_(slot['ids'][:B * T], non_blocking=True)
      slot['lens_h'][:B].copy_(slot['out_lens'][:B], non_blocking=True)
      slot['score_h'][:B].copy_(slot['score'][:B], non_blocking=True)
      slot['event'].record(decode_stream)
    return _PendingBeamDecode(slot, B, T)

  def fetch_losses(self, precise=False):
    """Per-utterance CTC losses [B] on the host, after checking the status words: both arrays come back in one
    pinned, asynchronous copy each and a single event wait (a step's only host synchronisation).  float32 like
    tf.nn.ctc_loss; ``precise=True`` returns float64 = hi + lo of the kernel's (hi, lo) pairs (-log p to ~1e-6 where one
    fp32 ulp of a 10 s utterance's loss is 1.2e-4)."""
    B = self.loss.numel()
    if not hasattr(self, '_loss_host') or self._loss_host[0].numel() < 2 * B:
      self._loss_host = (torch.empty(max(2 * B, 128), dtype=torch.float32, pin_memory=True),
                         torch.empty(max(B, 64), dtype=torch.int32, pin_memory=True), torch.cuda.Event(),
                         torch.empty(16, dtype=torch.float32, pin_memory=True))
    loss_h, status_h, event, gate_h = self._loss_host
    stream = self._stream if self._stream is not None else torch.cuda.current_stream(self.device)
    lost_h = self._streamk_lost_async(stream)
    with torch.cuda.stream(stream):
      loss_h[:2 * B].copy_(self.loss_pair, non_blocking=True)
      status_h[:B].copy_(self.ctc_status, non_blocking=True)
      gate_h[:1].copy_(self.gate, non_blocking=True)
      event.record(stream)
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
