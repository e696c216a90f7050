"""Device buffers of the Wav2Letter engine: the padded NWC tensor view, grow-only named storage, layer geometry.

HBM layout (DESIGN.md "Data layout"):
  * activations X[i] / gradients dZ[i]: padded NWC ``st_tensor3`` buffers, zero halos sized for
    the consuming convolution, channel pitch rounded to 16 floats (32 for wide tensors, ``channel_pitch``);
  * parameters, gradients, Adam m/v: four flat fp32 buffers with identical layout
    [F0 | b0 | F1 | b1 | ...], filters in the packed GEMM layout [k_pad][n_pad] -- so the
    gradient all-reduce and clip+Adam each see one contiguous buffer.
"""
import ctypes

import torch

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


class _StagedHostBatch:
  """One of the engine's two H2D staging buffers: ``event`` = copy finished, ``consumed`` = the compute stream has
  read it (``Wav2LetterEngine.stage_host_batch`` / ``load_batch``)."""

  def __init__(self, tensor):
    self.tensor = tensor
    self.event = torch.cuda.Event()
    self.consumed = torch.cuda.Event()
    self.consumed.record()
    self.taken = True             # handed to load_batch (host-side state; `consumed` is the device-side one)
