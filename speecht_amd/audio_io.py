"""Real-audio ingest for `speecht-cli preprocess`: what ``librosa.load(path)`` does
for the reference (preprocessing.py:169) -- decode the file, mix to mono, resample to librosa's default
22 050 Hz with the ``kaiser_best`` windowed-sinc filter, float32.

Everything here is host-side numpy: ingest runs once per corpus, off the hot path.  Neither libsndfile nor
resampy is installed, so both pieces restate their published algorithms:

* FLAC (the LibriSpeech container): frame / subframe / Rice-residual decoding per the FLAC format
  specification.  The decoder is pinned bit-exactly without any external tool: STREAMINFO carries the MD5 of
  the unencoded samples and ``decode_flac`` verifies it.
* ``kaiser_best``: resampy's band-limited interpolation (64 zero crossings, 512 table entries per crossing,
  roll-off 0.9475937167399596, Kaiser beta 14.769656459379492; Smith's algorithm, linear interpolation of the
  filter table), followed by librosa's ``fix_length`` to ``ceil(n * ratio)`` samples.  PARITY UNPINNED beyond
  the output length, which the reference's own test pins for its fixture
  (speecht/tests/test_speechCorpusReader.py:45: 114 881 samples).
"""
import hashlib
import math

import numpy as np


class FlacError(ValueError):
  pass


class _BitReader:
  """MSB-first bit reader over a bytes object with a small integer accumulator."""

  def __init__(self, data, pos=0):
    self.data, self.pos = data, pos           # pos: next byte to load
    self.acc, self.nacc = 0, 0

  def _fill(self, need):
    while self.nacc < need:
      chunk = self.data[self.pos:self.pos + 8]
      if not chunk:
        raise FlacError('unexpected end of stream')
      self.acc = (self.acc << (8 * len(chunk))) | int.from_bytes(chunk, 'big')
      self.nacc += 8 * len(chunk)
      self.pos += len(chunk)

  def read(self, n):
    if n == 0:
      return 0
    self._fill(n)
    self.nacc -= n
    v = self.acc >> self.nacc
    self.acc &= (1 << self.nacc) - 1
    return v

  def read_signed(self, n):
    v = self.read(n)
    return v - (1 << n) if v >> (n - 1) else v

  def read_unary(self):
    """Number of 0 bits before the next 1 bit (the 1 is consumed)."""
    q = 0
    while True:
      if self.nacc == 0:
        self._fill(1)
      if self.acc == 0:
        q += self.nacc
        self.nacc = 0
        continue
      lead = self.nacc - self.acc.bit_length()
      self.nacc -= lead + 1
      self.acc &= (1 << self.nacc) - 1
      return q + lead

  def align(self):
    drop = self.nacc % 8
    self.nacc -= drop
    self.acc &= (1 << self.nacc) - 1

  def byte_position(self):
    return self.pos - self.nacc // 8


def _read_residual(br, order, blocksize, out):
  method = br.read(2)
  if method > 1:
    raise FlacError('reserved residual coding method')
  pbits, escape = (4, 15) if method == 0 else (5, 31)
  part_order = br.read(4)
  parts = 1 << part_order
  i = order
  for part in range(parts):
    n = (blocksize >> part_order) - (order if part == 0 else 0)
    k = br.read(pbits)
    if k == escape:
      raw = br.read(5)
      for _ in range(n):
        out[i] = br.read_signed(raw) if raw else 0
        i += 1
    else:
      for _ in range(n):
        v = (br.read_unary() << k) | br.read(k)
        out[i] = (v >> 1) ^ -(v & 1)
        i += 1


_FIXED = {0: (), 1: (1,), 2: (2, -1), 3: (3, -3, 1), 4: (4, -6, 4, -1)}


def _read_subframe(br, bps, blocksize):
  if br.read(1):
    raise FlacError('subframe padding bit set')
  kind = br.read(6)
  wasted = 0
  if br.read(1):
    wasted = br.read_unary() + 1
    bps -= wasted
  out = [0] * blocksize
  if kind == 0:                                   # CONSTANT
    out = [br.read_signed(bps)] * blocksize
  elif kind == 1:                                 # VERBATIM
    out = [br.read_signed(bps) for _ in range(blocksize)]
  elif 8 <= kind <= 12 or 32 <= kind:             # FIXED / LPC
    if kind <= 12:
      order, shift, coefs = kind - 8, 0, _FIXED[kind - 8]
      for i in range(order):
        out[i] = br.read_signed(bps)
    else:
      order = kind - 31
      for i in range(order):
        out[i] = br.read_signed(bps)
      precision = br.read(4) + 1
      if precision == 16:
        raise FlacError('invalid LPC precision')
      shift = br.read_signed(5)
      if shift < 0:
        raise FlacError('negative LPC shift')
      coefs = tuple(br.read_signed(precision) for _ in range(order))
    _read_residual(br, order, blocksize, out)
    for i in range(order, blocksize):             # out[i] currently holds the residual
      pred = 0
      for j, c in enumerate(coefs):
        pred += c * out[i - 1 - j]
      out[i] += pred >> shift
  else:
    raise FlacError('reserved subframe type {}'.format(kind))
  if wasted:
    out = [v << wasted for v in out]
  return out


_BLOCKSIZES = {1: 192, 2: 576, 3: 1152, 4: 2304, 5: 4608}
_SAMPLE_SIZES = {1: 8, 2: 12, 4: 16, 5: 20, 6: 24}


def decode_flac(path_or_bytes, verify=True):
  """Decode a FLAC stream -> (int32 samples [n, channels], samplerate, bits_per_sample).  With ``verify`` the
  MD5 signature of STREAMINFO is checked against the decoded samples (when the encoder stored one)."""
  data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, 'rb').read()
  if data[:4] != b'fLaC':
    raise FlacError('not a FLAC stream')
  pos, info = 4, None
  while True:
    header = data[pos]
    length = int.from_bytes(data[pos + 1:pos + 4], 'big')
    if header & 0x7F == 0:
      br = _BitReader(data, pos + 4)
      br.read(16); br.read(16); br.read(24); br.read(24)
      info = dict(rate=br.read(20), channels=br.read(3) + 1, bps=br.read(5) + 1, total=br.read(36),
                  md5=data[pos + 4 + 18:pos + 4 + 34])
    pos += 4 + length
    if header & 0x80:
      break
  if info is None:
    raise FlacError('STREAMINFO missing')
  channels, bps = info['channels'], info['bps']
  blocks = []
  br = _BitReader(data, pos)
  end = len(data)
  while br.byte_position() < end:
    if br.read(14) != 0x3FFE:
      raise FlacError('lost frame sync at byte {}'.format(br.byte_position()))
    br.read(2)                                    # reserved + blocking strategy
    bs_code, sr_code = br.read(4), br.read(4)
    assignment, ss_code = br.read(4), br.read(3)
    br.read(1)
    first = br.read(8)                            # UTF-8 style frame / sample number
    extra = 0
    while first & (0x80 >> extra):
      extra += 1
    for _ in range(max(extra - 1, 0)):
      br.read(8)
    if bs_code == 6:
      blocksize = br.read(8) + 1
    elif bs_code == 7:
      blocksize = br.read(16) + 1
    elif bs_code >= 8:
      blocksize = 256 << (bs_code - 8)
    elif bs_code in _BLOCKSIZES:
      blocksize = _BLOCKSIZES[bs_code]
    else:
      raise FlacError('reserved block size code')
    if sr_code == 12:
      br.read(8)
    elif sr_code in (13, 14):
      br.read(16)
    frame_bps = _SAMPLE_SIZES.get(ss_code, bps)
    br.read(8)                                    # CRC-8 (the MD5 check covers integrity)
    if assignment < 8:
      chans = [_read_subframe(br, frame_bps, blocksize) for _ in range(assignment + 1)]
    elif assignment == 8:                         # left, side
      left = _read_subframe(br, frame_bps, blocksize)
      side = _read_subframe(br, frame_bps + 1, blocksize)
      chans = [left, [l - s for l, s in zip(left, side)]]
    elif assignment == 9:                         # side, right
      side = _read_subframe(br, frame_bps + 1, blocksize)
      right = _read_subframe(br, frame_bps, blocksize)
      chans = [[s + r for s, r in zip(side, right)], right]
    elif assignment == 10:                        # mid, side
      mid = _read_subframe(br, frame_bps, blocksize)
      side = _read_subframe(br, frame_bps + 1, blocksize)
      left = [(((m << 1) | (s & 1)) + s) >> 1 for m, s in zip(mid, side)]
      chans = [left, [l - s for l, s in zip(left, side)]]
    else:
      raise FlacError('reserved channel assignment')
    br.align()
    br.read(16)                                   # CRC-16
    blocks.append(np.array(chans, dtype=np.int32).T)
  samples = np.concatenate(blocks) if blocks else np.zeros((0, channels), np.int32)
  if info['total'] and samples.shape[0] != info['total']:
    raise FlacError('decoded {} samples, STREAMINFO says {}'.format(samples.shape[0], info['total']))
  if verify and any(info['md5']):
    width = (bps + 7) // 8
    raw = samples.astype('<i{}'.format(4 if width == 3 else width))
    payload = raw.tobytes() if width != 3 else raw.view(np.uint8).reshape(-1, 4)[:, :3].tobytes()
    if hashlib.md5(payload).digest() != info['md5']:
      raise FlacError('MD5 of the decoded samples does not match STREAMINFO')
  return samples, info['rate'], bps


# ---- resampy "kaiser_best" -------------------------------------------------------------------------------
_KAISER_BEST = dict(num_zeros=64, precision=9, rolloff=0.9475937167399596, beta=14.769656459379492)
_filter_cache = {}


def _kaiser_best_filter():
  if 'win' not in _filter_cache:
    cfg = _KAISER_BEST
    num_table = 2 ** cfg['precision']
    n = num_table * cfg['num_zeros']
    sinc_win = cfg['rolloff'] * np.sinc(cfg['rolloff'] * np.linspace(0, cfg['num_zeros'], num=n + 1, endpoint=True))
    taper = np.kaiser(2 * n + 1, cfg['beta'])[n:]           # right half of the symmetric window
    _filter_cache['win'] = (taper * sinc_win, num_table)
  return _filter_cache['win']


def resample_kaiser_best(y, sr_orig, sr_new):
  """resampy.resample(y, sr_orig, sr_new, filter='kaiser_best') for a 1-D signal (vectorised over outputs)."""
  y = np.asarray(y, dtype=np.float64)
  ratio = float(sr_new) / float(sr_orig)
  n_out = int(y.shape[0] * ratio)
  interp_win, num_table = _kaiser_best_filter()
  if ratio < 1:
    interp_win = interp_win * ratio
  interp_delta = np.zeros_like(interp_win)
  interp_delta[:-1] = np.diff(interp_win)
  scale = min(1.0, ratio)
  index_step = int(scale * num_table)
  nwin, n_orig = interp_win.shape[0], y.shape[0]
  time_register = np.arange(n_out, dtype=np.float64) / ratio          # accumulated by repeated addition upstream
  n = time_register.astype(np.int64)
  out = np.zeros(n_out)
  taps = np.arange((nwin + index_step - 1) // index_step)
  for wing in (0, 1):
    frac = scale * (time_register - n)
    if wing:
      frac = scale - frac
    index_frac = frac * num_table
    offset = index_frac.astype(np.int64)
    eta = index_frac - offset
    limit = (nwin - offset) // index_step                             # taps inside the filter table
    avail = (n + 1) if wing == 0 else (n_orig - n - 1)                # taps inside the signal
    count = np.minimum(limit, avail)
    idx = offset[:, None] + taps[None, :] * index_step
    valid = taps[None, :] < count[:, None]
    idx = np.where(valid, idx, 0)
    weight = interp_win[idx] + eta[:, None] * interp_delta[idx]
    src = (n[:, None] - taps[None, :]) if wing == 0 else (n[:, None] + taps[None, :] + 1)
    src = np.where(valid, src, 0)
    out += np.sum(np.where(valid, weight * y[src], 0.0), axis=1)
  return out


def librosa_load(path, sr=22050):
  """``librosa.load(path)`` as the reference calls it (preprocessing.py:169): decode, mono, resample to
  ``sr`` (None keeps the native rate), float32 in [-1, 1).  FLAC only; see preprocessing.load_audio for wav."""
  samples, rate, bps = decode_flac(path)
  y = samples.astype(np.float64).mean(axis=1) / float(1 << (bps - 1))
  if sr is not None and sr != rate:
    target = int(math.ceil(y.shape[0] * float(sr) / rate))             # librosa: fix_length(resampled, ceil(n * ratio))
    z = resample_kaiser_best(y, rate, sr)
    y = np.concatenate([z, np.zeros(max(0, target - z.shape[0]))])[:target]
    rate = sr
  return y.astype(np.float32), rate
