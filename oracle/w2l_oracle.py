"""CPU oracle for the speechT Wav2Letter hot path  --  TEST INFRASTRUCTURE ONLY.

This module is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The shipped path (``speecht_amd``) must never route through it.

PARITY UNPINNED BY THE REFERENCE: the arithmetic of this path lives in two un-vendored
third-party packages (tensorflow>=1.0.1, librosa>=0.5.0; /root/reference/requirements.txt:2,5)
that cannot be installed here, and the reference's own tests pin no numeric value of the
path (speecht/tests/test_speechCorpusReader.py:25-73).  The oracle therefore restates the
*published* op semantics at the reference's call sites and is pinned
three ways instead (tests/test_oracle_*.py): (1) an independent formulation with torch CPU
ops (F.conv1d + autograd, F.ctc_loss), (2) scipy.signal / scipy.fft for the STFT chain,
(3) analytical known-answer cases.  ``speecht.vocabulary`` is the one module that imports
here; its golden vectors are in tests/golden/ (scripts/make_golden.py).

All functions compute in float64 unless ``dtype`` says otherwise, are plain numpy, and cite
the reference file:line they follow (paths relative to /root/reference).
"""
import math

import numpy as np

BLANK_OFFSET = 1  # num_classes = vocabulary.SIZE + 1, blank = num_classes - 1 (speech_model.py:301)


# ----------------------------------------------------------------------------------------
# vocabulary  (speecht/vocabulary.py:16-81)
# ----------------------------------------------------------------------------------------
VOCAB_SIZE = 28
_ALPHABET = "abcdefghijklmnopqrstuvwxyz' "


def sentence_to_ids(sentence):
  """vocabulary.py:57-67: lower-case, a-z -> 0..25, apostrophe -> 26, space -> 27."""
  return [_ALPHABET.index(ch) for ch in sentence.lower()]


def ids_to_sentence(ids):
  """vocabulary.py:70-81."""
  return ''.join(_ALPHABET[int(i)] for i in ids)


# ----------------------------------------------------------------------------------------
# feature chain  (speecht/preprocessing.py:29-58; librosa semantics: SURVEY Appendix A7-A9)
# ----------------------------------------------------------------------------------------
def hz_to_mel_slaney(f):
  f = np.asarray(f, dtype=np.float64)
  f_sp = 200.0 / 3
  mels = f / f_sp
  min_log_hz = 1000.0
  min_log_mel = min_log_hz / f_sp
  logstep = math.log(6.4) / 27.0
  with np.errstate(divide='ignore', invalid='ignore'):
    log_part = min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep
  return np.where(f >= min_log_hz, log_part, mels)


def mel_to_hz_slaney(m):
  m = np.asarray(m, dtype=np.float64)
  f_sp = 200.0 / 3
  freqs = f_sp * m
  min_log_hz = 1000.0
  min_log_mel = min_log_hz / f_sp
  logstep = math.log(6.4) / 27.0
  return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(sr, n_fft, n_mels):
  """librosa.filters.mel(sr, n_fft, n_mels, fmin=0, fmax=sr/2, htk=False, norm=1) (Appendix A7).

  Triangular filters on the Slaney mel scale over the 1+n_fft/2 FFT bin centres, each scaled
  by 2/(f[i+2]-f[i]).  Returns [n_mels, 1+n_fft//2] float64.
  """
  n_bins = 1 + n_fft // 2
  fftfreqs = np.linspace(0.0, sr / 2.0, n_bins)
  mel_pts = np.linspace(hz_to_mel_slaney(0.0), hz_to_mel_slaney(sr / 2.0), n_mels + 2)
  mel_f = mel_to_hz_slaney(mel_pts)
  fdiff = np.diff(mel_f)
  ramps = mel_f[:, None] - fftfreqs[None, :]
  weights = np.zeros((n_mels, n_bins))
  for i in range(n_mels):
    lower = -ramps[i] / fdiff[i]
    upper = ramps[i + 2] / fdiff[i + 1]
    weights[i] = np.maximum(0.0, np.minimum(lower, upper))
  enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
  return weights * enorm[:, None]


def hann_periodic(n):
  """scipy.signal.get_window('hann', n, fftbins=True)."""
  return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def stft_power(y, n_fft=512, hop_length=160):
  """|librosa.stft(y, n_fft, hop_length, center=True, window='hann', pad_mode='reflect')|**2.

  Returns [1+n_fft//2, 1+len(y)//hop] (Appendix A7).
  """
  y = np.asarray(y, dtype=np.float64)
  ypad = np.pad(y, n_fft // 2, mode='reflect')
  n_frames = 1 + (len(ypad) - n_fft) // hop_length
  idx = np.arange(n_fft)[None, :] + hop_length * np.arange(n_frames)[:, None]
  frames = ypad[idx] * hann_periodic(n_fft)[None, :]
  spec = np.fft.rfft(frames, axis=1)
  return (spec.real ** 2 + spec.imag ** 2).T


def power_to_db(S, amin=1e-10, top_db=80.0):
  """librosa.power_to_db(S, ref=np.max) (preprocessing.py:53; Appendix A8)."""
  S = np.asarray(S, dtype=np.float64)
  ref = np.max(S)
  log_spec = 10.0 * np.log10(np.maximum(amin, S))
  log_spec -= 10.0 * np.log10(np.maximum(amin, ref))
  return np.maximum(log_spec, log_spec.max() - top_db)


def normalize(values):
  """preprocessing.py:29-33: one mean and one population std over all elements."""
  values = np.asarray(values, dtype=np.float64)
  return (values - np.mean(values)) / np.std(values)


def calc_power_spectrogram(audio_data, samplerate, n_mels=128, n_fft=512, hop_length=160):
  """preprocessing.py:36-58 -> [time, n_mels]."""
  S = mel_filterbank(samplerate, n_fft, n_mels) @ stft_power(audio_data, n_fft, hop_length)
  return normalize(power_to_db(S)).T


def dct_basis(n_mfcc, n_mels):
  """librosa.filters.dct (0.5/0.6) == orthonormal DCT-II rows 0..n_mfcc-1 (scipy.fftpack.dct(type=2,
  norm='ortho') in later releases)."""
  basis = np.empty((n_mfcc, n_mels))
  basis[0, :] = 1.0 / math.sqrt(n_mels)
  samples = np.arange(1, 2 * n_mels, 2) * math.pi / (2.0 * n_mels)
  for i in range(1, n_mfcc):
    basis[i, :] = np.cos(i * samples) * math.sqrt(2.0 / n_mels)
  return basis


def delta_lfilter(data, width=9, order=1):
  """librosa.feature.delta of the 0.5.x line the reference was written against (requirements.txt:5
  ``librosa>=0.5.0``; preprocessing.py:78-79): edge-pad by ``width`` on the time axis, run the FIR
  window [4..-4]/60 as a causal filter from rest ``order`` times over the PADDED signal, then cut
  [-5-T : -5].  For order 1 this is the usual regression delta with clamped edges; for order 2 the filter's
  zero initial state leaks into the first three frames (a quirk of that release, restated on purpose).
  PARITY UNPINNED (librosa >= 0.6.1 switched to a Savitzky-Golay filter)."""
  data = np.asarray(data, dtype=np.float64)
  half = 1 + width // 2
  window = np.arange(half - 1.0, -half, -1.0)
  window /= np.sum(window ** 2)
  T = data.shape[-1]
  x = np.pad(data, [(0, 0)] * (data.ndim - 1) + [(width, width)], mode='edge')
  for _ in range(order):
    y = np.zeros_like(x)
    for k, wk in enumerate(window):              # y[n] = sum_k w[k] x[n-k], x[<0] = 0
      y[..., k:] += wk * x[..., :x.shape[-1] - k]
    x = y
  return x[..., x.shape[-1] - half - T:x.shape[-1] - half]


def calc_mfccs(audio_data, samplerate, n_mfcc=13, n_fft=512, hop_length=160, n_mels=128):
  """preprocessing.py:61-84 -> [time, 3 * n_mfcc]: librosa.feature.mfcc (= DCT-II of
  power_to_db(melspectrogram(n_mels=128), ref=1.0, top_db=80)), its delta and delta-delta, each block
  z-normalised on its own."""
  S = mel_filterbank(samplerate, n_fft, n_mels) @ stft_power(audio_data, n_fft, hop_length)
  log_spec = 10.0 * np.log10(np.maximum(1e-10, S))
  log_spec = np.maximum(log_spec, log_spec.max() - 80.0)
  mfcc = dct_basis(n_mfcc, n_mels) @ log_spec
  return np.concatenate((normalize(mfcc), normalize(delta_lfilter(mfcc)),
                         normalize(delta_lfilter(mfcc, order=2))), axis=0).T


# ----------------------------------------------------------------------------------------
# batch assembly  (speecht/speech_input.py:27-69)
# ----------------------------------------------------------------------------------------
def pad_batch(input_list, input_size):
  """speech_input.py:37-45: zero-pad to [B, max_T, input_size]; lengths = unpadded frames."""
  lengths = np.array([inp.shape[0] for inp in input_list], dtype=np.int64)
  max_t = int(lengths.max())
  out = np.zeros((len(input_list), max_t, input_size))
  for i, inp in enumerate(input_list):
    out[i, :inp.shape[0], :] = inp
  return out, lengths, max_t


def sparse_labels(label_list, max_time):
  """speech_input.py:58-69: (indices [N,2], values [N], dense_shape [B, max_time])."""
  idx, vals = [], []
  for b, label in enumerate(label_list):
    for p, ident in enumerate(label):
      idx.append([b, p])
      vals.append(int(ident))
  return (np.array(idx, dtype=np.int64).reshape(-1, 2), np.array(vals, dtype=np.int64),
          np.array([len(label_list), max_time], dtype=np.int64))


# ----------------------------------------------------------------------------------------
# convolution stack  (speecht/speech_model.py:128-181, 275-295; tf.nn.conv1d 'SAME': Appendix A1)
# ----------------------------------------------------------------------------------------
def same_padding(t_in, width, stride):
  t_out = -(-t_in // stride)
  pad_total = max((t_out - 1) * stride + width - t_in, 0)
  pad_left = pad_total // 2
  return t_out, pad_left, pad_total - pad_left


def _im2col(x, width, stride):
  B, T, C = x.shape
  t_out, pl, pr = same_padding(T, width, stride)
  xp = np.pad(x, ((0, 0), (pl, pr + stride), (0, 0)))
  idx = (np.arange(t_out) * stride)[:, None] + np.arange(width)[None, :]
  cols = xp[:, idx, :]                      # [B, t_out, W, C]
  return cols.reshape(B, t_out, width * C), t_out, pl


def conv1d_same_fwd(x, filters, bias, stride=1, relu=True):
  """speech_model.py:155,173,177: NWC cross-correlation, SAME, + bias, optional ReLU.

  x [B,T,Cin], filters [W,Cin,Cout], bias [Cout] -> [B, ceil(T/stride), Cout].
  """
  W, Cin, Cout = filters.shape
  cols, t_out, _ = _im2col(x, W, stride)
  y = cols @ filters.reshape(W * Cin, Cout) + bias
  return np.maximum(y, 0.0) if relu else y


def conv1d_same_bwd(x, filters, y, dy, stride=1, relu=True, need_dx=True):
  """Back-prop through conv1d_same_fwd (what optimizer.compute_gradients does, speech_model.py:78).

  ``y`` is the layer's output (post-ReLU); returns (dx, dfilters, dbias).  tf.nn.relu's
  gradient is dy * (y > 0).
  """
  W, Cin, Cout = filters.shape
  B, T, _ = x.shape
  dz = dy * (y > 0) if relu else dy
  cols, t_out, pl = _im2col(x, W, stride)
  dF = (cols.reshape(-1, W * Cin).T @ dz.reshape(-1, Cout)).reshape(W, Cin, Cout)
  db = dz.reshape(-1, Cout).sum(axis=0)
  dx = None
  if need_dx:
    dcols = (dz @ filters.reshape(W * Cin, Cout).T).reshape(B, t_out, W, Cin)
    t_pad = (t_out - 1) * stride + W
    dxp = np.zeros((B, max(t_pad, pl + T), Cin), dtype=dz.dtype)
    for w in range(W):
      dxp[:, w:w + (t_out - 1) * stride + 1:stride, :] += dcols[:, :, w, :]
    dx = dxp[:, pl:pl + T, :]
  return dx, dF, db


def block_dft_conv(x, filters, bias, relu=True, dz=None, prev_act=None, store=None, block=64, need_y=True):
  """The SAME stride-1 convolution of ``conv1d_same_fwd`` (speech_model.py:155,173,177) and its gradients
  (``conv1d_same_bwd``, speech_model.py:78) in the block-DFT form the frequency-domain kernels compute
  (speecht_amd/csrc/conv_fft.hip): blocks of ``block`` output frames, N = block + W - 1 point DFTs over real input,
  one complex channel contraction per bin, overlap-save forward / overlap-add back-prop, lag products for the filters.
  Exactly equal to the direct form in exact arithmetic (tests/test_oracle_conv_ctc.py); ``store`` (e.g. ``bf16_round``)
  is the STORAGE MODEL of the bf16-plane variant: applied to the real and imaginary parts of the input spectra S, of the
  gradient spectra Z and of the filter spectra G -- the three things that variant writes to memory in bf16 -- and nowhere
  else (transforms, products and inverse transforms accumulate in higher precision).  Not a reference behaviour.

  x [B,T,Cin], filters [W,Cin,Cout] -> y [B,T,Cout]; with dz (gradient wrt the pre-activation output, [B,T,Cout]) also
  (dx, dfilters, dbias); dx is multiplied by (prev_act > 0) when prev_act is given (the ReLU of the layer below)."""
  q = store if store is not None else (lambda a: a)
  W, cin, cout = filters.shape
  B, T, _ = x.shape
  V = block
  N = V + W - 1
  bins = N // 2 + 1
  _, pl, _ = same_padding(T, W, 1)
  blocks = -(-T // V)
  k = np.arange(bins)[:, None]
  wk = np.where((k[:, 0] == 0) | (2 * k[:, 0] == N), 1.0, 2.0) / N

  def dft(seg):                                            # [rows, n <= N, C] real -> [bins, rows, C] complex
    E = np.exp(-2j * np.pi * k * np.arange(seg.shape[1])[None, :] / N)
    return np.tensordot(E.real, seg, axes=([1], [1])) + 1j * np.tensordot(E.imag, seg, axes=([1], [1]))

  def idft_real(spec, m):                                  # [bins, rows, C] -> points m: [rows, len(m), C]
    ph = wk[:, None] * np.exp(2j * np.pi * k * np.asarray(m)[None, :] / N)          # [bins, len(m)]
    out = np.tensordot(ph.real, spec.real, axes=([0], [0])) - np.tensordot(ph.imag, spec.imag, axes=([0], [0]))
    return np.transpose(out, (1, 0, 2))

  def qc(z):
    return q(z.real) + 1j * q(z.imag)

  rows = [(b, j) for b in range(B) for j in range(blocks)]
  xp = np.zeros((B, pl + blocks * V + N, cin))
  xp[:, pl:pl + T] = x
  S = qc(dft(np.stack([xp[b, j * V:j * V + N] for b, j in rows])))
  Gw = np.exp(-2j * np.pi * k * np.arange(W)[None, :] / N)                           # [bins, W]
  Fw = filters.reshape(W, cin * cout)
  G = qc((Gw.real @ Fw + 1j * (Gw.imag @ Fw)).reshape(bins, cin, cout))
  y = None
  if need_y:
    Y = np.matmul(S, np.conj(G))
    yb = idft_real(Y, np.arange(V)) + bias
    y = np.zeros((B, T, cout))
    for r, (b, j) in enumerate(rows):
      n = min(V, T - j * V)
      y[b, j * V:j * V + n] = yb[r, :n]
    if relu:
      y = np.maximum(y, 0.0)
  if dz is None:
    return y
  dzp = np.zeros((B, blocks * V, cout))
  dzp[:, :T] = dz
  Z = qc(dft(np.stack([dzp[b, j * V:j * V + V] for b, j in rows])))
  X = np.matmul(Z, np.transpose(G, (0, 2, 1)))
  tp = np.arange(V)
  own, below, above = idft_real(X, tp + pl), idft_real(X, tp + V + pl), idft_real(X, tp - V + pl)
  ok_below, ok_above = (tp + V + pl < N), (tp - V + pl >= 0)
  dx = np.zeros((B, T, cin))
  for r, (b, j) in enumerate(rows):
    v = own[r].copy()
    if j > 0:
      v[ok_below] += below[r - 1][ok_below]
    if j + 1 < blocks:
      v[ok_above] += above[r + 1][ok_above]
    n = min(V, T - j * V)
    dx[b, j * V:j * V + n] = v[:n]
  if prev_act is not None:
    dx = dx * (prev_act > 0)
  Qs = np.matmul(np.transpose(S, (0, 2, 1)), np.conj(Z))              # lag products per bin [bins, cin, cout]
  ang = 2 * np.pi * k[:, 0, None] * np.arange(W)[None, :] / N         # [bins, W]
  cw, sw = wk[:, None] * np.cos(ang), wk[:, None] * np.sin(ang)
  dF = (cw.T @ Qs.real.reshape(bins, -1) - sw.T @ Qs.imag.reshape(bins, -1)).reshape(W, cin, cout)
  # (the bias gradient is taken from the UNROUNDED block sums: the kernels keep bin 0 of the gradient spectra in fp32 for it)
  db = dz.reshape(-1, cout).sum(axis=0)
  return y, dx, dF, db


def wav2letter_layers(input_size, num_classes=29):
  """speech_model.py:275-292: (width, stride, cin, cout, relu) for the 11 layers."""
  layers = [(48, 2, input_size, 250, True)]
  layers += [(7, 1, 250, 250, True)] * 7
  layers += [(32, 1, 250, 2000, True), (1, 1, 2000, 2000, True), (1, 1, 2000, num_classes, False)]
  return layers


def xavier_init(layers, seed=42, bias_range=0.0, dtype=np.float64):
  """SURVEY 8(d) synthetic weights: Xavier-uniform (speech_model.py:150-151, Appendix A2),
  biases U(-bias_range, bias_range) (zeros in the reference, speech_model.py:152)."""
  rng = np.random.default_rng(seed)
  params = []
  for (W, s, cin, cout, relu) in layers:
    limit = math.sqrt(6.0 / (W * cin + W * cout))
    F = rng.uniform(-limit, limit, size=(W, cin, cout)).astype(dtype)
    b = rng.uniform(-bias_range, bias_range, size=(cout,)).astype(dtype) if bias_range else np.zeros(cout, dtype)
    params.append((F, b))
  return params


def bf16_round(a):
  """Round-to-nearest-even to bfloat16 precision (8 significand bits), returned as float64.
  Storage model of BASELINE config 4 ("bf16 activations"): not a reference behaviour."""
  u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
  u = (u + (((u >> 16) & 1) + 0x7FFF)) & np.uint32(0xFFFF0000)
  return u.view(np.float32).astype(np.float64)


def wav2letter_forward(x, params, layers, keep=False, store=None, spectral=()):
  """speech_model.py:275-295 -> logits time-major [T', B, C] (and the per-layer outputs).

  ``store`` (e.g. ``bf16_round``) models reduced-precision storage: it is applied to the input, to the
  filters used in the products and to every layer output that is written back for the next layer -- not to
  the biases, the accumulation or the logits.  Layers listed in ``spectral`` follow the storage model of the
  frequency-domain form instead (``block_dft_conv``: spectra rounded, the fp32 master filters' spectra rounded)."""
  q = store if store is not None else (lambda a: a)
  h = q(x)
  acts = [h]
  for i, ((F, b), (W, s, cin, cout, relu)) in enumerate(zip(params, layers)):
    h = block_dft_conv(h, F, b, relu, store=store) if i in spectral else conv1d_same_fwd(h, q(F), b, s, relu)
    if i + 1 < len(layers):
      h = q(h)
    acts.append(h)
  logits = np.transpose(h, (1, 0, 2))
  return (logits, acts) if keep else logits


def wav2letter_backward(acts, params, layers, dlogits_tm, store=None, spectral=()):
  """Gradients of all filters/biases given d(avg_loss)/d(logits) time-major [T',B,C].
  ``store`` / ``spectral``: see wav2letter_forward; ``store`` is also applied to every activation gradient that is written back."""
  q = store if store is not None else (lambda a: a)
  dy = q(np.transpose(dlogits_tm, (1, 0, 2)))
  grads = [None] * len(layers)
  for i in reversed(range(len(layers))):
    (F, b), (W, s, cin, cout, relu) = params[i], layers[i]
    if i in spectral:
      _, dx, dF, db = block_dft_conv(acts[i], F, b, relu, dz=dy * (acts[i + 1] > 0) if relu else dy, store=store, need_y=False)
    else:
      dx, dF, db = conv1d_same_bwd(acts[i], q(F), acts[i + 1], dy, s, relu, need_dx=(i > 0))
    grads[i] = (dF, db)
    dy = q(dx) if dx is not None else None
  return grads


# ----------------------------------------------------------------------------------------
# CTC  (speech_model.py:74 tf.nn.ctc_loss; Appendix A3)
# ----------------------------------------------------------------------------------------
def _logsumexp(a, axis=None):
  m = np.max(a, axis=axis, keepdims=True)
  m = np.where(np.isfinite(m), m, 0.0)
  with np.errstate(divide='ignore'):
    r = np.log(np.sum(np.exp(a - m), axis=axis, keepdims=True)) + m
  return np.squeeze(r, axis=axis) if axis is not None else r.reshape(())


def log_softmax(x, axis=-1):
  m = np.max(x, axis=axis, keepdims=True)
  z = x - m
  return z - np.log(np.sum(np.exp(z), axis=axis, keepdims=True))


def ctc_loss_and_grad(logits_tm, labels, seq_lens):
  """tf.nn.ctc_loss(labels, logits[T,B,C], seq_lens) with default flags + its gradient.

  labels: list of B id lists; seq_lens: [B] (already ``sequence_lengths // 2``).
  blank = C-1.  Returns (loss [B], grad [T,B,C] = d loss_b / d logits, zero for t >= len_b).
  Raises ValueError when a label needs more frames than available (TF 1.x InvalidArgument
  "Not enough time for target transition sequence").
  """
  logits_tm = np.asarray(logits_tm, dtype=np.float64)
  T, B, C = logits_tm.shape
  blank = C - 1
  loss = np.zeros(B)
  grad = np.zeros_like(logits_tm)
  NEG = -np.inf
  for b in range(B):
    Tb = int(seq_lens[b])
    lab = [int(v) for v in labels[b]]
    L = len(lab)
    repeats = sum(1 for i in range(1, L) if lab[i] == lab[i - 1])
    if L + repeats > Tb:
      raise ValueError('Not enough time for target transition sequence '
                       '(required: {}, available: {})'.format(L + repeats, Tb))
    if Tb == 0:                                       # no frames: only the empty labelling is possible, p = 1
      continue
    U = 2 * L + 1
    ext = np.full(U, blank, dtype=np.int64)
    ext[1::2] = lab
    logy = log_softmax(logits_tm[:Tb, b, :])          # [Tb, C]
    ly = logy[:, ext]                                 # [Tb, U]
    skip = np.zeros(U, dtype=bool)                    # may come from u-2
    skip[2:] = (ext[2:] != blank) & (ext[2:] != ext[:-2])
    alpha = np.full((Tb, U), NEG)
    alpha[0, 0] = ly[0, 0]
    if U > 1:
      alpha[0, 1] = ly[0, 1]
    def shift(v, k):                                  # v[u-k] for k > 0, v[u+|k|] for k < 0
      r = np.full(U, NEG)
      if k > 0 and U > k:
        r[k:] = v[:U - k]
      elif k < 0 and U > -k:
        r[:U + k] = v[-k:]
      return r
    for t in range(1, Tb):
      prev = alpha[t - 1]
      s1 = shift(prev, 1)
      s2 = np.where(skip, shift(prev, 2), NEG)
      alpha[t] = ly[t] + _logsumexp(np.stack([prev, s1, s2]), axis=0)
    beta = np.full((Tb, U), NEG)                      # excludes the emission at t (TF convention)
    beta[Tb - 1, U - 1] = 0.0
    if U > 1:
      beta[Tb - 1, U - 2] = 0.0
    skip_f = np.zeros(U, dtype=bool)                  # may go to u+2
    if U > 2:
      skip_f[:-2] = skip[2:]
    for t in range(Tb - 2, -1, -1):
      nxt = beta[t + 1] + ly[t + 1]
      s1 = shift(nxt, -1)
      s2 = np.where(skip_f, shift(nxt, -2), NEG)
      beta[t] = _logsumexp(np.stack([nxt, s1, s2]), axis=0)
    tail = alpha[Tb - 1, U - 2:] if U > 1 else alpha[Tb - 1, U - 1:]
    logp = float(_logsumexp(tail, axis=0))
    loss[b] = -logp
    ab = alpha + beta                                 # [Tb, U]
    occ = np.zeros((Tb, C))
    with np.errstate(under='ignore'):
      w = np.exp(ab - logp)
    for u in range(U):
      occ[:, ext[u]] += w[:, u]
    grad[:Tb, b, :] = np.exp(logy) - occ
  return loss, grad


def ctc_greedy_decode(logits_tm, seq_lens, merge_repeated=True):
  """tf.nn.ctc_greedy_decoder (speech_model.py:113-115; Appendix A4).

  Returns (list of id lists, neg_sum_logits [B,1]).
  """
  logits_tm = np.asarray(logits_tm)
  T, B, C = logits_tm.shape
  blank = C - 1
  out, score = [], np.zeros((B, 1))
  for b in range(B):
    ids, prev = [], -1
    for t in range(int(seq_lens[b])):
      row = logits_tm[t, b]
      k = int(np.argmax(row))            # first maximum on ties
      score[b, 0] -= row[k]
      if k != blank and not (merge_repeated and k == prev):
        ids.append(k)
      prev = k
    out.append(ids)
  return out, score


def ctc_beam_search_decode(logits_tm, seq_lens, beam_width=16, input_transform=None):
  """LM-free CTC prefix beam search, top path only (SURVEY 8(f) item 3 / BASELINE config 5).

  PARITY UNPINNED: the reference only ever calls a beam search through its KenLM TensorFlow fork
  (speech_model.py:101-111, beam_width=100, merge_repeated=False, top_paths=1), which is not vendored.
  This restates the stock ``tf.nn.ctc_beam_search_decoder`` recursion (ctc_beam_search.h ``Step``) without
  a scorer: a beam entry is a label prefix with (p_blank, p_label); per frame an entry is kept
  ("stay": blank or repeat of its last label, plus the mass flowing in from its parent prefix when that
  is also in the beam) and extended by every non-blank class whose child prefix is not already in the
  beam; the best ``beam_width`` candidates by total probability survive.  Ordering is made total so that
  an implementation can match it exactly: candidates are ranked by (total desc, slot*C + c asc) where
  ``slot`` is the parent entry's rank in the previous frame and c == blank marks the stay candidate.
  Scores are natural-log probabilities under the per-frame softmax.

  ``input_transform='log10_softmax'``: the search runs on ``tf.log(tf.nn.softmax(logits) + 1e-8) / math.log(10)``, what the
  reference hands its decoder (speech_model.py:102); the decoder normalises that per frame like any other input.  The labels
  returned are the top prefix itself (merge_repeated=False, speech_model.py:110).

  Returns (list of id lists, log_prob [B,1]).
  """
  logits_tm = np.asarray(logits_tm, dtype=np.float64)
  if input_transform == 'log10_softmax':
    z = logits_tm - logits_tm.max(axis=-1, keepdims=True)
    sm = np.exp(z) / np.exp(z).sum(axis=-1, keepdims=True)
    logits_tm = np.log(sm + 1e-8) / math.log(10)
  elif input_transform not in (None, 'logits'):
    raise ValueError(input_transform)
  T, B, C = logits_tm.shape
  blank, ninf = C - 1, -np.inf
  lse = np.logaddexp
  out, score = [], np.zeros((B, 1))
  for b in range(B):
    beams = [((), 0.0, ninf)]                      # (prefix, log p_blank, log p_label), best first
    for t in range(min(int(seq_lens[b]), T)):
      row = logits_tm[t, b]
      lp = row - row.max()
      lp = lp - math.log(np.exp(lp).sum())
      slot_of = {pre: i for i, (pre, _, _) in enumerate(beams)}
      cands = []
      for slot, (pre, pb, pl) in enumerate(beams):
        tot = lse(pb, pl)
        stay_b = tot + lp[blank]
        stay_l = ninf
        if pre:
          mass = pl
          if pre[:-1] in slot_of:
            ppre, ppb, ppl = beams[slot_of[pre[:-1]]]
            mass = lse(mass, ppb if (ppre and ppre[-1] == pre[-1]) else lse(ppb, ppl))
          stay_l = mass + lp[pre[-1]]
        cands.append((lse(stay_b, stay_l), slot * C + blank, pre, stay_b, stay_l))
        for c in range(C - 1):
          child = pre + (c,)
          if child in slot_of:
            continue
          v = (pb if (pre and pre[-1] == c) else tot) + lp[c]
          cands.append((v, slot * C + c, child, ninf, v))
      cands = [x for x in cands if x[0] > ninf]
      cands.sort(key=lambda x: (-x[0], x[1]))
      beams = [(pre, pb, pl) for _, _, pre, pb, pl in cands[:beam_width]]
    out.append([int(i) for i in beams[0][0]])
    score[b, 0] = lse(beams[0][1], beams[0][2])
  return out, score


def decoded_to_sparse(id_lists):
  """Sparse form of the decoder output: indices [N,2] row-major, values [N] int64, shape [B,max]."""
  idx, vals = [], []
  for b, ids in enumerate(id_lists):
    for p, v in enumerate(ids):
      idx.append([b, p])
      vals.append(v)
  max_len = max([len(i) for i in id_lists] + [0])
  return (np.array(idx, dtype=np.int64).reshape(-1, 2), np.array(vals, dtype=np.int64),
          np.array([len(id_lists), max_len], dtype=np.int64))


# ----------------------------------------------------------------------------------------
# optimizer  (speech_model.py:77-82; Appendix A5, A6)
# ----------------------------------------------------------------------------------------
def clip_by_global_norm(grads, clip_norm=5.0):
  gn = math.sqrt(sum(float(np.sum(np.square(g, dtype=np.float64))) for g in grads))
  scale = clip_norm / max(gn, clip_norm)
  return [g * scale for g in grads], gn


def adam_tf_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-3):
  """tf.train.AdamOptimizer update at 1-based ``step``: eps OUTSIDE the bias correction."""
  m = beta1 * m + (1.0 - beta1) * g
  v = beta2 * v + (1.0 - beta2) * g * g
  lr_t = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
  p = p - lr_t * m / (np.sqrt(v) + eps)
  return p, m, v


# ----------------------------------------------------------------------------------------
# one training step of the whole path  (speech_model.py:53-82 via step(), :197-235)
# ----------------------------------------------------------------------------------------
def train_step(x, seq_lens, labels, params, layers, opt_state, lr=1e-4, max_grad_norm=5.0,
               update=True, store=None):
  """x [B,T,C] padded batch, seq_lens [B] unpadded frames, labels list of id lists.

  opt_state = dict(step=int, m=[(mF,mb)...], v=[...]).  Returns dict with avg_loss, loss,
  logits (time-major), grads (unclipped), grad_norm and the new params/opt_state.
  """
  logits, acts = wav2letter_forward(x, params, layers, keep=True, store=store)
  B = x.shape[0]
  loss, g_logits = ctc_loss_and_grad(logits, labels, np.asarray(seq_lens) // 2)
  avg_loss = float(np.mean(loss))
  grads = wav2letter_backward(acts, params, layers, g_logits / B, store=store)
  flat = [g for pair in grads for g in pair]
  clipped, gn = clip_by_global_norm(flat, max_grad_norm)
  out = dict(avg_loss=avg_loss, loss=loss, logits=logits, grads=grads, grad_norm=gn)
  if update:
    step = opt_state['step'] + 1
    new_params, new_m, new_v = [], [], []
    for i, (F, b) in enumerate(params):
      mF, mb = opt_state['m'][i]
      vF, vb = opt_state['v'][i]
      F2, mF2, vF2 = adam_tf_step(F, clipped[2 * i], mF, vF, step, lr)
      b2, mb2, vb2 = adam_tf_step(b, clipped[2 * i + 1], mb, vb, step, lr)
      new_params.append((F2, b2))
      new_m.append((mF2, mb2))
      new_v.append((vF2, vb2))
    out['params'] = new_params
    out['opt_state'] = dict(step=step, m=new_m, v=new_v)
  return out


def zero_opt_state(params):
  return dict(step=0, m=[(np.zeros_like(F), np.zeros_like(b)) for F, b in params],
              v=[(np.zeros_like(F), np.zeros_like(b)) for F, b in params])


# ----------------------------------------------------------------------------------------
# evaluation statistics  (speecht/evaluation.py:40-49: editdistance.eval on chars / words)
# ----------------------------------------------------------------------------------------
def levenshtein(a, b):
  """Plain Levenshtein distance between two sequences (what ``editdistance.eval`` returns)."""
  a, b = list(a), list(b)
  prev = list(range(len(b) + 1))
  for i, ca in enumerate(a, 1):
    cur = [i]
    for j, cb in enumerate(b, 1):
      cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
    prev = cur
  return prev[-1]


# ----------------------------------------------------------------------------------------
# deterministic synthetic workload  
# ----------------------------------------------------------------------------------------
def synthetic_audio(utt_index, n_samples):
  rng = np.random.default_rng(1234 + utt_index)
  return np.clip(0.1 * rng.standard_normal(n_samples), -1.0, 1.0).astype(np.float32)


def synthetic_labels(utt_index, seconds, max_frames):
  """ids ~ U{0..27}, L = round(15*seconds), shrunk until L + repeats <= max_frames."""
  rng = np.random.default_rng(4321 + utt_index)
  L = int(round(15 * seconds))
  ids = rng.integers(0, VOCAB_SIZE, L).tolist()
  while ids and len(ids) + sum(1 for i in range(1, len(ids)) if ids[i] == ids[i - 1]) > max_frames:
    ids.pop()
  return ids
