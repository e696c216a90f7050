"""Seeded workloads shared by the parity tests, the golden generator, smoke() and bench.py
(SURVEY 8(d) synthetic inputs).  Pure numpy; imports nothing from the oracle or the product."""
import math

import numpy as np

VOCAB = 28


def w2l_layers(input_size, num_classes=29, width=250, fc=2000):
  """(filter_width, stride, cin, cout, relu) of Wav2LetterModel (speech_model.py:275-292);
  ``width``/``fc`` shrink the channel counts for small parity cases (250/2000 in the model)."""
  layers = [(48, 2, input_size, width, True)]
  layers += [(7, 1, width, width, True)] * 7
  layers += [(32, 1, width, fc, True), (1, 1, fc, fc, True), (1, 1, fc, num_classes, False)]
  return layers


def xavier_params(layers, seed=42, bias_range=0.05, dtype=np.float64):
  rng = np.random.default_rng(seed)
  params = []
  for (W, s, cin, cout, relu) in layers:
    limit = math.sqrt(6.0 / (W * cin + W * cout))
    F = rng.uniform(-limit, limit, size=(W, cin, cout)).astype(dtype)
    b = (rng.uniform(-bias_range, bias_range, size=(cout,)) if bias_range else np.zeros(cout)).astype(dtype)
    params.append((F, b))
  return params


def synthetic_features(seed, frames, n_feat):
  """z-normalised-looking features (what calc_power_spectrogram emits) without running the FFT."""
  rng = np.random.default_rng(9000 + seed)
  return rng.standard_normal((frames, n_feat))


def make_labels(seed, length, max_frames):
  rng = np.random.default_rng(4321 + seed)
  ids = rng.integers(0, VOCAB, length).tolist()
  while ids and len(ids) + sum(1 for i in range(1, len(ids)) if ids[i] == ids[i - 1]) > max_frames:
    ids.pop()
  return ids


def make_batch(frames_per_utt, n_feat, chars_per_frame=0.15, seed=0):
  """Zero-padded batch like BaseInputLoader._get_inputs_feed_item (speech_input.py:37-45)."""
  B, max_t = len(frames_per_utt), max(frames_per_utt)
  x = np.zeros((B, max_t, n_feat))
  labels = []
  for i, t in enumerate(frames_per_utt):
    x[i, :t] = synthetic_features(seed * 1000 + i, t, n_feat)
    labels.append(make_labels(seed * 1000 + i, int(round(chars_per_frame * t)), t // 2))
  return x, np.array(frames_per_utt, dtype=np.int64), labels


def small_train_case():
  """B=3 ragged, odd/even lengths, 16-mel, narrow channels: seconds on the float64 oracle."""
  layers = w2l_layers(16, width=24, fc=40)
  params = xavier_params(layers, seed=7)
  x, seq_lens, labels = make_batch([61, 48, 37], 16, seed=1)
  labels[1] = [2, 2, 2, 5]            # repeats
  labels[2] = []                      # empty transcript
  return dict(layers=layers, params=params, x=x, seq_lens=seq_lens, labels=labels)
