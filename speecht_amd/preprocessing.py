"""Feature extraction and the preprocessed-corpus reader (mirror of speecht/preprocessing.py).

``calc_power_spectrogram`` keeps the reference signature (preprocessing.py:36) but runs the whole
chain -- STFT, mel projection, power_to_db, z-normalisation, transpose -- in the HIP kernels of
csrc/melspec.hip.  Only the filterbank *constants* are built on the host (float64 numpy, cached
per (samplerate, n_fft, n_mels)), following librosa.filters.mel(htk=False, norm=1).
"""
import ctypes
import fnmatch
import functools
import logging
import math
import os
import random

import numpy as np

from . import vocabulary


@functools.lru_cache(maxsize=16)
def mel_filterbank(samplerate, n_fft, n_mels):
  """Slaney-scale, area-normalised triangular filters [n_mels, 1 + n_fft//2] (float64)."""
  f_sp, min_log_hz = 200.0 / 3, 1000.0
  min_log_mel, logstep = min_log_hz / f_sp, math.log(6.4) / 27.0

  def to_mel(hz):
    return min_log_mel + math.log(hz / min_log_hz) / logstep if hz >= min_log_hz else hz / f_sp

  def to_hz(mel):
    return np.where(mel >= min_log_mel, min_log_hz * np.exp(logstep * (mel - min_log_mel)), f_sp * mel)

  edges = to_hz(np.linspace(to_mel(0.0), to_mel(samplerate / 2.0), n_mels + 2))
  bins = np.linspace(0.0, samplerate / 2.0, 1 + n_fft // 2)
  rising = (bins[None, :] - edges[:-2, None]) / (edges[1:-1] - edges[:-2])[:, None]
  falling = (edges[2:, None] - bins[None, :]) / (edges[2:] - edges[1:-1])[:, None]
  tri = np.clip(np.minimum(rising, falling), 0.0, None)
  return tri * (2.0 / (edges[2:] - edges[:-2]))[:, None]


def calc_power_spectrogram_batch(audio_list, samplerate, n_mels=128, n_fft=512, hop_length=160, device='cuda:0'):
  """Features for several utterances in one launch pair; returns a list of [time, n_mels] float32
  arrays.  Utterances are concatenated in HBM; no padding, no per-utterance launches."""
  import torch
  from . import _lib
  dev = torch.device(device)
  lens = np.array([len(a) for a in audio_list], dtype=np.int64)
  if lens.min() <= n_fft // 2:
    raise ValueError('utterances must be longer than n_fft/2 samples (reflect padding)')
  s_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
  frames = 1 + lens // hop_length
  f_off = np.concatenate([[0], np.cumsum(frames)]).astype(np.int64)
  total = int(f_off[-1])
  audio = torch.as_tensor(np.concatenate([np.asarray(a, dtype=np.float32) for a in audio_list])).to(dev)
  basis = torch.as_tensor(mel_filterbank(float(samplerate), n_fft, n_mels).astype(np.float32)).contiguous().to(dev)
  d_soff, d_foff = torch.as_tensor(s_off).to(dev), torch.as_tensor(f_off).to(dev)
  out = torch.empty(total * n_mels, dtype=torch.float32, device=dev)
  lib = _lib.load()
  ws_bytes = lib.st_melspec_ws(len(audio_list), total, n_mels)
  ws = torch.empty(ws_bytes // 4 + 64, dtype=torch.float32, device=dev)
  P = lambda t: ctypes.c_void_p(t.data_ptr())
  _lib.call('st_melspec_f32', P(audio), P(d_soff), len(audio_list), int(lens.max()), P(basis), n_mels, n_fft,
            hop_length, P(d_foff), total, P(out), P(ws), ws.numel() * 4,
            ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
  host = out.view(total, n_mels).cpu().numpy()
  return [host[f_off[i]:f_off[i + 1]] for i in range(len(audio_list))]


def normalize(values):
  """(values - mean) / std over all elements (preprocessing.py:29-33); host helper kept for the
  API -- the device chain fuses this step."""
  values = np.asarray(values)
  return (values - values.mean()) / values.std()


def calc_power_spectrogram(audio_data, samplerate, n_mels=128, n_fft=512, hop_length=160):
  """Same contract as the reference (preprocessing.py:36-58): [time, n_mels]."""
  return calc_power_spectrogram_batch([audio_data], samplerate, n_mels, n_fft, hop_length)[0]


def calc_mfccs_batch(audio_list, samplerate, n_mfcc=13, n_fft=512, hop_length=160, device='cuda:0'):
  """MFCC + delta + delta-delta features for several utterances in one launch sequence; returns a list
  of [time, 3 * n_mfcc] float32 arrays (each block z-normalised per utterance)."""
  import torch
  from . import _lib
  n_mels = 128                                   # librosa.feature.mfcc's melspectrogram default
  dev = torch.device(device)
  lens = np.array([len(a) for a in audio_list], dtype=np.int64)
  if lens.min() <= n_fft // 2:
    raise ValueError('utterances must be longer than n_fft/2 samples (reflect padding)')
  s_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
  f_off = np.concatenate([[0], np.cumsum(1 + lens // hop_length)]).astype(np.int64)
  total = int(f_off[-1])
  audio = torch.as_tensor(np.concatenate([np.asarray(a, dtype=np.float32) for a in audio_list])).to(dev)
  basis = torch.as_tensor(mel_filterbank(float(samplerate), n_fft, n_mels).astype(np.float32)).contiguous().to(dev)
  d_soff, d_foff = torch.as_tensor(s_off).to(dev), torch.as_tensor(f_off).to(dev)
  out = torch.empty(total * 3 * n_mfcc, dtype=torch.float32, device=dev)
  ws_bytes = _lib.load().st_mfcc_ws(len(audio_list), total, n_mels, n_mfcc)
  ws = torch.empty(ws_bytes // 4 + 64, dtype=torch.float32, device=dev)
  P = lambda t: ctypes.c_void_p(t.data_ptr())
  _lib.call('st_mfcc_f32', P(audio), P(d_soff), len(audio_list), int(lens.max()), P(basis), n_mels, n_mfcc, n_fft,
            hop_length, P(d_foff), total, P(out), P(ws), ws.numel() * 4,
            ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
  host = out.view(total, 3 * n_mfcc).cpu().numpy()
  return [host[f_off[i]:f_off[i + 1]] for i in range(len(audio_list))]


def calc_mfccs(audio_data, samplerate, n_mfcc=13, n_fft=512, hop_length=160):
  """Same contract as the reference (preprocessing.py:61-84): [time, 3 * n_mfcc]."""
  return calc_mfccs_batch([audio_data], samplerate, n_mfcc, n_fft, hop_length)[0]


def iglob_recursive(directory, file_pattern):
  for root, _, names in os.walk(directory):
    for name in fnmatch.filter(names, file_pattern):
      yield os.path.join(root, name)


class SpeechCorpusReader:
  """Reads the preprocessed corpus cache written by `speecht-cli preprocess`: one ``.npz`` per
  utterance with ``audio_fragments [T, n_feat]`` and ``transcript [L]`` under
  ``<data>/preprocessed[-power]/<split>/`` (preprocessing.py:175-178, 199-206, 243-279).

  ``store_samples`` takes decoded waveforms from ``audio_loader(path) -> (float32 samples, samplerate)``;
  ``load_audio`` below bundles FLAC (with librosa's 22 050 Hz resampling, preprocessing.py:169), wav and npy.
  """

  def __init__(self, data_directory):
    self._data_directory = data_directory
    self._transcripts = None

  @staticmethod
  def _get_transcript_entries(transcript_directory):
    for path in iglob_recursive(transcript_directory, '*.trans.txt'):
      with open(path, 'r') as f:
        for line in f:
          yield line.rstrip('\n').split(' ', 1)

  @property
  def _transcript_dict(self):
    if not self._transcripts:
      self._transcripts = {k: vocabulary.sentence_to_ids(v)
                           for k, v in self._get_transcript_entries(self._data_directory)}
    return self._transcripts

  @staticmethod
  def _extract_audio_id(audio_file):
    return os.path.splitext(os.path.basename(audio_file))[0]

  def _get_directory(self, feature_type, sub_directory):
    name = 'preprocessed'
    if feature_type == calc_power_spectrogram or feature_type == 'power':
      name += '-power'
    return self._data_directory + '/' + name + '/' + sub_directory

  def generate_samples(self, directory, preprocess_fnc, audio_loader=None, pattern='*.flac'):
    """Generator of (audio_id, audio_fragments, transcript) straight from the audio files, without the
    cache (preprocessing.py:180-197).  ``audio_loader`` as in ``store_samples``; defaults to the bundled
    wav/npy decoder."""
    loader = audio_loader or load_audio
    for path in iglob_recursive(self._data_directory + '/' + directory, pattern):
      samples, rate = loader(path)
      audio_id = self._extract_audio_id(path)
      yield audio_id, preprocess_fnc(samples, rate), self._transcript_dict[audio_id]

  def store_samples(self, directory, preprocess_fnc, audio_loader=None, pattern='*.flac', batch=32):
    """Preprocess every audio file of ``<data>/<directory>`` and cache it as .npz.  The device
    extractor processes ``batch`` utterances per launch instead of the reference's process pool
    (preprocessing.py:229-241)."""
    if audio_loader is None:
      audio_loader = load_audio
    out_directory = self._get_directory(preprocess_fnc, directory)
    os.makedirs(out_directory, exist_ok=True)
    files = list(iglob_recursive(self._data_directory + '/' + directory, pattern))
    for i in range(0, len(files), batch):
      chunk = files[i:i + batch]
      loaded = [audio_loader(f) for f in chunk]
      rates = {sr for _, sr in loaded}
      if preprocess_fnc == calc_power_spectrogram and len(rates) == 1:
        feats = calc_power_spectrogram_batch([a for a, _ in loaded], rates.pop())
      elif preprocess_fnc == calc_mfccs and len(rates) == 1:
        feats = calc_mfccs_batch([a for a, _ in loaded], rates.pop())
      else:
        feats = [preprocess_fnc(a, sr) for a, sr in loaded]
      for f, feat in zip(chunk, feats):
        audio_id = self._extract_audio_id(f)
        np.savez(out_directory + '/' + audio_id, audio_fragments=feat, transcript=self._transcript_dict[audio_id])

  def load_samples(self, directory, max_size=False, loop_infinitely=False, limit_count=0, feature_type='mfcc', shuffle_seed=None):
    """Iterator over (audio_fragments, transcript), same semantics as preprocessing.py:243-279.
    ``shuffle_seed`` (not a reference argument): shuffle with a generator of its own seeded with it, over the SORTED file list,
    instead of the process-wide ``random`` state -- every rank of a data-parallel job must walk the samples in the same order."""
    load_directory = self._get_directory(feature_type, directory)
    if not os.path.exists(load_directory):
      raise ValueError('Directory {} does not exist'.format(load_directory))
    files = list(iglob_recursive(load_directory, '*.npz'))
    shuffle = random.shuffle
    if shuffle_seed is not None:
      files.sort()
      shuffle = random.Random(shuffle_seed).shuffle
    shuffle(files)
    if limit_count:
      files = files[:limit_count]
    while True:
      for path in files:
        with np.load(path) as data:
          frames = data['audio_fragments'].shape[0]
          if not max_size or frames <= max_size:
            yield data['audio_fragments'], data['transcript']
          else:
            logging.warning('Audio snippet too long: {}'.format(frames))
      if not loop_infinitely:
        break
      shuffle(files)


def load_audio(path):
  """Decode an audio file to (float32 mono samples in [-1, 1], samplerate).

  Bundled decoders: ``.flac`` (audio_io: FLAC decoding + librosa's implicit 22 050 Hz ``kaiser_best``
  resampling, i.e. ``librosa.load(path)`` of preprocessing.py:169), 16-bit PCM ``.wav`` (stdlib ``wave``,
  native rate) and raw ``.npy`` arrays (assumed 16 kHz).
  """
  ext = os.path.splitext(path)[1].lower()
  if ext == '.flac':
    # what the reference does for LibriSpeech: librosa.load(path) -> mono, resampled to 22 050 Hz
    from . import audio_io
    return audio_io.librosa_load(path)
  if ext == '.npy':
    return np.load(path).astype(np.float32), 16000
  if ext == '.wav':
    import wave
    with wave.open(path, 'rb') as w:
      if w.getsampwidth() != 2:
        raise ValueError('{}: only 16-bit PCM wav is supported'.format(path))
      data = np.frombuffer(w.readframes(w.getnframes()), dtype='<i2').astype(np.float32) / 32768.0
      if w.getnchannels() > 1:
        data = data.reshape(-1, w.getnchannels()).mean(axis=1)
      return data, w.getframerate()
  raise RuntimeError('{}: no decoder bundled for {} files'.format(path, ext or 'extension-less'))


class Preprocessing:
  """`speecht-cli preprocess` (preprocessing.py:282-311).  Corpus download (corpus.py) is network
  I/O and out of scope: the audio must already be under <data_dir>/{train,test,dev}."""

  AUDIO_PATTERNS = ('*.flac', '*.wav', '*.npy')

  def __init__(self, flags):
    self.flags = flags

  def run(self):
    corpus_reader = SpeechCorpusReader(self.flags.data_dir)
    if self.flags.feature_type == 'power':
      preprocess_fnc = calc_power_spectrogram
    elif self.flags.feature_type == 'mfcc':
      preprocess_fnc = calc_mfccs
    else:
      raise ValueError('Feature type must be mfcc or power.')
    preprocess_all = not (self.flags.train_only or self.flags.test_only or self.flags.dev_only)
    for split, only, title in (('train', self.flags.train_only, 'training'), ('test', self.flags.test_only, 'test'),
                               ('dev', self.flags.dev_only, 'development')):
      if only or preprocess_all:
        if not os.path.isdir(os.path.join(self.flags.data_dir, split)):
          print('Skipping {} data: {}/{} does not exist'.format(title, self.flags.data_dir, split))
          continue
        print('Preprocessing {} data'.format(title))
        for pattern in self.AUDIO_PATTERNS:
          corpus_reader.store_samples(split, preprocess_fnc, audio_loader=load_audio, pattern=pattern)
