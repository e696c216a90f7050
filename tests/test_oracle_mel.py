"""Pins the oracle's feature chain (preprocessing.py:29-58) against scipy and closed forms."""
import numpy as np
import pytest
import scipy.signal

from oracle import w2l_oracle as O


def test_hann_matches_scipy():
  np.testing.assert_allclose(O.hann_periodic(512), scipy.signal.get_window('hann', 512, fftbins=True),
                             atol=1e-15)


def test_stft_power_vs_scipy():
  y = O.synthetic_audio(0, 16000).astype(np.float64)
  S = O.stft_power(y, 512, 160)
  assert S.shape == (257, 1 + 16000 // 160)
  ypad = np.pad(y, 256, mode='reflect')
  _, _, Z = scipy.signal.stft(ypad, window='hann', nperseg=512, noverlap=512 - 160, nfft=512,
                              boundary=None, padded=False, return_onesided=True)
  Z = Z * scipy.signal.get_window('hann', 512).sum()     # undo scipy's spectrum scaling
  np.testing.assert_allclose(S, np.abs(Z) ** 2, rtol=1e-9, atol=1e-12)


def test_pure_tone_peak_bin():
  sr, f0 = 16000, 1000.0
  y = np.sin(2 * np.pi * f0 * np.arange(sr) / sr)
  S = O.stft_power(y, 512, 160)
  assert int(np.argmax(S[:, 50])) == round(f0 / (sr / 512))


@pytest.mark.parametrize('sr,n_mels', [(16000, 80), (22050, 128), (16000, 40)])
def test_mel_filterbank_slaney_properties(sr, n_mels):
  fb = O.mel_filterbank(sr, 512, n_mels)
  assert fb.shape == (n_mels, 257) and np.all(fb >= 0)
  # Slaney scale: linear below 1 kHz with 200/3 Hz per mel, so 1000 Hz == mel 15
  assert O.hz_to_mel_slaney(1000.0) == pytest.approx(15.0)
  assert O.mel_to_hz_slaney(O.hz_to_mel_slaney(4321.0)) == pytest.approx(4321.0)
  # area normalisation: the continuous triangle has unit area, i.e. sum(fb)*df ~= 1 where the
  # filter spans several bins
  df = sr / 512
  wide = fb[n_mels // 2:]
  np.testing.assert_allclose(wide.sum(axis=1) * df, 1.0, rtol=0.08)
  # peaks move monotonically upward
  peaks = np.argmax(fb[n_mels // 4:], axis=1)
  assert np.all(np.diff(peaks) >= 0)


def test_power_to_db_and_normalize():
  S = np.array([[1.0, 1e-3], [1e-12, 10.0]])
  D = O.power_to_db(S)
  np.testing.assert_allclose(D, [[-10.0, -40.0], [-80.0, 0.0]])   # floor at max-80; amin 1e-10
  n = O.normalize(np.array([[1.0, 2.0], [3.0, 6.0]]))
  assert abs(n.mean()) < 1e-15 and n.std() == pytest.approx(1.0)


def test_calc_power_spectrogram_shape_and_stats():
  y = O.synthetic_audio(3, 32000)
  f = O.calc_power_spectrogram(y, 16000, n_mels=80)
  assert f.shape == (201, 80)
  assert abs(f.mean()) < 1e-12 and f.std() == pytest.approx(1.0)


def test_delta_matches_librosa05_lfilter_formulation():
  """oracle.delta_lfilter vs the librosa 0.5.x recipe written with scipy.signal.lfilter (edge pad by 9,
  FIR [4..-4]/60 from rest, ``order`` passes over the padded axis, cut [-5-T:-5])."""
  rng = np.random.default_rng(0)
  x = rng.standard_normal((13, 50))
  window = np.arange(4.0, -5.0, -1.0)
  window /= np.sum(window ** 2)
  for order in (1, 2):
    d = np.pad(x, [(0, 0), (9, 9)], mode='edge')
    for _ in range(order):
      d = scipy.signal.lfilter(window, 1, d, axis=-1)
    np.testing.assert_allclose(O.delta_lfilter(x, order=order), d[:, -5 - 50:-5], atol=1e-14)
  # order 1 is the textbook regression delta with clamped edges
  t = 20
  ref = sum(k * (x[:, t + k] - x[:, t - k]) for k in range(1, 5)) / 60.0
  np.testing.assert_allclose(O.delta_lfilter(x)[:, t], ref, atol=1e-14)
  assert np.allclose(O.delta_lfilter(np.ones((2, 30))), 0.0)


def test_dct_basis_is_orthonormal_dct2():
  import scipy.fftpack
  rng = np.random.default_rng(1)
  S = rng.standard_normal((128, 7))
  np.testing.assert_allclose(O.dct_basis(13, 128) @ S, scipy.fftpack.dct(S, axis=0, type=2, norm='ortho')[:13], atol=1e-12)


def test_calc_mfccs_shape_and_block_normalisation():
  f = O.calc_mfccs(O.synthetic_audio(0, 16000), 16000)
  assert f.shape == (101, 39)
  for b in range(3):
    blk = f[:, 13 * b:13 * (b + 1)]
    assert abs(blk.mean()) < 1e-12 and abs(blk.std() - 1.0) < 1e-12
