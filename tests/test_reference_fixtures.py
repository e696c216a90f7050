"""Consumers of tests/golden/ref_{mel,logits,ctc,adam}.npz -- numbers recorded by RUNNING THE REFERENCE
(scripts/make_reference_fixtures.py: its own preprocessing / speech_input / speech_model modules under TensorFlow 1.x
+ librosa).  Those files cannot be produced in the build container (neither package exists here), so until
someone runs the generator in such an environment every test here SKIPS and the oracle stays "parity unpinned".  The
day the files exist these tests are what turns the pin green: the oracle against the reference on the CPU, the HIP
path against the reference on the GPU (tolerances of DESIGN 5: logits 1e-4, CTC loss 1e-4 relative, gradients 2e-4 of
the tensor max, features 1e-3, greedy ids identical).

``test_consumers_run_on_a_dry_run_fixture_set`` keeps the consumers themselves honest meanwhile: it generates the same
files from the ORACLE (``--dry-run``; marked as such, pinning nothing) and runs every CPU check on them.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import w2l_oracle as O
from tests import workloads as WL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
FILES = ('ref_mel.npz', 'ref_logits.npz', 'ref_ctc.npz', 'ref_adam.npz')
SKIP = ('reference fixtures absent: run scripts/make_reference_fixtures.py where TensorFlow 1.x + librosa + /root/reference '
        'exist (parity stays unpinned until then)')


def load(directory):
  if not all(os.path.exists(os.path.join(directory, f)) for f in FILES):
    return None
  return {f[4:-4]: dict(np.load(os.path.join(directory, f), allow_pickle=False)) for f in FILES}


def rel(a, b):
  return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))) / (np.max(np.abs(b)) + 1e-300))


def sample(a, n=4096):
  flat = np.asarray(a, dtype=np.float64).reshape(-1)
  idx = np.linspace(0, flat.size - 1, min(n, flat.size)).astype(np.int64)
  return np.array([flat.sum(), np.abs(flat).sum()]), flat[idx]


def case_of(fx):
  """The inputs of the recorded step, rebuilt from the seeds the fixture names (weights are not stored: 96 MB)."""
  lg = fx['logits']
  layers = WL.w2l_layers(int(lg['n_mels']))
  params = WL.xavier_params(layers, seed=int(lg['weights_seed']), dtype=np.float32)
  labels, at = [], 0
  for n in lg['label_lengths']:
    labels.append([int(v) for v in lg['label_values'][at:at + int(n)]])
    at += int(n)
  return layers, params, labels


def synthetic_clip(index, n):
  return np.clip(0.1 * np.random.default_rng(1234 + index).standard_normal(n), -1.0, 1.0).astype(np.float32)


# ---- the CPU checks (oracle vs recorded reference) --------------------------------------------------------------------
def check_features(fx, feature_tol=1e-3):
  for k in range(2):
    y = synthetic_clip(k, int(fx['mel']['samples_%d' % k])).astype(np.float64)
    for n_mels in (80, 128):
      want = fx['mel']['mel%d_%d' % (n_mels, k)]
      got = O.calc_power_spectrogram(y, 16000, n_mels=n_mels)
      assert got.shape == want.shape == (1 + len(y) // 160, n_mels)
      assert np.max(np.abs(got - want)) < feature_tol, (k, n_mels)
    want = fx['mel']['mfcc_%d' % k]
    got = O.calc_mfccs(y, 16000)
    assert got.shape == want.shape and np.max(np.abs(got - want)) < 2e-3, k


def check_batch_assembly(fx):
  layers, params, labels = case_of(fx)
  lg = fx['logits']
  frames = [int(t) for t in lg['frames']]
  x, seq, _ = WL.make_batch(frames, int(lg['n_mels']), seed=int(lg['batch_seed']))
  feats = [x[i, :t].astype(np.float32) for i, t in enumerate(frames)]
  px, pseq, max_t = O.pad_batch(feats, int(lg['n_mels']))                    # speech_input.py:27-45
  np.testing.assert_array_equal(px, lg['x'])
  np.testing.assert_array_equal(pseq, lg['seq'])
  idx, vals, shape = O.sparse_labels(labels, max_t)                          # speech_input.py:47-69
  np.testing.assert_array_equal(idx, lg['sparse_indices'])
  np.testing.assert_array_equal(vals, lg['sparse_values'])
  np.testing.assert_array_equal(shape, lg['sparse_shape'])


def check_step(fx):
  layers, params, labels = case_of(fx)
  lg, ctc, adam = fx['logits'], fx['ctc'], fx['adam']
  p64 = [(F.astype(np.float64), b.astype(np.float64)) for F, b in params]
  seq = lg['seq'].astype(np.int64)
  res = O.train_step(lg['x'].astype(np.float64), seq, labels, p64, layers, O.zero_opt_state(p64), lr=float(lg['lr']))
  assert res['logits'].shape == lg['logits'].shape
  assert np.max(np.abs(res['logits'] - lg['logits'])) < 1e-4                 # north_star: logits within 1e-4
  np.testing.assert_allclose(res['loss'], ctc['loss'], rtol=1e-4)            # CTC loss within 1e-4 (relative)
  assert abs(res['avg_loss'] - float(ctc['avg_loss'])) < 1e-4 * abs(float(ctc['avg_loss']))
  _, dl = O.ctc_loss_and_grad(res['logits'], labels, seq // 2)
  assert rel(dl / len(labels), ctc['dlogits']) < 2e-4
  dec, score = O.ctc_greedy_decode(res['logits'], seq // 2)
  d_idx, d_val, d_shape = O.decoded_to_sparse(dec)
  np.testing.assert_array_equal(d_idx, ctc['decoded_indices'])              # greedy strings bit-identical
  np.testing.assert_array_equal(d_val, ctc['decoded_values'])
  np.testing.assert_array_equal(d_shape, ctc['decoded_shape'])
  np.testing.assert_allclose(score, ctc['neg_sum_logits'], rtol=1e-5)
  assert res['grad_norm'] == pytest.approx(float(adam['grad_global_norm']), rel=1e-4)
  for i, (gF, gb) in enumerate(res['grads']):
    assert rel(gb, adam['grad_%d_bias' % i]) < 2e-4, i
    stats, picks = sample(gF)
    assert rel(picks, adam['grad_%d_filters_samples' % i]) < 2e-4, i
    assert abs(stats[1] - adam['grad_%d_filters_stats' % i][1]) < 2e-4 * adam['grad_%d_filters_stats' % i][1], i
  # every variable after one update: clip_by_global_norm(5) + Adam(eps = 1e-3) + global_step (speech_model.py:77-82)
  from speecht_amd.tf_checkpoint import reference_variable_names
  assert set(str(n) for n in adam['variable_names']) == reference_variable_names(len(layers))
  after = {'Variable': np.array(1), 'learning_rate': np.array(float(lg['lr'])),
           'training/beta1_power': np.array(0.9 ** 2), 'training/beta2_power': np.array(0.999 ** 2)}
  for i, ((F, b), (mF, mb), (vF, vb)) in enumerate(zip(res['params'], res['opt_state']['m'], res['opt_state']['v'])):
    base = 'convolution_layer_%d/' % i
    after.update({base + 'filters': F, base + 'bias': b, base + 'filters/Adam': mF, base + 'bias/Adam': mb,
                  base + 'filters/Adam_1': vF, base + 'bias/Adam_1': vb})
  for name, value in after.items():
    key = 'after__' + name.replace('/', '__')
    if key in adam:
      want, got = adam[key], np.asarray(value, np.float64)
    else:
      want, got = adam[key + '__samples'], sample(value)[1]
    # the first Adam step moves a weight by ~lr: compare the MOVEMENT for weights, values for everything else
    if name.endswith('filters') or name.endswith('bias'):
      i = int(name.split('_')[2].split('/')[0])
      start = params[i][0] if name.endswith('filters') else params[i][1]
      start = np.asarray(start, np.float64) if key in adam else sample(start)[1]
      assert np.max(np.abs((got - start) - (want - start))) < 1e-3 * float(lg['lr']) + 1.2e-7 * np.max(np.abs(start)), name
    else:
      assert rel(got, want) < 2e-4 or np.max(np.abs(want)) == 0.0, name


def test_oracle_matches_the_recorded_reference():
  fx = load(GOLD)
  if fx is None:
    pytest.skip(SKIP)
  assert str(fx['logits']['source']) == 'reference', 'tests/golden/ref_*.npz must come from the reference, not from --dry-run'
  check_features(fx)
  check_batch_assembly(fx)
  check_step(fx)


def test_consumers_run_on_a_dry_run_fixture_set(tmp_path):
  """The generator's file format and every CPU check above, exercised end to end on files written from the oracle
  (so the day real files arrive, a red test means a real difference, not a bug in the harness)."""
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'make_reference_fixtures.py'), '--dry-run', str(tmp_path)],
                     capture_output=True, text=True, timeout=900)
  assert r.returncode == 0, r.stdout + r.stderr
  fx = load(str(tmp_path))
  assert fx is not None and str(fx['adam']['source']) == 'oracle-dry-run'
  check_features(fx)
  check_batch_assembly(fx)
  check_step(fx)
  # and the generator refuses cleanly where the reference cannot run (this container: no tensorflow / librosa)
  import importlib.util
  if importlib.util.find_spec('tensorflow') is None or importlib.util.find_spec('librosa') is None:
    had = os.path.exists(os.path.join(GOLD, 'ref_mel.npz'))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'make_reference_fixtures.py')], capture_output=True, text=True,
                       timeout=300, cwd=str(tmp_path))
    assert r.returncode == 2 and 'nothing written' in r.stderr and os.path.exists(os.path.join(GOLD, 'ref_mel.npz')) == had


# ---- the GPU checks (HIP path vs recorded reference) -----------------------------------------------------------------
@pytest.mark.gpu
def test_hip_path_matches_the_recorded_reference():
  torch = pytest.importorskip('torch')
  fx = load(GOLD)
  if fx is None:
    pytest.skip(SKIP)
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  from speecht_amd import preprocessing as P
  from speecht_amd.engine import Wav2LetterEngine
  assert str(fx['logits']['source']) == 'reference'
  for k in range(2):
    y = synthetic_clip(k, int(fx['mel']['samples_%d' % k]))
    for n_mels in (80, 128):
      got = P.calc_power_spectrogram(y, 16000, n_mels=n_mels)
      assert np.max(np.abs(got - fx['mel']['mel%d_%d' % (n_mels, k)])) < 1e-3, (k, n_mels)
    assert np.max(np.abs(P.calc_mfccs(y, 16000) - fx['mel']['mfcc_%d' % k])) < 2e-3, k
  layers, params, labels = case_of(fx)
  lg, ctc, adam = fx['logits'], fx['ctc'], fx['adam']
  eng = Wav2LetterEngine(layers, device='cuda:0')
  eng.set_weights(params)
  eng.load_batch(lg['x'].astype(np.float32), lg['seq'])
  eng.set_labels(labels)
  eng.forward()
  eng.ctc_loss_grad(1.0 / len(labels))
  eng.backward()
  torch.cuda.synchronize()
  eng.check_ctc_status()
  assert np.max(np.abs(eng.logits_time_major().cpu().numpy() - lg['logits'])) < 1e-4
  np.testing.assert_allclose(eng.loss.cpu().numpy(), ctc['loss'], rtol=1e-4)
  dl = eng.dZ[-1].interior().cpu().numpy().transpose(1, 0, 2)
  assert rel(dl, ctc['dlogits']) < 2e-4
  dec, score = eng.greedy_decode()
  d_idx, d_val, d_shape = O.decoded_to_sparse(dec)
  np.testing.assert_array_equal(d_idx, ctc['decoded_indices'])
  np.testing.assert_array_equal(d_val, ctc['decoded_values'])
  for i, (gF, gb) in enumerate(eng.get_grads()):
    assert rel(gb, adam['grad_%d_bias' % i]) < 2e-4, i
    assert rel(sample(gF)[1], adam['grad_%d_filters_samples' % i]) < 2e-4, i
  eng.apply_update(lr=float(lg['lr']))
  torch.cuda.synchronize()
  assert float(eng.stats[0]) == pytest.approx(float(adam['grad_global_norm']), rel=1e-4)
  for i, ((F, b), (F0, b0)) in enumerate(zip(eng.get_weights(), params)):
    want = adam['after__convolution_layer_%d__bias' % i]
    assert np.max(np.abs((b - b0) - (want - b0))) < 1e-3 * float(lg['lr']) + 1.2e-7 * np.max(np.abs(b0)), i
    key = 'after__convolution_layer_%d__filters' % i
    want, got, start = ((adam[key], F, F0) if key in adam else (adam[key + '__samples'], sample(F)[1], sample(F0)[1]))
    assert np.max(np.abs((got - start) - (want - start))) < 1e-3 * float(lg['lr']) + 1.2e-7 * np.max(np.abs(start)), i
