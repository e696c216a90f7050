"""CPU tests of the host-side mirror of the reference API: batch/label layout, input pipeline,
evaluation statistics, CLI flags, corpus cache format, and the data-parallel all-reduce (gloo)."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from oracle import w2l_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_batch_and_label_layout_matches_oracle():
  from speecht_amd.speech_input import BaseInputLoader, sparse_to_label_lists
  rng = np.random.default_rng(0)
  items = [rng.standard_normal((t, 5)) for t in (7, 3, 9, 1)]
  labels = [[1, 2, 3], [], [27, 26], [4]]
  loader = BaseInputLoader(5)
  x, lens, max_t = loader._get_inputs_feed_item(items)
  xr, lr, mr = O.pad_batch(items, 5)
  np.testing.assert_allclose(x, xr.astype(np.float32))
  assert x.dtype == np.float32 and lens.tolist() == lr.tolist() and max_t == mr == 9
  sp = loader._get_labels_feed_item(labels, max_t)
  idx, vals, shape = O.sparse_labels(labels, max_t)
  np.testing.assert_array_equal(sp.indices, idx)
  np.testing.assert_array_equal(sp.values, vals)
  np.testing.assert_array_equal(sp.dense_shape, shape)       # dense_shape uses the INPUT max_time
  assert [a.tolist() for a in sparse_to_label_lists(sp)] == labels
  # unsorted sparse entries are brought to row-major order first
  perm = np.random.default_rng(0).permutation(len(sp.values))
  shuffled = type(sp)(sp.indices[perm], sp.values[perm], sp.dense_shape)
  assert [a.tolist() for a in sparse_to_label_lists(shuffled)] == labels


def test_input_batch_loader_drops_partial_batch_and_signals_end():
  from speecht_amd.speech_input import Coordinator, InputBatchLoader, OutOfRangeError

  def gen():
    for i in range(7):                                           # 7 samples, batch 3 -> 2 batches
      yield np.full((4 + i, 2), float(i)), [i % 28]
  loader = InputBatchLoader(2, 3, gen)
  coord = Coordinator()
  loader.start_threads(None, coord)
  seen = []
  with pytest.raises(OutOfRangeError):
    while True:
      x, lens, labels = loader.dequeue()
      seen.append((x.shape, lens.tolist(), labels.values.tolist()))
  assert seen == [((3, 6, 2), [4, 5, 6], [0, 1, 2]), ((3, 9, 2), [7, 8, 9], [3, 4, 5])]
  coord.request_stop(); coord.join()
  # max_steps caps the number of batches
  loader = InputBatchLoader(2, 1, gen, max_steps=2)
  loader.start_threads(None, Coordinator())
  assert loader.dequeue()[1].tolist() == [4] and loader.dequeue()[1].tolist() == [5]
  with pytest.raises(OutOfRangeError):
    loader.dequeue()


def test_single_input_loader_protocol():
  from speecht_amd.speech_input import SingleInputLoader
  loader = SingleInputLoader(3)
  with pytest.raises(ValueError):
    loader.get_feed_dict()
  loader.set_input(np.ones((5, 3)))
  x, lens, labels = loader.dequeue()
  assert x.shape == (1, 5, 3) and lens.tolist() == [5] and labels is None
  assert loader.speech_input is None                            # consumed, like the reference


def test_editdistance_and_eval_statistics():
  from speecht_amd import editdistance
  from speecht_amd.evaluation import EvalStatistics, Evaluation
  from speecht_amd.speech_input import SparseTensorValue
  rng = np.random.default_rng(1)
  for _ in range(50):
    a = ''.join(rng.choice(list('abc '), rng.integers(0, 12)))
    b = ''.join(rng.choice(list('abc '), rng.integers(0, 12)))
    assert editdistance.eval(a, b) == O.levenshtein(a, b)
    assert editdistance.eval(a.split(), b.split()) == O.levenshtein(a.split(), b.split())
  stats = EvalStatistics()
  stats.track_decoding('the cat sat', 'the cat sat on')
  assert (stats.letter_edit_distance, stats.word_edit_distance) == (3, 1)
  assert stats.letter_error_rate == pytest.approx(3 / 14) and stats.word_error_rate == pytest.approx(0.25)
  stats.track_decoding('x', 'y z')
  assert stats.global_letter_edit_distance == pytest.approx((3 + 3) / 2)
  assert stats.global_word_error_rate == pytest.approx((0.25 + 1.0) / 2)
  # extract_decoded_ids: faithful to the reference, including the empty-row quirk
  sp = SparseTensorValue(np.array([[0, 0], [0, 1], [2, 0]]), np.array([5, 6, 7]), np.array([3, 2]))
  assert [list(x) for x in Evaluation.extract_decoded_ids(sp)] == [[5, 6], [7]]   # row 1 (empty) vanishes


def test_run_step_pairs_like_the_reference_by_default():
  """evaluation.py:144-151: labels and decodings are walked with extract_decoded_ids in lock-step, so an utterance
  that decodes to the empty string shifts the later pairings and the walk ends when the decodings run out first (the
  reference: a bare StopIteration out of run_step; here the same abort as a RuntimeError that says why, after a
  warning that names the empty decodings); flags.pair_by_row (extension, --pair-by-row) pairs by batch row."""
  import types
  from speecht_amd.evaluation import Evaluation, EvalStatistics
  from speecht_amd.speech_input import SparseTensorValue
  from speecht_amd import vocabulary as V
  ids = V.sentence_to_ids
  labels = [ids('ab'), ids('cd'), ids('ef')]
  decoded = [ids('ab'), [], ids('ef')]                 # row 1 decodes to nothing

  def sparse(rows):
    idx = [[b, p] for b, r in enumerate(rows) for p in range(len(r))]
    return SparseTensorValue(np.array(idx).reshape(-1, 2), np.array([v for r in rows for v in r]), np.array([len(rows), 2]))
  model = types.SimpleNamespace(global_step=types.SimpleNamespace(eval=lambda: 0),
                                step=lambda sess, **kw: [np.float32(1.0), [sparse(decoded)], sparse(labels)])
  ev = Evaluation.__new__(Evaluation)
  ev.flags = types.SimpleNamespace(pair_by_row=False)
  stats = EvalStatistics()
  with pytest.raises(RuntimeError, match='ran out of decodings'):      # label 'ef' finds no third decoding
    ev.run_step(model, None, stats, save=False, verbose=False)
  assert stats.decodings_counter == 2 and stats.sum_letter_edit_distance == 0 + 2     # 'ab'~'ab', 'cd'~'ef'
  ev.flags.pair_by_row = True
  stats = EvalStatistics()
  ev.run_step(model, None, stats, save=False, verbose=False)
  assert stats.decodings_counter == 3 and stats.sum_letter_edit_distance == 0 + 2 + 0  # 'cd'~'' costs 2


def test_cli_flags_and_derived_values():
  import importlib.machinery
  import importlib.util
  loader = importlib.machinery.SourceFileLoader('speecht_cli', os.path.join(ROOT, 'speecht-cli'))
  cli = importlib.util.module_from_spec(importlib.util.spec_from_loader('speecht_cli', loader))
  loader.exec_module(cli)
  _, f = cli.parse(['train'])
  assert (f.feature_type, f.batch_size, f.run_name, f.learning_rate, f.max_gradient_norm,
          f.steps_per_checkpoint, f.run_type, f.run_train_dir) == ('power', 64, 'noname', 1e-4, 5.0, 1000, 'train',
                                                                     'train/noname')
  _, f = cli.parse(['evaluate', '--dev', '--step-count', '1', '--run-name', 'x', '--train-dir', 't', '--no-save'])
  assert (f.dataset, f.run_type, f.step_count, f.should_save, f.run_train_dir) == ('dev', 'dev', 1, False, 't/x')
  _, f = cli.parse(['evaluate'])
  assert f.dataset == 'test' and f.run_type == 'test' and f.should_save is True
  _, f = cli.parse(['preprocess', '--train-only', '--mfcc'])
  assert f.train_only and f.feature_type == 'mfcc' and f.run_type == 'other'


def test_corpus_reader_transcripts_and_npz_cache(tmp_path, golden_dir):
  """Same assertions as the reference's test__get_transcript_entries / test_load_samples
  (speecht/tests/test_speechCorpusReader.py:25-35,62-73) on its own transcript fixture."""
  from speecht_amd.preprocessing import SpeechCorpusReader
  from speecht_amd import vocabulary
  data = tmp_path / 'data'
  (data / 'train').mkdir(parents=True)
  src = open(os.path.join(golden_dir, '1089-134686.trans.txt')).read()
  (data / 'train' / '1089-134686.trans.txt').write_text(src)
  reader = SpeechCorpusReader(str(data))
  entries = list(reader._get_transcript_entries(str(data / 'train')))
  assert entries[0][0] == '1089-134686-0000' and entries[0][1].startswith('HE HOPED THERE WOULD BE STEW FOR DINNER')
  assert entries[-1][0] == '1089-134686-0037' and len(entries) == 38
  feats = np.random.default_rng(0).standard_normal((33, 13)).astype(np.float32)
  out = data / 'preprocessed' / 'train'
  out.mkdir(parents=True)
  np.savez(str(out / '1089-134686-0037'), audio_fragments=feats, transcript=reader._transcript_dict['1089-134686-0037'])
  samples = list(reader.load_samples('train', feature_type='mfcc'))
  assert len(samples) == 1
  np.testing.assert_array_equal(samples[0][0], feats)
  assert vocabulary.ids_to_sentence(samples[0][1]) == entries[-1][1].lower()
  with pytest.raises(ValueError):
    list(reader.load_samples('dev', feature_type='power'))


def test_shard_range_and_buckets():
  from speecht_amd.data_parallel import default_buckets, shard_range
  assert [shard_range(256, r, 8) for r in (0, 7)] == [(0, 32), (224, 256)]
  with pytest.raises(ValueError):
    shard_range(10, 0, 4)
  offs = [(0, 10), (10, 20), (20, 120), (120, 140), (140, 145)]
  assert default_buckets([e - s for s, e in offs], offs) == [(3, 120, 145), (2, 20, 120), (0, 0, 20)]
  # the Wav2Letter shape: the big layer is the ninth; the eight layers below it go in three pieces, the bottom layer alone last
  sizes = [10, 4, 4, 4, 4, 4, 4, 4, 100, 30, 1]
  offs, o = [], 0
  for n in sizes:
    offs.append((o, o + n))
    o += n
  assert default_buckets(sizes, offs) == [(9, offs[9][0], offs[10][1]), (8, offs[8][0], offs[8][1]), (4, offs[4][0], offs[7][1]),
                                          (1, offs[1][0], offs[3][1]), (0, 0, offs[0][1])]


DP_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["ST_ROOT"])
from oracle import w2l_oracle as O
from tests import workloads as WL
from speecht_amd.data_parallel import GradientAllReducer, shard_range, all_reduce_mean_scalar
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
if os.environ.get("ST_FULL_DEPTH"):
  layers = WL.w2l_layers(6, width=8, fc=16)                      # eleven layers: the five-bucket schedule of the real model
  frames = [80, 66, 56, 80, 71, 80, 49, 80]                      # global batch 8
else:
  layers = [(7, 2, 6, 8, True), (5, 1, 8, 8, True), (1, 1, 8, 29, False)]
  frames = [40, 33, 28, 40]                                      # global batch 4
params = WL.xavier_params(layers, seed=3)
x, seq, labels = WL.make_batch(frames, 6, seed=4)
lo, hi = shard_range(len(labels), rank, world)

def flat_grads(xs, ss, ls, scale_batch):
  logits, acts = O.wav2letter_forward(xs, params, layers, keep=True)
  loss, g = O.ctc_loss_and_grad(logits, ls, np.asarray(ss) // 2)
  grads = O.wav2letter_backward(acts, params, layers, g / scale_batch)     # 1 / GLOBAL batch
  return np.concatenate([np.concatenate([gF.ravel(), gb.ravel()]) for gF, gb in grads]), loss

local, loss = flat_grads(x[lo:hi], seq[lo:hi], labels[lo:hi], len(labels))
sizes = [F.size + b.size for F, b in params]
offs = np.concatenate([[0], np.cumsum(sizes)])
ranges = [(int(offs[i]), int(offs[i + 1])) for i in range(len(sizes))]
flat = torch.tensor(local)
red = GradientAllReducer(flat, ranges)
if os.environ.get("ST_FULL_DEPTH"):
  assert [b[0] for b in red.buckets] == [9, 8, 4, 1, 0], red.buckets   # L9+L10 | L8 | L4..L7 | L1..L3 | L0, in back-prop order
  assert red.buckets[0][2] == ranges[-1][1] and red.buckets[-1][1] == 0
  assert sum(e - s for _, s, e in red.buckets) == flat.numel()      # the buckets tile the flat gradient exactly
for i in reversed(range(len(layers))):
  red.on_layer_done(i)
red.finish()
full, full_loss = flat_grads(x, seq, labels, len(labels))
assert np.allclose(flat.numpy(), full, rtol=1e-9, atol=1e-12), np.abs(flat.numpy() - full).max()
mean_loss = all_reduce_mean_scalar(float(np.mean(loss)), "cpu")
assert abs(mean_loss - float(np.mean(full_loss))) < 1e-9
# identical replicas: every rank ends with the same buffer
gathered = [torch.zeros_like(flat) for _ in range(world)]
dist.all_gather(gathered, flat)
assert all(torch.equal(gathered[0], g) for g in gathered)
dist.destroy_process_group()
print("rank", rank, "ok")
'''


@pytest.mark.parametrize('world,full_depth', [(2, False), (4, True)])
def test_data_parallel_allreduce_equals_single_rank_gloo(tmp_path, world, full_depth):
  """world_size 2 and 4 on CPU (gloo): bucketed SUM all-reduce of per-rank gradients scaled by
  1/global_batch == gradient of the concatenated batch; replicas end bit-identical.  The world-4 case runs the
  eleven-layer model, i.e. the five buckets of the real schedule (L9+L10, L8, L4..L7, L1..L3, L0)."""
  script = tmp_path / 'dp_worker.py'
  script.write_text(DP_WORKER)
  env = dict(os.environ, ST_ROOT=ROOT, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29611 + world), WORLD_SIZE=str(world))
  if full_depth:
    env['ST_FULL_DEPTH'] = '1'
  procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                            stderr=subprocess.STDOUT) for r in range(world)]
  outs = [p.communicate(timeout=240)[0].decode() for p in procs]
  for r, (p, o) in enumerate(zip(procs, outs)):
    assert p.returncode == 0 and 'ok' in o, 'rank {} failed:\n{}'.format(r, o)


def test_flac_ingest_matches_the_references_fixture_expectations(golden_dir):
  """The reference's only numeric pin on this path: ``librosa.load`` of its LibriSpeech fixture has shape
  (114881,) (speecht/tests/test_speechCorpusReader.py:45).  The FLAC decoder is pinned bit-exactly by the MD5
  signature inside the file (decode_flac verifies it), the resampler by the output length, a pure tone and
  agreement with scipy's polyphase resampler."""
  import scipy.signal
  from speecht_amd import audio_io
  from speecht_amd.preprocessing import load_audio
  path = os.path.join(golden_dir, '1089-134686-0037.flac')
  samples, rate, bps = audio_io.decode_flac(path)                # raises if the MD5 does not match
  assert samples.shape == (83360, 1) and rate == 16000 and bps == 16
  y, sr = load_audio(path)
  assert y.shape == (114881,) and sr == 22050 and y.dtype == np.float32
  native = samples[:, 0] / 32768.0
  poly = scipy.signal.resample_poly(native, 441, 320)
  n = min(len(poly), len(y))
  assert np.corrcoef(y[:n], poly[:n])[0, 1] > 0.9999
  tone = np.sin(2 * np.pi * 1000.0 * np.arange(16000) / 16000.0)
  z = audio_io.resample_kaiser_best(tone, 16000, 22050)
  assert len(z) == 22050
  assert np.max(np.abs(z - np.sin(2 * np.pi * 1000.0 * np.arange(22050) / 22050.0))[200:-200]) < 1e-6
  with pytest.raises(audio_io.FlacError):
    audio_io.decode_flac(b'fLaC' + bytes(60))


def test_saver_index_is_relative_atomic_and_pruned(tmp_path, monkeypatch):
  """tf.train.Saver semantics kept by the .npz saver: the index names files relative to the checkpoint directory
  (restore works from another cwd / after moving the directory), keeps the last 5, never leaves temp files."""
  import shutil
  import types
  import torch
  from speecht_amd import speech_model as SM
  eng = types.SimpleNamespace(params=torch.arange(8.0), adam_m=torch.zeros(8), adam_v=torch.ones(8), n_flat=8,
                              layers=[types.SimpleNamespace(width=1, stride=1, cin=2, cout=2, relu=True)],
                              device='cpu', step_count=0, mark_weights_changed=lambda: None)
  model = types.SimpleNamespace(engine=eng, global_step=SM._Scalar(0, int), learning_rate=SM._Scalar(1e-3, float), _rank=0)
  saver = SM._Saver(model)
  run = tmp_path / 'train' / 'noname'
  run.mkdir(parents=True)
  monkeypatch.chdir(tmp_path)
  for step in range(1, 8):
    model.global_step.value = step
    saver.save(None, os.path.join('train', 'noname', 'speechT.ckpt'), global_step=model.global_step)
  files = sorted(os.listdir(run))
  assert files == ['checkpoint'] + ['speechT.ckpt-%d.npz' % s for s in range(3, 8)]
  index = json.load(open(run / 'checkpoint'))
  assert index['model_checkpoint_path'] == 'speechT.ckpt-7.npz'
  # another cwd and a moved directory still resolve
  monkeypatch.chdir('/')
  moved = tmp_path / 'elsewhere'
  shutil.move(str(run), str(moved))
  path = SM.latest_checkpoint(str(moved))
  assert path == os.path.join(str(moved), 'speechT.ckpt-7.npz')
  eng.params = torch.zeros(8)
  saver.restore(None, path)
  assert torch.equal(eng.params, torch.arange(8.0)) and model.global_step.eval() == 7 and eng.step_count == 7
  # a non-zero rank of a data-parallel job writes nothing
  model._rank = 1
  model.global_step.value = 9
  saver.save(None, os.path.join(str(moved), 'speechT.ckpt'), global_step=model.global_step)
  assert not os.path.exists(moved / 'speechT.ckpt-9.npz')
