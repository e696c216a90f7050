"""`torchrun --nproc-per-node N speecht-cli train` is data-parallel (reference loop:
speecht/training.py:44-98, speecht-cli:191-207).  CPU, gloo, world size 2: the CLI entry point itself is driven -- flags,
`Training.run`, the rank-aware `InputBatchLoader`, `SpeechModel.enable_data_parallel`, the bucketed gradient exchange with the
mean loss and the update gate riding in the first bucket, rank-0-only checkpoints / prints -- with the engine replaced by the
float64 oracle stand-in of tests/oracle_engine.py (the product engine needs a GPU and has no CPU path; the GPU form of this test
is tests/test_gpu_api.py::test_cli_train_under_a_launcher_is_data_parallel)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import importlib.machinery, importlib.util, os, sys
sys.path.insert(0, os.environ["ST_ROOT"])
from speecht_amd import speech_model
from tests.oracle_engine import OracleEngine

def _ensure_engine(self, sess):                    # the product engine needs a GPU: the oracle stand-in computes instead
  if self.engine is None:
    self.engine = OracleEngine(self._layer_specs)
  return self.engine
speech_model.SpeechModel._ensure_engine = _ensure_engine
loader = importlib.machinery.SourceFileLoader("speecht_cli", os.path.join(os.environ["ST_ROOT"], "speecht-cli"))
spec = importlib.util.spec_from_loader("speecht_cli", loader)
cli = importlib.util.module_from_spec(spec)
loader.exec_module(cli)
rc = cli.main(sys.argv[1:])
import torch.distributed as dist
if dist.is_initialized():
  dist.barrier()
  dist.destroy_process_group()
sys.exit(rc or 0)
'''


def make_corpus(directory, count, n_feat, seed=0):
  """Cached samples as `speecht-cli preprocess` leaves them (preprocessing.py:199-206): ragged lengths."""
  os.makedirs(directory, exist_ok=True)
  rng = np.random.default_rng(seed)
  for i in range(count):
    frames = int(rng.integers(30, 44))
    np.savez(os.path.join(directory, 'utt-{:03d}'.format(i)), audio_fragments=rng.standard_normal((frames, n_feat)),
             transcript=rng.integers(0, 26, 3))          # (letters only: a blank-only transcript has no words to rate)


def run_cli(tmp_path, name, world, batch, extra=()):
  script = tmp_path / 'cli_worker.py'
  script.write_text(WORKER)
  args = ['train', '--data-dir', str(tmp_path / 'data'), '--train-dir', str(tmp_path / ('train_' + name)), '--log-dir',
          str(tmp_path / ('log_' + name)), '--run-name', 'dp', '--batch-size', str(batch), '--device', 'cpu', '--seed', '11',
          '--steps-per-checkpoint', '2', '--max-steps', '4', '--learning-rate', '1e-3'] + list(extra)
  env = dict(os.environ, ST_ROOT=ROOT, ST_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29650 + world),
             OMP_NUM_THREADS='2')
  if world > 1:
    env['WORLD_SIZE'] = str(world)
  else:
    env.pop('WORLD_SIZE', None)
  procs = [subprocess.Popen([sys.executable, str(script)] + args, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), cwd=str(tmp_path),
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
  outs = [p.communicate(timeout=900) for p in procs]
  for r, (p, (out, err)) in enumerate(zip(procs, outs)):
    assert p.returncode == 0, 'rank {} of {} failed:\n{}\n{}'.format(r, world, out, err[-3000:])
  return [o for o, _ in outs]


@pytest.mark.timeout(1800)
def test_cli_train_world_two_equals_one_process_on_the_concatenated_batches(tmp_path):
  make_corpus(str(tmp_path / 'data' / 'preprocessed-power' / 'train'), 12, 6)
  single = run_cli(tmp_path, 'single', 1, 4)
  dp = run_cli(tmp_path, 'dp', 2, 2)
  # rank 0 alone speaks, and says what the single process says (same global batches, same mean losses to rounding)
  said = lambda out: [l for l in out.splitlines() if l.strip() and not l.startswith('[Gloo]')]      # (gloo's own banner aside)
  assert said(dp[1]) == [], dp[1]
  assert dp[0].count('Model saved') == single[0].count('Model saved') == 2
  lines = lambda out: [l for l in out.splitlines() if l.startswith('global step')]
  assert len(lines(dp[0])) == len(lines(single[0])) == 2
  for a, b in zip(lines(dp[0]), lines(single[0])):
    fa, fb = a.split(), b.split()
    assert fa[:6] == fb[:6]                                                  # global step N learning rate R
    assert abs(float(fa[-3]) - float(fb[-3])) < 1e-2, (a, b)                 # average loss (printed to 2 decimals)
  # one set of checkpoints per job, written by rank 0, and the same weights / Adam state as the single process
  ck = lambda name: np.load(str(tmp_path / ('train_' + name) / 'dp' / 'speechT.ckpt-4.npz'))
  a, b = ck('dp'), ck('single')
  assert int(a['global_step']) == int(b['global_step']) == 4
  for key in ('params', 'adam_m', 'adam_v'):
    scale = np.max(np.abs(b[key]))
    assert np.max(np.abs(a[key] - b[key])) <= 1e-9 * scale, (key, np.max(np.abs(a[key] - b[key])), scale)
  assert sorted(os.listdir(str(tmp_path / 'train_dp' / 'dp'))) == sorted(os.listdir(str(tmp_path / 'train_single' / 'dp')))
  # the summaries too: rank 0 only
  log = tmp_path / 'log_dp' / 'dp_train' / 'scalars.jsonl'
  assert log.exists() and len(log.read_text().splitlines()) == 2


def test_sharded_loader_pads_to_the_global_batch_and_takes_its_rows():
  """padding is never masked, so a rank's shard is padded to the GLOBAL batch's longest member."""
  from speecht_amd import speech_input
  rng = np.random.default_rng(0)
  samples = [(rng.standard_normal((t, 5)).astype(np.float32), [1, 2, t % 28]) for t in (9, 14, 11, 20, 8, 13, 7, 10)]
  whole = speech_input.InputBatchLoader(5, 4, lambda: iter(samples))
  ref = [whole._feed_item(b) for b in whole._batch(samples)]
  for rank in (0, 1):
    part = speech_input.InputBatchLoader(5, 2, lambda: iter(samples), shard=(rank, 2))
    got = [part._feed_item(b) for b in part._batch(samples)]
    assert len(got) == len(ref) == 2
    for (x, n, lab), (xr, nr, labr) in zip(got, ref):
      lo, hi = rank * 2, rank * 2 + 2
      assert x.shape == (2, xr.shape[1], 5) and np.array_equal(x, xr[lo:hi]) and np.array_equal(n, nr[lo:hi])
      assert lab.dense_shape.tolist() == [2, xr.shape[1]]
      rows = speech_input.sparse_to_label_lists(lab)
      assert [r.tolist() for r in rows] == [r.tolist() for r in speech_input.sparse_to_label_lists(labr)[lo:hi]]


def test_bucket_by_length_yields_every_sample_once_in_batches_of_neighbours():
  from speecht_amd import speech_input
  rng = np.random.default_rng(1)
  samples = [(np.zeros((int(t), 2), np.float32), [i]) for i, t in enumerate(rng.integers(20, 150, 70))]
  out = list(speech_input.bucket_by_length(iter(samples), 4, window=4, seed=3))
  assert sorted(s[1][0] for s in out) == list(range(70))
  pad = lambda seq: 1 - sum(s[0].shape[0] for s in seq) / (4.0 * sum(max(s[0].shape[0] for s in seq[i:i + 4]) for i in range(0, len(seq) - 3, 4)))
  assert pad(out[:68]) < 0.5 * pad(samples[:68])
  assert list(speech_input.bucket_by_length(iter(samples), 4, window=4, seed=3))[:8] == out[:8]      # seeded: reproducible
