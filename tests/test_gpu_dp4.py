"""Data-parallel first-run hardening on the one GPU a test box has (the 8-GPU exchange itself is the
driver's to run): FOUR ranks share cuda:0 with gloo as the transport, so everything but the wire is the production
path -- the five-bucket schedule in back-prop completion order, the per-layer hooks, and above all the `deferred` hook
of the frequency-domain layers, whose filter gradient runs on a side stream beside back-prop to the input and may only be
handed to the all-reduce once that stream is through (engine.backward).  Channel counts of 128 / 256 put every layer of
the reduced model on the frequency-domain path (the row thresholds are lowered through the engine's env knobs).

Also here: a label id the host refuses on ONE rank must not leave the other ranks hanging in the all-reduce (it
becomes a status word that travels with the gradients; every rank skips the update and raises), and `bench.py
--allreduce rccl` through its torch.distributed.run self-launch path at world size 1 (the library's own RCCL
communicator: rank count from ncclCommCount in the line).
"""
import json
import os
import subprocess
import sys

import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DP4_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["ST_ROOT"])
from tests import workloads as WL
from speecht_amd._lib import launch_trace
from speecht_amd.engine import Wav2LetterEngine
from speecht_amd.data_parallel import GradientAllReducer, shard_range
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)          # every rank drives cuda:0
layers = WL.w2l_layers(16, width=128, fc=256)
params = WL.xavier_params(layers, seed=3, bias_range=0.05)
frames = [200, 200, 161, 200, 133, 200, 97, 200]
x, seq, labels = WL.make_batch(frames, 16, seed=4)                     # global batch 8 -> 2 utterances per rank
lo, hi = shard_range(len(labels), rank, world)

def step(eng, xs, ss, ls, reducer, trace=None):
  eng.load_batch(xs, ss)
  eng.set_labels(ls)
  eng.forward()
  eng.ctc_loss_grad(1.0 / len(labels))                                # 1 / GLOBAL batch
  if trace is not None:
    with launch_trace() as tr:
      eng.backward(reducer.on_layer_done if reducer else None)
    trace.extend(tr.lines)
  else:
    eng.backward(reducer.on_layer_done if reducer else None)
  if reducer:
    reducer.finish()
  eng.apply_update(1e-3, 5.0)

eng = Wav2LetterEngine(layers, device="cuda:0")
eng.set_weights(params)
red = GradientAllReducer(eng.reduce_buffer, eng.reduce_ranges)
assert len(red.buckets) == 5 and [b[0] for b in red.buckets] == [9, 8, 4, 1, 0], red.buckets
lines = []
for k in range(3):
  step(eng, x[lo:hi], seq[lo:hi], labels[lo:hi], red, trace=lines if k == 0 else None)
torch.cuda.synchronize()
# the frequency-domain path with its side-stream filter gradients really ran (36-bin products of the 7-tap layers)
assert sorted(eng.fft) == list(range(9)) and all("ws2" in eng.fft[i] for i in range(1, 8)), sorted(eng.fft)
assert sum(1 for l in lines if l.startswith("gemm_tn<") and " batched bins=72 " in l) == 7, "\n".join(lines)
mine = eng.params.clone()
gathered = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(gathered, mine)
assert all(torch.equal(gathered[0], g) for g in gathered), "replicas diverged"
# ... and they land where one process stepping on the whole batch lands (same mean gradient; fp32 summation order differs)
solo = Wav2LetterEngine(layers, device="cuda:0")
solo.set_weights(params)
for _ in range(3):
  step(solo, x, seq, labels, None)
torch.cuda.synchronize()
start = Wav2LetterEngine(layers, device="cuda:0"); start.set_weights(params)
err, moved = float((mine - solo.params).abs().max()), float((solo.params - start.params).abs().max())
assert moved > 1e-3 and err < 2e-3 * moved, (err, moved)
del solo, start

# a label id the host refuses, on rank 1 only: nobody hangs, nobody updates, everybody raises
eng.defer_label_errors = True
bad = [list(l) for l in labels[lo:hi]]
if rank == 1:
  bad[0] = [3, 28, 4]                                                  # 28 is the blank: not a label
before = (eng.params.clone(), eng.adam_m.clone(), eng.step_count)
step(eng, x[lo:hi], seq[lo:hi], bad, red)
try:
  eng.fetch_losses()
  raised = None
except ValueError as e:
  raised = str(e)
torch.cuda.synchronize()
assert raised is not None, "rank %d did not raise" % rank
assert ("label ids must lie" in raised) == (rank == 1), raised
assert torch.equal(eng.params, before[0]) and torch.equal(eng.adam_m, before[1]) and eng.step_count == before[2]
# the next good batch trains again, replicas still identical
step(eng, x[lo:hi], seq[lo:hi], labels[lo:hi], red)
eng.fetch_losses()
mine = eng.params.clone()
dist.all_gather(gathered, mine)
assert all(torch.equal(gathered[0], g) for g in gathered) and not torch.equal(mine, before[0])
dist.destroy_process_group()
print("rank", rank, "ok", err, moved)
'''


def test_data_parallel_four_ranks_four_buckets_side_stream_gradients(tmp_path):
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  script = tmp_path / 'dp4_worker.py'
  script.write_text(DP4_WORKER)
  env = dict(os.environ, ST_ROOT=ROOT, MASTER_ADDR='127.0.0.1', MASTER_PORT='29641', WORLD_SIZE='4',
             ST_FFT_MIN_ROWS='1', ST_FFT_MIN_ROWS_NARROW='1')
  procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                            stderr=subprocess.STDOUT) for r in range(4)]
  outs = []
  for p in procs:
    try:
      outs.append(p.communicate(timeout=600)[0].decode())
    except subprocess.TimeoutExpired:
      for q in procs:
        q.kill()
      raise
  for r, (p, o) in enumerate(zip(procs, outs)):
    assert p.returncode == 0 and 'ok' in o, 'rank {} failed:\n{}'.format(r, o[-4000:])


DP_DELAY_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["ST_ROOT"])
from tests import workloads as WL
from speecht_amd.engine import Wav2LetterEngine
from speecht_amd.data_parallel import GradientAllReducer, shard_range
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
mode = os.environ["ST_TEST_MODE"]
dist.init_process_group("gloo", rank=rank, world_size=world)          # both ranks drive cuda:0
layers = WL.w2l_layers(16, width=128, fc=256)
params = WL.xavier_params(layers, seed=3, bias_range=0.05)
x, seq, labels = WL.make_batch([200, 200, 161, 200, 133, 200, 97, 200], 16, seed=4)
lo, hi = shard_range(len(labels), rank, world)

def one_step(eng, red, slow):
  # `slow`: every hand-over to a side stream is followed by a long spin on THAT stream before the work, so the filter gradients
  # that run there finish long after the compute stream has enqueued the bucket's exchange -- unless the engine orders the two
  if slow:
    plain = eng._on_side_stream
    def delayed(fn, second=False):
      def spun():
        with torch.cuda.stream(eng._stream):
          torch.cuda._sleep(int(2e6))                    # 20 ms of the 100 MHz wall clock (1 ms if it counted shader cycles)
        fn()
      plain(spun, second=second)
    eng._on_side_stream = delayed
  try:
    eng.load_batch(x[lo:hi], seq[lo:hi])
    eng.set_labels(labels[lo:hi])
    eng.forward()
    eng.ctc_loss_grad(1.0 / len(labels))
    eng.backward(red.on_layer_done, red.hook_layers)       # hooks only where a bucket ends, as SpeechModel.step and bench.py call it
    red.finish()
    torch.cuda.synchronize()
    g = eng.reduce_buffer.clone()
    eng.apply_update(1e-3, 5.0)
    torch.cuda.synchronize()
  finally:
    if slow:
      del eng._on_side_stream
  return g

def engine():
  e = Wav2LetterEngine(layers, device="cuda:0", conv_mode=mode)
  e.set_weights(params)
  return e

a, b = engine(), engine()
ra = GradientAllReducer(a.reduce_buffer, a.reduce_ranges)
rb = GradientAllReducer(b.reduce_buffer, b.reduce_ranges)
assert ra.hook_layers == {9, 8, 4, 1, 0}, ra.hook_layers
for k in range(2):
  ga, gb = one_step(a, ra, slow=False), one_step(b, rb, slow=True)
  assert torch.equal(ga, gb), ("reduced gradients depend on side-stream timing", k, float((ga - gb).abs().max()))
  assert torch.equal(a.params, b.params)
if mode == "fp32":
  assert all("ws2" in a.fft[i] for i in range(1, 8))     # the narrow layers' filter gradients did run on the side streams
else:
  assert set(range(1, 8)) <= set(a._side_wgrad_bf16), a._side_wgrad_bf16
mine = b.params.clone()
gathered = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(gathered, mine)
assert all(torch.equal(gathered[0], g) for g in gathered), "replicas diverged"
dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_bucket_exchange_waits_for_side_stream_filter_gradients(tmp_path, mode):
  """with hooks only at bucket-boundary layers, the bottom bucket (L0..L3) is handed over at L0,
  whose own gradient runs on the compute stream while those of L1-L3 are still on the side streams.  Two ranks on cuda:0
  (gloo): the same steps once as they come and once with every side-stream hand-over delayed by a long spin kernel must give
  bit-identical reduced gradients and weights."""
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  script = tmp_path / 'dp_delay_worker.py'
  script.write_text(DP_DELAY_WORKER)
  env = dict(os.environ, ST_ROOT=ROOT, MASTER_ADDR='127.0.0.1', MASTER_PORT='29647' if mode == 'fp32' else '29649', WORLD_SIZE='2',
             ST_FFT_MIN_ROWS='1', ST_FFT_MIN_ROWS_NARROW='1', ST_TEST_MODE=mode)
  procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                            stderr=subprocess.STDOUT) for r in range(2)]
  outs = []
  for p in procs:
    try:
      outs.append(p.communicate(timeout=600)[0].decode())
    except subprocess.TimeoutExpired:
      for q in procs:
        q.kill()
      raise
  for r, (p, o) in enumerate(zip(procs, outs)):
    assert p.returncode == 0 and 'ok' in o, 'rank {} failed:\n{}'.format(r, o[-4000:])


def test_bench_rccl_transport_through_the_self_launch_path_at_world_one():
  """`python bench.py --gpus 1 --self-launch --force-allreduce --allreduce rccl`: torch.distributed.run starts the one
  rank, the library's communicator is created from the broadcast id, every bucket goes through st_allreduce_buckets_f32,
  and the line says who exchanged: torch.distributed's world size AND the communicator's own rank count."""
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  env = dict(os.environ)
  for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
    env.pop(k, None)
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--self-launch', '--force-allreduce',
                      '--allreduce', 'rccl', '--steps', '3', '--warmup', '1', '--batch', '4', '--seconds', '2', '--no-alt',
                      '--no-cpu-baseline'], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, r.stdout
  out = json.loads(lines[0])
  assert out['n_gpus'] == 1 and out['config']['allreduce'] == 'rccl'
  assert out['rccl_ranks'] == dict(torch_distributed=1, backend='nccl', library_comm=1, transport='rccl', library_comm_spans_job=True,
                                   fallback=None), out['rccl_ranks']
  assert not any(l.strip() and not l.startswith('{') for l in r.stdout.splitlines()), r.stdout      # nothing but the line on stdout
  assert 'gradient all-reduce' in out['step_includes']
  assert out['parity']['passed'] is True and out['roofline']['frac'] > 0
