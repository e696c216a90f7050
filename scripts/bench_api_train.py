#!/usr/bin/env python3
"""End-to-end throughput of the reference-shaped Python API (InputBatchLoader feeder threads ->
SpeechModel.step): config-2 shapes (batch 32 of 10 s, 80-mel), synthetic cached samples.  Includes
everything bench.py leaves out on purpose: host-side padding, the queue, H2D copies, label upload.

--world N: the data-parallel form of the same loop (what `torchrun --nproc-per-node N speecht-cli train` runs): N ranks, each its
rows of every global batch, `SpeechModel.enable_data_parallel`.  On a 1-GPU box the ranks share cuda:0 over gloo (ST_SHARE_GPU=1,
ST_DIST_BACKEND=gloo: the transport is then the host's, so only the CONTROL FLOW is comparable -- against `bench.py --gpus N`
under the same two knobs, whose loop reads no loss)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speecht_amd import speech_input, speech_model    # noqa: E402


class Flags:
  command, learning_rate, learning_rate_decay_factor, max_gradient_norm, momentum = 'train', 1e-4, 0.0, 5.0, 0.9
  log_dir, run_name, run_type = '/tmp/speecht_api_bench', 'bench', 'train'


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--steps', type=int, default=100)
  ap.add_argument('--conv-mode', default=None)
  ap.add_argument('--world', type=int, default=1)
  args = ap.parse_args()
  if args.world > 1 and 'WORLD_SIZE' not in os.environ:
    import socket
    import subprocess
    with socket.socket() as sk:
      sk.bind(('127.0.0.1', 0))
      port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.setdefault('ST_SHARE_GPU', '1')
    env.setdefault('ST_DIST_BACKEND', 'gloo')
    sys.exit(subprocess.call([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.world),
                              '--master-addr', '127.0.0.1', '--master-port', str(port)] + sys.argv, env=env))
  from speecht_amd import data_parallel
  flags = Flags()
  flags.device = 'cuda:0'
  rank, world = data_parallel.init_job(flags)
  if args.conv_mode:
    os.environ['ST_CONV_MODE'] = args.conv_mode
  rng = np.random.default_rng(0)
  pool = [(rng.standard_normal((1001, 80)).astype(np.float32), rng.integers(0, 28, 150).tolist()) for _ in range(64)]

  def generator():
    while True:
      for s in pool:
        yield s

  loader = speech_input.InputBatchLoader(80, 32, generator, shard=(rank, world))
  model = speech_model.create_default_model(flags, 80, loader)
  with speech_model.Session(flags.device) as sess:
    model.init_session(sess)
    if world > 1:
      model.enable_data_parallel()
    coord = speech_input.Coordinator()
    loader.start_threads(sess=sess, coord=coord, n_threads=2 if world == 1 else 1)
    for _ in range(5):
      model.step(sess)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
      model.step(sess)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    coord.request_stop()
  if world > 1:
    import torch.distributed as dist
    t = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t[0])
  if rank == 0:
    print(json.dumps({'workload': 'SpeechModel.step through InputBatchLoader, batch 32 x 10 s per rank, 80-mel',
                      'conv_mode': model.engine.conv_mode, 'world': world, 'ms_per_step': round(dt * 1e3, 3),
                      'utterances_per_s': round(32 * world / dt, 1),
                      'transport': (model._reducer.transport if model._reducer else None),
                      'shared_gpu': bool(os.environ.get('ST_SHARE_GPU')) if world > 1 else None}))
  sys.stdout.flush()
  os._exit(0)


if __name__ == '__main__':
  main()
