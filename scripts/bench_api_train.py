#!/usr/bin/env python3
"""End-to-end throughput of the reference-shaped Python API (InputBatchLoader feeder threads ->
SpeechModel.step): config-2 shapes (batch 32 of 10 s, 80-mel), synthetic cached samples.  Includes
everything bench.py leaves out on purpose: host-side padding, the queue, H2D copies, label upload."""
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
  ap.add_argument('--graph', action='store_true', help='model.step_graph = True: the training step as one HIP-graph launch')
  args = ap.parse_args()
  if args.conv_mode:
    os.environ['ST_CONV_MODE'] = args.conv_mode
  rng = np.random.default_rng(0)
  pool = [(rng.standard_normal((1001, 80)).astype(np.float32), rng.integers(0, 28, 150).tolist()) for _ in range(64)]

  def generator():
    while True:
      for s in pool:
        yield s

  loader = speech_input.InputBatchLoader(80, 32, generator)
  model = speech_model.create_default_model(Flags(), 80, loader)
  model.step_graph = bool(args.graph)
  with speech_model.Session('cuda:0') as sess:
    model.init_session(sess)
    coord = speech_input.Coordinator()
    loader.start_threads(sess=sess, coord=coord, n_threads=2)
    for _ in range(5):
      model.step(sess)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
      model.step(sess)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    coord.request_stop()
  print(json.dumps({'workload': 'SpeechModel.step through InputBatchLoader, batch 32 x 10 s, 80-mel',
                    'conv_mode': model.engine.conv_mode, 'step_graph': bool(args.graph), 'ms_per_step': round(dt * 1e3, 3),
                    'utterances_per_s': round(32 / dt, 1)}))
  sys.stdout.flush()
  os._exit(0)


if __name__ == '__main__':
  main()
