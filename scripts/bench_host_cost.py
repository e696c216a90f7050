#!/usr/bin/env python3
"""CPU time the host spends ENQUEUEING one training step (not the paced wait of the bench loop): the step's Python + ctypes
calls timed with the GPU kept at most one step behind, per phase, for a conv mode -- what a C-side driver of the
step could give back.  bench.py's workload (32 x 10 s, 80-mel), data resident."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                               # noqa: E402
from speecht_amd.engine import Wav2LetterEngine            # noqa: E402
from tests import workloads as WL                          # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--conv-mode', default='fp32')
  ap.add_argument('--steps', type=int, default=40)
  args = ap.parse_args()
  layers = WL.w2l_layers(80)
  eng = Wav2LetterEngine(layers, device='cuda:0', conv_mode=args.conv_mode)
  eng.init_xavier(seed=1)
  x, seq, labels = WL.make_batch([1001] * 32, 80, seed=100)
  feed = bench.HostFeed(eng, x, seq, labels)
  names = ['feed', 'forward', 'ctc', 'backward', 'update']
  acc = dict.fromkeys(names, 0.0)
  calls = 0
  from speecht_amd import _lib
  lib = _lib.load()

  def step(timed):
    t = [time.perf_counter()]
    feed.next(); t.append(time.perf_counter())
    eng.forward(); t.append(time.perf_counter())
    eng.ctc_loss_grad(1.0 / 32); t.append(time.perf_counter())
    eng.backward(); t.append(time.perf_counter())
    eng.apply_update(1e-4); t.append(time.perf_counter())
    if timed:
      for n, a, b in zip(names, t, t[1:]):
        acc[n] += b - a

  for _ in range(5):
    step(False)
  torch.cuda.synchronize()
  done = [torch.cuda.Event() for _ in range(2)]
  t0 = time.perf_counter()
  for k in range(args.steps):
    if k >= 2:
      done[k % 2].synchronize()                            # at most two steps in flight: the enqueue cost is CPU time, not queue back-pressure
    step(True)
    done[k % 2].record()
  torch.cuda.synchronize()
  wall = (time.perf_counter() - t0) / args.steps
  host = sum(acc.values()) / args.steps
  print(json.dumps(dict(workload='32 x 10 s, 80-mel training step, resident data', conv_mode=args.conv_mode, 
                        ms_per_step=round(wall * 1e3, 3), host_enqueue_ms_per_step=round(host * 1e3, 3),
                        host_enqueue_by_phase_ms={n: round(v / args.steps * 1e3, 3) for n, v in acc.items()})))


if __name__ == '__main__':
  main()
