#!/usr/bin/env python3
"""Is the step-head bubble of the profiled timelines real?  (under rocprofv3 the bf16 step shows 177 us of
compute-stream idle between clip + Adam and the first layer's product -- ten small launches 15 us apart, launch-rate bound under
the tracer.)  UNPROFILED run of bench.py's step loop with two events per step on the compute stream: one right behind the
update's last compute-stream launch, one right in front of the first convolution launch of the NEXT step.  Their distance is the
head of the step: the input hand-over, the casts / transforms of the first layer's operands -- and whatever the stream idles."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                               # noqa: E402
from speecht_amd.engine import Wav2LetterEngine            # noqa: E402
from tests import workloads as WL                          # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--conv-mode', default='fp32')
  ap.add_argument('--steps', type=int, default=40)
  ap.add_argument('--frames', type=int, default=1001)
  args = ap.parse_args()
  layers = WL.w2l_layers(80)
  eng = Wav2LetterEngine(layers, device='cuda:0', conv_mode=args.conv_mode)
  eng.init_xavier(seed=1)
  x, seq, labels = WL.make_batch([args.frames] * 32, 80, seed=100)
  feed = bench.HostFeed(eng, x, seq, labels)
  import speecht_amd.modes.bf16 as mb
  import speecht_amd.modes.fp32 as mf
  first = {'fp32': 'st_conv1d_nwc_fwd_fft_chain_f32', 'bf16x6': 'st_conv1d_nwc_fwd_fft_chain_f32', 'bf16': 'st_conv1d_nwc_fwd_ws_bf16'}[args.conv_mode]
  state = dict(armed=False, before=None)

  def hooked(orig):
    def call(name, *a):
      if state['armed'] and name == first:
        state['armed'] = False
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        state['before'] = ev
      return orig(name, *a)
    return call
  mb.call = hooked(mb.call)
  mf.call = hooked(mf.call)
  heads, steps = [], []
  prev_after = None
  marks = []
  for k in range(args.steps + 6):
    if k >= 2:
      marks[k - 2].synchronize()
    state['armed'] = True
    feed.next()
    eng.forward()
    before = state['before']
    eng.ctc_loss_grad(1.0 / 32)
    eng.backward()
    eng.apply_update(1e-4)
    after = torch.cuda.Event(enable_timing=True)
    after.record()
    marks.append(after)
    if prev_after is not None and k >= 6:
      heads.append((prev_after, before))
      steps.append((prev_after, after))
    prev_after = after
  torch.cuda.synchronize()
  h = sorted(a.elapsed_time(b) * 1e3 for a, b in heads)
  s = sorted(a.elapsed_time(b) for a, b in steps)
  print(json.dumps(dict(conv_mode=args.conv_mode, frames=args.frames, steps=len(h), head_us_median=round(h[len(h) // 2], 1), head_us_min=round(h[0], 1),
                        head_us_max=round(h[-1], 1), step_ms_median=round(s[len(s) // 2], 3),
                        note='head = compute stream from behind the update (+ the bottom layer\'s operand refresh) to in front of the '
                             'first convolution launch of the next step: input hand-over copy, cast / forward transform inputs')))


if __name__ == '__main__':
  main()
