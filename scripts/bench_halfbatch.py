#!/usr/bin/env python3
"""Experiment: forward pass of one engine at batch 32 against two engines at batch 16 on two streams (the batch
dimension is independent through the whole network) -- does interleaving two half-batch chains fill the gaps the
narrow layers' under-filled launches leave?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speecht_amd.engine import Wav2LetterEngine
from tests import workloads as WL

dev = torch.device('cuda:0')
layers = WL.w2l_layers(80)
params = WL.xavier_params(layers, seed=42, dtype=np.float32)
x, seq, labels = WL.make_batch([1001] * 32, 80, seed=3)
x = x.astype(np.float32)


def timed(fn, reps=20):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps

full = Wav2LetterEngine(layers, device=dev)
full.set_weights(params)
full.load_batch(x, seq)
print('one engine, batch 32: forward %.3f ms' % timed(full.forward))
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
halves = []
for k, s in enumerate((s1, s2)):
  e = Wav2LetterEngine(layers, device=dev, stream=s)
  e.set_weights(params)
  with torch.cuda.stream(s):
    e.load_batch(x[16 * k:16 * k + 16], seq[16 * k:16 * k + 16])
  halves.append(e)
torch.cuda.synchronize()


def both():
  main = torch.cuda.current_stream(dev)
  ev = torch.cuda.Event(); ev.record(main)
  for e, s in zip(halves, (s1, s2)):
    s.wait_event(ev)
    e.forward()
  for s in (s1, s2):
    d = torch.cuda.Event(); d.record(s); main.wait_event(d)
print('two engines, batch 16 each, two streams: forward %.3f ms' % timed(both))
print('one half alone (batch 16): %.3f ms' % timed(lambda: (s1.wait_stream(torch.cuda.current_stream(dev)), halves[0].forward(), torch.cuda.current_stream(dev).wait_stream(s1))))
