#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: 30 s long-form utterances (T = 3001 -> T' = 1501), batch 16 per GPU,
forward + CTC decode.  Times forward, greedy decode and the LM-free prefix beam search (beam 16 by
default) separately with HIP events on resident inputs (parity of the decoders is pinned in
tests/test_gpu_parity.py, not here)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speecht_amd.engine import Wav2LetterEngine  # noqa: E402
from tests import workloads as WL                # noqa: E402


def timed(fn, reps):
  fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(reps):
    fn()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) / reps


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--batch', type=int, default=16)
  ap.add_argument('--seconds', type=float, default=30.0)
  ap.add_argument('--beam', type=int, default=16)
  ap.add_argument('--reps', type=int, default=5)
  args = ap.parse_args()
  dev = torch.device('cuda:0')
  frames = 1 + int(args.seconds * 16000) // 160
  layers = WL.w2l_layers(80)
  eng = Wav2LetterEngine(layers, device=dev)
  eng.set_weights(WL.xavier_params(layers, seed=42, bias_range=0.05, dtype=np.float32))
  x, seq_lens, _ = WL.make_batch([frames] * args.batch, 80, seed=7)
  eng.load_batch(x, seq_lens)
  t_fwd = timed(eng.forward, args.reps)
  # random-init logits are nearly flat (every candidate a near tie); decode seeded N(0, 3^2) logits instead
  g = torch.Generator(device='cpu').manual_seed(11)
  eng.X[-1].interior().copy_(torch.randn(eng.X[-1].interior().shape, generator=g) * 3.0)
  t_greedy = timed(lambda: eng.greedy_decode(), args.reps)
  t_beam = timed(lambda: eng.beam_search_decode(args.beam), args.reps)
  ids, logp = eng.beam_search_decode(args.beam)
  out = {'workload': 'configs[4]: batch {} of {:g} s, T\'={}, beam {}'.format(args.batch, args.seconds, eng.t_out, args.beam),
         'forward_ms': round(t_fwd, 3), 'greedy_ms_incl_d2h': round(t_greedy, 3),
         'beam_ms_incl_d2h': round(t_beam, 3), 'utt_per_s_forward_plus_beam': round(args.batch / (t_fwd + t_beam) * 1e3, 1),
         'mean_decoded_len': float(np.mean([len(i) for i in ids]))}
  print(json.dumps(out))


if __name__ == '__main__':
  main()
