#!/usr/bin/env python3
"""CTC loss + gradient alone (configs[1] shape: 32 utterances, T' = 501, 150 labels), HIP events, median of 5 x 20 calls."""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speecht_amd.engine import Wav2LetterEngine
from speecht_amd._lib import set_tuning

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--frames', type=int, default=501)
ap.add_argument('--labels', type=int, default=150)
ap.add_argument('--tune', action='append', default=[])
a = ap.parse_args()
for kv in a.tune:
  k, v = kv.split('=')
  set_tuning(k, int(v))
B, T, C = a.batch, a.frames, 29
rng = np.random.default_rng(0)
eng = Wav2LetterEngine([(1, 1, 16, C, False)], device='cuda:0')
eng.load_batch(np.zeros((B, T, 16)), [T] * B)
eng.X[-1].interior().copy_(torch.as_tensor(rng.normal(size=(B, T, C)).astype(np.float32)))
eng.ctc_lens = torch.full((B,), T, dtype=torch.int32, device='cuda:0')
eng.set_labels([list(rng.integers(0, C - 1, size=a.labels)) for _ in range(B)])
for _ in range(3):
  eng.ctc_loss_grad(1.0 / B)
torch.cuda.synchronize()
times = []
for _ in range(5):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(20):
    eng.ctc_loss_grad(1.0 / B)
  e1.record()
  torch.cuda.synchronize()
  times.append(e0.elapsed_time(e1) / 20 * 1e3)
print('ctc_loss_grad B=%d T=%d L=%d %s: %.1f us per call (median of 5; all %s), loss[0] %.4f' % (
    B, T, a.labels, a.tune, sorted(times)[2], [round(t, 1) for t in times], float(eng.loss[0])))
