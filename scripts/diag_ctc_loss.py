#!/usr/bin/env python3
"""Where the per-utterance CTC loss differs from the float64 oracle: the CTC kernels alone (oracle CTC evaluated on the DEVICE's
own logits) against the kernel's (hi, lo) loss pair and its fp32 hi part -- logits like the bench's last layer emits
(configs[1]: 32 x 501 frames, 150 labels) and N(0, 1) logits."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speecht_amd.engine import Wav2LetterEngine
from oracle import w2l_oracle as O

B, T, C, L = 8, 501, 29, 150
for name, scale in (('N(0,1) logits', 1.0), ('N(0,0.05) logits (fresh Xavier net)', 0.05), ('N(0,4) logits (trained net)', 4.0)):
  rng = np.random.default_rng(3)
  eng = Wav2LetterEngine([(1, 1, 16, C, False)], device='cuda:0')
  eng.load_batch(np.zeros((B, T, 16)), [2 * T] * B)
  logits = (scale * rng.normal(size=(B, T, C))).astype(np.float32)
  eng.X[-1].interior().copy_(torch.as_tensor(logits))
  eng.ctc_lens = torch.full((B,), T, dtype=torch.int32, device='cuda:0')
  labels = [list(rng.integers(0, C - 1, size=L)) for _ in range(B)]
  eng.set_labels(labels)
  eng.ctc_loss_grad(1.0 / B)
  torch.cuda.synchronize()
  ref, _ = O.ctc_loss_and_grad(logits.transpose(1, 0, 2).astype(np.float64), labels, [T] * B)
  hi = eng.loss.cpu().numpy().astype(np.float64)
  pr = eng.losses_precise()
  print('%-38s loss ~%.1f: |hi+lo - oracle| max %.2e (signed mean %+.2e), |hi - oracle| max %.2e, fp32 ulp %.1e' % (
      name, ref.mean(), np.abs(pr - ref).max(), (pr - ref).mean(), np.abs(hi - ref).max(), np.spacing(np.float32(ref.max()))))
