#!/usr/bin/env python3
"""What would a half-batch CTC schedule under a CU mask buy?  The pieces, measured directly:
  * the 2000 x 2000 layer's forward product (W-tap GEMM form, M = 16 032 rows) whole on 256 CUs, and as two halves of 8 016 rows
    on the 240-CU masked stream -- the cost of splitting (tile-grid quantisation: 1 008 tiles over 480 slots) and of 16 CUs less;
  * the CTC recursion of 16 utterances alone on the 16-CU masked stream, and beside a half product on the other 240.
A half-batch schedule hides at most the recursion (its length does not shrink with the batch: 501 dependent steps) and pays the
split twice (forward tail, backward head)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from speecht_amd._lib import call                          # noqa: E402
from speecht_amd.engine import Wav2LetterEngine            # noqa: E402
from speecht_amd.engine_streams import decoder_stream_pair  # noqa: E402
from tests import workloads as WL                          # noqa: E402

dev = torch.device('cuda:0')
layers = WL.w2l_layers(80)
params = WL.xavier_params(layers, seed=42, dtype=np.float32)
wide, narrow = decoder_stream_pair(dev)                    # 240 CUs / 16 CUs


def prepared(batch, stream=None):
  e = Wav2LetterEngine(layers, device=dev, stream=stream, fft_conv=False)
  e.set_weights(params)
  x, seq, labels = WL.make_batch([1001] * batch, 80, seed=3)
  ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream(dev))
  with ctx:
    e.load_batch(x.astype(np.float32), seq)
    e.set_labels(labels)
    e.forward()
    e.ctc_loss_grad(1.0 / batch)
  torch.cuda.synchronize()
  return e


def l9(e):
  i = 9
  l = e.layers[i]
  pf, pb = e._slice(e.params, i)
  call('st_conv1d_nwc_fwd_ws_f32', e.X[i].ref, e._ptr(pf), e._ptr(pb), l.width, l.stride, e.geo[i][2], int(l.relu), e.X[i + 1].ref,
       e._ptr(e.wgrad_ws), 0, e.stream_ptr)


def ctc(e):
  e.ctc_loss_grad(1.0 / 16)


def timed(fn, streams, reps=10):
  for _ in range(2):
    fn()
  torch.cuda.synchronize()
  main = torch.cuda.current_stream(dev)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record(main)
  for s in streams:
    s.wait_event(e0)
  for _ in range(reps):
    fn()
  for s in streams:
    d = torch.cuda.Event()
    d.record(s)
    main.wait_event(d)
  e1.record(main)
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e3


full = prepared(32)
half_plain = prepared(16)
half_wide = prepared(16, wide)
half_narrow = prepared(16, narrow)
out = dict(
    l9_forward_us_32_utterances_256cu=round(timed(lambda: l9(full), []), 1),
    l9_forward_us_16_utterances_256cu=round(timed(lambda: l9(half_plain), []), 1),
    l9_forward_us_16_utterances_240cu=round(timed(lambda: l9(half_wide), [wide]), 1),
    ctc_us_32_utterances_256cu=round(timed(lambda: full.ctc_loss_grad(1.0 / 32), []), 1),
    ctc_us_16_utterances_256cu=round(timed(lambda: ctc(half_plain), []), 1),
    ctc_us_16_utterances_16cu=round(timed(lambda: ctc(half_narrow), [narrow]), 1),
    l9_240cu_beside_ctc_16cu_us=round(timed(lambda: (l9(half_wide), ctc(half_narrow)), [wide, narrow]), 1),
    l9_256cu_beside_ctc_unmasked_us=round(timed(lambda: (l9(half_wide), ctc(half_plain)), [wide]), 1))
a = out
# the schedule: [L9 fwd A] [L9 fwd B || CTC A] [L9 wgrad A || CTC B] ... against [L9 fwd whole] [CTC whole]
serial = a['l9_forward_us_32_utterances_256cu'] + a['ctc_us_32_utterances_256cu']
split = a['l9_forward_us_16_utterances_240cu'] + a['l9_240cu_beside_ctc_16cu_us']
out['forward_tail_serial_us'] = round(serial, 1)
out['forward_tail_half_batch_us'] = round(split, 1)
out['note'] = ('serial = whole product + whole CTC; half-batch = first half product on 240 CUs, then the second half product beside '
               'the first half\'s CTC; the second half\'s CTC would have to hide under the backward pass\'s first product the same way')
print(json.dumps(out))
