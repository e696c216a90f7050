#!/usr/bin/env python3
"""Per-layer micro-benchmark of the bf16-activation conv kernels (BASELINE config 4) at the config-2 shapes."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speecht_amd._lib import call  # noqa: E402
from speecht_amd.engine import Wav2LetterEngine  # noqa: E402
from tests import workloads as WL  # noqa: E402
from bench_conv import timeit  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--batch', type=int, default=32)
  ap.add_argument('--frames', type=int, default=1001)
  ap.add_argument('--reps', type=int, default=10)
  ap.add_argument('--layers', type=str, default='0,1,8,9,10')
  ap.add_argument('--tune', action='append', default=[], help='name=value tuning override (st_set_tuning)')
  args = ap.parse_args()
  from speecht_amd._lib import set_tuning
  for kv in args.tune:
    k, v = kv.split('=')
    set_tuning(k, int(v))
  layers = WL.w2l_layers(80)
  eng = Wav2LetterEngine(layers, device='cuda:0', conv_mode='bf16')
  eng.set_weights(WL.xavier_params(layers, seed=42, dtype=np.float32))
  x, sl, labels = WL.make_batch([args.frames] * args.batch, 80, seed=0)
  eng.load_batch(x, sl)
  eng.set_labels(labels)
  eng.forward()
  eng.ctc_loss_grad(1.0 / args.batch)
  eng.backward()
  torch.cuda.synchronize()
  s, L = eng.stream_ptr, len(eng.layers)
  ws, wsb = eng._ptr(eng.wgrad_ws_b), eng.wgrad_ws_b.numel() * 4
  tot = [0.0, 0.0, 0.0]
  print('%-4s %-22s %11s %11s %11s   (ms | TF/s algorithmic)' % ('L', 'shape MxKxN', 'fwd', 'bwd_data', 'bwd_filt'))
  for i in [int(v) for v in args.layers.split(',')]:
    l = eng.layers[i]
    t_in, t_out, pl, pr = eng.geo[i]
    flops = 2.0 * args.batch * t_out * l.width * l.cin * l.cout
    last = i + 1 == L
    f = timeit(lambda: call('st_conv1d_nwc_fwd_ws_bf16', eng.X[i].ref, eng._ptr(eng.Xb[i]), eng._ptr(eng.Wb[i]),
                            eng._ptr(eng._slice(eng.params, i)[1]), l.width, l.stride, pl, int(l.relu), eng.X[i + 1].ref,
                            None if last else eng._ptr(eng.Xb[i + 1]), eng._ptr(eng.X[i + 1].buf) if last else None,
                            ws, wsb, s), args.reps)
    gf, gb = eng._slice(eng.grads, i)
    w = timeit(lambda: call('st_conv1d_nwc_bwd_filter_tr_bf16' if eng._wgrad_tr[i] else 'st_conv1d_nwc_bwd_filter_bf16', eng.X[i].ref, eng._ptr(eng.Xb[i]), eng.dZ[i].ref,
                            eng._ptr(eng.dZb[i]), l.width, l.stride, pl, eng._ptr(gf), eng._ptr(gb), ws, wsb, s), args.reps)
    d = 0.0
    if i > 0:
      relu_in = eng.layers[i - 1].relu
      d = timeit(lambda: call('st_conv1d_nwc_bwd_data_bf16', eng.dZ[i].ref, eng._ptr(eng.dZb[i]), eng._ptr(eng.WTb[i]),
                              l.width, pl, eng.X[i].ref if relu_in else None, eng._ptr(eng.Xb[i]) if relu_in else None,
                              eng.dZ[i - 1].ref, eng._ptr(eng.dZb[i - 1]), ws, wsb, s), args.reps)
    mult = 7 if i == 1 else 1
    for k, v in enumerate((f, d, w)):
      tot[k] += v * mult
    tf = lambda ms: flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    print('L%-3d %-22s %5.3f|%5.0f %5.3f|%5.0f %5.3f|%5.0f' % (
        i, '%dx%dx%d' % (args.batch * t_out, l.width * l.cin, l.cout), f, tf(f), d, tf(d), w, tf(w)))
  print('sum (L1 counted x7): fwd %.3f ms, bwd_data %.3f ms, bwd_filter %.3f ms, total %.3f ms' % (
      tot[0], tot[1], tot[2], sum(tot)))


if __name__ == '__main__':
  main()
