#!/usr/bin/env python3
"""Per-layer micro-benchmark of the conv kernels at the BASELINE config-2 shapes (HIP events)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speecht_amd._lib import call  # noqa: E402
from speecht_amd.engine import Wav2LetterEngine  # noqa: E402
from tests import workloads as WL  # noqa: E402


def timeit(fn, reps):
  fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  e1.synchronize()
  return e0.elapsed_time(e1) / reps


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--batch', type=int, default=32)
  ap.add_argument('--frames', type=int, default=1001)
  ap.add_argument('--reps', type=int, default=5)
  ap.add_argument('--layers', type=str, default='0,1,8,9,10')
  ap.add_argument('--only', type=str, default='', help='fwd|data|filter: time just that kernel')
  ap.add_argument('--tune', action='append', default=[], help='name=value tuning override (st_set_tuning)')
  args = ap.parse_args()
  from speecht_amd._lib import set_tuning
  for kv in args.tune:
    k, v = kv.split('=')
    set_tuning(k, int(v))
  layers = WL.w2l_layers(80)
  eng = Wav2LetterEngine(layers, device='cuda:0', fft_conv=False)
  eng.set_weights(WL.xavier_params(layers, seed=42, dtype=np.float32))
  x, sl, labels = WL.make_batch([args.frames] * args.batch, 80, seed=0)
  eng.load_batch(x, sl)
  eng.set_labels(labels)
  eng.forward()
  eng.ctc_loss_grad(1.0 / args.batch)
  eng.backward()
  torch.cuda.synchronize()
  s = eng.stream_ptr
  tot = {'fwd': 0.0, 'bwd_data': 0.0, 'bwd_filter': 0.0}
  print('%-4s %-22s %9s %9s %9s   (ms | TF/s algorithmic)' % ('L', 'shape MxKxN', 'fwd', 'bwd_data', 'bwd_filt'))
  for i in [int(v) for v in args.layers.split(',')]:
    l = eng.layers[i]
    t_in, t_out, pl, pr = eng.geo[i]
    flops = 2.0 * args.batch * t_out * l.width * l.cin * l.cout
    pf, pb = eng._slice(eng.params, i)
    gf, gb = eng._slice(eng.grads, i)
    f = 0.0 if args.only not in ('', 'fwd') else timeit(lambda: call('st_conv1d_nwc_fwd_ws_f32', eng.X[i].ref, eng._ptr(pf), eng._ptr(pb), l.width, l.stride, pl,
                            int(l.relu), eng.X[i + 1].ref, eng._ptr(eng.wgrad_ws), eng.wgrad_ws.numel() * 4, s), args.reps)
    w = 0.0 if args.only not in ('', 'filter') else timeit(lambda: call('st_conv1d_nwc_bwd_filter_f32', eng.X[i].ref, eng.dZ[i].ref, l.width, l.stride, pl,
                            eng._ptr(gf), eng._ptr(gb), eng._ptr(eng.wgrad_ws), eng.wgrad_ws.numel() * 4, s), args.reps)
    d = 0.0
    if i > 0 and args.only in ('', 'data'):
      d = timeit(lambda: call('st_conv1d_nwc_bwd_data_f32', eng.dZ[i].ref, eng._ptr(eng.packed_t[i]), l.width, pl,
                              eng.X[i].ref, eng.dZ[i - 1].ref, eng._ptr(eng.wgrad_ws), eng.wgrad_ws.numel() * 4, s), args.reps)
    mult = 7 if i == 1 else 1
    tot['fwd'] += f * mult; tot['bwd_data'] += d * mult; tot['bwd_filter'] += w * mult
    tf = lambda ms: flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    print('L%-3d %-22s %5.3f|%5.1f %5.3f|%5.1f %5.3f|%5.1f' % (
        i, '%dx%dx%d' % (args.batch * t_out, l.width * l.cin, l.cout), f, tf(f), d, tf(d), w, tf(w)))
  print('sum (L1 counted x7): fwd %.3f ms, bwd_data %.3f ms, bwd_filter %.3f ms, total %.3f ms' % (
      tot['fwd'], tot['bwd_data'], tot['bwd_filter'], sum(tot.values())))


if __name__ == '__main__':
  main()
