#!/usr/bin/env python3
"""Experiment: bf16x6 (fp32-accurate on bf16 MFMA) forward conv vs the fp32-MFMA kernel."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speecht_amd import _lib
from speecht_amd._lib import call
from speecht_amd.engine import Wav2LetterEngine, DevTensor3
from tests import workloads as WL
from scripts.bench_conv import timeit

lib = _lib.load()

layers = WL.w2l_layers(80)
eng = Wav2LetterEngine(layers, device='cuda:0')
eng.set_weights(WL.xavier_params(layers, seed=42, dtype=np.float32))
x, sl, labels = WL.make_batch([1001] * 32, 80, seed=0)
eng.load_batch(x, sl)
eng.forward()
torch.cuda.synchronize()
P = lambda t: ctypes.c_void_p(t.data_ptr())
for i in (1, 8, 9):
  l = eng.layers[i]
  X, Y = eng.X[i], eng.X[i + 1]
  n = X.buf.numel()
  xpl = torch.zeros(3 * n, dtype=torch.bfloat16, device='cuda:0')
  _lib.check(lib.st_exp_split3_bf16(P(X.buf), n, P(xpl), None), 'split3')
  pf, pb = eng._slice(eng.params, i)
  wpl = torch.zeros(3 * l.k_pad * l.n_pad, dtype=torch.bfloat16, device='cuda:0')
  _lib.check(lib.st_exp_split3_transpose_bf16(P(pf), l.k_pad, l.n_pad, P(wpl), None), 'split3t')
  out_store = torch.zeros(Y.buf.numel(), dtype=torch.float32, device='cuda:0')
  Y2 = DevTensor3(out_store, Y.batch, Y.frames, Y.channels, Y.halo, Y.t_pitch - Y.halo - Y.frames)
  run = lambda: _lib.check(lib.st_exp_conv1d_fwd_bf16x6(X.ref, P(xpl), P(wpl), P(pb), l.width, l.stride, eng.geo[i][2],
                                                        int(l.relu), Y2.ref, None, None), 'x6')
  run(); torch.cuda.synchronize()
  ref, got = Y.interior(), Y2.interior()
  err = float((ref - got).abs().max()); scale = float(ref.abs().max())
  ms = timeit(run, 10)
  ms32 = timeit(lambda: call('st_conv1d_nwc_fwd_f32', X.ref, P(pf), P(pb), l.width, l.stride, eng.geo[i][2], int(l.relu), Y.ref, eng.stream_ptr), 10)
  flops = 2.0 * 32 * eng.geo[i][1] * l.width * l.cin * l.cout
  print('L%d  bf16x6 %.3f ms (%.1f TF/s eq)   fp32-mfma %.3f ms (%.1f TF/s)   max|diff| %.3e (scale %.2f)' % (
      i, ms, flops / ms / 1e9, ms32, flops / ms32 / 1e9, err, scale))
