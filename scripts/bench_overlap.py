#!/usr/bin/env python3
"""Experiment: filter gradient and back-prop to the input of one layer are independent -- do they run faster side by side
on two streams than back to back?  (config-2 shapes, fp32 W-tap kernels)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speecht_amd._lib import call
from speecht_amd.engine import Wav2LetterEngine
from tests import workloads as WL
layers = WL.w2l_layers(80)
eng = Wav2LetterEngine(layers, device='cuda:0', fft_conv=False)
eng.set_weights(WL.xavier_params(layers, seed=42, dtype=np.float32))
x, sl, labels = WL.make_batch([1001] * 32, 80, seed=0)
eng.load_batch(x, sl); eng.set_labels(labels); eng.forward(); eng.ctc_loss_grad(1 / 32); eng.backward()
torch.cuda.synchronize()
side = torch.cuda.Stream()
ws2 = torch.empty_like(eng.wgrad_ws)
main = torch.cuda.current_stream()
P = eng._ptr
for i in (1, 9, 8):
  l = eng.layers[i]; pl = eng.geo[i][2]
  gf, gb = eng._slice(eng.grads, i)
  def wgrad(stream):
    call('st_conv1d_nwc_bwd_filter_f32', eng.X[i].ref, eng.dZ[i].ref, l.width, l.stride, pl, P(gf), None, P(ws2), ws2.numel() * 4,
         ctypes.c_void_p(stream.cuda_stream))
  def bwd(stream):
    call('st_conv1d_nwc_bwd_data_f32', eng.dZ[i].ref, P(eng.packed_t[i]), l.width, pl, eng.X[i].ref, eng.dZ[i - 1].ref,
         P(eng.wgrad_ws), eng.wgrad_ws.numel() * 4, ctypes.c_void_p(stream.cuda_stream))
  def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for _ in range(reps): fn()
    e1.record(main); e1.synchronize()
    return e0.elapsed_time(e1) / reps
  def serial():
    wgrad(main); bwd(main)
  def overlapped():
    ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
    wgrad(side); bwd(main)
    ev2 = torch.cuda.Event(); ev2.record(side); main.wait_event(ev2)
  print('L%d: back to back %.3f ms, two streams %.3f ms' % (i, timed(serial), timed(overlapped)))
