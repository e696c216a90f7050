"""Where does a train_step_graph step differ from the eager step?  (round 5 debugging aid)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from speecht_amd.engine import Wav2LetterEngine
from tests import workloads as WL
mode = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
layers = WL.w2l_layers(16, width=128, fc=256)
params = WL.xavier_params(layers, seed=5, bias_range=0.05)
x, seq, labels = WL.make_batch([200, 161, 200, 133], 16, seed=10)
def make():
  e = Wav2LetterEngine(layers, device='cuda:0', conv_mode=mode)
  e.fft_min_rows = e.fft_min_rows_narrow = 1
  e.set_weights(params)
  return e
a, b = make(), make()
b.enable_step_graph()
for step in range(4):
  for e in (a, b):
    e.load_batch(x, seq); e.set_labels(labels)
  a.forward(); a.ctc_loss_grad(0.25); a.backward()
  torch.cuda.synchronize()
  ga = a.grads.clone()
  a.apply_update(1e-3)
  b.train_step_graph(0.25, 1e-3)
  torch.cuda.synchronize()
  d = lambda u, v: float((u - v).abs().max())
  print('step', step, 'grads', d(ga, b.grads), 'params', d(a.params, b.params), 'm', d(a.adam_m, b.adam_m), 'v', d(a.adam_v, b.adam_v),
        'stats', a.stats.tolist(), b.stats.tolist(), 'logits', d(a.X[-1].buf, b.X[-1].buf), 'loss', d(a.loss, b.loss),
        'rate', [float(b._storage.bufs['adam_rate_par%d' % p][0]) for p in (0, 1) if 'adam_rate_par%d' % p in b._storage.bufs],
        'graphs', len(b._step_graphs))
  if d(ga, b.grads) > 0:
    for i, (s0, e0) in enumerate(a.layer_ranges):
      print('   layer', i, d(ga[s0:e0], b.grads[s0:e0]))
