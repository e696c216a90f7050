#!/usr/bin/env python3
"""Why does the bench line's side measurement (an engine of another arithmetic, started on the weights the fp32 run trained,
in the same process) run slower than the same mode in a process of its own?  Times the bf16 / bf16x6 step (a) on fresh
weights, (b) on the fp32 engine's trained weights, (c) the same after the fp32 engine is gone."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from speecht_amd.engine import Wav2LetterEngine
from tests import workloads as WL

dev = torch.device('cuda:0')
layers = WL.w2l_layers(80)
frames = 1001
x, seq_lens, labels = WL.make_batch([frames] * 32, 80, seed=100)

def timed(eng, steps=60, warm=8):
  feed = B.HostFeed(eng, x, seq_lens, labels)
  for _ in range(warm):
    B.train_step(eng, feed, None, 1e-4, 32)
  torch.cuda.synchronize()
  marks = [torch.cuda.Event() for _ in range(steps)]
  t0 = time.perf_counter()
  for k in range(steps):
    if k >= 3:
      marks[k - 3].synchronize()
    B.train_step(eng, feed, None, 1e-4, 32)
    marks[k].record()
  torch.cuda.synchronize()
  return round((time.perf_counter() - t0) / steps * 1e3, 3)


out = {}
w0 = WL.xavier_params(layers, seed=42, bias_range=0.05, dtype=np.float32)
for mode in ('bf16', 'bf16x6'):
  e = Wav2LetterEngine(layers, device=dev, conv_mode=mode); e.set_weights(w0)
  out[mode + '_fresh_alone'] = timed(e); del e; torch.cuda.empty_cache()
eng = Wav2LetterEngine(layers, device=dev); eng.set_weights(w0)
out['fp32'] = timed(eng)
for mode in ('bf16', 'bf16x6'):
  e = Wav2LetterEngine(layers, device=dev, conv_mode=mode); e.set_weights(w0)
  out[mode + '_fresh_beside_fp32_engine'] = timed(e); del e; torch.cuda.empty_cache()
  e = Wav2LetterEngine(layers, device=dev, conv_mode=mode); e.params.copy_(eng.params); e.mark_weights_changed()
  out[mode + '_trained_weights_beside_fp32_engine'] = timed(e); del e; torch.cuda.empty_cache()
trained = eng.params.clone()
del eng; torch.cuda.empty_cache()
for mode in ('bf16', 'bf16x6'):
  e = Wav2LetterEngine(layers, device=dev, conv_mode=mode); e.params.copy_(trained); e.mark_weights_changed()
  out[mode + '_trained_weights_alone'] = timed(e); del e; torch.cuda.empty_cache()
# (d) after the library's RCCL communicator has existed in the process (bench.comm_probe_world1)
eng = Wav2LetterEngine(layers, device=dev); eng.set_weights(w0)
feed = B.HostFeed(eng, x, seq_lens, labels)
out['comm_probe'] = B.comm_probe_world1(eng, feed, 1e-4, 32, 10, 3)
for mode in ('bf16', 'bf16x6'):
  e = Wav2LetterEngine(layers, device=dev, conv_mode=mode); e.params.copy_(eng.params); e.mark_weights_changed()
  out[mode + '_after_comm_probe'] = timed(e); del e; torch.cuda.empty_cache()
# (e) after the in-step roofline passes (timed launch trace) and the mel measurement
r = B.measure_dominant_kernel(eng, 32, lambda: B.train_step(eng, feed, None, 1e-4, 32), 7.0)
out['mel'] = B.measure_mel(dev, 32, 10.0, 80).get('utt_per_s') if hasattr(B, 'measure_mel') else None
for mode in ('bf16', 'bf16x6'):
  e = Wav2LetterEngine(layers, device=dev, conv_mode=mode); e.params.copy_(eng.params); e.mark_weights_changed()
  out[mode + '_after_roofline_passes'] = timed(e); del e; torch.cuda.empty_cache()
print(json.dumps(out))
