#!/usr/bin/env python3
"""Where the pipelined beam-search loop of inference.transcribe spends its wall time per batch (host side)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from speecht_amd.engine import Wav2LetterEngine, decoder_stream_pair
from speecht_amd import inference as I
from tests import workloads as WL
dev = torch.device('cuda:0')
layers = WL.w2l_layers(80)
eng = Wav2LetterEngine(layers, device=dev)
eng.set_weights(WL.xavier_params(layers, seed=42, bias_range=0.05, dtype=np.float32))
B, frames, beam, NB = 16, 3001, 16, 12
feats = [WL.synthetic_features(500 + i, frames, 80).astype(np.float32) for i in range(B)] * NB
lengths, buckets = I._plan(feats, B, False)
cs, ds = decoder_stream_pair(dev)
for masked in (True, False):
  if not masked:
    cs, ds = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
  for rep in range(2):
    torch.cuda.synchronize()
    st = I._Stager(dev, feats, lengths, buckets); st.start()
    t_get = t_enq = t_col = 0.0
    pending = None
    t00 = time.perf_counter()
    for idx in buckets:
      t0 = time.perf_counter(); staged = st.get(); t1 = time.perf_counter()
      with torch.cuda.stream(cs):
        eng.load_batch(staged, [lengths[i] for i in idx]); eng.forward(); h = eng.beam_search_decode_async(beam, ds)
      t2 = time.perf_counter()
      if pending is not None:
        pending.result()
      t3 = time.perf_counter()
      pending = h
      t_get += t1 - t0; t_enq += t2 - t1; t_col += t3 - t2
    pending.result(); st.close()
    tot = time.perf_counter() - t00
    print('masked=%s rep %d: %.2f ms per batch; stager wait %.2f, enqueue %.2f, collect wait %.2f' % (
        masked, rep, tot / NB * 1e3, t_get / NB * 1e3, t_enq / NB * 1e3, t_col / NB * 1e3))
# GPU-only: the same loop with a resident batch (no stager)
x, seq_lens, _ = WL.make_batch([frames] * B, 80, seed=7)
cs, ds = decoder_stream_pair(dev)
eng.load_batch(x, seq_lens)
torch.cuda.synchronize()
for rep in range(2):
  t0 = time.perf_counter(); pending = None
  for _ in range(NB):
    with torch.cuda.stream(cs):
      eng.forward(); h = eng.beam_search_decode_async(beam, ds)
    if pending is not None:
      pending.result()
    pending = h
  pending.result()
  print('resident batch, masked streams: %.2f ms per batch' % ((time.perf_counter() - t0) / NB * 1e3))
