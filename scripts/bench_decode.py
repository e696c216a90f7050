#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: 30 s long-form utterances (T = 3001 -> T' = 1501), batch 16 per GPU,
forward + CTC decode.  Times forward, greedy decode and the LM-free prefix beam search (beam 16 by
default) separately with HIP events on resident inputs (parity of the decoders is pinned in
tests/test_gpu_parity.py, not here)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speecht_amd.engine import Wav2LetterEngine  # noqa: E402
from tests import workloads as WL                # noqa: E402


def timed(fn, reps):
  fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(reps):
    fn()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) / reps


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--batch', type=int, default=16)
  ap.add_argument('--seconds', type=float, default=30.0)
  ap.add_argument('--beam', type=int, default=16)
  ap.add_argument('--reps', type=int, default=5)
  ap.add_argument('--samples', type=int, default=7)
  ap.add_argument('--pipeline-batches', type=int, default=12, help='batches through inference.transcribe for the pipelined figure')
  ap.add_argument('--resident-batches', type=int, default=48, help='batches of the resident forward + search loop (a search is still running when the last forward pass ends: ~one batch time of drain, amortised over these)')
  args = ap.parse_args()
  dev = torch.device('cuda:0')
  frames = 1 + int(args.seconds * 16000) // 160
  layers = WL.w2l_layers(80)
  eng = Wav2LetterEngine(layers, device=dev)
  eng.set_weights(WL.xavier_params(layers, seed=42, bias_range=0.05, dtype=np.float32))
  x, seq_lens, _ = WL.make_batch([frames] * args.batch, 80, seed=7)
  eng.load_batch(x, seq_lens)
  t_fwd = timed(eng.forward, args.reps)
  # random-init logits are nearly flat (every candidate a near tie); decode seeded N(0, 3^2) logits instead
  g = torch.Generator(device='cpu').manual_seed(11)
  eng.X[-1].interior().copy_(torch.randn(eng.X[-1].interior().shape, generator=g) * 3.0)
  t_greedy = timed(lambda: eng.greedy_decode(), args.reps)
  # the search: median of `samples` runs, the device part (events around the library call) apart from the host share
  # (launch + D2H of ids / lengths / scores + list building in engine.beam_search_decode)
  import ctypes
  import time
  from speecht_amd import _lib
  lib = _lib.load()
  B = eng.dec_lens.numel()
  ws = eng._storage.view('beam_ws', lib.st_ctc_beam_ws(B, eng.t_out, args.beam) // 4 + 16, torch.int32)[0]

  def device_only():
    _lib.call('st_ctc_beam_search_decode', eng.X[-1].ref, eng._ptr(eng.ctc_lens), args.beam, eng._ptr(eng.dec_ids), eng.t_out,
              eng._ptr(eng.dec_lens), eng._ptr(eng.dec_score), eng._ptr(ws), ws.numel() * 4, eng.stream_ptr)
  eng._wait_uploads()
  dev_ms, wall_ms = [], []
  for _ in range(args.samples):
    dev_ms.append(timed(device_only, 3))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ids, logp = eng.beam_search_decode(args.beam)
    wall_ms.append((time.perf_counter() - t0) * 1e3)
  t_beam = float(np.median(wall_ms))
  out = {'workload': 'configs[4]: batch {} of {:g} s, T\'={}, beam {}'.format(args.batch, args.seconds, eng.t_out, args.beam),
         'forward_ms': round(t_fwd, 3), 'greedy_ms_incl_d2h': round(t_greedy, 3),
         'beam_ms_incl_d2h': round(t_beam, 3), 'beam_ms_device_only': round(float(np.median(dev_ms)), 3),
         'beam_ms_host_share': round(t_beam - float(np.median(dev_ms)), 3),
         'beam_ms_incl_d2h_all': [round(v, 3) for v in wall_ms], 'beam_ms_device_only_all': [round(v, 3) for v in dev_ms],
         'method': 'median of {} runs; device = HIP events around st_ctc_beam_search_decode (log-softmax rows + search kernel), '
                   'wall = engine.beam_search_decode incl. D2H and list building'.format(args.samples),
         'utt_per_s_forward_plus_beam': round(args.batch / (t_fwd + t_beam) * 1e3, 1),
         'mean_decoded_len': float(np.mean([len(i) for i in ids]))}
  # the pipelined path a caller uses: inference.transcribe(beam_width=...) -- batch k's search on the decoder stream (CUs of its
  # own) under batch k + 1's forward pass; host padding, H2D and the read-back of the transcripts included
  from speecht_amd import engine as E
  from speecht_amd.inference import transcribe
  feats = [WL.synthetic_features(500 + i, frames, 80).astype(np.float32) for i in range(args.batch)] * args.pipeline_batches
  # logits of the sharpness the isolated figures above were taken on (std ~3; a random-init network's rows are nearly flat and
  # every candidate a near tie: the search then runs its many-survivors path, ~40 % slower): scale the output layer
  eng.load_batch(x, seq_lens)
  eng.forward()
  std = float(eng.X[-1].interior().std())
  w = eng.get_weights()
  w[-1] = (w[-1][0] * (3.0 / max(std, 1e-6)), w[-1][1] * (3.0 / max(std, 1e-6)))
  eng.set_weights(w)
  res = {'logit_std': 3.0, 'logit_std_random_init': round(std, 4)}
  for name, kw in (('pipelined', dict(pipeline=True)), ('serial_loop', dict(pipeline=False))):
    transcribe(eng, feats[:2 * args.batch], batch_size=args.batch, bucket=False, beam_width=args.beam, **kw)      # warm-up
    ts = []
    for _ in range(3):
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      got, _ = transcribe(eng, feats, batch_size=args.batch, bucket=False, beam_width=args.beam, **kw)
      ts.append(time.perf_counter() - t0)
    res[name] = dict(utt_per_s=round(len(feats) / float(np.median(ts)), 1), ms_per_batch=round(float(np.median(ts)) / args.pipeline_batches * 1e3, 3),
                     ids_digest=hash(tuple(tuple(s) for s in got)) & 0xffffffff)
  res['ids_equal'] = res['pipelined']['ids_digest'] == res['serial_loop']['ids_digest']
  # the same overlap on a RESIDENT batch (what `utt_per_s_forward_plus_beam` above is, serially): forward of batch k + 1 on the
  # compute stream while batch k is searched on the decoder stream
  eng.load_batch(x, seq_lens)
  torch.cuda.synchronize()
  for decoders in (1, 2):
    cs, ds = E.decoder_streams(dev, decoders)
    per = []
    for _ in range(3):
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      pending = []
      for _ in range(args.resident_batches):
        with torch.cuda.stream(cs):
          eng.forward()
          pending.append(eng.beam_search_decode_async(args.beam, ds))
        if len(pending) > decoders:
          pending.pop(0).result()
      for h in pending:
        h.result()
      per.append((time.perf_counter() - t0) / args.resident_batches * 1e3)
    tag = '' if decoders == 2 else '_one_decoder_stream'
    out['utt_per_s_forward_plus_beam_overlapped' + tag] = round(args.batch / float(np.median(per)) * 1e3, 1)
    out['ms_per_batch_forward_plus_beam_overlapped' + tag] = round(float(np.median(per)), 3)
  res['decoder_streams'] = 'CU-masked (hipExtStreamCreateWithCUMask: two decoder streams on 16 CUs, forward 240)' if isinstance(
      E.decoder_streams(dev, 2)[0], torch.cuda.ExternalStream) else 'plain streams'
  res['note'] = ('inference.transcribe(beam_width=%d) on %d batches of %d x %g s: host padding, H2D and read-back included; the '
                 'network\'s own logits with the output layer scaled to std 3' % (args.beam, args.pipeline_batches, args.batch, args.seconds))
  out['transcribe_beam'] = res
  print(json.dumps(out))


if __name__ == '__main__':
  main()
