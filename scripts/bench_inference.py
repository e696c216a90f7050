#!/usr/bin/env python3
"""BASELINE config 3 on one GPU: inference only (conv stack + CTC greedy decode), variable-length
utterances (2-15 s, 80-mel), batches of 64 with and without length bucketing.  Wall-clock per pool of
utterances including the host-side padding, the H2D copy of every batch and the D2H of the decodes
(pipelined across batches by inference.transcribe; the serial loop is timed beside it)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speecht_amd import inference                 # noqa: E402
from speecht_amd.engine import Wav2LetterEngine   # noqa: E402
from tests import workloads as WL                 # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--utterances', type=int, default=2048)
  ap.add_argument('--batch', type=int, default=64)
  ap.add_argument('--conv-mode', default=None)
  ap.add_argument('--samples', type=int, default=5, help='timed windows per variant (the median is reported)')
  ap.add_argument('--min-window', type=float, default=0.5, help='seconds: the pool is transcribed repeatedly inside one window until it is this long')
  args = ap.parse_args()
  rng = np.random.default_rng(3)
  samples = rng.integers(32000, 240001, args.utterances)            # 2..15 s at 16 kHz
  frames = 1 + samples // 160
  feats = [rng.standard_normal((int(t), 80)).astype(np.float32) for t in frames]
  layers = WL.w2l_layers(80)
  eng = Wav2LetterEngine(layers, device=torch.device('cuda:0'), conv_mode=args.conv_mode)
  eng.set_weights(WL.xavier_params(layers, seed=42, bias_range=0.05, dtype=np.float32))
  out = {'workload': 'configs[2]: inference, {} utterances of 2-15 s, batch {}, greedy decode'.format(args.utterances, args.batch),
         'conv_mode': eng.conv_mode, 'audio_seconds': float(samples.sum() / 16000.0),
         'method': 'per variant: one warm-up pass, then {} timed windows of >= {:g} s each (the pool transcribed back to back as often '
                   'as that takes), wall clock incl. host padding, H2D of every batch and D2H of the decodes; median window reported, '
                   'all windows listed'.format(args.samples, args.min_window),
         'default_pipeline': inference.DEFAULT_PIPELINE}
  results = {}
  for bucket, pipeline in ((True, True), (True, False), (False, True), (False, False)):
    buckets = inference.make_buckets(frames, args.batch) if bucket else [
        list(range(i, min(i + args.batch, len(feats)))) for i in range(0, len(feats), args.batch)]
    inference.transcribe(eng, feats, args.batch, bucket, pipeline)          # warm-up: a long-running service's steady state
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ref_ids, _ = inference.transcribe(eng, feats, args.batch, bucket, pipeline)
    torch.cuda.synchronize()
    once = time.perf_counter() - t0
    rounds = max(1, int(np.ceil(args.min_window / once)))
    rates, windows = [], []
    for _ in range(args.samples):
      t0 = time.perf_counter()
      for _ in range(rounds):
        ids, _ = inference.transcribe(eng, feats, args.batch, bucket, pipeline)
      torch.cuda.synchronize()
      dt = time.perf_counter() - t0
      windows.append(round(dt, 4))
      rates.append(rounds * args.utterances / dt)
    assert ids == ref_ids
    key = ('bucketed' if bucket else 'arrival_order') + ('_pipelined' if pipeline else '_serial_loop')
    results[key] = ids
    med = float(np.median(rates))
    out[key] = {'utterances_per_s': round(med, 1), 'utterances_per_s_min_max': [round(min(rates), 1), round(max(rates), 1)],
                'window_seconds': windows, 'passes_per_window': rounds,
                'padding_overhead': round(inference.padding_overhead(frames, buckets), 4),
                'realtime_factor': round(out['audio_seconds'] * med / args.utterances, 0)}
  assert results['bucketed_pipelined'] == results['bucketed_serial_loop']          # same launches on the same data
  assert results['arrival_order_pipelined'] == results['arrival_order_serial_loop']
  best = max((k for k in out if isinstance(out[k], dict) and 'utterances_per_s' in out[k]), key=lambda k: out[k]['utterances_per_s'])
  out['fastest'] = best
  # single-utterance latency (the SingleInputLoader / live path): eager launch sequence vs the captured HIP graph
  one = feats[0][:201]                                           # a 2 s utterance
  eng.load_batch(one[None], [one.shape[0]])
  for name, fn in (('eager', eng.forward), ('graph', eng.forward_graph)):
    for _ in range(3):
      fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
      fn()
      eng.greedy_decode()
    out.setdefault('single_utterance_2s_latency_ms', {})[name] = round((time.perf_counter() - t0) / 50 * 1e3, 3)
  print(json.dumps(out))


if __name__ == '__main__':
  main()
