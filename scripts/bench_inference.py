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
  ap.add_argument('--utterances', type=int, default=512)
  ap.add_argument('--batch', type=int, default=64)
  ap.add_argument('--conv-mode', default=None)
  args = ap.parse_args()
  rng = np.random.default_rng(3)
  samples = rng.integers(32000, 240001, args.utterances)            # 2..15 s at 16 kHz
  frames = 1 + samples // 160
  feats = [rng.standard_normal((int(t), 80)).astype(np.float32) for t in frames]
  layers = WL.w2l_layers(80)
  eng = Wav2LetterEngine(layers, device=torch.device('cuda:0'), conv_mode=args.conv_mode)
  eng.set_weights(WL.xavier_params(layers, seed=42, bias_range=0.05, dtype=np.float32))
  out = {'workload': 'configs[2]: inference, {} utterances of 2-15 s, batch {}, greedy decode'.format(args.utterances, args.batch),
         'conv_mode': eng.conv_mode, 'audio_seconds': float(samples.sum() / 16000.0)}
  for bucket, pipeline in ((True, True), (False, True), (True, False)):
    buckets = inference.make_buckets(frames, args.batch) if bucket else [
        list(range(i, min(i + args.batch, len(feats)))) for i in range(0, len(feats), args.batch)]
    inference.transcribe(eng, feats, args.batch, bucket, pipeline)          # warm-up: a long-running service's steady state
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ids, _ = inference.transcribe(eng, feats, args.batch, bucket, pipeline)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    key = ('bucketed' if bucket else 'arrival_order') + ('' if pipeline else '_serial_loop')
    out[key] = {'utterances_per_s': round(args.utterances / dt, 1), 'seconds': round(dt, 3),
                'padding_overhead': round(inference.padding_overhead(frames, buckets), 4),
                'realtime_factor': round(out['audio_seconds'] / dt, 0)}
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
