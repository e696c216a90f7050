#!/usr/bin/env python3
"""Training throughput in the reference's REAL regime: every batch padded to its own longest member, in arrival order from
a shuffled generator (speech_input.py:37-45,169-179) -- a new (B, max_T) shape almost every step -- through the
reference-shaped API (InputBatchLoader feeder threads -> SpeechModel.step: host padding, queue, H2D, label upload, loss
read-back).  Seeded pool of 2-15 s synthetic utterances (n ~ U{32 000..240 000} samples at 16 kHz, or the
reference's own 22 050 Hz framing with --rate 22050), batch 32 or the CLI default 64, 80 or 128 mel features.

Orders:  arrival  = the reference's behaviour (the pool shuffled, batches of consecutive samples);
         bucketed = the opt-in length-bucketed shuffled sampler (speech_input.bucket_by_length);
         fixed    = every utterance 10 s (the bench.py shape) through the same API, for the comparison the report makes.
Reports utterances/s, audio-seconds/s, the padding overhead, the number of distinct shapes, the host's time inside
`engine._ensure_shape` and the number of device allocations after warm-up."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speecht_amd import speech_input, speech_model    # noqa: E402
from tests import workloads as WL                      # noqa: E402


class Flags:
  command, learning_rate, learning_rate_decay_factor, max_gradient_norm, momentum = 'train', 1e-4, 0.0, 5.0, 0.9
  log_dir, run_name, run_type = '/tmp/speecht_varlen_bench', 'bench', 'train'


def make_pool(count, mels, rate, fixed_seconds=None, seed=0):
  rng = np.random.default_rng(seed)
  pool = []
  for i in range(count):
    seconds = fixed_seconds if fixed_seconds else float(rng.integers(32000, 240001)) / 16000.0
    frames = 1 + int(seconds * rate) // 160
    feats = WL.synthetic_features(seed * 100000 + i, frames, mels).astype(np.float32)
    labels = WL.make_labels(seed * 100000 + i, int(round(15 * seconds)), (frames // 2))
    pool.append((feats, labels, seconds))
  return pool


def run(order, batch, mels, rate, steps, warmup, pool_size, conv_mode, seed=0):
  if conv_mode:
    os.environ['ST_CONV_MODE'] = conv_mode
  pool = make_pool(pool_size, mels, rate, fixed_seconds=10.0 if order == 'fixed' else None, seed=seed)
  seconds_of = {id(p[0]): p[2] for p in pool}
  served = []          # (sum of true seconds, sum of true frames, padded frames) per batch, in the order the batches were built

  def generator():
    rng = np.random.default_rng(seed + 1)
    while True:
      for k in rng.permutation(len(pool)):
        yield pool[k][0], pool[k][1]

  creator = generator
  if order == 'bucketed':
    creator = lambda: speech_input.bucket_by_length(generator(), batch, window=16, seed=seed + 2)   # noqa: E731
  loader = speech_input.InputBatchLoader(mels, batch, creator)
  feed_item = loader._get_inputs_feed_item

  def counting_feed_item(input_list):
    out = feed_item(input_list)
    served.append((sum(seconds_of[id(f)] for f in input_list), int(out[1].sum()), len(input_list) * out[2]))
    return out
  loader._get_inputs_feed_item = counting_feed_item
  model = speech_model.create_default_model(Flags(), mels, loader)
  out = None
  with speech_model.Session('cuda:0') as sess:
    model.init_session(sess)
    eng = model.engine
    if hasattr(eng, 'reserve'):
      eng.reserve(batch, max(p[0].shape[0] for p in pool))
    ensure = eng._ensure_shape
    acc = dict(seconds=0.0, calls=0, changes=0)

    def timed_ensure(b, t):
      changed = eng._shape != (b, t)
      t0 = time.perf_counter()
      ensure(b, t)
      acc['seconds'] += time.perf_counter() - t0
      acc['calls'] += 1
      acc['changes'] += int(changed)
    eng._ensure_shape = timed_ensure
    coord = speech_input.Coordinator()
    loader.start_threads(sess=sess, coord=coord, n_threads=1)      # one feeder: the batch order is the generator's
    for _ in range(warmup):
      model.step(sess)
    torch.cuda.synchronize()
    acc.update(seconds=0.0, calls=0, changes=0)
    gen0 = eng._storage.generation
    mem0 = torch.cuda.memory_stats().get('allocation.all.allocated', 0)
    shapes = set()
    host, seen = [], []
    t0 = time.perf_counter()
    for _ in range(steps):
      h0 = time.perf_counter()
      model.step(sess)
      host.append(time.perf_counter() - h0)
      shapes.add(eng._shape)
      seen.append(eng._shape)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    coord.request_stop()
    used = served[warmup:warmup + steps]
    true_s = sum(u[0] for u in used)
    pad = 1.0 - sum(u[1] for u in used) / float(sum(u[2] for u in used))
    out = dict(order=order, batch=batch, mels=mels, rate=rate, conv_mode=eng.conv_mode, steps=steps,
               ms_per_step=round(dt / steps * 1e3, 3), utterances_per_s=round(batch * steps / dt, 1),
               audio_seconds_per_s=round(true_s / dt, 1), padding_fraction=round(pad, 4), distinct_shapes=len(shapes),
               shape_changes=acc['changes'], ensure_shape_ms_per_change=round(acc['seconds'] / max(acc['changes'], 1) * 1e3, 3),
               ensure_shape_ms_per_step=round(acc['seconds'] / steps * 1e3, 3),
               storage_reallocations_after_warmup=eng._storage.generation - gen0,
               torch_allocations_after_warmup=torch.cuda.memory_stats().get('allocation.all.allocated', 0) - mem0,
               step_call_ms_median=round(float(np.median(host)) * 1e3, 3),
               # the five longest step() calls and the (batch, frames) they ran at: a mean far above the median is a few of these
               slowest_step_calls=[dict(ms=round(host[k] * 1e3, 2), step=int(k), shape=[int(v) for v in seen[k]]) for k in np.argsort(host)[::-1][:5]])
  return out


def sweep(batch, mels, conv_mode, lengths, steps=12):
  """Fixed-shape training steps (engine loop, data resident) at each padded length: which shapes run below the rate of the
  bench shape -- per padded audio second -- and by how much."""
  import bench
  from speecht_amd.engine import Wav2LetterEngine
  layers = WL.w2l_layers(mels)
  eng = Wav2LetterEngine(layers, device='cuda:0', conv_mode=conv_mode)
  eng.init_xavier(seed=1)
  rows = []
  for frames in lengths:
    x, seq, labels = WL.make_batch([frames] * batch, mels, seed=7)
    feed = bench.HostFeed(eng, x, seq, labels)
    for _ in range(3):
      bench.train_step(eng, feed, None, 1e-4, batch)
    torch.cuda.synchronize()
    marks = []
    t0 = time.perf_counter()
    for k in range(steps):
      if k >= 2:
        marks[k - 2].synchronize()
      bench.train_step(eng, feed, None, 1e-4, batch)
      ev = torch.cuda.Event()
      ev.record()
      marks.append(ev)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    eng.discard_staged_batch(feed.staged)          # (the feed keeps one batch staged ahead: release its buffer for the next shape's)
    blocks = -(-((frames + 1) // 2) // 64)
    rows.append(dict(frames=frames, out_frames=(frames + 1) // 2, blocks=blocks, rows=batch * blocks, ms_per_step=round(ms, 3),
                     us_per_padded_audio_second=round(ms * 1e3 / (batch * (frames - 1) / 100.0), 2),
                     spectral_layers=sorted(getattr(eng, 'fft', {}) or getattr(eng, 'fftb', {}))))
    print(json.dumps(rows[-1]))
    sys.stdout.flush()
  return rows


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--sweep', type=int, nargs='*', default=None, help='fixed-shape step time at these padded lengths (frames) instead of the pool runs')
  ap.add_argument('--steps', type=int, default=60)
  ap.add_argument('--warmup', type=int, default=24)
  ap.add_argument('--pool', type=int, default=512)
  ap.add_argument('--batch', type=int, nargs='+', default=[32])
  ap.add_argument('--mels', type=int, nargs='+', default=[80])
  ap.add_argument('--rate', type=int, default=16000, help='frames = 1 + seconds * rate // 160 (the reference loads at 22 050 Hz)')
  ap.add_argument('--orders', nargs='+', default=['fixed', 'arrival', 'bucketed'])
  ap.add_argument('--conv-mode', default=None)
  ap.add_argument('--out', default=None)
  args = ap.parse_args()
  if args.sweep is not None:
    lengths = args.sweep or list(range(201, 1502, 100))
    rows = sweep(args.batch[0], args.mels[0], args.conv_mode, lengths)
    if args.out:
      with open(args.out, 'w') as f:
        json.dump(rows, f, indent=1)
    sys.stdout.flush()
    os._exit(0)
  results = []
  for mels in args.mels:
    for batch in args.batch:
      fixed = None
      for order in args.orders:
        if order == 'bucketed' and not hasattr(speech_input, 'bucket_by_length'):
          continue
        r = run(order, batch, mels, args.rate, args.steps, args.warmup, args.pool, args.conv_mode)
        if order == 'fixed':
          fixed = r
        elif fixed is not None:
          # what the fixed-shape rate would give on this order's padded frames: the share of the padded work that is real audio
          r['vs_fixed_shape_times_one_minus_padding'] = round(
              r['audio_seconds_per_s'] / (fixed['audio_seconds_per_s'] * (1.0 - r['padding_fraction'])), 4)
        results.append(r)
        print(json.dumps(r))
        sys.stdout.flush()
  if args.out:
    with open(args.out, 'w') as f:
      json.dump(results, f, indent=1)
  sys.stdout.flush()
  os._exit(0)


if __name__ == '__main__':
  main()
