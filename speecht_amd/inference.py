"""Batch inference with length bucketing (BASELINE config 3: conv stack + CTC greedy decode on
variable-length utterances).

The reference pads every batch to its longest member and masks nothing (speech_input.py:37-45,
SURVEY F7), so the ~73 output frames before an utterance's end depend on the padded length of the
batch it happens to be in.  Bucketing by length keeps padding -- and therefore wasted convolution
work -- small; results for an utterance equal what the reference would produce for the SAME batch
composition (that is what the tests check), not for an arbitrary one.
"""
import queue
import threading

import numpy as np

from . import vocabulary


def make_buckets(lengths, batch_size):
  """Indices sorted by length, cut into consecutive batches: returns a list of index lists."""
  order = np.argsort(np.asarray(lengths), kind='stable')
  return [order[i:i + batch_size].tolist() for i in range(0, len(order), batch_size)]


def padding_overhead(lengths, buckets):
  """Fraction of padded frames that are padding, for reporting."""
  lengths = np.asarray(lengths)
  padded = sum(len(b) * int(lengths[b].max()) for b in buckets)
  return 1.0 - float(lengths.sum()) / padded


def _plan(features, batch_size, bucket):
  lengths = [f.shape[0] for f in features]
  buckets = make_buckets(lengths, batch_size) if bucket else [
      list(range(i, min(i + batch_size, len(features)))) for i in range(0, len(features), batch_size)]
  return lengths, buckets


# Whether `transcribe` overlaps host staging / read-back with the GPU by default.  Decided by measurement
# (scripts/bench_inference.py, profiles/r3_inference_config3_*.json: 2 048 utterances, windows >= 0.5 s, median of 5).
DEFAULT_PIPELINE = True
# Decoder streams of the pipelined beam search: a search (one wavefront per utterance) takes a little longer than the forward
# pass of the next batch, so two consecutive batches are searched side by side on the decoder's compute units
# (scripts/bench_decode.py, profiles/r4_decode_config5.json).
BEAM_DECODERS = 2


def transcribe(engine, features, batch_size=64, bucket=True, pipeline=None, beam_width=None):
  """features: list of [T_i, input_size] arrays.  Returns (list of id lists, list of strings) in the
  input order, decoded greedily (speech_model.py:113-115) batch by batch -- or, with ``beam_width``, by the LM-free prefix
  beam search (the reference's beam search needs its KenLM fork, speech_model.py:101-111; configs[4] asks for beam 16).

  pipeline=True (default: ``DEFAULT_PIPELINE``) overlaps the three stages of consecutive batches: a stager thread pads batch k+1
  into pinned host memory and copies it to the device on its own stream while the GPU runs batch k, and the
  transcripts of batch k are read back (pinned, asynchronous) after batch k+1 has been enqueued.  Same
  launches on the same data as the serial loop, hence identical ids.  With ``beam_width`` the search of batch k runs on a
  decoder stream -- on compute units of its own, `engine.decoder_streams` -- under the forward passes of batches k+1 and
  k+2: one wavefront per utterance searches a little longer than the rest of the chip convolves (configs[4]: 3.9 against
  3.6 ms), so two searches are in flight."""
  if not features:
    return [], []
  if pipeline is None:
    pipeline = DEFAULT_PIPELINE
  lengths, buckets = _plan(features, batch_size, bucket)
  ids_out = [None] * len(features)
  if not pipeline:
    for idx in buckets:
      max_t = max(lengths[i] for i in idx)
      x = np.zeros((len(idx), max_t, features[0].shape[1]), dtype=np.float32)
      for row, i in enumerate(idx):
        x[row, :lengths[i]] = features[i]
      engine.load_batch(x, [lengths[i] for i in idx])
      engine.forward()
      ids, _ = engine.beam_search_decode(beam_width) if beam_width else engine.greedy_decode()
      for row, i in enumerate(idx):
        ids_out[i] = ids[row]
    return ids_out, [vocabulary.ids_to_sentence(s) for s in ids_out]

  def collect(handle, idx):
    res = handle.result()
    for row, ids in enumerate(res[0] if beam_width else res):
      ids_out[idx[row]] = ids

  import contextlib
  import torch
  depth = 1                                      # batches whose transcripts are still on their way when the next is enqueued
  if beam_width:
    from .engine import decoder_streams
    compute_stream, decode_stream = decoder_streams(engine.device, BEAM_DECODERS)
    depth = len(decode_stream)
    torch.cuda.synchronize(engine.device)        # weights / buffers written on other streams are in place
    on_compute = lambda: torch.cuda.stream(compute_stream)
  else:
    on_compute = contextlib.nullcontext
  with _PIPELINE_LOCK:               # the pinned staging ring of a device serves one pipeline at a time
    stager = _Stager(engine.device, features, lengths, buckets)
    stager.start()
    pending = []
    try:
      for idx in buckets:
        staged = stager.get()
        with on_compute():
          engine.load_batch(staged, [lengths[i] for i in idx])
          engine.forward()
          handle = engine.beam_search_decode_async(beam_width, decode_stream) if beam_width else engine.greedy_decode_async()
        pending.append((handle, idx))
        if len(pending) > depth:
          collect(*pending.pop(0))
      for item in pending:
        collect(*item)
    finally:
      stager.close()
      if beam_width:
        torch.cuda.synchronize(engine.device)    # nothing of this call is left on the masked streams
  return ids_out, [vocabulary.ids_to_sentence(s) for s in ids_out]


_STREAMS = {}      # device -> the stagers' copy stream
_PINNED = {}       # (device, depth) -> ring of pinned staging buffers, grow-only
_PIPELINE_LOCK = threading.Lock()


class _Stager(threading.Thread):
  """Pads batches into a small ring of pinned host buffers and copies them to the device on a private stream,
  at most ``depth`` batches ahead of the consumer."""

  def __init__(self, device, features, lengths, buckets, depth=2):
    super().__init__(daemon=True)
    import torch
    self.torch = torch
    self.device, self.features, self.lengths, self.buckets = device, features, lengths, buckets
    self.queue = queue.Queue(maxsize=depth)
    # one copy stream per device for the life of the process (creating a stream per call costs the short pools)
    self.stream = _STREAMS.get(str(device)) or _STREAMS.setdefault(str(device), torch.cuda.Stream(device))
    # [pinned buffer, event of its last H2D]; the buffers outlive the call (pinning host memory costs
    # milliseconds and synchronises the device) and are sized for the largest batch of the plan up front
    self.ring = _PINNED.setdefault((str(device), depth), [[None, None] for _ in range(depth + 2)])
    width = features[0].shape[1]
    need = max(len(idx) * max(lengths[i] for i in idx) for idx in buckets) * width
    for slot in self.ring:
      if slot[0] is None or slot[0].numel() < need:
        if slot[1] is not None:
          slot[1].synchronize()
        slot[0] = torch.empty(need + need // 4, dtype=torch.float32, pin_memory=True)
    self.stop = threading.Event()

  def run(self):
    torch = self.torch
    from .speech_input import StagedBatch
    try:
      width = self.features[0].shape[1]
      for k, idx in enumerate(self.buckets):
        if self.stop.is_set():
          return
        slot = self.ring[k % len(self.ring)]
        max_t = max(self.lengths[i] for i in idx)
        n = len(idx) * max_t * width
        if slot[1] is not None:
          slot[1].synchronize()                                   # the copy that last read this buffer is done
        host = slot[0][:n].view(len(idx), max_t, width)
        x = host.numpy()
        for row, i in enumerate(idx):
          x[row, :self.lengths[i]] = self.features[i]
          x[row, self.lengths[i]:] = 0.0
        with torch.cuda.stream(self.stream):
          dev = torch.empty((len(idx), max_t, width), dtype=torch.float32, device=self.device)
          dev.copy_(host, non_blocking=True)
          event = torch.cuda.Event()
          event.record(self.stream)
        slot[1] = event
        self._put(StagedBatch(dev, event))
    except BaseException as e:                                    # surfaces in the consumer's get()
      self._put(e)

  def _put(self, item):
    while not self.stop.is_set():
      try:
        self.queue.put(item, timeout=0.1)
        return
      except queue.Full:
        continue

  def get(self):
    item = self.queue.get()
    if isinstance(item, BaseException):
      raise item
    return item

  def close(self):
    self.stop.set()
    self.join(timeout=5.0)
