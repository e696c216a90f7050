"""Batch assembly and the input pipeline (mirror of speecht/speech_input.py).

The reference builds TF placeholders + a FIFOQueue(capacity=100) fed by Python threads
(speech_input.py:130-218).  Here the queue is a plain ``queue.Queue`` of host batches; the padded
batch layout and the sparse label layout -- the data formats either side of the hot path -- are kept
exactly (speech_input.py:27-69): inputs zero-padded to [B, max_T, input_size], sequence_lengths =
unpadded frame counts, labels as (indices [N,2], values [N], dense_shape [B, max_T]).
"""
import collections
import queue
import threading

import numpy as np

SparseTensorValue = collections.namedtuple('SparseTensorValue', ['indices', 'values', 'dense_shape'])


class StagedBatch:
  """A padded input batch that already lives in device memory: ``tensor`` [B, max_T, C] float32 and the
  event recorded on the copy stream after its H2D transfer (the consumer's stream waits on it)."""

  def __init__(self, tensor, event):
    self.tensor, self.event = tensor, event
    self.shape = tuple(tensor.shape)


class OutOfRangeError(Exception):
  """End-of-data signal, the role tf.errors.OutOfRangeError plays in training.py:92 / evaluation.py:109."""


class Coordinator:
  """The subset of tf.train.Coordinator the executors use (execution.py:54-58)."""

  def __init__(self):
    self._stop = threading.Event()
    self._threads = []

  def should_stop(self):
    return self._stop.is_set()

  def request_stop(self):
    self._stop.set()

  def register_thread(self, thread):
    self._threads.append(thread)

  def join(self, timeout=5.0):
    for t in self._threads:
      t.join(timeout)


class Placeholder:
  """Names an input slot; ``get_inputs()`` returns these, values arrive through ``dequeue()``/feed."""

  def __init__(self, name):
    self.name = name

  def __repr__(self):
    return '<Placeholder {}>'.format(self.name)


class BaseInputLoader:

  def __init__(self, input_size):
    self.input_size = input_size

  def _get_inputs_feed_item(self, input_list):
    """list of [time, input_size] arrays -> (input_tensor [B, max_T, C], sequence_lengths, max_time)."""
    sequence_lengths = np.fromiter((item.shape[0] for item in input_list), dtype=np.int64, count=len(input_list))
    max_time = int(sequence_lengths.max())
    input_tensor = np.zeros((len(input_list), max_time, self.input_size), dtype=np.float32)
    for row, item in zip(input_tensor, input_list):
      row[:item.shape[0]] = item
    return input_tensor, sequence_lengths, max_time

  @staticmethod
  def _get_labels_feed_item(label_list, max_time):
    """list of id sequences -> SparseTensorValue with dense_shape [B, max_time] (the reference
    uses the INPUT max_time here, speech_input.py:58)."""
    counts = [len(label) for label in label_list]
    total = sum(counts)
    indices = np.empty((total, 2), dtype=np.int64)
    indices[:, 0] = np.repeat(np.arange(len(label_list)), counts)
    indices[:, 1] = np.concatenate([np.arange(c) for c in counts]) if total else np.empty(0, np.int64)
    values = np.fromiter((v for label in label_list for v in label), dtype=np.int64, count=total)
    return SparseTensorValue(indices, values, np.array([len(label_list), max_time], dtype=np.int64))

  def get_inputs(self):
    raise NotImplementedError()

  def get_feed_dict(self):
    return None

  def dequeue(self):
    """Next (inputs, sequence_lengths, labels|None) for one model step."""
    raise NotImplementedError()


def sparse_to_label_lists(labels):
  """SparseTensorValue -> one int array of label ids per utterance (vectorised: a batch of 32 ten-second
  utterances carries ~5 000 label entries and this runs on the critical path of every step)."""
  batch = int(labels.dense_shape[0])
  indices = np.asarray(labels.indices).reshape(-1, 2)
  values = np.asarray(labels.values)
  rows, pos = indices[:, 0], indices[:, 1]
  if rows.size and (np.any(np.diff(rows) < 0) or np.any((np.diff(rows) == 0) & (np.diff(pos) < 0))):
    order = np.lexsort((pos, rows))                  # canonical row-major order
    rows, values = rows[order], values[order]
  counts = np.bincount(rows.astype(np.int64), minlength=batch)[:batch] if rows.size else np.zeros(batch, np.int64)
  return np.split(values, np.cumsum(counts)[:-1]) if batch else []


def bucket_by_length(samples, batch_size, window=16, seed=None):
  """Opt-in length-bucketed shuffled sampler (not reference behaviour: the reference batches consecutive samples of its
  shuffled generator, speech_input.py:169-179, and pads each batch to its own longest member -- ~40 % padding on 2-15 s
  speech).  Reads ``window`` batches' worth of samples from ``samples``, sorts them by frame count, cuts the sorted run into
  batches of neighbouring lengths and yields those batches in shuffled order, sample by sample -- so that the loader's
  ``zip(*[iter]*B)`` regroups exactly them.  Every sample is yielded once; a final short window is sorted and yielded too
  (the loader drops its last partial batch as always).  SURVEY F7: padding is never masked, so an utterance's logits near
  its end depend on the padded length of its batch -- bucketing changes those tails like any other batch composition does."""
  import random as _random
  rng = _random.Random(seed)
  it = iter(samples)
  while True:
    chunk = []
    for sample in it:
      chunk.append(sample)
      if len(chunk) == window * batch_size:
        break
    if not chunk:
      return
    chunk.sort(key=lambda s: s[0].shape[0])
    batches = [chunk[i:i + batch_size] for i in range(0, len(chunk), batch_size)]
    tail = batches.pop() if len(batches[-1]) < batch_size else None
    rng.shuffle(batches)
    for b in batches:
      for sample in b:
        yield sample
    if tail:
      for sample in tail:
        yield sample
    if len(chunk) < window * batch_size:
      return


class SingleInputLoader(BaseInputLoader):
  """Feeds one utterance per step (speech_input.py:79-127), used for live / ad-hoc inference."""

  def __init__(self, input_size):
    super().__init__(input_size)
    self.speech_input = None
    self.inputs = Placeholder('inputs')
    self.sequence_lengths = Placeholder('sequence_lengths')

  def get_inputs(self):
    return self.inputs, self.sequence_lengths, None

  def get_feed_dict(self):
    if self.speech_input is None:
      raise ValueError('Speech input must be provided using `set_input` first!')
    input_tensor, sequence_lengths, _ = self._get_inputs_feed_item([self.speech_input])
    self.speech_input = None
    return {self.inputs: input_tensor, self.sequence_lengths: sequence_lengths}

  def set_input(self, speech_input):
    self.speech_input = speech_input

  def dequeue(self):
    feed = self.get_feed_dict()
    return feed[self.inputs], feed[self.sequence_lengths], None


class InputBatchLoader(BaseInputLoader):
  """Background threads assemble padded batches into a bounded queue (capacity 100 like the
  reference's FIFOQueue); the final partial batch is dropped (``zip(*[iter]*B)``,
  speech_input.py:169-179) and ``max_steps`` caps the batches produced."""

  CAPACITY = 100
  DEVICE_PREFETCH = 3          # batches staged in HBM ahead of the consumer (copy stream, own thread)

  def __init__(self, input_size, batch_size, data_generator_creator, max_steps=None, shard=None):
    """``shard = (rank, world)`` (not a reference argument; data-parallel training): the loader forms the GLOBAL
    batches of ``batch_size * world`` consecutive samples -- the batches one process with that batch size would see, every rank
    from an identically ordered generator -- pads each to the global batch's longest member (padding is never masked, SURVEY F7:
    an utterance's logits depend on the padded length, so the shard must be padded like the whole batch) and keeps rows
    ``[rank * batch_size, (rank + 1) * batch_size)``."""
    super().__init__(input_size)
    self.batch_size = batch_size
    self.shard = tuple(shard) if shard is not None and int(shard[1]) > 1 else None
    self.data_generator_creator = data_generator_creator
    self.steps_left = max_steps
    self.inputs = Placeholder('inputs')
    self.sequence_lengths = Placeholder('sequence_lengths')
    self.labels = Placeholder('labels')
    self._queue = queue.Queue(maxsize=self.CAPACITY)
    self._closed = threading.Event()
    self._lock = threading.Lock()
    self._staged = None          # queue of device-resident batches once a stager thread runs
    self._staged_closed = threading.Event()

  def get_inputs(self):
    return self.inputs, self.sequence_lengths, self.labels

  def _batch(self, iterable):
    return zip(*([iter(iterable)] * (self.batch_size * (self.shard[1] if self.shard else 1))))

  def _feed_item(self, sample_batch):
    """One (inputs, sequence_lengths, labels) queue item from a tuple of samples; with ``shard`` this rank's rows of it."""
    input_list, label_list = zip(*sample_batch)
    if self.shard is None:
      input_tensor, sequence_lengths, max_time = self._get_inputs_feed_item(input_list)
      return input_tensor, sequence_lengths, self._get_labels_feed_item(label_list, max_time)
    rank = int(self.shard[0])
    lo, hi = rank * self.batch_size, (rank + 1) * self.batch_size
    max_time = max(item.shape[0] for item in input_list)                # of the GLOBAL batch
    mine = input_list[lo:hi]
    sequence_lengths = np.fromiter((item.shape[0] for item in mine), dtype=np.int64, count=len(mine))
    input_tensor = np.zeros((len(mine), max_time, self.input_size), dtype=np.float32)
    for row, item in zip(input_tensor, mine):
      row[:item.shape[0]] = item
    return input_tensor, sequence_lengths, self._get_labels_feed_item(label_list[lo:hi], max_time)

  def _enqueue(self, sess, coord):
    try:
      for sample_batch in self._batch(self.data_generator_creator()):
        item = self._feed_item(sample_batch)
        while not (self._closed.is_set() or coord.should_stop()):
          try:
            self._queue.put(item, timeout=0.1)
            break
          except queue.Full:
            continue
        with self._lock:
          if self.steps_left is not None:
            self.steps_left -= 1
            if self.steps_left == 0:
              break
        if coord.should_stop() or self._closed.is_set():
          break
    finally:
      self._closed.set()      # like sess.run(queue.close()): the first finished feeder closes the queue

  def start_threads(self, sess, coord, n_threads=1):
    threads = []
    for _ in range(n_threads):
      t = threading.Thread(target=self._enqueue, args=(sess, coord), daemon=True)
      t.start()
      coord.register_thread(t)
      threads.append(t)
    device = getattr(sess, 'device', None)
    if device is not None:
      import torch
      device = torch.device(device)          # Session.device is a plain string such as 'cuda:0'
    if device is not None and device.type == 'cuda':
      # the H2D copy of batch k+1.. overlaps the kernels of batch k (separate HIP stream, own thread)
      self._staged = queue.Queue(maxsize=self.DEVICE_PREFETCH)
      t = threading.Thread(target=self._stage, args=(device, coord), daemon=True)
      t.start()
      coord.register_thread(t)
      threads.append(t)
    return threads

  def _dequeue_host(self):
    while True:
      try:
        return self._queue.get(timeout=0.05)
      except queue.Empty:
        if self._closed.is_set() and self._queue.empty():
          raise OutOfRangeError('input queue is closed and has insufficient elements')

  def _stage(self, device, coord):
    import torch
    stream = torch.cuda.Stream(device)
    try:
      while not coord.should_stop():
        try:
          inputs, seq_lens, labels = self._dequeue_host()
        except OutOfRangeError:
          break
        with torch.cuda.stream(stream):
          tensor = torch.as_tensor(inputs).to(device)          # staged copy; the GIL is released meanwhile
          event = torch.cuda.Event()
          event.record(stream)
        item = (StagedBatch(tensor, event), seq_lens, labels)
        while not coord.should_stop():
          try:
            self._staged.put(item, timeout=0.1)
            break
          except queue.Full:
            continue
    finally:
      self._staged_closed.set()

  def dequeue(self):
    if self._staged is None:
      return self._dequeue_host()
    while True:
      try:
        return self._staged.get(timeout=0.05)
      except queue.Empty:
        if self._staged_closed.is_set() and self._staged.empty():
          raise OutOfRangeError('input queue is closed and has insufficient elements')
