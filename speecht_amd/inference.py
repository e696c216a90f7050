"""Batch inference with length bucketing (BASELINE config 3: conv stack + CTC greedy decode on
variable-length utterances).

The reference pads every batch to its longest member and masks nothing (speech_input.py:37-45,
SURVEY F7), so the ~73 output frames before an utterance's end depend on the padded length of the
batch it happens to be in.  Bucketing by length keeps padding -- and therefore wasted convolution
work -- small; results for an utterance equal what the reference would produce for the SAME batch
composition (that is what the tests check), not for an arbitrary one.
"""
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


def transcribe(engine, features, batch_size=64, bucket=True):
  """features: list of [T_i, input_size] arrays.  Returns (list of id lists, list of strings) in the
  input order, decoded greedily (speech_model.py:113-115) batch by batch."""
  lengths = [f.shape[0] for f in features]
  buckets = make_buckets(lengths, batch_size) if bucket else [
      list(range(i, min(i + batch_size, len(features)))) for i in range(0, len(features), batch_size)]
  ids_out = [None] * len(features)
  for idx in buckets:
    max_t = max(lengths[i] for i in idx)
    x = np.zeros((len(idx), max_t, features[0].shape[1]), dtype=np.float32)
    for row, i in enumerate(idx):
      x[row, :lengths[i]] = features[i]
    engine.load_batch(x, [lengths[i] for i in idx])
    engine.forward()
    ids, _ = engine.greedy_decode()
    for row, i in enumerate(idx):
      ids_out[i] = ids[row]
  return ids_out, [vocabulary.ids_to_sentence(s) for s in ids_out]
