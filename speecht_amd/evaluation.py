"""`speecht-cli evaluate` -- greedy decoding of a data set with letter / word error statistics.

Output lines and statistics follow speecht/evaluation.py (``validation average loss ...``,
``expected: ...``, ``decoded: ...``, ``LED: .. LER: .. WED: .. WER: ..``, ``Global statistics``).
"""
import itertools
import math

from . import editdistance, execution, speech_input, speech_model, vocabulary

STATS_FORMAT = 'LED: {} LER: {:.2f} WED: {} WER: {:.2f}'


class EvalStatistics:
  """Edit distances of the last decoding plus running sums over all of them
  (letter = per character, word = per whitespace-separated token; rates are relative to the
  expected string, evaluation.py:40-49)."""

  _FIELDS = ('letter_edit_distance', 'letter_error_rate', 'word_edit_distance', 'word_error_rate')

  def __init__(self):
    self.decodings_counter = 0
    for name in self._FIELDS:
      setattr(self, name, 0)
      setattr(self, 'sum_' + name, 0)

  def track_decoding(self, decoded_str, expected_str):
    expected_tokens, decoded_tokens = expected_str.split(), decoded_str.split()
    self.letter_edit_distance = editdistance.eval(expected_str, decoded_str)
    self.word_edit_distance = editdistance.eval(expected_tokens, decoded_tokens)
    self.letter_error_rate = self.letter_edit_distance / len(expected_str)
    self.word_error_rate = self.word_edit_distance / len(expected_tokens)
    for name in self._FIELDS:
      setattr(self, 'sum_' + name, getattr(self, 'sum_' + name) + getattr(self, name))
    self.decodings_counter += 1

  def __getattr__(self, name):
    # global_<field> = mean of <field> over all tracked decodings
    if name.startswith('global_') and name[len('global_'):] in self._FIELDS:
      return getattr(self, 'sum_' + name[len('global_'):]) / self.decodings_counter
    raise AttributeError(name)

  def last_line(self):
    return STATS_FORMAT.format(*(getattr(self, f) for f in self._FIELDS))

  def global_line(self):
    return STATS_FORMAT.format(*(getattr(self, 'global_' + f) for f in self._FIELDS))


class Evaluation(execution.DatasetExecutor):

  def create_sample_generator(self, limit_count: int):
    return self.reader.load_samples(self.flags.dataset, loop_infinitely=False, limit_count=limit_count,
                                    feature_type=self.flags.feature_type, shuffle_seed=self.shuffle_seed)

  def get_loader_limit_count(self):
    return self.flags.batch_size * self.world * self.flags.step_count

  def get_max_steps(self):
    return self.flags.step_count if self.flags.step_count else None

  def run(self):
    stats = EvalStatistics()
    with speech_model.Session(getattr(self.flags, 'device', 'cuda:0')) as sess, self.quiet_unless_rank0():
      model = self.create_model(sess)
      if self.world > 1:
        # replicas: every rank evaluates its rows of each global batch with the same checkpoint; the loss is averaged over
        # ranks per step, the statistics are gathered at the end (rank 0 prints its own rows and the global line)
        model.enable_data_parallel()
      print('Starting input pipeline')
      coordinator = self.start_pipeline(sess)
      print('Begin evaluation')
      try:
        for step in (range(self.flags.step_count) if self.flags.step_count else itertools.count()):
          if coordinator.should_stop():
            break
          self.run_step(model, sess, stats, save=self.flags.should_save and step == 0)
      except speech_input.OutOfRangeError:
        print('Done evaluating -- step limit reached')
      finally:
        coordinator.request_stop()
      stats = self.gather_statistics(stats)
      self.print_global_statistics(stats)
      coordinator.join()
    return stats

  def gather_statistics(self, stats):
    """Data-parallel evaluation: the running sums of every rank added up (the same object on a single process)."""
    if self.world <= 1:
      return stats
    import torch.distributed as dist
    mine = dict(decodings_counter=stats.decodings_counter, **{'sum_' + f: getattr(stats, 'sum_' + f) for f in stats._FIELDS})
    every = [None] * self.world
    dist.all_gather_object(every, mine)
    for key in mine:
      setattr(stats, key, sum(e[key] for e in every))
    return stats

  @staticmethod
  def print_global_statistics(stats):
    print('Global statistics')
    print(stats.global_line())

  def run_step(self, model, sess, stats: EvalStatistics, save: bool, verbose=True, feed_dict=None):
    """One batch: loss + greedy decoding + labels, then per-utterance statistics."""
    step_index = model.global_step.eval()
    fetched = model.step(sess, update=False, decode=True, return_label=True, summary=bool(save), feed_dict=feed_dict)
    avg_loss, decoded, label = fetched[:3]
    if save:
      model.summary_writer.add_summary(fetched[3], step_index)
    if verbose:
      perplexity = math.exp(float(avg_loss)) if avg_loss < 300 else float('inf')
      print('validation average loss {:.2f} perplexity {:.2f}'.format(avg_loss, perplexity))
    # Pairing of labels with decodings.  Default: the reference's own walk (evaluation.py:144-151) -- both sparse
    # tensors go through extract_decoded_ids and are consumed in lock-step, with its quirk that an utterance
    # decoding to the empty string yields nothing (every later pairing shifts, and ``next`` raises StopIteration
    # when the decodings run out first).  ``--pair-by-row`` (flags.pair_by_row; not a reference flag) pairs by batch
    # row instead, which is what the statistics mean to measure.
    if getattr(self.flags, 'pair_by_row', False):
      label_rows = self.rows_by_batch(label)
      path_iters = [iter(self.rows_by_batch(path)) for path in decoded]
    else:
      label_rows = self.extract_decoded_ids(label)
      path_iters = [self.extract_decoded_ids(path) for path in decoded]
      # the quirk, said out loud: every utterance that decoded to the empty string (routine for early checkpoints)
      # shifts the pairs behind it, and the walk ends early when the decodings run out
      empty = sum(1 for path in decoded for row in self.rows_by_batch(path) if not row)
      if empty:
        print('warning: {} decoding(s) of this batch are empty; the reference pairing (evaluation.py:144-151) skips them, '
              'so later expected/decoded pairs are shifted -- pass --pair-by-row to pair by utterance'.format(empty))
    for label_ids in label_rows:
      expected = vocabulary.ids_to_sentence(label_ids)
      if verbose:
        print('expected: {}'.format(expected))
      for path in path_iters:
        try:
          ids = next(path)
        except StopIteration:
          # the reference dies here with a bare StopIteration out of run_step; same outcome, with the reason
          raise RuntimeError('ran out of decodings before labels: an utterance of this batch decoded to the empty string and '
                             'the reference\'s lock-step pairing (evaluation.py:144-151) cannot continue; '
                             'use --pair-by-row') from None
        hypothesis = vocabulary.ids_to_sentence(ids)
        stats.track_decoding(hypothesis, expected)
        if verbose:
          print('decoded: {}'.format(hypothesis))
          print(stats.last_line())

  @staticmethod
  def rows_by_batch(sparse_tensor):
    """Sparse (indices, values, dense_shape) -> one id list per batch row (empty rows included)."""
    rows = [[] for _ in range(int(sparse_tensor.dense_shape[0]))]
    for (batch_id, _), value in zip(sparse_tensor.indices, sparse_tensor.values):
      rows[int(batch_id)].append(value)
    return rows

  @staticmethod
  def extract_decoded_ids(sparse_tensor):
    """Generator with the reference's semantics (evaluation.py:160-171), quirk included: a row is only
    emitted once a LATER row starts, so an empty row yields nothing and shifts what follows."""
    current, current_row = [], 0
    for position, (batch_id, _) in enumerate(sparse_tensor.indices):
      if batch_id > current_row:
        yield current
        current, current_row = [], batch_id
      current.append(sparse_tensor.values[position])
    yield current
