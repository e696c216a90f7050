"""`speecht-cli evaluate`: decode, compare, report LED/LER/WED/WER (mirror of speecht/evaluation.py)."""
import itertools

import numpy as np

from . import editdistance, vocabulary
from .execution import DatasetExecutor
from .speech_input import OutOfRangeError
from .speech_model import Session, SpeechModel


class EvalStatistics:
  """Running letter / word edit distances and error rates (evaluation.py:27-65)."""

  def __init__(self):
    self.decodings_counter = 0
    self.sum_letter_edit_distance = self.sum_letter_error_rate = 0
    self.sum_word_edit_distance = self.sum_word_error_rate = 0
    self.letter_edit_distance = self.letter_error_rate = 0
    self.word_edit_distance = self.word_error_rate = 0

  def track_decoding(self, decoded_str, expected_str):
    expected_words = expected_str.split()
    self.letter_edit_distance = editdistance.eval(expected_str, decoded_str)
    self.letter_error_rate = self.letter_edit_distance / len(expected_str)
    self.word_edit_distance = editdistance.eval(expected_words, decoded_str.split())
    self.word_error_rate = self.word_edit_distance / len(expected_words)
    self.sum_letter_edit_distance += self.letter_edit_distance
    self.sum_letter_error_rate += self.letter_error_rate
    self.sum_word_edit_distance += self.word_edit_distance
    self.sum_word_error_rate += self.word_error_rate
    self.decodings_counter += 1

  def _mean(self, total):
    return total / self.decodings_counter

  global_letter_edit_distance = property(lambda self: self._mean(self.sum_letter_edit_distance))
  global_letter_error_rate = property(lambda self: self._mean(self.sum_letter_error_rate))
  global_word_edit_distance = property(lambda self: self._mean(self.sum_word_edit_distance))
  global_word_error_rate = property(lambda self: self._mean(self.sum_word_error_rate))


STATS_FORMAT = 'LED: {} LER: {:.2f} WED: {} WER: {:.2f}'


class Evaluation(DatasetExecutor):

  def create_sample_generator(self, limit_count: int):
    return self.reader.load_samples(self.flags.dataset, loop_infinitely=False, limit_count=limit_count,
                                    feature_type=self.flags.feature_type)

  def get_loader_limit_count(self):
    return self.flags.step_count * self.flags.batch_size

  def get_max_steps(self):
    return self.flags.step_count or None

  def run(self):
    stats = EvalStatistics()
    with Session(getattr(self.flags, 'device', 'cuda:0')) as sess:
      model = self.create_model(sess)
      print('Starting input pipeline')
      coord = self.start_pipeline(sess)
      try:
        print('Begin evaluation')
        steps = range(self.flags.step_count) if self.flags.step_count else itertools.count()
        for step in steps:
          if coord.should_stop():
            break
          self.run_step(model, sess, stats, self.flags.should_save and step == 0)
      except OutOfRangeError:
        print('Done evaluating -- step limit reached')
      finally:
        coord.request_stop()
      self.print_global_statistics(stats)
      coord.join()
    return stats

  @staticmethod
  def print_global_statistics(stats):
    print('Global statistics')
    print(STATS_FORMAT.format(stats.global_letter_edit_distance, stats.global_letter_error_rate,
                              stats.global_word_edit_distance, stats.global_word_error_rate))

  def run_step(self, model: SpeechModel, sess, stats: EvalStatistics, save: bool, verbose=True, feed_dict=None):
    global_step = model.global_step.eval()
    if save:
      avg_loss, decoded, label, summary = model.step(sess, update=False, decode=True, return_label=True,
                                                     summary=True, feed_dict=feed_dict)
      model.summary_writer.add_summary(summary, global_step)
    else:
      avg_loss, decoded, label = model.step(sess, update=False, decode=True, return_label=True, feed_dict=feed_dict)
    if verbose:
      perplexity = np.exp(float(avg_loss)) if avg_loss < 300 else float('inf')
      print('validation average loss {:.2f} perplexity {:.2f}'.format(avg_loss, perplexity))
    # Deliberate fix of a reference defect: evaluation.py:144-151 pairs labels with decodings via
    # extract_decoded_ids, which skips utterances that decode to the empty string (shifting every
    # later pairing, and raising StopIteration when the last ones are empty).  Rows are paired by
    # their batch index here; extract_decoded_ids itself is kept faithful for other callers.
    decoded_rows = [Evaluation.rows_by_batch(path) for path in decoded]
    for row, label_ids in enumerate(Evaluation.rows_by_batch(label)):
      expected_str = vocabulary.ids_to_sentence(label_ids)
      if verbose:
        print('expected: {}'.format(expected_str))
      for rows in decoded_rows:
        decoded_str = vocabulary.ids_to_sentence(rows[row])
        stats.track_decoding(decoded_str, expected_str)
        if verbose:
          print('decoded: {}'.format(decoded_str))
          print(STATS_FORMAT.format(stats.letter_edit_distance, stats.letter_error_rate,
                                    stats.word_edit_distance, stats.word_error_rate))

  @staticmethod
  def rows_by_batch(sparse_tensor):
    rows = [[] for _ in range(int(sparse_tensor.dense_shape[0]))]
    for (batch_id, _), value in zip(sparse_tensor.indices, sparse_tensor.values):
      rows[int(batch_id)].append(value)
    return rows

  @staticmethod
  def extract_decoded_ids(sparse_tensor):
    """Groups sparse values by batch row.  Faithful to evaluation.py:160-171, including its quirk:
    a row is only emitted when a LATER row starts, so an utterance that decodes to the empty string
    yields nothing and shifts the pairing of every following utterance."""
    ids = []
    last_batch_id = 0
    for (batch_id, _), value in zip(sparse_tensor.indices, sparse_tensor.values):
      if batch_id > last_batch_id:
        yield ids
        ids = []
        last_batch_id = batch_id
      ids.append(value)
    yield ids
