"""`speecht-cli train` -- the optimisation loop around ``SpeechModel.step``.

Behaviour follows speecht/training.py:44-98: statistics are averaged over windows of
``--steps-per-checkpoint`` steps; at the end of every window the model is checkpointed as
``<train_dir>/<run_name>/speechT.ckpt-<global_step>`` and, when a decay factor is set, the learning
rate is multiplied by it if the window loss is worse than each of the previous three windows.
"""
import math
import os
import time

from . import execution, speech_input, speech_model


class _Window:
  """Running means over one checkpoint window."""

  def __init__(self, size):
    self.size = size
    self.reset()

  def reset(self):
    self.mean_step_time = 0.0
    self.mean_loss = 0.0

  def add(self, seconds, loss):
    self.mean_step_time += seconds / self.size
    self.mean_loss += loss / self.size


class Training(execution.DatasetExecutor):

  CHECKPOINT_NAME = 'speechT.ckpt'

  def create_sample_generator(self, limit_count: int):
    samples = self.reader.load_samples('train', loop_infinitely=True, limit_count=limit_count,
                                       feature_type=self.flags.feature_type, shuffle_seed=self.shuffle_seed)
    window = getattr(self.flags, 'bucket_window', 0)
    if window and not getattr(self, 'peeking', False):
      # opt-in: batches of neighbouring lengths (4 % padding instead of ~40 % on 2-15 s speech, scripts/bench_varlen_train.py);
      # under data parallelism the GLOBAL batches are bucketed, with the job's seed, so every rank cuts the same batches
      samples = speech_input.bucket_by_length(samples, self.flags.batch_size * self.world, window, seed=self.shuffle_seed)
    return samples

  def get_loader_limit_count(self) -> int:
    return self.flags.limit_training_set

  def create_model(self, sess):
    model = speech_model.create_default_model(self.flags, self.input_size, self.speech_input)
    reset_to = self.flags.learning_rate if self.flags.reset_learning_rate else None
    model.restore_or_create(sess, self.flags.run_train_dir, reset_to)
    if self.world > 1:
      # data parallel: rank 0's weights / Adam state / counters go to every rank, gradients are SUM-all-reduced in buckets
      # under back-prop, the global mean loss rides in the first bucket (speech_model.enable_data_parallel)
      model.enable_data_parallel()
    return model

  def _end_of_window(self, sess, model, window, last_loss, summary, history):
    step = model.global_step.eval()
    perplexity = math.exp(float(last_loss)) if last_loss < 300 else float('inf')
    print('global step {:d} learning rate {:.4f} step-time {:.2f} average loss {:.2f} perplexity {:.2f}'.format(
        step, model.learning_rate.eval(), window.mean_step_time, last_loss, perplexity))
    if self.rank == 0:
      model.summary_writer.add_summary(summary, step)
    decay = self.flags.learning_rate_decay_factor
    if decay > 0 and len(history) > 2 and window.mean_loss > max(history[-3:]):
      sess.run(model.learning_rate_decay_op)
    history.append(window.mean_loss)
    model.saver.save(sess, os.path.join(self.flags.run_train_dir, self.CHECKPOINT_NAME), global_step=model.global_step)
    print('Model saved')
    window.reset()

  def run(self, max_steps=None):
    """``max_steps`` (not a reference flag) bounds the loop for tests and smoke runs."""
    with speech_model.Session(getattr(self.flags, 'device', 'cuda:0')) as sess, self.quiet_unless_rank0():
      model = self.create_model(sess)
      coordinator = self.start_pipeline(sess, n_threads=2)
      window = _Window(self.flags.steps_per_checkpoint)
      history = []
      steps_done = 0
      print('Begin training')
      try:
        while not coordinator.should_stop():
          steps_done += 1
          closes_window = steps_done % window.size == 0
          began = time.time()
          fetched = model.step(sess, summary=closes_window)     # [avg_loss, None(update), summary?]
          window.add(time.time() - began, fetched[0])
          if closes_window:
            self._end_of_window(sess, model, window, fetched[0], fetched[2], history)
          if max_steps and steps_done >= max_steps:
            break
      except speech_input.OutOfRangeError:
        print('Done training -- step limit reached')
      finally:
        coordinator.request_stop()
      coordinator.join()
      return model
