"""`speecht-cli train`: the step loop with checkpoint cadence and LR decay (mirror of speecht/training.py)."""
import os
import time

import numpy as np

from .execution import DatasetExecutor
from .speech_input import OutOfRangeError
from .speech_model import Session, create_default_model


class Training(DatasetExecutor):

  def create_sample_generator(self, limit_count: int):
    return self.reader.load_samples('train', loop_infinitely=True, limit_count=limit_count,
                                    feature_type=self.flags.feature_type)

  def get_loader_limit_count(self) -> int:
    return self.flags.limit_training_set

  def create_model(self, sess):
    model = create_default_model(self.flags, self.input_size, self.speech_input)
    model.restore_or_create(sess, self.flags.run_train_dir,
                            self.flags.learning_rate if self.flags.reset_learning_rate else None)
    return model

  def run(self, max_steps=None):
    """Same bookkeeping as training.py:44-98: every ``steps_per_checkpoint`` steps print the window
    statistics, decay the learning rate if the window loss exceeds the last three, and checkpoint.
    ``max_steps`` (not in the reference) bounds the loop for tests."""
    with Session(getattr(self.flags, 'device', 'cuda:0')) as sess:
      model = self.create_model(sess)
      coord = self.start_pipeline(sess, n_threads=2)
      step_time, loss = 0.0, 0.0
      current_step = 0
      previous_losses = []
      every = self.flags.steps_per_checkpoint
      try:
        print('Begin training')
        while not coord.should_stop():
          current_step += 1
          is_checkpoint_step = current_step % every == 0
          start_time = time.time()
          step_result = model.step(sess, summary=is_checkpoint_step)
          avg_loss = step_result[0]
          step_time += (time.time() - start_time) / every
          loss += avg_loss / every
          if is_checkpoint_step:
            global_step = model.global_step.eval()
            perplexity = np.exp(float(avg_loss)) if avg_loss < 300 else float('inf')
            print('global step {:d} learning rate {:.4f} step-time {:.2f} average loss {:.2f} perplexity {:.2f}'
                  .format(global_step, model.learning_rate.eval(), step_time, avg_loss, perplexity))
            model.summary_writer.add_summary(step_result[2], global_step)
            if self.flags.learning_rate_decay_factor > 0 and len(previous_losses) > 2 and loss > max(previous_losses[-3:]):
              sess.run(model.learning_rate_decay_op)
            previous_losses.append(loss)
            checkpoint_path = os.path.join(self.flags.run_train_dir, 'speechT.ckpt')
            model.saver.save(sess, checkpoint_path, global_step=model.global_step)
            print('Model saved')
            step_time, loss = 0.0, 0.0
          if max_steps and current_step >= max_steps:
            break
      except OutOfRangeError:
        print('Done training -- step limit reached')
      finally:
        coord.request_stop()
      coord.join()
      return model
