"""Shared plumbing of the dataset-driven commands (`train`, `evaluate`).

Plays the role of speecht/execution.py: owns the corpus reader, peeks at one cached sample to learn
the feature width, wires an ``InputBatchLoader`` to a sample generator and builds/restores the model.
Sub-classes say where samples come from (``create_sample_generator``), how many of them to use
(``get_loader_limit_count``) and, optionally, how many batches to produce (``get_max_steps``).
"""
import abc
import functools

from . import preprocessing, speech_input, speech_model


class DatasetExecutor(abc.ABC):

  # -- hooks ---------------------------------------------------------------------------------------
  @abc.abstractmethod
  def create_sample_generator(self, limit_count: int):
    """Iterator of (features [T, C], transcript ids)."""

  @abc.abstractmethod
  def get_loader_limit_count(self) -> int:
    """How many cached samples the loader may draw from (0 = all)."""

  def get_max_steps(self):
    """Batches to produce before the queue closes (None = unbounded)."""
    return None

  # -- construction --------------------------------------------------------------------------------
  def __init__(self, flags):
    self.flags = flags
    self.reader = preprocessing.SpeechCorpusReader(flags.data_dir)
    print('Determine input size from first sample')
    self.input_size = self.determine_input_size()
    print('Initialize InputBatchLoader')
    generator_factory = functools.partial(self.create_sample_generator, self.get_loader_limit_count())
    self.speech_input = speech_input.InputBatchLoader(self.input_size, flags.batch_size, generator_factory,
                                                      self.get_max_steps())

  def determine_input_size(self):
    first_features, _ = next(self.create_sample_generator(limit_count=1))
    return first_features.shape[1]

  # -- runtime -------------------------------------------------------------------------------------
  def start_pipeline(self, sess, n_threads=1):
    """Starts the feeder threads; the returned coordinator stops and joins them."""
    coordinator = speech_input.Coordinator()
    self.speech_input.start_threads(sess=sess, coord=coordinator, n_threads=n_threads)
    return coordinator

  def create_model(self, sess):
    """Default: the evaluation behaviour -- a checkpoint must exist (FileNotFoundError otherwise)."""
    model = speech_model.create_default_model(self.flags, self.input_size, self.speech_input)
    model.restore(sess, self.flags.run_train_dir)
    return model
