"""Shared plumbing of the dataset-driven commands (`train`, `evaluate`).

Plays the role of speecht/execution.py: owns the corpus reader, peeks at one cached sample to learn
the feature width, wires an ``InputBatchLoader`` to a sample generator and builds/restores the model.
Sub-classes say where samples come from (``create_sample_generator``), how many of them to use
(``get_loader_limit_count``) and, optionally, how many batches to produce (``get_max_steps``).
"""
import abc
import contextlib
import functools
import io

from . import data_parallel, preprocessing, speech_input, speech_model


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
    # under a launcher (`torchrun --nproc-per-node N speecht-cli ...`) this process is one rank of a data-parallel job: it
    # takes its slice of every global batch (`--batch-size` utterances per GPU); rank 0 alone prints and writes
    self.rank, self.world = data_parallel.job()
    # every rank must walk the samples in the same order: one seed for the shuffles, drawn by rank 0
    self.shuffle_seed = getattr(flags, 'seed', None)
    if self.shuffle_seed is None and self.world > 1:
      self.shuffle_seed = data_parallel.broadcast_seed()
    self.reader = preprocessing.SpeechCorpusReader(flags.data_dir)
    self.say('Determine input size from first sample')
    self.input_size = self.determine_input_size()
    self.say('Initialize InputBatchLoader')
    generator_factory = functools.partial(self.create_sample_generator, self.get_loader_limit_count())
    self.speech_input = speech_input.InputBatchLoader(self.input_size, flags.batch_size, generator_factory,
                                                      self.get_max_steps(), shard=(self.rank, self.world))

  def say(self, *args):
    if self.rank == 0:
      print(*args)

  def quiet_unless_rank0(self):
    """Context: stdout of ranks > 0 goes nowhere (the model and the loop print what the reference prints -- once per job)."""
    return contextlib.redirect_stdout(io.StringIO()) if self.rank != 0 else contextlib.nullcontext()

  def determine_input_size(self):
    self.peeking = True            # (sub-classes that wrap the stream -- a bucketing sampler -- leave the one-sample peek alone)
    try:
      first_features, _ = next(self.create_sample_generator(limit_count=1))
    finally:
      self.peeking = False
    return first_features.shape[1]

  # -- runtime -------------------------------------------------------------------------------------
  def start_pipeline(self, sess, n_threads=1):
    """Starts the feeder threads; the returned coordinator stops and joins them."""
    coordinator = speech_input.Coordinator()
    if self.world > 1 or self.shuffle_seed is not None:
      # feeder threads each walk a generator of their own and race for the queue: ONE keeps the batch order the same on every
      # rank of a job -- and reproducible under --seed (two seeded threads would also feed every batch twice)
      n_threads = 1
    self.speech_input.start_threads(sess=sess, coord=coordinator, n_threads=n_threads)
    return coordinator

  def create_model(self, sess):
    """Default: the evaluation behaviour -- a checkpoint must exist (FileNotFoundError otherwise)."""
    model = speech_model.create_default_model(self.flags, self.input_size, self.speech_input)
    model.restore(sess, self.flags.run_train_dir)
    return model
