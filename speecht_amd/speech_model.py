"""Wav2Letter acoustic model with the reference's Python surface (mirror of speecht/speech_model.py).

``SpeechModel`` keeps the call protocol of the TF1 graph class -- ``add_training_ops``,
``add_decoding_ops``, ``finalize``, ``init_session``, ``step`` (same fetch order), ``restore`` /
``restore_or_create``, ``global_step`` / ``learning_rate`` handles with ``.eval()``, ``saver.save`` --
but nothing is traced or compiled: ``_create_network`` records the layer list and ``step`` runs the
hand-written HIP kernels through ``engine.Wav2LetterEngine``.  ``sess`` is an opaque ``Session``
(device + stream) instead of a tf.Session.
"""
import json
import os

import numpy as np

from . import vocabulary
from .speech_input import BaseInputLoader, SparseTensorValue, sparse_to_label_lists


class Session:
  """Execution context handed to ``step`` (the reference passes a tf.Session, training.py:46)."""

  def __init__(self, device='cuda:0'):
    self.device = device

  def __enter__(self):
    return self

  def __exit__(self, *exc):
    return False

  def run(self, op, feed_dict=None):
    """Runs host-side ops such as ``learning_rate_decay_op`` (training.py:84)."""
    if isinstance(op, (list, tuple)):
      return [self.run(o) for o in op]
    return op()


class _Scalar:
  """A host-side variable with the ``.eval()`` / ``.assign()`` protocol callers use."""

  def __init__(self, value, dtype=float):
    self._dtype = dtype
    self.value = dtype(value)

  def eval(self, session=None):
    return self.value

  def assign(self, value):
    def op():
      self.value = self._dtype(value() if callable(value) else value)
      return self.value
    return op


class _SummaryWriter:
  """TensorBoard diagnostics (tf.summary.*, speech_model.py:50-51,158-170) are out of scope; scalar
  summaries are appended as JSON lines to <log_dir>/<run_name>_<run_type>/scalars.jsonl."""

  def __init__(self, directory):
    self.directory = directory

  def add_graph(self, graph=None):
    pass

  def add_summary(self, summary, global_step=None):
    if not summary:
      return
    os.makedirs(self.directory, exist_ok=True)
    with open(os.path.join(self.directory, 'scalars.jsonl'), 'a') as f:
      f.write(json.dumps(dict(summary, step=int(global_step) if global_step is not None else None)) + '\n')


class _Saver:
  """Checkpoints: ``<path>-<global_step>.npz`` + a ``checkpoint`` index file naming the latest,
  holding what tf.train.Saver(tf.global_variables()) holds (speech_model.py:122): weights, Adam
  m/v, global_step, learning_rate."""

  MAX_TO_KEEP = 5        # tf.train.Saver's default

  def __init__(self, model):
    self.model = model

  def save(self, sess, save_path, global_step=None):
    """Like tf.train.Saver.save: ``<save_path>-<step>`` + the ``checkpoint`` index, which names files RELATIVE to
    the checkpoint directory (so a moved train dir or another cwd still restores) and lists the last
    MAX_TO_KEEP checkpoints; older ones are deleted.  Files are written to a temporary name and renamed, so a
    crash never leaves a truncated checkpoint behind the index.  In data-parallel training only rank 0 writes
    (the replicas are identical).  ``model.checkpoint_format = 'tf'`` (CLI ``--tf-checkpoints``) writes TensorFlow
    bundles instead (``save_tf``)."""
    if getattr(self.model, 'checkpoint_format', 'npz') == 'tf':
      return self.save_tf(sess, save_path, global_step)
    step = global_step.eval() if hasattr(global_step, 'eval') else global_step
    path = '{}-{}'.format(save_path, step) if step is not None else save_path
    if getattr(self.model, '_rank', 0) != 0:
      return path
    eng = self.model.engine
    directory = os.path.dirname(path) or '.'
    tmp = path + '.tmp.npz'
    np.savez(tmp, params=eng.params.cpu().numpy(), adam_m=eng.adam_m.cpu().numpy(),
             adam_v=eng.adam_v.cpu().numpy(), global_step=self.model.global_step.eval(),
             learning_rate=self.model.learning_rate.eval() if hasattr(self.model, 'learning_rate') else 0.0,
             layers=np.array([(l.width, l.stride, l.cin, l.cout, int(l.relu)) for l in eng.layers]))
    os.replace(tmp, path + '.npz')
    index_path = os.path.join(directory, 'checkpoint')
    name = os.path.basename(path) + '.npz'
    kept = []
    if os.path.exists(index_path):
      try:
        kept = [n for n in json.load(open(index_path)).get('all_model_checkpoint_paths', []) if n != name]
      except ValueError:
        kept = []
    kept.append(name)
    for old in kept[:-self.MAX_TO_KEEP]:
      try:
        os.remove(os.path.join(directory, os.path.basename(old)))
      except OSError:
        pass
    kept = kept[-self.MAX_TO_KEEP:]
    with open(index_path + '.tmp', 'w') as f:
      json.dump({'model_checkpoint_path': name, 'all_model_checkpoint_paths': kept}, f)
    os.replace(index_path + '.tmp', index_path)
    return path

  # ---- the reference's own format: TensorFlow V2 bundles (tf.train.Saver, speech_model.py:122) --------------------
  def save_tf(self, sess, save_path, global_step=None):
    """``<save_path>-<step>.index`` / ``.data-00000-of-00001`` + TF's text ``checkpoint`` file under the variable names
    the reference graph's Saver looks up (``tf_checkpoint.reference_variable_names``: Adam's beta powers live under the
    ``training/`` name scope, everything else is unscoped), so that anything that reads TF checkpoints can load this
    model.  Restoring into the reference itself is the intent and follows from TF 1.x naming rules, but has not been
    tried against a TensorFlow installation (none is available here)."""
    from . import tf_checkpoint as tfc
    step = global_step.eval() if hasattr(global_step, 'eval') else global_step
    path = '{}-{}'.format(save_path, step) if step is not None else save_path
    if getattr(self.model, '_rank', 0) != 0:
      return path
    eng = self.model.engine
    weights = eng.get_weights()
    m, v = eng.get_adam_state()
    t = eng.step_count
    tensors = {'Variable': np.array(self.model.global_step.eval(), dtype=np.int32),
               'learning_rate': np.array(self.model.learning_rate.eval() if hasattr(self.model, 'learning_rate') else 0.0,
                                         dtype=np.float32),
               # Adam's non-slot accumulators are tf.Variables made inside tf.name_scope('training') (speech_model.py:72-82)
               'training/beta1_power': np.array(0.9 ** (t + 1), dtype=np.float32),
               'training/beta2_power': np.array(0.999 ** (t + 1), dtype=np.float32)}
    for i, ((F, b), (mF, mb), (vF, vb)) in enumerate(zip(weights, m, v)):
      scope = 'convolution_layer_{}'.format(i)
      tensors.update({scope + '/filters': F, scope + '/bias': b, scope + '/filters/Adam': mF, scope + '/bias/Adam': mb,
                      scope + '/filters/Adam_1': vF, scope + '/bias/Adam_1': vb})
    assert set(tensors) == tfc.reference_variable_names(len(weights)), sorted(set(tensors) ^ tfc.reference_variable_names(len(weights)))
    tfc.write_bundle(path, tensors)
    directory = os.path.dirname(path) or '.'
    state = tfc.read_checkpoint_state(directory)
    kept = [os.path.basename(p) for p in (state[1] if state else []) if os.path.basename(p) != os.path.basename(path)]
    kept.append(os.path.basename(path))
    for old in kept[:-self.MAX_TO_KEEP]:
      for suffix in ('.index', '.data-00000-of-00001'):
        try:
          os.remove(os.path.join(directory, old + suffix))
        except OSError:
          pass
    tfc.write_checkpoint_state(directory, os.path.basename(path), kept[-self.MAX_TO_KEEP:])
    return path

  def restore_tf(self, sess, prefix):
    """Restore a checkpoint written by the reference (or by ``save_tf``): filters / biases are required, Adam
    slots, global_step and learning_rate are taken when present (``export --weights``-style files have none)."""
    from . import tf_checkpoint as tfc
    eng = self.model.engine
    layers, scalars = tfc.split_variables(tfc.read_bundle(prefix))
    def per_layer(slot):
      out = []
      for i, l in enumerate(eng.layers):
        vs = layers.get(i, {})
        if ('filters', slot) not in vs or ('bias', slot) not in vs:
          return None
        out.append((vs[('filters', slot)], vs[('bias', slot)]))
      return out
    weights = per_layer(None)
    if weights is None:
      raise ValueError('checkpoint {} does not hold convolution_layer_0..{} filters and biases'.format(prefix, len(eng.layers) - 1))
    for (F, b), l in zip(weights, eng.layers):
      if F.shape != (l.width, l.cin, l.cout) or b.shape != (l.cout,):
        raise ValueError('checkpoint {} does not match the model layout'.format(prefix))
    eng.set_weights(weights)
    step = int(scalars['Variable']) if 'Variable' in scalars else int(scalars.get('global_step', 0))
    m, v = per_layer('Adam'), per_layer('Adam_1')
    if m is not None and v is not None:
      eng.set_adam_state(m, v, step)
    else:
      eng.adam_m.zero_(); eng.adam_v.zero_()
      eng.step_count = 0
    self.model.global_step.value = step
    if hasattr(self.model, 'learning_rate') and 'learning_rate' in scalars:
      self.model.learning_rate.value = float(scalars['learning_rate'])

  def restore(self, sess, path):
    import torch
    if not path.endswith('.npz'):
      return self.restore_tf(sess, path)
    eng = self.model.engine
    with np.load(path) as ck:
      if ck['params'].shape[0] != eng.n_flat:
        raise ValueError('checkpoint {} does not match the model layout'.format(path))
      dev = eng.device
      eng.params.copy_(torch.as_tensor(ck['params']).to(dev))
      eng.adam_m.copy_(torch.as_tensor(ck['adam_m']).to(dev))
      eng.adam_v.copy_(torch.as_tensor(ck['adam_v']).to(dev))
      eng.step_count = int(ck['global_step'])
      eng.mark_weights_changed()
      self.model.global_step.value = int(ck['global_step'])
      if hasattr(self.model, 'learning_rate'):
        self.model.learning_rate.value = float(ck['learning_rate'])


def latest_checkpoint(checkpoint_directory):
  index = os.path.join(checkpoint_directory, 'checkpoint')
  if not os.path.exists(index):
    return None
  try:
    path = json.load(open(index)).get('model_checkpoint_path')
  except ValueError:
    # not this package's JSON index: TensorFlow's text CheckpointState (the reference's own train directories and
    # its published weights, README.md:72-79) names a bundle prefix
    from . import tf_checkpoint as tfc
    state = tfc.read_checkpoint_state(checkpoint_directory)
    if not state:
      return None
    for prefix in (state[0], os.path.join(checkpoint_directory, os.path.basename(state[0]))):
      if os.path.exists(prefix + '.index'):
        return prefix
    return None
  if not path:
    return None
  # names in the index are relative to the checkpoint directory (indices written before that hold a path
  # relative to the cwd of the run that wrote them: its basename still resolves here)
  local = os.path.join(checkpoint_directory, os.path.basename(path))
  if os.path.exists(local):
    return local
  return path if os.path.exists(path) else None


class SpeechModel:

  def __init__(self, input_loader: BaseInputLoader, input_size: int, num_classes: int):
    """input_loader provides the batches, input_size = values per time step, num_classes =
    vocabulary size + 1 for the CTC blank (speech_model.py:29-51)."""
    self.input_loader = input_loader
    self.input_size = input_size
    self.num_classes = num_classes
    self.convolution_count = 0
    self._layer_specs = []
    self.global_step = _Scalar(0, int)
    self.inputs, self.sequence_lengths, self.labels = input_loader.get_inputs()
    self.logits = self._create_network(num_classes)
    self.engine = None
    self._training = False
    self._decoding = False
    self._reducer = None
    self._world = 1
    self._rank = 0

  # ---- graph-building protocol ---------------------------------------------------------------
  def _convolution(self, value, filter_width, stride, input_channels, out_channels, apply_non_linearity=True):
    """Registers conv1d(SAME) + bias (+ ReLU) as layer ``convolution_layer_<id>`` (speech_model.py:128-181);
    returns (symbolic output, out_channels) like the reference."""
    layer_id = self.convolution_count
    self.convolution_count += 1
    self._layer_specs.append((filter_width, stride, input_channels, out_channels, bool(apply_non_linearity)))
    return ('convolution_layer_{}'.format(layer_id), out_channels), out_channels

  def _create_network(self, num_classes):
    raise NotImplementedError()

  def add_training_ops(self, learning_rate=1e-3, learning_rate_decay_factor=0, max_gradient_norm=5.0, momentum=0.9):
    """CTC loss -> mean -> clip_by_global_norm -> Adam(epsilon=1e-3) (speech_model.py:53-82).
    ``momentum`` is accepted and unused, exactly as in the reference."""
    self.learning_rate = _Scalar(learning_rate, float)
    self.learning_rate_decay_op = self.learning_rate.assign(
        lambda: self.learning_rate.value * learning_rate_decay_factor)
    self.max_gradient_norm = max_gradient_norm
    self._training = self.labels is not None

  def add_decoding_ops(self, language_model=None, lm_weight=0.8, word_count_weight=0.0, valid_word_count_weight=2.3,
                       beam_width=0, beam_input=None):
    """Greedy CTC decoding (speech_model.py:112-115).  The LM beam search needs the reference's
    custom tensorflow-with-kenlm fork (speech_model.py:101-111) and is not part of this path.
    ``beam_width`` > 0 (not a reference argument) selects the LM-free prefix beam search instead (up to 128; the reference's
    operating point is 100 with merge_repeated=False on ``beam_input='log10_softmax'``, i.e. log10(softmax + 1e-8),
    speech_model.py:102-110 -- without its KenLM scorer)."""
    if language_model:
      raise NotImplementedError('KenLM beam-search decoding depends on a TensorFlow fork that is not vendored')
    self.beam_width = int(beam_width or 0)
    self.beam_input = beam_input
    self.lm_weight, self.word_count_weight = lm_weight, word_count_weight
    self.valid_word_count_weight = valid_word_count_weight
    self._decoding = True

  def finalize(self, log_dir: str, run_name: str, run_type: str):
    self.saver = _Saver(self)
    self.summary_writer = _SummaryWriter('{}/{}_{}'.format(log_dir, run_name, run_type))

  # ---- session protocol ------------------------------------------------------------------------
  def _ensure_engine(self, sess):
    if self.engine is None:
      from .engine import Wav2LetterEngine
      self.engine = Wav2LetterEngine(self._layer_specs, device=sess.device)
    return self.engine

  def init_session(self, sess, init_variables=True):
    eng = self._ensure_engine(sess)
    if init_variables:
      eng.init_xavier(getattr(self, 'init_seed', None))        # xavier_initializer filters, zero biases (speech_model.py:150-152)
      eng.adam_m.zero_(); eng.adam_v.zero_()
      eng.step_count = 0
      self.global_step.value = 0
    self.summary_writer.add_graph(None)

  def enable_data_parallel(self, group=None, transport=None):
    """Shard-by-utterance data parallelism: see data_parallel.py.  Call after init_session / restore.  ``transport``: None =
    the library's own RCCL communicator when the job runs on the nccl backend with a GPU per rank and the communicator spans the
    job (`data_parallel.make_reducer`, the rule bench.py follows), torch.distributed otherwise; 'rccl' / 'torch' force one."""
    import torch.distributed as dist
    from .data_parallel import make_reducer
    self._world = dist.get_world_size(group) if dist.is_initialized() else 1
    eng = self.engine
    self._rank = dist.get_rank(group) if dist.is_initialized() else 0
    self._reducer, self._transport_note = (make_reducer(eng.reduce_buffer, eng.reduce_ranges, group, transport)
                                           if self._world > 1 else (None, None))
    # a label the host refuses must not raise on one rank while the others wait in the all-reduce: it becomes a status
    # word that travels with the gradients, and every rank raises together after the step (engine.set_labels)
    eng.defer_label_errors = self._world > 1
    if self._world > 1:
      # Only gradients are exchanged afterwards, so the replicas must START identical: rank 0's weights, Adam
      # moments, step counters and learning rate go to everyone (init_session draws an unseeded Xavier sample
      # per process, and a restore may have read different files).
      src = dist.get_global_rank(group, 0) if group is not None else 0
      for t in (eng.params, eng.adam_m, eng.adam_v):
        dist.broadcast(t, src=src, group=group)
      box = [(eng.step_count, self.global_step.value, self.learning_rate.value if hasattr(self, 'learning_rate') else None)]
      dist.broadcast_object_list(box, src=src, group=group)
      eng.step_count, self.global_step.value = int(box[0][0]), int(box[0][1])
      if box[0][2] is not None and hasattr(self, 'learning_rate'):
        self.learning_rate.value = float(box[0][2])
      eng.mark_weights_changed()
      # the read-back of a step's status words / gate / mean loss is enqueued behind the FIRST gradient bucket's all-reduce (the
      # gate and the mean-loss slot travel in it) on the collective's own stream -- never on a stream that shares a hardware
      # queue with the compute stream, where the wait would hold back-prop back: `step` returns while back-prop is still running
      self._early = None

      def after_first_bucket(wait):
        stream = self._reducer.readback_stream()
        wait(stream)
        self._early = eng.fetch_losses_begin(stream=stream)
      self._reducer.after_first_bucket = after_first_bucket

  def step(self, sess, loss=True, update=True, decode=False, return_label=False, summary=False, feed_dict=None):
    """One evaluation of the path.  Returns, in this order and only when requested:
    avg_loss, decoded, label, update (None), summary  (speech_model.py:197-235)."""
    eng = self._ensure_engine(sess)
    inputs, seq_lens, labels = self.input_loader.dequeue()
    if feed_dict:
      inputs = feed_dict.get(self.inputs, inputs)
      seq_lens = feed_dict.get(self.sequence_lengths, seq_lens)
      labels = feed_dict.get(self.labels, labels) if self.labels is not None else labels
    if (loss or update) and labels is None:
      raise ValueError('loss/update requested but the input loader provides no labels')
    eng.load_batch(inputs, seq_lens)
    eng.forward()
    out = []
    avg_loss = None
    if loss or update:
      eng.set_labels(sparse_to_label_lists(labels))
      # d(avg_loss)/d(loss_b) = 1 / global batch (speech_model.py:75)
      eng.ctc_loss_grad(1.0 / (len(seq_lens) * self._world))
      # single process: the loss read-back is enqueued right behind CTC and waited for after back-prop and the update have been
      # enqueued -- step() returns while the GPU is still in the backward pass, and the caller's next step() overlaps its host
      # side (dequeue, batch hand-over, first launches) with it.  Data parallel: the update gate and the global mean loss ride
      # in the FIRST gradient bucket (engine.gate_slots); the read-back is enqueued behind that bucket's all-reduce on a stream of
      # its own (`enable_data_parallel`), so the step has the same single, early host wait and no collective of its own.
      early = eng.fetch_losses_begin() if (self._world == 1 or not update) else None
      self._early = None
      if update:
        if not self._training:
          raise RuntimeError('add_training_ops() was not called with labelled inputs')
        eng.backward(self._reducer.on_layer_done if self._reducer else None, self._reducer.hook_layers if self._reducer else None)
        if self._reducer:
          self._reducer.finish()
        eng.apply_update(self.learning_rate.value, self.max_gradient_norm)   # no-op on the device if CTC rejected the batch
      # raises on a CTC status word (of any rank) -- before global_step moves: like TF's failed sess.run, a rejected
      # batch leaves weights, Adam state and counters as they were
      if early is None:
        early = self._early if self._early is not None else eng.fetch_losses_begin()
      losses = eng.fetch_losses_end(early, precise=True)
      avg_loss = np.float32(losses.mean())     # mean of -log p in float64, returned as TF's float32
      if update:
        self.global_step.value += 1
      if self._world > 1:
        if update:
          avg_loss = np.float32(eng.mean_loss_reduced)         # the global mean, out of the gradient exchange itself
        else:
          from .data_parallel import all_reduce_mean_scalar     # evaluation steps exchange no gradients: one scalar all-reduce
          avg_loss = np.float32(all_reduce_mean_scalar(float(avg_loss), eng.device))
    if loss:
      out.append(avg_loss)
    if decode:
      if not self._decoding:
        raise RuntimeError('add_decoding_ops() was not called')
      ids, _ = eng.beam_search_decode(self.beam_width, getattr(self, 'beam_input', None)) if self.beam_width else eng.greedy_decode()
      idx = [[b, p] for b, seq in enumerate(ids) for p in range(len(seq))]
      out.append([SparseTensorValue(np.array(idx, dtype=np.int64).reshape(-1, 2),
                                    np.array([v for seq in ids for v in seq], dtype=np.int64),
                                    np.array([len(ids), max([len(s) for s in ids] + [0])], dtype=np.int64))])
    if return_label:
      out.append(labels)
    if update:
      out.append(None)
    if summary:
      scalars = {}
      if avg_loss is not None:
        scalars['loss'] = float(avg_loss)
      if hasattr(self, 'learning_rate'):
        scalars['learning_rate'] = self.learning_rate.value
      out.append(scalars)
    return out

  def restore(self, session, checkpoint_directory: str, reset_learning_rate: float = None):
    path = latest_checkpoint(checkpoint_directory)
    if not path:
      raise FileNotFoundError('No checkpoint for evaluation found')
    print('Reading model parameters from {}'.format(path))
    self._ensure_engine(session)
    self.saver.restore(session, path)
    self.init_session(session, init_variables=False)
    if reset_learning_rate:
      self.learning_rate.value = float(reset_learning_rate)

  def restore_or_create(self, session, checkpoint_directory: str, reset_learning_rate: float = None):
    try:
      self.restore(session, checkpoint_directory, reset_learning_rate)
    except FileNotFoundError:
      print('Created model with fresh parameters.')
      self.init_session(session, init_variables=True)

  # ---- weight interchange: the `export --weights` layout (exporting.py:30-40) -------------------
  def export_weights(self, directory):
    """<dir>/convolution_layer_<i>/filters:0.npy [W,Cin,Cout] and .../bias:0.npy [Cout]."""
    for i, (F, b) in enumerate(self.engine.get_weights()):
      layer_dir = os.path.join(directory, 'convolution_layer_{}'.format(i))
      os.makedirs(layer_dir, exist_ok=True)
      np.save(os.path.join(layer_dir, 'filters:0.npy'), F)
      np.save(os.path.join(layer_dir, 'bias:0.npy'), b)

  def load_weights(self, session, directory):
    eng = self._ensure_engine(session)
    params = []
    for i in range(len(self._layer_specs)):
      layer_dir = os.path.join(directory, 'convolution_layer_{}'.format(i))
      params.append((np.load(os.path.join(layer_dir, 'filters:0.npy')), np.load(os.path.join(layer_dir, 'bias:0.npy'))))
    eng.set_weights(params)


class Wav2LetterModel(SpeechModel):
  """11 convolutions: 48/2 -> 7x (7/1) -> 32/1 (x8 channels) -> 1x1 -> 1x1 to classes (speech_model.py:275-295)."""

  def __init__(self, input_loader: BaseInputLoader, input_size: int, num_classes: int):
    super().__init__(input_loader, input_size, num_classes)

  def _create_network(self, num_classes):
    outputs, channels = self._convolution(self.inputs, 48, 2, self.input_size, 250)
    for _ in range(7):
      outputs, channels = self._convolution(outputs, 7, 1, channels, channels)
    outputs, channels = self._convolution(outputs, 32, 1, channels, channels * 8)
    outputs, channels = self._convolution(outputs, 1, 1, channels, channels)
    outputs, channels = self._convolution(outputs, 1, 1, channels, num_classes, False)
    # time-major [max_time / 2, batch_size, num_classes] like tf.transpose(outputs, (1, 0, 2))
    return ('logits_time_major', outputs)


def create_default_model(flags, input_size: int, speech_input: BaseInputLoader) -> SpeechModel:
  """speech_model.py:298-324: training ops are always added so that checkpoints restore fully."""
  model = Wav2LetterModel(input_loader=speech_input, input_size=input_size, num_classes=vocabulary.SIZE + 1)
  if flags.command == 'train':
    model.add_training_ops(learning_rate=flags.learning_rate,
                           learning_rate_decay_factor=flags.learning_rate_decay_factor,
                           max_gradient_norm=flags.max_gradient_norm, momentum=flags.momentum)
    model.add_decoding_ops()
  elif flags.command == 'export':
    model.add_training_ops()
    model.add_decoding_ops()
  else:
    model.add_training_ops()
    model.add_decoding_ops(language_model=getattr(flags, 'language_model', None),
                           lm_weight=getattr(flags, 'lm_weight', 0.8),
                           word_count_weight=getattr(flags, 'word_count_weight', 0.0),
                           valid_word_count_weight=getattr(flags, 'valid_word_count_weight', 2.3),
                           beam_width=getattr(flags, 'beam_width', 0), beam_input=getattr(flags, 'beam_input', None))
  model.finalize(log_dir=flags.log_dir, run_name=flags.run_name, run_type=flags.run_type)
  model.checkpoint_format = 'tf' if getattr(flags, 'tf_checkpoints', False) else 'npz'
  model.init_seed = getattr(flags, 'seed', None)       # (extension; None = unseeded like the reference)
  return model
