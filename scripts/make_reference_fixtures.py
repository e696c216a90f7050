#!/usr/bin/env python3
"""Oracle falsification harness: writes tests/golden/ref_{mel,logits,ctc,adam}.npz by RUNNING THE REFERENCE.

    python scripts/make_reference_fixtures.py                 # needs /root/reference + tensorflow 1.x + librosa
    python scripts/make_reference_fixtures.py --dry-run DIR   # same files from the numpy oracle (self-test of the
                                                              # consumers; marked source='oracle-dry-run', pins nothing)

Why it exists: the arithmetic of the path lives in TensorFlow 1.x and librosa, neither
of which can be installed in the build container, and the reference's tests hold no number of this path -- so the
oracle (oracle/w2l_oracle.py) is "parity unpinned".  This script is the one-command fix for the day such an environment
exists: it imports the reference's OWN modules (/root/reference/speecht/{preprocessing,speech_input,speech_model}.py --
imported where they lie, nothing copied) on SURVEY 8(d)'s synthetic inputs and records what they compute; the tests in
tests/test_reference_fixtures.py then hold the oracle (CPU) and the HIP path (GPU) against those files and skip while
the files are absent.  It has NOT been run against TensorFlow/librosa (neither exists here); the reference-side code
below follows the call sites cited next to each block.

What is recorded (all from the reference's own functions / graph):
  ref_mel.npz     calc_power_spectrogram (preprocessing.py:36-58) at 80 and 128 mels and calc_mfccs (:61-84) on
                  clip(0.1 N(0,1)) clips of 32 000 and 16 077 samples at 16 kHz (rng 1234 + index)
  ref_logits.npz  the padded batch and sparse labels out of BaseInputLoader._get_inputs_feed_item /
                  _get_labels_feed_item (speech_input.py:27-69), and Wav2LetterModel's logits [T', B, 29]
                  (speech_model.py:275-295) for seeded weights (tests/workloads.xavier_params(seed=42), non-zero biases)
  ref_ctc.npz     tf.nn.ctc_loss per utterance + avg_loss (speech_model.py:74-75), d avg_loss / d logits, the greedy
                  decoder's ids and neg-sum-logits (:113-115)
  ref_adam.npz    the 22 gradients of avg_loss (before clipping), the global norm, and every variable after ONE
                  `update` (clip_by_global_norm(5) + Adam(eps=1e-3), :77-82) -- weights, Adam slots, beta powers, global
                  step -- plus the NAMES of tf.global_variables() (pins tf_checkpoint.reference_variable_names).
                  Full-width tensors are stored as (sum, abs-sum, 4096 strided samples); biases in full.
"""
import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')
REFERENCE = '/root/reference'

from tests import workloads as WL  # noqa: E402

N_MELS = 80
FRAMES = [121, 100, 77]                       # ragged batch: odd / even lengths, all >= the 48-tap first layer
LR = 1e-4
SAMPLES = 4096


def synthetic_clip(index, n):
  """clip(0.1 * N(0,1), -1, 1) float32, rng seeded 1234 + utterance index."""
  return np.clip(0.1 * np.random.default_rng(1234 + index).standard_normal(n), -1.0, 1.0).astype(np.float32)


def batch_case():
  layers = WL.w2l_layers(N_MELS)
  params = WL.xavier_params(layers, seed=42, dtype=np.float32)           # non-zero biases
  x, seq, labels = WL.make_batch(FRAMES, N_MELS, seed=21)
  feats = [x[i, :t].astype(np.float32) for i, t in enumerate(FRAMES)]
  labels[1] = labels[1][:6] + [labels[1][5]] * 2 + labels[1][6:12]      # repeated labels need separating blanks
  return layers, params, feats, labels


def sample(a):
  """(sum, abs-sum, SAMPLES strided elements) of a tensor: a 96 MB gradient becomes 16 KB that still pins it."""
  flat = np.asarray(a, dtype=np.float64).reshape(-1)
  idx = np.linspace(0, flat.size - 1, min(SAMPLES, flat.size)).astype(np.int64)
  return np.array([flat.sum(), np.abs(flat).sum()]), flat[idx]


# ---------------------------------------------------------------------------------------------------------------------
class ReferenceBackend:
  """The reference itself: its modules imported from /root/reference, TensorFlow 1.x graph mode, librosa."""
  source = 'reference'

  def __init__(self):
    missing = []
    try:
      import tensorflow as tf
      if not hasattr(tf, 'placeholder'):
        if hasattr(tf, 'compat') and hasattr(tf.compat, 'v1'):
          missing.append('tensorflow 1.x (found %s; the reference uses tf.contrib / tf.placeholder directly)' % tf.__version__)
    except ImportError:
      missing.append('tensorflow (1.x; requirements.txt:2)')
    try:
      import librosa  # noqa: F401
    except ImportError:
      missing.append('librosa (>= 0.5.0; requirements.txt:5)')
    if not os.path.isdir(REFERENCE):
      missing.append(REFERENCE)
    if missing:
      raise RuntimeError('cannot run the reference here, missing: ' + '; '.join(missing))
    sys.path.insert(0, REFERENCE)
    import tensorflow as tf
    import librosa
    from speecht import preprocessing, speech_input, speech_model          # the reference's modules, where they lie
    self.tf, self.pre, self.si, self.sm = tf, preprocessing, speech_input, speech_model
    self.versions = 'tensorflow %s, librosa %s' % (tf.__version__, librosa.__version__)

  def mel(self, y, n_mels):
    return self.pre.calc_power_spectrogram(y, 16000, n_mels=n_mels)        # preprocessing.py:36-58

  def mfcc(self, y):
    return self.pre.calc_mfccs(y, 16000)                                   # preprocessing.py:61-84

  def step(self, layers, params, feats, labels):
    tf, si, sm = self.tf, self.si, self.sm
    tf.reset_default_graph()

    class FeedLoader(si.BaseInputLoader):
      """placeholders fed with the reference's own feed items (speech_input.py:27-69)"""

      def __init__(self, input_size):
        super().__init__(input_size)
        self.inputs = tf.placeholder(tf.float32, [None, None, input_size], name='inputs')
        self.sequence_lengths = tf.placeholder(tf.int32, [None], name='sequence_lengths')
        self.labels = tf.sparse_placeholder(tf.int32, name='labels')

      def get_inputs(self):
        return self.inputs, self.sequence_lengths, self.labels

    loader = FeedLoader(N_MELS)
    model = sm.Wav2LetterModel(loader, N_MELS, 29)                         # speech_model.py:270-295, num_classes = 28 + 1
    model.add_training_ops(learning_rate=LR)                               # :53-82 (defaults: clip 5.0)
    model.add_decoding_ops()                                               # greedy branch, :112-115
    model.finalize(log_dir=tempfile.mkdtemp(), run_name='fixtures', run_type='train')
    x, seq, max_time = loader._get_inputs_feed_item(feats)                 # speech_input.py:27-45
    sparse = loader._get_labels_feed_item(labels, max_time)                # :47-69
    feed = {loader.inputs: x, loader.sequence_lengths: seq, loader.labels: sparse}
    by_name = {v.op.name: v for v in tf.global_variables()}
    weights = [(by_name['convolution_layer_%d/filters' % i], by_name['convolution_layer_%d/bias' % i])
               for i in range(len(layers))]
    flat_vars = [v for pair in weights for v in pair]
    grads_op = tf.gradients(model.avg_loss, flat_vars)
    dlogits_op = tf.gradients(model.avg_loss, model.logits)[0]
    out = dict(x=x, seq=seq, sparse_indices=sparse.indices, sparse_values=sparse.values, sparse_shape=sparse.dense_shape)
    with tf.Session() as sess:
      model.init_session(sess)
      for (fv, bv), (F, b) in zip(weights, params):
        sess.run([fv.assign(F), bv.assign(b)])
      logits, cost, avg, dlogits, grads, decoded, logp = sess.run(
          [model.logits, model.cost, model.avg_loss, dlogits_op, grads_op, model.decoded, model.log_probabilities], feed_dict=feed)
      out.update(logits=logits, loss=cost, avg_loss=avg, dlogits=dlogits, grads=grads,
                 decoded_indices=decoded[0].indices, decoded_values=decoded[0].values, decoded_shape=decoded[0].dense_shape,
                 neg_sum_logits=logp)
      sess.run(model.update, feed_dict=feed)                               # clip + Adam + global_step += 1, :80-82
      names = sorted(by_name)
      out['variable_names'] = names
      out['after'] = {n: sess.run(by_name[n]) for n in names}
    return out


class OracleBackend:
  """Dry run: the same files from the numpy oracle, to exercise the consumers.  Pins nothing."""
  source = 'oracle-dry-run'
  versions = 'oracle/w2l_oracle.py (float64 numpy)'

  def __init__(self):
    from oracle import w2l_oracle as O
    self.O = O

  def mel(self, y, n_mels):
    return self.O.calc_power_spectrogram(y.astype(np.float64), 16000, n_mels=n_mels)

  def mfcc(self, y):
    return self.O.calc_mfccs(y.astype(np.float64), 16000)

  def step(self, layers, params, feats, labels):
    O = self.O
    from speecht_amd.tf_checkpoint import reference_variable_names
    x, seq, max_time = O.pad_batch([f.astype(np.float64) for f in feats], N_MELS)
    idx, vals, shape = O.sparse_labels(labels, max_time)
    p64 = [(F.astype(np.float64), b.astype(np.float64)) for F, b in params]
    res = O.train_step(x, seq, labels, p64, layers, O.zero_opt_state(p64), lr=LR)
    _, dl = O.ctc_loss_and_grad(res['logits'], labels, seq // 2)
    dec, score = O.ctc_greedy_decode(res['logits'], seq // 2)
    d_idx, d_val, d_shape = O.decoded_to_sparse(dec)
    after = {'Variable': np.array(1, np.int32), 'learning_rate': np.array(LR, np.float32),
             'training/beta1_power': np.array(0.9 ** 2, np.float32), 'training/beta2_power': np.array(0.999 ** 2, np.float32)}
    for i, ((F, b), (mF, mb), (vF, vb)) in enumerate(zip(res['params'], res['opt_state']['m'], res['opt_state']['v'])):
      base = 'convolution_layer_%d/' % i
      after.update({base + 'filters': F, base + 'bias': b, base + 'filters/Adam': mF, base + 'bias/Adam': mb,
                    base + 'filters/Adam_1': vF, base + 'bias/Adam_1': vb})
    assert set(after) == reference_variable_names(len(layers))
    return dict(x=x, seq=seq, sparse_indices=idx, sparse_values=vals, sparse_shape=shape, logits=res['logits'], loss=res['loss'],
                avg_loss=res['avg_loss'], dlogits=dl / len(labels),          # time-major [T', B, C] like tf.gradients(avg_loss, logits)
                grads=[g for pair in res['grads'] for g in pair], decoded_indices=d_idx, decoded_values=d_val,
                decoded_shape=d_shape, neg_sum_logits=score, variable_names=sorted(after), after=after)


# ---------------------------------------------------------------------------------------------------------------------
def write_fixtures(backend, out_dir):
  os.makedirs(out_dir, exist_ok=True)
  meta = dict(source=backend.source, versions=backend.versions, generator='scripts/make_reference_fixtures.py')

  mel = dict(meta)
  for k, n in enumerate((32000, 16077)):
    y = synthetic_clip(k, n)
    mel['samples_%d' % k] = np.int64(n)
    mel['mel80_%d' % k] = np.asarray(backend.mel(y, 80), dtype=np.float64)
    mel['mel128_%d' % k] = np.asarray(backend.mel(y, 128), dtype=np.float64)
    mel['mfcc_%d' % k] = np.asarray(backend.mfcc(y), dtype=np.float64)
  np.savez_compressed(os.path.join(out_dir, 'ref_mel.npz'), **mel)

  layers, params, feats, labels = batch_case()
  r = backend.step(layers, params, feats, labels)
  case = dict(frames=np.array(FRAMES), n_mels=np.int64(N_MELS), weights_seed=np.int64(42), batch_seed=np.int64(21), lr=np.float64(LR),
              label_lengths=np.array([len(l) for l in labels]), label_values=np.array([v for l in labels for v in l], dtype=np.int64))
  np.savez_compressed(os.path.join(out_dir, 'ref_logits.npz'), x=np.asarray(r['x']), seq=np.asarray(r['seq']),
                      sparse_indices=np.asarray(r['sparse_indices']), sparse_values=np.asarray(r['sparse_values']),
                      sparse_shape=np.asarray(r['sparse_shape']), logits=np.asarray(r['logits']), **case, **meta)
  np.savez_compressed(os.path.join(out_dir, 'ref_ctc.npz'), loss=np.asarray(r['loss']), avg_loss=np.float64(r['avg_loss']),
                      dlogits=np.asarray(r['dlogits']), decoded_indices=np.asarray(r['decoded_indices']),
                      decoded_values=np.asarray(r['decoded_values']), decoded_shape=np.asarray(r['decoded_shape']),
                      neg_sum_logits=np.asarray(r['neg_sum_logits']), **meta)
  adam = dict(meta)
  adam['variable_names'] = np.array(r['variable_names'])
  gn2 = 0.0
  for k, g in enumerate(r['grads']):
    layer, kind = k // 2, ('filters', 'bias')[k % 2]
    gn2 += float(np.sum(np.asarray(g, np.float64) ** 2))
    if kind == 'bias':
      adam['grad_%d_bias' % layer] = np.asarray(g)
    else:
      adam['grad_%d_filters_stats' % layer], adam['grad_%d_filters_samples' % layer] = sample(g)
  adam['grad_global_norm'] = np.float64(np.sqrt(gn2))
  for name, value in r['after'].items():
    key = 'after__' + name.replace('/', '__')
    if np.asarray(value).size > 4 * SAMPLES:
      adam[key + '__stats'], adam[key + '__samples'] = sample(value)
    else:
      adam[key] = np.asarray(value)
  np.savez_compressed(os.path.join(out_dir, 'ref_adam.npz'), **adam)
  print('wrote ref_mel / ref_logits / ref_ctc / ref_adam .npz to %s (source: %s; %s)' % (out_dir, backend.source, backend.versions))


def main():
  ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
  ap.add_argument('--dry-run', metavar='DIR', help='write the files from the numpy oracle into DIR (consumer self-test)')
  args = ap.parse_args()
  if args.dry_run:
    write_fixtures(OracleBackend(), args.dry_run)
    return 0
  try:
    backend = ReferenceBackend()
  except RuntimeError as e:
    print('make_reference_fixtures: %s\nnothing written; parity stays unpinned.' % e, file=sys.stderr)
    return 2
  write_fixtures(backend, GOLD)
  return 0


if __name__ == '__main__':
  sys.exit(main())
