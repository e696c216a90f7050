"""TEST INFRASTRUCTURE: a stand-in for `speecht_amd.engine.Wav2LetterEngine` that computes with the float64 numpy oracle
(oracle/w2l_oracle.py) on host tensors, so that the CONTROL FLOW around the engine -- `SpeechModel.step`, the data-parallel
exchange (`data_parallel.GradientAllReducer` over gloo), `Training.run`, `speecht-cli train` -- can be exercised without a GPU.
Never imported by the product: tests patch `SpeechModel._ensure_engine` with it (the product's own engine raises without a GPU).

Only the attributes `SpeechModel`, `_Saver` and `GradientAllReducer` touch exist here.  The flat buffers are float64 torch
tensors laid out [F0 | b0 | F1 | b1 | ...] in the reference's [W, Cin, Cout] order (no padding), plus the 16 gate slots."""
import math

import numpy as np
import torch

from oracle import w2l_oracle as O


class _Layer:
  def __init__(self, width, stride, cin, cout, relu):
    self.width, self.stride, self.cin, self.cout, self.relu = width, stride, cin, cout, relu


class OracleEngine:
  conv_mode = 'oracle-f64'

  def __init__(self, layers, device='cpu', **_):
    self.device = torch.device('cpu')
    self.layer_tuples = [tuple(l) for l in layers]
    self.layers = [_Layer(*l) for l in layers]
    self.num_classes = self.layers[-1].cout
    self.offsets, off = [], 0
    for l in self.layers:
      nf = l.width * l.cin * l.cout
      self.offsets.append((off, off + nf))
      off += nf + l.cout
    self.n_flat = off
    z = lambda: torch.zeros(self.n_flat, dtype=torch.float64)
    self.params, self.adam_m, self.adam_v = z(), z(), z()
    self.reduce_buffer = torch.zeros(self.n_flat + 16, dtype=torch.float64)
    self.grads = self.reduce_buffer[:self.n_flat]
    self.gate_slots = self.reduce_buffer[self.n_flat:self.n_flat + 2]
    self.step_count = 0
    self.defer_label_errors = False
    self.shapes_seen = []

  # ---- layout ----
  @property
  def layer_ranges(self):
    return [(fo, bo + l.cout) for (fo, bo), l in zip(self.offsets, self.layers)]

  @property
  def reduce_ranges(self):
    r = self.layer_ranges
    r[-1] = (r[-1][0], self.n_flat + 16)
    return r

  def _lists(self, flat):
    out = []
    a = flat.numpy()
    for (fo, bo), l in zip(self.offsets, self.layers):
      out.append((a[fo:bo].reshape(l.width, l.cin, l.cout).copy(), a[bo:bo + l.cout].copy()))
    return out

  def _fill(self, flat, lists):
    a = flat.numpy()
    for (fo, bo), l, (F, b) in zip(self.offsets, self.layers, lists):
      a[fo:bo] = np.asarray(F, dtype=np.float64).reshape(-1)
      a[bo:bo + l.cout] = np.asarray(b, dtype=np.float64)

  def set_weights(self, params):
    self._fill(self.params, params)

  def get_weights(self):
    return self._lists(self.params)

  def get_adam_state(self):
    return self._lists(self.adam_m), self._lists(self.adam_v)

  def set_adam_state(self, m, v, step):
    self._fill(self.adam_m, m)
    self._fill(self.adam_v, v)
    self.step_count = int(step)

  def mark_weights_changed(self):
    pass

  def init_xavier(self, seed=None):
    rng = np.random.default_rng(seed)
    self.set_weights([(rng.uniform(-1, 1, (l.width, l.cin, l.cout)) * math.sqrt(6.0 / (l.width * l.cin + l.width * l.cout)),
                       np.zeros(l.cout)) for l in self.layers])

  # ---- the path ----
  def load_batch(self, inputs, seq_lens):
    self.x = np.asarray(inputs, dtype=np.float64)
    self.seq_lens = np.asarray(seq_lens, dtype=np.int64)
    self.shapes_seen.append(self.x.shape)

  def forward(self):
    self.logits, self.acts = O.wav2letter_forward(self.x, self.get_weights(), self.layer_tuples, keep=True)

  def set_labels(self, label_list):
    self.labels = [np.asarray(l, dtype=np.int64).tolist() for l in label_list]

  def ctc_loss_grad(self, grad_scale):
    self.loss, g = O.ctc_loss_and_grad(self.logits, self.labels, self.seq_lens // 2)
    self.dlogits = g * grad_scale
    self.gate_slots[0] = 0.0
    self.gate_slots[1] = float(np.sum(self.loss) * grad_scale)

  def backward(self, on_layer_done=None, hook_layers=None):
    grads = O.wav2letter_backward(self.acts, self.get_weights(), self.layer_tuples, self.dlogits)
    self._fill(self.grads, grads)
    for i in reversed(range(len(self.layers))):
      if on_layer_done is not None and (hook_layers is None or i in hook_layers):
        on_layer_done(i)

  def apply_update(self, lr, max_grad_norm=5.0, beta1=0.9, beta2=0.999, eps=1e-3):
    if float(self.gate_slots[0]) != 0.0:
      return
    self.step_count += 1
    g = self.grads.numpy()
    gn = math.sqrt(float(np.sum(g * g)))
    g = g * (max_grad_norm / max(gn, max_grad_norm))
    m, v, p = self.adam_m.numpy(), self.adam_v.numpy(), self.params.numpy()
    m[:] = beta1 * m + (1 - beta1) * g
    v[:] = beta2 * v + (1 - beta2) * g * g
    lr_t = lr * math.sqrt(1 - beta2 ** self.step_count) / (1 - beta1 ** self.step_count)
    p -= lr_t * m / (np.sqrt(v) + eps)

  def fetch_losses_begin(self, stream=None):
    return (float(self.gate_slots[0]), float(self.gate_slots[1]))

  def fetch_losses_end(self, handle, precise=False):
    self.mean_loss_reduced = handle[1]
    if handle[0] != 0.0:
      raise ValueError('batch rejected')
    return np.asarray(self.loss, dtype=np.float64 if precise else np.float32)

  def fetch_losses(self, precise=False):
    return self.fetch_losses_end(self.fetch_losses_begin(), precise)

  def greedy_decode(self, merge_repeated=True):
    return O.ctc_greedy_decode(self.logits, self.seq_lens // 2, merge_repeated)
