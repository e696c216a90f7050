"""Independent CPU formulation of the training step with torch CPU ops (checker only, like oracle/).

``F.conv1d`` (oneDNN / native im2col GEMM) + ``F.ctc_loss`` + autograd share no code with
``oracle/w2l_oracle.py`` (numpy im2col + hand-written alpha/beta) nor with the HIP kernels, so agreement of
all three at BASELINE's full sizes is the strongest pin available while TF1 cannot be installed.
Used by the full-size GPU parity tests and by ``bench.py``'s second ``cpu_baseline`` leg.
Reference lines restated: speech_model.py:74-75 (ctc_loss + reduce_mean), :155,173,177 (conv1d SAME + bias
+ relu), :275-295 (layer table), :78 (compute_gradients).
"""
import numpy as np
import torch
import torch.nn.functional as F


def same_padding(t_in, width, stride):
  t_out = -(-t_in // stride)
  total = max((t_out - 1) * stride + width - t_in, 0)
  return t_out, total // 2, total - total // 2


def make_params(params, dtype, requires_grad=True):
  """[(filters [W,Cin,Cout], bias [Cout])] numpy -> torch leaf tensors in conv1d's [Cout,Cin,W] layout."""
  out = []
  for Fw, b in params:
    w = torch.tensor(np.ascontiguousarray(np.transpose(Fw, (2, 1, 0))), dtype=dtype, requires_grad=requires_grad)
    bb = torch.tensor(np.asarray(b), dtype=dtype, requires_grad=requires_grad)
    out.append((w, bb))
  return out


def forward(x, tparams, layers, keep=None, relu_masks=None):
  """x [B,T,Cin] torch -> logits [B,T',C] (channels-last like the reference).  ``keep``: a list that receives
  every layer's output [B,T_i,C_i] (detached numpy).  ``relu_masks``: one boolean [B,T_i,C_i] array per layer (None for
  a layer without ReLU): the activation becomes ``h * mask`` instead of ``relu(h)`` -- the network with its ReLU
  pattern pinned, a smooth (piecewise-linear -> linear) function of inputs and weights."""
  h = x.permute(0, 2, 1)
  for i, ((w, b), (W, s, cin, cout, relu)) in enumerate(zip(tparams, layers)):
    _, pl, pr = same_padding(h.shape[2], W, s)
    h = F.conv1d(F.pad(h, (pl, pr)), w, b, stride=s)
    if relu and relu_masks is not None:
      h = h * torch.as_tensor(np.asarray(relu_masks[i]), dtype=h.dtype).permute(0, 2, 1)
    elif relu:
      h = torch.relu(h)
    if keep is not None:
      keep.append(h.detach().permute(0, 2, 1).numpy())
  return h.permute(0, 2, 1)


def backward_from_acts(acts, params, layers, dlogits, dtype=torch.float64):
  """Back-prop evaluated on GIVEN activations: acts[0] = the input batch, acts[i+1] = output of layer i (all
  [B,T_i,C_i]), dlogits [B,T',C] = d avg_loss / d logits.  ReLU masks are (acts[i+1] > 0), the filter gradients
  use acts[i] -- so feeding the device's own stored activations makes this the exact linear map the device's
  backward kernels have to reproduce, free of the one discontinuity of the step (a pre-activation within rounding
  of zero takes different sides in fp32 and float64 and changes every gradient below it by far more than any
  rounding error).  torch.nn.grad.conv1d_input / conv1d_weight in float64.
  Returns [(dF [W,Cin,Cout], db [Cout])] and the list of dz_i [B,T_i+1,C_i+1] (gradient wrt layer i's output)."""
  from torch.nn import grad as G
  tparams = make_params(params, dtype, requires_grad=False)
  dy = torch.as_tensor(np.asarray(dlogits), dtype=dtype).permute(0, 2, 1)          # [B, C, T']
  grads, dzs = [None] * len(layers), [None] * len(layers)
  for i in reversed(range(len(layers))):
    (w, b), (W, s, cin, cout, relu) = tparams[i], layers[i]
    out = torch.as_tensor(np.asarray(acts[i + 1]), dtype=dtype).permute(0, 2, 1)
    dz = dy * (out > 0).to(dtype) if relu else dy
    x = torch.as_tensor(np.asarray(acts[i]), dtype=dtype).permute(0, 2, 1)
    _, pl, pr = same_padding(x.shape[2], W, s)
    xp = F.pad(x, (pl, pr))
    dw = G.conv1d_weight(xp, w.shape, dz.contiguous(), stride=s)
    grads[i] = (np.transpose(dw.numpy(), (2, 1, 0)).astype(np.float64), dz.sum(dim=(0, 2)).numpy().astype(np.float64))
    dzs[i] = dz.permute(0, 2, 1).numpy()
    if i > 0:
      dxp = G.conv1d_input(xp.shape, w, dz.contiguous(), stride=s)
      dy = dxp[:, :, pl:pl + x.shape[2]]
  return grads, dzs


def loss_and_grads(x, seq_lens, labels, params, layers, dtype=torch.float64, threads=None, relu_masks=None):
  """One forward + CTC + backward.  Returns dict(logits [T',B,C], loss [B], avg_loss,
  grads [(dF [W,Cin,Cout], db [Cout])], acts (input + every layer output, [B,T_i,C_i]), dlogits [B,T',C]) as
  numpy -- gradients of avg_loss = mean_b loss_b.

  ``relu_masks`` (see ``forward``) pins the ReLU pattern, e.g. to the one the device took (``relu_masks_of``): the step's
  only discontinuity -- a pre-activation within fp32 rounding of zero that lands on the other side in float64 and
  switches a unit's whole gradient on or off -- is then out of the comparison, and what remains is a fully independent
  float64 evaluation (own forward activations, own CTC, autograd) of the same piecewise-linear branch."""
  if threads:
    torch.set_num_threads(threads)
  tparams = make_params(params, dtype)
  xt = torch.tensor(np.asarray(x), dtype=dtype)
  acts = []
  logits = forward(xt, tparams, layers, keep=acts, relu_masks=relu_masks)            # [B, T', C]
  logits.retain_grad()
  tm = logits.permute(1, 0, 2)
  lens = torch.as_tensor(np.asarray(seq_lens) // 2, dtype=torch.long)
  flat = torch.tensor([v for l in labels for v in l], dtype=torch.long)
  llen = torch.tensor([len(l) for l in labels], dtype=torch.long)
  per_utt = F.ctc_loss(torch.log_softmax(tm, dim=-1), flat, lens, llen, blank=tm.shape[2] - 1, reduction='none',
                       zero_infinity=False)
  avg = per_utt.mean()
  avg.backward()
  grads = [(np.transpose(w.grad.numpy(), (2, 1, 0)).astype(np.float64), b.grad.numpy().astype(np.float64))
           for w, b in tparams]
  return dict(logits=tm.detach().numpy().astype(np.float64), loss=per_utt.detach().numpy().astype(np.float64),
              avg_loss=float(avg.detach()), grads=grads, acts=[np.asarray(x)] + acts,
              dlogits=logits.grad.numpy().astype(np.float64))


def relu_masks_of(acts, layers):
  """The ReLU pattern of a forward pass: acts[i + 1] = stored output of layer i ([B,T_i,C_i]) -> one boolean mask per
  layer (None where the layer has no ReLU), in the form ``forward(relu_masks=...)`` takes."""
  return [(np.asarray(acts[i + 1]) > 0) if layers[i][4] else None for i in range(len(layers))]


class TorchCpuTrainer:
  """The same step with clip_by_global_norm(5) + TF-Adam (eps outside the bias correction,
  speech_model.py:77-82) for the CPU baseline timing in bench.py: fp32, oneDNN convolutions, all host cores."""

  def __init__(self, params, layers, lr=1e-4, dtype=torch.float32):
    self.layers, self.lr, self.dtype = layers, lr, dtype
    self.p = [t for pair in make_params(params, dtype) for t in pair]
    self.m = [torch.zeros_like(t) for t in self.p]
    self.v = [torch.zeros_like(t) for t in self.p]
    self.t = 0

  def step(self, x, seq_lens, labels):
    for t in self.p:
      t.grad = None
    pairs = list(zip(self.p[0::2], self.p[1::2]))
    logits = forward(torch.as_tensor(np.asarray(x), dtype=self.dtype), pairs, self.layers).permute(1, 0, 2)
    lens = torch.as_tensor(np.asarray(seq_lens) // 2, dtype=torch.long)
    flat = torch.tensor([v for l in labels for v in l], dtype=torch.long)
    llen = torch.tensor([len(l) for l in labels], dtype=torch.long)
    loss = F.ctc_loss(torch.log_softmax(logits, dim=-1), flat, lens, llen, blank=logits.shape[2] - 1,
                      reduction='none').mean()
    loss.backward()
    with torch.no_grad():
      gn = torch.sqrt(sum((t.grad.double() ** 2).sum() for t in self.p)).item()
      scale = 5.0 / max(gn, 5.0)
      self.t += 1
      lr_t = self.lr * (1.0 - 0.999 ** self.t) ** 0.5 / (1.0 - 0.9 ** self.t)
      for p, m, v in zip(self.p, self.m, self.v):
        g = p.grad * scale
        m.mul_(0.9).add_(g, alpha=0.1)
        v.mul_(0.999).addcmul_(g, g, value=0.001)
        p.sub_(lr_t * m / (v.sqrt() + 1e-3))
    return float(loss.detach())
