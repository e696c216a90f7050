"""Diagnostic: where does the full-size backward pass lose accuracy?  Runs the config-2 step on the device and
compares, for two utterances, dlogits and every activation gradient dZ[i] with the float64 oracle (back-prop to the
inputs is independent per utterance), then every filter gradient computed from the DEVICE's own dZ / X in float64."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import w2l_oracle as O
from tests import workloads as WL
from speecht_amd.engine import Wav2LetterEngine

mode = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
layers = WL.w2l_layers(80)
params = WL.xavier_params(layers, seed=42, dtype=np.float32)
frames = [1001] * 30 + [777, 500]
x, seq, labels = WL.make_batch(frames, 80, seed=3)
x = x.astype(np.float32)
B = len(frames)
eng = Wav2LetterEngine(layers, device='cuda:0', conv_mode=mode)
eng.set_weights(params)
eng.load_batch(x, seq)
eng.set_labels(labels)
eng.forward()
eng.ctc_loss_grad(1.0 / B)
eng.backward()
torch.cuda.synchronize()
rows = [0, 31]
p64 = [(F.astype(np.float64), b.astype(np.float64)) for F, b in params]
logits, acts = O.wav2letter_forward(x[rows].astype(np.float64), p64, layers, keep=True)
loss, g = O.ctc_loss_and_grad(logits, [labels[r] for r in rows], seq[rows] // 2)
dev_logits = eng.logits_time_major().cpu().numpy()[:, rows]
print('logits err', np.abs(dev_logits - logits).max())
# CTC gradient on the device's own logits (isolates the CTC kernel from the forward error)
_, g_dev_logits = O.ctc_loss_and_grad(dev_logits.astype(np.float64), [labels[r] for r in rows], seq[rows] // 2)
dl_dev = eng.dZ[-1].interior().cpu().numpy()[rows].astype(np.float64)          # [2, T', 29]
ref_dl = np.transpose(g, (1, 0, 2)) / B
print('dlogits: err vs oracle %.2e, vs oracle-on-device-logits %.2e (of max %.3e)' % (
    np.abs(dl_dev - ref_dl).max() / np.abs(ref_dl).max(),
    np.abs(dl_dev - np.transpose(g_dev_logits, (1, 0, 2)) / B).max() / np.abs(ref_dl).max(), np.abs(ref_dl).max()))
dy = ref_dl
for i in reversed(range(len(layers))):
  (F, b), (W, s, cin, cout, relu) = p64[i], layers[i]
  dz = dy * (acts[i + 1] > 0) if relu else dy
  got = eng.dZ[i].interior().cpu().numpy()[rows].astype(np.float64)
  print('dZ[%d] err %.2e of max (max %.3e, rms %.3e)' % (i, np.abs(got - dz).max() / np.abs(dz).max(), np.abs(dz).max(),
                                                        np.sqrt((dz ** 2).mean())))
  if i > 0:
    dy, _, _ = O.conv1d_same_bwd(acts[i], F, acts[i + 1], dy, s, relu, need_dx=True)
# filter gradients from the device's own operands, float64 on the host, one layer at a time
grads = eng.get_grads()
for i in reversed(range(len(layers))):
  (W, s, cin, cout, relu) = layers[i]
  X = eng.X[i].interior().cpu().numpy().astype(np.float64)
  dZ = eng.dZ[i].interior().cpu().numpy().astype(np.float64)
  cols, t_out, pl = O._im2col(X, W, s)
  dF = (cols.reshape(-1, W * cin).T @ dZ.reshape(-1, cout)).reshape(W, cin, cout)
  print('wgrad L%d from device operands: err %.2e of max; bias %.2e' % (
      i, np.abs(grads[i][0] - dF).max() / np.abs(dF).max(),
      np.abs(grads[i][1] - dZ.reshape(-1, cout).sum(0)).max() / np.abs(dZ.reshape(-1, cout).sum(0)).max()))
