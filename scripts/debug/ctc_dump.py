"""Debug aid: run the CTC kernels on a tiny case and dump the lattice records next to the oracle's alpha."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from speecht_amd.engine import Wav2LetterEngine
from oracle import w2l_oracle as O

T, C = 8, 29
labels = [[], [1], [2, 2]]
B = len(labels)
rng = np.random.default_rng(0)
logits = rng.normal(size=(T, B, C)).astype(np.float32)
lens = [T, T - 1, T]
eng = Wav2LetterEngine([(1, 1, 16, C, False)], device='cuda:0')
eng.load_batch(np.zeros((B, T, 16)), [T] * B)
eng.X[-1].interior().copy_(torch.as_tensor(np.transpose(logits, (1, 0, 2))))
eng.ctc_lens = torch.as_tensor(np.asarray(lens, dtype=np.int32)).to('cuda:0')
eng.set_labels(labels)
eng.ctc_loss_grad(1.0)
torch.cuda.synchronize()
print('status', eng.ctc_status.cpu().numpy(), 'loss', eng.loss.cpu().numpy())
ref_loss, ref_grad = O.ctc_loss_and_grad(logits, labels, lens)
print('oracle loss', ref_loss)
ws = eng.ctc_ws.cpu().numpy()
rows = B * T
logy = ws[:rows * 32].reshape(B, T, 32)
emis = ws[rows * 32: rows * 96].reshape(B, T, 32, 2)
print('logy[0,0,:4]', logy[0, 0, :4], 'emis m', emis[0, 0, :4, 0], 'emis e', emis[0, 0, :4, 1].view(np.int32))
print('check 2^logy', 2.0 ** logy[0, 0, :4], emis[0, 0, :4, 0] * 2.0 ** emis[0, 0, :4, 1].view(np.int32))
alpha = ws[rows * 96: rows * 96 + rows * 64 * 2].reshape(B, T, 64, 2)
for b in range(B):
  for t in (0, 1, lens[b] - 1):
    m, e = alpha[b, t, :5, 0], alpha[b, t, :5, 1].view(np.int32)
    print('b', b, 't', t, 'm', m, 'e', e)
