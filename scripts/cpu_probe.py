import sys, time, os, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from tests import torch_ref as TR, workloads as WL
layers = WL.w2l_layers(80); params = WL.xavier_params(layers, seed=42, bias_range=0.0, dtype=np.float32)
x, sl, labels = WL.make_batch([1001]*4, 80, seed=0); x=x.astype(np.float32)
print('default threads', torch.get_num_threads(), 'cpus', os.cpu_count())
for threads in (None, 64, 128):
  if threads: torch.set_num_threads(threads)
  for mk in (False, True):
    tr = TR.TorchCpuTrainer(params, layers)
    with torch.backends.mkldnn.flags(enabled=mk):
      tr.step(x[:1,:201],[201],[labels[0][:20]])
      t0=time.time(); tr.step(x, sl, labels); dt=time.time()-t0
    print('threads', torch.get_num_threads(), 'mkldnn', mk, '%.2f s for 4 utt' % dt, flush=True)
    if dt > 60: break
