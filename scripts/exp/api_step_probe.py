import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
os.environ['ST_CONV_MODE'] = 'bf16'
from speecht_amd import speech_input, speech_model
import scripts.bench_api_train as B
rng = np.random.default_rng(0)
pool = [(rng.standard_normal((1001, 80)).astype(np.float32), rng.integers(0, 28, 150).tolist()) for _ in range(64)]
def generator():
  while True:
    for s in pool: yield s
for threads in (2, 1):
  loader = speech_input.InputBatchLoader(80, 32, generator)
  model = speech_model.create_default_model(B.Flags(), 80, loader)
  with speech_model.Session('cuda:0') as sess:
    model.init_session(sess)
    coord = speech_input.Coordinator()
    loader.start_threads(sess=sess, coord=coord, n_threads=threads)
    eng = None
    for _ in range(8): model.step(sess)
    torch.cuda.synchronize()
    eng = model.engine
    # instrument
    acc = {}
    def wrap(obj, name):
      fn = getattr(obj, name)
      def timed(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return r
      setattr(obj, name, timed)
    for n in ('load_batch', 'forward', 'set_labels', 'ctc_loss_grad', 'backward', 'apply_update', 'fetch_losses_begin', 'fetch_losses_end'):
      wrap(eng, n)
    wrap(loader, 'dequeue')
    N = 60
    t0 = time.perf_counter()
    for _ in range(N): model.step(sess)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    coord.request_stop()
  print(json.dumps(dict(threads=threads, ms_per_step=round(dt * 1e3, 3), phases_ms={k: round(v / N * 1e3, 3) for k, v in acc.items()},
                        host_sum_ms=round(sum(acc.values()) / N * 1e3, 3))))
sys.stdout.flush(); os._exit(0)
