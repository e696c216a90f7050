#!/usr/bin/env python3
"""bench.py -- utterances/sec of one Wav2Letter TRAINING step on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): per GPU a batch of 32 synthetic 10 s @ 16 kHz
utterances = 1001 frames x 80 mel features (z-normalised synthetic features; the reference
extracts features offline, preprocessing.py:212-241), 150-character labels, default Wav2Letter
depth (speech_model.py:275-295), fp32.  One step = batch -> 11-layer conv forward -> CTC loss +
gradient -> back-prop -> (gradient all-reduce) -> global-norm clip -> TF-Adam, inputs resident in
HBM.  Weak scaling: every rank processes its own 32 utterances; value = N*32 / max-over-ranks time.

Prints ONE JSON line on rank 0 with the driver's contract plus `roofline` (dominant kernel, HIP
event timing) and `cpu_baseline` (the numpy oracle timed on the host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from speecht_amd.data_parallel import GradientAllReducer  # noqa: E402
from speecht_amd.engine import Wav2LetterEngine  # noqa: E402
from tests import workloads as WL  # noqa: E402

PEAK_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32 MFMA == f32 vector peak
PEAK_HBM_GBS = 8000.0


def conv_flops(engine, batch):
  """Algorithmic FLOPs per launch of layer i forward: 2*B*T_out*W*Cin*Cout (unpadded dims)."""
  out = []
  for l, (t_in, t_out, pl, pr) in zip(engine.layers, engine.geo):
    out.append(2.0 * batch * t_out * l.width * l.cin * l.cout)
  return out


def train_step(eng, x_dev, reducer, lr, global_batch):
  eng.X[0].interior().copy_(x_dev)
  eng.forward()
  eng.ctc_loss_grad(1.0 / global_batch)
  eng.backward(reducer.on_layer_done if reducer else None)
  if reducer:
    reducer.finish()
  eng.apply_update(lr)


DOMINANT = 'gemm_nn_kernel<128,128,2,2,0,true>'


def measure_dominant_kernel(eng, batch, reps=10):
  """HIP-event timing of the dominant kernel symbol gemm_nn_kernel<128,128,2,2,0,true>: the forward
  convolution of every layer whose packed width is a multiple of 128 and whose k-tiles are all whole
  (L1..L9 = 95.8 % of the forward MACs; 9 launches per step, the same launches rocprofv3 --stats averages;
  L0's 80-channel input takes the clamped-address variant of the same kernel).  Events are recorded on
  the stream the kernels are launched on; two untimed passes first so clocks are ramped."""
  from speecht_amd._lib import call
  flops = conv_flops(eng, batch)
  wide = [i for i, l in enumerate(eng.layers) if l.n_pad % 128 == 0 and l.cin_pitch % 32 == 0]
  if not wide:
    return None
  s = eng.stream_ptr

  def launch(i):
    l = eng.layers[i]
    pf, pb = eng._slice(eng.params, i)
    call('st_conv1d_nwc_fwd_f32', eng.X[i].ref, eng._ptr(pf), eng._ptr(pb), l.width, l.stride,
         eng.geo[i][2], int(l.relu), eng.X[i + 1].ref, s)
  for _ in range(2):
    for i in wide:
      launch(i)
  evs = []
  for _ in range(reps):
    for i in wide:
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      launch(i)
      e1.record()
      evs.append((i, e0, e1))
  torch.cuda.synchronize()
  tot_ms = sum(e0.elapsed_time(e1) for _, e0, e1 in evs) / reps
  tot_flops = sum(flops[i] for i in wide)
  tot_bytes = sum(4.0 * (batch * eng.geo[i][0] * eng.layers[i].cin + eng.layers[i].width * eng.layers[i].cin *
                         eng.layers[i].cout + batch * eng.geo[i][1] * eng.layers[i].cout) for i in wide)
  n = len(wide)
  avg_ms = tot_ms / n
  achieved = tot_flops / (tot_ms * 1e-3) / 1e12
  return dict(bound='mfma', kernel=DOMINANT + ' (conv forward, layers %s)' % ','.join('L%d' % i for i in wide),
              achieved=round(achieved, 2), peak=PEAK_F32_TFLOPS, unit='TFLOP/s',
              frac=round(achieved / PEAK_F32_TFLOPS, 4), traffic=None, traffic_unit='bytes/launch (PMC, see profiles/)',
              avg_launch_ms=round(avg_ms, 4), launches_per_step=n,
              algorithmic_gflop_per_launch=round(tot_flops / n / 1e9, 2),
              algorithmic_mb_per_launch=round(tot_bytes / n / 1e6, 2),
              hbm_frac_of_peak=round(tot_bytes / (tot_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4))


def measure_mel(dev, batch, seconds, n_mels, reps=5):
  """calc_power_spectrogram (preprocessing.py:36-58) on `batch` resident 16 kHz clips: the reference
  runs it offline (speecht-cli preprocess), so it is reported beside, not inside, the step metric."""
  import ctypes
  from speecht_amd import _lib
  from speecht_amd.preprocessing import mel_filterbank
  n = int(seconds * 16000)
  # deterministic clips of SURVEY 8(d): clip(0.1 * N(0,1), -1, 1), rng seeded 1234 + utterance index
  clips = [np.clip(0.1 * np.random.default_rng(1234 + i).standard_normal(n), -1.0, 1.0).astype(np.float32) for i in range(batch)]
  audio = torch.as_tensor(np.concatenate(clips)).to(dev)
  frames = 1 + n // 160
  s_off = torch.arange(batch + 1, dtype=torch.int64, device=dev) * n
  f_off = torch.arange(batch + 1, dtype=torch.int64, device=dev) * frames
  basis = torch.as_tensor(mel_filterbank(16000.0, 512, n_mels).astype(np.float32)).to(dev)
  out = torch.empty(batch * frames * n_mels, dtype=torch.float32, device=dev)
  ws = torch.empty(_lib.load().st_melspec_ws(batch, batch * frames, n_mels) // 4 + 64, dtype=torch.float32, device=dev)
  P = lambda t: ctypes.c_void_p(t.data_ptr())
  run = lambda: _lib.call('st_melspec_f32', P(audio), P(s_off), batch, n, P(basis), n_mels, 512, 160, P(f_off),
                          batch * frames, P(out), P(ws), ws.numel() * 4,
                          ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
  run()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    run()
  e1.record()
  e1.synchronize()
  ms = e0.elapsed_time(e1) / reps
  bytes_alg = batch * (n * 4 + frames * n_mels * 4)
  return dict(utterances_per_s=round(batch / (ms * 1e-3), 1), ms_per_batch=round(ms, 4),
              algorithmic_gbs=round(bytes_alg / (ms * 1e-3) / 1e9, 1), bound='hbm/launch', n_mels=n_mels)


def cpu_baseline(n_mels, frames, utts=2, steps=2):
  """The numpy oracle (CPU restatement of the reference path; TF1 cannot be installed) timed on
  the host cores: `utts` utterances of the same 10 s workload, fp32, full-width network."""
  from oracle import w2l_oracle as O
  layers = WL.w2l_layers(n_mels)
  params = WL.xavier_params(layers, seed=42, bias_range=0.0, dtype=np.float32)
  x, sl, labels = WL.make_batch([frames] * utts, n_mels, seed=0)
  x = x.astype(np.float32)
  st = O.zero_opt_state(params)
  O.train_step(x[:1, :201], [201], [labels[0][:20]], params, layers, st, lr=1e-4)     # warm-up (BLAS threads)
  t0 = time.time()
  for _ in range(steps):
    O.train_step(x, sl, labels, params, layers, st, lr=1e-4)
  dt = (time.time() - t0) / steps
  return dict(value=round(utts / dt, 4), unit='utterances/s', cores=os.cpu_count(), kind='port',
              sample='{} x 10 s utterances, {} full training steps of oracle/w2l_oracle.py (numpy fp32, '
                     'BLAS threads = host cores)'.format(utts, steps))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--warmup', type=int, default=5)
  ap.add_argument('--batch', type=int, default=32, help='utterances per GPU')
  ap.add_argument('--seconds', type=float, default=10.0)
  ap.add_argument('--mels', type=int, default=80)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-alt', action='store_true', help='skip the bf16x6 / bf16 side measurements')
  ap.add_argument('--force-allreduce', action='store_true', help='run the RCCL all-reduce path even on 1 rank (self-test)')
  ap.add_argument('--conv-mode', choices=('fp32', 'bf16x6', 'bf16'), default=None,
                  help='arithmetic of the timed loop (default fp32 = BASELINE configs[1]; bf16 = configs[3] arithmetic)')
  ap.add_argument('--allreduce', choices=('torch', 'rccl'), default=None,
                  help='gradient exchange transport: torch.distributed (default) or the library\'s st_allreduce_* (RCCL)')
  args = ap.parse_args()

  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world > 1 or args.force_allreduce:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29577')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    # test knobs (not used by the driver): ST_SHARE_GPU=1 maps every rank to cuda:0 and ST_DIST_BACKEND=gloo swaps
    # the transport, so that the world_size > 1 control flow can be exercised on a single-GPU box
    backend = os.environ.get('ST_DIST_BACKEND', 'nccl')
    if os.environ.get('ST_SHARE_GPU'):
      local_rank = 0
    torch.cuda.set_device(local_rank)
    if backend == 'nccl':
      dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
    else:
      dist.init_process_group(backend, rank=rank, world_size=world)
  assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node {} for --gpus {}'.format(args.gpus, args.gpus)
  dev = torch.device('cuda', local_rank)
  torch.cuda.set_device(dev)

  frames = 1 + int(args.seconds * 16000) // 160
  layers = WL.w2l_layers(args.mels)
  eng = Wav2LetterEngine(layers, device=dev, conv_mode=args.conv_mode)
  eng.set_weights(WL.xavier_params(layers, seed=42, bias_range=0.0, dtype=np.float32))   # same replica everywhere
  x, seq_lens, labels = WL.make_batch([frames] * args.batch, args.mels, seed=100 + rank)
  eng.load_batch(x, seq_lens)
  eng.set_labels(labels)
  x_dev = torch.as_tensor(x, dtype=torch.float32).to(dev)
  reducer = GradientAllReducer(eng.grads, eng.layer_ranges, force=args.force_allreduce, transport=args.allreduce) if (world > 1 or args.force_allreduce) else None
  global_batch = args.batch * world
  lr = 1e-4

  def sync():
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  for _ in range(args.warmup):
    train_step(eng, x_dev, reducer, lr, global_batch)
  sync()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    train_step(eng, x_dev, reducer, lr, global_batch)
  sync()
  elapsed = time.perf_counter() - t0
  if world > 1:
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t[0])
  eng.check_ctc_status()
  loss = float(eng.loss.mean())
  replicas_identical = None
  if world > 1:
    # data-parallel invariant: every rank applied the same averaged gradient to the same weights, so the
    # replicas must still be bit-identical (checked outside the timed region: two small all-reduces)
    p64 = eng.params.double()
    digest = torch.stack([p64.sum(), (p64 * p64).sum(), p64[::4097].abs().sum()])
    lo, hi = digest.clone(), digest.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    replicas_identical = bool(torch.equal(lo, hi))

  if rank == 0:
    ms = elapsed / args.steps * 1e3
    fl = conv_flops(eng, args.batch)
    step_gflop = (3.0 * sum(fl) - fl[0]) / 1e9     # fwd + bwd-data (not for L0) + bwd-filter
    out = {
        'metric': 'utterances/sec (training step, 10 s@16 kHz, batch 32)',
        'value': round(global_batch / (elapsed / args.steps), 2), 'unit': 'utterances/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': {'fp32': 'f32', 'bf16x6': 'f32 (bf16x6 split)', 'bf16': 'bf16 activations, f32 accumulate/CTC/Adam'}[eng.conv_mode],
        'data': 'synthetic',
        'config': {'workload': ('configs[3] arithmetic: data-parallel training step, batch {} per GPU of {:g} s synthetic clips, '
                                '{}-mel, bf16 activations / fp32 CTC' if eng.conv_mode == 'bf16' else
                                'configs[1]: {}xMI355X training step, batch {} per GPU of {:g} s synthetic clips, {}-mel, '
                                'default Wav2Letter depth, fp32').format(*(([] if eng.conv_mode == 'bf16' else [world]) +
                                                                           [args.batch, args.seconds, args.mels])),
                   'global_batch': global_batch, 'frames': frames, 'parallelism': 'dp%d' % world,
                   'allreduce': reducer.transport if reducer else None},
        'final_avg_loss': round(loss, 4),
        'replicas_identical': replicas_identical,
        'step_tflops_algorithmic': round(step_gflop / ms, 2),
    }
    out['roofline'] = measure_dominant_kernel(eng, args.batch)
    traffic_file = os.path.join(ROOT, 'profiles', 'traffic.json')
    if out['roofline'] and os.path.exists(traffic_file):
      out['roofline']['traffic'] = json.load(open(traffic_file)).get('bytes_per_launch')
    util_file = os.path.join(ROOT, 'profiles', 'mfma_util.json')
    if out['roofline'] and os.path.exists(util_file):      # PMC pass (scripts/gpu_mfma_util.sh), padded work included
      out['roofline']['mfma_busy_pmc'] = json.load(open(util_file)).get('gemm_nn_kernel<128, 128, 2, 2, 0, true>', {}).get('mfma_busy_frac_at_2p4ghz')
    if world == 1:
      out['mel_features'] = measure_mel(dev, args.batch, args.seconds, args.mels)
    if world == 1 and eng.conv_mode == 'fp32' and not args.no_alt:
      # Side measurements on the same box and inputs, NOT the headline:
      #  * bf16x6 (experimental): fp32 operands split exactly into 3 bf16 pieces, 6 cross terms on the bf16
      #    matrix pipe, fp32 accumulate; passes the same parity tests as the fp32 path.
      #  * bf16: BASELINE configs[3]'s arithmetic ("bf16 activations / fp32 CTC") on one GPU -- reduced
      #    precision by design, parity against the oracle's bf16 storage model (tests/test_gpu_bf16.py).
      alts = [('alt_bf16x6', 'bf16x6', 'f32 operands, exact 3-way bf16 split, 6 bf16 MFMA terms, f32 accumulate',
               'experimental opt-in (ST_CONV_MODE=bf16x6); not the headline value'),
              ('alt_bf16', 'bf16', 'bf16 activations + activation gradients, f32 masters/accumulate/logits/CTC/Adam',
               'configs[3] arithmetic on 1 GPU (ST_CONV_MODE=bf16); reduced precision, not the headline value')]
      for key, mode, dtype, note in alts:
        alt = Wav2LetterEngine(layers, device=dev, conv_mode=mode)
        alt.params.copy_(eng.params)
        alt.mark_weights_changed()
        alt.load_batch(x, seq_lens)
        alt.set_labels(labels)
        for _ in range(args.warmup):
          train_step(alt, x_dev, None, lr, global_batch)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
          train_step(alt, x_dev, None, lr, global_batch)
        torch.cuda.synchronize()
        alt_ms = (time.perf_counter() - t1) / args.steps * 1e3
        out[key] = {'value': round(args.batch / alt_ms * 1e3, 2), 'unit': 'utterances/s', 'ms_per_step': round(alt_ms, 3),
                    'step_tflops_algorithmic': round(step_gflop / alt_ms, 2),
                    'final_avg_loss': round(float(alt.loss.mean()), 4), 'dtype': dtype, 'note': note}
        del alt
        torch.cuda.empty_cache()
    if world == 1 and not args.no_cpu_baseline:
      out['cpu_baseline'] = cpu_baseline(args.mels, frames)
    print(json.dumps(out))
  if dist.is_initialized():
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
