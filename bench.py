#!/usr/bin/env python3
"""bench.py -- utterances/sec of one Wav2Letter TRAINING step on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1 without WORLD_SIZE in the environment: re-launches itself under torch.distributed.run, one rank per
  GPU over RCCL; under an external torch.distributed.run it just joins the job)

Workload (BASELINE.json configs[1]): per GPU a batch of 32 synthetic 10 s @ 16 kHz
utterances = 1001 frames x 80 mel features (z-normalised synthetic features; the reference
extracts features offline, preprocessing.py:212-241), 150-character labels, default Wav2Letter
depth (speech_model.py:275-295), fp32.  One step = batch -> 11-layer conv forward -> CTC loss +
gradient -> back-prop -> (gradient all-reduce) -> global-norm clip -> TF-Adam.  The timed step also carries what
SURVEY 8(d) counts into it: the H2D copy of the padded feature batch from pinned host memory (on a copy stream,
double-buffered, so batch k+1 travels while batch k computes) and the upload of the labels.
Weak scaling: every rank processes its own 32 utterances; value = N*32 / max-over-ranks time.

Prints ONE JSON line on rank 0 with the driver's contract plus `roofline` (dominant kernel, HIP
event timing) and `cpu_baseline` (the numpy oracle timed on the host cores, N=1 only).
"""
import argparse
import json
import re
import os
import socket
import subprocess
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


class HostFeed:
  """The input side of a step: pinned host batch -> HBM (async, double-buffered) + label upload."""

  def __init__(self, eng, x, seq_lens, labels):
    self.eng, self.seq_lens, self.labels = eng, seq_lens, labels
    self.x_pinned = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).pin_memory()
    self.staged = eng.stage_host_batch(self.x_pinned)

  def next(self):
    cur = self.staged
    self.staged = self.eng.stage_host_batch(self.x_pinned)      # H2D of the next step's batch, behind this step's kernels
    self.eng.load_batch(cur, self.seq_lens)
    self.eng.set_labels(self.labels)


def train_step(eng, feed, reducer, lr, global_batch):
  feed.next()
  eng.forward()
  eng.ctc_loss_grad(1.0 / global_batch)
  eng.backward(reducer.on_layer_done if reducer else None, reducer.hook_layers if reducer else None)
  if reducer:
    reducer.finish()
  eng.apply_update(lr)


def _lib_handle():
  from speecht_amd import _lib
  return _lib.load()


def trace_symbol(line):
  """Launch-trace line -> the kernel symbol the roofline groups by (conv GEMMs keep their epilogue flavour)."""
  tok = line.split()
  if tok[0].startswith('gemm_nn<'):
    return tok[0] + (' epi=1' if (len(tok) > 1 and tok[1] == 'epi=1') else ' epi=0')
  return tok[0]


def trace_key(line):
  """A trace line without its measurements: identifies (kernel, shape, policy) of a launch."""
  return ' '.join(t for t in line.split() if not (t.startswith('gflop=') or t.startswith('ms=') or t.startswith('mb=')))


def trace_field(line, name):
  m = re.search(r'(?:^| )%s=([-0-9.e+]+)' % name, line)
  return float(m.group(1)) if m else None


def measure_dominant_kernel(eng, batch, step_fn, step_ms, reps=3, profiled_steps=6):
  """`roofline`: the hardware rate of every matrix-pipe kernel of the training step, measured INSIDE real steps.

  In-step (the contract number): `profiled_steps` real training steps run under the library's timed launch trace
  (st_trace_begin_timed: a pair of HIP events on the launch's own stream around every matrix-pipe launch -- per-bin
  products, W-tap GEMMs, the DFT / inverse-DFT transforms; side streams included, so a launch that shares the chip with
  the other stream's launch shows the time it really took).  Launches are grouped by kernel symbol; `roofline` is the
  GEMM symbol with the largest share of the step, `by_kernel` the others; rocprofv3 --kernel-trace --stats of the same
  command (profiles/) averages the same launches under the same symbol.
  Isolated (reported beside it): each launch of the step on its own, back to back on one stream, `reps` times (median).
  FLOPs: the ALGORITHMIC ones of each launch -- 2*B*T'*W*Cin*Cout for a W-tap launch, 2 * rows * 2Cin * 2Cout * bins
  for a per-bin product (complex as 4 real multiplies; unpadded channel counts, real row count) -- matched to the
  in-step launches through the trace line (kernel, shape, policy); EXECUTED FLOPs (padding included) come from the
  trace lines themselves (`gflop=`).  Which kernel instantiation a call runs is read from the trace, not assumed."""
  import ctypes
  from speecht_amd._lib import call, launch_trace
  if eng.conv_mode == 'bf16':
    block = roofline_bf16_in_step(eng, step_fn, step_ms)
    if block is not None:
      block['isolated_forward_wide_layers'] = measure_dominant_kernel_bf16(eng, batch, conv_flops(eng, batch), 10)
    return block, None
  flops = conv_flops(eng, batch)
  s, P = eng.stream_ptr, eng._ptr
  ws, wsb = P(eng.wgrad_ws), eng.wgrad_ws.numel() * 4
  launches = []                                  # (label, flops, bytes, fn)
  for i, l in enumerate(eng.layers):
    pl = eng.geo[i][2]
    pf, pb = eng._slice(eng.params, i)
    gf, gb = eng._slice(eng.grads, i)
    io_bytes = 4.0 * (batch * eng.geo[i][0] * l.cin + l.width * l.cin * l.cout + batch * eng.geo[i][1] * l.cout)
    if i in eng.fft and eng.fft_conv:
      f = eng.fft[i]                             # (layer 0 runs on its polyphase view: f['width'] taps over f['cin'] channels)
      blocks, bins, rows_pad = (ctypes.c_int() for _ in range(3))
      call('st_conv1d_fft_plan', f['width'], eng.geo[i][1], batch, None, None, ctypes.byref(blocks), ctypes.byref(bins),
           ctypes.byref(rows_pad))
      rows, nb, rp = batch * blocks.value, bins.value, rows_pad.value
      ka, nf, nbk = 2 * (-(-f['cin_pitch'] // 64) * 64), 2 * l.n_pad, 2 * l.nt_pad
      cin_real = l.cin * (l.stride if f['shift'] is not None else 1)
      fl = 2.0 * rows * (2 * cin_real) * (2 * l.cout) * nb
      nbytes = 4.0 * nb * (rows * 2 * cin_real + 4 * cin_real * l.cout + rows * 2 * l.cout)
      # the layer's workspace as its own entry points lay it out: [stream-K area of the per-bin products | output spectra]
      tail_bytes = _lib_handle().st_gemm_nn_batched_ws_bytes()
      tail, outp = P(f['ws']), ctypes.c_void_p(f['ws'].data_ptr() + tail_bytes)
      form = _lib_handle().st_conv1d_fft_three_products(f['width'], f['cin_pitch'], l.cout)
      if form == 2:
        # the three-product form (conv_fft.hip g3_form): rows [S_r + S_i | S_i | S_r | S_i - S_r] / [Z_r + Z_i | Z_r | Z_i], filter planes
        # G_r, G_i - G_r, -(G_r + G_i) -- the calls below are the library's own (st_conv1d_nwc_*_fft_f32).  Algorithmic FLOPs: three
        # real products per complex one, what the form executes
        fl *= 0.75
        half, npo = ka // 2, nf // 2
        i3 = lambda *v: (ctypes.c_int64 * 3)(*v)
        launches.append(('L%d fwd x%d bins' % (i, nb), fl, nbytes, lambda f=f, half=half, npo=npo, rp=rp, nb=nb, outp=outp: call(
            'st_gemm_nn_g3_batched_f32', P(f['sf']), 4 * half, rp * 4 * half, i3(0, half, 2 * half), P(f['gfwd']), npo, 3 * half * npo,
            i3(0, half * npo, 2 * half * npo), 0, outp, nf, rp * nf, npo, rp, half, npo, nb, s)))
        launches.append(('L%d bwd x%d bins' % (i, nb), fl, nbytes, lambda f=f, half=half, npo=npo, rp=rp, nb=nb, outp=outp: call(
            'st_gemm_nn_g3_batched_f32', P(f['zf']), 3 * npo, rp * 3 * npo, i3(0, 2 * npo, npo), P(f['gfwd']), npo, 3 * half * npo,
            i3(0, 2 * half * npo, half * npo), 1, outp, ka, rp * ka, half, rp, npo, half, nb, s)))
        launches.append(('L%d wgrad x%d bins' % (i, nb), fl, nbytes, lambda f=f, half=half, npo=npo, rp=rp, nb=nb, outp=outp: call(
            'st_gemm_tn_g3_batched_f32', P(f['sf']), 4 * half, rp * 4 * half, i3(2 * half, 3 * half, 0), P(f['zf']), 3 * npo, rp * 3 * npo,
            i3(0, 2 * npo, npo), outp, 2 * half * npo, half * npo, rp, half, npo, nb, s)))
        continue
      assert form == 0, 'bench: isolated launches of a layer with three-part gradient spectra rows are not described here'
      launches.append(('L%d fwd x%d bins' % (i, nb), fl, nbytes, lambda f=f, ka=ka, nf=nf, rp=rp, nb=nb, tail=tail, outp=outp: call(
          'st_gemm_nn_batched_ws_f32', P(f['sf']), ka, 2 * rp * ka, P(f['gfwd']), ka * nf, outp, nf, rp * nf, rp, ka, nf, nb, tail, tail_bytes, s)))
      if i > 0:
        # back-prop to the input: the same spectra read as a transposed operand (X = Z gfwd^T)
        launches.append(('L%d bwd x%d bins' % (i, nb), fl, nbytes, lambda f=f, ka=ka, nf=nf, rp=rp, nb=nb, tail=tail, outp=outp: call(
            'st_gemm_nn_batched_bt_ws_f32', P(f['zf']), nf, rp * nf, P(f['gfwd']), ka * nf, outp, ka, rp * ka, rp, nf, ka, nb, tail, tail_bytes, s)))
      if (ka // 2) % 128 == 0:
        # the lag products as the library runs them: spectra rows read as two half-length rows, the real and the imaginary
        # product of a bin as batches 2 b and 2 b + 1 over the spectra and their rotated copy, both against the bin's dz spectra
        launches.append(('L%d wgrad x%d bins' % (i, nb), fl, nbytes, lambda f=f, ka=ka, nf=nf, rp=rp, nb=nb, outp=outp: call(
            'st_gemm_tn_batched_shared_f32', P(f['sf']), ka // 2, rp * ka, P(f['zf']), nf // 2, rp * nf, outp, (ka // 2) * (nf // 2), 2 * rp,
            ka // 2, nf // 2, 2 * nb, 1, s)))
      else:
        launches.append(('L%d wgrad x%d bins' % (i, nb), fl, nbytes, lambda f=f, ka=ka, nf=nf, rp=rp, nb=nb, outp=outp: call(
            'st_gemm_tn_batched_f32', P(f['sf']), ka, 2 * rp * ka, P(f['zf']), nf, rp * nf, outp, ka * nf, rp, ka, nf, nb, s)))
      continue
    launches.append(('L%d fwd' % i, flops[i], io_bytes, lambda i=i, l=l, pl=pl, pf=pf, pb=pb: call(
        'st_conv1d_nwc_fwd_ws_f32', eng.X[i].ref, P(pf), P(pb), l.width, l.stride, pl, int(l.relu), eng.X[i + 1].ref, ws, wsb, s)))
    if i > 0:
      if eng._transposed_in_place(i):
        launches.append(('L%d bwd' % i, flops[i], io_bytes, lambda i=i, l=l, pf=pf: call(
            'st_conv1d_1tap_bwd_data_bias_f32', eng.dZ[i].ref, P(pf), eng.X[i].ref if eng.layers[i - 1].relu else None,
            eng.dZ[i - 1].ref, P(eng._slice(eng.grads, i - 1)[1]), ws, wsb, s)))
      else:
        launches.append(('L%d bwd' % i, flops[i], io_bytes, lambda i=i, l=l, pl=pl: call(
            'st_conv1d_nwc_bwd_data_bias_f32', eng.dZ[i].ref, P(eng.packed_t[i]), l.width, pl,
            eng.X[i].ref if eng.layers[i - 1].relu else None, eng.dZ[i - 1].ref, P(eng._slice(eng.grads, i - 1)[1]), ws, wsb, s)))
    launches.append(('L%d wgrad' % i, flops[i], io_bytes, lambda i=i, l=l, pl=pl, gf=gf: call(
        'st_conv1d_nwc_bwd_filter_f32', eng.X[i].ref, eng.dZ[i].ref, l.width, l.stride, pl, P(gf), None, ws, wsb, s)))
  if not launches:
    return None, None

  # ---- isolated: the same launches one by one; their trace lines give the (kernel, shape) -> algorithmic work map ----
  symbols, keys = [], []
  for _, _, _, fn in launches:
    with launch_trace() as tr:
      fn()
    symbols.append(trace_symbol(tr.lines[0]))
    keys.append(trace_key(tr.lines[0]))
  evs = []
  for _ in range(reps):
    for k, (_, _, _, fn) in enumerate(launches):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      fn()
      e1.record()
      evs.append((k, e0, e1))
  torch.cuda.synchronize()
  samples = [[] for _ in launches]
  for k, e0, e1 in evs:
    samples[k].append(e0.elapsed_time(e1))
  iso = [float(np.median(v)) for v in samples]   # median over the repetitions: one disturbed interval does not move a launch
  alg = {}                                       # trace key -> (flops, bytes) of one such launch (L1..L7 share a key: same work)
  for k, key in enumerate(keys):
    alg.setdefault(key, (launches[k][1], launches[k][2]))

  # ---- in-step: real training steps under the timed trace, ONE kernel symbol timed per pass ----
  # (a timed launch costs its stream ~7 us -- hipExtLaunchKernel with events gives up back-to-back dispatch -- so timing all
  # ~100 launches of a step at once stretches it by 10 % and every overlapped launch with it; one symbol's <= 16 launches: ~1 %)
  with launch_trace() as tr:
    step_fn()
  torch.cuda.synchronize()
  name_lines = list(tr.lines)                                    # every traced launch of a step, with its executed GFLOP
  tokens = sorted({l.split()[0] for l in name_lines})
  step_lines, pass_ms = [], {}
  for tok in tokens:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with launch_trace(timed=True, only=tok + ' ') as tr:
      for _ in range(profiled_steps):
        step_fn()
      torch.cuda.synchronize()
      pass_ms[tok] = (time.perf_counter() - t0) / profiled_steps * 1e3
    step_lines += [l for l in tr.lines if trace_field(l, 'ms') is not None and trace_field(l, 'ms') >= 0.0]
  profiled_ms = float(np.max(list(pass_ms.values())))

  def rate(fl, ms):
    return fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0

  def iso_group(sym):
    idx = [k for k, s_ in enumerate(symbols) if s_ == sym]
    ms, fl = sum(iso[k] for k in idx), sum(launches[k][1] for k in idx)
    return dict(ms_per_step=round(ms, 4), avg_launch_ms=round(ms / len(idx), 4), achieved=round(rate(fl, ms), 2),
                frac=round(rate(fl, ms) / PEAK_F32_TFLOPS, 4),
                per_launch={launches[k][0]: dict(ms=round(iso[k], 4), tflops=round(rate(launches[k][1], iso[k]), 1)) for k in idx},
                timing='each launch alone, back to back on one stream, median of %d' % reps)

  def step_group(sym):
    rows = [l for l in step_lines if trace_symbol(l) == sym]
    ms = sum(trace_field(l, 'ms') for l in rows) / profiled_steps
    ex = sum(trace_field(l, 'gflop') or 0.0 for l in rows) / profiled_steps
    known = [l for l in rows if trace_key(l) in alg]
    # (a shape the isolated pass did not launch counts with its executed FLOPs)
    fl = sum(alg[trace_key(l)][0] if trace_key(l) in alg else 1e9 * (trace_field(l, 'gflop') or 0.0) for l in rows) / profiled_steps
    by = sum(alg[trace_key(l)][1] for l in known) / profiled_steps
    n = len(rows) / profiled_steps
    per = {}
    for l in rows:
      per.setdefault(trace_key(l), []).append(trace_field(l, 'ms'))
    g = dict(kernel=sym, launches_per_step=round(n, 2), ms_per_step=round(ms, 4), avg_launch_ms=round(ms / n, 4),
             executed_gflop_per_launch=round(ex / n, 2), executed_tflops=round(rate(ex * 1e9, ms), 2))
    if known:
      g.update(achieved=round(rate(fl, ms), 2), frac=round(rate(fl, ms) / PEAK_F32_TFLOPS, 4),
               algorithmic_gflop_per_launch=round(fl / n / 1e9, 2), algorithmic_mb_per_launch=round(by / n / 1e6, 2))
    mb = [trace_field(l, 'mb') for l in rows]
    if rows and all(v is not None for v in mb):
      # the DFT / inverse-DFT transforms: bound by the bytes they move (HBM roof) with their matrix-pipe share beside it -- the
      # trace line carries both the executed FLOPs and the algorithmic bytes of the launch (csrc/conv_fft.hip)
      g.update(bound='hbm', algorithmic_mb_per_launch=round(sum(mb) / len(rows), 2),
               hbm_gbs=round(sum(mb) / profiled_steps * 1e6 / (ms * 1e-3) / 1e9, 1),
               hbm_frac=round(sum(mb) / profiled_steps * 1e6 / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
               mfma_frac=round(rate(ex * 1e9, ms) / PEAK_F32_TFLOPS, 4))
    g['per_shape'] = {k: dict(launches_per_step=round(len(v) / profiled_steps, 2), avg_ms=round(float(np.mean(v)), 4),
                              tflops=round(rate(alg[k][0], float(np.mean(v))), 1) if k in alg else None) for k, v in per.items()}
    return g

  in_step = sorted((step_group(sym) for sym in sorted({trace_symbol(l) for l in step_lines})), key=lambda g: -g['ms_per_step'])
  gemms = [g for g in in_step if 'frac' in g]
  dom = gemms[0]
  out = dict(bound='mfma', peak=PEAK_F32_TFLOPS, unit='TFLOP/s', traffic=None, traffic_unit='bytes/launch (PMC, see profiles/)')
  out.update({k: v for k, v in dom.items() if k != 'per_shape'})
  out['timing'] = ('in-step: every launch of this kernel inside %d real training steps (side streams running) launched through '
                   'hipExtLaunchKernel with start/stop events = the dispatch\'s own begin/end time stamps, what rocprofv3\'s kernel '
                   'trace reports (st_trace_begin_timed, one kernel symbol per pass); achieved = algorithmic FLOPs of those '
                   'launches / their summed time' % profiled_steps)
  out['per_shape'] = dom['per_shape']
  out['isolated'] = iso_group(dom['kernel']) if dom['kernel'] in symbols else None
  out['hbm_frac_of_peak'] = round(dom['algorithmic_mb_per_launch'] * 1e6 / (dom['avg_launch_ms'] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
  out['by_kernel'] = []
  for g in in_step:
    if g is dom:
      continue
    e = {k: v for k, v in g.items() if k != 'per_shape'}
    if g['kernel'] in symbols:
      ig = iso_group(g['kernel'])
      e['isolated'] = dict(frac=ig['frac'], achieved=ig['achieved'], ms_per_step=ig['ms_per_step'])
    out['by_kernel'].append(e)
  all_ms = sum(g['ms_per_step'] for g in gemms)
  all_fl = sum(g['algorithmic_gflop_per_launch'] * g['launches_per_step'] for g in gemms) * 1e9
  out['all_matrix_launches'] = dict(launches_per_step=round(sum(g['launches_per_step'] for g in gemms), 2), ms_per_step=round(all_ms, 3),
                                    achieved=round(rate(all_fl, all_ms), 2), frac=round(rate(all_fl, all_ms) / PEAK_F32_TFLOPS, 4),
                                    note='summed launch times; launches on the two streams of the backward pass overlap, so '
                                         'this can exceed their share of the step')
  executed = sum(trace_field(l, 'gflop') or 0.0 for l in name_lines)
  step = dict(step_executed_gflop=round(executed, 1), step_hw_tflops=round(executed / step_ms, 2),
              step_hw_frac=round(executed / step_ms / PEAK_F32_TFLOPS, 4),
              step_hw_note='FLOPs the matrix pipe really executed in one step (every traced launch incl. the DFT / inverse-DFT '
                           'transforms and the 29-class layer, channel padding included) / ms_per_step / %.1f TFLOP/s' % PEAK_F32_TFLOPS,
              profiled_ms_per_step=round(profiled_ms, 3),
              profiled_note='slowest of the per-symbol timed passes (wall clock per step, host synchronised per pass)')
  return out, step


def measure_dominant_kernel_bf16(eng, batch, flops, reps):
  """configs[3] arithmetic: the forward launches of gemm_nn_bf16_kernel<256,...> (L8, L9) against the dense bf16 MFMA
  peak (2.5 PFLOP/s nominal; the board sustains ~1.8 on random operands and clocks down to ~2.0 GHz under this
  kernel)."""
  from speecht_amd._lib import call
  s = eng.stream_ptr
  L = len(eng.layers)
  wide = [i for i, l in enumerate(eng.layers) if l.n_pad >= 256 and batch * eng.geo[i][1] >= 4096]
  if not wide:
    return None

  def launch(i):
    l = eng.layers[i]
    last = i + 1 == L
    call('st_conv1d_nwc_fwd_ws_bf16', eng.X[i].ref, eng._ptr(eng.Xb[i]), eng._ptr(eng.Wb[i]), eng._ptr(eng._slice(eng.params, i)[1]),
         l.width, l.stride, eng.geo[i][2], int(l.relu), eng.X[i + 1].ref, None if last else eng._ptr(eng.Xb[i + 1]),
         eng._ptr(eng.X[i + 1].buf) if last else None, eng._ptr(eng.wgrad_ws_b), eng.wgrad_ws_b.numel() * 4, s)
  for _ in range(2):
    for i in wide:
      launch(i)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    for i in wide:
      launch(i)
  e1.record()
  e1.synchronize()
  ms = e0.elapsed_time(e1) / reps
  fl = sum(flops[i] for i in wide)
  achieved = fl / (ms * 1e-3) / 1e12
  return dict(bound='mfma', kernel='gemm_nn_bf16_kernel<256,2,4,32,1,4,true,true> (conv forward, layers %s)' % ','.join('L%d' % i for i in wide),
              achieved=round(achieved, 1), peak=2500.0, unit='TFLOP/s', frac=round(achieved / 2500.0, 4), traffic=None,
              avg_launch_ms=round(ms / len(wide), 4), launches_per_step=len(wide),
              algorithmic_gflop_per_launch=round(fl / len(wide) / 1e9, 2))


def roofline_bf16_in_step(eng, step_fn, ms_per_step, steps=4):
  """`roofline` block of the bf16-activation step (configs[3] arithmetic), measured INSIDE real steps: every launch of the bf16
  matrix kernels (W-tap / per-bin GEMMs `gemm_nn_bf16<..>`, transposing-read filter gradients `wgrad_tr_bf16<..>`) carries
  start / stop events from its own dispatch (the library's timed launch trace), grouped by kernel symbol; the symbol with the
  largest share of the step is the block, the others go to `by_kernel`.  FLOPs are the EXECUTED ones of each launch (its trace
  line: tile padding included), so `achieved` is a hardware rate against the dense bf16 MFMA peak (2.5 PFLOP/s nominal; the
  board sustains ~1.8 on random operands).  The HBM view beside it: algorithmic bytes of the step (SURVEY 8(d) with
  2-byte activations) over the step time against 8 TB/s, and -- when profiles/traffic_bf16.json was collected for these sources
  -- the fabric bytes the TCC counters saw."""
  from speecht_amd._lib import launch_trace
  from speecht_amd.build import source_digest
  torch.cuda.synchronize()
  with launch_trace(timed=True, only='bf16<') as tr:
    for _ in range(steps):
      step_fn()
    torch.cuda.synchronize()
  groups = {}
  for line in tr.lines:
    ms, gf = trace_field(line, 'ms'), trace_field(line, 'gflop')
    if ms is None or ms < 0 or gf is None:
      continue
    g = groups.setdefault(line.split()[0], dict(ms=0.0, gflop=0.0, n=0))
    g['ms'] += ms; g['gflop'] += gf; g['n'] += 1
  if not groups:
    return None
  def block(sym):
    g = groups[sym]
    ach = g['gflop'] / g['ms']                       # GFLOP / ms = TFLOP/s
    return dict(kernel=sym, launches_per_step=round(g['n'] / steps, 2), avg_launch_ms=round(g['ms'] / g['n'], 4),
                ms_per_step=round(g['ms'] / steps, 4), executed_gflop_per_launch=round(g['gflop'] / g['n'], 3),
                achieved=round(ach, 1), peak=2500.0, unit='TFLOP/s', frac=round(ach / 2500.0, 4))
  dom = max(groups, key=lambda k: groups[k]['ms'])
  out = dict(bound='mfma', **block(dom))
  out['by_kernel'] = [block(k) for k in sorted(groups, key=lambda k: -groups[k]['ms']) if k != dom]
  total_ms = sum(g['ms'] for g in groups.values()) / steps
  total_gf = sum(g['gflop'] for g in groups.values()) / steps
  out['matrix_launches'] = dict(ms_per_step=round(total_ms, 3), executed_gflop_per_step=round(total_gf, 1),
                                achieved=round(total_gf / total_ms, 1), frac=round(total_gf / total_ms / 2500.0, 4),
                                note='summed launch times of all traced bf16 matrix launches; launches on different streams overlap')
  out['step_executed_gflop_traced'] = round(total_gf, 1)
  out['step_hw_frac'] = round(total_gf / ms_per_step / 2500.0, 4)
  # HBM side: activations (X 1..10, dZ 0..10) at 2 bytes, weights / gradients / Adam state at 4 (3 392.6 MB in fp32,
  # of which activations 1 757 MB)
  B = eng.X[0].batch
  act = sum(2.0 * B * t_out * l.cout for l, (_, t_out, _, _) in zip(eng.layers, eng.geo))       # one write of every layer output
  alg_bytes = 4.0 * B * eng.geo[0][0] * eng.layers[0].cin + 3.0 * act + 2.0 * act + 4.0 * 8 * eng.n_flat
  out['hbm'] = dict(algorithmic_bytes_per_step=round(alg_bytes), achieved_gbs=round(alg_bytes / ms_per_step / 1e6, 1), peak_gbs=PEAK_HBM_GBS,
                    frac=round(alg_bytes / ms_per_step / 1e6 / PEAK_HBM_GBS, 4),
                    note='algorithmic bytes: fp32 input, every activation written once and read by forward, back-prop mask and filter '
                         'gradient (2 bytes), every activation gradient written once and read twice, weights / gradient / Adam moments '
                         '(4 bytes: read p g m v, write p m v, norm pass); over the step time, against 8 TB/s')
  path = os.path.join(ROOT, 'profiles', 'traffic_bf16.json')
  out['traffic'] = None
  if os.path.exists(path):
    data = json.load(open(path))
    if data.get('source_digest') == source_digest():
      out['traffic'] = data.get('step_bytes')
      out['traffic_over_algorithmic'] = round(data['step_bytes'] / alg_bytes, 2) if data.get('step_bytes') else None
      out['traffic_gbs'] = round(data['step_bytes'] / ms_per_step / 1e6, 1) if data.get('step_bytes') else None
      out['traffic_source'] = 'profiles/traffic_bf16.json: FETCH_SIZE / WRITE_SIZE passes over `bench.py --steps-only --conv-mode bf16`, fabric bytes of one step'
    else:
      out['traffic_stale'] = 'profiles/traffic_bf16.json was collected for other sources (%s)' % str(data.get('source_digest'))[:12]
  return out


def host_cost(eng, feed, lr, global_batch, steps, ahead):
  """What the host side of a step costs: `host_enqueue_ms` is the CPU time inside the calls that enqueue one step (Python + ctypes;
  the pacing wait that keeps two steps in flight is NOT in it), `ms_per_step` the wall clock per step of the same loop.  (Rounds
  5 carried a whole-step HIP graph beside it: bit-identical, half the enqueue time, but a SLOWER replay -- 7.08 against 6.83 ms
  fp32, 2.49 against 2.42 bf16 -- because the capture froze a worse stream -> queue assignment; removed in round 6, docs/history/r5.md.)"""
  for _ in range(3):
    train_step(eng, feed, None, lr, global_batch)
  torch.cuda.synchronize()
  marks = [torch.cuda.Event() for _ in range(steps)]
  host = 0.0
  t0 = time.perf_counter()
  for k in range(steps):
    if ahead and k >= ahead:
      marks[k - ahead].synchronize()
    h0 = time.perf_counter()
    train_step(eng, feed, None, lr, global_batch)
    host += time.perf_counter() - h0
    marks[k].record()
  torch.cuda.synchronize()
  return dict(ms_per_step=round((time.perf_counter() - t0) / steps * 1e3, 3), host_enqueue_ms=round(host / steps * 1e3, 3))


def comm_model(eng, feed, lr, global_batch, step_ms, world=8, reps=5):
  """MODEL, not a measurement (every test box has one GPU): when, inside the backward pass, each of the four
  gradient buckets is complete (events recorded where the data-parallel hook would hand the bucket to RCCL: engine.backward with
  the bucket-boundary hooks, side streams joined as in a real exchange), and what an 8-rank all-reduce of that bucket would add
  to the step for an ASSUMED bus bandwidth: buckets go out in order on one collective stream, a ring / tree moves
  2 (N - 1) / N x bytes at `busbw`, the transfer that is not over when back-prop ends is exposed.  Nothing is claimed about the
  slowdown the transfers cause the kernels they run beside."""
  from speecht_amd.data_parallel import default_buckets
  ranges = eng.reduce_ranges
  buckets = default_buckets([e - s for s, e in ranges], ranges)
  hook_layers = {lo for lo, _, _ in buckets}
  samples = []
  for _ in range(reps):
    ready = {}
    def hook(i):
      ev = torch.cuda.Event(enable_timing=True)
      ev.record()
      ready[i] = ev
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    feed.next()
    eng.forward()
    eng.ctc_loss_grad(1.0 / global_batch)
    e0.record()
    eng.backward(hook, hook_layers)
    e1.record()
    eng.apply_update(lr)
    torch.cuda.synchronize()
    samples.append((e0.elapsed_time(e1), {i: e0.elapsed_time(ev) for i, ev in ready.items()}))
  bwd_ms = float(np.median([s[0] for s in samples]))
  per_bucket = []
  for lo, s, e in buckets:
    t = float(np.median([smp[1][lo] for smp in samples]))
    per_bucket.append(dict(first_layer=lo, mb=round((e - s) * 4 / 1e6, 2), ready_ms_after_backward_starts=round(t, 3),
                           ms_of_backward_left=round(bwd_ms - t, 3)))
  model = {}
  for busbw in (100.0, 200.0, 300.0):
    clock = 0.0
    for b in per_bucket:
      clock = max(clock, b['ready_ms_after_backward_starts']) + 2.0 * (world - 1) / world * b['mb'] / busbw     # MB / (GB/s) = ms
    exposed = max(0.0, clock - bwd_ms)
    model['busbw_%d_GBs' % int(busbw)] = dict(exposed_ms=round(exposed, 3), step_ms=round(step_ms + exposed, 3),
                                             utterances_per_s_8gpu=round(world * global_batch / (step_ms + exposed) * 1e3, 1),
                                             scaling_vs_8x=round(step_ms / (step_ms + exposed), 4))
  return dict(kind='MODEL from single-GPU event times, not a measurement', world_assumed=world, backward_ms=round(bwd_ms, 3),
              buckets=per_bucket, assumed=model,
              note='exposed = what the in-order all-reduces of the buckets (2 (N - 1) / N x bytes / busbw each, started when the bucket '
                   'is ready) still have to do when back-prop ends; the measured compute step is `step_ms` of this mode.  xGMI: 7 links x '
                   '~153 GB/s per GPU; RCCL on 8 MI300-class GPUs reaches 200-300 GB/s bus bandwidth on large messages, less on 7-16 MB ones')


def config2_rate(dev, layers, budget_s=1.0):
  """BASELINE configs[2] in the line: inference only, 256 utterances of 2-15 s (seeded), batch 64, bucketed by length and pipelined
  (inference.transcribe: host padding, H2D of every batch and read-back of the transcripts included), greedy decode; the shortest
  utterance's logits and greedy ids are checked against the float64 oracle."""
  from oracle import w2l_oracle as O          # checker only
  from speecht_amd import inference
  rng = np.random.default_rng(3)
  samples = rng.integers(32000, 240001, 256)
  frames = 1 + samples // 160
  feats = [rng.standard_normal((int(t), 80)).astype(np.float32) for t in frames]
  params = WL.xavier_params(layers, seed=42, bias_range=0.05, dtype=np.float32)
  eng = Wav2LetterEngine(layers, device=dev)
  eng.set_weights(params)
  for _ in range(2):                                                           # warm-up: every bucket shape described, pinned rings and
    ids, _ = inference.transcribe(eng, feats, 64, True, True)                  # decoder slots allocated, the pipeline's threads started
  torch.cuda.synchronize()
  t0, passes = time.perf_counter(), 0
  while time.perf_counter() - t0 < budget_s:
    ids, _ = inference.transcribe(eng, feats, 64, True, True)
    passes += 1
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  # oracle check on the shortest utterance, alone in its batch (same batch composition on both sides)
  k = int(np.argmin(frames))
  one = feats[k]
  eng.load_batch(one[None], [one.shape[0]])
  eng.forward()
  dec, _ = eng.greedy_decode()
  torch.cuda.synchronize()
  p64 = [(F.astype(np.float64), b.astype(np.float64)) for F, b in params]
  ref = O.wav2letter_forward(one[None].astype(np.float64), p64, layers)
  ref_dec, _ = O.ctc_greedy_decode(ref, np.array([one.shape[0] // 2]))
  err = float(np.max(np.abs(eng.logits_time_major().cpu().numpy() - ref)))
  buckets = inference.make_buckets(frames, 64)
  del eng
  torch.cuda.empty_cache()
  return dict(workload='configs[2]: inference, 256 utterances of 2-15 s (seeded), batch 64, length-bucketed + pipelined, greedy decode',
              utterances_per_s=round(passes * len(feats) / dt, 1), passes=passes, seconds=round(dt, 3),
              protocol='warm (two untimed passes over the pool first), then whole passes of the 256-utterance pool for ~1 s; every pass '
                       'fills and drains its 4-batch pipeline, so the rate is below scripts/bench_inference.py\'s, which streams a '
                       '2 048-utterance pool (32 batches per pass) through the same code',
              padding_overhead=round(inference.padding_overhead(frames, buckets), 4),
              audio_seconds_per_s=round(passes * float(samples.sum()) / 16000.0 / dt, 0),
              oracle_check=dict(utterance_frames=int(one.shape[0]), max_logit_err=err, greedy_ids_equal=bool(dec == ref_dec),
                                passed=bool(err < 1e-4 and dec == ref_dec)))


def config4_rate(dev, layers, batches=12, beam=16):
  """BASELINE configs[4] on one GPU's shard in the line: 16 utterances of 30 s (T' = 1501) resident, forward pass + LM-free prefix
  beam search (beam 16), the search of batch k on CU-masked decoder streams under the forward passes of the next batches
  (engine.beam_search_decode_async); the beam ids of utterance 0 are checked against the oracle search on the device's own logits."""
  from oracle import w2l_oracle as O          # checker only
  from speecht_amd import engine as E
  frames = 3001
  params = WL.xavier_params(layers, seed=42, bias_range=0.05, dtype=np.float32)
  eng = Wav2LetterEngine(layers, device=dev)
  eng.set_weights(params)
  x, seq_lens, _ = WL.make_batch([frames] * 16, 80, seed=7)
  eng.load_batch(x, seq_lens)
  eng.forward()
  std = float(eng.X[-1].interior().std())        # a random-init network's rows are nearly flat: scale the output layer to std 3
  w = eng.get_weights()
  w[-1] = (w[-1][0] * (3.0 / max(std, 1e-6)), w[-1][1] * (3.0 / max(std, 1e-6)))
  eng.set_weights(w)
  eng.load_batch(x, seq_lens)
  cs, ds = E.decoder_streams(dev, 2)
  def run(n):
    pending, last = [], None
    for _ in range(n):
      with torch.cuda.stream(cs):
        eng.forward()
        pending.append(eng.beam_search_decode_async(beam, ds))
      if len(pending) > 2:
        last = pending.pop(0).result()
    for h in pending:
      last = h.result()
    return last
  run(3)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  got, _ = run(batches)
  dt = time.perf_counter() - t0
  torch.cuda.synchronize()
  logits = eng.logits_time_major()[:, :1, :].cpu().numpy().astype(np.float64)
  ref_ids, _ = O.ctc_beam_search_decode(logits, np.array([frames // 2]), beam)
  del eng
  torch.cuda.empty_cache()
  return dict(workload="configs[4] shard: 16 x 30 s resident (T' = 1501), forward + prefix beam search (beam %d), searches overlapped "
                       'with the next forward passes on CU-masked decoder streams' % beam,
              utterances_per_s=round(batches * 16 / dt, 1), ms_per_batch=round(dt / batches * 1e3, 3), batches=batches,
              protocol='3 untimed batches, then %d timed ones INCLUDING the drain of the last two searches (nothing overlaps them); '
                       'scripts/bench_decode.py times 40 batches the same way, so its drain weighs a third as much' % batches,
              oracle_check=dict(utterance=0, decoded_len=len(got[0]), beam_ids_equal=bool(got[0] == ref_ids[0]), passed=bool(got[0] == ref_ids[0])))


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
  stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
  # the filterbank is compiled into a device-side plan once per (sample rate, n_mels), like the reference computes
  # its mel basis once; the timed call is the per-batch work
  plan = torch.empty(_lib.load().st_melspec_plan_bytes() // 4, dtype=torch.float32, device=dev)
  _lib.call('st_melspec_plan_f32', P(basis), n_mels, 512, P(plan), plan.numel() * 4, stream)
  run = lambda: _lib.call('st_melspec_planned_f32', P(audio), P(s_off), batch, n, P(plan), n_mels, 512, 160, P(f_off),
                          batch * frames, P(out), P(ws), ws.numel() * 4, stream)
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


def cpu_baseline_torch(n_mels, frames, batch, budget_s=40.0):
  """The same training step with torch CPU ops (F.conv1d through oneDNN + native CTC + autograd, fp32), the fastest
  CPU formulation available here (TF1 cannot be installed): median of 3 steps after warm-up.  The intra-op thread count
  is probed (torch's default = physical cores, and half of it: on the 128-core hosts of the GPU boxes half was 1.6x
  faster; all 256 hardware threads were 100x slower).  The batch is the full 32 utterances unless the probe says
  three steps of it would not fit the time budget."""
  from tests import torch_ref as TR
  layers = WL.w2l_layers(n_mels)
  params = WL.xavier_params(layers, seed=42, bias_range=0.0, dtype=np.float32)
  x, sl, labels = WL.make_batch([frames] * batch, n_mels, seed=0)
  x = x.astype(np.float32)
  trainer = TR.TorchCpuTrainer(params, layers, lr=1e-4)
  default_threads = torch.get_num_threads()
  probe, best = min(2, batch), None
  for threads in (default_threads, max(1, default_threads // 2)):
    torch.set_num_threads(threads)
    trainer.step(x[:1, :201], [201], [labels[0][:20]])                # warm-up (thread pool, primitive cache)
    t0 = time.time()
    trainer.step(x[:probe], sl[:probe], labels[:probe])
    per_utt = (time.time() - t0) / probe
    if best is None or per_utt < best[1]:
      best = (threads, per_utt)
  torch.set_num_threads(best[0])
  utts = batch
  while utts > probe and 4 * utts * best[1] > budget_s:
    utts //= 2
  trainer.step(x[:utts], sl[:utts], labels[:utts])                    # warm-up at the timed shape
  times = []
  for _ in range(3):
    t0 = time.time()
    trainer.step(x[:utts], sl[:utts], labels[:utts])
    times.append(time.time() - t0)
  torch.set_num_threads(default_threads)
  med = sorted(times)[1]
  return dict(value=round(utts / med, 3), unit='utterances/s', cores=best[0], host_threads=os.cpu_count(), kind='port',
              sample='batch of {} x 10 s utterances, median of 3 full training steps (forward + CTC + backward + clip + '
                     'TF-Adam) of tests/torch_ref.py: torch {} CPU ops (oneDNN conv1d, native ctc_loss), fp32, {} intra-op '
                     'threads; CPU restatement of the reference path (TF1 not installable)'.format(utts, torch.__version__, best[0]),
              step_seconds=[round(t, 3) for t in times])


def parity_on_bench_inputs(eng, x, seq_lens, labels, rows=(0, -1)):
  """|CTC loss - oracle| and max|logit - oracle| on the bench inputs (BASELINE metric): the device
  runs forward + CTC on the full batch with the initial weights, the float64 oracle on a 2-utterance slice of the
  same batch (utterances are independent: nothing is masked, so a row's logits depend on that row only)."""
  from oracle import w2l_oracle as O
  rows = [r % len(seq_lens) for r in rows]
  eng.load_batch(x, seq_lens)
  eng.set_labels(labels)
  eng.forward()
  eng.ctc_loss_grad(1.0 / len(seq_lens))
  torch.cuda.synchronize()
  eng.check_ctc_status()
  got = eng.logits_time_major().cpu().numpy()[:, rows]
  dec, _ = eng.greedy_decode()
  params64 = [(F.astype(np.float64), b.astype(np.float64)) for F, b in eng.get_weights()]
  layers = [(l.width, l.stride, l.cin, l.cout, l.relu) for l in eng.layers]
  ref = O.wav2letter_forward(np.asarray(x, np.float32)[rows].astype(np.float64), params64, layers)
  loss_ref, _ = O.ctc_loss_and_grad(ref, [labels[r] for r in rows], np.asarray(seq_lens)[rows] // 2)
  ref_dec, _ = O.ctc_greedy_decode(ref, np.asarray(seq_lens)[rows] // 2)
  loss_f32 = eng.loss.cpu().numpy()[rows].astype(np.float64)      # the fp32 value tf.nn.ctc_loss would return
  loss_dev = eng.losses_precise()[rows]                            # (hi, lo) float pair of the kernel: -log p as it knows it
  out = dict(rows=rows, max_logit_err=float(np.max(np.abs(got - ref))),
             max_logit_err_rel=float(np.max(np.abs(got - ref)) / np.max(np.abs(ref))), max_abs_logit=float(np.max(np.abs(ref))),
             ctc_loss_delta=float(np.max(np.abs(loss_dev - loss_ref))),
             ctc_loss_delta_rel=float(np.max(np.abs(loss_dev - loss_ref) / np.abs(loss_ref))),
             ctc_loss_delta_fp32_output=float(np.max(np.abs(loss_f32 - loss_ref))),
             ctc_loss_oracle=[round(float(v), 4) for v in loss_ref],
             greedy_strings_equal=bool([dec[r] for r in rows] == ref_dec),
             oracle='oracle/w2l_oracle.py float64 on utterances {} of the bench batch, initial weights'.format(rows))
  # What is ASSERTED (bench.py exits non-zero otherwise): north_star's "within 1e-4": 1e-4 ABSOLUTE for the logits (O(1)
  # numbers) and 1e-4 ABSOLUTE for the per-utterance CTC loss -- an unnormalised sum over ~500 frames, O(1000), where one fp32
  # ulp is 1.2e-4: the kernel therefore returns -log p as a (hi, lo) float pair (st_ctc_loss_grad_hilo_f32); hi alone, the fp32
  # number TF's op returns, is reported as `ctc_loss_delta_fp32_output` and cannot meet an absolute 1e-4 at this magnitude.
  # The logits of fresh weights are small numbers (|logit| < 0.1): 1e-4 absolute alone would let a 1000x regression pass, so the
  # error is also bounded relative to the largest logit -- measured 2e-6, asserted at 2e-5.
  out['asserted'] = dict(max_logit_err='< 1e-4 absolute', max_logit_err_rel='< 2e-5 of the largest |logit|', ctc_loss_delta='< 1e-4 absolute (hi + lo of the kernel\'s loss pair)',
                         ctc_loss_delta_rel='< 1e-4 relative', greedy_strings_equal=True,
                         ctc_loss_delta_fp32_output='reported, not asserted (one fp32 ulp of the loss is %.1e)'
                                                    % float(np.spacing(np.float32(np.max(np.abs(loss_ref))))))
  out['passed'] = bool(out['max_logit_err'] < 1e-4 and out['max_logit_err_rel'] < 2e-5 and out['ctc_loss_delta'] < 1e-4 and out['ctc_loss_delta_rel'] < 1e-4 and
                       out['greedy_strings_equal'])
  return out


def attach_pmc_profiles(roofline):
  """`traffic` (fabric bytes per launch of the dominant kernel) and `mfma_busy_pmc` cannot be read from inside the run:
  they come from rocprofv3 --pmc passes over this same command (scripts/gpu_traffic.sh, scripts/gpu_mfma_util.sh), kept
  under profiles/ and stamped with the digest of the sources they were collected for.  A file collected for OTHER
  sources is not quoted: `traffic` stays null and the line says so (and the mismatch is shouted on stderr)."""
  from speecht_amd.build import source_digest
  digest = source_digest()
  for fname, fill in (('traffic.json', 'traffic'), ('mfma_util.json', 'mfma_busy_pmc')):
    path = os.path.join(ROOT, 'profiles', fname)
    if not os.path.exists(path):
      continue
    data = json.load(open(path))
    if data.get('source_digest') != digest:
      roofline[fill + '_stale'] = ('profiles/%s was collected for sources %s, this build is %s: not quoted (re-run scripts/%s)'
                                   % (fname, str(data.get('source_digest'))[:12], digest[:12],
                                      'gpu_traffic.sh' if fill == 'traffic' else 'gpu_mfma_util.sh'))
      print('bench.py: WARNING ' + roofline[fill + '_stale'], file=sys.stderr)
      continue
    if fill == 'traffic':
      k = data.get('by_kernel', {}).get(roofline['kernel'], {})
      roofline['traffic'] = k.get('bytes_per_launch')
      roofline['traffic_over_algorithmic'] = (round(k['bytes_per_launch'] / (roofline['algorithmic_mb_per_launch'] * 1e6), 2)
                                              if k.get('bytes_per_launch') else None)
      roofline['step_fabric_bytes'] = data.get('step_bytes')
      roofline['traffic_source'] = ('profiles/traffic.json (sources %s): FETCH_SIZE / WRITE_SIZE PMC passes of rocprofv3 over this '
                                    'command, read side doubled per MI355X_MICROARCH.md (scripts/gpu_traffic.sh)' % digest[:12])
    else:
      m = re.match(r'gemm_nn<(\d+),(\d+),(\d+),(\d+),(fast-bt|fast|clamped)> epi=(\d)', roofline['kernel'])
      key = ('gemm_nn_kernel<%s, %s, %s, %s, %s, %s, %s>' % (m.group(1), m.group(2), m.group(3), m.group(4), m.group(6),
                                                             'false' if m.group(5) == 'clamped' else 'true',
                                                             'true' if m.group(5) == 'fast-bt' else 'false') if m else 'gemm_tn_kernel<128, 2, 2>')
      roofline['mfma_busy_pmc'] = data.get(key, {}).get('mfma_busy_frac_at_2p4ghz')


class c_stdout_to_stderr:
  """RCCL prints a version banner to the C-level stdout when a communicator is created; the driver reads ONE JSON line from
  stdout.  Inside this context file descriptor 1 is stderr (and C stdio is flushed before it comes back)."""

  def __enter__(self):
    import ctypes
    self._libc = ctypes.CDLL(None)
    sys.stdout.flush()
    self._libc.fflush(None)
    self._saved = os.dup(1)
    os.dup2(2, 1)
    return self

  def __exit__(self, *exc):
    sys.stdout.flush()
    self._libc.fflush(None)
    os.dup2(self._saved, 1)
    os.close(self._saved)
    return False


def comm_probe_world1(eng, feed, lr, global_batch, steps, ahead):
  """What the data-parallel plumbing costs a step on ONE GPU: the same steps with the bucketed exchange forced through the
  library's RCCL communicator at world size 1 (per-layer hooks, the collective stream, its events, ncclAllReduce launches
  beside back-prop) against the plain steps.  It cannot show what the transfers of N > 1 cost the GEMMs (a 1-rank all-reduce
  moves nothing over xGMI); it bounds everything else."""
  try:
    with c_stdout_to_stderr():
      probe = GradientAllReducer(eng.reduce_buffer, eng.reduce_ranges, force=True, transport='rccl')
  except Exception as e:      # noqa: BLE001
    return dict(error=repr(e))
  res = {}
  for name, red in (('compute_only_ms', None), ('with_collective_ms', probe), ('compute_only_ms_again', None)):
    for _ in range(3):
      train_step(eng, feed, red, lr, global_batch)
    torch.cuda.synchronize()
    marks = [torch.cuda.Event() for _ in range(steps)]
    t0 = time.perf_counter()
    for k in range(steps):
      if ahead and k >= ahead:
        marks[k - ahead].synchronize()
      train_step(eng, feed, red, lr, global_batch)
      marks[k].record()
    torch.cuda.synchronize()
    res[name] = round((time.perf_counter() - t0) / steps * 1e3, 3)
  res.update(library_comm=probe.comm.count(), buckets=len(probe.buckets),
             note='world size 1: hooks + collective stream + ncclAllReduce launches, no xGMI traffic')
  with c_stdout_to_stderr():
    probe.comm.close()
  return res


def free_port():
  with socket.socket() as sk:
    sk.bind(('127.0.0.1', 0))
    return sk.getsockname()[1]


def self_launch(args):
  """``python bench.py --gpus N`` with no torch.distributed environment: start N ranks of this script (one per
  GPU, RCCL) under torch.distributed.run on this node and hand its output through."""
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
         '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + [
             a for a in sys.argv[1:] if a != '--self-launch']
  return subprocess.call(cmd, env=env)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--warmup', type=int, default=5)
  ap.add_argument('--batch', type=int, default=32, help='utterances per GPU')
  ap.add_argument('--seconds', type=float, default=10.0)
  ap.add_argument('--mels', type=int, default=80)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-alt', action='store_true', help='skip the bf16 side measurement (configs[3] arithmetic on this GPU)')
  ap.add_argument('--alt-bf16x6', action='store_true', help='(accepted for old command lines: the bf16x6 side measurement is part of the default run)')
  ap.add_argument('--no-alt-bf16x6', action='store_true', help='skip the bf16x6 side measurement (its launches share kernel symbols with the '
                  'fp32 step; the profiled runs of scripts/gpu_profile_round.sh use --no-alt / --steps-only anyway)')
  ap.add_argument('--steps-only', action='store_true', help='warm-up + timed steps and a minimal line: no parity pass, roofline, mel, '
                  'alt modes or CPU baseline (the command the PMC passes of scripts/gpu_profile_round.sh run: every launch is a step\'s)')
  ap.add_argument('--self-launch', action='store_true', help='go through torch.distributed.run even for --gpus 1 (self-test of the launcher path)')
  ap.add_argument('--force-allreduce', action='store_true', help='run the RCCL all-reduce path even on 1 rank (self-test)')
  ap.add_argument('--conv-mode', choices=('fp32', 'bf16x6', 'bf16'), default=None,
                  help='arithmetic of the timed loop (default fp32 = BASELINE configs[1]; bf16 = configs[3] arithmetic)')
  ap.add_argument('--allreduce', choices=('torch', 'rccl'), default=None,
                  help='gradient exchange transport: the library\'s own RCCL communicator behind the C ABI (st_allreduce_buckets_f32; '
                       'default for --gpus > 1, falls back to torch.distributed if it cannot be set up) or torch.distributed')
  ap.add_argument('--tune', action='append', default=[], help='name=value override of a library policy (st_set_tuning; experiments only)')
  args = ap.parse_args()
  for kv in args.tune:
    from speecht_amd._lib import set_tuning
    set_tuning(kv.split('=')[0], int(kv.split('=')[1]))

  if (args.gpus > 1 or args.self_launch) and 'WORLD_SIZE' not in os.environ:
    sys.exit(self_launch(args))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world > 1 or args.force_allreduce or 'WORLD_SIZE' in os.environ:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29577')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    # test knobs (not used by the driver): ST_SHARE_GPU=1 maps every rank to cuda:0 and ST_DIST_BACKEND=gloo swaps
    # the transport, so that the world_size > 1 control flow can be exercised on a single-GPU box
    backend = os.environ.get('ST_DIST_BACKEND', 'nccl')
    if os.environ.get('ST_SHARE_GPU'):
      local_rank = 0
    torch.cuda.set_device(local_rank)
    with c_stdout_to_stderr():                     # (RCCL's version banner goes to the C stdout of rank 0)
      if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        dist.all_reduce(torch.zeros(1, device=torch.device('cuda', local_rank)))      # the communicator exists from here on
        torch.cuda.synchronize()
      else:
        dist.init_process_group(backend, rank=rank, world_size=world)
  assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node {} for --gpus {}'.format(args.gpus, args.gpus)
  dev = torch.device('cuda', local_rank)
  torch.cuda.set_device(dev)

  frames = 1 + int(args.seconds * 16000) // 160
  layers = WL.w2l_layers(args.mels)
  eng = Wav2LetterEngine(layers, device=dev, conv_mode=args.conv_mode)
  x, seq_lens, labels = WL.make_batch([frames] * args.batch, args.mels, seed=100 + rank)
  # parity on the bench inputs twice: with NON-ZERO biases U(-0.05, 0.05) first (zero biases hide F7 -- the padded
  # region then stays zero through every layer), then with the zero-bias fresh-training weights the timed steps start from
  parity_b = None
  if rank == 0 and not args.steps_only:
    eng.set_weights(WL.xavier_params(layers, seed=42, bias_range=0.05, dtype=np.float32))
    parity_b = parity_on_bench_inputs(eng, x, seq_lens, labels)
  eng.set_weights(WL.xavier_params(layers, seed=42, bias_range=0.0, dtype=np.float32))   # same replica everywhere
  parity = parity_on_bench_inputs(eng, x, seq_lens, labels) if (rank == 0 and not args.steps_only) else None
  for pr in (parity, parity_b):
    if pr is not None and eng.conv_mode == 'bf16':
      pr['asserted'], pr['passed'] = None, None        # bf16 storage: parity is tests/test_gpu_bf16.py's (bf16 ulps), not 1e-4
    if pr is not None and pr['passed'] is False:
      print('bench.py: PARITY FAILED against the oracle on the bench inputs, no result line is printed: ' + json.dumps(pr),
            file=sys.stderr)
      sys.exit(3)
  feed = HostFeed(eng, x, seq_lens, labels)
  reducer, transport_note = None, None
  if world > 1 or args.force_allreduce:
    # Default transport for N > 1: the library's own RCCL communicator (the C ABI's st_comm_* / st_allreduce_buckets_f32) -- but
    # only if it really spans the job: its rank count (ncclCommCount) must equal the world size on every rank, else every rank
    # falls back to torch.distributed together and the line says so.
    # (ranks that share one GPU -- the ST_SHARE_GPU test knob -- cannot form an RCCL communicator: torch.distributed there)
    want = args.allreduce or os.environ.get('ST_ALLREDUCE') or ('rccl' if (world > 1 and not os.environ.get('ST_SHARE_GPU')) else 'torch')
    if want == 'rccl':
      ok = 1
      try:
        with c_stdout_to_stderr():
          reducer = GradientAllReducer(eng.reduce_buffer, eng.reduce_ranges, force=args.force_allreduce, transport='rccl')
        ok = int(reducer.comm is not None and reducer.comm.count() == world)
      except Exception as e:      # noqa: BLE001 -- whatever the set-up raises, the exchange must still happen somehow
        ok, transport_note = 0, 'library communicator failed: %r' % (e,)
      if dist.is_initialized() and world > 1:
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = int(flag[0])
      if not ok:
        transport_note = transport_note or 'library communicator does not span the job (count != world) on some rank'
        if args.allreduce == 'rccl':
          print('bench.py: --allreduce rccl asked for, but ' + transport_note, file=sys.stderr)
          sys.exit(5)
        reducer = None
    if reducer is None:
      reducer = GradientAllReducer(eng.reduce_buffer, eng.reduce_ranges, force=args.force_allreduce, transport='torch')
  global_batch = args.batch * world
  lr = 1e-4

  def sync():
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  for _ in range(args.warmup):
    train_step(eng, feed, reducer, lr, global_batch)
  sync()
  marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # per-step times for the median
  t0 = time.perf_counter()
  marks[0].record()
  host_ms = []
  # the host keeps at most two steps in flight (measured: one leaves the GPU waiting for launches between steps, +0.13 ms;
  # an unbounded burst after the sync runs the first three steps 0.3-2 ms slow each)
  ahead = int(os.environ.get('ST_BENCH_AHEAD', '2'))
  for k in range(args.steps):
    h0 = time.perf_counter()
    if ahead and k >= ahead:
      marks[k + 1 - ahead].synchronize()
    train_step(eng, feed, reducer, lr, global_batch)
    marks[k + 1].record()
    host_ms.append((time.perf_counter() - h0) * 1e3)
  sync()
  elapsed = time.perf_counter() - t0
  step_ms = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
  median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
  rank_ms = [elapsed / args.steps * 1e3]
  eng.check_ctc_status()
  loss = float(eng.loss.mean())
  replicas_identical, comm = None, None
  if world > 1:
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    every = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(every, t)
    rank_ms = [float(e[0]) / args.steps * 1e3 for e in every]
    elapsed = max(float(e[0]) for e in every)
    # data-parallel invariant: every rank applied the same averaged gradient to the same weights, so the
    # replicas must still be bit-identical (checked outside the timed region: two small all-reduces)
    p64 = eng.params.double()
    digest = torch.stack([p64.sum(), (p64 * p64).sum(), p64[::4097].abs().sum()])
    lo, hi = digest.clone(), digest.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    replicas_identical = bool(torch.equal(lo, hi))
    del p64
    # also outside the timed region (and after the digest: un-reduced steps let the replicas drift): the same
    # steps without the exchange, and the exchange alone, to see how much of it the back-prop kernels hide
    reps = max(3, min(10, args.steps))
    sync()
    t1 = time.perf_counter()
    for _ in range(reps):
      train_step(eng, feed, None, lr, global_batch)
    sync()
    compute_ms = (time.perf_counter() - t1) / reps * 1e3
    t1 = time.perf_counter()
    for _ in range(reps):
      for i in reversed(range(len(eng.layers))):
        reducer.on_layer_done(i)
      reducer.finish()
    sync()
    allreduce_ms = (time.perf_counter() - t1) / reps * 1e3
    # each bucket's all-reduce on its own (events on the collective's stream; max over ranks), in launch order
    bucket_ms = []
    for lo, bs, be in reducer.buckets:
      sync()
      t1 = time.perf_counter()
      for _ in range(reps):
        reducer.on_layer_done(lo)
        reducer.finish()
      torch.cuda.synchronize()
      bucket_ms.append((time.perf_counter() - t1) / reps * 1e3)
    c = torch.tensor([compute_ms, allreduce_ms] + bucket_ms, dtype=torch.float64, device=dev)
    dist.all_reduce(c, op=dist.ReduceOp.MAX)
    comm = dict(compute_only_ms_per_step=round(float(c[0]), 3), allreduce_alone_ms=round(float(c[1]), 3),
                exposed_comm_ms=round(max(0.0, elapsed / args.steps * 1e3 - float(c[0])), 3),
                gradient_mb=round(eng.n_flat * 4 / 1e6, 1), buckets=len(reducer.buckets),
                per_bucket=[dict(first_layer=lo, mb=round((be - bs) * 4 / 1e6, 2), allreduce_ms=round(float(c[2 + k]), 3),
                                 bus_gbs=round(2.0 * (world - 1) / world * (be - bs) * 4 / (float(c[2 + k]) * 1e-3) / 1e9, 1))
                            for k, (lo, bs, be) in enumerate(reducer.buckets)])
  ranks_info = None
  if reducer is not None:
    # who is really exchanging: torch.distributed's view and -- with the library transport -- the RCCL communicator's own count
    ranks_info = dict(torch_distributed=dist.get_world_size() if dist.is_initialized() else 1,
                      backend=dist.get_backend() if dist.is_initialized() else None,
                      library_comm=reducer.comm.count() if reducer.comm is not None else None, transport=reducer.transport,
                      library_comm_spans_job=(reducer.comm.count() == world) if reducer.comm is not None else None,
                      fallback=transport_note)
    if world > 1 and not replicas_identical:
      if rank == 0:
        print('bench.py: the replicas are NOT bit-identical after %d data-parallel steps (rccl_ranks %s): the exchange is '
              'broken, no result line is printed' % (args.warmup + args.steps, json.dumps(ranks_info)), file=sys.stderr)
      dist.destroy_process_group()
      sys.exit(4)

  if rank == 0 and args.steps_only:
    print(json.dumps({'metric': 'utterances/sec (training step, 10 s@16 kHz, batch 32)', 'value': round(global_batch / (elapsed / args.steps), 2),
                      'unit': 'utterances/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                      'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'steps_only': True}))
  elif rank == 0:
    ms = elapsed / args.steps * 1e3
    fl = conv_flops(eng, args.batch)
    step_gflop = (3.0 * sum(fl) - fl[0]) / 1e9     # fwd + bwd-data (not for L0) + bwd-filter
    out = {
        'metric': 'utterances/sec (training step, 10 s@16 kHz, batch 32)',
        'value': round(global_batch / (elapsed / args.steps), 2), 'unit': 'utterances/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3),
        'ms_per_step_median': round(median_ms, 3),
        'ms_per_step_quantiles': [round(step_ms[int(q * (len(step_ms) - 1))], 3) for q in (0.0, 0.25, 0.5, 0.75, 1.0)],
        'ms_each_step': [round(marks[k].elapsed_time(marks[k + 1]), 3) for k in range(args.steps)],
        'host_ms_each_step': [round(v, 2) for v in host_ms],
        'host_enqueue_ms_per_step': [round(float(np.median(host_ms)), 3), round(float(np.max(host_ms)), 3)],   # median, max (the host runs ahead of the GPU)
        'step_includes': 'pinned H2D of the feature batch (copy stream, double-buffered) + label upload + forward + '
                         'CTC loss/grad + back-prop' + (' + gradient all-reduce' if reducer else '') + ' + clip + Adam',
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
        'ctc_loss_delta': parity['ctc_loss_delta'], 'max_logit_err': parity['max_logit_err'],
        'ctc_loss_delta_kind': 'absolute, ASSERTED < 1e-4: |hi + lo - oracle| of the kernel\'s (hi, lo) loss pair; the fp32 hi part alone '
                               '(what TF returns) is ctc_loss_delta_fp32_output in `parity`',
        'ctc_loss_delta_rel': parity['ctc_loss_delta_rel'], 'parity': parity, 'parity_nonzero_bias': parity_b,
        'replicas_identical': replicas_identical, 'rccl_ranks': ranks_info,
        'per_rank_ms_per_step': [round(v, 3) for v in rank_ms], 'comm': comm,
        'step_tflops_algorithmic': round(step_gflop / ms, 2),
        'step_tflops_note': ('W-tap FLOPs of the reference formulation (SURVEY 8(d): 71.4 GFLOP per utterance) / step time; '
                             'nine of the eleven layers run in the frequency domain with 3-10x fewer multiplications, so this is an '
                             'equivalent rate, not a hardware rate (the hardware rate is roofline.achieved)')
                            if (eng.fft and eng.fft_conv) else None,
    }
    # The side measurements come BEFORE the roofline passes: those launch through hipExtLaunchKernel with start / stop events,
    # which switches the process's HSA queues into profiling mode for good -- every later dispatch carries a completion
    # signal.  The 7 ms fp32 step does not notice; steps of many short launches do (measured round 4, scripts/exp/
    # alt_timing_probe.py: bf16 2.66 -> 2.86 ms, bf16x6 6.46 -> 7.11 ms when timed after those passes).
    if world == 1 and reducer is None and eng.conv_mode == 'fp32':
      out['comm_probe_world1'] = comm_probe_world1(eng, feed, lr, global_batch, args.steps, ahead)
    if world == 1 and reducer is None and not args.no_alt:
      out['comm_model_8gpu'] = comm_model(eng, feed, lr, global_batch, ms)
      out['host_cost'] = host_cost(eng, feed, lr, global_batch, args.steps, ahead)
    if world == 1 and eng.conv_mode == 'fp32' and not args.no_alt:
      # Side measurements on the same box and inputs, NOT the headline:
      #  * bf16x6 (experimental): fp32 operands split exactly into 3 bf16 pieces, 6 cross terms on the bf16
      #    matrix pipe, fp32 accumulate; passes the same parity tests as the fp32 path.
      #  * bf16: BASELINE configs[3]'s arithmetic ("bf16 activations / fp32 CTC") on one GPU -- reduced
      #    precision by design, parity against the oracle's bf16 storage model (tests/test_gpu_bf16.py).
      alts = [('alt_bf16x6', 'bf16x6', 'f32 tensors; the 2000 x 2000 layer\'s three products with every operand split exactly into 3 bf16 pieces, '
               '6 bf16 MFMA terms, f32 accumulate (the other layers as in the headline)',
               'opt-in (ST_CONV_MODE=bf16x6): fp32-level accuracy (same parity tests and tolerances as the headline path) on the '
               'bf16 matrix pipe; reported beside the headline, which stays on the fp32 MFMA instruction'),
              ('alt_bf16', 'bf16', 'bf16 activations + activation gradients, f32 masters/accumulate/logits/CTC/Adam',
               'configs[3] arithmetic on 1 GPU (ST_CONV_MODE=bf16); reduced precision, not the headline value')]
      for key, mode, dtype, note in alts:
        if mode == 'bf16x6' and args.no_alt_bf16x6:
          continue
        alt = Wav2LetterEngine(layers, device=dev, conv_mode=mode)
        alt.params.copy_(eng.params)
        alt.mark_weights_changed()
        alt_feed = HostFeed(alt, x, seq_lens, labels)
        for _ in range(args.warmup):
          train_step(alt, alt_feed, None, lr, global_batch)
        torch.cuda.synchronize()
        # timed like the headline loop: at most `ahead` steps in flight (an unbounded burst after a sync runs its first
        # steps slow -- it made this 3.1 ms step read 3.5 ms)
        alt_marks = [torch.cuda.Event() for _ in range(args.steps)]
        t1 = time.perf_counter()
        for k in range(args.steps):
          if ahead and k >= ahead:
            alt_marks[k - ahead].synchronize()
          train_step(alt, alt_feed, None, lr, global_batch)
          alt_marks[k].record()
        torch.cuda.synchronize()
        alt_ms = (time.perf_counter() - t1) / args.steps * 1e3
        out[key] = {'value': round(args.batch / alt_ms * 1e3, 2), 'unit': 'utterances/s', 'ms_per_step': round(alt_ms, 3),
                    'step_tflops_algorithmic': round(step_gflop / alt_ms, 2),
                    'final_avg_loss': round(float(alt.loss.mean()), 4), 'dtype': dtype, 'note': note}
        if mode == 'bf16':
          # the comm model first (plain events), the roofline pass last: its timed launches switch the queues to profiling mode
          out[key]['host_cost'] = host_cost(alt, alt_feed, lr, global_batch, args.steps, ahead)
          out[key]['comm_probe_world1'] = comm_probe_world1(alt, alt_feed, lr, global_batch, args.steps, ahead)
          out[key]['comm_model_8gpu'] = comm_model(alt, alt_feed, lr, global_batch, alt_ms)
          out[key]['roofline'] = roofline_bf16_in_step(alt, lambda: train_step(alt, alt_feed, None, lr, global_batch), alt_ms)
        del alt
        torch.cuda.empty_cache()
    # (rank 0 alone runs these profiled steps: without the exchange when there are other ranks)
    step_fn = lambda: train_step(eng, feed, reducer if world == 1 else None, lr, global_batch)
    out['roofline'], step_hw = measure_dominant_kernel(eng, args.batch, step_fn, ms)
    if step_hw:
      out.update(step_hw)
    if out['roofline'] and eng.conv_mode != 'bf16':
      attach_pmc_profiles(out['roofline'])
    if world == 1:
      out['mel_features'] = measure_mel(dev, args.batch, args.seconds, args.mels)
      if not args.no_alt:
        out['mel_features_batch512'] = measure_mel(dev, 512, args.seconds, args.mels)    # (a preprocessing job's batch: launch costs amortised)
    if world == 1 and not args.no_alt and args.mels == 80:
      # the other single-GPU configurations of BASELINE.json, a few seconds each, each with an oracle check of its own
      out['configs2_inference'] = config2_rate(dev, layers)
      out['configs4_decode'] = config4_rate(dev, layers)
    if world == 1 and not args.no_cpu_baseline:
      # two CPU restatements of the reference path on this box's host cores; the faster one is `cpu_baseline`
      out['cpu_baseline'] = cpu_baseline_torch(args.mels, frames, args.batch)
      out['cpu_baseline_numpy_oracle'] = cpu_baseline(args.mels, frames)
    print(json.dumps(out))
  if dist.is_initialized():
    if world > 1:
      # rank 0 measured on its own for a while (roofline passes, side measurements): the others wait for it here instead of
      # tearing the communicator down under it
      dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
