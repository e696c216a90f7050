#!/usr/bin/env python3
"""Times st_gemm_nn_batched_f32 on the three per-bin product shapes of the frequency-domain L8 (config 2: 48 bins,
256 rows, 2*256 x 2*2048 channels) for tile / split experiments (--tune name=value)."""
import argparse, ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speecht_amd._lib import call, set_tuning  # noqa: E402
from bench_conv import timeit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--tune', action='append', default=[])
ap.add_argument('--bins', type=int, default=48)
ap.add_argument('--rows', type=int, default=256)
args = ap.parse_args()
for kv in args.tune:
  k, v = kv.split('=')
  set_tuning(k, int(v))
dev = torch.device('cuda:0')
P = lambda t: ctypes.c_void_p(t.data_ptr())
from speecht_amd import _lib  # noqa: E402
WS_BYTES = _lib.load().st_gemm_nn_batched_ws_bytes()
WS = torch.zeros(WS_BYTES // 4, device=dev)        # stream-K scratch of the per-bin products (control words zero)
shapes = [('fwd', args.rows, 512, 4096), ('bwd', args.rows, 4096, 512)]
for name, M, K, N in shapes:
  A = torch.randn(args.bins * M * K, device=dev)
  B = torch.randn(args.bins * K * N, device=dev)
  C = torch.empty(args.bins * M * N, device=dev)
  fn = lambda: call('st_gemm_nn_batched_ws_f32', P(A), K, M * K, P(B), K * N, P(C), N, M * N, M, K, N, args.bins, P(WS), WS_BYTES, None)
  ms = timeit(fn, 20)
  print('%-6s M=%d K=%d N=%d x%d: %.3f ms  %.1f TF/s' % (name, M, K, N, args.bins, ms, 2.0 * M * K * N * args.bins / ms / 1e9))
# the 7-tap 250 -> 250 layers: 36 bins
for name, M, K, N, bins in [('fwd7', args.rows, 512, 512, 36), ('fwd0', args.rows, 384, 512, 45), ('bwd0-ish', args.rows, 512, 384 + 128, 45)]:
  A = torch.randn(bins * M * K, device=dev); B = torch.randn(bins * K * N, device=dev); C = torch.empty(bins * M * N, device=dev)
  fn = lambda: call('st_gemm_nn_batched_ws_f32', P(A), K, M * K, P(B), K * N, P(C), N, M * N, M, K, N, bins, P(WS), WS_BYTES, None)
  ms = timeit(fn, 20)
  print('%-6s M=%d K=%d N=%d x%d: %.3f ms  %.1f TF/s' % (name, M, K, N, bins, ms, 2.0 * M * K * N * bins / ms / 1e9))

# filter-gradient lag products on the TN kernel (no transposed spectra needed): out = A^T Z
import numpy as np
for name, M, K, N, bins in [('tn8', args.rows, 512, 4096, 48), ('tn7', args.rows, 512, 512, 36)]:
  A = torch.randn(bins * M * K, device=dev); Z = torch.randn(bins * M * N, device=dev); C = torch.empty(bins * K * N, device=dev)
  fn = lambda: call('st_gemm_tn_batched_f32', P(A), K, M * K, P(Z), N, M * N, P(C), K * N, M, K, N, bins, None)
  ms = timeit(fn, 20)
  ref = A.view(bins, M, K)[1].double().T @ Z.view(bins, M, N)[1].double()
  err = float((C.view(bins, K, N)[1].double() - ref).abs().max() / ref.abs().max())
  print('%-6s M=%d K=%d N=%d x%d: %.3f ms  %.1f TF/s  (check %.1e)' % (name, M, K, N, bins, ms, 2.0 * M * K * N * bins / ms / 1e9, err))

# where the time of the small per-bin products goes: fixed cost (K -> 0), the slope per k-tile, and the step at
# 8-bin boundaries (bins are dealt to the 8 XCDs)
if os.environ.get('SWEEP'):
  for bins in (32, 36, 40, 64):
    for K in (32, 128, 256, 512, 1024):
      M, N = args.rows, 512
      A = torch.randn(bins * M * K, device=dev); B = torch.randn(bins * K * N, device=dev); C = torch.empty(bins * M * N, device=dev)
      fn = lambda: call('st_gemm_nn_batched_ws_f32', P(A), K, M * K, P(B), K * N, P(C), N, M * N, M, K, N, bins, P(WS), WS_BYTES, None)
      ms = timeit(fn, 20)
      print('sweep  bins=%d M=%d K=%d N=%d: %.1f us  %.1f TF/s' % (bins, M, K, N, ms * 1e3, 2.0 * M * K * N * bins / ms / 1e9))
