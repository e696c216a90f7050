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
shapes = [('fwd', args.rows, 512, 4096), ('bwd', args.rows, 4096, 512), ('wgrad', 512, args.rows, 4096)]
for name, M, K, N in shapes:
  A = torch.randn(args.bins * M * K, device=dev)
  B = torch.randn(args.bins * K * N, device=dev)
  C = torch.empty(args.bins * M * N, device=dev)
  fn = lambda: call('st_gemm_nn_batched_f32', P(A), K, M * K, P(B), K * N, P(C), N, M * N, M, K, N, args.bins, None)
  ms = timeit(fn, 20)
  print('%-6s M=%d K=%d N=%d x%d: %.3f ms  %.1f TF/s' % (name, M, K, N, args.bins, ms, 2.0 * M * K * N * args.bins / ms / 1e9))
