#!/usr/bin/env python3
"""Experiment (round 4): does a CU-masked stream keep a one-wave-per-utterance decoder fast next to the fp32 GEMMs of the next
batch's forward?  fp32 MFMAs execute on the VALU datapath, so a VALU chain that shares its SIMD with a GEMM wave crawls (the CTC
recursion under a GEMM: 122 -> 560 us).  Streams created with hipExtStreamCreateWithCUMask: the decoder on `k` CUs, the forward
on the complement.  Prints forward / beam times alone, overlapped on plain streams, overlapped on masked streams."""
import ctypes, glob, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from speecht_amd.engine import Wav2LetterEngine
from speecht_amd import _lib
from tests import workloads as WL

hip = ctypes.CDLL(glob.glob(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64*'))[0])


def masked_stream(mask_words):
  arr = (ctypes.c_uint32 * len(mask_words))(*mask_words)
  s = ctypes.c_void_p()
  rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(mask_words), arr)
  assert rc == 0, rc
  return torch.cuda.ExternalStream(s.value)


dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
layers = WL.w2l_layers(80)
eng = Wav2LetterEngine(layers, device=dev)
eng.set_weights(WL.xavier_params(layers, seed=42, bias_range=0.05, dtype=np.float32))
B, frames, beam = 16, 3001, 16
x, seq_lens, _ = WL.make_batch([frames] * B, 80, seed=7)
eng.load_batch(x, seq_lens)
eng.forward(); torch.cuda.synchronize()
g = torch.Generator(device='cpu').manual_seed(11)
logits = (torch.randn(eng.X[-1].interior().shape, generator=g) * 3.0).to(dev)
lib = _lib.load()
ws = torch.zeros(lib.st_ctc_beam_ws(B, eng.t_out, beam) // 4 + 16, dtype=torch.int32, device=dev)
from speecht_amd._lib import Tensor3
lbuf = torch.zeros_like(eng.X[-1].buf)
ldesc = Tensor3(lbuf.data_ptr(), B, eng.X[-1].frames, eng.X[-1].channels, 0, eng.X[-1].t_pitch, eng.X[-1].c_pitch)
lbuf.view(B, eng.X[-1].t_pitch, eng.X[-1].c_pitch)[:, :eng.X[-1].frames, :29].copy_(logits)
eng._wait_uploads(); torch.cuda.synchronize()


def beam_on(stream):
  _lib.call('st_ctc_beam_search_decode', ctypes.byref(ldesc), eng._ptr(eng.ctc_lens), beam, eng._ptr(eng.dec_ids), eng.t_out,
            eng._ptr(eng.dec_lens), eng._ptr(eng.dec_score), eng._ptr(ws), ws.numel() * 4, ctypes.c_void_p(stream.cuda_stream))


def fwd_on(stream):
  saved, eng._stream = eng._stream, stream
  try:
    eng.forward()
  finally:
    eng._stream = saved


def run(fs, bs, reps=6, what='both'):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(reps):
    if what in ('both', 'fwd'):
      fwd_on(fs)
    if what in ('both', 'beam'):
      beam_on(bs)
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / reps * 1e3


plain_a, plain_b = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
for s in (plain_a, plain_b):
  run(s, s, 2)
print('plain streams: forward alone %.2f ms, beam alone %.2f ms, serial on one stream %.2f ms, two streams %.2f ms' % (
    run(plain_a, plain_b, what='fwd'), run(plain_a, plain_b, what='beam'), run(plain_a, plain_a), run(plain_a, plain_b)))
layouts = []
for k in (8, 16):
  layouts.append(('%d CUs, low bits' % k, list(range(k))))
for per in (1, 2, 4):
  layouts.append(('%d CUs, %d low bits of each 32-bit word' % (8 * per, per), [w * 32 + j for w in range(8) for j in range(per)]))
  layouts.append(('%d CUs, %d low bits of each 16-bit half word' % (16 * per, per), [w * 16 + j for w in range(16) for j in range(per)]))
layouts.append(('16 CUs, bits 8i', [8 * i for i in range(16)]))
layouts.append(('16 CUs, bits 0-1 of words + bits 16-17', [w * 32 + j for w in range(4) for j in (0, 1, 16, 17)]))
for name, dec_bits in layouts:
  words_dec = [0] * 8
  for b in dec_bits:
    words_dec[b // 32] |= 1 << (b % 32)
  words_fwd = [(~w) & 0xffffffff for w in words_dec]
  try:
    ds, fs = masked_stream(words_dec), masked_stream(words_fwd)
  except AssertionError as e:
    print('mask create failed', e); continue
  run(fs, ds, 2)
  print('decoder on %-48s forward alone (masked) %.2f ms, beam alone (masked) %.2f ms, overlapped %.2f ms per batch' % (
      name + ':', run(fs, ds, what='fwd'), run(fs, ds, what='beam'), run(fs, ds)))
# the forward on an all-ones mask: what a masked queue costs by itself
fs = masked_stream([0xffffffff] * 8)
run(fs, fs, 2)
print('all 256 bits set: forward alone %.2f ms' % run(fs, fs, what='fwd'))
