"""Streams of the engine: one set of ROLE streams per process (which hardware queue a HIP stream lands on decides what
overlaps) and the CU-masked decoder streams of the pipelined beam search."""
import os

import torch


def decoder_streams(device, decoders=2):
  """(compute stream, [decoder streams]) for overlapping a one-wave-per-utterance decoder with the NEXT batches' forward passes.

  fp32 MFMAs execute on the VALU datapath: a VALU / LDS chain that shares its SIMD with the waves of an fp32 GEMM gets an
  issue slot every ~25 cycles instead of every ~4 (measured round 4: the CTC recursion under a GEMM 122 -> 560 us, the beam
  search beside the next forward pass 3.9 -> 6.5 ms per batch).  So the two run on DISJOINT compute units: streams created
  with hipExtStreamCreateWithCUMask, the decoders on 16 CUs (mask bits 0..15 -- observed on MI355X: two CUs of every XCD; any
  other layout tried costs the GEMMs 10-70 %), the forward pass on the other 240 (+11 % on the forward pass alone, 3.6 -> 4.0
  ms at configs[4]).  A search is one wavefront per utterance -- 16 of the decoder CUs' 64 SIMDs at configs[4] -- and takes a
  little longer than the forward pass: with ONE decoder stream the search sets the pace (4.18 ms per batch against 7.4 serial),
  with two the searches of consecutive batches run side by side on the same 16 CUs and the forward pass does (4.02 ms).
  Placement is a matter of speed only.  Falls back to plain streams where the runtime lacks the call."""
  import ctypes as C
  import glob
  key = (str(device), int(decoders))
  if key in _DECODER_STREAMS:
    return _DECODER_STREAMS[key]
  made = None
  if os.environ.get('ST_DECODER_CU_MASK', '1') != '0':
    try:
      hip = C.CDLL(glob.glob(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64*'))[0])
      made = []
      with torch.cuda.device(device):
        for words in [[0xffff0000] + [0xffffffff] * 7] + [[0x0000ffff] + [0] * 7] * int(decoders):
          arr = (C.c_uint32 * 8)(*words)
          handle = C.c_void_p()
          if hip.hipExtStreamCreateWithCUMask(C.byref(handle), 8, arr) != 0 or not handle.value:
            raise OSError('hipExtStreamCreateWithCUMask failed')
          made.append(torch.cuda.ExternalStream(handle.value, device=device))
    except (OSError, IndexError, AttributeError):
      made = None
  if made is None:
    made = [torch.cuda.Stream(device) for _ in range(1 + int(decoders))]
  _DECODER_STREAMS[key] = (made[0], made[1:])
  return _DECODER_STREAMS[key]


def decoder_stream_pair(device):
  """(compute stream, decoder stream): `decoder_streams` with one decoder."""
  compute, decoders = decoder_streams(device, 1)
  return compute, decoders[0]


_DECODER_STREAMS = {}
_ROLE_STREAMS = {}


def role_stream(device, role):
  """The process-wide stream of a role ('h2d', 'upload', 'side', 'side2', 'collective') on a device.

  Which HARDWARE queue a HIP stream lands on is decided when it is first used, round-robin over the runtime's four queues,
  and two streams on one queue do not overlap: the queue serialises them.  With a stream per engine the same engine ran
  its step at different speeds depending on how many streams the process had used before it was built (measured round 4:
  the bf16x6 step 6.4 ms in a process of its own, 7.0 ms as the bench's side measurement behind an fp32 engine and a
  collective stream; the bf16 step 2.66 / 2.86 ms; GPU_MAX_HW_QUEUES=8 instead: fp32 step 7.0 -> 9.2 ms).  So the streams
  of all roles are created -- and used once, in a fixed order -- when the first of them is asked for, every engine of the
  process shares them (ordering between engines is by the events each engine records anyway), and the role -> queue map is
  the same in every process: compute stream q0, h2d q1, side q2, side2 q3, upload q0, collective q1 -- the two side streams
  on queues of their own (with side2 on the compute stream's queue, as creation order had it before: bf16 step 2.68 instead
  of 2.57 ms, fp32 and bf16x6 within 0.5 %; DESIGN 4.8 item 5 has the orders that were measured)."""
  key = str(device)
  pool = _ROLE_STREAMS.get(key)
  if pool is None:
    pool = {}
    with torch.cuda.device(device):
      torch.zeros(1, device=device)                       # the compute (current) stream has its queue first
      for r in ('h2d', 'side', 'side2', 'upload', 'collective'):
        pool[r] = torch.cuda.Stream(device)
        with torch.cuda.stream(pool[r]):
          torch.zeros(1, device=device)
      torch.cuda.synchronize(device)
    _ROLE_STREAMS[key] = pool
  if role not in pool:                                   # (a role the order left out)
    pool[role] = torch.cuda.Stream(device)
  return pool[role]
