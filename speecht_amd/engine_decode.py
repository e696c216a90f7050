"""Decoders of the engine (speech_model.py:101-115): greedy CTC decoding and the LM-free prefix beam search, synchronous and
with their outputs on the way to pinned host memory (`inference.transcribe` overlaps them with the next batches)."""
import ctypes

import torch

from . import _lib
from ._lib import Tensor3, call


class _PendingDecode:
  """Decoder outputs on their way to pinned host memory (``Wav2LetterEngine.greedy_decode_async``).  The engine
  alternates between two host slots: read a handle before issuing the second decode after it."""

  def __init__(self, slot, batch, t_out):
    self._slot, self._batch, self._t_out = slot, batch, t_out

  def result(self):
    ids_host, lens_host, event = self._slot
    event.synchronize()
    lens = lens_host[:self._batch].numpy()
    ids = ids_host[:self._batch * self._t_out].numpy().reshape(self._batch, self._t_out)
    return [ids[b, :lens[b]].tolist() for b in range(self._batch)]


class _PendingBeamDecode:
  """Prefix-beam-search outputs on their way to pinned host memory (``Wav2LetterEngine.beam_search_decode_async``):
  ``result()`` waits for that batch only and returns (list of id lists, log_prob [B, 1])."""

  def __init__(self, slot, batch, t_out):
    self._slot, self._batch, self._t_out = slot, batch, t_out
    self._generation = slot['generation']

  def result(self):
    s = self._slot
    if s['generation'] != self._generation:
      # the slots form a ring of (decoder streams + 1): this handle's pinned buffers and event now belong to a later batch
      raise RuntimeError('beam-search handle read too late: {} further beam_search_decode_async call(s) re-used its slot; read a '
                         'handle before issuing more than len(decode streams) further calls'.format(s['generation'] - self._generation))
    s['event'].synchronize()
    lens = s['lens_h'][:self._batch].numpy()
    ids = s['ids_h'][:self._batch * self._t_out].numpy().reshape(self._batch, self._t_out)
    return ([ids[b, :lens[b]].tolist() for b in range(self._batch)],
            s['score_h'][:self._batch].numpy().reshape(-1, 1).copy())


def beam_input_transform(name):
  """The `input_transform` code of st_ctc_beam_search_decode_ex: None / 'logits' -> 0, 'log10_softmax' -> 1 (the reference's
  decoder input, tf.log(tf.nn.softmax(logits) + 1e-8) / log(10), speech_model.py:102)."""
  if name in (None, 'logits', 0):
    return 0
  if name in ('log10_softmax', 1):
    return 1
  raise ValueError("input_transform must be None, 'logits' or 'log10_softmax', got {!r}".format(name))


def merge_repeated_labels(seq):
  """tf.nn.ctc_beam_search_decoder(merge_repeated=True) on an output prefix: consecutive equal labels collapse (TF's LabelSeq walk;
  it also collapses genuine double letters, which is why the reference passes False, speech_model.py:110)."""
  return [v for i, v in enumerate(seq) if i == 0 or v != seq[i - 1]]


class DecodeMixin:
  """The decoding entry points of `Wav2LetterEngine` (they read the logits X[-1] and the lengths the batch was loaded with)."""

  def greedy_decode(self, merge_repeated=True):
    """tf.nn.ctc_greedy_decoder (speech_model.py:113-115) -> (list of id lists, neg_sum_logits [B,1])."""
    self._wait_uploads()
    call('st_ctc_greedy_decode', self.X[-1].ref, self._ptr(self.ctc_lens), int(merge_repeated),
         self._ptr(self.dec_ids), self.t_out, self._ptr(self.dec_lens), self._ptr(self.dec_score), self.stream_ptr)
    lens = self.dec_lens.cpu().numpy()
    ids = self.dec_ids.view(-1, self.t_out).cpu().numpy()
    return [ids[b, :lens[b]].tolist() for b in range(len(lens))], self.dec_score.cpu().numpy().reshape(-1, 1)

  def greedy_decode_async(self, merge_repeated=True):
    """``greedy_decode`` without the host synchronisation: launches the decoder and the D2H copies of its
    outputs into pinned host buffers and returns a handle; ``handle.result()`` waits for that batch only.  Lets
    a caller enqueue the next batch's forward before it reads this batch's transcripts (inference.transcribe)."""
    self._wait_uploads()
    call('st_ctc_greedy_decode', self.X[-1].ref, self._ptr(self.ctc_lens), int(merge_repeated),
         self._ptr(self.dec_ids), self.t_out, self._ptr(self.dec_lens), self._ptr(self.dec_score), self.stream_ptr)
    B, n = self.dec_lens.numel(), self.dec_ids.numel()
    if not hasattr(self, '_dec_host'):
      self._dec_host, self._dec_turn = [None, None], 0
    self._dec_turn ^= 1
    slot = self._dec_host[self._dec_turn]
    if slot is None or slot[0].numel() < n or slot[1].numel() < B:
      if slot is not None:
        slot[2].synchronize()                                      # a copy into the old buffers may be in flight
      slot = [torch.empty(max(n, 1), dtype=torch.int32, pin_memory=True),
              torch.empty(max(B, 1), dtype=torch.int32, pin_memory=True), torch.cuda.Event()]
      self._dec_host[self._dec_turn] = slot
    stream = self._stream if self._stream is not None else torch.cuda.current_stream(self.device)
    with torch.cuda.stream(stream):
      slot[0][:n].copy_(self.dec_ids, non_blocking=True)
      slot[1][:B].copy_(self.dec_lens, non_blocking=True)
      slot[2].record(stream)
    return _PendingDecode(slot, B, self.t_out)

  def beam_search_decode(self, beam_width=16, input_transform=None, merge_repeated=False):
    """LM-free CTC prefix beam search, top path (stock tf.nn.ctc_beam_search_decoder semantics; the
    reference's own beam search needs its KenLM fork, speech_model.py:101-111)
    -> (list of id lists, log_prob [B,1]).  Beams up to 128 (the reference runs 100).  ``input_transform='log10_softmax'``
    searches on log10(softmax(logits) + 1e-8), the reference's decoder input (speech_model.py:102); ``merge_repeated``
    (reference: False, speech_model.py:110) collapses repeated labels of the returned prefix the way TF's decoder does."""
    lib = _lib.load()
    B = self.dec_lens.numel()
    need = lib.st_ctc_beam_ws(B, self.t_out, int(beam_width))
    ws = self._storage.view('beam_ws', need // 4 + 16, torch.int32)[0]
    self._wait_uploads()
    call('st_ctc_beam_search_decode_ex', self.X[-1].ref, self._ptr(self.ctc_lens), int(beam_width), beam_input_transform(input_transform),
         self._ptr(self.dec_ids), self.t_out, self._ptr(self.dec_lens), self._ptr(self.dec_score),
         self._ptr(ws), ws.numel() * 4, self.stream_ptr)
    lens = self.dec_lens.cpu().numpy()
    ids = self.dec_ids.view(-1, self.t_out).cpu().numpy()
    out = [ids[b, :lens[b]].tolist() for b in range(len(lens))]
    if merge_repeated:
      out = [merge_repeated_labels(seq) for seq in out]
    return out, self.dec_score.cpu().numpy().reshape(-1, 1)

  def beam_search_decode_async(self, beam_width=16, decode_stream=None, input_transform=None):
    """``beam_search_decode`` without the host synchronisation and OFF the compute stream: the logits and lengths of this
    batch are copied into a decoder slot, the search runs on ``decode_stream`` (default: a stream of the engine's own;
    `decoder_streams` gives CU-masked ones -- a list of streams is used in turn, consecutive batches' searches then run side by
    side) and its outputs go to pinned host memory; returns a handle whose
    ``result()`` waits for this batch only.  The caller enqueues the next batch's forward pass meanwhile -- the search is ONE
    wavefront per utterance (3.9 ms for 16 x 30 s, beam 16: as long as the forward pass) and leaves the chip to it."""
    lib = _lib.load()
    B, T = self.dec_lens.numel(), self.t_out
    xl = self.X[-1]
    need = lib.st_ctc_beam_ws(B, T, int(beam_width))
    streams = list(decode_stream) if isinstance(decode_stream, (list, tuple)) else [decode_stream]
    # one slot more than decoder streams: the forward pass fills a slot while every stream searches one
    if not hasattr(self, '_beam_slots') or len(self._beam_slots) != len(streams) + 1:
      for old in getattr(self, '_beam_slots', []):
        if old is not None:
          old['event'].synchronize()
      self._beam_slots, self._beam_turn = [None] * (len(streams) + 1), 0
    self._beam_turn += 1
    which = self._beam_turn % len(self._beam_slots)
    slot = self._beam_slots[which]
    decode_stream = streams[self._beam_turn % len(streams)]
    main = self._stream if self._stream is not None else torch.cuda.current_stream(self.device)
    if decode_stream is None:
      if getattr(self, '_decode_stream', None) is None:
        self._decode_stream = torch.cuda.Stream(self.device)
      decode_stream = self._decode_stream
    if slot is None or slot['logits'].numel() < xl.buf.numel() or slot['ids'].numel() < B * T or slot['ws'].numel() * 4 < need or \
        slot['lens'].numel() < B:
      if slot is not None:
        slot['event'].synchronize()                               # the old buffers may still be in use
      i32 = lambda n, **kw: torch.empty(max(n, 1), dtype=torch.int32, **kw)
      slot = dict(logits=torch.empty(xl.buf.numel(), dtype=torch.float32, device=self.device), lens=i32(B, device=self.device),
                  ids=i32(B * T, device=self.device), out_lens=i32(B, device=self.device),
                  score=torch.empty(max(B, 1), dtype=torch.float32, device=self.device), ws=i32(need // 4 + 16, device=self.device),
                  ids_h=i32(B * T, pin_memory=True), lens_h=i32(B, pin_memory=True),
                  score_h=torch.empty(max(B, 1), dtype=torch.float32, pin_memory=True), event=torch.cuda.Event(), generation=0)
      slot['event'].record(decode_stream)
      self._beam_slots[which] = slot
    slot['generation'] += 1                                       # handles of the batch that last used this slot are stale from here on
    self._wait_uploads()
    main.wait_event(slot['event'])                               # the search that last read this slot is through
    with torch.cuda.stream(main):
      slot['logits'][:xl.buf.numel()].copy_(xl.buf, non_blocking=True)
      slot['lens'][:B].copy_(self.ctc_lens, non_blocking=True)
      ready = torch.cuda.Event()
      ready.record(main)
    desc = Tensor3(slot['logits'].data_ptr(), xl.batch, xl.frames, xl.channels, xl.halo, xl.t_pitch, xl.c_pitch)
    decode_stream.wait_event(ready)
    call('st_ctc_beam_search_decode_ex', ctypes.byref(desc), self._ptr(slot['lens']), int(beam_width), beam_input_transform(input_transform),
         self._ptr(slot['ids']), T,
         self._ptr(slot['out_lens']), self._ptr(slot['score']), self._ptr(slot['ws']), slot['ws'].numel() * 4,
         ctypes.c_void_p(decode_stream.cuda_stream))
    with torch.cuda.stream(decode_stream):
      slot['ids_h'][:B * T].copy_(slot['ids'][:B * T], non_blocking=True)
      slot['lens_h'][:B].copy_(slot['out_lens'][:B], non_blocking=True)
      slot['score_h'][:B].copy_(slot['score'][:B], non_blocking=True)
      slot['event'].record(decode_stream)
    return _PendingBeamDecode(slot, B, T)
