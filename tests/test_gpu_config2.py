"""BASELINE configs[2] at its REAL size: 1 x MI355X inference-only, conv stack + CTC greedy decode, batch 64 of
variable-length utterances (2-15 s at 16 kHz -> 201..1501 frames of 80 mel features), bucketed by length -- the
full-width network (250 / 2000 channels), through ``inference.transcribe`` as the CLI's evaluate path would.

What is checked, per bucket (the reference pads a batch to its longest member and masks nothing, speech_input.py:37-45,
so an utterance's logits are defined per batch composition -- the oracle gets the same padded rows):
  * logits of two rows of the shortest, of the longest full and of the ragged tail bucket against the float64 oracle
    (oracle/w2l_oracle.py, speech_model.py:275-295) < 1e-4;
  * greedy ids of ALL rows of EVERY bucket identical to the oracle decoder (speech_model.py:113-115) fed the device's
    own logits (ties cannot flip), and identical to what ``transcribe`` (pipelined, default) returned;
  * the launch trace of every bucket: which layers ran in the frequency domain (the policy is by output rows B x T',
    engine._use_fft), so the test cannot pass on the small-problem kernels -- full buckets: all nine layers (45- / 36- /
    48-bin batched products); the 3-utterance tail bucket: the 32-tap layer only, W-tap kernels for the rest.
"""
import time

import numpy as np
import pytest

from oracle import w2l_oracle as O
from tests import workloads as WL

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

BATCH = 64
N_UTT = 4 * BATCH + 3                    # four full buckets and a ragged tail of the three longest


def test_config2_bucketed_inference_at_full_size_matches_oracle():
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  from speecht_amd._lib import launch_trace
  from speecht_amd.engine import Wav2LetterEngine
  from speecht_amd.inference import make_buckets, padding_overhead, transcribe
  layers = WL.w2l_layers(80)
  params = WL.xavier_params(layers, seed=42, dtype=np.float32)
  p64 = [(F.astype(np.float64), b.astype(np.float64)) for F, b in params]
  rng = np.random.default_rng(2024)
  samples = rng.integers(32000, 240001, N_UTT)                       # 2 .. 15 s of 16 kHz audio
  lengths = (1 + samples // 160).tolist()                            # frames: hop 160, centred STFT
  assert min(lengths) >= 201 and max(lengths) <= 1501
  feats = [WL.synthetic_features(5000 + i, t, 80).astype(np.float32) for i, t in enumerate(lengths)]
  eng = Wav2LetterEngine(layers, device='cuda:0')
  eng.set_weights(params)

  t0 = time.time()
  ids, text = transcribe(eng, feats, batch_size=BATCH)               # bucketed + pipelined: the default
  torch.cuda.synchronize()
  print('transcribe of %d utterances (first call, allocations included): %.2f s' % (N_UTT, time.time() - t0))
  assert len(ids) == N_UTT and all(text[i] == O.ids_to_sentence(ids[i]) for i in range(N_UTT))

  buckets = make_buckets(lengths, BATCH)
  assert [len(b) for b in buckets] == [BATCH] * 4 + [3]
  arrival = [list(range(i, min(i + BATCH, N_UTT))) for i in range(0, N_UTT, BATCH)]
  print('padding: %.1f %% bucketed vs %.1f %% in arrival order' % (100 * padding_overhead(lengths, buckets),
                                                                  100 * padding_overhead(lengths, arrival)))
  assert padding_overhead(lengths, buckets) < 0.2 < padding_overhead(lengths, arrival)

  oracle_buckets = {0: 'shortest', 3: 'longest full', 4: 'ragged tail'}
  for k, idx in enumerate(buckets):
    max_t = max(lengths[i] for i in idx)
    x = np.zeros((len(idx), max_t, 80), dtype=np.float32)
    for row, i in enumerate(idx):
      x[row, :lengths[i]] = feats[i]
    seq = np.asarray([lengths[i] for i in idx], dtype=np.int64)
    eng.load_batch(x, seq)
    with launch_trace() as tr:
      eng.forward()
    torch.cuda.synchronize()
    trace = '\n'.join(tr.lines)
    t_out = (max_t + 1) // 2
    rows = len(idx) * t_out
    batched = lambda bins: sum(1 for l in tr.lines if l.startswith(('gemm_nn<', 'gemm_nn_bins<', 'gemm_nn_g3<')) and ' batched bins=%d ' % bins in l)
    # the policy of engine._use_fft: the 32-tap layer from 1 000 output rows, the 7-tap layers and the first layer from 3 000
    # (the wide layer's product runs as TWO launches when a bin's rows -- utterances x 64-frame blocks, padded to 64 -- end in a
    #  half 128-row tile: the whole tiles, then the last 64 rows on the 64-row kernel, st::gemm_nn_batched)
    rows_pad = -(-(len(idx) * -(-t_out // 64)) // 64) * 64
    wide_launches = 2 if (rows_pad % 128 == 64 and rows_pad > 128) else 1
    assert batched(48) == (wide_launches if rows >= 1000 else 0), trace
    assert batched(36) == (7 if rows >= 3000 else 0), trace
    assert batched(45) == (1 if rows >= 3000 else 0), trace
    wtap = [l for l in tr.lines if l.startswith('gemm_nn<') and 'batched' not in l]
    # W-tap launches: always L9 and L10; L0..L7 only below the narrow threshold
    assert len(wtap) == (2 if rows >= 3000 else 10), trace
    if len(idx) == BATCH:
      assert rows >= 3000                                              # every full bucket runs all nine layers there
    else:
      assert 1000 <= rows < 3000, rows                                 # the tail: three 14-15 s utterances

    got = eng.logits_time_major().cpu().numpy()                        # [T', B, 29]
    assert got.shape == (t_out, len(idx), 29)
    # every row of the bucket: device decoder == oracle decoder on the device's logits == what transcribe returned
    dec, score = eng.greedy_decode()
    ref_dec, ref_score = O.ctc_greedy_decode(got, seq // 2)
    assert dec == ref_dec, 'bucket %d' % k
    np.testing.assert_allclose(score, ref_score, rtol=1e-5)
    assert [ids[i] for i in idx] == dec, 'bucket %d: transcribe() and the serial loop disagree' % k
    if k in oracle_buckets:
      check = sorted({0, len(idx) - 1})
      t0 = time.time()
      ref = O.wav2letter_forward(x[check].astype(np.float64), p64, layers)
      err = float(np.max(np.abs(got[:, check] - ref)))
      print('bucket %d (%s, B = %d, T = %d, %d output rows): max|logit err| on rows %s = %.2e (oracle %.1f s); '
            'frequency-domain layers: %d' % (k, oracle_buckets[k], len(idx), max_t, rows, check, err, time.time() - t0,
                                             batched(48) + batched(36) + batched(45)))
      assert err < 1e-4 and err < 2e-5 * float(np.max(np.abs(ref))), (k, err, float(np.max(np.abs(ref))))     # absolute (north_star) and of the logits' own size
      # and the decode of those rows from the ORACLE's logits (ties aside, the strings the reference would print)
      o_dec, _ = O.ctc_greedy_decode(ref, (seq // 2)[check])
      assert [dec[r] for r in check] == o_dec, 'bucket %d' % k
