#!/usr/bin/env python3
"""Generates tests/golden/* .  Run in the build container only (needs /root/reference).

1. vocabulary_golden.json -- produced by IMPORTING the reference's speecht.vocabulary (the one
   reference module that imports without tensorflow/librosa) on the transcripts the reference's
   own test fixture holds (speecht/tests/data/train/1089-134686.trans.txt) plus edge strings.
2. w2l_small_golden.npz -- oracle outputs (float64) for a small seeded Wav2Letter train step and
   a mel-feature case; regenerated deterministically from oracle/w2l_oracle.py so that the GPU
   box can check both the oracle (drift) and the HIP path against committed numbers.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')


def vocabulary_golden():
  sys.path.insert(0, '/root/reference')
  import speecht.vocabulary as ref_vocab   # reference module, imported not copied
  lines = open('/root/reference/speecht/tests/data/train/1089-134686.trans.txt').read().splitlines()
  cases = []
  for line in lines:
    audio_id, sentence = line.split(' ', 1)
    ids = ref_vocab.sentence_to_ids(sentence)
    cases.append(dict(audio_id=audio_id, sentence=sentence, ids=ids,
                      roundtrip=ref_vocab.ids_to_sentence(ids)))
  for s in ["DON'T", "a z ' ", "", "Hello World"]:
    ids = ref_vocab.sentence_to_ids(s)
    cases.append(dict(audio_id=None, sentence=s, ids=ids, roundtrip=ref_vocab.ids_to_sentence(ids)))
  meta = dict(SIZE=ref_vocab.SIZE, APOSTROPHE=ref_vocab.APOSTROPHE, SPACE_ID=ref_vocab.SPACE_ID,
              id_to_letter=[ref_vocab.id_to_letter(i) for i in range(ref_vocab.SIZE)])
  json.dump(dict(meta=meta, cases=cases), open(os.path.join(GOLD, 'vocabulary_golden.json'), 'w'), indent=1)
  print('vocabulary_golden.json:', len(cases), 'cases')


def small_w2l_golden():
  from oracle import w2l_oracle as O
  from tests.workloads import small_train_case
  case = small_train_case()
  out = O.train_step(case['x'], case['seq_lens'], case['labels'], case['params'], case['layers'],
                     O.zero_opt_state(case['params']), lr=1e-4)
  dec, score = O.ctc_greedy_decode(out['logits'], case['seq_lens'] // 2)
  save = dict(avg_loss=out['avg_loss'], loss=out['loss'], logits=out['logits'].astype(np.float32),
              grad_norm=out['grad_norm'],
              decoded=np.array([ids + [-1] * (64 - len(ids)) for ids in dec], dtype=np.int64),
              neg_sum_logits=score)
  for i, ((gF, gb), (pF, pb)) in enumerate(zip(out['grads'], out['params'])):
    save['gb%d' % i] = gb
    save['gF%d_sum' % i] = gF.sum()
    save['gF%d_abs' % i] = np.abs(gF).sum()
    save['pb%d' % i] = pb
    save['pF%d_sum' % i] = pF.sum()
  y = O.synthetic_audio(7, 16000 + 77)
  save['mel80'] = O.calc_power_spectrogram(y, 16000, n_mels=80).astype(np.float32)
  np.savez_compressed(os.path.join(GOLD, 'w2l_small_golden.npz'), **save)
  print('w2l_small_golden.npz avg_loss', out['avg_loss'], 'norm', out['grad_norm'])


if __name__ == '__main__':
  os.makedirs(GOLD, exist_ok=True)
  if os.path.isdir('/root/reference'):
    vocabulary_golden()
  small_w2l_golden()
