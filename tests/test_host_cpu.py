"""CPU-only tests: host logic, golden vectors, and that the C-ABI library loads and exports
every symbol include/speecht_hip.h declares (no compute calls without a GPU)."""
import json
import os
import re

import numpy as np
import pytest

from oracle import w2l_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_vocabulary_matches_reference_golden(golden_dir):
  from speecht_amd import vocabulary as V
  gold = json.load(open(os.path.join(golden_dir, 'vocabulary_golden.json')))
  assert V.SIZE == gold['meta']['SIZE'] and V.APOSTROPHE == gold['meta']['APOSTROPHE']
  assert V.SPACE_ID == gold['meta']['SPACE_ID']
  assert [V.id_to_letter(i) for i in range(V.SIZE)] == gold['meta']['id_to_letter']
  for case in gold['cases']:
    assert V.sentence_to_ids(case['sentence']) == case['ids']
    assert V.ids_to_sentence(case['ids']) == case['roundtrip']
    assert O.sentence_to_ids(case['sentence']) == case['ids']          # the oracle too
    assert O.ids_to_sentence(case['ids']) == case['roundtrip']


def test_oracle_golden_drift(golden_dir):
  """The committed small-case golden must still be what the oracle produces."""
  from tests.workloads import small_train_case
  gold = np.load(os.path.join(golden_dir, 'w2l_small_golden.npz'))
  case = small_train_case()
  out = O.train_step(case['x'], case['seq_lens'], case['labels'], case['params'], case['layers'],
                     O.zero_opt_state(case['params']), lr=1e-4)
  assert out['avg_loss'] == pytest.approx(float(gold['avg_loss']), rel=1e-12)
  np.testing.assert_allclose(out['logits'], gold['logits'], atol=1e-6)
  mel = O.calc_power_spectrogram(O.synthetic_audio(7, 16000 + 77), 16000, n_mels=80)
  np.testing.assert_allclose(mel, gold['mel80'], atol=1e-5)


def test_mel_filterbank_host_matches_oracle():
  from speecht_amd.preprocessing import mel_filterbank
  for sr, n_mels in [(16000.0, 80), (22050.0, 128), (16000.0, 40)]:
    np.testing.assert_allclose(mel_filterbank(sr, 512, n_mels), O.mel_filterbank(sr, 512, n_mels), atol=1e-15)


def test_library_exports_every_declared_symbol():
  import ctypes
  from speecht_amd import _lib
  from speecht_amd.build import LIB_PATH, build_library
  build_library(verbose=False)
  header = open(os.path.join(ROOT, 'include', 'speecht_hip.h')).read()
  header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
  declared = set(re.findall(r'\b(st_[a-z0-9_]+)\s*\(', header))
  assert declared, 'no declarations parsed'
  lib = ctypes.CDLL(LIB_PATH)
  for name in sorted(declared):
    assert hasattr(lib, name), name
  assert declared == set(_lib.EXPORTED_SYMBOLS)
  exp = open(os.path.join(ROOT, 'include', 'speecht_hip_experimental.h')).read()
  exp = set(re.findall(r'\b(st_exp_[a-z0-9_]+)\s*\(', re.sub(r'/\*.*?\*/', '', exp, flags=re.S)))
  assert exp == set(_lib._EXPERIMENTAL) and all(hasattr(lib, n) for n in exp)
  assert _lib.load().st_version() >= 100


def test_argument_validation_without_gpu():
  """Host-side precondition checks run before any launch."""
  import ctypes
  from speecht_amd import _lib
  lib = _lib.load()
  kv, kp, npad = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
  assert lib.st_packed_dims(32, 256, 2000, ctypes.byref(kv), ctypes.byref(kp), ctypes.byref(npad)) == 0
  assert (kv.value, kp.value, npad.value) == (8192, 8192, 2048)
  assert lib.st_packed_dims(1, 2000, 29, ctypes.byref(kv), ctypes.byref(kp), ctypes.byref(npad)) == 0
  assert (kv.value, kp.value, npad.value) == (2000, 2016, 32)
  assert lib.st_packed_dims(7, 250, 250, None, None, None) != 0          # pitch not a multiple of 16
  assert b'st_packed_dims' in lib.st_last_error()
  x = _lib.Tensor3(1, 2, 10, 16, 3, 16, 16)
  y = _lib.Tensor3(1, 2, 10, 16, 0, 10, 16)
  # halo too small for pad_left = 5
  assert lib.st_conv1d_nwc_fwd_f32(ctypes.byref(x), 1, None, 11, 1, 5, 1, ctypes.byref(y), None) == -1
  assert b'halo' in lib.st_last_error()
  # log2-softmax [32] + emission factors [32][2] + alpha, beta records [5 * 64][2] floats per (utterance, frame)
  assert lib.st_ctc_ws(32, 501, 150) == 32 * 501 * 4 * (32 * 3 + 4 * 5 * 64) + 512
  assert lib.st_ctc_ws(1, 10, 600) == 0                                    # label too long


def test_engine_refuses_cpu():
  from speecht_amd import _lib
  from speecht_amd.engine import Wav2LetterEngine
  with pytest.raises(_lib.SpeechtHipError):
    Wav2LetterEngine([(1, 1, 16, 29, False)], device='cpu')


def test_bench_trace_line_parsing():
  """bench.py's in-step roofline matches the library's launch-trace lines to the work they did: the symbol a line groups
  under, its key (kernel + shape + policy, measurements stripped) and its fields."""
  import importlib.util
  spec = importlib.util.spec_from_file_location('bench_module', os.path.join(ROOT, 'bench.py'))
  bench = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(bench)
  nn = 'gemm_nn<128,128,2,2,fast> epi=1 splits=2 M=16032 Np=256 Kp=64512 taps=32 xcd=8x1 gflop=1059.5 ms=0.54321'
  bt = 'gemm_nn<64,128,2,2,fast> batched bins=36 M=256 Np=512 Kp=512 gflop=4.832'
  sk = 'gemm_nn_bins<64,128,2,2> batched bins=36 M=256 Np=512 Kp=512 streamk wgs=512 upw=18 gflop=4.832'
  tn = 'gemm_tn<128> batched bins=36 M=256 Kp=512 Np=512 gflop=4.832 ms=0.06100'
  assert bench.trace_symbol(nn) == 'gemm_nn<128,128,2,2,fast> epi=1'
  assert bench.trace_symbol(bt) == 'gemm_nn<64,128,2,2,fast> epi=0'          # batched products: the plain epilogue
  assert bench.trace_symbol(sk) == 'gemm_nn_bins<64,128,2,2>'                 # the persistent stream-K form: a symbol of its own
  assert bench.trace_symbol(tn) == 'gemm_tn<128>' and bench.trace_symbol('dft_rows<3> rows=256 chunks=8 bins=36 gflop=0.9') == 'dft_rows<3>'
  assert bench.trace_key(nn) == 'gemm_nn<128,128,2,2,fast> epi=1 splits=2 M=16032 Np=256 Kp=64512 taps=32 xcd=8x1'
  assert bench.trace_key(tn) == bench.trace_key(tn.replace('ms=0.06100', 'ms=0.09'))
  assert bench.trace_field(nn, 'ms') == pytest.approx(0.54321) and bench.trace_field(nn, 'gflop') == pytest.approx(1059.5)
  assert bench.trace_field(bt, 'ms') is None and bench.trace_field(bt, 'bins') == 36.0


def test_speecht_import_names_resolve_to_the_native_modules():
  """The reference's callers import ``speecht.<module>`` (execution.py:20-23, evaluation.py:20-23): the alias package hands them
  the speecht_amd modules themselves (same objects, not copies); what is out of scope raises ImportError."""
  import importlib
  import speecht
  for name in ('vocabulary', 'preprocessing', 'speech_input', 'speech_model', 'evaluation', 'training', 'execution', 'exporting'):
    assert importlib.import_module('speecht.' + name) is importlib.import_module('speecht_amd.' + name), name
    assert getattr(speecht, name) is importlib.import_module('speecht_amd.' + name)
    assert importlib.import_module('speecht_amd.' + name).__spec__.name == 'speecht_amd.' + name       # the alias import leaves the module's own spec
  from speecht.speech_model import Wav2LetterModel, create_default_model      # noqa: F401
  from speecht.speech_input import InputBatchLoader, SingleInputLoader       # noqa: F401
  from speecht import vocabulary
  assert vocabulary.sentence_to_ids("don't") == [3, 14, 13, 26, 19]
  with pytest.raises(ImportError):
    importlib.import_module('speecht.corpus')
