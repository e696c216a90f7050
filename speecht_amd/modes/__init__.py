"""Arithmetic modes of the Wav2Letter engine: what each one allocates beyond the shared buffers, and its launch sequences.

  fp32     exact-fp32 MFMA kernels; nine of the eleven layers in the frequency domain (BASELINE configs[1], the headline)
  bf16x6   the same with the wide 1 x 1 layers as fp32-accurate three-piece bf16 products (experimental, opt-in)
  bf16     bf16 activations and activation gradients, fp32 masters / accumulation / CTC / Adam (BASELINE configs[3])

A mode object holds no tensors of its own: it reads and writes the attributes of the engine it belongs to (`self.e`), so
`eng.fft`, `eng.Xb`, ... stay where tests, bench.py and the profiling scripts look for them.
"""
from .bf16 import Bf16Mode
from .bf16x6 import Bf16x6Mode
from .fp32 import Fp32Mode

MODES = {'fp32': Fp32Mode, 'bf16x6': Bf16x6Mode, 'bf16': Bf16Mode}


def make_mode(engine):
  return MODES[engine.conv_mode](engine)
