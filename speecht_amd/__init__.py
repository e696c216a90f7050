"""speecht_amd: MI355X-native Wav2Letter training/inference path of louiskirsch/speechT.

Host-side mirror of the reference's Python API (speech_model / speech_input / preprocessing /
vocabulary / evaluation / training) over hand-written gfx950 HIP kernels (csrc/) reached through
the C ABI in include/speecht_hip.h.  See DESIGN.md.
"""
__version__ = '0.1.0'
