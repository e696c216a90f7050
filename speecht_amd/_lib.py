"""ctypes binding of libspeecht_hip.so (include/speecht_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

from .build import LIB_PATH


class Tensor3(ctypes.Structure):
  """st_tensor3: padded NWC activation view."""
  _fields_ = [('base', c_void_p), ('batch', c_int32), ('frames', c_int32), ('channels', c_int32),
              ('halo', c_int32), ('t_pitch', c_int32), ('c_pitch', c_int32)]


class SpeechtHipError(RuntimeError):
  pass


_T3P = POINTER(Tensor3)
_SIGNATURES = {
    'st_version': (c_int, []),
    'st_last_error': (c_char_p, []),
    'st_trace_begin': (c_int, []),
    'st_trace_begin_timed': (c_int, []),
    'st_trace_timed_filter': (c_int, [c_char_p]),
    'st_trace_end': (c_size_t, [c_char_p, c_size_t]),
    'st_set_tuning': (c_int, [c_char_p, c_int]),
    'st_host_crc32c': (ctypes.c_uint32, [c_void_p, c_size_t, ctypes.c_uint32]),
    'st_packed_dims': (c_int, [c_int, c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    'st_pack_filters_f32': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'st_unpack_filters_f32': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'st_filters_flip_transpose_f32': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'st_conv1d_nwc_fwd_f32': (c_int, [_T3P, c_void_p, c_void_p, c_int, c_int, c_int, c_int, _T3P, c_void_p]),
    'st_conv1d_fwd_ws': (c_size_t, [_T3P, _T3P, c_int]),
    'st_conv1d_nwc_fwd_ws_f32': (c_int, [_T3P, c_void_p, c_void_p, c_int, c_int, c_int, c_int, _T3P, c_void_p, c_size_t,
                                          c_void_p]),
    'st_conv1d_fft_plan': (c_int, [c_int, c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int),
                                   POINTER(c_int)]),
    'st_gemm_nn_batched_f32': (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                                       c_int, c_int, c_void_p]),
    'st_gemm_nn_batched_ws_bytes': (c_size_t, []),
    'st_gemm_nn_batched_ctrl_bytes': (c_size_t, []),
    'st_streamk_lost_ptr': (c_int, [POINTER(c_void_p)]),
    'st_streamk_lost_count': (c_int, [POINTER(ctypes.c_uint32)]),
    'st_streamk_lost_fetch_async': (c_int, [c_void_p, c_void_p]),
    'st_gemm_nn_batched_ws_f32': (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                                          c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'st_gemm_nn_batched_bt_ws_f32': (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                                             c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'st_gemm_tn_batched_f32': (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_int,
                                       c_int, c_int, c_void_p]),
    'st_gemm_tn_batched_shared_f32': (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_int,
                                              c_int, c_int, c_int, c_void_p]),
    'st_gemm_nn_g3_batched_f32': (c_int, [c_void_p, c_int64, c_int64, POINTER(c_int64), c_void_p, c_int64, c_int64, POINTER(c_int64), c_int,
                                          c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    'st_gemm_tn_g3_batched_f32': (c_int, [c_void_p, c_int64, c_int64, POINTER(c_int64), c_void_p, c_int64, c_int64, POINTER(c_int64),
                                          c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    'st_conv1d_fft_three_products': (c_int, [c_int, c_int, c_int]),
    'st_conv1d_fft_table_floats': (c_size_t, []),
    'st_conv1d_fft_tables_f32': (c_int, [c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'st_conv1d_fft_filter_floats': (c_size_t, [c_int, c_int, c_int]),
    'st_conv1d_fft_filters_f32': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'st_conv1d_fft_sf_floats': (c_size_t, [_T3P, _T3P, c_int]),
    'st_conv1d_fft_zf_floats': (c_size_t, [_T3P, c_int]),
    'st_conv1d_fft_ws': (c_size_t, [_T3P, _T3P, c_int]),
    'st_conv1d_nwc_fwd_fft_f32': (c_int, [_T3P, c_void_p, c_void_p, c_int, c_int, c_int, _T3P, c_void_p, c_void_p, c_void_p,
                                          c_size_t, c_void_p]),
    'st_conv1d_nwc_fwd_fft_chain_f32': (c_int, [_T3P, c_void_p, c_void_p, c_int, c_int, c_int, _T3P, c_void_p, c_void_p, c_int, c_void_p,
                                                c_void_p, c_int, c_int, POINTER(c_int), c_void_p, c_size_t, c_void_p]),
    'st_conv1d_nwc_bwd_data_fft_chain_f32': (c_int, [_T3P, c_void_p, c_void_p, c_int, c_int, _T3P, _T3P, c_void_p, c_void_p, c_void_p, c_int,
                                                     POINTER(c_int), c_void_p, c_size_t, c_void_p]),
    'st_conv1d_fft_filter_plane_elems': (c_size_t, [c_int, c_int, c_int]),
    'st_conv1d_fft_filters_planes': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'st_conv1d_fft_planes_ws': (c_size_t, [_T3P, _T3P, c_int, c_int]),
    'st_conv1d_nwc_fwd_fft_planes': (c_int, [_T3P, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, _T3P, c_void_p, c_void_p, c_void_p,
                                             c_int, c_void_p, c_size_t, c_void_p]),
    'st_conv1d_fft_dz_spectra_planes': (c_int, [_T3P, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'st_conv1d_fft_bias_grad_dc_f32': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'st_conv1d_nwc_bwd_data_fft_planes': (c_int, [_T3P, c_void_p, c_void_p, c_int, c_int, _T3P, c_void_p, _T3P, c_void_p, c_void_p, c_int,
                                                  c_void_p, c_size_t, c_void_p]),
    'st_conv1d_nwc_bwd_filter_fft_planes': (c_int, [_T3P, _T3P, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_size_t,
                                                    c_void_p]),
    'st_conv1d_fft_dz_spectra_f32': (c_int, [_T3P, c_int, c_void_p, c_void_p, c_void_p]),
    'st_conv1d_fft_bias_grad_f32': (c_int, [_T3P, c_int, c_void_p, c_void_p, c_void_p]),
    'st_conv1d_nwc_bwd_data_fft_f32': (c_int, [_T3P, c_void_p, c_void_p, c_int, c_int, _T3P, _T3P, c_void_p, c_void_p, c_size_t,
                                               c_void_p]),
    'st_conv1d_nwc_bwd_filter_fft_f32': (c_int, [_T3P, _T3P, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                                 c_size_t, c_void_p]),
    'st_conv1d_bwd_data_ws': (c_size_t, [_T3P, _T3P, c_int]),
    'st_conv1d_nwc_bwd_data_f32': (c_int, [_T3P, c_void_p, c_int, c_int, _T3P, _T3P, c_void_p, c_size_t, c_void_p]),
    'st_conv1d_bwd_data_bias_ws': (c_size_t, [_T3P, _T3P, c_int]),
    'st_conv1d_nwc_bwd_data_bias_f32': (c_int, [_T3P, c_void_p, c_int, c_int, _T3P, _T3P, c_void_p, c_void_p, c_size_t,
                                                c_void_p]),
    'st_conv1d_1tap_bwd_data_bias_f32': (c_int, [_T3P, c_void_p, _T3P, _T3P, c_void_p, c_void_p, c_size_t, c_void_p]),
    'st_conv1d_bwd_filter_ws': (c_size_t, [_T3P, _T3P, c_int]),
    'st_conv1d_nwc_bwd_filter_f32': (c_int, [_T3P, _T3P, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                             c_size_t, c_void_p]),
    'st_bias_grad_ws': (c_size_t, [_T3P]),
    'st_bias_grad_f32': (c_int, [_T3P, c_void_p, c_void_p, c_size_t, c_void_p]),
    'st_ctc_ws': (c_size_t, [c_int, c_int, c_int]),
    'st_ctc_loss_grad_f32': (c_int, [_T3P, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, _T3P,
                                     c_void_p, c_void_p, c_size_t, c_void_p]),
    'st_ctc_loss_grad_hilo_f32': (c_int, [_T3P, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, _T3P,
                                     c_void_p, c_void_p, c_size_t, c_void_p]),
    'st_ctc_greedy_decode': (c_int, [_T3P, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'st_ctc_beam_ws': (c_size_t, [c_int, c_int, c_int]),
    'st_ctc_beam_search_decode': (c_int, [_T3P, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                          c_size_t, c_void_p]),
    'st_ctc_beam_search_decode_ex': (c_int, [_T3P, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                             c_size_t, c_void_p]),
    'st_global_norm_ws': (c_size_t, [c_size_t]),
    'st_global_norm_clip_adam_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float,
                                             c_float, c_float, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    'st_ctc_status_gate_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    'st_ctc_status_gate_loss_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    'st_global_norm_clip_adam_gated_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float,
                                                   c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_size_t,
                                                   c_void_p]),
    'st_global_norm_f32': (c_int, [c_void_p, c_size_t, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    'st_melspec_plan_bytes': (c_size_t, []),
    'st_melspec_plan_f32': (c_int, [c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'st_melspec_planned_f32': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_int, c_int, c_int, c_void_p,
                                       c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    'st_melspec_ws': (c_size_t, [c_int, c_int64, c_int]),
    'st_melspec_f32': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_int, c_int, c_int, c_void_p,
                               c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    'st_cast_bf16': (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
    'st_filters_bf16': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'st_filters_bwd_bf16': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'st_conv1d_nwc_fwd_bf16': (c_int, [_T3P, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, _T3P, c_void_p,
                                       c_void_p, c_void_p]),
    'st_conv1d_fwd_bf16_ws': (c_size_t, [_T3P, _T3P, c_int]),
    'st_conv1d_nwc_fwd_ws_bf16': (c_int, [_T3P, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, _T3P, c_void_p,
                                          c_void_p, c_void_p, c_size_t, c_void_p]),
    'st_conv1d_bwd_data_bf16_ws': (c_size_t, [_T3P, _T3P, c_int]),
    'st_conv1d_nwc_bwd_data_bf16': (c_int, [_T3P, c_void_p, c_void_p, c_int, c_int, _T3P, c_void_p, _T3P, c_void_p,
                                            c_void_p, c_size_t, c_void_p]),
    'st_conv1d_bwd_filter_bf16_ws': (c_size_t, [_T3P, _T3P, c_int, c_int, c_int]),
    'st_conv1d_nwc_bwd_filter_bf16': (c_int, [_T3P, c_void_p, _T3P, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                              c_void_p, c_size_t, c_void_p]),
    'st_conv1d_bwd_filter_tr_bf16_slack_rows': (c_int, []),
    'st_conv1d_bwd_filter_tr_bf16_ws': (c_size_t, [_T3P, _T3P, c_int, c_int, c_int]),
    'st_conv1d_nwc_bwd_filter_tr_bf16': (c_int, [_T3P, c_void_p, _T3P, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                                 c_void_p, c_size_t, c_void_p]),
    'st_comm_unique_id_bytes': (c_int, []),
    'st_comm_unique_id': (c_int, [c_void_p, c_size_t]),
    'st_comm_init': (c_int, [c_void_p, c_size_t, c_int, c_int, POINTER(c_void_p)]),
    'st_comm_destroy': (c_int, [c_void_p]),
    'st_comm_count': (c_int, [c_void_p, POINTER(c_int)]),
    'st_allreduce_f32': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'st_allreduce_buckets_f32': (c_int, [c_void_p, c_void_p, POINTER(c_size_t), POINTER(c_size_t), c_int, c_void_p]),
    'st_mfcc_ws': (c_size_t, [c_int, c_int64, c_int, c_int]),
    'st_mfcc_f32': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                            c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    'st_fill_f32': (c_int, [c_void_p, c_float, c_size_t, c_void_p]),
    'st_zero_halos_f32': (c_int, [_T3P, c_void_p]),
    'st_zero_regions': (c_int, [c_void_p, c_int, c_void_p]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

# experimental entry points (csrc/conv_bf16.hip), declared in include/speecht_hip_experimental.h
_EXPERIMENTAL = {
    'st_exp_split3_bf16': (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
    'st_exp_split3_transpose_bf16': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'st_exp_conv1d_fwd_bf16x6': (c_int, [_T3P, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, _T3P,
                                         c_void_p, c_void_p]),
    'st_exp_conv1d_bwd_data_bf16x6_ws': (c_size_t, [_T3P, _T3P, c_int]),
    'st_exp_conv1d_bwd_data_bf16x6': (c_int, [_T3P, c_void_p, c_void_p, c_int, c_int, _T3P, _T3P, c_void_p, c_void_p, c_size_t,
                                              c_void_p]),
    'st_exp_transpose_split3_bf16': (c_int, [_T3P, c_int, c_int, c_int, c_size_t, c_void_p, c_void_p]),
    'st_exp_conv1d_bwd_filter_bf16x6': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
}

_lib = None


def load():
  """Load the shared library (once).  Raises if it has not been built."""
  global _lib
  if _lib is None:
    # The process must run ONE HIP runtime.  torch bundles its own libamdhip64.so (soname without
    # version), hipcc links ours against /opt/rocm's libamdhip64.so.7: make torch's runtime globally
    # visible first so that our HIP calls bind to the runtime that owns torch's streams and memory.
    import torch
    bundled = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
    if os.path.exists(bundled):
      ctypes.CDLL(bundled, mode=ctypes.RTLD_GLOBAL)
    if not os.path.exists(LIB_PATH):
      raise SpeechtHipError('{} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                            '(there is no CPU fallback)'.format(LIB_PATH))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in list(_SIGNATURES.items()) + list(_EXPERIMENTAL.items()):
      fn = getattr(lib, name)
      fn.restype = res
      fn.argtypes = args
    _lib = lib
    # ST_TUNE="knob=value,knob=value": st_set_tuning overrides for A/B runs of scripts that have no --tune option (experiments only)
    for item in filter(None, os.environ.get('ST_TUNE', '').split(',')):
      name, _, value = item.partition('=')
      check(lib.st_set_tuning(name.strip().encode(), int(value)), 'ST_TUNE ' + item)
  return _lib


def check(code, what):
  if code != 0:
    raise SpeechtHipError('{} failed ({}): {}'.format(what, code, load().st_last_error().decode()))


def call(name, *args):
  """Invoke an int-returning entry point and raise on a non-zero status."""
  check(getattr(load(), name)(*args), name)


class launch_trace:
  """``with launch_trace() as tr: ...`` -> ``tr.lines``: one line per kernel launch the library made inside
  the block, naming the variant and split policy (st_trace_begin / st_trace_end).  ``timed=True``: every traced
  launch is bracketed by HIP events on its own stream and its line ends in `` ms=<duration>`` (collecting the trace
  waits for the launches)."""

  def __init__(self, timed=False, only=None):
    """``only``: in timed mode, time just the launches whose trace line contains this text (a timed launch costs the
    stream a few microseconds: timing one kernel's launches leaves the step undisturbed)."""
    self.timed, self.only = timed, only

  def __enter__(self):
    if self.timed:
      load().st_trace_timed_filter((self.only or '').encode())
      load().st_trace_begin_timed()
    else:
      load().st_trace_begin()
    self.lines = []
    return self

  def __exit__(self, *exc):
    lib = load()
    need = lib.st_trace_end(None, 0)
    buf = ctypes.create_string_buffer(int(need))
    lib.st_trace_end(buf, need)
    self.lines = [l for l in buf.value.decode().split('\n') if l]
    return False


TUNING_EPOCH = [0]        # bumped by set_tuning: policies size workspaces, so an engine's cached shape descriptions are keyed by it


def set_tuning(name, value):
  """Performance-experiment override (0 = library policy); see include/speecht_hip.h."""
  TUNING_EPOCH[0] += 1
  call('st_set_tuning', name.encode(), int(value))
