// Internal helpers shared by the gfx950 kernels of libspeecht_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdarg.h>
#include <stdio.h>

#include "speecht_hip.h"

namespace st {

void set_error(const char* fmt, ...);

// launch trace (st_trace_begin / st_trace_end) and tuning overrides (st_set_tuning); api.hip
bool trace_on();
void trace(const char* fmt, ...);
// Timed mode (st_trace_begin_timed): the launch that follows a trace() line goes through hipExtLaunchKernel with a start and
// a stop event, which take their times from the kernel dispatch's own completion signal -- the begin / end time stamps a
// profiler's kernel trace shows -- without putting an event marker (a barrier packet) between launches, so the launch
// sequence of a real step, side streams and all, runs as it does untimed.  (Tried first: event markers around each launch --
// +12...20 % on the step; begin / end time stamps by device atomics -- +7 %, and blind to the write-back at a kernel's end.)
// The line gets " ms=..." when the trace is collected.  Outside timed mode launch_timed() is a plain launch.
class LaunchTimer {
 public:
  explicit LaunchTimer(hipStream_t s);
  bool active() const { return start_ != nullptr; }
  hipEvent_t start() const { return start_; }
  hipEvent_t stop() const { return stop_; }

 private:
  hipEvent_t start_, stop_;
};
// (the arguments are converted to the kernel's own parameter types here: hipExtLaunchKernelGGL packs what it is given)
template <typename... KArgs, typename... Args>
inline void launch_timed(const LaunchTimer& t, void (*kernel)(KArgs...), const dim3& grid, const dim3& block, hipStream_t s,
                         Args&&... args) {
  static_assert(sizeof...(KArgs) == sizeof...(Args), "launch_timed: argument count does not match the kernel");
  if (t.active()) hipExtLaunchKernelGGL(kernel, grid, block, 0, s, t.start(), t.stop(), 0, static_cast<KArgs>(args)...);
  else hipLaunchKernelGGL(kernel, grid, block, 0, s, static_cast<KArgs>(args)...);
}
enum { TUNE_GEMM_TILE, TUNE_GEMM_SPLITS, TUNE_FWD_SPLITS, TUNE_XCD_GM, TUNE_NO_FAST, TUNE_BF16_TILE,
       TUNE_BF16_WGRAD_SPLITS, TUNE_BF16_SCHED, TUNE_STREAMK, TUNE_TRANSFORM_WGS, TUNE_BF16_WGRAD_TARGET, TUNE_STREAMK_SLOTS, TUNE_NO_FUSED_TRANSFORMS,
       TUNE_STREAMK_TEST_DROP, TUNE_BF16_LAG_COPIES, TUNE_BF16_WGRAD_RING, TUNE_BF16_TAPS_PANEL, TUNE_BF16_WGRAD_BIAS_PASS, TUNE_BF16_WGRAD_PLAIN_ORDER, TUNE_FILTERS_IDFT_VALU, TUNE_NO_ROW_SPLIT, TUNE_NO_G3, TUNE_G3_TILE, TUNE_COUNT };
int tuning(int key);

// conv_gemm.hip: batched plain GEMM on the fp32 MFMA convolution kernel (used by conv_fft.hip)
// sk_ws: optional SK_WS_FLOATS floats of scratch for the persistent stream-K form of an uneven launch (conv_gemm.hip,
// gemm_nn_bins_kernel): [control words: 768 flags | timeout count] then 768 partial tiles of 64 x 128.  Any content before the
// first call (flags carry the launch's epoch and are put back to zero by their reader); one scratch per stream.
constexpr int SK_FLAGS = 0, SK_TIMEOUTS = 768, SK_CTRL_WORDS = 1024;
constexpr long SK_WS_FLOATS = SK_CTRL_WORDS + 768L * 64 * 128;
int gemm_nn_batched(const float* A, long lda, long a_batch, const float* B, long b_batch, float* C, long ldc, long c_batch,
                    int M, int K, int N, int batches, hipStream_t s, float* sk_ws = nullptr, bool b_transposed = false);
int gemm_tn_batched(const float* A, long lda, long a_batch, const float* Z, long ldz, long z_batch, float* out, long o_batch,
                    int M, int K, int N, int batches, hipStream_t s, int z_batch_shift = 0);
// the same products of COMPLEX operands as three real products per bin instead of four (Gauss; conv_gemm.hip gemm_nn_g3_kernel /
// gemm_tn_g3_kernel): operand planes A_p = A + a_off[p] etc., the real and the imaginary result c_off2 columns / o_part floats apart
int gemm_nn_g3_batched(const float* A, long lda, long a_batch, const long a_off[3], const float* B, long ldb, long b_batch,
                       const long b_off[3], float* C, long ldc, long c_batch, long c_off2, int M, int K, int N, int batches,
                       hipStream_t s, bool b_transposed = false);
int gemm_tn_g3_batched(const float* A, long lda, long a_batch, const long a_off[3], const float* Z, long ldz, long z_batch,
                       const long z_off[3], float* out, long o_batch, long o_part, int M, int K, int N, int batches, hipStream_t s);

// conv_bf16.hip: the same per-bin products on the bf16 matrix pipe (planes = 1: bf16 operands; 3: the exact 3-way split of fp32
// operands, six product terms -- fp32-accurate), and the reduction-major copies of spectra the lag products read
int gemm_bf16_bins(int planes, const void* a_planes, size_t a_plane, long a_rows_apart, long a_bin, const void* bt_planes,
                   size_t b_plane, long ldb, long b_bin, float* c, long ldc, int rows, int k, int n, int bins, hipStream_t s,
                   int b_bin_shift = 0);
// wgrad_tr_bf16.hip: the lag products of the filter gradient straight from the spectra planes (LDS transposing reads)
int lag_products_tr_bf16(const void* s_plane, const void* z_plane, int bins, int rows, int half, int npo, float* q, hipStream_t s);
// conv_taps_bf16.hip: stride-1 W-tap layers on 256-channel bf16 rows with the input panel resident in LDS (tap w of output frame
// t reads operand frame t + w - lead; `act`: keep the result where the bf16 plane of `act` is positive)
bool conv_taps_bf16_eligible(const st_tensor3& a, const st_tensor3& y, int width, int lead, const st_tensor3* act);
int conv_taps_bf16(const st_tensor3& a, const void* a_plane, const void* filters, const float* bias, int width, int lead, int relu,
                   const st_tensor3* act, const void* act_plane, const st_tensor3& y, void* y_plane, hipStream_t s);
int transpose_bf16_bins(const void* src, void* dst, int bins, int rows, int cols, hipStream_t s);
int transpose_bf16_bins_split(const void* src, void* dst, int bins, int rows, int cols, int forms, hipStream_t s);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return ST_ELAUNCH;
  }
  return ST_OK;
}

#define ST_REQUIRE(cond, ...)      \
  do {                             \
    if (!(cond)) {                 \
      st::set_error(__VA_ARGS__);  \
      return ST_EINVAL;            \
    }                              \
  } while (0)

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline size_t round_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// float offset of row (b, t) of a padded NWC tensor
inline long row_offset(const st_tensor3& x, int b, int t) {
  return ((long)b * x.t_pitch + x.halo + t) * (long)x.c_pitch;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace st
