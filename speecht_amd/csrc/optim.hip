// clip_by_global_norm + TF-style Adam on the flat parameter / gradient buffers (gfx950).
//
// Replaces tf.clip_by_global_norm(grads, 5.0) and tf.train.AdamOptimizer(lr, epsilon=1e-3)
// .apply_gradients (speech_model.py:77-82).  TF adds epsilon OUTSIDE
// the bias correction: p -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps).
//
// HBM-bound streaming kernels: 16-byte loads, grid-stride, ~2048 blocks.  The global norm is a
// fixed-shape two-level sum (deterministic): NORM_BLOCKS partials, each re-summed in the same
// order by every block of the update kernel -- no atomics, no host round trip for the scale.
#include <algorithm>

#include "st_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NORM_BLOCKS = 1024;

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = st::wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, size_t n,
                                                            float* __restrict__ partial) {
  __shared__ float red[4];
  const size_t n4 = n / 4;
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 x = reinterpret_cast<const f32x4*>(g)[i];
    s += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { float x = g[n4 * 4 + threadIdx.x]; s += x * x; }
  float t = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__device__ __forceinline__ float total_sumsq(const float* __restrict__ partial, float* red) {
  float s = 0.f;
  for (int i = threadIdx.x; i < NORM_BLOCKS; i += 256) s += partial[i];
  return block_sum_256(s, red);
}

__global__ __launch_bounds__(256) void norm_stats_kernel(const float* __restrict__ partial, float clip,
                                                         float* __restrict__ stats) {
  __shared__ float red[4];
  float ss = total_sumsq(partial, red);
  if (threadIdx.x == 0) {
    float gn = sqrtf(ss);
    stats[0] = gn;
    stats[1] = clip / fmaxf(gn, clip);
  }
}

__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, size_t n,
                                                        const float* __restrict__ partial, float clip, float lr_t_value,
                                                        float b1, float b2, float eps, float* __restrict__ stats,
                                                        const float* __restrict__ gate) {
  __shared__ float red[4];
  const float lr_t = lr_t_value;
  const float gn = sqrtf(total_sumsq(partial, red));
  const float scale = clip / fmaxf(gn, clip);
  if (blockIdx.x == 0 && threadIdx.x == 0 && stats) { stats[0] = gn; stats[1] = scale; }
  // a batch that tf.nn.ctc_loss would have rejected (InvalidArgument fails the whole sess.run before any variable
  // is touched, speech_model.py:74,82) must leave weights and Adam state alone: uniform early exit
  if (gate && gate[0] != 0.f) return;
  // a gradient that is not finite (a poisoned stream-K tile, an overflow) must not reach params / m / v: fmaxf(NaN, clip)
  // would keep the scale at 1 and Adam would store the NaN.  stats[0] carries the norm, so the host sees why nothing moved.
  if (!(gn <= 3.0e38f)) return;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 gg = reinterpret_cast<const f32x4*>(g)[i] * scale;
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mm[e] = b1 * mm[e] + (1.f - b1) * gg[e];
      vv[e] = b2 * vv[e] + (1.f - b2) * gg[e] * gg[e];
      pp[e] -= lr_t * mm[e] / (sqrtf(vv[e]) + eps);
    }
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    reinterpret_cast<f32x4*>(p)[i] = pp;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    size_t i = n4 * 4 + threadIdx.x;
    float gg = g[i] * scale;
    float mm = b1 * m[i] + (1.f - b1) * gg;
    float vv = b2 * v[i] + (1.f - b2) * gg * gg;
    m[i] = mm; v[i] = vv;
    p[i] -= lr_t * mm / (sqrtf(vv) + eps);
  }
}

// gate[0] = number of utterances CTC refused; with `loss_hi`: gate[1] = sum_b (hi_b + lo_b) * loss_scale, summed in double in a
// fixed order (lane-strided partials, then a butterfly) -- this rank's share of the global mean loss, which travels with the
// gradients through the all-reduce (the slots sit inside the first reduce bucket) instead of a second, blocking scalar exchange
__global__ void status_gate_kernel(const int* __restrict__ status, int n, float* __restrict__ gate,
                                   const float* __restrict__ loss_hi, const float* __restrict__ loss_lo, float loss_scale) {
  int bad = 0;
  double sum = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) {
    bad += status[i] != 0;
    if (loss_hi) sum += (double)loss_hi[i] + (loss_lo ? (double)loss_lo[i] : 0.0);
  }
  bad = (int)st::wave_sum((float)bad);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if (threadIdx.x == 0) {
    gate[0] = (float)bad;
    if (loss_hi) gate[1] = (float)(sum * (double)loss_scale);
  }
}

}  // namespace

extern "C" {

int st_ctc_status_gate_f32(const int32_t* status, int batch, float* gate, void* stream) {
  ST_REQUIRE(status && gate && batch > 0, "status_gate: bad args");
  hipLaunchKernelGGL(status_gate_kernel, dim3(1), dim3(64), 0, st::as_stream(stream), status, batch, gate, (const float*)nullptr,
                     (const float*)nullptr, 0.f);
  return st::check_launch("status_gate");
}

int st_ctc_status_gate_loss_f32(const int32_t* status, int batch, const float* loss_hi, const float* loss_lo, float loss_scale,
                                float* gate, void* stream) {
  ST_REQUIRE(status && gate && loss_hi && batch > 0, "status_gate_loss: bad args");
  hipLaunchKernelGGL(status_gate_kernel, dim3(1), dim3(64), 0, st::as_stream(stream), status, batch, gate, loss_hi, loss_lo, loss_scale);
  return st::check_launch("status_gate_loss");
}

size_t st_global_norm_ws(size_t n) { (void)n; return NORM_BLOCKS * sizeof(float); }

int st_global_norm_f32(const float* grads, size_t n, float clip_norm, float* stats, void* workspace,
                       size_t workspace_bytes, void* stream) {
  ST_REQUIRE(grads && stats && workspace && workspace_bytes >= st_global_norm_ws(n), "global_norm: bad args");
  ST_REQUIRE(((uintptr_t)grads & 15) == 0, "global_norm: buffer must be 16-byte aligned");
  hipStream_t s = st::as_stream(stream);
  float* partial = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(NORM_BLOCKS), dim3(256), 0, s, grads, n, partial);
  hipLaunchKernelGGL(norm_stats_kernel, dim3(1), dim3(256), 0, s, partial, clip_norm, stats);
  return st::check_launch("global_norm");
}

int st_global_norm_clip_adam_f32(float* params, const float* grads, float* m, float* v, size_t n, float clip_norm,
                                 float lr_t, float beta1, float beta2, float eps, float* stats, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  return st_global_norm_clip_adam_gated_f32(params, grads, m, v, n, clip_norm, lr_t, beta1, beta2, eps, stats, nullptr,
                                            workspace, workspace_bytes, stream);
}

int st_global_norm_clip_adam_gated_f32(float* params, const float* grads, float* m, float* v, size_t n, float clip_norm,
                                       float lr_t, float beta1, float beta2, float eps, float* stats, const float* gate,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  ST_REQUIRE(params && grads && m && v && workspace && workspace_bytes >= st_global_norm_ws(n), "clip_adam: bad args");
  ST_REQUIRE((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
             "clip_adam: buffers must be 16-byte aligned");
  hipStream_t s = st::as_stream(stream);
  float* partial = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(NORM_BLOCKS), dim3(256), 0, s, grads, n, partial);
  const int blocks = (int)std::max<size_t>(1, std::min<size_t>((n / 4 + 255) / 256, 2048));
  hipLaunchKernelGGL(clip_adam_kernel, dim3(blocks), dim3(256), 0, s, params, grads, m, v, n, partial, clip_norm,
                     lr_t, beta1, beta2, eps, stats, gate);
  return st::check_launch("clip_adam");
}

}  // extern "C"
