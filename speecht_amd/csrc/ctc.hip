// CTC loss + gradient and greedy decoding on gfx950.
//
// Replaces tf.nn.ctc_loss (speech_model.py:74; CPU-only op in TF1) and
// tf.nn.ctc_greedy_decoder(merge_repeated=True) (speech_model.py:113-115).
// Semantics follow TF: blank = C-1, log-space f32, beta excludes the
// emission at t, loss = -log p(l|x) unnormalised, gradient wrt the *logits*.
//
// Structure:
//  1. ctc_logsoftmax: one thread per (b, t) row of <= 32 classes.
//  2. ctc_alpha_beta<KPL>: the sequential part.  One WAVE per (utterance, direction): the
//     2L+1 lattice states are dealt KPL-contiguous per lane, so a whole time step is register
//     arithmetic plus two cross-lane shifts -- no LDS round trip, no barrier on the 500-1500
//     step critical path.  Emissions are staged through LDS in 64-frame chunks, prefetched
//     one chunk ahead.  alpha and beta run concurrently on different CUs.
//  3. ctc_grad: fully parallel over (b, t): occupancy per class from alpha+beta with a fixed
//     summation order (per-class position lists), so results are run-to-run deterministic.
#include <algorithm>

#include "st_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int CP = 32;        // class pitch of the log-softmax scratch
constexpr int TC = 64;        // frames per LDS emission chunk
#define NEG_INF (-__builtin_inff())

struct RowMap2 {   // (b, t) -> float offset
  long batch_stride;
  long row0;
  int row_stride;
  __device__ __forceinline__ long off(int b, int t) const {
    return (long)b * batch_stride + row0 + (long)t * row_stride;
  }
};

// ---- base-2 log-domain helpers: v_exp_f32 / v_log_f32 are 2^x / log2(x) natively -------------
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float lg2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float lse2_b2(float a, float b) {
  float m = fmaxf(a, b);
  float ms = m == NEG_INF ? 0.f : m;
  return ms + lg2(ex2(a - ms) + ex2(b - ms));
}
__device__ __forceinline__ float lse3_b2(float a, float b, float c) {   // branch free; all -inf -> -inf
  float m = fmaxf(fmaxf(a, b), c);
  float ms = m == NEG_INF ? 0.f : m;
  return ms + lg2(ex2(a - ms) + ex2(b - ms) + ex2(c - ms));
}

// cross-lane moves on the VALU (DPP), no LDS round trip
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v, float fill) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill),
                                                               __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float from_lane_below(float v) { return dpp_move<0x138>(v, NEG_INF); }   // wave_shr:1
__device__ __forceinline__ float from_lane_above(float v) { return dpp_move<0x130>(v, NEG_INF); }   // wave_shl:1
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, dpp_move<0xB1>(v, v));     // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp_move<0x4E>(v, v));     // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp_move<0x141>(v, v));    // row_half_mirror
  v = fmaxf(v, dpp_move<0x140>(v, v));    // row_mirror: every lane of a 16-row holds the row max
  // the four rows through the scalar unit.  (v_permlane16_swap / v_permlane32_swap would stay on the vector side, but
  // hipcc folds max(r[0], r[1]) of a swap's two results to r[0]: the "wave maximum" then was the maximum of lanes
  // 0..15 -- uniform, so the re-centred lattice stayed consistent, but not centred: found when a linear-domain
  // variant of the recursion, which needs the true maximum, overflowed.)
  const int b = __builtin_bit_cast(int, v);
  const float m0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
  const float m1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float m2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
  const float m3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
}

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr int NORM_EVERY = 4;   // re-centre the lattice column every 4 frames (every frame measured no more accurate)

// log2-softmax of every (b, t) row into the scratch [B*T][32]
__global__ __launch_bounds__(256) void ctc_logsoftmax_kernel(const float* __restrict__ logits, RowMap2 map,
                                                             int B, int T, int C, float* __restrict__ logy) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * T) return;
  int b = i / T, t = i - b * T;
  const float* row = logits + map.off(b, t);
  float v[CP];
#pragma unroll
  for (int q = 0; q < CP / 4; ++q) {
    f32x4 x = *reinterpret_cast<const f32x4*>(row + 4 * q);
    v[4 * q] = x[0]; v[4 * q + 1] = x[1]; v[4 * q + 2] = x[2]; v[4 * q + 3] = x[3];
  }
  float m = NEG_INF;
#pragma unroll
  for (int c = 0; c < CP; ++c) if (c < C) m = fmaxf(m, v[c]);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CP; ++c) if (c < C) s += expf(v[c] - m);
  const float lz = m + logf(s);
  float* out = logy + (long)i * CP;
#pragma unroll
  for (int q = 0; q < CP / 4; ++q) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (4 * q + e < C) ? (v[4 * q + e] - lz) * LOG2E : 0.f;
    *reinterpret_cast<f32x4*>(out + 4 * q) = o;
  }
}

// One wave per (utterance, direction).  blockIdx.y: 0 = alpha, 1 = beta.
// Lattice columns are kept RE-CENTRED: stored value = log2 alpha_t(u) - off[t], where off (double,
// one per frame) accumulates the wave maximum subtracted every NORM_EVERY frames.  The f32 state
// therefore stays O(10) instead of O(-1000) and keeps ~1e-6 absolute precision over 500+ frames.
// Storage is lane-major: value of state u = lane*KPL + j at frame t sits at ((t*KPL + j)*64 + lane).
template <int KPL>
__global__ __launch_bounds__(64) void ctc_alpha_beta_kernel(const float* __restrict__ logy, int T, int C,
                                                            const int* __restrict__ label_ids,
                                                            const int* __restrict__ label_off,
                                                            const int* __restrict__ seq_lens,
                                                            float* __restrict__ alpha, float* __restrict__ beta,
                                                            double* __restrict__ aoff, double* __restrict__ boff,
                                                            int* __restrict__ status) {
  constexpr int UP = KPL * 64;
  __shared__ __attribute__((aligned(16))) float E[2][TC * CP];
  const int b = blockIdx.x;
  const bool is_beta = blockIdx.y != 0;
  const int lane = threadIdx.x;
  const int blank = C - 1;
  const int* lab = label_ids + label_off[b];
  const int L = label_off[b + 1] - label_off[b];
  const int U = 2 * L + 1;
  const int Tb = seq_lens[b];

  // "Not enough time for target transition sequence": L + #adjacent repeats must fit in Tb
  int rep = 0;
  for (int i = 1 + lane; i < L; i += 64) rep += lab[i] == lab[i - 1];
  rep = (int)st::wave_sum((float)rep);
  const bool bad = Tb < 0 || Tb > T || L + rep > Tb || U > UP;
  if (bad) {
    if (!is_beta && lane == 0) status[b] = 1;
    return;
  }
  if (!is_beta && lane == 0) status[b] = 0;
  if (Tb == 0) return;      // no frames and (checked above) an empty label: p = 1, nothing to recurse over

  int coff[KPL];          // class column of each state (LDS float offset inside an emission row)
  bool valid[KPL], skip[KPL];
#pragma unroll
  for (int j = 0; j < KPL; ++j) {
    const int u = lane * KPL + j;
    valid[j] = u < U;
    const bool odd = (u & 1) && valid[j];
    const int li = (u - 1) >> 1;
    coff[j] = odd ? lab[li] : blank;
    if (!is_beta) skip[j] = odd && u >= 3 && lab[li] != lab[li - 1];          // may arrive from u-2
    else skip[j] = odd && u + 2 < U && lab[li + 1] != lab[li];                  // may leave to u+2
  }

  const float* ly = logy + (long)b * T * CP;
  float* dst = (is_beta ? beta : alpha) + (long)b * T * UP + lane;
  double* doff = (is_beta ? boff : aoff) + (long)b * T;

  // stage one 64-frame chunk of emissions [chunk*TC, +TC) into E[buf]; rows past T read row T-1
  f32x4 stage[TC * CP / 4 / 64];
  auto chunk_load = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < TC * CP / 4 / 64; ++i) {
      int f = lane + 64 * i;                 // float4 index inside the chunk
      int t = min(chunk * TC + f / (CP / 4), T - 1);
      stage[i] = *reinterpret_cast<const f32x4*>(ly + (long)t * CP + (f % (CP / 4)) * 4);
    }
  };
  auto chunk_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < TC * CP / 4 / 64; ++i)
      *reinterpret_cast<f32x4*>(&E[buf][(lane + 64 * i) * 4]) = stage[i];
  };
  auto recentre = [&](float (&s)[KPL], double& off) {
    float m = NEG_INF;
#pragma unroll
    for (int j = 0; j < KPL; ++j) m = fmaxf(m, s[j]);
    m = wave_max_dpp(m);
    if (m != NEG_INF) {
#pragma unroll
      for (int j = 0; j < KPL; ++j) s[j] -= m;
      off += (double)m;
    }
  };

  float s[KPL];   // re-centred log2 alpha_t(u) resp. log2 beta_t(u)
  double off = 0.0;
  if (!is_beta) {
    // ---- alpha: t ascending ------------------------------------------------------------
    chunk_load(0);
    chunk_store(0);
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      const int u = lane * KPL + j;
      s[j] = (u < 2 && valid[j]) ? E[0][coff[j]] : NEG_INF;
    }
#pragma unroll
    for (int j = 0; j < KPL; ++j) dst[j * 64] = s[j];
    if (lane == 0) doff[0] = 0.0;
    const int nchunks = (Tb + TC - 1) / TC;
    for (int ch = 0; ch < nchunks; ++ch) {
      const int buf = ch & 1;
      if (ch + 1 < nchunks) chunk_load(ch + 1);
      const int t_lo = max(1, ch * TC), t_hi = min(Tb, (ch + 1) * TC);
      for (int t = t_lo; t < t_hi; ++t) {
        const float* e = &E[buf][(t - ch * TC) * CP];
        float ev[KPL];
#pragma unroll
        for (int j = 0; j < KPL; ++j) ev[j] = e[coff[j]];
        const float up1 = from_lane_below(s[KPL - 1]);
        const float up2 = KPL >= 2 ? from_lane_below(s[KPL >= 2 ? KPL - 2 : 0]) : from_lane_below(up1);
        float n[KPL];
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
          const float p1 = j >= 1 ? s[j >= 1 ? j - 1 : 0] : up1;
          const float p2 = j >= 2 ? s[j >= 2 ? j - 2 : 0] : (j == 1 ? up1 : up2);
          const float v = ev[j] + lse3_b2(s[j], p1, skip[j] ? p2 : NEG_INF);
          n[j] = valid[j] ? v : NEG_INF;
        }
#pragma unroll
        for (int j = 0; j < KPL; ++j) s[j] = n[j];
        if ((t & (NORM_EVERY - 1)) == 0) recentre(s, off);
        float* o = dst + (long)t * UP;
#pragma unroll
        for (int j = 0; j < KPL; ++j) o[j * 64] = s[j];
        if (lane == 0) doff[t] = off;
      }
      if (ch + 1 < nchunks) chunk_store(buf ^ 1);
    }
  } else {
    // ---- beta: t descending; step t consumes the emissions of frame t+1 -------------------
    const int last = (Tb - 1) / TC;
    chunk_load(last);
    chunk_store(last & 1);
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      const int u = lane * KPL + j;
      s[j] = (valid[j] && u >= U - 2) ? 0.f : NEG_INF;
    }
    {
      float* o = dst + (long)(Tb - 1) * UP;
#pragma unroll
      for (int j = 0; j < KPL; ++j) o[j * 64] = s[j];
      if (lane == 0) doff[Tb - 1] = 0.0;
    }
    for (int ch = last; ch >= 0; --ch) {
      const int buf = ch & 1;
      if (ch > 0) chunk_load(ch - 1);
      // frames f = t+1 of this chunk: f in [max(1, ch*TC), min(Tb-1, ch*TC+TC-1)]
      const int f_hi = min(Tb - 1, ch * TC + TC - 1), f_lo = max(1, ch * TC);
      for (int f = f_hi; f >= f_lo; --f) {
        const float* e = &E[buf][(f - ch * TC) * CP];
        float g[KPL];
#pragma unroll
        for (int j = 0; j < KPL; ++j) g[j] = s[j] + e[coff[j]];
        const float dn1 = from_lane_above(g[0]);
        const float dn2 = KPL >= 2 ? from_lane_above(g[KPL >= 2 ? 1 : 0]) : from_lane_above(dn1);
        float n[KPL];
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
          const float p1 = j + 1 < KPL ? g[j + 1 < KPL ? j + 1 : 0] : dn1;
          const float p2 = j + 2 < KPL ? g[j + 2 < KPL ? j + 2 : 0] : (j + 2 == KPL ? dn1 : dn2);
          const float v = lse3_b2(g[j], p1, skip[j] ? p2 : NEG_INF);
          n[j] = valid[j] ? v : NEG_INF;
        }
#pragma unroll
        for (int j = 0; j < KPL; ++j) s[j] = n[j];
        if ((f & (NORM_EVERY - 1)) == 0) recentre(s, off);
        float* o = dst + (long)(f - 1) * UP;
#pragma unroll
        for (int j = 0; j < KPL; ++j) o[j * 64] = s[j];
        if (lane == 0) doff[f - 1] = off;
      }
      if (ch > 0) chunk_store(buf ^ 1);
    }
  }
}

// grad[b,t,c] = scale * (y_t(c) - sum_{u: l'_u = c} exp(alpha_t(u) + beta_t(u) - log p))
constexpr int GF = 16;   // frames per block (4 waves x 4)
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ logy, const float* __restrict__ alpha,
                                                       const float* __restrict__ beta,
                                                       const double* __restrict__ aoff,
                                                       const double* __restrict__ boff, int T, int C, int KPL,
                                                       const int* __restrict__ label_ids,
                                                       const int* __restrict__ label_off,
                                                       const int* __restrict__ seq_lens,
                                                       const int* __restrict__ status, float scale,
                                                       float* __restrict__ grad, RowMap2 gmap, int gcols,
                                                       float* __restrict__ loss, int lmax) {
  extern __shared__ __attribute__((aligned(16))) int smem[];
  int* pos_off = smem;                 // [32]
  int* pos_list = smem + 32;           // [lmax]
  float* wbuf = reinterpret_cast<float*>(smem + 32 + lmax);   // [4][lmax]
  const int UP = KPL * 64;
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int* lab = label_ids + label_off[b];
  const int L = label_off[b + 1] - label_off[b];
  const int U = 2 * L + 1;
  const int Tb = seq_lens[b];
  const bool bad = status[b] != 0;
  const int blank = C - 1;
  auto sidx = [&](int u) { return (u % KPL) * 64 + u / KPL; };     // lane-major state index

  if (wave == 0) {
    // per-class position lists in increasing position order (fixed summation order)
    int cnt = 0;
    if (lane < blank) for (int i = 0; i < L; ++i) cnt += lab[i] == lane;
    int inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
    int start = inc - cnt;
    if (lane < 32) pos_off[lane] = lane < blank ? start : L;
    if (lane < blank) { int w = start; for (int i = 0; i < L; ++i) if (lab[i] == lane) pos_list[w++] = i; }
  }
  __syncthreads();

  double logp2 = 0.0;   // log2 p(l|x); stays 0 (p = 1) for an utterance without frames
  if (!bad && Tb > 0) {
    const float* al = alpha + ((long)b * T + (Tb - 1)) * UP;
    logp2 = aoff[(long)b * T + Tb - 1] + (double)lse2_b2(al[sidx(U - 1)], U > 1 ? al[sidx(U - 2)] : NEG_INF);
  }
  if (blockIdx.x == 0 && tid == 0) loss[b] = bad ? __builtin_inff() : (float)(-logp2 * (double)LN2);

  float* wb = wbuf + wave * lmax;
  for (int k = 0; k < GF / 4; ++k) {
    const int t = blockIdx.x * GF + k * 4 + wave;
    const bool live = !bad && t < Tb && t < T;
    float blank_sum = 0.f, occ_scale = 1.f;
    if (live) {
      const float* al = alpha + ((long)b * T + t) * UP;
      const float* be = beta + ((long)b * T + t) * UP;
      const float shift = (float)(aoff[(long)b * T + t] + boff[(long)b * T + t] - logp2);
      float label_sum = 0.f;
      for (int u = lane; u < U; u += 64) {
        const int i = sidx(u);
        float w = ex2(al[i] + be[i] + shift);
        if (u & 1) { wb[u >> 1] = w; label_sum += w; } else blank_sum += w;
      }
      blank_sum = st::wave_sum(blank_sum);
      // sum_u alpha_t(u) beta_t(u) == p(l|x) for EVERY t; the recursions accumulate the (biased)
      // 1-ULP error of v_exp/v_log over hundreds of dependent steps, so the per-frame sum can be off
      // by ~1e-4 relative.  Normalising by the frame's own total removes that common factor.
      const float total = blank_sum + st::wave_sum(label_sum);
      occ_scale = total > 0.f ? 1.f / total : 1.f;
      blank_sum *= occ_scale;
    }
    __syncthreads();
    if (t < T && lane < gcols) {
      float g = 0.f;
      if (live && lane < C) {
        float occ = blank_sum;
        if (lane < blank) {
          occ = 0.f;
          for (int i = pos_off[lane]; i < pos_off[lane + 1]; ++i) occ += wb[pos_list[i]];
          occ *= occ_scale;
        }
        g = (ex2(logy[((long)b * T + t) * CP + lane]) - occ) * scale;
      }
      grad[gmap.off(b, t) + lane] = g;
    }
    __syncthreads();
  }
}

// greedy decode: argmax per frame, collapse repeats, drop blanks; one block per utterance
__global__ __launch_bounds__(256) void ctc_greedy_kernel(const float* __restrict__ logits, RowMap2 map, int T, int C,
                                                         const int* __restrict__ seq_lens, int merge_repeated,
                                                         int* __restrict__ ids, int max_out,
                                                         int* __restrict__ out_lens, float* __restrict__ neg_sum) {
  extern __shared__ int ksm[];          // [T] argmax per frame, then scan scratch [256]
  int* kbuf = ksm;
  int* scan = ksm + T;
  float* fsum = reinterpret_cast<float*>(scan + 256);
  const int b = blockIdx.x, tid = threadIdx.x;
  const int Tb = min(seq_lens[b], T);
  const int blank = C - 1;
  float msum = 0.f;
  for (int t = tid; t < Tb; t += 256) {
    const float* row = logits + map.off(b, t);
    float best = row[0];
    int bi = 0;
    for (int c = 1; c < C; ++c) { float v = row[c]; if (v > best) { best = v; bi = c; } }   // first max wins
    kbuf[t] = bi;
    msum += best;
  }
  __syncthreads();
  // contiguous segment per thread so positions stay ordered
  const int seg = (Tb + 255) / 256;
  const int lo = min(tid * seg, Tb), hi = min(lo + seg, Tb);
  int cnt = 0;
  for (int t = lo; t < hi; ++t) {
    int k = kbuf[t];
    cnt += (k != blank) && !(merge_repeated && t > 0 && k == kbuf[t - 1]);
  }
  scan[tid] = cnt;
  fsum[tid] = msum;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    int v = tid >= o ? scan[tid - o] : 0;
    float f = tid + o < 256 ? fsum[tid + o] : 0.f;
    __syncthreads();
    scan[tid] += v;
    fsum[tid] += f;
    __syncthreads();
  }
  int w = scan[tid] - cnt;
  for (int t = lo; t < hi; ++t) {
    int k = kbuf[t];
    if ((k != blank) && !(merge_repeated && t > 0 && k == kbuf[t - 1])) {
      if (w < max_out) ids[(long)b * max_out + w] = k;
      ++w;
    }
  }
  if (tid == 255) out_lens[b] = scan[255];
  if (tid == 0) neg_sum[b] = -fsum[0];
}

int pick_kpl(int max_label_len) {
  static const int opts[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16};
  const int U = 2 * max_label_len + 1;
  for (int k : opts) if (k * 64 >= U) return k;
  return -1;
}

RowMap2 make_map2(const st_tensor3& t) {
  RowMap2 m;
  m.batch_stride = (long)t.t_pitch * t.c_pitch;
  m.row0 = (long)t.halo * t.c_pitch;
  m.row_stride = t.c_pitch;
  return m;
}

template <int KPL>
void launch_ab(int B, hipStream_t s, const float* logy, int T, int C, const int* ids, const int* off,
               const int* lens, float* alpha, float* beta, double* aoff, double* boff, int* status) {
  hipLaunchKernelGGL((ctc_alpha_beta_kernel<KPL>), dim3(B, 2), dim3(64), 0, s, logy, T, C, ids, off, lens,
                     alpha, beta, aoff, boff, status);
}

}  // namespace

extern "C" {

size_t st_ctc_ws(int batch, int frames, int max_label_len) {
  int kpl = pick_kpl(std::max(max_label_len, 0));
  if (kpl < 0 || batch <= 0 || frames <= 0) return 0;
  size_t rows = (size_t)batch * frames;
  return rows * CP * sizeof(float) + 2 * rows * kpl * 64 * sizeof(float) + 2 * rows * sizeof(double) + 512;
}

int st_ctc_loss_grad_f32(const st_tensor3* logits, const int32_t* label_ids, const int32_t* label_offsets,
                         const int32_t* seq_lens, int max_label_len, float grad_scale, float* loss,
                         const st_tensor3* grad, int32_t* status, void* workspace, size_t workspace_bytes,
                         void* stream) {
  ST_REQUIRE(logits && logits->base && grad && grad->base && label_ids && label_offsets && seq_lens && loss &&
                 status && workspace, "ctc: null argument");
  ST_REQUIRE(logits->channels >= 2 && logits->channels <= CP && logits->c_pitch >= CP && logits->c_pitch % 4 == 0,
             "ctc: num_classes must be 2..32 with c_pitch >= 32");
  ST_REQUIRE(grad->batch == logits->batch && grad->frames == logits->frames && grad->c_pitch >= logits->channels,
             "ctc: grad tensor mismatch");
  const int kpl = pick_kpl(max_label_len);
  ST_REQUIRE(kpl > 0, "ctc: label length %d exceeds 511", max_label_len);
  ST_REQUIRE(workspace_bytes >= st_ctc_ws(logits->batch, logits->frames, max_label_len), "ctc: workspace too small");
  hipStream_t s = st::as_stream(stream);
  const int B = logits->batch, T = logits->frames, C = logits->channels;
  const size_t rows = (size_t)B * T;
  float* logy = reinterpret_cast<float*>(workspace);
  float* alpha = logy + rows * CP;
  float* beta = alpha + rows * kpl * 64;
  double* aoff = reinterpret_cast<double*>(beta + rows * kpl * 64);   // 8-byte aligned: all counts are even
  double* boff = aoff + rows;
  hipLaunchKernelGGL(ctc_logsoftmax_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, logits->base,
                     make_map2(*logits), B, T, C, logy);
  switch (kpl) {
#define ST_AB(K) case K: launch_ab<K>(B, s, logy, T, C, label_ids, label_offsets, seq_lens, alpha, beta, aoff, boff, status); break;
    ST_AB(1) ST_AB(2) ST_AB(3) ST_AB(4) ST_AB(5) ST_AB(6) ST_AB(8) ST_AB(10) ST_AB(12) ST_AB(16)
#undef ST_AB
  }
  if (int e = st::check_launch("ctc_alpha_beta")) return e;
  const int lmax = std::max(1, kpl * 32);
  const size_t shm = (32 + (size_t)lmax * 5) * sizeof(int);
  hipLaunchKernelGGL(ctc_grad_kernel, dim3(st::ceil_div(T, GF), B), dim3(256), shm, s, logy, alpha, beta, aoff, boff,
                     T, C, kpl, label_ids, label_offsets, seq_lens, status, grad_scale, grad->base, make_map2(*grad),
                     std::min(grad->c_pitch, CP), loss, lmax);
  return st::check_launch("ctc_grad");
}

int st_ctc_greedy_decode(const st_tensor3* logits, const int32_t* seq_lens, int merge_repeated, int32_t* ids,
                         int max_out, int32_t* out_lens, float* neg_sum_logits, void* stream) {
  ST_REQUIRE(logits && logits->base && seq_lens && ids && out_lens && neg_sum_logits, "greedy: null argument");
  ST_REQUIRE(logits->channels >= 2 && max_out >= 1, "greedy: bad shape");
  ST_REQUIRE(logits->frames <= 12000, "greedy: more than 12000 frames per utterance not supported");
  const size_t shm = ((size_t)logits->frames + 512) * sizeof(int);
  hipLaunchKernelGGL(ctc_greedy_kernel, dim3(logits->batch), dim3(256), shm, st::as_stream(stream), logits->base,
                     make_map2(*logits), logits->frames, logits->channels, seq_lens, merge_repeated, ids, max_out,
                     out_lens, neg_sum_logits);
  return st::check_launch("ctc_greedy");
}

}  // extern "C"
