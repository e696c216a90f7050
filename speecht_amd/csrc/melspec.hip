// Mel-filterbank feature extractor on gfx950: calc_power_spectrogram (preprocessing.py:36-58).
//
//   librosa.feature.melspectrogram(y, sr, n_mels, n_fft=512, hop_length=160)   (:50)
//   -> librosa.power_to_db(S, ref=np.max)                                      (:53)
//   -> normalize: (x - mean) / std over the whole matrix                       (:29-33, :56)
//   -> transpose to [time, n_mels]                                             (:58)
// Semantics: SURVEY Appendix A7-A9 (center=True reflect padding, periodic Hann, power 2,
// amin 1e-10, top_db 80, population std).
//
// mel_plan_kernel   : once per (sample rate, n_mels): the sparse triangular filterbank cut into pieces of <= 16
//                     bins (~2*257 non-zeros in an [n_mels x 257] basis) + every per-lane constant of the FFT.
// mel_pair_kernel   : a WAVE transforms two consecutive frames with one 512-point complex FFT (frame 2p in the
//                     real part, 2p+1 in the imaginary part; csrc/mel_fft.h): three radix-8 Stockham stages, 8
//                     points per lane in registers, two exchanges through a wave-private LDS buffer, no workgroup
//                     barrier in the frame loop.  Power spectra of both frames from Z[k] and Z[512-k] (one cross-
//                     lane fetch), sparse mel projection from the plan, log2 of the mel power written as one
//                     coalesced run, the wave's maximum into its own slot (no atomics, no memset).
// mel_stats_kernel  : dB relative to the utterance maximum + -80 dB floor, per-utterance sum / sum of squares in
//                     double, fixed-shape partials (deterministic).
// mel_finish_kernel : mean / population std from the partials, normalised write in the reference's
//                     [time, n_mels] layout.
// Roofline: 640 KB in + 320 KB out per 10 s utterance (0.96 MB algorithmic; HBM floor ~0.12 us per utterance);
// in practice bound by instruction issue of the FFT + projection, reported separately from the training step.
#include <algorithm>

#include "mel_fft.h"
#include "st_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NFFT = 512;
constexpr int NBINS = NFFT / 2 + 1;
constexpr int STAT_CHUNKS = 64; // partial sums per utterance
constexpr float LOG2_AMIN = -33.219280948873624f;    // log2(1e-10): power_to_db's amin
constexpr float DB_PER_LOG2 = 3.0102999566398120f;   // 10 * log10(2)

constexpr int PLAN_PIECE = 16;                  // bins per piece
constexpr int PLAN_MAX_ROUNDS = 12;             // 64 pieces per round; n_mels <= 256 needs <= 2 * (256 + 34) pieces
constexpr int PAIRS_PER_WAVE = 4;
constexpr int MAX_OUT_PER_LANE = 8;             // 2 * 256 mel values of a frame pair over 64 lanes

struct PlanItem { int mel, sel, k_lo, n; };     // mel < 0: idle slot
// Device-side plan (st_melspec_plan_f32): the sparse filterbank as pieces + every per-lane constant of the
// transform, so that the frame kernel starts without a single transcendental.
struct Plan {
  int rounds, n_mels, per_frame, pad;           // per_frame = pieces of ONE frame; slot = sel * per_frame + piece
  PlanItem items[PLAN_MAX_ROUNDS * 64];
  float weights[PLAN_MAX_ROUNDS * 64][PLAN_PIECE];
  int oinfo[2 * 256];                           // output sel * n_mels + m: (first slot << 8) | number of pieces
  float win[NFFT];                              // periodic Hann
  melfft::cf tw1[8][64];                        // [r][lane]: W64^((lane & 7) r)
  melfft::cf tw2[8][64];                        // [r][lane]: W512^(lane r)
};

__global__ __launch_bounds__(256) void mel_plan_kernel(const float* __restrict__ basis, int n_mels, Plan* __restrict__ plan) {
  __shared__ int lo_s[256], hi_s[256], first_s[257];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int m = wave; m < n_mels; m += 4) {                 // first / last non-zero bin of row m by ballots
    int lo = NBINS, hi = 0;
    for (int k0 = 0; k0 < NBINS; k0 += 64) {
      const int k = k0 + lane;
      const unsigned long long nz = __ballot(k < NBINS && basis[(long)m * NBINS + k] != 0.f);
      if (nz) {
        lo = min(lo, k0 + __ffsll((long long)nz) - 1);
        hi = max(hi, k0 + 64 - __clzll((long long)nz));
      }
    }
    if (lane == 0) { lo_s[m] = min(lo, hi); hi_s[m] = hi; }
  }
  __syncthreads();
  if (tid == 0) {
    int total = 0;
    for (int m = 0; m < n_mels; ++m) { first_s[m] = total; total += max(1, (hi_s[m] - lo_s[m] + PLAN_PIECE - 1) / PLAN_PIECE); }
    first_s[n_mels] = total;
    plan->rounds = (2 * total + 63) / 64;
    plan->n_mels = n_mels;
    plan->per_frame = total;
    plan->pad = 0;
  }
  __syncthreads();
  const int per_frame = first_s[n_mels];
  const int rounds = (2 * per_frame + 63) / 64;
  for (int slot = tid; slot < rounds * 64; slot += 256) {
    PlanItem it{-1, 0, 0, 0};
    const int sel = slot >= per_frame ? 1 : 0, piece = slot - sel * per_frame;
    if (piece < per_frame) {
      int m = 0;
      while (first_s[m + 1] <= piece) ++m;                  // n_mels <= 256: a short scan, once per plan
      const int k_lo = lo_s[m] + (piece - first_s[m]) * PLAN_PIECE;
      it = PlanItem{m, sel, k_lo, max(0, min(PLAN_PIECE, hi_s[m] - k_lo))};
    }
    plan->items[slot] = it;
    for (int i = 0; i < PLAN_PIECE; ++i)
      plan->weights[slot][i] = (it.mel >= 0 && i < it.n) ? basis[(long)it.mel * NBINS + it.k_lo + i] : 0.f;
  }
  for (int i = tid; i < 2 * n_mels; i += 256) {
    const int sel = i >= n_mels ? 1 : 0, m = i - sel * n_mels;
    plan->oinfo[i] = ((sel * per_frame + first_s[m]) << 8) | (first_s[m + 1] - first_s[m]);
  }
  for (int k = tid; k < NFFT; k += 256) plan->win[k] = 0.5f - 0.5f * cospif(2.0f * (float)k / (float)NFFT);
  for (int i = tid; i < 8 * 64; i += 256) {
    const int r = i >> 6, l = i & 63;
    sincospif(-2.0f * (float)((l & 7) * r) / 64.0f, &plan->tw1[r][l].y, &plan->tw1[r][l].x);
    sincospif(-2.0f * (float)(l * r) / 512.0f, &plan->tw2[r][l].y, &plan->tw2[r][l].x);
  }
}

__device__ __forceinline__ void wave_lds_sync() {
  // LDS traffic of one wave is executed in order; this only keeps the compiler from moving accesses across
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int reflect_index(int j, int n) {
  if (j < 0) j = -j;
  if (j >= n) j = 2 * (n - 1) - j;
  return min(max(j, 0), n - 1);
}

// number of per-wave maximum slots an utterance owns (a function of the launch grid only)
__host__ __device__ inline int wave_slots(int max_frames) {
  return ((max_frames + 1) / 2 + 4 * PAIRS_PER_WAVE - 1) / (4 * PAIRS_PER_WAVE) * 4;
}

__global__ __launch_bounds__(256, 4) void mel_pair_kernel(const float* __restrict__ audio,
                                                          const long* __restrict__ sample_off,
                                                          const Plan* __restrict__ plan, int n_mels, int hop,
                                                          const long* __restrict__ frame_off,
                                                          float* __restrict__ logpow, float* __restrict__ wmax,
                                                          double* __restrict__ wstat) {
  // wstat (nullable; the power-spectrogram path): per wave slot {min, sum, sum of squares} of the log2 values it wrote -- with
  // the maximum they give the utterance's mean / variance in closed form whenever power_to_db's 80 dB floor touches no element
  // (normalize(), preprocessing.py:29-33), so the statistics pass over the matrix only runs for utterances it does touch
  using melfft::cf;
  __shared__ cf zbuf[4][melfft::LDS_COMPLEX];
  __shared__ float pw[4][2][NBINS + PLAN_PIECE - 1];       // + read-ahead pad (zero weights meet zero values there)
  __shared__ cf tw1_s[8][64], tw2_s[8][64];                // twiddles of stage 1 / 2, [r][lane]
  const int u = blockIdx.y;
  const long s0 = sample_off[u];
  const int n = (int)(sample_off[u + 1] - s0);
  const int frames = 1 + n / hop;
  const int pairs = (frames + 1) / 2;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int p_begin = (blockIdx.x * 4 + wave) * PAIRS_PER_WAVE;
  float* my_max = wmax + (long)u * (gridDim.x * 4) + blockIdx.x * 4 + wave;
  double* my_stat = wstat ? wstat + ((long)u * (gridDim.x * 4) + blockIdx.x * 4 + wave) * 3 : nullptr;
  auto idle_slot = [&]() {
    if (lane == 0) {
      *my_max = LOG2_AMIN;
      if (my_stat) { my_stat[0] = 1e30; my_stat[1] = 0.0; my_stat[2] = 0.0; }
    }
  };
  if (blockIdx.x * 4 * PAIRS_PER_WAVE >= pairs) {          // whole workgroup past the end of a short utterance
    idle_slot();
    return;
  }
  for (int i = tid; i < 8 * 64; i += 256) {
    (&tw1_s[0][0])[i] = (&plan->tw1[0][0])[i];
    (&tw2_s[0][0])[i] = (&plan->tw2[0][0])[i];
  }
  __syncthreads();                                         // the only workgroup barrier
  if (p_begin >= pairs) {
    idle_slot();
    return;
  }
  const float* y = audio + s0;
  const int rounds = plan->rounds;
  const int out_iters = (2 * n_mels + 63) >> 6;
  if (lane < 2 * (PLAN_PIECE - 1)) pw[wave][lane / (PLAN_PIECE - 1)][NBINS + lane % (PLAN_PIECE - 1)] = 0.f;
  float win[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) win[r] = plan->win[lane + 64 * r];
  int oinfo[MAX_OUT_PER_LANE];
#pragma unroll
  for (int j = 0; j < MAX_OUT_PER_LANE; ++j) oinfo[j] = lane + 64 * j < 2 * n_mels ? plan->oinfo[lane + 64 * j] : 0;
  cf* zb = zbuf[wave];
  float* part = reinterpret_cast<float*>(zb);              // piece sums: the FFT buffer is idle during the projection
  float vmax = LOG2_AMIN, vmin = 1e30f;
  double sum1 = 0.0, sum2 = 0.0;
  const int p_end = min(pairs, p_begin + PAIRS_PER_WAVE);
  // raw samples of a frame pair: frame 2p -> xa, frame 2p + 1 -> xb; centre=True: padded index t*hop + k  <->  sample
  // t*hop + k - NFFT/2, reflected at the ends.  Loaded one pair ahead so that the HBM/L2 latency hides behind the
  // previous pair's FFT.
  float xa[8], xb[8];
  auto load_pair = [&](int p) {
    const int j0 = 2 * p * hop - NFFT / 2;
    if (j0 >= 0 && j0 + hop + NFFT <= n) {                 // both frames inside the utterance (wave-uniform)
      const float* src = y + j0 + lane;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        xa[r] = src[64 * r];
        xb[r] = src[64 * r + hop];
      }
    } else {
      const bool has_b = 2 * p + 1 < frames;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int k = lane + 64 * r;
        xa[r] = y[reflect_index(j0 + k, n)];
        xb[r] = has_b ? y[reflect_index(j0 + hop + k, n)] : 0.f;
      }
    }
  };
  load_pair(p_begin);
  for (int p = p_begin; p < p_end; ++p) {
    const int ta = 2 * p, tb = ta + 1;
    cf v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      v[r].x = xa[r] * win[r];
      v[r].y = xb[r] * win[r];
    }
    if (p + 1 < p_end) load_pair(p + 1);
    melfft::dft8(v);
#pragma unroll
    for (int r = 0; r < 8; ++r) zb[melfft::pad(melfft::out_index(0, lane, r))] = v[r];
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const cf z = zb[melfft::pad(lane + 64 * r)];
      v[r] = r ? melfft::zmul(z, tw1_s[r][lane]) : z;
    }
    wave_lds_sync();
    melfft::dft8(v);
#pragma unroll
    for (int r = 0; r < 8; ++r) zb[melfft::pad(melfft::out_index(1, lane, r))] = v[r];
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const cf z = zb[melfft::pad(lane + 64 * r)];
      v[r] = r ? melfft::zmul(z, tw2_s[r][lane]) : z;
    }
    melfft::dft8(v);                                       // Z[lane + 64 r] = v[r]
    // Z[512 - k] for k = lane + 64 r: lane (64 - lane) & 63, register 7 - r (lane 0: register 8 - r, own Z[0] for r = 0)
    const int partner = (64 - lane) & 63;
    cf q[4];                                               // partner registers 4..7
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      q[r].x = __shfl(v[4 + r].x, partner, 64);
      q[r].y = __shfl(v[4 + r].y, partner, 64);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // lanes >= 1 pair register r with partner register 7 - r = q[3 - r]; lane 0 with 8 - r = q[4 - r] (r = 0: itself)
      cf b = q[3 - r];
      if (lane == 0) b = r == 0 ? v[0] : q[4 - r];
      float pa, pb;
      melfft::pair_power(v[r], b, &pa, &pb);
      pw[wave][0][lane + 64 * r] = pa;
      pw[wave][1][lane + 64 * r] = pb;
    }
    if (lane == 0) {                                       // k = 256 pairs with itself
      float pa, pb;
      melfft::pair_power(v[4], v[4], &pa, &pb);
      pw[wave][0][256] = pa;
      pw[wave][1][256] = pb;
    }
    wave_lds_sync();
    for (int round = 0; round < rounds; ++round) {
      const int slot = round * 64 + lane;
      const PlanItem it = plan->items[slot];
      const f32x4* w4 = reinterpret_cast<const f32x4*>(plan->weights[slot]);
      const f32x4 w0 = w4[0], w1 = w4[1], w2 = w4[2], w3 = w4[3];      // zero beyond the piece (and for idle slots)
      const float* src = &pw[wave][it.sel][it.k_lo];
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        sum = fmaf(w0[i], src[i], sum);
        sum = fmaf(w1[i], src[4 + i], sum);
        sum = fmaf(w2[i], src[8 + i], sum);
        sum = fmaf(w3[i], src[12 + i], sum);
      }
      part[slot] = sum;
    }
    wave_lds_sync();
    // both mel rows are adjacent in memory (frames ta and ta + 1 of the same utterance); a filter's pieces are
    // summed in plan order (deterministic); stored as log2(max(amin, power)): the dB chain downstream is affine in it
    float* dst = logpow + (frame_off[u] + ta) * (long)n_mels;
    const int valid = (tb < frames ? 2 : 1) * n_mels;
#pragma unroll
    for (int j = 0; j < MAX_OUT_PER_LANE; ++j) {
      if (j < out_iters) {
        const int i = lane + 64 * j;
        const int first = oinfo[j] >> 8, count = oinfo[j] & 255;
        float a = part[first];
        if (count > 1) a += part[first + 1];
        for (int c = 2; c < count; ++c) a += part[first + c];
        const float l2 = fmaxf(__log2f(a), LOG2_AMIN);     // log2(max(1e-10, a)); a >= 0
        if (i < valid) {
          dst[i] = l2;
          vmax = fmaxf(vmax, l2);
          vmin = fminf(vmin, l2);
          sum1 += (double)l2;
          sum2 += (double)l2 * (double)l2;
        }
      }
    }
    wave_lds_sync();
  }
  vmax = st::wave_max(vmax);
  if (lane == 0) *my_max = vmax;
  if (my_stat) {
    vmin = -st::wave_max(-vmin);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {                       // fixed-shape tree: deterministic
      sum1 += __shfl_xor(sum1, o, 64);
      sum2 += __shfl_xor(sum2, o, 64);
    }
    if (lane == 0) { my_stat[0] = (double)vmin; my_stat[1] = sum1; my_stat[2] = sum2; }
  }
}

__device__ __forceinline__ double block_sum_d(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
  __syncthreads();
  return t;
}

// log2 of the utterance's largest mel power: maximum over the per-wave slots, every thread of the block gets it
__device__ __forceinline__ float utterance_max(const float* __restrict__ wmax, int u, int slots, float* red) {
  float m = LOG2_AMIN;
  for (int i = threadIdx.x; i < slots; i += blockDim.x) m = fmaxf(m, wmax[(long)u * slots + i]);
  m = st::wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = red[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, red[i]);
  __syncthreads();
  return m;
}

// {max, min, sum, sum of squares} of the utterance's log2 values from the per-wave slots: every WAVE walks all slots in the same
// fixed order (identical values in every wave of every workgroup), no LDS, no barrier -- the 64 workgroups an utterance has in
// mel_stats_kernel / mel_finish_kernel each pay this once, and for the statistics pass it is all an unfloored utterance costs
// (with block-wide reductions behind barriers it was 2 us per workgroup: 73 us of nothing at batch 512)
__device__ __forceinline__ void utterance_stats(const float* __restrict__ wmax, const double* __restrict__ wstat, int u, int slots,
                                                float* mx, double* mn, double* s1, double* s2) {
  const int lane = threadIdx.x & 63;
  float m = LOG2_AMIN;
  double a = 1e30, b = 0.0, c = 0.0;
  for (int i = lane; i < slots; i += 64) {
    const double* w = wstat + ((long)u * slots + i) * 3;
    m = fmaxf(m, wmax[(long)u * slots + i]);
    a = fmin(a, w[0]); b += w[1]; c += w[2];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    m = fmaxf(m, __shfl_xor(m, o, 64));
    a = fmin(a, __shfl_xor(a, o, 64));
    b += __shfl_xor(b, o, 64);
    c += __shfl_xor(c, o, 64);
  }
  *mx = m; *mn = a; *s1 = b; *s2 = c;
}

// power_to_db(ref = max, amin 1e-10, top_db 80) from log2 values: 10 log10(x) = DB_PER_LOG2 * log2(x); the maximum of
// the dB matrix is 0 by construction, so the floor is -80
__device__ __forceinline__ float to_db(float l2, float l2_max) { return fmaxf(DB_PER_LOG2 * (l2 - l2_max), -80.f); }

__global__ __launch_bounds__(256) void mel_stats_kernel(const float* __restrict__ logpow,
                                                        const long* __restrict__ sample_off,
                                                        const long* __restrict__ frame_off, int n_mels, int hop,
                                                        const float* __restrict__ wmax, int slots,
                                                        const double* __restrict__ wstat, double* __restrict__ partial) {
  __shared__ double red[4];
  const int u = blockIdx.y;
  const int n = (int)(sample_off[u + 1] - sample_off[u]);
  const long count = (long)(1 + n / hop) * n_mels;
  const float* src = logpow + frame_off[u] * (long)n_mels;
  float l2_max;
  {
    // the smallest element stays above the floor -> so does every element (to_db is monotone): mel_finish_kernel takes the closed
    // form from the per-wave sums and this pass has nothing to do (the same value in every wave: a uniform exit)
    double mn, s1, s2;
    utterance_stats(wmax, wstat, u, slots, &l2_max, &mn, &s1, &s2);
    if (DB_PER_LOG2 * ((float)mn - l2_max) >= -80.f) return;
  }
  const long per = (count + STAT_CHUNKS - 1) / STAT_CHUNKS;
  const long lo = blockIdx.x * per, hi = min(count, lo + per);
  double s = 0.0, ss = 0.0;
  if ((((uintptr_t)src | (uintptr_t)(lo * 4)) & 15) == 0) {             // 16-byte chunks (fixed order per thread)
    const long n4 = (hi - lo) / 4;
    const f32x4* src4 = reinterpret_cast<const f32x4*>(src + lo);
    for (long i = threadIdx.x; i < n4; i += 256) {
      const f32x4 v = src4[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double d = (double)to_db(v[e], l2_max);
        s += d;
        ss += d * d;
      }
    }
    for (long i = lo + n4 * 4 + threadIdx.x; i < hi; i += 256) {
      const double d = (double)to_db(src[i], l2_max);
      s += d;
      ss += d * d;
    }
  } else {
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
      const double d = (double)to_db(src[i], l2_max);
      s += d;
      ss += d * d;
    }
  }
  s = block_sum_d(s, red);
  ss = block_sum_d(ss, red);
  if (threadIdx.x == 0) {
    partial[((long)u * STAT_CHUNKS + blockIdx.x) * 2] = s;
    partial[((long)u * STAT_CHUNKS + blockIdx.x) * 2 + 1] = ss;
  }
}

__global__ __launch_bounds__(256) void mel_finish_kernel(const float* __restrict__ logpow,
                                                         const long* __restrict__ sample_off,
                                                         const long* __restrict__ frame_off, int n_mels, int hop,
                                                         const float* __restrict__ wmax, int slots,
                                                         const double* __restrict__ wstat, const double* __restrict__ partial,
                                                         float* __restrict__ out) {
  __shared__ double tot[2];
  const int u = blockIdx.y;
  const int n = (int)(sample_off[u + 1] - sample_off[u]);
  const long count = (long)(1 + n / hop) * n_mels;
  static_assert(STAT_CHUNKS == 64, "one partial per lane of the first wave");
  float l2_max_early;
  double mn, w1, w2;
  utterance_stats(wmax, wstat, u, slots, &l2_max_early, &mn, &w1, &w2);
  const bool unfloored = DB_PER_LOG2 * ((float)mn - l2_max_early) >= -80.f;      // (the test mel_stats_kernel exits on)
  if (unfloored) {
    // d = c (l - M) for every element: sum d = c (S1 - N M), sum d^2 = c^2 (S2 - 2 M S1 + N M^2), in double
    if (threadIdx.x == 0) {
      const double c = (double)DB_PER_LOG2, M = (double)l2_max_early, N = (double)count;
      tot[0] = c * (w1 - N * M);
      tot[1] = c * c * (w2 - 2.0 * M * w1 + N * M * M);
    }
  } else if (threadIdx.x < 64) {                   // same fixed-shape tree in every block: deterministic
    double s = partial[((long)u * STAT_CHUNKS + threadIdx.x) * 2];
    double ss = partial[((long)u * STAT_CHUNKS + threadIdx.x) * 2 + 1];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s += __shfl_xor(s, o, 64);
      ss += __shfl_xor(ss, o, 64);
    }
    if (threadIdx.x == 0) { tot[0] = s; tot[1] = ss; }
  }
  __syncthreads();
  const double s = tot[0], ss = tot[1];
  const double mean = s / (double)count;
  const double var = fmax(ss / (double)count - mean * mean, 0.0);
  const float inv_std = (float)(1.0 / sqrt(var));
  const float meanf = (float)mean;
  const float* src = logpow + frame_off[u] * (long)n_mels;
  float* dst = out + frame_off[u] * (long)n_mels;
  const float l2_max = l2_max_early;
  const long per = (count + STAT_CHUNKS - 1) / STAT_CHUNKS;
  const long lo = blockIdx.x * per, hi = min(count, lo + per);
  if ((((uintptr_t)src | (uintptr_t)dst | (uintptr_t)(lo * 4)) & 15) == 0) {
    const long n4 = (hi - lo) / 4;
    const f32x4* src4 = reinterpret_cast<const f32x4*>(src + lo);
    f32x4* dst4 = reinterpret_cast<f32x4*>(dst + lo);
    for (long i = threadIdx.x; i < n4; i += 256) {
      f32x4 v = src4[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (to_db(v[e], l2_max) - meanf) * inv_std;
      dst4[i] = v;
    }
    for (long i = lo + n4 * 4 + threadIdx.x; i < hi; i += 256) dst[i] = (to_db(src[i], l2_max) - meanf) * inv_std;
  } else {
    for (long i = lo + threadIdx.x; i < hi; i += 256) dst[i] = (to_db(src[i], l2_max) - meanf) * inv_std;
  }
}

// ---- MFCC + delta + delta-delta (preprocessing.py:61-84) --------------------------------------------
// librosa.feature.mfcc = DCT-II (orthonormal) of power_to_db(mel power, ref = 1.0, top_db = 80);
// librosa.feature.delta (0.5.x: FIR [4..-4]/60 run causally from rest over the edge-padded signal, once or
// twice); each of the three [n_mfcc, T] blocks is z-normalised on its own.  Reuses the plan and mel_pair_kernel.
constexpr int MAX_MFCC = 32;

// one wave per frame: dB of the mel bins in registers, n_mfcc cosine projections, wave reduction
__global__ __launch_bounds__(256) void mfcc_dct_kernel(const float* __restrict__ logpow,
                                                       const long* __restrict__ sample_off,
                                                       const long* __restrict__ frame_off, int n_mels, int n_mfcc,
                                                       int hop, const float* __restrict__ wmax, int slots,
                                                       float* __restrict__ coef) {
  __shared__ float redf[4];
  const int u = blockIdx.y;
  const int frames = 1 + (int)(sample_off[u + 1] - sample_off[u]) / hop;
  const float floor_db = DB_PER_LOG2 * utterance_max(wmax, u, slots, redf) - 80.f;     // ref = 1.0, top_db = 80
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= frames) return;
  const int lane = threadIdx.x & 63;
  const float* src = logpow + (frame_off[u] + t) * (long)n_mels;
  float db[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = lane + 64 * j;
    db[j] = m < n_mels ? fmaxf(DB_PER_LOG2 * src[m], floor_db) : 0.f;
  }
  const float inv2n = 0.5f / (float)n_mels;
  for (int c = 0; c < n_mfcc; ++c) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = fmaf(db[j], cospif((float)(c * (2 * (lane + 64 * j) + 1)) * inv2n), acc);
    acc = st::wave_sum(acc);
    if (lane == 0) coef[(frame_off[u] + t) * (long)n_mfcc + c] = acc * (c ? sqrtf(2.f / n_mels) : rsqrtf((float)n_mels));
  }
}

// x edge-padded by 9 frames on both sides; index j of the padded axis, zero before it starts (filter at rest)
__device__ __forceinline__ float mfcc_padded(const float* __restrict__ x, int n_mfcc, int frames, int c, int j) {
  return j < 0 ? 0.f : x[(long)min(max(j - 9, 0), frames - 1) * n_mfcc + c];
}
__device__ __forceinline__ float mfcc_delta1(const float* __restrict__ x, int n_mfcc, int frames, int c, int j) {
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) acc = fmaf((float)(4 - k) * (1.f / 60.f), mfcc_padded(x, n_mfcc, frames, c, j - k), acc);
  return j < 0 ? 0.f : acc;
}

__global__ __launch_bounds__(256) void mfcc_delta_kernel(const float* __restrict__ coef,
                                                         const long* __restrict__ sample_off,
                                                         const long* __restrict__ frame_off, int n_mfcc, int hop,
                                                         float* __restrict__ d1, float* __restrict__ d2,
                                                         double* __restrict__ partial) {
  __shared__ double red[4];
  const int u = blockIdx.y;
  const int frames = 1 + (int)(sample_off[u + 1] - sample_off[u]) / hop;
  const long count = (long)frames * n_mfcc;
  const float* x = coef + frame_off[u] * (long)n_mfcc;
  const long per = (count + STAT_CHUNKS - 1) / STAT_CHUNKS;
  const long lo = blockIdx.x * per, hi = min(count, lo + per);
  double s[3] = {0, 0, 0}, ss[3] = {0, 0, 0};
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
    const int t = (int)(i / n_mfcc), c = (int)(i - (long)t * n_mfcc);
    const float v0 = x[i];
    const float v1 = mfcc_delta1(x, n_mfcc, frames, c, 13 + t);
    float v2 = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) v2 = fmaf((float)(4 - k) * (1.f / 60.f), mfcc_delta1(x, n_mfcc, frames, c, 13 + t - k), v2);
    d1[frame_off[u] * (long)n_mfcc + i] = v1;
    d2[frame_off[u] * (long)n_mfcc + i] = v2;
    s[0] += v0; ss[0] += (double)v0 * v0;
    s[1] += v1; ss[1] += (double)v1 * v1;
    s[2] += v2; ss[2] += (double)v2 * v2;
  }
  for (int b = 0; b < 3; ++b) {
    const double a = block_sum_d(s[b], red), q = block_sum_d(ss[b], red);
    if (threadIdx.x == 0) {
      double* dst = partial + (((long)u * 3 + b) * STAT_CHUNKS + blockIdx.x) * 2;
      dst[0] = a;
      dst[1] = q;
    }
  }
}

__global__ __launch_bounds__(256) void mfcc_finish_kernel(const float* __restrict__ coef, const float* __restrict__ d1,
                                                          const float* __restrict__ d2,
                                                          const long* __restrict__ sample_off,
                                                          const long* __restrict__ frame_off, int n_mfcc, int hop,
                                                          const double* __restrict__ partial, float* __restrict__ out) {
  __shared__ float mean_s[3], inv_s[3];
  const int u = blockIdx.y;
  const int frames = 1 + (int)(sample_off[u + 1] - sample_off[u]) / hop;
  const long count = (long)frames * n_mfcc;
  if (threadIdx.x < 3) {
    double s = 0.0, ss = 0.0;
    for (int c = 0; c < STAT_CHUNKS; ++c) {
      s += partial[(((long)u * 3 + threadIdx.x) * STAT_CHUNKS + c) * 2];
      ss += partial[(((long)u * 3 + threadIdx.x) * STAT_CHUNKS + c) * 2 + 1];
    }
    const double mean = s / (double)count;
    mean_s[threadIdx.x] = (float)mean;
    inv_s[threadIdx.x] = (float)(1.0 / sqrt(fmax(ss / (double)count - mean * mean, 0.0)));
  }
  __syncthreads();
  const long base = frame_off[u] * (long)n_mfcc;
  const long per = (count + STAT_CHUNKS - 1) / STAT_CHUNKS;
  const long lo = blockIdx.x * per, hi = min(count, lo + per);
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
    const long t = i / n_mfcc, c = i - t * n_mfcc;
    float* row = out + (frame_off[u] + t) * (long)(3 * n_mfcc);
    row[c] = (coef[base + i] - mean_s[0]) * inv_s[0];
    row[n_mfcc + c] = (d1[base + i] - mean_s[1]) * inv_s[1];
    row[2 * n_mfcc + c] = (d2[base + i] - mean_s[2]) * inv_s[2];
  }
}

size_t pow_bytes(int64_t total_frames, int n_mels) { return st::round_up((size_t)total_frames * n_mels * sizeof(float), 256); }
size_t slot_bytes(int n_utts, int64_t max_samples, int hop) {
  return st::round_up((size_t)n_utts * wave_slots((int)(1 + max_samples / hop)) * sizeof(float), 256);
}
// upper bound for workspace queries that do not know max_samples: one utterance holding every frame
size_t slot_bytes_bound(int n_utts, int64_t total_frames) {
  return st::round_up((size_t)n_utts * wave_slots((int)total_frames) * sizeof(float), 256);
}

// log2 mel power of every frame + per-wave maxima
int launch_frames(const float* audio, const long* soff, int n_utts, int64_t max_samples, const Plan* plan, int n_mels,
                  int hop, const long* foff, float* logpow, float* wmax, hipStream_t s, double* wstat = nullptr) {
  const int max_frames = (int)(1 + max_samples / hop);
  hipLaunchKernelGGL(mel_pair_kernel, dim3(wave_slots(max_frames) / 4, n_utts), dim3(256), 0, s, audio, soff, plan, n_mels,
                     hop, foff, logpow, wmax, wstat);
  return st::check_launch("mel_frames");
}

}  // namespace

extern "C" {

size_t st_melspec_plan_bytes(void) { return st::round_up(sizeof(Plan), 256); }

int st_melspec_plan_f32(const float* mel_basis, int n_mels, int n_fft, void* plan, size_t plan_bytes, void* stream) {
  ST_REQUIRE(mel_basis && plan && plan_bytes >= st_melspec_plan_bytes(), "melspec plan: bad args");
  ST_REQUIRE(n_fft == NFFT && n_mels > 0 && n_mels <= 256, "melspec plan: n_fft must be 512 and n_mels <= 256");
  ST_REQUIRE(((uintptr_t)plan & 15) == 0, "melspec plan: buffer must be 16-byte aligned");
  hipLaunchKernelGGL(mel_plan_kernel, dim3(1), dim3(256), 0, st::as_stream(stream), mel_basis, n_mels,
                     reinterpret_cast<Plan*>(plan));
  return st::check_launch("mel_plan");
}

size_t st_melspec_ws(int n_utts, int64_t total_frames, int n_mels) {
  if (n_utts <= 0 || total_frames <= 0 || n_mels <= 0) return 0;
  // log2 mel powers | per-wave maxima | statistics partials | per-wave {min, sum, sum of squares} (3 doubles per slot) | plan
  return pow_bytes(total_frames, n_mels) + slot_bytes_bound(n_utts, total_frames) +
         (size_t)n_utts * STAT_CHUNKS * 2 * sizeof(double) + 6 * slot_bytes_bound(n_utts, total_frames) + st_melspec_plan_bytes() + 256;
}

int st_melspec_planned_f32(const float* audio, const int64_t* sample_offsets, int n_utts, int64_t max_samples,
                           const void* plan, int n_mels, int n_fft, int hop, const int64_t* frame_offsets,
                           int64_t total_frames, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  ST_REQUIRE(audio && sample_offsets && plan && frame_offsets && out && workspace, "melspec: null argument");
  ST_REQUIRE(n_fft == NFFT, "melspec: only n_fft = 512 (the reference default, preprocessing.py:36) is built");
  ST_REQUIRE(n_utts > 0 && n_mels > 0 && n_mels <= 256 && hop > 0 && max_samples > NFFT / 2 && total_frames > 0 &&
                 1 + max_samples / hop <= total_frames,
             "melspec: bad shape");
  ST_REQUIRE(workspace_bytes >= st_melspec_ws(n_utts, total_frames, n_mels) - st_melspec_plan_bytes() - 256,
             "melspec: workspace too small");
  hipStream_t s = st::as_stream(stream);
  char* w = reinterpret_cast<char*>(workspace);
  float* logpow = reinterpret_cast<float*>(w);
  w += pow_bytes(total_frames, n_mels);
  float* wmax = reinterpret_cast<float*>(w);
  w += slot_bytes(n_utts, max_samples, hop);
  double* partial = reinterpret_cast<double*>(w);
  w += (size_t)n_utts * STAT_CHUNKS * 2 * sizeof(double);
  double* wstat = reinterpret_cast<double*>(w);
  const long* soff = reinterpret_cast<const long*>(sample_offsets);
  const long* foff = reinterpret_cast<const long*>(frame_offsets);
  const int slots = wave_slots((int)(1 + max_samples / hop));
  if (int e = launch_frames(audio, soff, n_utts, max_samples, reinterpret_cast<const Plan*>(plan), n_mels, hop, foff, logpow,
                            wmax, s, wstat))
    return e;
  hipLaunchKernelGGL(mel_stats_kernel, dim3(STAT_CHUNKS, n_utts), dim3(256), 0, s, logpow, soff, foff, n_mels, hop, wmax,
                     slots, wstat, partial);
  hipLaunchKernelGGL(mel_finish_kernel, dim3(STAT_CHUNKS, n_utts), dim3(256), 0, s, logpow, soff, foff, n_mels, hop, wmax,
                     slots, wstat, partial, out);
  return st::check_launch("melspec");
}

int st_melspec_f32(const float* audio, const int64_t* sample_offsets, int n_utts, int64_t max_samples,
                   const float* mel_basis, int n_mels, int n_fft, int hop, const int64_t* frame_offsets,
                   int64_t total_frames, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  ST_REQUIRE(mel_basis && workspace, "melspec: null argument");
  ST_REQUIRE(n_utts > 0 && total_frames > 0 && n_mels > 0, "melspec: bad shape");
  ST_REQUIRE(workspace_bytes >= st_melspec_ws(n_utts, total_frames, n_mels), "melspec: workspace too small");
  // the plan lives at the end of the workspace; everything before it belongs to the planned call
  const size_t front = st_melspec_ws(n_utts, total_frames, n_mels) - st_melspec_plan_bytes() - 256;
  char* plan = reinterpret_cast<char*>(st::round_up((size_t)((uintptr_t)workspace + front), 256));
  if (int e = st_melspec_plan_f32(mel_basis, n_mels, n_fft, plan, st_melspec_plan_bytes(), stream)) return e;
  return st_melspec_planned_f32(audio, sample_offsets, n_utts, max_samples, plan, n_mels, n_fft, hop, frame_offsets,
                                total_frames, out, workspace, front, stream);
}

size_t st_mfcc_ws(int n_utts, int64_t total_frames, int n_mels, int n_mfcc) {
  if (n_utts <= 0 || total_frames <= 0 || n_mels <= 0 || n_mfcc <= 0) return 0;
  return pow_bytes(total_frames, n_mels) + slot_bytes_bound(n_utts, total_frames) + 3 * pow_bytes(total_frames, n_mfcc) +
         (size_t)n_utts * 3 * STAT_CHUNKS * 2 * sizeof(double) + st_melspec_plan_bytes() + 256;
}

int st_mfcc_f32(const float* audio, const int64_t* sample_offsets, int n_utts, int64_t max_samples,
                const float* mel_basis, int n_mels, int n_mfcc, int n_fft, int hop, const int64_t* frame_offsets,
                int64_t total_frames, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  ST_REQUIRE(audio && sample_offsets && mel_basis && frame_offsets && out && workspace, "mfcc: null argument");
  ST_REQUIRE(n_fft == NFFT, "mfcc: only n_fft = 512 (the reference default, preprocessing.py:61) is built");
  ST_REQUIRE(n_utts > 0 && n_mels > 0 && n_mels <= 256 && n_mfcc > 0 && n_mfcc <= MAX_MFCC && n_mfcc <= n_mels &&
                 hop > 0 && max_samples > NFFT / 2 && total_frames > 0 && 1 + max_samples / hop <= total_frames,
             "mfcc: bad shape");
  ST_REQUIRE(workspace_bytes >= st_mfcc_ws(n_utts, total_frames, n_mels, n_mfcc), "mfcc: workspace too small");
  hipStream_t s = st::as_stream(stream);
  char* w = reinterpret_cast<char*>(workspace);
  float* logpow = reinterpret_cast<float*>(w);
  w += pow_bytes(total_frames, n_mels);
  float* wmax = reinterpret_cast<float*>(w);
  w += slot_bytes(n_utts, max_samples, hop);
  float* coef = reinterpret_cast<float*>(w);
  float* d1 = reinterpret_cast<float*>(w + pow_bytes(total_frames, n_mfcc));
  float* d2 = reinterpret_cast<float*>(w + 2 * pow_bytes(total_frames, n_mfcc));
  w += 3 * pow_bytes(total_frames, n_mfcc);
  double* partial = reinterpret_cast<double*>(w);
  w += (size_t)n_utts * 3 * STAT_CHUNKS * 2 * sizeof(double);
  char* plan = reinterpret_cast<char*>(st::round_up((size_t)(uintptr_t)w, 256));
  if (int e = st_melspec_plan_f32(mel_basis, n_mels, n_fft, plan, st_melspec_plan_bytes(), stream)) return e;
  const long* soff = reinterpret_cast<const long*>(sample_offsets);
  const long* foff = reinterpret_cast<const long*>(frame_offsets);
  const int max_frames = (int)(1 + max_samples / hop);
  const int slots = wave_slots(max_frames);
  if (int e = launch_frames(audio, soff, n_utts, max_samples, reinterpret_cast<const Plan*>(plan), n_mels, hop, foff, logpow,
                            wmax, s))
    return e;
  hipLaunchKernelGGL(mfcc_dct_kernel, dim3(st::ceil_div(max_frames, 4), n_utts), dim3(256), 0, s, logpow, soff, foff,
                     n_mels, n_mfcc, hop, wmax, slots, coef);
  hipLaunchKernelGGL(mfcc_delta_kernel, dim3(STAT_CHUNKS, n_utts), dim3(256), 0, s, coef, soff, foff, n_mfcc, hop, d1, d2,
                     partial);
  hipLaunchKernelGGL(mfcc_finish_kernel, dim3(STAT_CHUNKS, n_utts), dim3(256), 0, s, coef, d1, d2, soff, foff, n_mfcc, hop,
                     partial, out);
  return st::check_launch("mfcc");
}

}  // extern "C"
