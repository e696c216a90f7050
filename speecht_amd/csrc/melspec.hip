// Mel-filterbank feature extractor on gfx950: calc_power_spectrogram (preprocessing.py:36-58).
//
//   librosa.feature.melspectrogram(y, sr, n_mels, n_fft=512, hop_length=160)   (:50)
//   -> librosa.power_to_db(S, ref=np.max)                                      (:53)
//   -> normalize: (x - mean) / std over the whole matrix                       (:29-33, :56)
//   -> transpose to [time, n_mels]                                             (:58)
// Semantics: SURVEY Appendix A7-A9 (center=True reflect padding, periodic Hann, power 2,
// amin 1e-10, top_db 80, population std).
//
// mel_ranges_kernel : first/last non-zero bin of every triangular filter (they are sparse:
//                     ~2*257 non-zeros in an [n_mels x 257] basis).
// mel_frame_kernel  : a workgroup walks FPB consecutive frames of one utterance: coalesced gather of
//                     the 512 reflect-padded samples, Hann window, 512-point radix-2 Stockham FFT in
//                     LDS with twiddles tabulated once per workgroup, |.|^2, sparse mel projection,
//                     running max (integer atomicMax: order independent) for the dB reference.
// mel_stats_kernel  : dB + -80 dB floor, per-utterance sum / sum of squares in double, fixed-shape
//                     partials (deterministic).
// mel_finish_kernel : mean / population std from the partials, normalised write in the reference's
//                     [time, n_mels] layout.
// Roofline: 640 KB in + 320 KB out per 10 s utterance (HBM floor ~0.1 us); in practice bound by the
// FFT's LDS passes and launch latency, reported separately from the training step by bench.py.
#include <algorithm>

#include "mel_fft.h"
#include "st_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NFFT = 512;
constexpr int NBINS = NFFT / 2 + 1;
constexpr int FPB = 8;          // frames per workgroup
constexpr int STAT_CHUNKS = 64; // partial sums per utterance

__global__ void mel_ranges_kernel(const float* __restrict__ basis, int n_mels, int* __restrict__ ranges) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n_mels) return;
  const float* row = basis + (long)m * NBINS;
  int lo = NBINS, hi = 0;
  for (int k = 0; k < NBINS; ++k)
    if (row[k] != 0.f) { lo = min(lo, k); hi = k + 1; }
  ranges[2 * m] = min(lo, hi);
  ranges[2 * m + 1] = hi;
}

__global__ __launch_bounds__(256) void mel_frame_kernel(const float* __restrict__ audio,
                                                        const long* __restrict__ sample_off,
                                                        const float* __restrict__ basis,
                                                        const int* __restrict__ ranges, int n_mels, int hop,
                                                        const long* __restrict__ frame_off,
                                                        float* __restrict__ melpow, unsigned* __restrict__ umax) {
  __shared__ float re[2][NFFT];
  __shared__ float im[2][NFFT];
  __shared__ float twr[NFFT / 2], twi[NFFT / 2];   // W512^k = exp(-2*pi*i*k/512)
  __shared__ float pw[NBINS + 3];
  __shared__ float wmax[4];
  const int u = blockIdx.y;
  const long s0 = sample_off[u];
  const int n = (int)(sample_off[u + 1] - s0);
  const int frames = 1 + n / hop;
  const int t_begin = blockIdx.x * FPB;
  if (t_begin >= frames) return;
  const int tid = threadIdx.x;
  const float* y = audio + s0;
  {
    float sn, cs;
    sincospif(-2.0f * (float)tid / (float)NFFT, &sn, &cs);
    twr[tid] = cs;
    twi[tid] = sn;
  }
  const float w0 = 0.5f - 0.5f * cospif(2.0f * (float)tid / (float)NFFT);            // periodic Hann
  const float w1 = 0.5f - 0.5f * cospif(2.0f * (float)(tid + 256) / (float)NFFT);
  int m_lo = 0, m_hi = 0;
  if (tid < n_mels) { m_lo = ranges[2 * tid]; m_hi = ranges[2 * tid + 1]; }
  const float* brow = basis + (long)min(tid, n_mels - 1) * NBINS;
  float vmax = 0.f;
  __syncthreads();

  const int t_end = min(frames, t_begin + FPB);
  for (int t = t_begin; t < t_end; ++t) {
    // windowed frame; centre=True: padded index p = t*hop + k  <->  sample p - NFFT/2, reflected
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int k = tid + 256 * r;
      int j = t * hop + k - NFFT / 2;
      if (j < 0) j = -j;
      if (j >= n) j = 2 * (n - 1) - j;
      j = min(max(j, 0), n - 1);
      re[0][k] = y[j] * (r ? w1 : w0);
      im[0][k] = 0.f;
    }
    __syncthreads();
    // Stockham autosort radix-2 (decimation in frequency): 9 stages, one butterfly per thread per
    // stage, natural-order result.  stage st: stride s = 2^st, twiddle W_n^p = W512^(p << st).
    int cur = 0;
#pragma unroll
    for (int st = 0; st < 9; ++st) {
      const int s = 1 << st;
      const int p = tid >> st, q = tid & (s - 1);
      const float cs = twr[p << st], sn = twi[p << st];
      const float ar = re[cur][tid], ai = im[cur][tid];
      const float br = re[cur][tid + NFFT / 2], bi = im[cur][tid + NFFT / 2];
      const float dr = ar - br, di = ai - bi;
      const int o0 = q + s * 2 * p, o1 = o0 + s;
      re[cur ^ 1][o0] = ar + br; im[cur ^ 1][o0] = ai + bi;
      re[cur ^ 1][o1] = dr * cs - di * sn; im[cur ^ 1][o1] = dr * sn + di * cs;
      __syncthreads();
      cur ^= 1;
    }
    for (int k = tid; k < NBINS; k += 256) pw[k] = re[cur][k] * re[cur][k] + im[cur][k] * im[cur][k];
    __syncthreads();
    if (tid < n_mels) {
      float acc = 0.f;
      for (int k = m_lo; k < m_hi; ++k) acc = fmaf(brow[k], pw[k], acc);
      melpow[(frame_off[u] + t) * (long)n_mels + tid] = acc;
      vmax = fmaxf(vmax, acc);
    }
    // (n_mels > 256 is rejected by the host wrapper)
  }
  vmax = st::wave_max(vmax);
  if ((tid & 63) == 0) wmax[tid >> 6] = vmax;
  __syncthreads();
  if (tid == 0) {
    float v = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    atomicMax(&umax[u], __float_as_uint(v));     // non-negative floats order like their bit patterns
  }
}

// ---- wave-per-frame-pair kernel (the default) -----------------------------------------------------------------
// A wave transforms TWO consecutive frames with one 512-point complex FFT (frame 2p in the real part, 2p+1 in the
// imaginary part; csrc/mel_fft.h): three radix-8 Stockham stages, 8 points per lane in registers, two exchanges
// through a wave-private LDS buffer, no workgroup barrier anywhere.  The power spectra of both frames come from
// Z[k] and Z[512-k] (one cross-lane fetch per value), the sparse triangular mel filters are applied from a plan
// (st_melspec_plan_f32: every filter cut into pieces of <= 16 bins, 64 pieces per round, both frames side by side)
// with the partial sums combined by LDS adds, and the mel rows of both frames leave as one coalesced run.
constexpr int PLAN_PIECE = 16;                  // bins per piece
constexpr int PLAN_MAX_ROUNDS = 12;             // 64 pieces per round; n_mels <= 256 needs <= 2 * (256 + 34) pieces
constexpr int PAIRS_PER_WAVE = 4;

struct PlanItem { int mel, sel, k_lo, n; };     // mel < 0: idle slot
struct PlanHeader { int rounds, n_mels, per_frame, pad; };   // per_frame = pieces of one frame
// plan = PlanHeader | PlanItem[rounds * 64] | float weights[rounds * 64][PLAN_PIECE] | int first[n_mels + 1]
// slot of piece j of filter m for frame `sel`: sel * per_frame + first[m] + j
constexpr size_t PLAN_BYTES = sizeof(PlanHeader) + (size_t)PLAN_MAX_ROUNDS * 64 * (sizeof(PlanItem) + PLAN_PIECE * sizeof(float)) +
                              257 * sizeof(int);

__global__ __launch_bounds__(256) void mel_plan_kernel(const float* __restrict__ basis, int n_mels, char* __restrict__ plan) {
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
    first_s[n_mels] = total;                                // pieces of ONE frame
    PlanHeader* h = reinterpret_cast<PlanHeader*>(plan);
    h->rounds = (2 * total + 63) / 64;
    h->n_mels = n_mels;
    h->per_frame = total;
    h->pad = 0;
  }
  __syncthreads();
  const int per_frame = first_s[n_mels];
  const int rounds = (2 * per_frame + 63) / 64;
  PlanItem* items = reinterpret_cast<PlanItem*>(plan + sizeof(PlanHeader));
  float* weights = reinterpret_cast<float*>(plan + sizeof(PlanHeader) + (size_t)rounds * 64 * sizeof(PlanItem));
  int* first = reinterpret_cast<int*>(weights + (size_t)rounds * 64 * PLAN_PIECE);
  for (int m = tid; m <= n_mels; m += 256) first[m] = first_s[m];
  for (int slot = tid; slot < rounds * 64; slot += 256) {
    PlanItem it{-1, 0, 0, 0};
    const int sel = slot >= per_frame ? 1 : 0, piece = slot - sel * per_frame;
    if (piece < per_frame) {
      int m = 0;
      while (first_s[m + 1] <= piece) ++m;                  // n_mels <= 256: a short scan, once per plan
      const int k_lo = lo_s[m] + (piece - first_s[m]) * PLAN_PIECE;
      it = PlanItem{m, sel, k_lo, max(0, min(PLAN_PIECE, hi_s[m] - k_lo))};
    }
    items[slot] = it;
    for (int i = 0; i < PLAN_PIECE; ++i)
      weights[(size_t)slot * PLAN_PIECE + i] = (it.mel >= 0 && i < it.n) ? basis[(long)it.mel * NBINS + it.k_lo + i] : 0.f;
  }
}

__device__ __forceinline__ void wave_lds_sync() {
  // LDS traffic of one wave is executed in order; this only keeps the compiler from moving accesses across
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256) void mel_pair_kernel(const float* __restrict__ audio,
                                                       const long* __restrict__ sample_off,
                                                       const char* __restrict__ plan, int n_mels, int hop,
                                                       const long* __restrict__ frame_off,
                                                       float* __restrict__ melpow, unsigned* __restrict__ umax) {
  using melfft::cf;
  __shared__ cf zbuf[4][melfft::LDS_COMPLEX];
  __shared__ float pw[4][2][NBINS + 7];
  const int u = blockIdx.y;
  const long s0 = sample_off[u];
  const int n = (int)(sample_off[u + 1] - s0);
  const int frames = 1 + n / hop;
  const int pairs = (frames + 1) / 2;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int p_begin = (blockIdx.x * 4 + wave) * PAIRS_PER_WAVE;
  if (p_begin >= pairs) return;
  const float* y = audio + s0;
  const PlanHeader hdr = *reinterpret_cast<const PlanHeader*>(plan);
  const PlanItem* items = reinterpret_cast<const PlanItem*>(plan + sizeof(PlanHeader));
  const float* weights = reinterpret_cast<const float*>(plan + sizeof(PlanHeader) + (size_t)hdr.rounds * 64 * sizeof(PlanItem));
  const int* first = reinterpret_cast<const int*>(weights + (size_t)hdr.rounds * 64 * PLAN_PIECE);
  if (lane < 14) pw[wave][lane / 7][NBINS + lane % 7] = 0.f;      // the read-ahead pad of both rows (weights there are 0)
  // per-lane constants: periodic Hann at lane + 64 r, twiddles of stage 1 (W64^((lane & 7) r)) and 2 (W512^(lane r))
  float win[8];
  cf tw1[8], tw2[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    win[r] = 0.5f - 0.5f * cospif(2.0f * (float)(lane + 64 * r) / (float)NFFT);
    sincospif(-2.0f * (float)((lane & 7) * r) / 64.0f, &tw1[r].y, &tw1[r].x);
    sincospif(-2.0f * (float)(lane * r) / 512.0f, &tw2[r].y, &tw2[r].x);
  }
  cf* zb = zbuf[wave];
  float* part = reinterpret_cast<float*>(zb);              // piece sums: the FFT buffer is idle during the projection
  float vmax = 0.f;
  const int p_end = min(pairs, p_begin + PAIRS_PER_WAVE);
  for (int p = p_begin; p < p_end; ++p) {
    const int ta = 2 * p, tb = ta + 1;
    cf v[8];
    // windowed frames; centre=True: padded index t*hop + k  <->  sample t*hop + k - NFFT/2, reflected at the ends
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int k = lane + 64 * r;
      int ja = ta * hop + k - NFFT / 2, jb = ja + hop;
      if (ja < 0) ja = -ja;
      if (ja >= n) ja = 2 * (n - 1) - ja;
      ja = min(max(ja, 0), n - 1);
      if (jb < 0) jb = -jb;
      if (jb >= n) jb = 2 * (n - 1) - jb;
      jb = min(max(jb, 0), n - 1);
      v[r].x = y[ja] * win[r];
      v[r].y = tb < frames ? y[jb] * win[r] : 0.f;
    }
    melfft::dft8(v);
#pragma unroll
    for (int r = 0; r < 8; ++r) zb[melfft::pad(melfft::out_index(0, lane, r))] = v[r];
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const cf z = zb[melfft::pad(lane + 64 * r)];
      v[r] = r ? melfft::zmul(z, tw1[r]) : z;
    }
    wave_lds_sync();
    melfft::dft8(v);
#pragma unroll
    for (int r = 0; r < 8; ++r) zb[melfft::pad(melfft::out_index(1, lane, r))] = v[r];
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const cf z = zb[melfft::pad(lane + 64 * r)];
      v[r] = r ? melfft::zmul(z, tw2[r]) : z;
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
    for (int round = 0; round < hdr.rounds; ++round) {
      const PlanItem it = items[round * 64 + lane];
      if (it.mel >= 0) {
        const f32x4* w4 = reinterpret_cast<const f32x4*>(weights + (size_t)(round * 64 + lane) * PLAN_PIECE);
        const float* src = &pw[wave][it.sel][it.k_lo];
        float sum = 0.f;
        for (int i = 0; i < it.n; i += 4) {
          const f32x4 w = w4[i >> 2];                      // zero beyond it.n; pw rows carry 7 readable pad floats
          sum = fmaf(w[0], src[i], sum);
          sum = fmaf(w[1], src[i + 1], sum);
          sum = fmaf(w[2], src[i + 2], sum);
          sum = fmaf(w[3], src[i + 3], sum);
        }
        part[round * 64 + lane] = sum;
      }
    }
    wave_lds_sync();
    // both mel rows are adjacent in memory (frames ta and ta + 1 of the same utterance); a filter's pieces are
    // summed in plan order (deterministic)
    float* dst = melpow + (frame_off[u] + ta) * (long)n_mels;
    const int valid = (tb < frames ? 2 : 1) * n_mels;
    for (int i = lane; i < valid; i += 64) {
      const int sel = i >= n_mels ? 1 : 0, m = i - sel * n_mels;
      float a = 0.f;
      for (int j = first[m]; j < first[m + 1]; ++j) a += part[sel * hdr.per_frame + j];
      dst[i] = a;
      vmax = fmaxf(vmax, a);
    }
    wave_lds_sync();
  }
  vmax = st::wave_max(vmax);
  if (lane == 0) atomicMax(&umax[u], __float_as_uint(vmax));     // non-negative floats order like their bit patterns
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

// power_to_db(ref = max, amin 1e-10, top_db 80): the maximum of the dB matrix is 0 by construction
__device__ __forceinline__ float to_db(float s, float ref_db) {
  return fmaxf(10.f * log10f(fmaxf(1e-10f, s)) - ref_db, -80.f);
}

__global__ __launch_bounds__(256) void mel_stats_kernel(const float* __restrict__ melpow,
                                                        const long* __restrict__ sample_off,
                                                        const long* __restrict__ frame_off, int n_mels, int hop,
                                                        const unsigned* __restrict__ umax,
                                                        double* __restrict__ partial) {
  __shared__ double red[4];
  const int u = blockIdx.y;
  const int n = (int)(sample_off[u + 1] - sample_off[u]);
  const long count = (long)(1 + n / hop) * n_mels;
  const float* src = melpow + frame_off[u] * (long)n_mels;
  const float ref_db = 10.f * log10f(fmaxf(1e-10f, __uint_as_float(umax[u])));
  const long per = (count + STAT_CHUNKS - 1) / STAT_CHUNKS;
  const long lo = blockIdx.x * per, hi = min(count, lo + per);
  double s = 0.0, ss = 0.0;
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
    const double d = (double)to_db(src[i], ref_db);
    s += d;
    ss += d * d;
  }
  s = block_sum_d(s, red);
  ss = block_sum_d(ss, red);
  if (threadIdx.x == 0) {
    partial[((long)u * STAT_CHUNKS + blockIdx.x) * 2] = s;
    partial[((long)u * STAT_CHUNKS + blockIdx.x) * 2 + 1] = ss;
  }
}

__global__ __launch_bounds__(256) void mel_finish_kernel(const float* __restrict__ melpow,
                                                         const long* __restrict__ sample_off,
                                                         const long* __restrict__ frame_off, int n_mels, int hop,
                                                         const unsigned* __restrict__ umax,
                                                         const double* __restrict__ partial,
                                                         float* __restrict__ out) {
  const int u = blockIdx.y;
  const int n = (int)(sample_off[u + 1] - sample_off[u]);
  const long count = (long)(1 + n / hop) * n_mels;
  double s = 0.0, ss = 0.0;
  for (int c = 0; c < STAT_CHUNKS; ++c) {          // same order in every block: deterministic
    s += partial[((long)u * STAT_CHUNKS + c) * 2];
    ss += partial[((long)u * STAT_CHUNKS + c) * 2 + 1];
  }
  const double mean = s / (double)count;
  const double var = fmax(ss / (double)count - mean * mean, 0.0);
  const float inv_std = (float)(1.0 / sqrt(var));
  const float meanf = (float)mean;
  const float* src = melpow + frame_off[u] * (long)n_mels;
  float* dst = out + frame_off[u] * (long)n_mels;
  const float ref_db = 10.f * log10f(fmaxf(1e-10f, __uint_as_float(umax[u])));
  const long per = (count + STAT_CHUNKS - 1) / STAT_CHUNKS;
  const long lo = blockIdx.x * per, hi = min(count, lo + per);
  for (long i = lo + threadIdx.x; i < hi; i += 256) dst[i] = (to_db(src[i], ref_db) - meanf) * inv_std;
}

// ---- MFCC + delta + delta-delta (preprocessing.py:61-84) --------------------------------------------
// librosa.feature.mfcc = DCT-II (orthonormal) of power_to_db(mel power, ref = 1.0, top_db = 80);
// librosa.feature.delta (0.5.x: FIR [4..-4]/60 run causally from rest over the edge-padded signal, once or
// twice); each of the three [n_mfcc, T] blocks is z-normalised on its own.  Reuses mel_ranges / mel_frame.
constexpr int MAX_MFCC = 32;

// one wave per frame: dB of the mel bins in registers, n_mfcc cosine projections, wave reduction
__global__ __launch_bounds__(256) void mfcc_dct_kernel(const float* __restrict__ melpow,
                                                       const long* __restrict__ sample_off,
                                                       const long* __restrict__ frame_off, int n_mels, int n_mfcc,
                                                       int hop, const unsigned* __restrict__ umax,
                                                       float* __restrict__ coef) {
  const int u = blockIdx.y;
  const int frames = 1 + (int)(sample_off[u + 1] - sample_off[u]) / hop;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= frames) return;
  const int lane = threadIdx.x & 63;
  const float floor_db = 10.f * log10f(fmaxf(1e-10f, __uint_as_float(umax[u]))) - 80.f;
  const float* src = melpow + (frame_off[u] + t) * (long)n_mels;
  float db[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = lane + 64 * j;
    db[j] = m < n_mels ? fmaxf(10.f * log10f(fmaxf(1e-10f, src[m])), floor_db) : 0.f;
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

}  // namespace

extern "C" {

size_t st_melspec_plan_bytes(void) { return st::round_up(PLAN_BYTES, 256); }

int st_melspec_plan_f32(const float* mel_basis, int n_mels, int n_fft, void* plan, size_t plan_bytes, void* stream) {
  ST_REQUIRE(mel_basis && plan && plan_bytes >= st_melspec_plan_bytes(), "melspec plan: bad args");
  ST_REQUIRE(n_fft == NFFT && n_mels > 0 && n_mels <= 256, "melspec plan: n_fft must be 512 and n_mels <= 256");
  ST_REQUIRE(((uintptr_t)plan & 15) == 0, "melspec plan: buffer must be 16-byte aligned");
  hipLaunchKernelGGL(mel_plan_kernel, dim3(1), dim3(256), 0, st::as_stream(stream), mel_basis, n_mels,
                     reinterpret_cast<char*>(plan));
  return st::check_launch("mel_plan");
}

size_t st_melspec_ws(int n_utts, int64_t total_frames, int n_mels) {
  if (n_utts <= 0 || total_frames <= 0 || n_mels <= 0) return 0;
  return pow_bytes(total_frames, n_mels) + st::round_up((size_t)n_utts * 4, 256) +
         st::round_up((size_t)n_mels * 2 * sizeof(int), 256) + (size_t)n_utts * STAT_CHUNKS * 2 * sizeof(double) +
         st_melspec_plan_bytes() + 256;
}

// mel power of every frame + per-utterance maximum: the wave-per-frame-pair kernel, or (tuning "mel_variant" = 1,
// for A/B measurements) the first-generation workgroup-per-frame kernel
static int launch_frames(const float* audio, const long* soff, int n_utts, int64_t max_samples, const float* mel_basis,
                         const char* plan, int* ranges, int n_mels, int hop, const long* foff, float* melpow,
                         unsigned* umax, hipStream_t s) {
  if (hipMemsetAsync(umax, 0, (size_t)n_utts * 4, s) != hipSuccess) {
    st::set_error("melspec: memset failed");
    return ST_ELAUNCH;
  }
  const unsigned max_frames = (unsigned)(1 + max_samples / hop);
  if (st::tuning(st::TUNE_MEL_VARIANT) == 1 && mel_basis && ranges) {
    hipLaunchKernelGGL(mel_ranges_kernel, dim3(st::ceil_div(n_mels, 64)), dim3(64), 0, s, mel_basis, n_mels, ranges);
    hipLaunchKernelGGL(mel_frame_kernel, dim3(st::ceil_div((int)max_frames, FPB), n_utts), dim3(256), 0, s, audio, soff,
                       mel_basis, ranges, n_mels, hop, foff, melpow, umax);
  } else {
    const int pairs = (int)(max_frames + 1) / 2;
    hipLaunchKernelGGL(mel_pair_kernel, dim3(st::ceil_div(pairs, 4 * PAIRS_PER_WAVE), n_utts), dim3(256), 0, s, audio, soff,
                       plan, n_mels, hop, foff, melpow, umax);
  }
  return st::check_launch("mel_frames");
}

int st_melspec_planned_f32(const float* audio, const int64_t* sample_offsets, int n_utts, int64_t max_samples,
                           const void* plan, int n_mels, int n_fft, int hop, const int64_t* frame_offsets,
                           int64_t total_frames, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  ST_REQUIRE(audio && sample_offsets && plan && frame_offsets && out && workspace, "melspec: null argument");
  ST_REQUIRE(n_fft == NFFT, "melspec: only n_fft = 512 (the reference default, preprocessing.py:36) is built");
  ST_REQUIRE(n_utts > 0 && n_mels > 0 && n_mels <= 256 && hop > 0 && max_samples > NFFT / 2 && total_frames > 0,
             "melspec: bad shape");
  ST_REQUIRE(workspace_bytes >= st_melspec_ws(n_utts, total_frames, n_mels) - st_melspec_plan_bytes() - 256,
             "melspec: workspace too small");
  hipStream_t s = st::as_stream(stream);
  char* w = reinterpret_cast<char*>(workspace);
  float* melpow = reinterpret_cast<float*>(w);
  w += pow_bytes(total_frames, n_mels);
  unsigned* umax = reinterpret_cast<unsigned*>(w);
  w += st::round_up((size_t)n_utts * 4, 256);
  w += st::round_up((size_t)n_mels * 2 * sizeof(int), 256);
  double* partial = reinterpret_cast<double*>(w);
  const long* soff = reinterpret_cast<const long*>(sample_offsets);
  const long* foff = reinterpret_cast<const long*>(frame_offsets);
  if (int e = launch_frames(audio, soff, n_utts, max_samples, nullptr, reinterpret_cast<const char*>(plan), nullptr, n_mels,
                            hop, foff, melpow, umax, s))
    return e;
  hipLaunchKernelGGL(mel_stats_kernel, dim3(STAT_CHUNKS, n_utts), dim3(256), 0, s, melpow, soff, foff, n_mels, hop,
                     umax, partial);
  hipLaunchKernelGGL(mel_finish_kernel, dim3(STAT_CHUNKS, n_utts), dim3(256), 0, s, melpow, soff, foff, n_mels, hop,
                     umax, partial, out);
  return st::check_launch("melspec");
}

int st_melspec_f32(const float* audio, const int64_t* sample_offsets, int n_utts, int64_t max_samples,
                   const float* mel_basis, int n_mels, int n_fft, int hop, const int64_t* frame_offsets,
                   int64_t total_frames, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  ST_REQUIRE(audio && sample_offsets && mel_basis && frame_offsets && out && workspace, "melspec: null argument");
  ST_REQUIRE(n_fft == NFFT, "melspec: only n_fft = 512 (the reference default, preprocessing.py:36) is built");
  ST_REQUIRE(n_utts > 0 && n_mels > 0 && n_mels <= 256 && hop > 0 && max_samples > NFFT / 2 && total_frames > 0,
             "melspec: bad shape");
  ST_REQUIRE(workspace_bytes >= st_melspec_ws(n_utts, total_frames, n_mels), "melspec: workspace too small");
  hipStream_t s = st::as_stream(stream);
  char* w = reinterpret_cast<char*>(workspace);
  float* melpow = reinterpret_cast<float*>(w);
  w += pow_bytes(total_frames, n_mels);
  unsigned* umax = reinterpret_cast<unsigned*>(w);
  w += st::round_up((size_t)n_utts * 4, 256);
  int* ranges = reinterpret_cast<int*>(w);
  w += st::round_up((size_t)n_mels * 2 * sizeof(int), 256);
  double* partial = reinterpret_cast<double*>(w);
  w += (size_t)n_utts * STAT_CHUNKS * 2 * sizeof(double);
  char* plan = reinterpret_cast<char*>(st::round_up((size_t)(uintptr_t)w, 256));
  if (st::tuning(st::TUNE_MEL_VARIANT) != 1)
    if (int e = st_melspec_plan_f32(mel_basis, n_mels, n_fft, plan, st_melspec_plan_bytes(), stream)) return e;
  const long* soff = reinterpret_cast<const long*>(sample_offsets);
  const long* foff = reinterpret_cast<const long*>(frame_offsets);
  if (int e = launch_frames(audio, soff, n_utts, max_samples, mel_basis, plan, ranges, n_mels, hop, foff, melpow, umax, s))
    return e;
  hipLaunchKernelGGL(mel_stats_kernel, dim3(STAT_CHUNKS, n_utts), dim3(256), 0, s, melpow, soff, foff, n_mels, hop,
                     umax, partial);
  hipLaunchKernelGGL(mel_finish_kernel, dim3(STAT_CHUNKS, n_utts), dim3(256), 0, s, melpow, soff, foff, n_mels, hop,
                     umax, partial, out);
  return st::check_launch("melspec");
}

size_t st_mfcc_ws(int n_utts, int64_t total_frames, int n_mels, int n_mfcc) {
  if (n_utts <= 0 || total_frames <= 0 || n_mels <= 0 || n_mfcc <= 0) return 0;
  return pow_bytes(total_frames, n_mels) + st::round_up((size_t)n_utts * 4, 256) +
         st::round_up((size_t)n_mels * 2 * sizeof(int), 256) + 3 * pow_bytes(total_frames, n_mfcc) +
         (size_t)n_utts * 3 * STAT_CHUNKS * 2 * sizeof(double) + st_melspec_plan_bytes() + 256;
}

int st_mfcc_f32(const float* audio, const int64_t* sample_offsets, int n_utts, int64_t max_samples,
                const float* mel_basis, int n_mels, int n_mfcc, int n_fft, int hop, const int64_t* frame_offsets,
                int64_t total_frames, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  ST_REQUIRE(audio && sample_offsets && mel_basis && frame_offsets && out && workspace, "mfcc: null argument");
  ST_REQUIRE(n_fft == NFFT, "mfcc: only n_fft = 512 (the reference default, preprocessing.py:61) is built");
  ST_REQUIRE(n_utts > 0 && n_mels > 0 && n_mels <= 256 && n_mfcc > 0 && n_mfcc <= MAX_MFCC && n_mfcc <= n_mels &&
                 hop > 0 && max_samples > NFFT / 2 && total_frames > 0,
             "mfcc: bad shape");
  ST_REQUIRE(workspace_bytes >= st_mfcc_ws(n_utts, total_frames, n_mels, n_mfcc), "mfcc: workspace too small");
  hipStream_t s = st::as_stream(stream);
  char* w = reinterpret_cast<char*>(workspace);
  float* melpow = reinterpret_cast<float*>(w);
  w += pow_bytes(total_frames, n_mels);
  unsigned* umax = reinterpret_cast<unsigned*>(w);
  w += st::round_up((size_t)n_utts * 4, 256);
  int* ranges = reinterpret_cast<int*>(w);
  w += st::round_up((size_t)n_mels * 2 * sizeof(int), 256);
  float* coef = reinterpret_cast<float*>(w);
  float* d1 = reinterpret_cast<float*>(w + pow_bytes(total_frames, n_mfcc));
  float* d2 = reinterpret_cast<float*>(w + 2 * pow_bytes(total_frames, n_mfcc));
  w += 3 * pow_bytes(total_frames, n_mfcc);
  double* partial = reinterpret_cast<double*>(w);
  w += (size_t)n_utts * 3 * STAT_CHUNKS * 2 * sizeof(double);
  char* plan = reinterpret_cast<char*>(st::round_up((size_t)(uintptr_t)w, 256));
  if (st::tuning(st::TUNE_MEL_VARIANT) != 1)
    if (int e = st_melspec_plan_f32(mel_basis, n_mels, n_fft, plan, st_melspec_plan_bytes(), stream)) return e;
  const long* soff = reinterpret_cast<const long*>(sample_offsets);
  const long* foff = reinterpret_cast<const long*>(frame_offsets);
  const unsigned max_frames = (unsigned)(1 + max_samples / hop);
  if (int e = launch_frames(audio, soff, n_utts, max_samples, mel_basis, plan, ranges, n_mels, hop, foff, melpow, umax, s))
    return e;
  hipLaunchKernelGGL(mfcc_dct_kernel, dim3(st::ceil_div((int)max_frames, 4), n_utts), dim3(256), 0, s, melpow, soff, foff,
                     n_mels, n_mfcc, hop, umax, coef);
  hipLaunchKernelGGL(mfcc_delta_kernel, dim3(STAT_CHUNKS, n_utts), dim3(256), 0, s, coef, soff, foff, n_mfcc, hop, d1, d2,
                     partial);
  hipLaunchKernelGGL(mfcc_finish_kernel, dim3(STAT_CHUNKS, n_utts), dim3(256), 0, s, coef, d1, d2, soff, foff, n_mfcc, hop,
                     partial, out);
  return st::check_launch("mfcc");
}

}  // extern "C"
