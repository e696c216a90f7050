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

#include "st_common.h"

namespace {

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

size_t st_melspec_ws(int n_utts, int64_t total_frames, int n_mels) {
  if (n_utts <= 0 || total_frames <= 0 || n_mels <= 0) return 0;
  return pow_bytes(total_frames, n_mels) + st::round_up((size_t)n_utts * 4, 256) +
         st::round_up((size_t)n_mels * 2 * sizeof(int), 256) + (size_t)n_utts * STAT_CHUNKS * 2 * sizeof(double);
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
  if (hipMemsetAsync(umax, 0, (size_t)n_utts * 4, s) != hipSuccess) {
    st::set_error("melspec: memset failed");
    return ST_ELAUNCH;
  }
  const long* soff = reinterpret_cast<const long*>(sample_offsets);
  const long* foff = reinterpret_cast<const long*>(frame_offsets);
  const unsigned max_frames = (unsigned)(1 + max_samples / hop);
  hipLaunchKernelGGL(mel_ranges_kernel, dim3(st::ceil_div(n_mels, 64)), dim3(64), 0, s, mel_basis, n_mels, ranges);
  hipLaunchKernelGGL(mel_frame_kernel, dim3(st::ceil_div((int)max_frames, FPB), n_utts), dim3(256), 0, s, audio, soff,
                     mel_basis, ranges, n_mels, hop, foff, melpow, umax);
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
         (size_t)n_utts * 3 * STAT_CHUNKS * 2 * sizeof(double);
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
  if (hipMemsetAsync(umax, 0, (size_t)n_utts * 4, s) != hipSuccess) {
    st::set_error("mfcc: memset failed");
    return ST_ELAUNCH;
  }
  const long* soff = reinterpret_cast<const long*>(sample_offsets);
  const long* foff = reinterpret_cast<const long*>(frame_offsets);
  const unsigned max_frames = (unsigned)(1 + max_samples / hop);
  hipLaunchKernelGGL(mel_ranges_kernel, dim3(st::ceil_div(n_mels, 64)), dim3(64), 0, s, mel_basis, n_mels, ranges);
  hipLaunchKernelGGL(mel_frame_kernel, dim3(st::ceil_div((int)max_frames, FPB), n_utts), dim3(256), 0, s, audio, soff,
                     mel_basis, ranges, n_mels, hop, foff, melpow, umax);
  hipLaunchKernelGGL(mfcc_dct_kernel, dim3(st::ceil_div((int)max_frames, 4), n_utts), dim3(256), 0, s, melpow, soff, foff,
                     n_mels, n_mfcc, hop, umax, coef);
  hipLaunchKernelGGL(mfcc_delta_kernel, dim3(STAT_CHUNKS, n_utts), dim3(256), 0, s, coef, soff, foff, n_mfcc, hop, d1, d2,
                     partial);
  hipLaunchKernelGGL(mfcc_finish_kernel, dim3(STAT_CHUNKS, n_utts), dim3(256), 0, s, coef, d1, d2, soff, foff, n_mfcc, hop,
                     partial, out);
  return st::check_launch("mfcc");
}

}  // extern "C"
