// Mel-filterbank feature extractor on gfx950: calc_power_spectrogram (preprocessing.py:36-58).
//
//   librosa.feature.melspectrogram(y, sr, n_mels, n_fft=512, hop_length=160)   (:50)
//   -> librosa.power_to_db(S, ref=np.max)                                      (:53)
//   -> normalize: (x - mean) / std over the whole matrix                       (:29-33, :56)
//   -> transpose to [time, n_mels]                                             (:58)
// Semantics: SURVEY Appendix A7-A9 (center=True reflect padding, periodic Hann, power 2,
// amin 1e-10, top_db 80, population std).
//
// Kernel 1 (one workgroup per frame): gather 512 reflect-padded samples (coalesced), window,
// 512-point radix-2 Stockham FFT in LDS, |.|^2 of the 257 bins, mel projection against the
// L2-resident basis, per-utterance running max by integer atomicMax (order independent).
// Kernel 2 (one workgroup per utterance): dB conversion with the utterance max as reference,
// -80 dB floor, mean and population std by fixed-shape tree sums in double (deterministic),
// normalised write-out in the reference's [time, n_mels] layout.  HBM-bound: 640 KB in,
// 320 KB out per 10 s utterance.
#include <algorithm>

#include "st_common.h"

namespace {

constexpr int NFFT = 512;
constexpr int NBINS = NFFT / 2 + 1;

__global__ __launch_bounds__(256) void mel_frame_kernel(const float* __restrict__ audio,
                                                        const long* __restrict__ sample_off,
                                                        const float* __restrict__ basis, int n_mels, int hop,
                                                        const long* __restrict__ frame_off,
                                                        float* __restrict__ melpow, unsigned* __restrict__ umax) {
  __shared__ float re[2][NFFT];
  __shared__ float im[2][NFFT];
  __shared__ float pw[NBINS + 3];
  __shared__ float wmax[4];
  const int u = blockIdx.y;
  const long s0 = sample_off[u];
  const int n = (int)(sample_off[u + 1] - s0);
  const int frames = 1 + n / hop;
  const int t = blockIdx.x;
  if (t >= frames) return;
  const int tid = threadIdx.x;
  const float* y = audio + s0;

  // windowed frame; centre=True: padded index p = t*hop + k  <->  sample p - NFFT/2, reflected
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int k = tid + 256 * r;
    int j = t * hop + k - NFFT / 2;
    if (j < 0) j = -j;
    if (j >= n) j = 2 * (n - 1) - j;
    j = min(max(j, 0), n - 1);
    const float w = 0.5f - 0.5f * cospif(2.0f * (float)k / (float)NFFT);
    re[0][k] = y[j] * w;
    im[0][k] = 0.f;
  }
  __syncthreads();

  // Stockham autosort radix-2 (decimation in frequency): 9 stages, one butterfly per thread per
  // stage, result in natural order.  stage st: stride s = 2^st, current length n = NFFT >> st.
  int cur = 0;
#pragma unroll
  for (int st = 0; st < 9; ++st) {
    const int s = 1 << st;
    const int p = tid >> st, q = tid & (s - 1);
    float sn, cs;
    sincospif(-2.0f * (float)p / (float)(NFFT >> st), &sn, &cs);      // w = exp(-2*pi*i*p/n)
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

  float vmax = 0.f;
  for (int m = tid; m < n_mels; m += 256) {
    const float* row = basis + (long)m * NBINS;
    float acc = 0.f;
    for (int k = 0; k < NBINS; ++k) acc = fmaf(row[k], pw[k], acc);
    melpow[(frame_off[u] + t) * (long)n_mels + m] = acc;
    vmax = fmaxf(vmax, acc);
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

__global__ __launch_bounds__(1024) void mel_normalize_kernel(const float* __restrict__ melpow,
                                                             const long* __restrict__ sample_off,
                                                             const long* __restrict__ frame_off, int n_mels,
                                                             int hop, const unsigned* __restrict__ umax,
                                                             float* __restrict__ out) {
  __shared__ double red[16];
  const int u = blockIdx.x;
  const int n = (int)(sample_off[u + 1] - sample_off[u]);
  const long count = (long)(1 + n / hop) * n_mels;
  const float* src = melpow + frame_off[u] * (long)n_mels;
  float* dst = out + frame_off[u] * (long)n_mels;
  const float amin = 1e-10f;
  const float ref_db = 10.f * log10f(fmaxf(amin, __uint_as_float(umax[u])));
  // max over the matrix of (10 log10(max(amin,S)) - ref_db) is attained at S = max(S):
  const float top = 10.f * log10f(fmaxf(amin, __uint_as_float(umax[u]))) - ref_db;   // == 0
  const float floor_db = top - 80.f;
  auto db = [&](float s) { return fmaxf(10.f * log10f(fmaxf(amin, s)) - ref_db, floor_db); };
  double sum = 0.0;
  for (long i = threadIdx.x; i < count; i += blockDim.x) sum += (double)db(src[i]);
  const double mean = block_sum_d(sum, red) / (double)count;
  double ss = 0.0;
  for (long i = threadIdx.x; i < count; i += blockDim.x) { double d = (double)db(src[i]) - mean; ss += d * d; }
  const double var = block_sum_d(ss, red) / (double)count;
  const float inv_std = (float)(1.0 / sqrt(var));
  const float meanf = (float)mean;
  for (long i = threadIdx.x; i < count; i += blockDim.x) dst[i] = (db(src[i]) - meanf) * inv_std;
}

}  // namespace

extern "C" {

size_t st_melspec_ws(int n_utts, int64_t total_frames, int n_mels) {
  if (n_utts <= 0 || total_frames <= 0 || n_mels <= 0) return 0;
  return st::round_up((size_t)total_frames * n_mels * sizeof(float), 256) + st::round_up((size_t)n_utts * 4, 256);
}

int st_melspec_f32(const float* audio, const int64_t* sample_offsets, int n_utts, int64_t max_samples,
                   const float* mel_basis, int n_mels, int n_fft, int hop, const int64_t* frame_offsets,
                   int64_t total_frames, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  ST_REQUIRE(audio && sample_offsets && mel_basis && frame_offsets && out && workspace, "melspec: null argument");
  ST_REQUIRE(n_fft == NFFT, "melspec: only n_fft = 512 (the reference default, preprocessing.py:36) is built");
  ST_REQUIRE(n_utts > 0 && n_mels > 0 && hop > 0 && max_samples > NFFT / 2 && total_frames > 0, "melspec: bad shape");
  ST_REQUIRE(workspace_bytes >= st_melspec_ws(n_utts, total_frames, n_mels), "melspec: workspace too small");
  hipStream_t s = st::as_stream(stream);
  float* melpow = reinterpret_cast<float*>(workspace);
  unsigned* umax = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(workspace) +
                                               st::round_up((size_t)total_frames * n_mels * sizeof(float), 256));
  if (hipMemsetAsync(umax, 0, (size_t)n_utts * 4, s) != hipSuccess) {
    st::set_error("melspec: memset failed");
    return ST_ELAUNCH;
  }
  const unsigned max_frames = (unsigned)(1 + max_samples / hop);
  hipLaunchKernelGGL(mel_frame_kernel, dim3(max_frames, n_utts), dim3(256), 0, s, audio,
                     reinterpret_cast<const long*>(sample_offsets), mel_basis, n_mels, hop,
                     reinterpret_cast<const long*>(frame_offsets), melpow, umax);
  hipLaunchKernelGGL(mel_normalize_kernel, dim3(n_utts), dim3(1024), 0, s, melpow,
                     reinterpret_cast<const long*>(sample_offsets), reinterpret_cast<const long*>(frame_offsets),
                     n_mels, hop, umax, out);
  return st::check_launch("melspec");
}

}  // extern "C"
