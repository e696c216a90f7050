// Frequency-domain convolution for long, wide filters (the model's L8: 32 taps, 250 -> 2000 channels, 66 % of the
// MACs of the step; speech_model.py:285, tf.nn.conv1d 'SAME' + bias + relu and its gradients).
//
// Time is cut into blocks of V output frames; a block's receptive window has N = V + W - 1 frames.  With the
// length-N DFT along time (real input: bins k = 0 .. N/2), per bin and per (row = utterance x block):
//     forward    Y[k] = S[k] . conj(G[k])            S = DFT of x[jV - pad_left + n],  n < N   (overlap-save)
//     to input   X[k] = D[k] . G[k]                  D = DFT of dz[jV + pad_left - (W-1) + n], outputs m >= W-1
//     filters    Q[k] = sum_rows conj-pairing S[k]^T Z[k]   Z = DFT of dz[jV + t'], t' < V, zero padded; lags w < W
// (G = DFT of the zero-padded filter) -- one complex [rows x Cin] x [Cin x Cout] product per bin instead of W taps
// per frame: 8 bins-flops per 63 frames against 64 flops per frame, a 10x cut of the multiplications (N = 94).
// The complex products run as REAL GEMMs on the exact-fp32 MFMA convolution kernel (gemm_nn_batched, one bin per
// XCD at a time) through the embedding [re | im] x [[Gr, -Gi], [Gi, Gr]]; the transforms are direct DFTs
// (N <= 128: a few hundred multiply-adds per value, no butterflies, fp32 twiddles from a float64 table) in
// thread-per-channel kernels whose twiddles come through the scalar cache.
// Accuracy: every step is fp32 with exact products; the direct DFT sums N terms -- errors of 1e-6 of the tensor
// scale, the same class as the fp32 accumulation of the direct kernel (tests/test_gpu_fft_conv.py).
#include <algorithm>

#include "st_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NMAX = 128;
constexpr int CH = 64;                       // channels per workgroup (one per lane), 4 waves share them

int npad_of(int c) { return c <= 32 ? 32 : (c <= 64 ? 64 : (int)st::round_up(c, 128)); }

struct Plan {
  int n, v, blocks, bins, rows, rows_pad;
};

// N even in [2W, 128]: fewest GEMM tile-steps  ceil(bins / 8) * 8 * round_up(rows, 128)  (8 bins run side by side,
// one per XCD; row tiles are 128 deep)
Plan make_plan(int width, int frames, int batch) {
  Plan best{};
  long best_cost = -1;
  for (int n = std::max(2 * width, 16); n <= NMAX; n += 2) {
    Plan p;
    p.n = n;
    p.v = n - width + 1;
    p.blocks = st::ceil_div(frames, p.v);
    p.bins = n / 2 + 1;
    p.rows = batch * p.blocks;
    p.rows_pad = (int)st::round_up(p.rows, 128);
    const long cost = (long)st::round_up(p.bins, 8) * p.rows_pad;
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = p; }
  }
  return best;
}

// twiddle table tw[j] = (cos, sin)(2 pi j / n), j < n, in a caller-provided device buffer (st_conv1d_fft_twiddles_f32)
__global__ void twiddle_kernel(int n, f32x2* __restrict__ tw) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) {
    float sn, cs;
    sincospif(2.0f * (float)j / (float)n, &sn, &cs);
    tw[j] = f32x2{cs, sn};
  }
}

struct RowsIn {                  // a padded NWC tensor, read frame-wise
  const float* base;
  long batch_stride;             // floats between utterances
  int c_pitch, channels_read;    // floats per frame; channels to transform (<= c_pitch)
  int t_lo, t_hi;                // readable frames [t_lo, t_hi) relative to frame 0 (halos included: they hold zeros)
};

// ---- forward DFT of time segments ---------------------------------------------------------------------------
// row = b * blocks + j  ->  out[k][row][c] = sum_{n < seg_len} x[b][j * v + start + n][c] * e^{-2 pi i k n / N}
// stored [bins][rows_pad][2 * half]: re at column c, im at column half + c; zero for rows >= rows and for the
// columns in [channels_read, half).  outT (optional): the same values as [bins][2 * half][rows_pad].
// One lane per channel; the four waves of a workgroup take bins g, g+4, ... four at a time per pass over the
// segment (8 multiply-adds per LDS read), twiddles through the scalar cache (their index is wave-uniform).
__global__ __launch_bounds__(256) void dft_rows_kernel(RowsIn x, int blocks, int rows, int rows_pad, int n, int v, int start,
                                                       int seg_len, int bins, int half, const f32x2* __restrict__ tw,
                                                       float* __restrict__ out, float* __restrict__ outT) {
  __shared__ float seg[NMAX][CH];
  const int row = blockIdx.x, c0 = blockIdx.y * CH;
  const int lane = threadIdx.x & 63, g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = c0 + lane;
  const bool live = row < rows && c < x.channels_read;
  const int b = row / blocks, j = row - b * blocks;
  const int t0 = j * v + start;
  for (int nn = g; nn < n; nn += 4) {
    const int t = t0 + nn;
    float val = 0.f;
    if (live && nn < seg_len && t >= x.t_lo && t < x.t_hi) val = x.base[(long)b * x.batch_stride + (long)t * x.c_pitch + c];
    seg[nn][lane] = val;
  }
  __syncthreads();
  if (c >= half) return;
  const long plane = (long)rows_pad * 2 * half;
  for (int k0 = g; k0 < bins; k0 += 16) {
    float re[4] = {0.f, 0.f, 0.f, 0.f}, im[4] = {0.f, 0.f, 0.f, 0.f};
    int idx[4] = {0, 0, 0, 0};
    for (int nn = 0; nn < n; ++nn) {
      const float xv = seg[nn][lane];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x2 w = tw[__builtin_amdgcn_readfirstlane(idx[q])];
        re[q] = fmaf(xv, w[0], re[q]);
        im[q] = fmaf(-xv, w[1], im[q]);
        idx[q] += k0 + 4 * q;                              // (k n) mod N, k = k0 + 4 q < N
        if (idx[q] >= n) idx[q] -= n;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = k0 + 4 * q;
      if (k < bins) {
        float* o = out + (long)k * plane + (long)row * 2 * half;
        o[c] = re[q];
        o[half + c] = im[q];
        if (outT) {
          float* ot = outT + (long)k * plane;
          ot[(long)c * rows_pad + row] = re[q];
          ot[(long)(half + c) * rows_pad + row] = im[q];
        }
      }
    }
  }
}

// ---- inverse DFT of spectra back to frames, with the layer epilogue -----------------------------------------
// in [bins][rows_pad][2 * half] (re | im).  For row = (b, j) and output offsets m in [m0, m0 + v):
//   val[m][c] = (1 / N) * sum_k w_k (re[k][c] cos(2 pi k m / N) - im[k][c] sin(2 pi k m / N)),  w_k = 1 for k = 0 and
//   k = N / 2, else 2;   frame t = j * v + (m - m0) < frames gets  act(val + bias[c])  or  mask * val.
struct RowsOut {
  float* base;
  long batch_stride;
  int c_pitch, channels, frames;
};
__global__ __launch_bounds__(256) void idft_rows_kernel(const float* __restrict__ in, int blocks, int rows, int rows_pad, int n,
                                                        int v, int m0, int bins, int half, const f32x2* __restrict__ tw,
                                                        RowsOut y, const float* __restrict__ bias, int relu,
                                                        const float* __restrict__ mask, long mask_batch_stride,
                                                        int mask_c_pitch) {
  __shared__ float sre[NMAX / 2 + 1][CH], sim[NMAX / 2 + 1][CH];
  const int row = blockIdx.x, c0 = blockIdx.y * CH;
  if (row >= rows) return;
  const int lane = threadIdx.x & 63, g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = c0 + lane;
  const long plane = (long)rows_pad * 2 * half;
  const float inv_n = 1.f / (float)n;
  for (int k = g; k < bins; k += 4) {
    const float wk = (k == 0 || 2 * k == n) ? inv_n : 2.f * inv_n;
    const float* src = in + (long)k * plane + (long)row * 2 * half;
    sre[k][lane] = c < half ? src[c] * wk : 0.f;
    sim[k][lane] = c < half ? src[half + c] * wk : 0.f;
  }
  __syncthreads();
  if (c >= y.c_pitch) return;
  const int b = row / blocks, j = row - b * blocks;
  const float bv = (bias && c < y.channels) ? bias[c] : 0.f;
  for (int t0 = g; t0 < v; t0 += 16) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int idx[4] = {0, 0, 0, 0}, step[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) step[q] = (m0 + t0 + 4 * q) % n;
    for (int k = 0; k < bins; ++k) {
      const float r = sre[k][lane], i = sim[k][lane];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x2 w = tw[__builtin_amdgcn_readfirstlane(idx[q])];
        acc[q] = fmaf(r, w[0], acc[q]);
        acc[q] = fmaf(-i, w[1], acc[q]);
        idx[q] += step[q];                                 // (k m) mod N
        if (idx[q] >= n) idx[q] -= n;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int tl = t0 + 4 * q, t = j * v + tl;
      if (tl < v && t < y.frames) {
        float val = c < y.channels ? acc[q] + bv : 0.f;    // pad channels stay zero
        if (relu) val = fmaxf(val, 0.f);
        if (mask) val = mask[(long)b * mask_batch_stride + (long)t * mask_c_pitch + c] > 0.f ? val : 0.f;
        y.base[(long)b * y.batch_stride + (long)t * y.c_pitch + c] = val;
      }
    }
  }
}

// ---- filters -> their spectra in the two GEMM operand layouts -------------------------------------------------
// G[k][c][o] = sum_w F[w][c][o] e^{-2 pi i k w / N}.
//  forward operand  gfwd [bins][2 cpi][2 npo]:  [[Gr, -Gi], [Gi, Gr]]      (Y = S conj(G))
//  from packed [w * cpi + c][npo]; one thread per (c, o), o fastest (coalesced reads and writes).
// (WT = compile-time width: the taps stay in registers; WT = 0: run-time width, taps in scratch)
template <int WT>
__global__ __launch_bounds__(256) void filters_dft_fwd_kernel(const float* __restrict__ packed, int width_rt, int cin, int cout,
                                                              int cpi, int npo, int n, int bins,
                                                              const f32x2* __restrict__ tw, float* __restrict__ gfwd) {
  const int width = WT ? WT : width_rt;
  const int o = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
  if (o >= npo) return;
  float f[WT ? WT : 64];
  const bool live = c < cin && o < cout;
#pragma unroll
  for (int w = 0; w < width; ++w) f[w] = live ? packed[((long)w * cpi + c) * npo + o] : 0.f;
  const long plane = (long)2 * cpi * 2 * npo;
  for (int k = 0; k < bins; ++k) {
    float gr = 0.f, gi = 0.f;
    int idx = 0;
#pragma unroll
    for (int w = 0; w < width; ++w) {
      const f32x2 t = tw[idx];
      gr = fmaf(f[w], t[0], gr);
      gi = fmaf(-f[w], t[1], gi);
      idx += k;
      if (idx >= n) idx -= n;
    }
    float* g = gfwd + (long)k * plane;
    g[(long)c * 2 * npo + o] = gr;
    g[(long)c * 2 * npo + npo + o] = -gi;
    g[(long)(cpi + c) * 2 * npo + o] = gi;
    g[(long)(cpi + c) * 2 * npo + npo + o] = gr;
  }
}

//  back-prop operand  gbwd [bins][2 cpo][2 npi]:  rows (re o | im o), columns (re c | im c):  [[Gr^T, Gi^T], [-Gi^T, Gr^T]]
//  from the flipped / transposed copy packed_t [w' * cpo + o][npi] with w' = W - 1 - w (c fastest there).
template <int WT>
__global__ __launch_bounds__(256) void filters_dft_bwd_kernel(const float* __restrict__ packed_t, int width_rt, int cin, int cout,
                                                              int cpo, int npi, int n, int bins,
                                                              const f32x2* __restrict__ tw, float* __restrict__ gbwd) {
  const int width = WT ? WT : width_rt;
  const int c = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y;
  if (c >= npi) return;
  float f[WT ? WT : 64];
  const bool live = c < cin && o < cout;
#pragma unroll
  for (int w = 0; w < width; ++w) f[w] = live ? packed_t[((long)(width - 1 - w) * cpo + o) * npi + c] : 0.f;
  const long plane = (long)2 * cpo * 2 * npi;
  for (int k = 0; k < bins; ++k) {
    float gr = 0.f, gi = 0.f;
    int idx = 0;
#pragma unroll
    for (int w = 0; w < width; ++w) {
      const f32x2 t = tw[idx];
      gr = fmaf(f[w], t[0], gr);
      gi = fmaf(-f[w], t[1], gi);
      idx += k;
      if (idx >= n) idx -= n;
    }
    float* g = gbwd + (long)k * plane;
    g[(long)o * 2 * npi + c] = gr;
    g[(long)o * 2 * npi + npi + c] = gi;
    g[(long)(cpo + o) * 2 * npi + c] = -gi;
    g[(long)(cpo + o) * 2 * npi + npi + c] = gr;
  }
}

// ---- filter gradient: spectra of the lag products back to the W taps ---------------------------------------------
// q [bins][2 cpi][2 npo] = [S_r | S_i]^T [Z_r | Z_i]:  Re = P00 + P11, Im = P10 - P01;
// dF[w][c][o] = (1 / N) sum_k w_k (Re cos(2 pi k w / N) - Im sin(2 pi k w / N)) into dpacked [w * cpi + c][npo].
template <int WT>
__global__ __launch_bounds__(256) void filters_idft_kernel(const float* __restrict__ q, int width_rt, int cin, int cout, int cpi,
                                                           int npo, int n, int bins, const f32x2* __restrict__ tw,
                                                           float* __restrict__ dpacked) {
  const int width = WT ? WT : width_rt;
  const int o = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
  if (o >= npo) return;
  float acc[WT ? WT : 64];
#pragma unroll
  for (int w = 0; w < width; ++w) acc[w] = 0.f;
  const bool live = c < cin && o < cout;
  const long plane = (long)2 * cpi * 2 * npo;
  const float inv_n = 1.f / (float)n;
  if (live) {
    for (int k = 0; k < bins; ++k) {
      const float* p = q + (long)k * plane;
      const float wk = (k == 0 || 2 * k == n) ? inv_n : 2.f * inv_n;
      const float re = (p[(long)c * 2 * npo + o] + p[(long)(cpi + c) * 2 * npo + npo + o]) * wk;
      const float im = (p[(long)(cpi + c) * 2 * npo + o] - p[(long)c * 2 * npo + npo + o]) * wk;
      int idx = 0;
#pragma unroll
      for (int w = 0; w < width; ++w) {
        const f32x2 t = tw[idx];
        acc[w] = fmaf(re, t[0], acc[w]);
        acc[w] = fmaf(-im, t[1], acc[w]);
        idx += k;
        if (idx >= n) idx -= n;
      }
    }
  }
#pragma unroll
  for (int w = 0; w < width; ++w) dpacked[((long)w * cpi + c) * npo + o] = acc[w];
}

bool tensor_ok(const st_tensor3* t) {
  return t && t->base && t->batch > 0 && t->frames > 0 && t->channels > 0 && t->halo >= 0 && t->c_pitch % 16 == 0 &&
         t->c_pitch >= t->channels && t->t_pitch >= t->halo + t->frames;
}

RowsIn rows_in(const st_tensor3& t, int channels_read) {
  RowsIn r;
  r.base = t.base + (long)t.halo * t.c_pitch;        // frame 0 of utterance 0
  r.batch_stride = (long)t.t_pitch * t.c_pitch;
  r.c_pitch = t.c_pitch;
  r.channels_read = channels_read;
  r.t_lo = -t.halo;
  r.t_hi = t.t_pitch - t.halo;
  return r;
}

void launch_dft(const st_tensor3& t, const Plan& pl, int start, int seg_len, int half, const f32x2* tw, float* out, float* outT,
                hipStream_t s) {
  hipLaunchKernelGGL(dft_rows_kernel, dim3(pl.rows_pad, st::ceil_div(half, CH)), dim3(256), 0, s, rows_in(t, t.c_pitch), pl.blocks,
                     pl.rows, pl.rows_pad, pl.n, pl.v, start, seg_len, pl.bins, half, tw, out, outT);
}

}  // namespace

extern "C" {

int st_conv1d_fft_plan(int width, int frames, int batch, int* n, int* valid, int* blocks, int* bins, int* rows_pad) {
  ST_REQUIRE(width >= 2 && 2 * width <= NMAX && width <= 64 && frames > 0 && batch > 0, "fft plan: filter width must be in [2, 64]");
  const Plan p = make_plan(width, frames, batch);
  if (n) *n = p.n;
  if (valid) *valid = p.v;
  if (blocks) *blocks = p.blocks;
  if (bins) *bins = p.bins;
  if (rows_pad) *rows_pad = p.rows_pad;
  return ST_OK;
}

size_t st_conv1d_fft_filter_floats(int width, int frames, int batch, int cin_pitch, int cout_pitch, int cin, int cout,
                                   int backward) {
  const Plan p = make_plan(width, frames, batch);
  return backward ? (size_t)p.bins * 2 * cout_pitch * 2 * npad_of(cin) : (size_t)p.bins * 2 * cin_pitch * 2 * npad_of(cout);
}

int st_conv1d_fft_twiddles_f32(int width, int frames, int batch, float* tw, size_t tw_floats, void* stream) {
  ST_REQUIRE(width >= 2 && 2 * width <= NMAX && tw, "fft twiddles: bad argument");
  const Plan p = make_plan(width, frames, batch);
  ST_REQUIRE(tw_floats >= 2 * (size_t)p.n, "fft twiddles: table needs 2 * n floats");
  hipLaunchKernelGGL(twiddle_kernel, dim3(1), dim3(NMAX), 0, st::as_stream(stream), p.n, reinterpret_cast<f32x2*>(tw));
  return st::check_launch("fft twiddles");
}

int st_conv1d_fft_filters_f32(const float* packed, const float* packed_t, int width, int frames, int batch, int cin,
                              int cout, int cin_pitch, int cout_pitch, const float* twiddles, float* gfwd, float* gbwd,
                              void* stream) {
  ST_REQUIRE(width >= 2 && width <= 64 && 2 * width <= NMAX && cin_pitch % 16 == 0 && cout_pitch % 16 == 0,
             "fft filters: bad shape");
  ST_REQUIRE(npad_of(cout) % 128 == 0 && npad_of(cin) % 128 == 0, "fft filters: both channel counts must pack to multiples of 128");
  hipStream_t s = st::as_stream(stream);
  const Plan p = make_plan(width, frames, batch);
  ST_REQUIRE(twiddles, "fft filters: twiddle table missing");
  const f32x2* tw = reinterpret_cast<const f32x2*>(twiddles);
  const int npo = npad_of(cout), npi = npad_of(cin);
  if (gfwd) {
    ST_REQUIRE(packed, "fft filters: packed filters missing");
    // rows of pad channels (c in [cin, cin_pitch)) are written as zeros by the kernel's `live` test
    const dim3 grid(st::ceil_div(npo, 256), cin_pitch);
    if (width == 32) hipLaunchKernelGGL(filters_dft_fwd_kernel<32>, grid, dim3(256), 0, s, packed, width, cin, cout, cin_pitch, npo, p.n, p.bins, tw, gfwd);
    else hipLaunchKernelGGL(filters_dft_fwd_kernel<0>, grid, dim3(256), 0, s, packed, width, cin, cout, cin_pitch, npo, p.n, p.bins, tw, gfwd);
  }
  if (gbwd) {
    ST_REQUIRE(packed_t, "fft filters: flipped / transposed filters missing");
    const dim3 grid(st::ceil_div(npi, 256), cout_pitch);
    if (width == 32) hipLaunchKernelGGL(filters_dft_bwd_kernel<32>, grid, dim3(256), 0, s, packed_t, width, cin, cout, cout_pitch, npi, p.n, p.bins, tw, gbwd);
    else hipLaunchKernelGGL(filters_dft_bwd_kernel<0>, grid, dim3(256), 0, s, packed_t, width, cin, cout, cout_pitch, npi, p.n, p.bins, tw, gbwd);
  }
  return st::check_launch("fft filters");
}

// floats: sf, sft (each), and the scratch spectra the three passes need
size_t st_conv1d_fft_sf_floats(const st_tensor3* x, const st_tensor3* y, int width) {
  if (!x || !y) return 0;
  const Plan p = make_plan(width, y->frames, y->batch);
  return (size_t)p.bins * p.rows_pad * 2 * x->c_pitch;
}

size_t st_conv1d_fft_ws(const st_tensor3* x, const st_tensor3* y, int width) {
  if (!x || !y) return 0;
  const Plan p = make_plan(width, y->frames, y->batch);
  const size_t nf = 2 * (size_t)npad_of(y->channels), kb = 2 * (size_t)y->c_pitch, ka = 2 * (size_t)x->c_pitch,
               nb = 2 * (size_t)npad_of(x->channels);
  const size_t fwd = (size_t)p.bins * p.rows_pad * nf;                             // Yf
  const size_t bwd = (size_t)p.bins * p.rows_pad * (kb + nb);                      // Df + Xf
  const size_t wgr = (size_t)p.bins * p.rows_pad * nf + (size_t)p.bins * ka * nf;  // Zf + Qf
  return (std::max(fwd, std::max(bwd, wgr)) + 64) * sizeof(float);
}

int st_conv1d_nwc_fwd_fft_f32(const st_tensor3* x, const float* gfwd, const float* bias, int width, int pad_left, int relu,
                              const st_tensor3* y, const float* twiddles, float* sf, float* sft, void* workspace,
                              size_t workspace_bytes, void* stream) {
  ST_REQUIRE(tensor_ok(x) && tensor_ok(y) && gfwd && sf && workspace, "conv fft fwd: bad argument");
  ST_REQUIRE(x->batch == y->batch && x->frames == y->frames && pad_left >= 0 && pad_left < width, "conv fft fwd: stride-1 SAME layers only");
  ST_REQUIRE(npad_of(y->channels) % 128 == 0 && workspace_bytes >= st_conv1d_fft_ws(x, y, width), "conv fft fwd: workspace / shape");
  hipStream_t s = st::as_stream(stream);
  const Plan p = make_plan(width, y->frames, y->batch);
  ST_REQUIRE(twiddles, "conv fft: twiddle table missing");
  const f32x2* tw = reinterpret_cast<const f32x2*>(twiddles);
  const int ka = 2 * x->c_pitch, npo = npad_of(y->channels), nf = 2 * npo;
  float* yf = reinterpret_cast<float*>(workspace);
  launch_dft(*x, p, -pad_left, p.n, x->c_pitch, tw, sf, sft, s);
  if (int e = st::gemm_nn_batched(sf, ka, (long)p.rows_pad * ka, gfwd, (long)ka * nf, yf, nf, (long)p.rows_pad * nf, p.rows_pad, ka,
                                  nf, p.bins, s))
    return e;
  RowsOut out{y->base + (long)y->halo * y->c_pitch, (long)y->t_pitch * y->c_pitch, y->c_pitch, y->channels, y->frames};
  hipLaunchKernelGGL(idft_rows_kernel, dim3(p.rows, st::ceil_div(y->c_pitch, CH)), dim3(256), 0, s, yf, p.blocks, p.rows,
                     p.rows_pad, p.n, p.v, 0, p.bins, npo, tw, out, bias, relu, (const float*)nullptr, 0L, 0);
  return st::check_launch("conv fft fwd");
}

int st_conv1d_nwc_bwd_data_fft_f32(const st_tensor3* dz, const float* gbwd, int width, int pad_left, const st_tensor3* act,
                                   const st_tensor3* dx, const float* twiddles, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  ST_REQUIRE(tensor_ok(dz) && tensor_ok(dx) && gbwd && workspace, "conv fft bwd_data: bad argument");
  ST_REQUIRE(dz->batch == dx->batch && dz->frames == dx->frames && pad_left >= 0 && pad_left < width, "conv fft bwd_data: stride-1 layers only");
  ST_REQUIRE(npad_of(dx->channels) % 128 == 0 && workspace_bytes >= st_conv1d_fft_ws(dx, dz, width), "conv fft bwd_data: workspace / shape");
  if (act) ST_REQUIRE(tensor_ok(act) && act->batch == dx->batch && act->frames == dx->frames && act->c_pitch >= dx->c_pitch,
                      "conv fft bwd_data: mask tensor mismatch");
  hipStream_t s = st::as_stream(stream);
  const Plan p = make_plan(width, dz->frames, dz->batch);
  ST_REQUIRE(twiddles, "conv fft: twiddle table missing");
  const f32x2* tw = reinterpret_cast<const f32x2*>(twiddles);
  const int kb = 2 * dz->c_pitch, npi = npad_of(dx->channels), nb = 2 * npi;
  float* df = reinterpret_cast<float*>(workspace);
  float* xf = df + (size_t)p.bins * p.rows_pad * kb;
  // overlap-save on dz: segment of block j starts at frame j*V + pad_left - (W - 1); outputs m in [W-1, N)
  launch_dft(*dz, p, pad_left - (width - 1), p.n, dz->c_pitch, tw, df, nullptr, s);
  if (int e = st::gemm_nn_batched(df, kb, (long)p.rows_pad * kb, gbwd, (long)kb * nb, xf, nb, (long)p.rows_pad * nb, p.rows_pad, kb,
                                  nb, p.bins, s))
    return e;
  RowsOut out{dx->base + (long)dx->halo * dx->c_pitch, (long)dx->t_pitch * dx->c_pitch, dx->c_pitch, dx->channels, dx->frames};
  hipLaunchKernelGGL(idft_rows_kernel, dim3(p.rows, st::ceil_div(dx->c_pitch, CH)), dim3(256), 0, s, xf, p.blocks, p.rows,
                     p.rows_pad, p.n, p.v, width - 1, p.bins, npi, tw, out, (const float*)nullptr, 0,
                     act ? act->base + (long)act->halo * act->c_pitch : nullptr, act ? (long)act->t_pitch * act->c_pitch : 0L,
                     act ? act->c_pitch : 0);
  return st::check_launch("conv fft bwd_data");
}

int st_conv1d_nwc_bwd_filter_fft_f32(const st_tensor3* x, const st_tensor3* dz, const float* sft, int width,
                                     const float* twiddles, float* dpacked, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  ST_REQUIRE(tensor_ok(x) && tensor_ok(dz) && sft && dpacked && workspace, "conv fft bwd_filter: bad argument");
  ST_REQUIRE(x->batch == dz->batch && x->frames == dz->frames, "conv fft bwd_filter: stride-1 layers only");
  ST_REQUIRE(npad_of(dz->channels) % 128 == 0 && workspace_bytes >= st_conv1d_fft_ws(x, dz, width), "conv fft bwd_filter: workspace / shape");
  hipStream_t s = st::as_stream(stream);
  const Plan p = make_plan(width, dz->frames, dz->batch);
  ST_REQUIRE(twiddles, "conv fft: twiddle table missing");
  const f32x2* tw = reinterpret_cast<const f32x2*>(twiddles);
  const int ka = 2 * x->c_pitch, npo = npad_of(dz->channels), nf = 2 * npo;
  float* zf = reinterpret_cast<float*>(workspace);
  float* qf = zf + (size_t)p.bins * p.rows_pad * nf;
  // Z: the V frames of block j, zero padded to N
  launch_dft(*dz, p, 0, p.v, npo, tw, zf, nullptr, s);
  // Q[bin] = SfT[bin] (2 cpi x rows_pad) * Zf[bin] (rows_pad x 2 npo)
  if (int e = st::gemm_nn_batched(sft, p.rows_pad, (long)ka * p.rows_pad, zf, (long)p.rows_pad * nf, qf, nf, (long)ka * nf, ka,
                                  p.rows_pad, nf, p.bins, s))
    return e;
  const dim3 grid(st::ceil_div(npo, 256), x->c_pitch);
  if (width == 32) hipLaunchKernelGGL(filters_idft_kernel<32>, grid, dim3(256), 0, s, qf, width, x->channels, dz->channels, x->c_pitch, npo, p.n, p.bins, tw, dpacked);
  else hipLaunchKernelGGL(filters_idft_kernel<0>, grid, dim3(256), 0, s, qf, width, x->channels, dz->channels, x->c_pitch, npo, p.n, p.bins, tw, dpacked);
  return st::check_launch("conv fft bwd_filter");
}

}  // extern "C"
