// Frequency-domain convolution (speech_model.py:279-285, tf.nn.conv1d 'SAME' + bias + relu and its gradients): the
// model's 32-tap 250 -> 2000 layer (66 % of the MACs of the step), its seven 7-tap layers and -- on its polyphase view,
// see include/speecht_hip.h -- the stride-2 48-tap first layer.
//
// Time is cut into blocks of V = 64 output frames; a block's receptive window has N = V + W - 1 frames (95 for 32
// taps).  With the length-N DFT along time (real input: bins k = 0 .. N/2), per bin and per (row = utterance x block):
//     forward    Y[k] = S[k] . conj(G[k])       S = DFT of x[jV - pad_left + n], n < N          (overlap-save)
//     to input   X[k] = Z[k] . G[k]             Z = DFT of dz[jV + t'], t' < V, zero padded      (overlap-add: frame
//                                               jV + t' sums block j at m = t' + pad_left and its two neighbours)
//     filters    Q[k] = sum_rows S[k]^T conj(Z[k]),  lags w < W
//     bias       sum_rows Z[0]                  (bin 0 is the plain sum of the block's frames)
// (G = DFT of the zero-padded filter): one complex [rows x Cin] x [Cin x Cout] product per bin instead of W taps per
// frame -- for 32 taps 48 bins x 8 flops per 64 frames against 64 flops per frame, a 10x cut of the multiplications.
// The complex products run as REAL GEMMs on the exact-fp32 MFMA convolution kernels (st::gemm_nn_batched, one bin per
// XCD at a time; st::gemm_tn_batched for the lag products) through the embedding [re | im] x [[Gr, -Gi], [Gi, Gr]]
// (back-prop: x its transpose, read in place).
// The transforms are dense DFTs on the same matrix instruction (v_mfma_f32_32x32x2_f32): a wavefront owns 32 channels
// of one row, reads the 96 x 96 (forward) or 64 x 96 (inverse) DFT matrix from LDS as MFMA A fragments, loads the
// frames straight from the NWC tensor as B fragments (128-byte runs per frame) in stages that stay one ahead of the
// MFMAs, and writes 128-byte runs; the inverse carries the bias / ReLU / mask epilogue.  The wide layer's transforms
// are bound by the 330 MB they move (3.5 TB/s), the narrow ones by latency (35 MB in 15-18 us).
// Accuracy: fp32 throughout, exact products, N-term sums: ~1e-6 of the tensor scale (tests/test_gpu_fft_conv.py).
#include <algorithm>

#include "st_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int V = 64;                        // output frames per block
constexpr int KP = 96;                       // padded DFT length (N = V + W - 1 <= 96) and padded 2 * bins
constexpr int HB = KP / 2;                   // rows [0, HB) of a spectrum matrix are real parts, [HB, KP) imaginary
// device tables (st_conv1d_fft_tables_f32), floats:
constexpr int T_FS = 0;                      // forward DFT of N-frame segments          [KP][KP]
constexpr int T_FZ = T_FS + KP * KP;         // forward DFT of V-frame, zero-padded ones [KP][KP]
constexpr int T_IY = T_FZ + KP * KP;         // inverse at m = t'                        [V][KP]
constexpr int T_IX = T_IY + V * KP;          // inverse at m = t' + pl, t' + V + pl, t' - V + pl   [3][V][KP]
constexpr int T_TW = T_IX + 3 * V * KP;      // (cos, sin)(2 pi j / N), j < N            [128][2]
constexpr int T_FW = T_TW + 256;             // (cos, sin)(2 pi k w / N), k < bins, w < W: [HB][W][2] (filter kernels: one
constexpr int T_FP = T_FW + HB * 34 * 2;     // contiguous row per bin, read with wide scalar loads)
// T_FP: the forward matrix T_FS with its columns in the order the inverse transform's accumulators hold frames, k-step
// interleaved [FUSE_Q][KP][2] (idft_dft_rows_kernel): what the LAYER BELOW stages when it hands its frames over in registers
constexpr int FUSE_BLOCKS = 8, FUSE_HALO = 4, FUSE_Q = 32 + FUSE_HALO;
constexpr int T_IW = T_FP + FUSE_Q * KP * 2; // inverse at the WHOLE window of a block [KP][KP]: rows 0..63 m = t' + pl (the block's own
                                             // frames), 64 + e m = e (e < pl: lands in the block before), 68 + e m = V + pl + e (the block after)
constexpr int T_ZP = T_IW + KP * KP;         // T_FZ with its columns in accumulator order, k-step interleaved [32][KP][2]
constexpr int T_END = T_ZP + 32 * KP * 2;
// window index of the frame lanes h supply in k-step q of the fused transform (-1: none): q < 32 pairs the frames of accumulator
// register q & 15 of the 32-frame half q >> 4; k-step 32 + e pairs frame e of the left halo (h = 0) with frame e of the right one
__host__ __device__ inline int fused_column(int q, int hh, int n, int pad_left) {
  const int right = n - V - pad_left;
  if (q < 32) return pad_left + 32 * (q >> 4) + (q & 3) + 8 * ((q & 15) >> 2) + 4 * hh;
  if (hh == 0) return (q - 32) < pad_left ? (q - 32) : -1;
  return (q - 32) < right ? V + pad_left + (q - 32) : -1;
}

int npad_of(int c) { return c <= 32 ? 32 : (c <= 64 ? 64 : (int)st::round_up(c, 128)); }
// spectra of a tensor with channel pitch cp keep `half_of(cp)` columns for the real and for the imaginary parts: a
// multiple of 64, so that [re | im] rows tile the filter-gradient kernel (128 columns) whatever the pitch
int half_of(int cp) { return (int)st::round_up(cp, 64); }

struct Plan {
  int n, blocks, bins, rows, rows_pad;
};
// rows of a bin's plane are padded to whole GEMM row tiles.  ROWS_F32 (the fp32 entry points): 64 -- the persistent per-bin
// kernel's tile height; the 128-row kernels take a partial last tile.  Real training batches change (B, max_T) every step
// (speech_input.py:37-45): with 32 utterances every odd block count padded to 128 rows wasted up to 37 % of the per-bin products
// (601 input frames: 160 rows -> 256; scripts/bench_varlen_train.py --sweep).  The plane entry points (bf16 matrix pipe: 256- and
// 128-row tiles without partial-tile handling) and the sizing functions keep 128 (sizes: an upper bound for both).
constexpr int ROWS_F32 = 64;
Plan make_plan(int width, int frames, int batch, int row_granule = 128) {
  Plan p;
  p.n = V + width - 1;
  p.blocks = st::ceil_div(frames, V);
  p.bins = p.n / 2 + 1;
  p.rows = batch * p.blocks;
  p.rows_pad = (int)st::round_up(p.rows, row_granule);
  return p;
}

__global__ void tables_kernel(int n, int pad_left, float* __restrict__ t) {
  const int bins = n / 2 + 1, width = n - V + 1;
  const float inv_n = 1.f / (float)n;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < T_END; i += gridDim.x * blockDim.x) {
    float val = 0.f;
    if (i < T_IY) {                                        // forward matrices: row m = (re | im, bin), column = frame
      const int z = i >= T_FZ, r = (i - (z ? T_FZ : T_FS)) / KP, col = (i - (z ? T_FZ : T_FS)) % KP;
      const int k = r < HB ? r : r - HB;
      if (k < bins && col < (z ? V : n)) {
        float sn, cs;
        sincospif(2.0f * (float)((k * col) % n) * inv_n, &sn, &cs);
        val = r < HB ? cs : -sn;
      }
    } else if (i < T_TW) {                                 // inverse matrices: row t', column = (re | im, bin)
      const int which = i < T_IX ? 0 : 1 + (i - T_IX) / (V * KP);
      const int rem = i < T_IX ? i - T_IY : (i - T_IX) % (V * KP);
      const int tp = rem / KP, col = rem % KP;
      const int k = col < HB ? col : col - HB;
      int m = tp;
      if (which == 1) m = tp + pad_left;
      if (which == 2) m = tp + V + pad_left;
      if (which == 3) m = tp - V + pad_left;
      if (k < bins && m >= 0 && m < n) {
        const float wk = (k == 0 || 2 * k == n) ? inv_n : 2.f * inv_n;
        float sn, cs;
        sincospif(2.0f * (float)((k * m) % n) * inv_n, &sn, &cs);
        val = col < HB ? wk * cs : -wk * sn;
      }
    } else if (i < T_FW) {
      const int j = (i - T_TW) / 2;
      if (j < n) {
        float sn, cs;
        sincospif(2.0f * (float)j * inv_n, &sn, &cs);
        val = (i - T_TW) % 2 ? sn : cs;
      }
    } else if (i < T_FP) {
      const int e = (i - T_FW) / 2, k = e / width, w = e % width;
      if (k < bins) {
        float sn, cs;
        sincospif(2.0f * (float)((k * w) % n) * inv_n, &sn, &cs);
        val = (i - T_FW) % 2 ? sn : cs;
      }
    } else if (i < T_IW) {
      const int e = i - T_FP, q = e / (KP * 2), r = (e - q * (KP * 2)) >> 1, hh = e & 1;
      const int k = r < HB ? r : r - HB, col = fused_column(q, hh, n, pad_left);
      if (k < bins && col >= 0 && col < n) {
        float sn, cs;
        sincospif(2.0f * (float)((k * col) % n) * inv_n, &sn, &cs);
        val = r < HB ? cs : -sn;
      }
    } else if (i < T_ZP) {
      const int e = i - T_IW, r = e / KP, col = e % KP;
      const int k = col < HB ? col : col - HB;
      int m = -1;
      if (r < V) m = r + pad_left;
      else if (r < V + 4) m = (r - V) < pad_left ? r - V : -1;
      else if (r < V + 8) m = (r - V - 4) < width - 1 - pad_left ? V + pad_left + (r - V - 4) : -1;
      if (k < bins && m >= 0 && m < n) {
        const float wk = (k == 0 || 2 * k == n) ? inv_n : 2.f * inv_n;
        float sn, cs;
        sincospif(2.0f * (float)((k * m) % n) * inv_n, &sn, &cs);
        val = col < HB ? wk * cs : -wk * sn;
      }
    } else {
      const int e = i - T_ZP, q = e / (KP * 2), r = (e - q * (KP * 2)) >> 1, hh = e & 1;
      const int k = r < HB ? r : r - HB, col = fused_column(q, hh, V, 0);      // frame t' of the zero-padded block
      if (k < bins && col < V) {
        float sn, cs;
        sincospif(2.0f * (float)((k * col) % n) * inv_n, &sn, &cs);
        val = r < HB ? cs : -sn;
      }
    }
    t[i] = val;
  }
}

struct RowsIn {                  // a padded NWC tensor, read frame-wise
  const float* base;             // frame 0 of utterance 0 (fp32 tensor) ...
  const unsigned short* base_b;  // ... or of its bf16 form (same geometry; the kernels' IN_BF variants)
  long batch_stride;             // floats between utterances
  int c_pitch, channels_read;    // floats per frame; channels to transform (<= c_pitch)
  int t_lo, t_hi;                // readable frames [t_lo, t_hi) relative to frame 0 (halos included: they hold zeros)
};

// ---- forward DFT of time segments on the matrix pipe ----------------------------------------------------------
// row = b * blocks + j:  out[k][row][c] = sum_n Wm[(re | im, k)][n] * x[b][j * V + start + n][c], stored
// [bins][rows_pad][2 * half] with re at column c and im at column half + c; rows >= rows and channels >=
// channels_read give zeros.
// One (row, 32 channels) item per wavefront.
// The frames are loaded ONCE for the three 32-row tiles of the DFT matrix (three independent accumulator chains), only
// the `nstages * CH` frame pairs the matrix has non-zero columns for, and in stages of CH pairs: the loads of stage
// st + 1 are in flight while the 3 * CH MFMAs of stage st run.  The matrix sits in LDS pair-interleaved
// ((2s, 2s + 1) of every row adjacent), so an A fragment is one ds_read_b32 at consecutive addresses across the wave.
constexpr int CH = 12;                       // frame (bin) pairs per pipeline stage; KP / 2 = 4 * CH

template <int R, int COUNT = 1>
__device__ inline void stage_matrix(float* __restrict__ dst, const float* __restrict__ src) {      // COUNT x src [R][KP] row-major
  // all loads first, then the LDS writes: a load -> wait -> write loop exposes one memory latency per 4 KB
  constexpr int IT = R * KP / 4 / 256;
  static_assert(IT * 256 * 4 == R * KP, "the matrix must be a whole number of 256-thread passes");
  f32x4 v[COUNT * IT];
#pragma unroll
  for (int it = 0; it < COUNT * IT; ++it) v[it] = reinterpret_cast<const f32x4*>(src)[threadIdx.x + 256 * it];
#pragma unroll
  for (int it = 0; it < COUNT * IT; ++it) {
    const int m = it / IT, i = threadIdx.x + 256 * (it % IT);
    const int row = (i * 4) / KP, k = (i * 4) % KP;          // KP % 4 == 0: four columns of one row
    f32x2* d = reinterpret_cast<f32x2*>(dst + m * R * KP) + ((k >> 1) * R + row);
    d[0] = f32x2{v[it][0], v[it][1]};
    d[R] = f32x2{v[it][2], v[it][3]};
  }
}

// IN_BF: the tensor is read in its bf16 form.  OUTP: 0 -> fp32 spectra `out`; 4 / 5 -> fp32 spectra in the three-product layouts, a row
// [S_r + S_i | S_i | S_r | S_i - S_r] (input spectra) or [Z_r + Z_i | Z_r | Z_i] (gradient spectra) of `half` columns each;
// 1 -> ONE bf16 plane `outb` (bf16 activations:
// the per-bin products run on the bf16 matrix pipe); 3 -> the exact 3-way bf16 split of the fp32 value in three planes
// `out_plane` elements apart (fp32-accurate products from six bf16 terms, conv_bf16.hip).  `dc` (optional): the fp32 value of
// spectrum row 0 -- bin 0's real part, the plain sum of the block's frames -- [rows_pad][half]: the bias gradient is a sum of
// these, and a bf16 spectrum would cost it its cancellation digits.
__device__ __forceinline__ void split3_bits(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  const __bf16 bh = (__bf16)x;
  const float r1 = x - (float)bh;          // exact
  const __bf16 bm = (__bf16)r1;
  const __bf16 bl = (__bf16)(r1 - (float)bm);
  h = __builtin_bit_cast(unsigned short, bh); m = __builtin_bit_cast(unsigned short, bm); l = __builtin_bit_cast(unsigned short, bl);
}
template <int NST, bool IN_BF = false, int OUTP = 0>   // NST stages of CH frame pairs: the matrix has no columns past 2 * NST * CH
__global__ __launch_bounds__(256, 2) void dft_rows_kernel(RowsIn x, const float* __restrict__ wm, int blocks, int rows,
                                                          int rows_pad, int start, int bins, int half, int nchunks,
                                                          float* __restrict__ out, unsigned short* __restrict__ outb,
                                                          size_t out_plane, float* __restrict__ dc, long bin_stride,
                                                          float* __restrict__ out2) {
  __shared__ __attribute__((aligned(16))) float wl[KP * KP];
  const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), total = gridDim.x * 4;
  stage_matrix<KP>(wl, wm);
  __syncthreads();
  // (bin_stride: elements between the planes of consecutive bins -- rows_pad * 2 * half, or twice that when the spectra of a
  // bin are followed by their rotated copy out2 = [S_i | -S_r], the operand of the imaginary lag products)
  const long plane = bin_stride;
  const float* afrag = wl + 2 * l31 + h;
  for (int item = gw; item < rows_pad * nchunks; item += total) {
    const int row = item / nchunks, c = (item - row * nchunks) * 32 + l31;
    f32x16 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    if (row < rows) {
      const int b = row / blocks, j = row - b * blocks;
      const int t0 = j * V + start + h;
      const bool cok = c < x.channels_read;
      const long col = (long)b * x.batch_stride + min(c, x.channels_read - 1);
      const float* src = x.base + col;
      const unsigned short* srcb = x.base_b + col;
      // branch-free: every lane loads from a clamped, readable address; what it must not see is masked off when the
      // value is consumed (a select next to the load would make the wave wait for it there)
      auto load = [&](float (&bf)[CH], int st) {
#pragma unroll
        for (int s = 0; s < CH; ++s) {
          const int t = t0 + 2 * (st * CH + s);
          const long o = (long)min(max(t, x.t_lo), x.t_hi - 1) * x.c_pitch;
          if (IN_BF) bf[s] = __uint_as_float((unsigned)srcb[o] << 16);
          else bf[s] = src[o];
        }
      };
      auto mac = [&](const float (&bf)[CH], int st) {
        const float* a = afrag + st * (CH * KP * 2);
#pragma unroll
        for (int s = 0; s < CH; ++s) {
          const int t = t0 + 2 * (st * CH + s);
          const unsigned keep = (cok && t >= x.t_lo && t < x.t_hi) ? 0xffffffffu : 0u;
          const float v = __uint_as_float(__float_as_uint(bf[s]) & keep);
#pragma unroll
          for (int i = 0; i < 3; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(s * KP + i * 32) * 2], v, acc[i], 0, 0, 0);
        }
      };
      // the scheduling fences keep the compiler from hoisting later stages' loads (and their address registers)
      // above the MFMAs of this one; the stage loop is unrolled at compile time so that the waits are exact counts
      // (`vmcnt(12 + k)`: the older stage has landed, the newer one stays in flight)
      float bfr[2][CH];
      load(bfr[0], 0);
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        if (st + 1 < NST) load(bfr[(st + 1) & 1], st + 1);
        __builtin_amdgcn_sched_barrier(0);
        mac(bfr[st & 1], st);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (OUTP == 4 || OUTP == 5) {
      // the three-product layouts (conv_gemm.hip gemm_nn_g3_kernel): the real part of bin m (accumulator row m < HB) and its
      // imaginary part (row m + HB = m + 32 + 16: the next tile's register r + 8, or the one after's r - 8) sit in the SAME lane,
      // so the sums the three products need cost one VALU add per value here, nothing elsewhere
      constexpr int NP = OUTP == 4 ? 4 : 3;
      if (c < half) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (i == 1 && (r >> 2) >= 2) continue;                      // rows >= HB hold imaginary parts
            const int bin = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float re = acc[i][r];
            const float im = (r >> 2) < 2 ? acc[i + 1][(r + 8) & 15] : acc[(i + 2) % 3][(r - 8) & 15];
            if (bin < bins) {
              float* o = out + (long)bin * plane + (long)row * NP * half + c;
              if (OUTP == 4) { o[0] = re + im; o[half] = im; o[2 * half] = re; o[3 * half] = im - re; }
              else { o[0] = re + im; o[half] = re; o[2 * half] = im; }
            }
          }
      }
    } else if (c < half) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int bin = m < HB ? m : m - HB, col = (m < HB ? 0 : half) + c;
          if (bin < bins) {
            const long o = (long)bin * plane + (long)row * 2 * half + col;
            if (OUTP == 0) {
              out[o] = acc[i][r];
              if (out2) out2[(long)bin * plane + (long)row * 2 * half + (m < HB ? half : 0) + c] = m < HB ? -acc[i][r] : acc[i][r];
            }
            else if (OUTP == 1) outb[o] = __builtin_bit_cast(unsigned short, (__bf16)acc[i][r]);
            else {
              unsigned short sh, sm, sl;
              split3_bits(acc[i][r], sh, sm, sl);
              outb[o] = sh; outb[out_plane + o] = sm; outb[2 * out_plane + o] = sl;
            }
            if (dc && m == 0) dc[(long)row * half + c] = acc[i][r];
          }
        }
    }
  }
}

// ---- inverse DFT of spectra back to frames, with the layer epilogue ---------------------------------------------
// in [bins][rows_pad][2 * half_in] (re | im).  Frame t = j * V + t' of utterance b gets
//   val[c] = sum over TERMS of  Winv[term][t'][(re | im, k)] * in[k][row + off(term)][...]     off = 0, -1, +1 (same b)
// then  act(val + bias[c])  (forward)  or  mask * val  (back-prop); pad channels and nothing beyond y.frames.
struct RowsOut {
  float* base;
  long batch_stride;
  int c_pitch, channels, frames;
  unsigned short* base_b;        // BF kernels: the output is written in bf16 here (same geometry), `base` is not touched
};
// One (row, 32 channels) item per wavefront: both 32-frame halves of the block from one pass over
// the spectra (HP bin pairs of real parts, HP of imaginary parts: 2 * HP >= bins), loads one stage ahead of the MFMAs
// across the terms; a neighbour term only feeds the half of the block it reaches.
// BF (bf16 activations): the output goes out in bf16 (y.base_b) and the ReLU mask source `mask` is a bf16 tensor
template <int TERMS, int HP, bool BF = false>
__global__ __launch_bounds__(256, 2) void idft_rows_kernel(const float* __restrict__ in, const float* __restrict__ winv, int blocks,
                                                           int rows, int rows_pad, int bins, int half_in, int nchunks,
                                                           RowsOut y, const float* __restrict__ bias, int relu,
                                                           const float* __restrict__ mask, long mask_batch_stride,
                                                           int mask_c_pitch) {
  constexpr int NST = 2 * HP / CH;                      // stages per term
  static_assert(2 * HP % CH == 0 && HP <= HB / 2, "pairs per term must fill whole stages");
  __shared__ __attribute__((aligned(16))) float wl[TERMS * V * KP];
  const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), total = gridDim.x * 4;
  stage_matrix<V, TERMS>(wl, winv);
  __syncthreads();
  const long plane = (long)rows_pad * 2 * half_in;
  const float* afrag = wl + 2 * l31 + h;
  for (int item = gw; item < rows * nchunks; item += total) {
    const int row = item / nchunks, c = (item - row * nchunks) * 32 + l31;
    const int b = row / blocks, j = row - b * blocks;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // the ReLU mask of the 64 frames first, branch-free (clamped addresses): gathered next to the stores each load would
    // expose a full memory latency -- for all the compiler knows the output aliases the mask
    float mk[2][16];
    if (mask) {
      const long mcol = (long)b * mask_batch_stride + min(c, mask_c_pitch - 1);
      const float* mp = mask + mcol;
      const unsigned short* mpb = reinterpret_cast<const unsigned short*>(mask) + mcol;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long o = (long)min(j * V + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, y.frames - 1) * mask_c_pitch;
          if (BF) mk[i][r] = __uint_as_float((unsigned)mpb[o] << 16);
          else mk[i][r] = mp[o];
        }
    }
    const float* src = in + (long)row * 2 * half_in + min(c, half_in - 1);
    const bool cok = c < half_in;
    // global stage g = term * NST + st; pair p = st * CH + s of a term: p < HP -> real parts of bins (2p, 2p + 1),
    // else imaginary parts of bins (2 (p - HP), 2 (p - HP) + 1)
    auto load = [&](float (&bf)[CH], int g) {
      const int term = g / NST, st = g % NST;
      const int off = term == 0 ? 0 : (term == 1 ? -1 : 1);
      const bool there = j + off >= 0 && j + off < blocks;                 // wave-uniform
      const float* sp = src + (there ? (long)off * 2 * half_in : 0L);
      // branch-free: clamped, readable addresses; what must not be seen is masked off in mac()
#pragma unroll
      for (int s = 0; s < CH; ++s) {
        const int p = st * CH + s;
        const int bin = (p < HP ? 2 * p : 2 * (p - HP)) + h;
        bf[s] = sp[(long)min(bin, bins - 1) * plane + (p < HP ? 0 : half_in)];
      }
    };
    auto mac = [&](const float (&bf)[CH], int g) {
      const int term = g / NST, st = g % NST;
      const int off = term == 0 ? 0 : (term == 1 ? -1 : 1);
      const bool live = cok && j + off >= 0 && j + off < blocks;
      const float* a = afrag + term * (V * KP);
#pragma unroll
      for (int s = 0; s < CH; ++s) {
        const int p = st * CH + s;
        const int bin = (p < HP ? 2 * p : 2 * (p - HP)) + h;
        const unsigned keep = (live && bin < bins) ? 0xffffffffu : 0u;
        const float v = __uint_as_float(__float_as_uint(bf[s]) & keep);
        const int sidx = p < HP ? p : HB / 2 + (p - HP);                    // pair index into the [V][KP] matrix
        // the neighbours only reach W - 1 frames into this block: frames [0, 32) never see block j + 1, [32, 64) never j - 1
        if (term != 2) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(sidx * V) * 2], v, acc[0], 0, 0, 0);
        if (term != 1) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(sidx * V + 32) * 2], v, acc[1], 0, 0, 0);
      }
    };
    float bfr[2][CH];
    load(bfr[0], 0);
#pragma unroll
    for (int g = 0; g < TERMS * NST; ++g) {                       // fences: see dft_rows_kernel
      if (g + 1 < TERMS * NST) load(bfr[(g + 1) & 1], g + 1);
      __builtin_amdgcn_sched_barrier(0);
      mac(bfr[g & 1], g);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (c < y.c_pitch) {
      const float bv = (bias && c < y.channels) ? bias[c] : 0.f;
      // all values first, then nothing but stores: a load-dependent select next to a predicated store makes the
      // compiler wait for every earlier store as well
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float val = c < y.channels ? acc[i][r] + bv : 0.f;           // pad channels stay zero
          if (relu) val = fmaxf(val, 0.f);
          if (mask) val = mk[i][r] > 0.f ? val : 0.f;
          asm volatile("" : "+v"(val));                               // pinned here: not sunk into the store's predicate block
          acc[i][r] = val;
        }
      float* yp = y.base + (long)b * y.batch_stride + c;
      unsigned short* ypb = y.base_b + (long)b * y.batch_stride + c;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int t = j * V + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (t < y.frames) {
            if (BF) ypb[(long)t * y.c_pitch] = __builtin_bit_cast(unsigned short, (__bf16)acc[i][r]);
            else yp[(long)t * y.c_pitch] = acc[i][r];
          }
        }
    }
  }
}

// ---- inverse DFT of layer i fused with the forward DFT of layer i + 1 (round 4) ---------------------------------------
// The narrow layers' transforms are latency chains (one (block, 32 channels) item per wavefront, 14 + 19 us per layer
// boundary with the chip mostly idle): here the frames an inverse transform has just produced -- bias and ReLU applied --
// go into the NEXT layer's forward transform without leaving the registers.  A workgroup is one (utterance, 32 channels)
// column of up to FUSE_BLOCKS blocks, a wavefront per block.  The inverse transform leaves lane (channel, h) with frames
// {32 i + (r & 3) + 8 (r >> 2) + 4 h}: for every accumulator register r the pair (lanes h = 0, lanes h = 1) IS a valid
// k-step of `v_mfma_f32_32x32x2_f32` for the next DFT once that DFT's matrix has its columns in the same order -- the
// workgroup builds that column-permuted copy of the next layer's T_FS table in LDS.  The next block window also covers
// pad_left frames of the block before and W - 1 - pad_left of the block after: those few frames go through LDS (one
// barrier), as up to FUSE_HALO extra k-steps (left-neighbour frame on the h = 0 lanes, right-neighbour frame on h = 1).
// Writes y (the backward pass needs it for the ReLU mask) and the next layer's input spectra (+ rotated copy) exactly as
// idft_rows_kernel and dft_rows_kernel would; frames past y.frames and pad channels enter the transform as zeros.
template <int HP>
__global__ __launch_bounds__(64 * FUSE_BLOCKS, 1) void idft_dft_rows_kernel(
    const float* __restrict__ in, const float* __restrict__ winv, int blocks, int rows_pad, int bins, int half_in, int nchunks, RowsOut y,
    const float* __restrict__ bias, int relu, const float* __restrict__ wm_next, int n_next, int pl_next, int bins_next, int half_next,
    float* __restrict__ out, long bin_stride, float* __restrict__ out2) {
  constexpr int NST = 2 * HP / CH;
  static_assert(2 * HP % CH == 0 && HP <= HB / 2, "pairs per term must fill whole stages");
  __shared__ __attribute__((aligned(16))) float wl[V * KP];
  __shared__ __attribute__((aligned(16))) float wf[FUSE_Q * KP * 2];          // [k-step][matrix row][h]
  __shared__ float halo[FUSE_BLOCKS][2][FUSE_HALO][32];                       // [block][head | tail][frame][channel]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / nchunks, chunk = blockIdx.x - b * nchunks;
  const int j = wave, row = b * blocks + j, c = chunk * 32 + l31;
  const bool active = j < blocks;
  const int right_next = n_next - V - pl_next;                               // frames of the next block inside the window
  if (tid < 256) stage_matrix<V, 1>(wl, winv);
  // the next layer's forward matrix with its columns in accumulator order (its table T_FP): a straight copy
  {
    constexpr int N4 = FUSE_Q * KP * 2 / 4;
    for (int idx = tid; idx < N4; idx += 64 * FUSE_BLOCKS) reinterpret_cast<f32x4*>(wf)[idx] = reinterpret_cast<const f32x4*>(wm_next)[idx];
  }
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const long plane = (long)rows_pad * 2 * half_in;
  const bool cok = c < half_in;
  const float* src = in + (long)(active ? row : 0) * 2 * half_in + min(c, half_in - 1);
  auto load = [&](float (&bf)[CH], int st) {
#pragma unroll
    for (int s = 0; s < CH; ++s) {
      const int p = st * CH + s;
      const int bin = (p < HP ? 2 * p : 2 * (p - HP)) + h;
      bf[s] = src[(long)min(bin, bins - 1) * plane + (p < HP ? 0 : half_in)];
    }
  };
  float bfr[NST][CH];
  if (active) {
#pragma unroll
    for (int st = 0; st < NST; ++st) load(bfr[st], st);                      // (in flight while the matrices are staged)
  }
  __syncthreads();
  if (active) {
    const float* afrag = wl + 2 * l31 + h;
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < CH; ++s) {
        const int p = st * CH + s;
        const int bin = (p < HP ? 2 * p : 2 * (p - HP)) + h;
        const unsigned keep = (cok && bin < bins) ? 0xffffffffu : 0u;
        const float v = __uint_as_float(__float_as_uint(bfr[st][s]) & keep);
        const int sidx = p < HP ? p : HB / 2 + (p - HP);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[(sidx * V) * 2], v, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[(sidx * V + 32) * 2], v, acc[1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // layer epilogue; what the next transform must see as zero (pad channels, frames past the end) becomes zero here
    const float bv = (bias && c < y.channels) ? bias[c] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = j * V + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float val = c < y.channels ? acc[i][r] + bv : 0.f;
        if (relu) val = fmaxf(val, 0.f);
        val = t < y.frames ? val : 0.f;
        asm volatile("" : "+v"(val));
        acc[i][r] = val;
      }
    if (c < y.c_pitch) {
      float* yp = y.base + (long)b * y.batch_stride + c;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int t = j * V + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (t < y.frames) yp[(long)t * y.c_pitch] = acc[i][r];
        }
    }
    // frames 0 .. 3 of this block (registers 0 .. 3 of the h = 0 lanes) and 60 .. 63 (registers 12 .. 15 of half 1, h = 1 lanes)
#pragma unroll
    for (int e = 0; e < FUSE_HALO; ++e) {
      if (h == 0) halo[j][0][e][l31] = acc[0][e];
      else halo[j][1][e][l31] = acc[1][12 + e];
    }
  }
  __syncthreads();
  if (!active) return;
  f32x16 sacc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[i][r] = 0.f;
  const float* af2 = wf + 2 * l31 + h;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const float v = acc[q >> 4][q & 15];
#pragma unroll
    for (int i = 0; i < 3; ++i) sacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af2[(q * KP + i * 32) * 2], v, sacc[i], 0, 0, 0);
  }
#pragma unroll
  for (int e = 0; e < FUSE_HALO; ++e) {
    // left: frame 64 - pl + e of block j - 1 (tail slot 4 - pl + e); right: frame e of block j + 1
    float v = 0.f;
    if (h == 0) { if (j > 0 && e < pl_next) v = halo[j - 1][1][FUSE_HALO - pl_next + e][l31]; }
    else { if (j + 1 < blocks && e < right_next) v = halo[j + 1][0][e][l31]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) sacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af2[((32 + e) * KP + i * 32) * 2], v, sacc[i], 0, 0, 0);
  }
  if (c < half_next) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int bin = m < HB ? m : m - HB, col = (m < HB ? 0 : half_next) + c;
        if (bin < bins_next) {
          const long o = (long)bin * bin_stride + (long)row * 2 * half_next;
          out[o + col] = sacc[i][r];
          if (out2) out2[o + (m < HB ? half_next : 0) + c] = m < HB ? -sacc[i][r] : sacc[i][r];
        }
      }
  }
}

// ---- back-prop: inverse DFT over the whole window + overlap-add through LDS + mask + the forward DFT of the layer below ---
// idft_rows_kernel<3> gives every block three inverse terms: its own spectra and both neighbours', each neighbour for the
// W - 1 frames it reaches into the block -- twice the matrix instructions and three times the loads of the forward inverse.
// Here a workgroup is one (utterance, 32 channels) column of up to FUSE_BLOCKS blocks, a wavefront per block: the block's
// spectra are inverted ONCE at all N points of its window (table T_IW: the 64 own frames and, in a third 32-row tile,
// the few frames that spill into either neighbour), the spills change hands through LDS (one barrier; sum own + the block
// before's + the block after's, a fixed order), the ReLU mask is applied, dx is stored, and the frames -- still in registers,
// in accumulator order -- go straight into the zero-padded forward DFT that the layer below needs of its dz (table T_ZP of
// THAT layer): its st_conv1d_fft_dz_spectra_f32 launch and the re-read of dx fall away.
template <int HP>
__global__ __launch_bounds__(64 * FUSE_BLOCKS, 1) void idft_ola_dft_rows_kernel(
    const float* __restrict__ in, const float* __restrict__ winv, int blocks, int rows_pad, int bins, int half_in, int nchunks, RowsOut y,
    int pad_left, int right, const float* __restrict__ mask, long mask_batch_stride, int mask_c_pitch,
    const float* __restrict__ wz_below, int bins_below, int half_below, float* __restrict__ zf_below) {
  constexpr int NST = 2 * HP / CH;
  static_assert(2 * HP % CH == 0 && HP <= HB / 2, "pairs per term must fill whole stages");
  __shared__ __attribute__((aligned(16))) float wl[KP * KP];
  __shared__ __attribute__((aligned(16))) float wf[32 * KP * 2];
  __shared__ float spill[FUSE_BLOCKS][2][FUSE_HALO][32];                      // [block][into the block before | after][frame][channel]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / nchunks, chunk = blockIdx.x - b * nchunks;
  const int j = wave, row = b * blocks + j, c = chunk * 32 + l31;
  const bool active = j < blocks;
  const long plane = (long)rows_pad * 2 * half_in;
  const bool cok = c < half_in;
  const float* src = in + (long)(active ? row : 0) * 2 * half_in + min(c, half_in - 1);
  auto load = [&](float (&bf)[CH], int st) {
#pragma unroll
    for (int s = 0; s < CH; ++s) {
      const int p = st * CH + s;
      const int bin = (p < HP ? 2 * p : 2 * (p - HP)) + h;
      bf[s] = src[(long)min(bin, bins - 1) * plane + (p < HP ? 0 : half_in)];
    }
  };
  float bfr[NST][CH];
  float mk[2][16];
  if (active) {
#pragma unroll
    for (int st = 0; st < NST; ++st) load(bfr[st], st);
    if (mask) {
      const float* mp = mask + (long)b * mask_batch_stride + min(c, mask_c_pitch - 1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          mk[i][r] = mp[(long)min(j * V + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, y.frames - 1) * mask_c_pitch];
    }
  }
  if (tid < 256) stage_matrix<KP, 1>(wl, winv);
  if (wz_below) {
    constexpr int N4 = 32 * KP * 2 / 4;
    for (int idx = tid; idx < N4; idx += 64 * FUSE_BLOCKS) reinterpret_cast<f32x4*>(wf)[idx] = reinterpret_cast<const f32x4*>(wz_below)[idx];
  }
  __syncthreads();
  f32x16 acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  if (active) {
    const float* afrag = wl + 2 * l31 + h;
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < CH; ++s) {
        const int p = st * CH + s;
        const int bin = (p < HP ? 2 * p : 2 * (p - HP)) + h;
        const unsigned keep = (cok && bin < bins) ? 0xffffffffu : 0u;
        const float v = __uint_as_float(__float_as_uint(bfr[st][s]) & keep);
        const int sidx = p < HP ? p : HB / 2 + (p - HP);
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[(sidx * KP + i * 32) * 2], v, acc[i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // rows 64 .. 67 of the window (h = 0 lanes, registers 0 .. 3 of the third tile) belong to the block before, 68 .. 71 (h = 1) to the one after
#pragma unroll
    for (int e = 0; e < FUSE_HALO; ++e) spill[j][h][e][l31] = acc[2][e];
  }
  __syncthreads();
  if (!active) return;
  // overlap-add: frames 0 .. right - 1 get what the block before spilled forward, frames V - pl .. V - 1 what the block after spilled back
#pragma unroll
  for (int e = 0; e < FUSE_HALO; ++e) {
    if (h == 0) { if (j > 0 && e < right) acc[0][e] += spill[j - 1][1][e][l31]; }
    else { if (j + 1 < blocks && e >= FUSE_HALO - pad_left) acc[1][12 + e] += spill[j + 1][0][e - (FUSE_HALO - pad_left)][l31]; }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = j * V + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      float val = c < y.channels ? acc[i][r] : 0.f;
      if (mask) val = mk[i][r] > 0.f ? val : 0.f;
      val = t < y.frames ? val : 0.f;
      asm volatile("" : "+v"(val));
      acc[i][r] = val;
    }
  if (c < y.c_pitch) {
    float* yp = y.base + (long)b * y.batch_stride + c;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = j * V + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (t < y.frames) yp[(long)t * y.c_pitch] = acc[i][r];
      }
  }
  if (!wz_below) return;
  f32x16 sacc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[i][r] = 0.f;
  const float* af2 = wf + 2 * l31 + h;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const float v = acc[q >> 4][q & 15];
#pragma unroll
    for (int i = 0; i < 3; ++i) sacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af2[(q * KP + i * 32) * 2], v, sacc[i], 0, 0, 0);
  }
  if (c < half_below) {
    const long zplane = (long)rows_pad * 2 * half_below;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int bin = m < HB ? m : m - HB, col = (m < HB ? 0 : half_below) + c;
        if (bin < bins_below) zf_below[(long)bin * zplane + (long)row * 2 * half_below + col] = sacc[i][r];
      }
  }
}

// ---- filters -> their spectra as a GEMM operand -------------------------------------------------------------
// G[k][c][o] = sum_w F[w][c][o] e^{-2 pi i k w / N}.
//  gfwd [bins][2 cph][2 npo]:  [[Gr, -Gi], [Gi, Gr]]      (Y = S conj(G); cph = spectra half width).  Back-prop to the input
//  (X = Z G) needs [[Gr^T, Gi^T], [-Gi^T, Gr^T]] = gfwd^T: the per-bin product kernels read gfwd as a TRANSPOSED operand
//  (round 4) -- rounds 2-3 built that second set of spectra from a flipped / transposed copy of the weights every step
//  (the 32-tap layer: 403 MB written and read again, plus the 128 MB of the flip).
//  from packed [w * cpi + c][npo]; one thread per (c, o), o fastest (coalesced reads and writes); rows c >= cin zero.
// (WT = compile-time width: the taps stay in registers; WT = 0: run-time width, taps in scratch)
// OUTP 0: fp32 `gfwd`; 4: fp32 `gfwd` as [bins][3][cph][npo] (the three-product form, see g3_form);
// 1 / 3: the matrix as one bf16 plane / the exact 3-way bf16 split in three planes at `gb`
template <int WT, int OUTP = 0>
__global__ __launch_bounds__(256) void filters_dft_fwd_kernel(const float* __restrict__ packed, int width_rt, int cin, int cout,
                                                              int cpi, int cph, int npo, int n, int bins,
                                                              const f32x2* __restrict__ tw, float* __restrict__ gfwd,
                                                              unsigned short* __restrict__ gb, size_t g_plane) {
  const int width = WT ? WT : width_rt;
  const int o = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
  if (o >= npo) return;
  float f[WT ? WT : 64];
  const bool live = c < cin && o < cout;
#pragma unroll
  for (int w = 0; w < width; ++w) f[w] = live ? packed[((long)w * cpi + c) * npo + o] : 0.f;
  const long plane = (long)2 * cph * 2 * npo;
  // blockIdx.z: a share of the bins (narrow layers have too few (c, o) pairs to fill the chip with one thread each)
  const int per = (bins + gridDim.z - 1) / gridDim.z, k_lo = blockIdx.z * per, k_hi = min(bins, k_lo + per);
  for (int k = k_lo; k < k_hi; ++k) {
    float gr = 0.f, gi = 0.f;
    const f32x2* row = tw + k * width;                     // uniform: wide scalar loads
#pragma unroll
    for (int w = 0; w < width; ++w) {
      const f32x2 t = row[w];
      gr = fmaf(f[w], t[0], gr);
      gi = fmaf(-f[w], t[1], gi);
    }
    auto put = [&](long o_, float v) {
      if (OUTP == 0) gfwd[o_] = v;
      else if (OUTP == 1) gb[o_] = __builtin_bit_cast(unsigned short, (__bf16)v);
      else {
        unsigned short sh, sm, sl;
        split3_bits(v, sh, sm, sl);
        gb[o_] = sh; gb[g_plane + o_] = sm; gb[2 * g_plane + o_] = sl;
      }
    };
    if constexpr (OUTP == 4) {
      // the three-product form: planes [c][o] of G_r, G_i - G_r and -(G_r + G_i) -- the forward product (S conj(G)) and back-prop
      // to the input (Z G, the planes read transposed) take their second and third factor from the same two, signs included
      const long g3 = (long)k * 3 * cph * npo + (long)c * npo + o;
      gfwd[g3] = gr;
      gfwd[g3 + (long)cph * npo] = gi - gr;
      gfwd[g3 + 2L * cph * npo] = -(gr + gi);
      continue;
    }
    const long g0 = (long)k * plane;
    put(g0 + (long)c * 2 * npo + o, gr);
    put(g0 + (long)c * 2 * npo + npo + o, -gi);
    put(g0 + (long)(cph + c) * 2 * npo + o, gi);
    put(g0 + (long)(cph + c) * 2 * npo + npo + o, gr);
  }
}

// ---- filter gradient: spectra of the lag products back to the W taps ---------------------------------------------
// q [bins][2 cph][2 npo] = [S_r | S_i]^T [Z_r | Z_i]:  Re = P00 + P11, Im = P10 - P01;
// dF[w][c][o] = (1 / N) sum_k w_k (Re cos(2 pi k w / N) - Im sin(2 pi k w / N)) into dpacked [w * cpi + c][npo].
// split != 0: q is [bins][2][cph][npo] -- the real and the imaginary lag products themselves (half the floats)
template <int WT>
__global__ __launch_bounds__(256) void filters_idft_kernel(const float* __restrict__ q, int width_rt, int cin, int cout, int cpi, int cph,
                                                           int npo, int n, int bins, const f32x2* __restrict__ tw,
                                                           float* __restrict__ dpacked, int split) {
  const int width = WT ? WT : width_rt;
  const int o = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
  if (o >= npo) return;
  float acc[WT ? WT : 64];
#pragma unroll
  for (int w = 0; w < width; ++w) acc[w] = 0.f;
  const bool live = c < cin && o < cout;
  const long plane = split ? (long)2 * cph * npo : (long)2 * cph * 2 * npo;
  const float inv_n = 1.f / (float)n;
  if (live) {
    for (int k = 0; k < bins; ++k) {
      const float* p = q + (long)k * plane;
      const float wk = (k == 0 || 2 * k == n) ? inv_n : 2.f * inv_n;      // DC (and the Nyquist bin of an even N) count once
      float re, im;
      if (split) {
        re = p[(long)c * npo + o] * wk;
        im = p[(long)cph * npo + (long)c * npo + o] * wk;
      } else {
        re = (p[(long)c * 2 * npo + o] + p[(long)(cph + c) * 2 * npo + npo + o]) * wk;
        im = (p[(long)(cph + c) * 2 * npo + o] - p[(long)c * 2 * npo + npo + o]) * wk;
      }
      const f32x2* row = tw + k * width;                   // uniform: wide scalar loads
#pragma unroll
      for (int w = 0; w < width; ++w) {
        const f32x2 t = row[w];
        acc[w] = fmaf(re, t[0], acc[w]);
        acc[w] = fmaf(-im, t[1], acc[w]);
      }
    }
  }
#pragma unroll
  for (int w = 0; w < width; ++w) dpacked[((long)w * cpi + c) * npo + o] = acc[w];
}

// ---- the same inverse transform on the matrix pipe (round 6) -------------------------------------------------------
// dF[w][(c, o)] = sum over (bin, part) of T[w][(bin, part)] * q[(bin, part)][(c, o)] is a product with a CONSTANT 32 x K matrix
// (K = 2 or 4 values per bin, <= 192): as one thread per (c, o) it cost 2 * W fmas per value on the vector ALU -- the 32-tap layer
// 3.2 GFLOP there, 78 us for 201 MB on the compute stream's critical path -- and with few (c, o) pairs it was a chain of dependent
// loads on a fraction of the chip (the 25-tap first layer: 160 workgroups, 45 bins, 60 us at the very end of the backward pass).
// Here a wavefront owns 32 * VW consecutive output channels o of one input channel c: T sits in LDS pair-interleaved (an A
// fragment is one ds_read_b32, as in the transforms above), a lane loads VW consecutive floats of a part's row per k-step (the
// half-waves read the two parts of a k-step: 128 * VW-byte runs), VW accumulator tiles, loads one stage ahead of the MFMAs.
// Rows w >= W of T are zero; the taps come out as accumulator rows and leave as VW-float stores.
constexpr int FI_CH = 12;                    // k-steps per stage
constexpr int FI_SMAX = 96;                  // k-steps: 4 parts x 48 bins / 2
struct IdftParts { long off[4]; };           // float offsets of a bin's parts inside its plane (c = 0, o = 0)
template <int VW>
__global__ __launch_bounds__(256, 2) void filters_idft_mfma_kernel(const float* __restrict__ q, int width, int cin, int cout, int cpi,
                                                                   int npo, int n, int bins, const f32x2* __restrict__ tw,
                                                                   float* __restrict__ dpacked, int pshift, long plane, IdftParts parts,
                                                                   int ld, int ochunks, int items) {
  typedef float vec __attribute__((ext_vector_type(VW)));
  __shared__ float tl[FI_SMAX * 32 * 2];                                      // [k-step][tap w][h]
  const int nparts = 1 << pshift, S = (nparts * bins) >> 1, stages = (S + FI_CH - 1) / FI_CH;
  {
    // split form: parts {Re, Im} -> {+cos, -sin};  block form: {P00, P11, P10, P01} (Re = P00 + P11, Im = P10 - P01) -> {+cos, +cos, -sin, +sin}
    const float inv_n = 1.f / (float)n;
    for (int i = threadIdx.x; i < stages * FI_CH * 64; i += 256) {
      const int hh = i & 1, m = (i >> 1) & 31, sidx = i >> 6;
      float v = 0.f;
      if (sidx < S && m < width) {
        const int kk = 2 * sidx + hh, k = kk >> pshift, part = kk & (nparts - 1);
        const float wk = (k == 0 || 2 * k == n) ? inv_n : 2.f * inv_n;         // DC (and the Nyquist bin of an even N) count once
        const f32x2 t = tw[k * width + m];
        if (nparts == 2) v = part == 0 ? wk * t[0] : -wk * t[1];
        else v = part < 2 ? wk * t[0] : (part == 2 ? -wk * t[1] : wk * t[1]);
      }
      tl[i] = v;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), total = gridDim.x * 4;
  const float* afrag = tl + 2 * l31 + h;
  for (int item = gw; item < items; item += total) {
    const int c = item / ochunks, o0 = (item - c * ochunks) * (32 * VW) + l31 * VW;
    f32x16 acc[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[e][r] = 0.f;
    if (c < cin) {                                                            // (wave-uniform; pad channels get zeros)
      const float* base = q + (long)c * ld + o0;
      auto load = [&](vec (&bf)[FI_CH], int st) {
#pragma unroll
        for (int j = 0; j < FI_CH; ++j) {
          const int kk = 2 * min(st * FI_CH + j, S - 1) + h;                   // (k-steps past the end re-read the last one against zero rows of T)
          bf[j] = *reinterpret_cast<const vec*>(base + (long)(kk >> pshift) * plane + parts.off[kk & (nparts - 1)]);
        }
      };
      auto mac = [&](const vec (&bf)[FI_CH], int st) {
        const float* a = afrag + st * (FI_CH * 64);
#pragma unroll
        for (int j = 0; j < FI_CH; ++j)
#pragma unroll
          for (int e = 0; e < VW; ++e) {
            acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j * 64], bf[j][e], acc[e], 0, 0, 0);
          }
      };
      vec bfr[2][FI_CH];
      load(bfr[0], 0);
      for (int st = 0; st < stages; st += 2) {                                // (two stages per trip: the buffers keep compile-time names)
        if (st + 1 < stages) load(bfr[1], st + 1);
        __builtin_amdgcn_sched_barrier(0);
        mac(bfr[0], st);
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < stages) {
          if (st + 2 < stages) load(bfr[0], st + 2);
          __builtin_amdgcn_sched_barrier(0);
          mac(bfr[1], st + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (c < cpi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int w = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (w < width) {
          vec v;
#pragma unroll
          for (int e = 0; e < VW; ++e) {
            v[e] = o0 + e < cout ? acc[e][r] : 0.f;
          }
          *reinterpret_cast<vec*>(dpacked + ((long)w * cpi + c) * npo + o0) = v;
        }
      }
    }
  }
}

// ---- bias gradient from the spectra of dz: bin 0 of a block is the plain sum of its 64 frames, so
// dbias[o] = sum_rows Z[0][row][o] -- 256 x n floats instead of a pass over the whole gradient tensor.
// 32 columns x 8 row-lanes per block, fixed summation order.
__global__ __launch_bounds__(256) void bias_from_spectra_kernel(const float* __restrict__ zf, int rows, int ld, int channels, int np,
                                                                float* __restrict__ dbias) {
  __shared__ float red[8][33];
  const int cl = threadIdx.x & 31, r = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s = 0.f;
  if (c < channels) {
    // four independent partial sums: the loads of a pass are in flight together (one dependent chain would expose
    // a cache latency per row)
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    int k = r;
    for (; k + 24 < rows; k += 32) {
      const float a = zf[(long)k * ld + c], b = zf[(long)(k + 8) * ld + c], d = zf[(long)(k + 16) * ld + c],
                  e = zf[(long)(k + 24) * ld + c];
      p0 += a; p1 += b; p2 += d; p3 += e;
    }
    for (; k < rows; k += 8) p0 += zf[(long)k * ld + c];
    s = (p0 + p1) + (p2 + p3);
  }
  red[r][cl] = s;
  __syncthreads();
  if (r == 0 && c < np) {
    float t = red[0][cl];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += red[k][cl];
    dbias[c] = t;
  }
}

bool tensor_ok(const st_tensor3* t) {
  return t && t->base && t->batch > 0 && t->frames > 0 && t->channels > 0 && t->halo >= 0 && t->c_pitch % 16 == 0 &&
         t->c_pitch >= t->channels && t->t_pitch >= t->halo + t->frames;
}

RowsIn rows_in(const st_tensor3& t) {
  RowsIn r;
  r.base_b = nullptr;
  r.base = t.base + (long)t.halo * t.c_pitch;        // frame 0 of utterance 0
  r.batch_stride = (long)t.t_pitch * t.c_pitch;
  r.c_pitch = t.c_pitch;
  r.channels_read = t.c_pitch;
  r.t_lo = -t.halo;
  r.t_hi = t.t_pitch - t.halo;
  return r;
}

constexpr int TRANSFORM_WGS_DEFAULT = 512;       // persistent: two workgroups per CU, every wave walks its share of the items
inline int transform_wgs() { const int t = st::tuning(st::TUNE_TRANSFORM_WGS); return t > 0 ? t : TRANSFORM_WGS_DEFAULT; }

// tb: the tensor's bf16 form (null: read the fp32 tensor); planes: 0 -> fp32 spectra `out`, 1 / 3 -> bf16 plane(s) `outb`
// form: 0, or 4 / 5 -- the fp32 spectra in the three-product layouts (rows of 4 / 3 parts, dft_rows_kernel OUTP)
void launch_dft(const st_tensor3& t, const void* tb, const Plan& pl, const float* wm, int start, int frames_used, int half, float* out,
                void* outb, int planes, size_t out_plane, float* dc, hipStream_t s, long bin_stride = 0, float* out2 = nullptr,
                int form = 0) {
  if (bin_stride == 0) bin_stride = (long)pl.rows_pad * (form == 4 ? 4 : form == 5 ? 3 : 2) * half;
  const int nchunks = st::ceil_div(half, 32);
  const int wgs = std::min(transform_wgs(), st::ceil_div(pl.rows_pad * nchunks, 4));
  const int nst = frames_used <= 6 * CH ? 3 : 4;                           // the matrix has no columns past frames_used
  // mb = the bytes the transform has to move: every frame of the tensor once, every spectrum value (and its rotated copy) once
  const double esz_in = tb ? 2.0 : 4.0, esz_out = planes == 0 ? 4.0 : 2.0 * planes;
  st::trace("dft_rows<%d%s%s> rows=%d chunks=%d bins=%d gflop=%.3f mb=%.2f", nst, tb ? ",bf16-in" : "",
            planes == 1 ? ",bf16-out" : planes == 3 ? ",x3-out" : form == 4 ? ",4-part" : form == 5 ? ",3-part" : "",
            pl.rows, nchunks, pl.bins, 4096e-9 * pl.rows * (double)nchunks * nst * CH * 3,
            1e-6 * ((double)t.batch * t.frames * t.c_pitch * esz_in +
                    (double)pl.bins * pl.rows * half * esz_out * (form == 4 ? 4 : form == 5 ? 3 : out2 ? 4 : 2)));
  RowsIn x = rows_in(t);
  if (tb) x.base_b = reinterpret_cast<const unsigned short*>(tb) + (long)t.halo * t.c_pitch;
  unsigned short* ob = reinterpret_cast<unsigned short*>(outb);
  st::LaunchTimer timer(s);
#define ST_DFT(NSTV, INB, OUTPV)                                                                                         \
  st::launch_timed(timer, dft_rows_kernel<NSTV, INB, OUTPV>, dim3(wgs), dim3(256), s, x, wm, pl.blocks, pl.rows, pl.rows_pad, start, \
                   pl.bins, half, nchunks, out, ob, out_plane, dc, bin_stride, out2)
#define ST_DFT_N(INB, OUTPV) do { if (nst == 3) ST_DFT(3, INB, OUTPV); else ST_DFT(4, INB, OUTPV); } while (0)
  if (!tb && planes == 0 && form == 4) ST_DFT_N(false, 4);
  else if (!tb && planes == 0 && form == 5) ST_DFT_N(false, 5);
  else if (!tb && planes == 0) ST_DFT_N(false, 0);
  else if (!tb && planes == 3) ST_DFT_N(false, 3);
  else if (tb && planes == 1) ST_DFT_N(true, 1);
  else st::set_error("dft: unsupported operand form (bf16 in: %d, planes %d)", tb ? 1 : 0, planes);
#undef ST_DFT_N
#undef ST_DFT
}

template <int TERMS>
void launch_idft(const float* in, const float* winv, const Plan& p, int half_in, int nchunks, const RowsOut& out, const float* bias,
                 int relu, const void* mask, long mask_batch_stride, int mask_c_pitch, hipStream_t s) {
  const dim3 grid(std::min(transform_wgs(), st::ceil_div(p.rows * nchunks, 4)));
  const int hp = p.bins <= 36 ? 18 : 24;
  const bool bf = out.base_b != nullptr;                       // bf16 activations: bf16 output and bf16 mask source
  st::trace("idft_rows<%d,%d%s> rows=%d chunks=%d bins=%d gflop=%.3f mb=%.2f", TERMS, hp, bf ? ",bf16" : "", p.rows, nchunks, p.bins,
            4096e-9 * p.rows * (double)nchunks * 2 * hp * (TERMS + 1),
            1e-6 * ((double)p.bins * p.rows * 2 * half_in * 4 + (double)p.rows * V * out.c_pitch * (bf ? 2.0 : 4.0) * (mask ? 2 : 1)));
  const float* mk = reinterpret_cast<const float*>(mask);
  st::LaunchTimer timer(s);
#define ST_IDFT(HPV, BFV)                                                                                                     \
  st::launch_timed(timer, idft_rows_kernel<TERMS, HPV, BFV>, grid, dim3(256), s, in, winv, p.blocks, p.rows, p.rows_pad, p.bins, half_in, \
                   nchunks, out, bias, relu, mk, mask_batch_stride, mask_c_pitch)
  if (p.bins <= 36) { if (bf) ST_IDFT(18, true); else ST_IDFT(18, false); }
  else { if (bf) ST_IDFT(24, true); else ST_IDFT(24, false); }
#undef ST_IDFT
}

bool width_ok(int width) { return width >= 2 && V + width - 1 <= KP; }
// The three-product form of a layer's per-bin complex products (Gauss; conv_gemm.hip gemm_nn_g3_kernel / gemm_tn_g3_kernel): wide
// layers only -- many output channels (the narrow layers' products run on the persistent per-bin kernel, and their launches are
// too short for another split of the work) and more than 9 taps (the fused transforms of a chain write a neighbour layer's
// spectra in the four-product layout: they stop at 9 taps, so the two never meet on the input side).  The GRADIENT spectra's
// layout [Z_r + Z_i | Z_r | Z_i] is a function of (taps, output channels) alone -- st_conv1d_fft_dz_spectra_f32 does not know
// the layer's input -- and a layer whose input spectra do not tile the kernels (half % 128 != 0) reads its [Z_r | Z_i] out of
// the same rows with the four-product kernels.  st_set_tuning("no_g3", 1): the four-product form everywhere (A/B runs, parity
// of the two forms) -- set before the filter spectra are built.
constexpr int G3_MIN_WIDTH = 2 * FUSE_HALO + 2;
bool zf3_form(int width, int cout) { return width >= G3_MIN_WIDTH && npad_of(cout) >= 512 && st::tuning(st::TUNE_NO_G3) == 0; }
bool g3_form(int width, int cin_pitch, int cout) { return zf3_form(width, cout) && half_of(cin_pitch) % 128 == 0; }
// the filter gradient's lag products as separate real / imaginary products over half-length rows (see bwd_filter): needs the
// spectra halves to tile the filter-gradient kernel (128 columns)
bool split_lag_products(int half) { return half % 128 == 0; }

template <int OUTP>
void launch_filters_planes(int width, const dim3& grid, hipStream_t s, const float* packed, int cin, int cout, int cin_pitch, int cph,
                           int npo, int n, int bins, const f32x2* tw, unsigned short* gb, size_t g_plane) {
  float* none = nullptr;
  if (width == 32) hipLaunchKernelGGL((filters_dft_fwd_kernel<32, OUTP>), grid, dim3(256), 0, s, packed, width, cin, cout, cin_pitch, cph, npo, n, bins, tw, none, gb, g_plane);
  else if (width == 25) hipLaunchKernelGGL((filters_dft_fwd_kernel<25, OUTP>), grid, dim3(256), 0, s, packed, width, cin, cout, cin_pitch, cph, npo, n, bins, tw, none, gb, g_plane);
  else if (width == 7) hipLaunchKernelGGL((filters_dft_fwd_kernel<7, OUTP>), grid, dim3(256), 0, s, packed, width, cin, cout, cin_pitch, cph, npo, n, bins, tw, none, gb, g_plane);
  else hipLaunchKernelGGL((filters_dft_fwd_kernel<0, OUTP>), grid, dim3(256), 0, s, packed, width, cin, cout, cin_pitch, cph, npo, n, bins, tw, none, gb, g_plane);
}

bool planes_ok(int planes) { return planes == 1 || planes == 3; }

// can layer i's inverse transform carry the next layer's forward transform (idft_dft_rows_kernel)?
bool can_fuse_next(const Plan& p, const st_tensor3& y, int next_width, int next_pad_left) {
  if (next_width < 2 || !width_ok(next_width) || next_pad_left < 0 || next_pad_left >= next_width) return false;
  const int right = next_width - 1 - next_pad_left;
  return p.blocks <= FUSE_BLOCKS && p.rows == p.rows_pad && next_pad_left <= FUSE_HALO && right <= FUSE_HALO &&
         half_of(y.c_pitch) == y.c_pitch && y.c_pitch % 32 == 0 && st::tuning(st::TUNE_NO_FUSED_TRANSFORMS) == 0;
}


// the filter gradient's inverse transform: the matrix-pipe kernel (widths up to 32, output channels packing to whole 32 * VW
// chunks), else -- or with st_set_tuning("filters_idft_valu", 1), for A/B runs and the parity test of the two -- one thread per (c, o)
void launch_filters_idft(const float* qf, int width, int cin, int cout, int cpi, int cph, int npo, int n, int bins, const f32x2* tw,
                         float* dpacked, int split, hipStream_t s) {
  if (width <= 32 && npo % 128 == 0 && (split ? 2 : 4) * bins <= 2 * FI_SMAX && st::tuning(st::TUNE_FILTERS_IDFT_VALU) == 0) {
    IdftParts parts;
    long plane;
    int ld;
    if (split) {                       // q [bins][2][cph][npo]
      plane = 2L * cph * npo; ld = npo;
      parts.off[0] = 0; parts.off[1] = (long)cph * npo; parts.off[2] = parts.off[3] = 0;
    } else {                           // q [bins][2 cph][2 npo]: P00 (c, o), P11 (cph + c, npo + o), P10 (cph + c, o), P01 (c, npo + o)
      plane = 2L * cph * 2 * npo; ld = 2 * npo;
      parts.off[0] = 0; parts.off[1] = (long)cph * 2 * npo + npo; parts.off[2] = (long)cph * 2 * npo; parts.off[3] = npo;
    }
    // the widest loads that still leave the chip a few thousand wavefront items
    const int vw = (long)cpi * (npo / 128) >= 2048 ? 4 : ((long)cpi * (npo / 64) >= 2048 ? 2 : 1);
    const int ochunks = npo / (32 * vw), items = cpi * ochunks;
    const dim3 grid(std::min(1024, st::ceil_div(items, 4)));
    st::trace("filters_idft_mfma<%d> taps=%d bins=%d parts=%d items=%d gflop=%.3f mb=%.2f", vw, width, bins, split ? 2 : 4, items,
              4096e-9 * items * (double)vw * st::ceil_div((split ? 2 : 4) * bins / 2, FI_CH) * FI_CH,
              1e-6 * 4 * ((double)(split ? 2 : 4) * bins * cin * npo + (double)width * cpi * npo));
    st::LaunchTimer timer(s);
#define ST_FI(VWV) st::launch_timed(timer, filters_idft_mfma_kernel<VWV>, grid, dim3(256), s, qf, width, cin, cout, cpi, npo, n, bins, tw, dpacked, \
                                    split ? 1 : 2, plane, parts, ld, ochunks, items)
    if (vw == 4) ST_FI(4); else if (vw == 2) ST_FI(2); else ST_FI(1);
#undef ST_FI
    return;
  }
  const dim3 grid(st::ceil_div(npo, 256), cpi);
  if (width == 32) hipLaunchKernelGGL(filters_idft_kernel<32>, grid, dim3(256), 0, s, qf, width, cin, cout, cpi, cph, npo, n, bins, tw, dpacked, split);
  else if (width == 25) hipLaunchKernelGGL(filters_idft_kernel<25>, grid, dim3(256), 0, s, qf, width, cin, cout, cpi, cph, npo, n, bins, tw, dpacked, split);
  else if (width == 7) hipLaunchKernelGGL(filters_idft_kernel<7>, grid, dim3(256), 0, s, qf, width, cin, cout, cpi, cph, npo, n, bins, tw, dpacked, split);
  else hipLaunchKernelGGL(filters_idft_kernel<0>, grid, dim3(256), 0, s, qf, width, cin, cout, cpi, cph, npo, n, bins, tw, dpacked, split);
}

}  // namespace

extern "C" {

int st_gemm_nn_batched_f32(const float* a, int64_t lda, int64_t a_batch, const float* b, int64_t b_batch, float* c,
                           int64_t ldc, int64_t c_batch, int m, int k, int n, int batches, void* stream) {
  return st::gemm_nn_batched(a, lda, a_batch, b, b_batch, c, ldc, c_batch, m, k, n, batches, st::as_stream(stream));
}

size_t st_gemm_nn_batched_ws_bytes(void) { return (size_t)st::SK_WS_FLOATS * sizeof(float); }
size_t st_gemm_nn_batched_ctrl_bytes(void) { return (size_t)st::SK_CTRL_WORDS * sizeof(unsigned); }

int st_gemm_nn_batched_ws_f32(const float* a, int64_t lda, int64_t a_batch, const float* b, int64_t b_batch, float* c, int64_t ldc,
                              int64_t c_batch, int m, int k, int n, int batches, void* workspace, size_t workspace_bytes,
                              void* stream) {
  float* sk = (workspace && workspace_bytes >= st_gemm_nn_batched_ws_bytes()) ? reinterpret_cast<float*>(workspace) : nullptr;
  return st::gemm_nn_batched(a, lda, a_batch, b, b_batch, c, ldc, c_batch, m, k, n, batches, st::as_stream(stream), sk);
}

int st_gemm_nn_batched_bt_ws_f32(const float* a, int64_t lda, int64_t a_batch, const float* bt, int64_t bt_batch, float* c, int64_t ldc,
                                 int64_t c_batch, int m, int k, int n, int batches, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  float* sk = (workspace && workspace_bytes >= st_gemm_nn_batched_ws_bytes()) ? reinterpret_cast<float*>(workspace) : nullptr;
  return st::gemm_nn_batched(a, lda, a_batch, bt, bt_batch, c, ldc, c_batch, m, k, n, batches, st::as_stream(stream), sk, true);
}

int st_gemm_tn_batched_f32(const float* a, int64_t lda, int64_t a_batch, const float* z, int64_t ldz, int64_t z_batch, float* out,
                           int64_t out_batch, int m, int k, int n, int batches, void* stream) {
  return st::gemm_tn_batched(a, lda, a_batch, z, ldz, z_batch, out, out_batch, m, k, n, batches, st::as_stream(stream));
}

int st_gemm_tn_batched_shared_f32(const float* a, int64_t lda, int64_t a_batch, const float* z, int64_t ldz, int64_t z_batch, float* out,
                                  int64_t out_batch, int m, int k, int n, int batches, int z_batch_shift, void* stream) {
  ST_REQUIRE(z_batch_shift >= 0 && z_batch_shift < 8, "gemm_tn_batched_shared: bad shift");
  return st::gemm_tn_batched(a, lda, a_batch, z, ldz, z_batch, out, out_batch, m, k, n, batches, st::as_stream(stream), z_batch_shift);
}

int st_gemm_nn_g3_batched_f32(const float* a, int64_t lda, int64_t a_batch, const int64_t* a_off, const float* b, int64_t ldb, int64_t b_batch,
                              const int64_t* b_off, int b_transposed, float* c, int64_t ldc, int64_t c_batch, int64_t c_off2, int m, int k, int n,
                              int batches, void* stream) {
  ST_REQUIRE(a_off && b_off, "gemm_nn_g3_batched: null offsets");
  const long ao[3] = {(long)a_off[0], (long)a_off[1], (long)a_off[2]}, bo[3] = {(long)b_off[0], (long)b_off[1], (long)b_off[2]};
  return st::gemm_nn_g3_batched(a, lda, a_batch, ao, b, ldb, b_batch, bo, c, ldc, c_batch, c_off2, m, k, n, batches, st::as_stream(stream),
                                b_transposed != 0);
}

int st_gemm_tn_g3_batched_f32(const float* a, int64_t lda, int64_t a_batch, const int64_t* a_off, const float* z, int64_t ldz, int64_t z_batch,
                              const int64_t* z_off, float* out, int64_t out_batch, int64_t out_part, int m, int k, int n, int batches,
                              void* stream) {
  ST_REQUIRE(a_off && z_off, "gemm_tn_g3_batched: null offsets");
  const long ao[3] = {(long)a_off[0], (long)a_off[1], (long)a_off[2]}, zo[3] = {(long)z_off[0], (long)z_off[1], (long)z_off[2]};
  return st::gemm_tn_g3_batched(a, lda, a_batch, ao, z, ldz, z_batch, zo, out, out_batch, out_part, m, k, n, batches, st::as_stream(stream));
}

// 0: the layer's per-bin products run in the four-product form; 1: its gradient spectra take three-part rows but its input does
// not tile the three-product kernels; 2: the three-product form (g3_form)
int st_conv1d_fft_three_products(int width, int cin_pitch, int cout) {
  return g3_form(width, cin_pitch, cout) ? 2 : (zf3_form(width, cout) ? 1 : 0);
}

int st_conv1d_fft_plan(int width, int frames, int batch, int* n, int* valid, int* blocks, int* bins, int* rows_pad) {
  ST_REQUIRE(width_ok(width) && frames > 0 && batch > 0, "fft plan: filter width must be in [2, 33]");
  const Plan p = make_plan(width, frames, batch);
  if (n) *n = p.n;
  if (valid) *valid = V;
  if (blocks) *blocks = p.blocks;
  if (bins) *bins = p.bins;
  if (rows_pad) *rows_pad = p.rows_pad;
  return ST_OK;
}

size_t st_conv1d_fft_table_floats(void) { return T_END; }

int st_conv1d_fft_tables_f32(int width, int pad_left, float* tables, size_t table_floats, void* stream) {
  ST_REQUIRE(width_ok(width) && pad_left >= 0 && pad_left < width && tables && table_floats >= (size_t)T_END,
             "fft tables: bad argument");
  hipLaunchKernelGGL(tables_kernel, dim3(64), dim3(256), 0, st::as_stream(stream), V + width - 1, pad_left, tables);
  return st::check_launch("fft tables");
}

size_t st_conv1d_fft_filter_floats(int width, int cin_pitch, int cout) {
  if (!width_ok(width)) return 0;
  const size_t bins = (V + width - 1) / 2 + 1;
  return bins * 2 * half_of(cin_pitch) * 2 * npad_of(cout);
}

int st_conv1d_fft_filters_f32(const float* packed, int width, int cin, int cout, int cin_pitch, const float* tables, float* gfwd,
                              void* stream) {
  ST_REQUIRE(width_ok(width) && cin_pitch % 16 == 0 && tables && packed && gfwd, "fft filters: bad argument");
  ST_REQUIRE(npad_of(cout) % 128 == 0, "fft filters: the output channels must pack to a multiple of 128");
  hipStream_t s = st::as_stream(stream);
  const int n = V + width - 1, bins = n / 2 + 1;
  const f32x2* tw = reinterpret_cast<const f32x2*>(tables + T_FW);
  const int npo = npad_of(cout);
  // compile-time widths for the layers of the model (taps in registers); any other width runs the generic form
#define ST_FFT_WIDTH_DISPATCH(KERNEL, ...)                                                                   \
  do {                                                                                                       \
    if (width == 32) hipLaunchKernelGGL(KERNEL<32>, grid, dim3(256), 0, s, __VA_ARGS__);                     \
    else if (width == 25) hipLaunchKernelGGL(KERNEL<25>, grid, dim3(256), 0, s, __VA_ARGS__);                \
    else if (width == 7) hipLaunchKernelGGL(KERNEL<7>, grid, dim3(256), 0, s, __VA_ARGS__);                  \
    else hipLaunchKernelGGL(KERNEL<0>, grid, dim3(256), 0, s, __VA_ARGS__);                                  \
  } while (0)
  // rows of pad channels (c in [cin, half)) are written as zeros by the kernel's `live` test
  const int gx = st::ceil_div(npo, 256), gy = half_of(cin_pitch);
  const dim3 grid(gx, gy, gx * gy < 1024 ? 4 : 1);
  if (g3_form(width, cin_pitch, cout)) {               // three planes per bin instead of the 2 x 2 block matrix (g3_form)
    if (width == 32) hipLaunchKernelGGL((filters_dft_fwd_kernel<32, 4>), grid, dim3(256), 0, s, packed, width, cin, cout, cin_pitch, half_of(cin_pitch), npo, n, bins, tw, gfwd, (unsigned short*)nullptr, (size_t)0);
    else hipLaunchKernelGGL((filters_dft_fwd_kernel<0, 4>), grid, dim3(256), 0, s, packed, width, cin, cout, cin_pitch, half_of(cin_pitch), npo, n, bins, tw, gfwd, (unsigned short*)nullptr, (size_t)0);
    return st::check_launch("fft filters (three planes)");
  }
  ST_FFT_WIDTH_DISPATCH(filters_dft_fwd_kernel, packed, width, cin, cout, cin_pitch, half_of(cin_pitch), npo, n, bins, tw, gfwd,
                        (unsigned short*)nullptr, (size_t)0);
  return st::check_launch("fft filters");
}

// floats of the input spectra sf and of the dz spectra zf
// (the fp32 form keeps, per bin, the spectra [S_r | S_i] followed by their rotated copy [S_i | -S_r]: read as matrices of
// half-length rows the two are the operands of the real and of the imaginary lag products of the filter gradient)
size_t st_conv1d_fft_sf_floats(const st_tensor3* x, const st_tensor3* y, int width) {
  if (!x || !y || !width_ok(width)) return 0;
  const Plan p = make_plan(width, y->frames, y->batch);
  return (size_t)p.bins * 2 * p.rows_pad * 2 * half_of(x->c_pitch);
}

size_t st_conv1d_fft_zf_floats(const st_tensor3* dz, int width) {
  if (!dz || !width_ok(width)) return 0;
  const Plan p = make_plan(width, dz->frames, dz->batch);
  // (three parts per row, [Z_r + Z_i | Z_r | Z_i], where the layer's gradient spectra take the three-product layout: zf3_form --
  // sized for it whatever the tuning knob says)
  const bool three = width >= G3_MIN_WIDTH && npad_of(dz->channels) >= 512;
  return (size_t)p.bins * p.rows_pad * (three ? 3 : 2) * npad_of(dz->channels);
}

size_t st_conv1d_fft_ws(const st_tensor3* x, const st_tensor3* y, int width) {
  if (!x || !y || !width_ok(width)) return 0;
  const Plan p = make_plan(width, y->frames, y->batch);
  const size_t nf = 2 * (size_t)npad_of(y->channels), ka = 2 * (size_t)half_of(x->c_pitch), nb = ka;
  const size_t yf = (size_t)p.bins * p.rows_pad * nf, xf = (size_t)p.bins * p.rows_pad * nb, qf = (size_t)p.bins * ka * nf;
  // [stream-K area of the per-bin products (control words + partial tiles, st_common.h) | spectra of the call's output]
  return (st::SK_WS_FLOATS + std::max(yf, std::max(xf, qf)) + 64) * sizeof(float);
}

// The forward call of a CHAIN of frequency-domain layers: `sf_ready` != 0 -- this layer's input spectra are in `sf` already (the
// previous layer's call wrote them); next_* given -- when the shapes allow (can_fuse_next: at most 8 blocks per utterance, no pad
// rows, a next window reaching at most 4 frames into either neighbour block) the inverse transform also produces the NEXT layer's
// input spectra into `next_sf` and *next_sf_written becomes 1 (else 0: the next call transforms y itself).
int st_conv1d_nwc_fwd_fft_chain_f32(const st_tensor3* x, const float* gfwd, const float* bias, int width, int pad_left, int relu,
                                    const st_tensor3* y, const float* tables, float* sf, int sf_ready, const float* next_tables,
                                    float* next_sf, int next_width, int next_pad_left, int* next_sf_written, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  ST_REQUIRE(tensor_ok(x) && tensor_ok(y) && gfwd && sf && workspace && tables && width_ok(width), "conv fft fwd: bad argument");
  ST_REQUIRE(x->batch == y->batch && x->frames == y->frames && pad_left >= 0 && pad_left < width, "conv fft fwd: stride-1 SAME layers only");
  ST_REQUIRE(npad_of(y->channels) % 128 == 0 && workspace_bytes >= st_conv1d_fft_ws(x, y, width), "conv fft fwd: workspace / shape");
  if (next_sf_written) *next_sf_written = 0;
  hipStream_t s = st::as_stream(stream);
  const Plan p = make_plan(width, y->frames, y->batch, ROWS_F32);
  const int ka = 2 * half_of(x->c_pitch), npo = npad_of(y->channels), nf = 2 * npo;
  float* const sk = reinterpret_cast<float*>(workspace);
  float* yf = sk + st::SK_WS_FLOATS;
  const int half = half_of(x->c_pitch);
  const long s_bin = 2L * p.rows_pad * ka;                      // [S | rotated copy] per bin
  if (g3_form(width, x->c_pitch, y->channels)) {
    // three real products per bin: rows [S_r + S_i | S_i | S_r | S_i - S_r] (the same floats per bin as [S | rotated copy]) against
    // the planes G_r, G_i - G_r, -(G_r + G_i):  Re Y = (S_r + S_i) G_r + S_i (G_i - G_r),  Im Y = (S_r + S_i) G_r - S_r (G_r + G_i)
    ST_REQUIRE(!sf_ready, "conv fft fwd: a layer in the three-product form transforms its own input");
    launch_dft(*x, nullptr, p, tables + T_FS, -pad_left, p.n, half, sf, nullptr, 0, 0, nullptr, s, 0, nullptr, 4);
    const long plane = (long)half * npo;
    const long a_off[3] = {0, half, 2L * half}, b_off[3] = {0, plane, 2 * plane};
    if (int e = st::gemm_nn_g3_batched(sf, 4L * half, s_bin, a_off, gfwd, npo, 3 * plane, b_off, yf, nf, (long)p.rows_pad * nf, npo,
                                       p.rows_pad, half, npo, p.bins, s))
      return e;
  } else {
  if (!sf_ready)
    launch_dft(*x, nullptr, p, tables + T_FS, -pad_left, p.n, half, sf, nullptr, 0, 0, nullptr, s, s_bin,
               split_lag_products(half) ? sf + (long)p.rows_pad * ka : nullptr);
  if (int e = st::gemm_nn_batched(sf, ka, s_bin, gfwd, (long)ka * nf, yf, nf, (long)p.rows_pad * nf, p.rows_pad, ka, nf, p.bins, s, sk))
    return e;
  }
  RowsOut out{y->base + (long)y->halo * y->c_pitch, (long)y->t_pitch * y->c_pitch, y->c_pitch, y->channels, y->frames, nullptr};
  const int nchunks = st::ceil_div(y->c_pitch, 32);
  if (next_tables && next_sf && can_fuse_next(p, *y, next_width, next_pad_left)) {
    const Plan pn = make_plan(next_width, y->frames, y->batch, ROWS_F32);
    const int half_n = half_of(y->c_pitch), ka_n = 2 * half_n;
    const long sn_bin = 2L * pn.rows_pad * ka_n;
    float* rot = split_lag_products(half_n) ? next_sf + (long)pn.rows_pad * ka_n : nullptr;
    // gflop: the inverse transform's 2 * HP k-steps on two 32-frame tiles, then 32 + FUSE_HALO k-steps on the next layer's three
    // 32-row tiles; mb: this layer's output spectra in, the activation tensor and the next layer's input spectra (+ rotated copy) out
    st::trace("idft_dft_rows<%d> rows=%d chunks=%d bins=%d next_bins=%d gflop=%.3f mb=%.2f", p.bins <= 36 ? 18 : 24, p.rows, nchunks, p.bins, pn.bins,
              4096e-9 * p.rows * (double)nchunks * (2 * (p.bins <= 36 ? 18 : 24) * 2 + (32 + FUSE_HALO) * 3),
              1e-6 * 4 * ((double)p.bins * p.rows * 2 * npo + (double)p.rows * V * y->c_pitch + (double)pn.bins * p.rows * 2 * half_n * (rot ? 2 : 1)));
    const dim3 grid(y->batch * nchunks), block(64 * FUSE_BLOCKS);
    st::LaunchTimer timer(s);
    if (p.bins <= 36)
      st::launch_timed(timer, idft_dft_rows_kernel<18>, grid, block, s, yf, tables + T_IY, p.blocks, p.rows_pad, p.bins, npo, nchunks, out, bias, relu,
                       next_tables + T_FP, pn.n, next_pad_left, pn.bins, half_n, next_sf, sn_bin, rot);
    else
      st::launch_timed(timer, idft_dft_rows_kernel<24>, grid, block, s, yf, tables + T_IY, p.blocks, p.rows_pad, p.bins, npo, nchunks, out, bias, relu,
                       next_tables + T_FP, pn.n, next_pad_left, pn.bins, half_n, next_sf, sn_bin, rot);
    if (next_sf_written) *next_sf_written = 1;
    return st::check_launch("conv fft fwd (fused transforms)");
  }
  launch_idft<1>(yf, tables + T_IY, p, npo, nchunks, out, bias, relu, nullptr, 0L, 0, s);
  return st::check_launch("conv fft fwd");
}

int st_conv1d_nwc_fwd_fft_f32(const st_tensor3* x, const float* gfwd, const float* bias, int width, int pad_left, int relu,
                              const st_tensor3* y, const float* tables, float* sf, void* workspace,
                              size_t workspace_bytes, void* stream) {
  return st_conv1d_nwc_fwd_fft_chain_f32(x, gfwd, bias, width, pad_left, relu, y, tables, sf, 0, nullptr, nullptr, 0, 0, nullptr, workspace,
                                         workspace_bytes, stream);
}

int st_conv1d_fft_dz_spectra_f32(const st_tensor3* dz, int width, const float* tables, float* zf, void* stream) {
  ST_REQUIRE(tensor_ok(dz) && tables && zf && width_ok(width) && npad_of(dz->channels) % 128 == 0, "conv fft dz spectra: bad argument");
  const Plan p = make_plan(width, dz->frames, dz->batch, ROWS_F32);
  launch_dft(*dz, nullptr, p, tables + T_FZ, 0, V, npad_of(dz->channels), zf, nullptr, 0, 0, nullptr, st::as_stream(stream), 0, nullptr,
             zf3_form(width, dz->channels) ? 5 : 0);
  return st::check_launch("conv fft dz spectra");
}

int st_conv1d_fft_bias_grad_f32(const st_tensor3* dz, int width, const float* zf, float* dbias, void* stream) {
  ST_REQUIRE(tensor_ok(dz) && zf && dbias && width_ok(width) && npad_of(dz->channels) % 128 == 0, "conv fft bias grad: bad argument");
  const Plan p = make_plan(width, dz->frames, dz->batch, ROWS_F32);
  const int np = npad_of(dz->channels);
  const bool zf3 = zf3_form(width, dz->channels);              // rows [Z_r + Z_i | Z_r | Z_i]: bin 0's real parts one part in
  hipLaunchKernelGGL(bias_from_spectra_kernel, dim3(st::ceil_div(np, 32)), dim3(256), 0, st::as_stream(stream), zf + (zf3 ? np : 0), p.rows,
                     (zf3 ? 3 : 2) * np, dz->channels, np, dbias);
  return st::check_launch("conv fft bias grad");
}

// Back-prop to the input in a CHAIN of frequency-domain layers: below_tables / below_zf given (the layer below, whose output
// gradient dx is) -- when the shapes allow (at most 8 blocks per utterance, no pad rows, spills of at most 4 frames into either
// neighbour block, dx channels packing like the layer below's output) ONE launch inverts every block at its whole window,
// overlap-adds through LDS, masks, stores dx and leaves the layer below's dz spectra in below_zf (*below_zf_written = 1: that
// layer skips st_conv1d_fft_dz_spectra_f32).  Without them, or when the shapes do not fit: the three-term inverse of rounds 2-3.
int st_conv1d_nwc_bwd_data_fft_chain_f32(const st_tensor3* dz, const float* zf, const float* gfwd, int width, int pad_left,
                                         const st_tensor3* act, const st_tensor3* dx, const float* tables, const float* below_tables,
                                         float* below_zf, int below_width, int* below_zf_written, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  ST_REQUIRE(tensor_ok(dz) && tensor_ok(dx) && zf && gfwd && workspace && tables && width_ok(width), "conv fft bwd_data: bad argument");
  ST_REQUIRE(dz->batch == dx->batch && dz->frames == dx->frames && pad_left >= 0 && pad_left < width, "conv fft bwd_data: stride-1 layers only");
  ST_REQUIRE(npad_of(dz->channels) % 128 == 0 && workspace_bytes >= st_conv1d_fft_ws(dx, dz, width), "conv fft bwd_data: workspace / shape");
  if (act) ST_REQUIRE(tensor_ok(act) && act->batch == dx->batch && act->frames == dx->frames && act->c_pitch >= dx->c_pitch,
                      "conv fft bwd_data: mask tensor mismatch");
  if (below_zf_written) *below_zf_written = 0;
  hipStream_t s = st::as_stream(stream);
  const Plan p = make_plan(width, dz->frames, dz->batch, ROWS_F32);
  // X[bin] = Z[bin] (rows x 2 npo) * gfwd[bin]^T (2 npo x 2 cph): the forward spectra read as a transposed operand
  const int kz = 2 * npad_of(dz->channels), cph = half_of(dx->c_pitch), nb = 2 * cph;
  float* const sk = reinterpret_cast<float*>(workspace);
  float* xf = sk + st::SK_WS_FLOATS;
  const int npz = npad_of(dz->channels);
  const bool zf3 = zf3_form(width, dz->channels);
  if (g3_form(width, dx->c_pitch, dz->channels)) {
    // X = Z G in three products, the filter planes read transposed in place:
    // Re X = (Z_r + Z_i) G_r - Z_i (G_r + G_i),  Im X = (Z_r + Z_i) G_r + Z_r (G_i - G_r)
    const long plane = (long)cph * npz;
    const long a_off[3] = {0, 2L * npz, npz}, b_off[3] = {0, 2 * plane, plane};
    if (int e = st::gemm_nn_g3_batched(zf, 3L * npz, (long)p.rows_pad * 3 * npz, a_off, gfwd, npz, 3 * plane, b_off, xf, nb, (long)p.rows_pad * nb,
                                       cph, p.rows_pad, npz, cph, p.bins, s, true))
      return e;
  } else if (int e = st::gemm_nn_batched(zf + (zf3 ? npz : 0), zf3 ? 3L * npz : kz, (long)p.rows_pad * (zf3 ? 3 * npz : kz), gfwd, (long)nb * kz, xf, nb,
                                         (long)p.rows_pad * nb, p.rows_pad, kz, nb, p.bins, s, sk, true))
    return e;
  RowsOut out{dx->base + (long)dx->halo * dx->c_pitch, (long)dx->t_pitch * dx->c_pitch, dx->c_pitch, dx->channels, dx->frames, nullptr};
  const int nchunks = st::ceil_div(dx->c_pitch, 32);
  const float* mask = act ? act->base + (long)act->halo * act->c_pitch : nullptr;
  const long mask_bs = act ? (long)act->t_pitch * act->c_pitch : 0L;
  const int right = width - 1 - pad_left;
  const bool window_form = p.blocks <= FUSE_BLOCKS && p.rows == p.rows_pad && pad_left <= FUSE_HALO && right <= FUSE_HALO &&
                           st::tuning(st::TUNE_NO_FUSED_TRANSFORMS) == 0;
  if (window_form) {
    // the layer below's dz spectra ride along when its zf rows are laid out like this call's columns
    const bool below = below_tables && below_zf && width_ok(below_width) && npad_of(dx->channels) % 128 == 0 &&
                       npad_of(dx->channels) == cph && 32 * nchunks == cph && !zf3_form(below_width, dx->channels);
    const Plan pb = make_plan(below ? below_width : width, dx->frames, dx->batch, ROWS_F32);
    // gflop: the whole-window inverse (2 * HP k-steps, three 32-row tiles), then the 32 k-steps of the layer below's zero-padded
    // forward transform (three tiles); mb: this layer's dx spectra and the ReLU mask in, dx and the layer below's dz spectra out
    st::trace("idft_ola_dft_rows<%d%s> rows=%d chunks=%d bins=%d gflop=%.3f mb=%.2f", p.bins <= 36 ? 18 : 24, below ? ",dz-spectra" : "", p.rows, nchunks, p.bins,
              4096e-9 * p.rows * (double)nchunks * (2 * (p.bins <= 36 ? 18 : 24) * 3 + (below ? 32 * 3 : 0)),
              1e-6 * 4 * ((double)p.bins * p.rows * 2 * cph + (double)p.rows * V * dx->c_pitch * (mask ? 2 : 1) + (below ? (double)pb.bins * p.rows * 2 * cph : 0.0)));
    const dim3 grid(dx->batch * nchunks), block(64 * FUSE_BLOCKS);
    const float* wz = below ? below_tables + T_ZP : nullptr;
    st::LaunchTimer timer(s);
    if (p.bins <= 36)
      st::launch_timed(timer, idft_ola_dft_rows_kernel<18>, grid, block, s, xf, tables + T_IW, p.blocks, p.rows_pad, p.bins, cph, nchunks, out,
                       pad_left, right, mask, mask_bs, act ? act->c_pitch : 0, wz, pb.bins, cph, below ? below_zf : nullptr);
    else
      st::launch_timed(timer, idft_ola_dft_rows_kernel<24>, grid, block, s, xf, tables + T_IW, p.blocks, p.rows_pad, p.bins, cph, nchunks, out,
                       pad_left, right, mask, mask_bs, act ? act->c_pitch : 0, wz, pb.bins, cph, below ? below_zf : nullptr);
    if (below && below_zf_written) *below_zf_written = 1;
    return st::check_launch("conv fft bwd_data (window form)");
  }
  launch_idft<3>(xf, tables + T_IX, p, cph, nchunks, out, nullptr, 0, mask, mask_bs, act ? act->c_pitch : 0, s);
  return st::check_launch("conv fft bwd_data");
}

int st_conv1d_nwc_bwd_data_fft_f32(const st_tensor3* dz, const float* zf, const float* gfwd, int width, int pad_left,
                                   const st_tensor3* act, const st_tensor3* dx, const float* tables, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  return st_conv1d_nwc_bwd_data_fft_chain_f32(dz, zf, gfwd, width, pad_left, act, dx, tables, nullptr, nullptr, 0, nullptr, workspace,
                                              workspace_bytes, stream);
}

// ======== the same three operations with the per-bin products on the bf16 matrix pipe (conv_bf16.hip) ===================
// planes = 1: bf16 activations (BASELINE configs[3]) -- tensors are read / written in their bf16 form, spectra and filter
//             spectra are ONE bf16 plane, every accumulation (DFT, products, inverse DFT) is fp32;
// planes = 3: fp32 tensors; every spectrum value is split EXACTLY into three bf16 planes and a product evaluated as the six
//             largest cross terms with fp32 accumulation (the bf16x6 scheme of conv_bf16.hip: at least as accurate as an fp32
//             FMA chain).
// Layouts: spectra planes [bins][rows_pad][cols] as in the fp32 form; filter spectra twice -- `g_planes` [bins][2 cph][2 npo]
// (the matrix gfwd; back-prop to the input multiplies by its transpose, so this is that product's k-contiguous operand) and
// `gt_planes` [2 npo][bins][2 cph] (its transpose per bin, rows holding all bins: the forward product's operand).

size_t st_conv1d_fft_filter_plane_elems(int width, int cin_pitch, int cout) { return st_conv1d_fft_filter_floats(width, cin_pitch, cout); }

int st_conv1d_fft_filters_planes(const float* packed, int width, int cin, int cout, int cin_pitch, const float* tables, void* g_planes,
                                 void* gt_planes, int planes, void* stream) {
  ST_REQUIRE(width_ok(width) && cin_pitch % 16 == 0 && tables && packed && g_planes && gt_planes && planes_ok(planes), "fft filter planes: bad argument");
  ST_REQUIRE(npad_of(cout) % 128 == 0, "fft filter planes: the output channels must pack to a multiple of 128");
  hipStream_t s = st::as_stream(stream);
  const int n = V + width - 1, bins = n / 2 + 1, npo = npad_of(cout), cph = half_of(cin_pitch);
  const f32x2* tw = reinterpret_cast<const f32x2*>(tables + T_FW);
  const size_t plane = (size_t)bins * 2 * cph * 2 * npo;
  const int gx = st::ceil_div(npo, 256), gy = cph;
  const dim3 grid(gx, gy, gx * gy < 1024 ? 4 : 1);
  unsigned short* g = reinterpret_cast<unsigned short*>(g_planes);
  if (planes == 1) launch_filters_planes<1>(width, grid, s, packed, cin, cout, cin_pitch, cph, npo, n, bins, tw, g, plane);
  else launch_filters_planes<3>(width, grid, s, packed, cin, cout, cin_pitch, cph, npo, n, bins, tw, g, plane);
  if (int e = st::check_launch("fft filter planes")) return e;
  for (int pl = 0; pl < planes; ++pl)
    if (int e = st::transpose_bf16_bins(g + pl * plane, reinterpret_cast<unsigned short*>(gt_planes) + pl * plane, bins, 2 * cph, 2 * npo, s))
      return e;
  return ST_OK;
}

// elements per plane of the input spectra / of the dz spectra (the fp32 forms' float counts)
size_t st_conv1d_fft_planes_ws(const st_tensor3* x, const st_tensor3* y, int width, int planes) {
  if (!x || !y || !width_ok(width) || !planes_ok(planes)) return 0;
  const Plan p = make_plan(width, y->frames, y->batch);
  const size_t nf = 2 * (size_t)npad_of(y->channels), ka = 2 * (size_t)half_of(x->c_pitch);
  const size_t red = (size_t)p.bins * p.rows_pad;
  // [stream-K area (unused here, kept for a common layout) | fp32 product spectra: max(yf, xf, qf) | reduction-major bf16 copies
  //  of both spectra for the lag products, `planes` each]
  const size_t prod = std::max((size_t)p.bins * p.rows_pad * nf, (size_t)p.bins * ka * nf);
  // (the input spectra twice when the lag products run in their split form: both operands of it, st::transpose_bf16_bins_split)
  return (st::SK_WS_FLOATS + prod + 64) * sizeof(float) + planes * (2 * ka + nf) * red * 2 + 512;
}

int st_conv1d_nwc_fwd_fft_planes(const st_tensor3* x, const void* x_bf16, const void* gt_planes, const float* bias, int width,
                                 int pad_left, int relu, const st_tensor3* y, void* y_bf16, const float* tables, void* sf_planes,
                                 int planes, void* workspace, size_t workspace_bytes, void* stream) {
  ST_REQUIRE(tensor_ok(x) && tensor_ok(y) && gt_planes && sf_planes && workspace && tables && width_ok(width) && planes_ok(planes),
             "conv fft planes fwd: bad argument");
  ST_REQUIRE((planes == 1) == (x_bf16 != nullptr) && (planes == 1) == (y_bf16 != nullptr),
             "conv fft planes fwd: one plane goes with bf16 tensors, three planes with fp32 tensors");
  ST_REQUIRE(x->batch == y->batch && x->frames == y->frames && pad_left >= 0 && pad_left < width, "conv fft planes fwd: stride-1 SAME layers only");
  ST_REQUIRE(npad_of(y->channels) % 128 == 0 && workspace_bytes >= st_conv1d_fft_planes_ws(x, y, width, planes), "conv fft planes fwd: workspace / shape");
  hipStream_t s = st::as_stream(stream);
  const Plan p = make_plan(width, y->frames, y->batch);
  const int ka = 2 * half_of(x->c_pitch), npo = npad_of(y->channels), nf = 2 * npo;
  const size_t s_plane = (size_t)p.bins * p.rows_pad * ka, g_plane = (size_t)p.bins * ka * nf;
  float* yf = reinterpret_cast<float*>(workspace) + st::SK_WS_FLOATS;
  launch_dft(*x, x_bf16, p, tables + T_FS, -pad_left, p.n, half_of(x->c_pitch), nullptr, sf_planes, planes, s_plane, nullptr, s);
  // Y[bin] = S[bin] (rows x ka) * gfwd[bin] (ka x nf): the operand is gt[n][bin * ka + k]
  if (int e = st::gemm_bf16_bins(planes, sf_planes, s_plane, ka, (long)p.rows_pad * ka, gt_planes, g_plane, (long)p.bins * ka, ka, yf, nf,
                                 p.rows_pad, ka, nf, p.bins, s))
    return e;
  RowsOut out{y->base + (long)y->halo * y->c_pitch, (long)y->t_pitch * y->c_pitch, y->c_pitch, y->channels, y->frames,
              y_bf16 ? reinterpret_cast<unsigned short*>(y_bf16) + (long)y->halo * y->c_pitch : nullptr};
  launch_idft<1>(yf, tables + T_IY, p, npo, st::ceil_div(y->c_pitch, 32), out, bias, relu, nullptr, 0L, 0, s);
  return st::check_launch("conv fft planes fwd");
}

int st_conv1d_fft_dz_spectra_planes(const st_tensor3* dz, const void* dz_bf16, int width, const float* tables, void* zf_planes,
                                    int planes, float* dc, void* stream) {
  ST_REQUIRE(tensor_ok(dz) && tables && zf_planes && width_ok(width) && planes_ok(planes) && npad_of(dz->channels) % 128 == 0 &&
                 (planes == 1) == (dz_bf16 != nullptr), "conv fft planes dz spectra: bad argument");
  const Plan p = make_plan(width, dz->frames, dz->batch);
  const int npo = npad_of(dz->channels);
  launch_dft(*dz, dz_bf16, p, tables + T_FZ, 0, V, npo, nullptr, zf_planes, planes, (size_t)p.bins * p.rows_pad * 2 * npo, dc,
             st::as_stream(stream));
  return st::check_launch("conv fft planes dz spectra");
}

// dbias[o] = sum over the blocks of their fp32 frame sums dc[row][o] (st_conv1d_fft_dz_spectra_planes)
int st_conv1d_fft_bias_grad_dc_f32(const float* dc, int rows, int channels, int n_pad, float* dbias, void* stream) {
  ST_REQUIRE(dc && dbias && rows > 0 && channels > 0 && n_pad >= channels, "conv fft bias grad (dc): bad argument");
  hipLaunchKernelGGL(bias_from_spectra_kernel, dim3(st::ceil_div(n_pad, 32)), dim3(256), 0, st::as_stream(stream), dc, rows, n_pad, channels,
                     n_pad, dbias);
  return st::check_launch("conv fft bias grad (dc)");
}

int st_conv1d_nwc_bwd_data_fft_planes(const st_tensor3* dz, const void* zf_planes, const void* g_planes, int width, int pad_left,
                                      const st_tensor3* act, const void* act_bf16, const st_tensor3* dx, void* dx_bf16,
                                      const float* tables, int planes, void* workspace, size_t workspace_bytes, void* stream) {
  ST_REQUIRE(tensor_ok(dz) && tensor_ok(dx) && zf_planes && g_planes && workspace && tables && width_ok(width) && planes_ok(planes),
             "conv fft planes bwd_data: bad argument");
  ST_REQUIRE(dz->batch == dx->batch && dz->frames == dx->frames && pad_left >= 0 && pad_left < width, "conv fft planes bwd_data: stride-1 layers only");
  ST_REQUIRE((planes == 1) == (dx_bf16 != nullptr) && (!act || (planes == 1) == (act_bf16 != nullptr)),
             "conv fft planes bwd_data: one plane goes with bf16 tensors, three planes with fp32 tensors");
  ST_REQUIRE(npad_of(dz->channels) % 128 == 0 && workspace_bytes >= st_conv1d_fft_planes_ws(dx, dz, width, planes), "conv fft planes bwd_data: workspace / shape");
  if (act) ST_REQUIRE(tensor_ok(act) && act->batch == dx->batch && act->frames == dx->frames && act->c_pitch >= dx->c_pitch,
                      "conv fft planes bwd_data: mask tensor mismatch");
  hipStream_t s = st::as_stream(stream);
  const Plan p = make_plan(width, dz->frames, dz->batch);
  const int kz = 2 * npad_of(dz->channels), cph = half_of(dx->c_pitch), nb = 2 * cph;
  float* xf = reinterpret_cast<float*>(workspace) + st::SK_WS_FLOATS;
  // X[bin] = Z[bin] (rows x kz) * gfwd[bin]^T (kz x nb): gfwd itself is the k-contiguous operand
  if (int e = st::gemm_bf16_bins(planes, zf_planes, (size_t)p.bins * p.rows_pad * kz, kz, (long)p.rows_pad * kz, g_planes, (size_t)p.bins * nb * kz,
                                 kz, (long)nb * kz, xf, nb, p.rows_pad, kz, nb, p.bins, s))
    return e;
  RowsOut out{dx->base + (long)dx->halo * dx->c_pitch, (long)dx->t_pitch * dx->c_pitch, dx->c_pitch, dx->channels, dx->frames,
              dx_bf16 ? reinterpret_cast<unsigned short*>(dx_bf16) + (long)dx->halo * dx->c_pitch : nullptr};
  const void* mask = nullptr;
  if (act) mask = act_bf16 ? static_cast<const void*>(reinterpret_cast<const unsigned short*>(act_bf16) + (long)act->halo * act->c_pitch)
                           : static_cast<const void*>(act->base + (long)act->halo * act->c_pitch);
  launch_idft<3>(xf, tables + T_IX, p, cph, st::ceil_div(dx->c_pitch, 32), out, nullptr, 0, mask, act ? (long)act->t_pitch * act->c_pitch : 0L,
                 act ? act->c_pitch : 0, s);
  return st::check_launch("conv fft planes bwd_data");
}

int st_conv1d_nwc_bwd_filter_fft_planes(const st_tensor3* x, const st_tensor3* dz, const void* sf_planes, const void* zf_planes, int width,
                                        const float* tables, float* dpacked, int planes, void* workspace, size_t workspace_bytes,
                                        void* stream) {
  ST_REQUIRE(tensor_ok(x) && tensor_ok(dz) && sf_planes && zf_planes && dpacked && workspace && tables && width_ok(width) && planes_ok(planes),
             "conv fft planes bwd_filter: bad argument");
  ST_REQUIRE(x->batch == dz->batch && x->frames == dz->frames, "conv fft planes bwd_filter: stride-1 layers only");
  ST_REQUIRE(npad_of(dz->channels) % 128 == 0 && workspace_bytes >= st_conv1d_fft_planes_ws(x, dz, width, planes), "conv fft planes bwd_filter: workspace / shape");
  hipStream_t s = st::as_stream(stream);
  const Plan p = make_plan(width, dz->frames, dz->batch);
  const f32x2* tw = reinterpret_cast<const f32x2*>(tables + T_FW);
  const int half = half_of(x->c_pitch), ka = 2 * half, npo = npad_of(dz->channels), nf = 2 * npo;
  const long red = (long)p.bins * p.rows_pad;
  const size_t prod = std::max((size_t)p.bins * p.rows_pad * nf, (size_t)p.bins * ka * nf);
  float* qf = reinterpret_cast<float*>(workspace) + st::SK_WS_FLOATS;
  unsigned short* st_planes = reinterpret_cast<unsigned short*>(qf + prod + 64);           // [planes][ka][red] (split: [half][4 red])
  unsigned short* zt_planes = st_planes + (size_t)planes * 2 * ka * red;                   // [planes][nf][red] (split: [npo][2 red])
  const unsigned short* sfp = reinterpret_cast<const unsigned short*>(sf_planes);
  const unsigned short* zfp = reinterpret_cast<const unsigned short*>(zf_planes);
  if (split_lag_products(half) && planes == 1 && npo % 128 == 0 && (2 * p.rows_pad) % 32 == 0 && st::tuning(st::TUNE_BF16_LAG_COPIES) == 0) {
    // round 5: the same two products per bin straight from the spectra planes -- both operands are reduction-major as they lie,
    // and ds_read_b64_tr_b16 hands the matrix pipe its reduction-minor fragments (wgrad_tr_bf16.hip); the rotated operand is a
    // register shuffle there.  No transposing copies (2 launches, 52 us per step at config 2); st_set_tuning("bf16_lag_copies", 1)
    // keeps the round-4 form for A/B runs and for the parity test that holds the two against each other.
    if (int e = st::lag_products_tr_bf16(sfp, zfp, p.bins, p.rows_pad, half, npo, qf, s)) return e;
    launch_filters_idft(qf, width, x->channels, dz->channels, x->c_pitch, half, npo, p.n, p.bins, tw, dpacked, 1, s);
    return st::check_launch("conv fft planes bwd_filter");
  }
  if (split_lag_products(half)) {
    // the split form of st_conv1d_nwc_bwd_filter_fft_f32: Re Q and Im Q as plain products over 2 * rows_pad (part, row) pairs,
    // the rotated operand S' = [S_i | -S_r] formed by the transposing copy (a sign flip of bf16 is exact); batch 2 b + j reads
    // the dz spectra of bin b
    for (int pl = 0; pl < planes; ++pl) {
      if (int e = st::transpose_bf16_bins_split(sfp + (size_t)pl * red * ka, st_planes + (size_t)pl * 2 * ka * red, p.bins, p.rows_pad, ka, 2, s)) return e;
      if (int e = st::transpose_bf16_bins_split(zfp + (size_t)pl * red * nf, zt_planes + (size_t)pl * nf * red, p.bins, p.rows_pad, nf, 1, s)) return e;
    }
    if (int e = st::gemm_bf16_bins(planes, st_planes, (size_t)2 * ka * red, 4 * red, 2L * p.rows_pad, zt_planes, (size_t)nf * red, 2 * red,
                                   2L * p.rows_pad, qf, npo, half, 2 * p.rows_pad, npo, 2 * p.bins, s, 1))
      return e;
    launch_filters_idft(qf, width, x->channels, dz->channels, x->c_pitch, half, npo, p.n, p.bins, tw, dpacked, 1, s);
    return st::check_launch("conv fft planes bwd_filter");
  }
  for (int pl = 0; pl < planes; ++pl) {
    if (int e = st::transpose_bf16_bins(sfp + (size_t)pl * red * ka, st_planes + (size_t)pl * ka * red, p.bins, p.rows_pad, ka, s)) return e;
    if (int e = st::transpose_bf16_bins(zfp + (size_t)pl * red * nf, zt_planes + (size_t)pl * nf * red, p.bins, p.rows_pad, nf, s)) return e;
  }
  // Q[bin] = S[bin]^T (ka x rows) * Z[bin] (rows x nf): both operands reduction-major, rows of all bins behind each other
  if (int e = st::gemm_bf16_bins(planes, st_planes, (size_t)ka * red, red, p.rows_pad, zt_planes, (size_t)nf * red, red, p.rows_pad, qf, nf, ka,
                                 p.rows_pad, nf, p.bins, s))
    return e;
  launch_filters_idft(qf, width, x->channels, dz->channels, x->c_pitch, half, npo, p.n, p.bins, tw, dpacked, 0, s);
  return st::check_launch("conv fft planes bwd_filter");
}

int st_conv1d_nwc_bwd_filter_fft_f32(const st_tensor3* x, const st_tensor3* dz, const float* sf, const float* zf, int width,
                                     const float* tables, float* dpacked, void* workspace, size_t workspace_bytes, void* stream) {
  ST_REQUIRE(tensor_ok(x) && tensor_ok(dz) && sf && zf && dpacked && workspace && tables && width_ok(width), "conv fft bwd_filter: bad argument");
  ST_REQUIRE(x->batch == dz->batch && x->frames == dz->frames, "conv fft bwd_filter: stride-1 layers only");
  ST_REQUIRE(npad_of(dz->channels) % 128 == 0 && workspace_bytes >= st_conv1d_fft_ws(x, dz, width), "conv fft bwd_filter: workspace / shape");
  hipStream_t s = st::as_stream(stream);
  const Plan p = make_plan(width, dz->frames, dz->batch, ROWS_F32);
  const f32x2* tw = reinterpret_cast<const f32x2*>(tables + T_FW);
  const int half = half_of(x->c_pitch), ka = 2 * half, npo = npad_of(dz->channels), nf = 2 * npo;
  float* qf = reinterpret_cast<float*>(workspace) + st::SK_WS_FLOATS;
  const long s_bin = 2L * p.rows_pad * ka;                      // [S | rotated copy] per bin (st_conv1d_fft_sf_floats)
  const bool zf3 = zf3_form(width, dz->channels);
  if (g3_form(width, x->c_pitch, dz->channels)) {
    // the lag products in three products per bin (gemm_tn_g3_kernel):  Re Q = S_r^T (Z_r + Z_i) + (S_i - S_r)^T Z_i,
    // Im Q = (S_r + S_i)^T Z_r - S_r^T (Z_r + Z_i);  q comes out as the split form's [bins][2][half][npo]
    const long a_off[3] = {2L * half, 3L * half, 0}, z_off[3] = {0, 2L * npo, npo};
    if (int e = st::gemm_tn_g3_batched(sf, 4L * half, s_bin, a_off, zf, 3L * npo, (long)p.rows_pad * 3 * npo, z_off, qf, 2L * half * npo,
                                       (long)half * npo, p.rows_pad, half, npo, p.bins, s))
      return e;
    launch_filters_idft(qf, width, x->channels, dz->channels, x->c_pitch, half, npo, p.n, p.bins, tw, dpacked, 1, s);
  } else if (split_lag_products(half)) {
    // The lag products Q = S^H-like sums over the rows of a bin:  Re Q = S_r^T Z_r + S_i^T Z_i,  Im Q = S_i^T Z_r - S_r^T Z_i.
    // A spectra row [re | im] read as TWO rows of half length turns each into ONE plain product over 2 * rows_pad rows:
    // Re Q = S2^T Z2 with S2 = S as [2 rows][half], Z2 = Z as [2 rows][npo]; Im Q the same with the rotated copy [S_i | -S_r]
    // the forward pass wrote behind S.  Against the 2 x 2 block form of rounds 2-3 ([S_r | S_i]^T [Z_r | Z_i], four blocks
    // that filters_idft combined): half the output floats (L8: 403 -> 201 MB written and read back), a reduction twice as
    // long per output tile (16 stages instead of 8 of a kernel whose prologue and 64 KB epilogue were most of its time), and
    // the real and the imaginary product of a bin in one launch (batch 2 b + j reads Z of bin b).
    if (int e = st::gemm_tn_batched(sf, half, (long)p.rows_pad * ka, zf, npo, (long)p.rows_pad * nf, qf, (long)half * npo, 2 * p.rows_pad,
                                    half, npo, 2 * p.bins, s, 1))
      return e;
    launch_filters_idft(qf, width, x->channels, dz->channels, x->c_pitch, half, npo, p.n, p.bins, tw, dpacked, 1, s);
  } else {
    // (spectra halves that do not tile the kernel -- the polyphase first layer: 192 columns): Q[bin] = [S_r | S_i]^T [Z_r | Z_i],
    // 2 half x 2 npo, combined by filters_idft
    // (gradient spectra in three-part rows -- zf3_form -- are read from their [Z_r | Z_i] columns)
    if (int e = st::gemm_tn_batched(sf, ka, s_bin, zf + (zf3 ? npo : 0), zf3 ? 3L * npo : nf, (long)p.rows_pad * (zf3 ? 3 * npo : nf), qf, (long)ka * nf,
                                    p.rows_pad, ka, nf, p.bins, s))
      return e;
    launch_filters_idft(qf, width, x->channels, dz->channels, x->c_pitch, half, npo, p.n, p.bins, tw, dpacked, 0, s);
  }
  return st::check_launch("conv fft bwd_filter");
}

}  // extern "C"
