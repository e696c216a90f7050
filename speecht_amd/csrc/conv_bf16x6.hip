// EXPERIMENTAL (opt-in): fp32-accurate convolution GEMM on the bf16 matrix pipe ("bf16x6").
//
// Every fp32 operand is split EXACTLY into three bf16 pieces a = a_h + a_m + a_l (8 + 8 + 8
// significand bits, same exponent range as fp32), and a*b is evaluated as the six largest cross
// terms  a_h b_h + a_h b_m + a_m b_h + a_h b_l + a_l b_h + a_m b_m  on v_mfma_f32_32x32x16_bf16 with
// fp32 accumulation; the dropped terms are below 2^-24 relative.  Products of bf16 pairs are exact in
// fp32, so the result is at least as accurate as an fp32 FMA chain (numpy model, K = 8000: error
// 2.0e-6 vs 5.9e-6 for the chain, scale 2.6) while the matrix pipe runs 16/6 = 2.7x the fp32-MFMA
// rate.  (The same idea as cuBLAS's "BF16x9" fp32 emulation.)
//
// Operands are pre-split in HBM: activation planes keep the padded NWC geometry of the fp32 tensor
// (2-byte elements), filter planes are stored transposed [n_pad][k_pad] so that both MFMA operands
// are reduction-contiguous 16-byte fragments.
#include <algorithm>
#include <cstdlib>

#include "st_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int BKB = 32;           // bf16 elements of reduction per LDS stage (2 MFMA k-steps)
constexpr int NT_ = 256;

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;          // exact
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);            // exact: at most 8 significant bits remain
}

// elementwise split of a padded tensor (same geometry for the three planes)
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ src, size_t n4,
                                                     __bf16* __restrict__ ph, __bf16* __restrict__ pm,
                                                     __bf16* __restrict__ pl) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
    __bf16 h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split3(v[e], h[e], m[e], l[e]);
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    reinterpret_cast<bf16x4*>(ph)[i] = bf16x4{h[0], h[1], h[2], h[3]};
    reinterpret_cast<bf16x4*>(pm)[i] = bf16x4{m[0], m[1], m[2], m[3]};
    reinterpret_cast<bf16x4*>(pl)[i] = bf16x4{l[0], l[1], l[2], l[3]};
  }
}

// packed filters [Kp][Np] fp32 -> three planes [Np][Kp] bf16 (32x32 LDS transpose)
__global__ __launch_bounds__(256) void split3_transpose_kernel(const float* __restrict__ packed, int Kp, int Np,
                                                               __bf16* __restrict__ planes) {
  __shared__ float tile[32][33];
  const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) tile[r][tx] = (k0 + r < Kp) ? packed[(long)(k0 + r) * Np + n0 + tx] : 0.f;
  __syncthreads();
  const size_t plane = (size_t)Np * Kp;
  for (int r = ty; r < 32; r += 8) {
    if (k0 + tx < Kp) {
      __bf16 h, m, l;
      split3(tile[tx][r], h, m, l);
      const size_t o = (size_t)(n0 + r) * Kp + k0 + tx;
      planes[o] = h; planes[plane + o] = m; planes[2 * plane + o] = l;
    }
  }
}

// src: padded NWC fp32 tensor.  dst planes [c_rows][batch * tq] (reduction-major for the filter
// gradient): plane[c][b * tq + j] = src[b][row0 + j][c] for j < rows; everything else stays zero.
__global__ __launch_bounds__(256) void transpose_split3_kernel(const float* __restrict__ src, int rows, int row0,
                                                               int t_pitch, int c_pitch, int tq, size_t plane,
                                                               __bf16* __restrict__ dst) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int j0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* s = src + ((long)b * t_pitch + row0) * c_pitch;
  for (int r = ty; r < 32; r += 8)
    tile[r][tx] = (j0 + r < rows && c0 + tx < c_pitch) ? s[(long)(j0 + r) * c_pitch + c0 + tx] : 0.f;
  __syncthreads();
  const long row_len = (long)gridDim.z * tq;
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, j = j0 + tx;
    if (c < c_pitch && j < rows) {
      __bf16 h, m, l;
      split3(tile[tx][r], h, m, l);
      const size_t o = (size_t)c * row_len + (size_t)b * tq + j;
      dst[o] = h; dst[plane + o] = m; dst[2 * plane + o] = l;
    }
  }
}

struct RowMapB {
  int frames, row_stride;
  long batch_stride, row0;
  __device__ __forceinline__ long off(int m) const {
    int b = m / frames;
    int t = m - b * frames;
    return (long)b * batch_stride + row0 + (long)t * row_stride;
  }
};

struct X6Params {
  const __bf16* A; size_t a_plane;       // three activation planes, element stride between planes
  RowMapB amap;
  const __bf16* B; size_t b_plane;       // three transposed filter planes [Np][Kp]
  float* C; RowMapB cmap;
  const float* bias;
  const float* mask; RowMapB mmap;       // relu mask source (back-prop to the input), may be null
  __bf16* Cp; size_t c_plane;            // optional: the three planes of the output (same geometry as C)
  int M, Kvalid, Kp, Np, n_store, relu, taps, cp;
  int tiles_m, tiles_n, chunk;
};

// Square tile BM = BN = 32 * (number of waves); every wave stages rows [32w, 32w+32) of each of the
// three planes of both operands (1 KiB DMA pieces), so a stage costs 6 * (64 / rows-per-piece)
// DMA instructions per wave: 12 for <128, 2x2, BK 32>, 6 for <256, 2x4, BK 16>.
template <int BT, int WM, int WN, int BK>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nn_bf16x6_kernel(X6Params p) {
  constexpr int NW = WM * WN;
  constexpr int BM = BT, BN = BT;
  static_assert(BT == 32 * NW && (BK == 16 || BK == 32), "tile config");
  constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 32, NT = WTN / 32;
  constexpr int KS = BK / 16;                        // MFMA k-steps per stage
  constexpr int SLOTS = BK / 8;                      // 16-byte slots per row
  constexpr int RPB = 128 / BK;                      // rows per 256-byte bank span
  constexpr int RPP = 512 / BK;                      // rows per 1-KiB DMA piece
  constexpr int PPW = 32 / RPP;                      // pieces per wave, plane and operand
  constexpr int PL = BM * BK;                        // elements per plane tile
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * 3 * PL * 2 + 12 * BM];
  unsigned short* const As = smem;                   // [buf][plane][BM][BK]
  unsigned short* const Bs = smem + 2 * 3 * PL;
  long* const a_off = reinterpret_cast<long*>(smem + 4 * 3 * PL);
  long* const c_off = a_off + BM;
  long* const m_off = c_off + BM;

  const int bid = blockIdx.x;
  const int idx = (bid & 7) * p.chunk + (bid >> 3);
  if ((bid >> 3) >= p.chunk || idx >= p.tiles_m * p.tiles_n) return;
  const int tile_n = idx / p.tiles_m;
  const int tile_m = idx - tile_n * p.tiles_m;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  if (tid < BM) {
    int m = m0 + tid;
    bool valid = m < p.M;
    int mm = valid ? m : p.M - 1;
    a_off[tid] = p.amap.off(mm);
    c_off[tid] = valid ? p.cmap.off(mm) : -1;
    m_off[tid] = p.mask ? p.mmap.off(mm) : 0;
  }
  __syncthreads();

  // physical 16-byte slot s of row r holds source slot s ^ ((r / RPB) % SLOTS)
  const int prow = lane / SLOTS, pslot = lane % SLOTS;
  const __bf16* asrc[PPW];
  const __bf16* bsrc[PPW];
  int slot8[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int r = wave * 32 + i * RPP + prow;
    asrc[i] = p.A + a_off[r];
    slot8[i] = (pslot ^ ((r / RPB) % SLOTS)) * 8;
    bsrc[i] = p.B + (long)min(n0 + r, p.Np - 1) * p.Kp;
  }
  const int ktail = p.Kvalid - 8;
  constexpr int N_DMA = 6 * PPW;
  auto dma_piece = [&](int pc, int k0, int buf) {        // pc -> (operand, plane, i)
    const int op = pc / (3 * PPW), pl = (pc / PPW) % 3, i = pc % PPW;
    if (op == 0) {
      const __bf16* g = asrc[i] + pl * p.a_plane + min(k0 + slot8[i], ktail);
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(As + (buf * 3 + pl) * PL + (wave * 32 + i * RPP) * BK), 16, 0, 0);
    } else {
      const __bf16* g = bsrc[i] + pl * p.b_plane + min(k0 + slot8[i], p.Kp - 8);
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Bs + (buf * 3 + pl) * PL + (wave * 32 + i * RPP) * BK), 16, 0, 0);
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses (elements): row * BK + ((KS-step slot) ^ swizzle) * 8
  int a_frag[KS], b_frag[KS];
  {
    const int ra = wm * WTM + l31, rb = wn * WTN + l31;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      a_frag[ks] = ra * BK + (((2 * ks + h) ^ ((ra / RPB) % SLOTS)) * 8);
      b_frag[ks] = rb * BK + (((2 * ks + h) ^ ((rb / RPB) % SLOTS)) * 8);
    }
  }

  const bool tap_inner = p.taps > 1;
  const int chunks = (p.cp + BK - 1) / BK;
  const int nk = tap_inner ? chunks * p.taps : p.Kp / BK;
  int tap = 0, chunk = 0;
  auto tile_k0 = [&](int t, int c) { return tap_inner ? t * p.cp + c * BK : c * BK; };
  auto tile_ks = [&](int t, int c) {
    const int valid = tap_inner ? p.cp - c * BK : p.Kvalid - c * BK;
    return valid >= BK ? KS : (valid + 15) / 16;
  };
#pragma unroll
  for (int pc = 0; pc < N_DMA; ++pc) dma_piece(pc, tile_k0(0, 0), 0);
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const int nks = tile_ks(tap, chunk);
    int ntap = tap, nchunk = chunk;
    if (tap_inner) { if (++ntap == p.taps) { ntap = 0; ++nchunk; } } else { ++nchunk; }
    const bool more = kt + 1 < nk;
    const int nk0 = tile_k0(ntap, nchunk);
    tap = ntap; chunk = nchunk;
    const unsigned short* as = As + cur * 3 * PL;
    const unsigned short* bs = Bs + cur * 3 * PL;
    bf16x8 af[KS][3][MT], bf[KS][3][NT];
    auto read_frags = [&](int ks) {
#pragma unroll
      for (int pl = 2; pl >= 0; --pl) {              // low planes first: their MFMAs are issued first
#pragma unroll
        for (int i = 0; i < MT; ++i) af[ks][pl][i] = *reinterpret_cast<const bf16x8*>(as + pl * PL + a_frag[ks] + i * 32 * BK);
#pragma unroll
        for (int n = 0; n < NT; ++n) bf[ks][pl][n] = *reinterpret_cast<const bf16x8*>(bs + pl * PL + b_frag[ks] + n * 32 * BK);
      }
    };
    read_frags(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (more) {
#pragma unroll
        for (int pc = ks * (N_DMA / KS); pc < (ks + 1) * (N_DMA / KS); ++pc) dma_piece(pc, nk0, cur ^ 1);
      }
      if (ks + 1 < KS) read_frags(ks + 1);
      __builtin_amdgcn_sched_barrier(0);
      if (ks < nks) {
#pragma unroll
        for (int t = 0; t < 6; ++t) {               // smallest terms first
          constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][TA[t]][i], bf[ks][TB[t]][n], acc[i][n], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: tile column = lane&31 -> output column, row = (r&3) + 8*(r>>2) + 4*h
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = n0 + wn * WTN + n * 32 + l31;
    const bool col_ok = col < p.n_store;
    const float bv = (p.bias && col_ok) ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const long co = c_off[row];
        if (co >= 0 && col_ok) {
          float v = acc[i][n][r] + bv;
          if (p.relu) v = fmaxf(v, 0.f);
          if (p.mask) v = p.mask[m_off[row] + col] > 0.f ? v : 0.f;
          p.C[co + col] = v;
          if (p.Cp) {                                   // the consumer's operand planes, split here once
            __bf16 sh, sm, sl;
            split3(v, sh, sm, sl);
            p.Cp[co + col] = sh;
            p.Cp[p.c_plane + co + col] = sm;
            p.Cp[2 * p.c_plane + co + col] = sl;
          }
        }
      }
  }
}

int npad_of(int cout) { return cout <= 32 ? 32 : (cout <= 64 ? 64 : (int)st::round_up(cout, 128)); }

}  // namespace

extern "C" {

// planes: 3 * n bf16 elements (h | m | l)
int st_exp_split3_bf16(const float* src, size_t n, void* planes, void* stream) {
  ST_REQUIRE(src && planes && n % 4 == 0, "split3: bad args");
  __bf16* pl = reinterpret_cast<__bf16*>(planes);
  const int blocks = (int)std::min<size_t>((n / 4 + 255) / 256, 4096);
  hipLaunchKernelGGL(split3_kernel, dim3(blocks), dim3(256), 0, st::as_stream(stream), src, n / 4, pl, pl + n, pl + 2 * n);
  return st::check_launch("split3");
}

// packed [k_pad][n_pad] fp32 -> planes 3 x [n_pad][k_pad] bf16
int st_exp_split3_transpose_bf16(const float* packed, int k_pad, int n_pad, void* planes, void* stream) {
  ST_REQUIRE(packed && planes && k_pad % 32 == 0 && n_pad % 32 == 0, "split3_transpose: bad args");
  hipLaunchKernelGGL(split3_transpose_kernel, dim3(k_pad / 32, n_pad / 32), dim3(256), 0, st::as_stream(stream), packed,
                     k_pad, n_pad, reinterpret_cast<__bf16*>(planes));
  return st::check_launch("split3_transpose");
}

static int launch_x6(X6Params& p, hipStream_t s) {
  static const int big = getenv("ST_X6_TILE") ? atoi(getenv("ST_X6_TILE")) : 256;
  const int BT = (big == 256 && p.Np % 256 == 0 && (long)st::ceil_div(p.M, 256) * (p.Np / 256) >= 192) ? 256 : 128;
  p.tiles_m = st::ceil_div(p.M, BT);
  p.tiles_n = p.Np / BT;
  p.chunk = st::ceil_div(p.tiles_m * p.tiles_n, 8);
  if (BT == 256) hipLaunchKernelGGL((gemm_nn_bf16x6_kernel<256, 2, 4, 16>), dim3(p.chunk * 8), dim3(512), 0, s, p);
  else if (big == 16) hipLaunchKernelGGL((gemm_nn_bf16x6_kernel<128, 2, 2, 16>), dim3(p.chunk * 8), dim3(256), 0, s, p);
  else hipLaunchKernelGGL((gemm_nn_bf16x6_kernel<128, 2, 2, 32>), dim3(p.chunk * 8), dim3(256), 0, s, p);
  return st::check_launch("gemm_nn_bf16x6");
}

static RowMapB map_of(const st_tensor3& t, int first_row, int frame_stride, int frames) {
  RowMapB m;
  m.frames = frames;
  m.row_stride = frame_stride * t.c_pitch;
  m.batch_stride = (long)t.t_pitch * t.c_pitch;
  m.row0 = (long)first_row * t.c_pitch;
  return m;
}

// forward conv on the bf16x6 path; x gives the geometry of the three activation planes;
// y_planes (nullable) receives the three planes of the output for the next layer
int st_exp_conv1d_fwd_bf16x6(const st_tensor3* x, const void* x_planes, const void* w_planes, const float* bias,
                             int width, int stride, int pad_left, int relu, const st_tensor3* y, void* y_planes,
                             void* stream) {
  ST_REQUIRE(x && y && x_planes && w_planes && y->base, "conv bf16x6: null argument");
  ST_REQUIRE(x->halo >= pad_left && y->frames == st::ceil_div(x->frames, stride) && x->c_pitch % 16 == 0,
             "conv bf16x6: bad geometry");
  X6Params p{};
  p.A = reinterpret_cast<const __bf16*>(x_planes);
  p.a_plane = (size_t)x->batch * x->t_pitch * x->c_pitch;
  p.amap = map_of(*x, x->halo - pad_left, stride, y->frames);
  p.Np = npad_of(y->channels);
  ST_REQUIRE(p.Np % 128 == 0, "conv bf16x6: n_pad must be a multiple of 128");
  p.Kvalid = width * x->c_pitch;
  p.Kp = (int)st::round_up(p.Kvalid, 32);
  p.B = reinterpret_cast<const __bf16*>(w_planes);
  p.b_plane = (size_t)p.Np * p.Kp;
  p.C = y->base;
  p.cmap = map_of(*y, y->halo, 1, y->frames);
  p.Cp = reinterpret_cast<__bf16*>(y_planes);
  p.c_plane = (size_t)y->batch * y->t_pitch * y->c_pitch;
  p.bias = bias;
  p.M = y->batch * y->frames;
  p.n_store = std::min(y->c_pitch, p.Np);
  p.relu = relu;
  p.taps = width;
  p.cp = x->c_pitch;
  return launch_x6(p, st::as_stream(stream));
}

// back-prop to the layer input on the bf16x6 path (stride-1 layers): dz planes x planes of the
// flipped/transposed filter operand; act (nullable) is the ReLU mask source
int st_exp_conv1d_bwd_data_bf16x6(const st_tensor3* dz, const void* dz_planes, const void* wt_planes, int width,
                                  int pad_left, const st_tensor3* act, const st_tensor3* dx, void* dx_planes,
                                  void* stream) {
  ST_REQUIRE(dz && dx && dz_planes && wt_planes && dx->base, "conv bwd bf16x6: null argument");
  const int lead = width - 1 - pad_left;
  ST_REQUIRE(lead >= 0 && dz->halo >= lead && dz->frames == dx->frames && dz->batch == dx->batch, "conv bwd bf16x6: bad geometry");
  X6Params p{};
  p.A = reinterpret_cast<const __bf16*>(dz_planes);
  p.a_plane = (size_t)dz->batch * dz->t_pitch * dz->c_pitch;
  p.amap = map_of(*dz, dz->halo - lead, 1, dx->frames);
  p.Np = npad_of(dx->channels);
  ST_REQUIRE(p.Np % 128 == 0, "conv bwd bf16x6: n_pad must be a multiple of 128");
  p.Kvalid = width * dz->c_pitch;
  p.Kp = (int)st::round_up(p.Kvalid, 32);
  p.B = reinterpret_cast<const __bf16*>(wt_planes);
  p.b_plane = (size_t)p.Np * p.Kp;
  p.C = dx->base;
  p.cmap = map_of(*dx, dx->halo, 1, dx->frames);
  p.Cp = reinterpret_cast<__bf16*>(dx_planes);
  p.c_plane = (size_t)dx->batch * dx->t_pitch * dx->c_pitch;
  if (act) {
    p.mask = act->base;
    p.mmap = map_of(*act, act->halo, 1, act->frames);
  }
  p.M = dx->batch * dx->frames;
  p.n_store = std::min(dx->c_pitch, p.Np);
  p.taps = width;
  p.cp = dz->c_pitch;
  return launch_x6(p, st::as_stream(stream));
}

// planes [c_rows][batch * tq] of a padded tensor, rows [row0, row0 + rows) of every utterance
int st_exp_transpose_split3_bf16(const st_tensor3* t, int row0, int rows, int tq, size_t plane_elems, void* planes,
                                 void* stream) {
  ST_REQUIRE(t && t->base && planes && rows > 0 && row0 >= 0 && row0 + rows <= t->t_pitch && tq >= rows &&
                 plane_elems >= (size_t)t->c_pitch * t->batch * tq, "transpose_split3: bad args");
  dim3 grid(st::ceil_div(rows, 32), st::ceil_div(t->c_pitch, 32), t->batch);
  hipLaunchKernelGGL(transpose_split3_kernel, grid, dim3(256), 0, st::as_stream(stream), t->base, rows, row0, t->t_pitch,
                     t->c_pitch, tq, plane_elems, reinterpret_cast<__bf16*>(planes));
  return st::check_launch("transpose_split3");
}

// filter gradient on the bf16x6 path (stride-1 layers): dF[(w,c)][n] = sum_r XT[c][r + w + lead] * dZT[n][r],
// r = b * tq + t.  xt_planes: [x.c_pitch][batch*tq (+ slack)], dzt_planes: [n_pad][batch*tq].
int st_exp_conv1d_bwd_filter_bf16x6(const void* xt_planes, const void* dzt_planes, int batch, int tq, int width,
                                    int cin_pitch, int x_first_row, int cout, float* dpacked, void* stream) {
  ST_REQUIRE(xt_planes && dzt_planes && dpacked && tq % 32 == 0 && cin_pitch % 16 == 0, "bwd_filter bf16x6: bad args");
  X6Params p{};
  const long red = (long)batch * tq;                   // reduction length
  p.A = reinterpret_cast<const __bf16*>(xt_planes);
  p.a_plane = (size_t)cin_pitch * red + 4096;          // planes are allocated with slack behind the last row
  p.amap.frames = cin_pitch;                           // output row k = w * cin_pitch + c
  p.amap.batch_stride = 1;                             // tap w shifts the window by one frame
  p.amap.row_stride = (int)red;                        // channel c selects the plane row
  p.amap.row0 = x_first_row;
  p.Np = npad_of(cout);
  ST_REQUIRE(p.Np % 128 == 0 && red < (1L << 31), "bwd_filter bf16x6: unsupported shape");
  p.Kvalid = (int)red;
  p.Kp = (int)red;
  p.B = reinterpret_cast<const __bf16*>(dzt_planes);
  p.b_plane = (size_t)p.Np * red;
  p.C = dpacked;
  p.M = width * cin_pitch;
  p.cmap.frames = p.M;
  p.cmap.batch_stride = 0;
  p.cmap.row_stride = p.Np;
  p.cmap.row0 = 0;
  p.n_store = p.Np;
  p.taps = 1;
  p.cp = cin_pitch;
  return launch_x6(p, st::as_stream(stream));
}

}  // extern "C"
