// Convolution GEMMs on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, fp32 accumulation).  One kernel
// template, two operand models selected by NP (bf16 planes per operand):
//
//   NP = 1  "bf16 activations" (BASELINE config 4): activations and activation gradients live in HBM as
//           bf16 (round-to-nearest-even, written once by the producing kernel's epilogue), filters are
//           fp32 masters with a bf16 transposed copy, every accumulation and the bias/ReLU epilogue is
//           fp32, the last layer's logits and everything in CTC / clip / Adam stay fp32.
//   NP = 3  "bf16x6" (opt-in, fp32-accurate), described next.
//
// bf16x6: fp32-accurate convolution GEMM on the bf16 matrix pipe.
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
#include <type_traits>
#include <utility>

namespace {

// f(integral_constant<int, 0>{}), ..., f(integral_constant<int, N - 1>{}): a loop whose index is a compile-time value
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

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

// one fp32 value -> NP planes at element offset o (NP = 1: plain round-to-nearest-even cast)
template <int NP>
__device__ __forceinline__ void store_planes(float v, __bf16* dst, size_t plane, size_t o) {
  if constexpr (NP == 1) {
    dst[o] = (__bf16)v;
  } else {
    __bf16 h, m, l;
    split3(v, h, m, l);
    dst[o] = h; dst[plane + o] = m; dst[2 * plane + o] = l;
  }
}

// elementwise split / cast of a padded tensor (same geometry for every plane)
template <int NP>
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ src, size_t n4, __bf16* __restrict__ dst) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
    if constexpr (NP == 1) {
      reinterpret_cast<bf16x4*>(dst)[i] = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    } else {
      __bf16 h[4], m[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split3(v[e], h[e], m[e], l[e]);
      reinterpret_cast<bf16x4*>(dst)[i] = bf16x4{h[0], h[1], h[2], h[3]};
      reinterpret_cast<bf16x4*>(dst + 4 * n4)[i] = bf16x4{m[0], m[1], m[2], m[3]};
      reinterpret_cast<bf16x4*>(dst + 8 * n4)[i] = bf16x4{l[0], l[1], l[2], l[3]};
    }
  }
}

// packed filters [Kp][Np] fp32 -> NP planes [Np][Kp] bf16 (32x32 LDS transpose)
template <int NP>
__global__ __launch_bounds__(256) void split_transpose_kernel(const float* __restrict__ packed, int Kp, int Np,
                                                               __bf16* __restrict__ planes) {
  __shared__ float tile[32][33];
  const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) tile[r][tx] = (k0 + r < Kp) ? packed[(long)(k0 + r) * Np + n0 + tx] : 0.f;
  __syncthreads();
  const size_t plane = (size_t)Np * Kp;
  for (int r = ty; r < 32; r += 8) {
    if (k0 + tx < Kp) store_planes<NP>(tile[tx][r], planes, plane, (size_t)(n0 + r) * Kp + k0 + tx);
  }
}

// src: padded NWC fp32 tensor.  dst planes [c_rows][batch * tq] (reduction-major for the filter
// gradient): plane[c][b * tq + j] = src[b][row0 + j][c] for j < rows; everything else stays zero.
template <int NP>
__global__ __launch_bounds__(256) void transpose_split_kernel(const float* __restrict__ src, int rows, int row0,
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
    if (c < c_pitch && j < rows) store_planes<NP>(tile[tx][r], dst, plane, (size_t)c * row_len + (size_t)b * tq + j);
  }
}

// bf16 source (bf16-activation mode): dst[c][b * tq + j] = src[b][row0 + step * j][c] for j < rows, c < c_pitch;
// zero for every other (c < c_rows, j < tq), so the GEMM never meets stale bits.
// step 2 de-interleaves the frames of a stride-2 layer's input into two phase planes (one job each).
// One launch serves all operands of a filter gradient (the phase planes of x, then dz): the jobs' channel tiles
// are stacked along blockIdx.y.  zero_tail: 4096 elements behind a plane that shifted taps may read.
struct TransposeJob {
  const unsigned short* src;
  unsigned short* dst;
  unsigned short* zero_tail;
  int rows, row0, step, t_pitch, c_pitch, c_rows, y_tiles;
  unsigned short flip;                                 // 0x8000: the copy is negated (sign bit of every element)
};
struct TransposeJobs {
  TransposeJob job[4];
  int n, tq;
  long pitch;                                          // row pitch of the transposed planes, >= batch * tq
};

__global__ __launch_bounds__(256) void transpose_bf16_kernel(TransposeJobs js) {
  // 64 frames x 64 channels per workgroup, 8-byte global accesses on both sides
  typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
  __shared__ unsigned short tile[64][66];
  int ty = blockIdx.y, k = 0;
  while (k + 1 < js.n && ty >= js.job[k].y_tiles) ty -= js.job[k++].y_tiles;
  const TransposeJob jb = js.job[k];
  const int tq = js.tq;
  const int b = blockIdx.z;
  const int j0 = blockIdx.x * 64, c0 = ty * 64;
  const int q = threadIdx.x & 15, r16 = threadIdx.x >> 4;
  if (jb.zero_tail && blockIdx.x == 0 && ty == 0 && b == 0) {
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
    const u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    reinterpret_cast<u16x8*>(jb.zero_tail)[threadIdx.x] = z;
    reinterpret_cast<u16x8*>(jb.zero_tail)[256 + threadIdx.x] = z;
  }
  const unsigned short* s = jb.src + ((long)b * jb.t_pitch + jb.row0) * jb.c_pitch;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int r = it * 16 + r16, c = c0 + q * 4;
    u16x4 v = {0, 0, 0, 0};
    if (j0 + r < jb.rows && c < jb.c_pitch) v = *reinterpret_cast<const u16x4*>(s + (long)(j0 + r) * jb.step * jb.c_pitch + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[r][q * 4 + e] = v[e];
  }
  __syncthreads();
  const long row_len = js.pitch;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int c = c0 + it * 16 + r16, j = j0 + q * 4;
    if (c < jb.c_rows && j < tq) {                    // zeros outside the source
      u16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = tile[q * 4 + e][it * 16 + r16] ^ jb.flip;
      *reinterpret_cast<u16x4*>(jb.dst + (size_t)c * row_len + (size_t)b * tq + j) = v;
    }
  }
  // the pad behind the last utterance of every row (shifted taps read into it)
  const int pad = (int)(row_len - (long)gridDim.z * tq);
  if (pad > 0 && blockIdx.x == 0 && b == (int)gridDim.z - 1) {
    for (int i = threadIdx.x; i < 64 * (pad / 4); i += 256) {
      const int c = c0 + i / (pad / 4), e = (i % (pad / 4)) * 4;
      if (c < jb.c_rows) *reinterpret_cast<u16x4*>(jb.dst + (size_t)c * row_len + (size_t)gridDim.z * tq + e) = u16x4{0, 0, 0, 0};
    }
  }
}

// back-prop operand of the bf16 path straight from the packed fp32 filters: the flipped/transposed filter
// stored reduction-contiguous is a row copy,  out[c][(W-1-w) * cout_pitch + o] = packed[w * cin_pitch + c][o]
__global__ __launch_bounds__(256) void filters_bwd_bf16_kernel(const float* __restrict__ packed, int width, int cin,
                                                               int cin_pitch, int cout_pitch, int n_pad, int kt_pad,
                                                               __bf16* __restrict__ out) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  const int q4 = cout_pitch / 4;
  const long total = (long)width * cin * q4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int o = (int)(i % q4) * 4;
    const long wc = i / q4;
    const int c = (int)(wc % cin), w = (int)(wc / cin);
    const f32x4 v = *reinterpret_cast<const f32x4*>(packed + ((long)w * cin_pitch + c) * n_pad + o);
    *reinterpret_cast<bf16x4*>(out + (long)c * kt_pad + (long)(width - 1 - w) * cout_pitch + o) =
        bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  }
}

struct RowMapB {
  int frames, row_stride;
  long batch_stride, row0;
  int phase_shift, phase_mask;     // b -> (b >> shift) * batch_stride + (b & mask) * phase_stride
  long phase_stride;
  __device__ __forceinline__ long off(int m) const {
    int b = m / frames;
    int t = m - b * frames;
    return (long)(b >> phase_shift) * batch_stride + (long)(b & phase_mask) * phase_stride + row0 +
           (long)t * row_stride;
  }
};

struct X6Params {
  const __bf16* A; size_t a_plane;       // NP activation planes, element stride between planes
  RowMapB amap;
  const __bf16* B; size_t b_plane;       // NP transposed filter planes [Np][Kp]
  float* C; RowMapB cmap;                // fp32 output (nullable when only planes are wanted)
  const float* bias;
  const float* mask;                     // relu mask source (back-prop to the input): fp32 tensor ...
  const __bf16* mask_b;                  // ... or bf16 tensor; both may be null
  RowMapB mmap;
  __bf16* Cp; size_t c_plane;            // optional: the NP planes of the output (same geometry as C)
  int M, Kvalid, Kp, Np, n_store, relu, taps, cp;
  int tiles_m, tiles_n, chunk;           // XCD-aware tile order: the 8 XCDs as a gm x gn grid over the tile grid,
  int gm, tm_per, tn_per;                // each XCD owns tm_per x tn_per tiles (chunk = tm_per * tn_per), see launch_gemm
  int splits; long slab_stride;          // reduction split (taps == 1): split s stores to C + s * slab_stride
  int b_bin_shift;                       // matrices 2^shift at a time share a B operand (B + (bin >> shift) * b_bin)
  int rows_per_bin; long b_bin;          // > 0: A / C are `M / rows_per_bin` matrices stacked along M (the frequency bins of
                                         // conv_fft.hip), the B operand of the bin a tile lies in starts at B + bin * b_bin
};

// Square tile BM = BN = 32 * (number of waves); every wave stages rows [32w, 32w+32) of each of the
// three planes of both operands (1 KiB DMA pieces), so a stage costs 6 * (64 / rows-per-piece)
// DMA instructions per wave: 12 for <128, 2x2, BK 32>, 6 for <256, 2x4, BK 16>.
// ST = LDS ring depth: stages kt+1 .. kt+ST-1 are in flight while stage kt feeds the matrix pipe; a wave
// waits with a counted s_waitcnt (its own oldest stage) before the per-stage barrier, not vmcnt(0).
// PP = ping-pong: the waves form two groups (one wave of each per SIMD) that run half a stage out of phase --
// while one group feeds the matrix pipe the other reads its fragments of the next stage from LDS and issues
// DMA, two barriers per stage.  Without it all waves leave the stage barrier together, burst-read LDS, and the
// matrix pipe idles for the whole burst.
// FAST = every reduction stage is whole (channel pitch / reduction length a multiple of BK): DMA sources are a
// per-lane pointer plus the uniform k0 (no clamps) and every stage runs all its MFMA k-steps (no tail test).
// SCH = 1 (round 3; 256 x 256 tile, FOUR waves = one per SIMD, 128 x 128 per wave, NP = 1, BK = 32, FAST): the wave
// interleaves its own fragment reads and DMA pieces between its MFMAs instead of sharing the SIMD with a partner wave.
// Why: in ping-pong every LDS / DMA instruction of the reading wave is issued beside the partner's streaming MFMAs and
// costs ~100+ cycles there (PMC, DESIGN 4.3: matrix pipe 44 % busy, the read phase twice the MFMA phase); one wave per
// SIMD hides up to ~5 single-issue instructions in each 32-cycle MFMA shadow (MI355X_MICROARCH.md), and the 128 x 128
// wave tile needs 0.5 fragment reads + 0.25 DMA pieces per MFMA instead of 0.75 + 0.25.  The stage is software
// pipelined across the per-stage barrier: the fragments of k-step 0 of stage kt + 1 are read during the second half of
// the MFMAs of stage kt (the barrier sits in the middle of a stage's second k-step), so no stage opens with an exposed
// LDS round trip.
template <int BT, int WM, int WN, int BK, int NP, int ST = 2, bool PP = false, bool FAST = false, int SCH = 0>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nn_bf16_kernel(X6Params p) {
  constexpr int NW = WM * WN;
  constexpr int BM = BT, BN = BT;
  static_assert(BT % (32 * NW) == 0 && (BK == 16 || BK == 32 || BK == 64) && (NP == 1 || NP == 3) && ST >= 2, "tile config");
  constexpr int RW = BT / NW;                        // rows of each operand a wave stages
  constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 32, NT = WTN / 32;
  constexpr int KS = BK / 16;                        // MFMA k-steps per stage
  constexpr int SLOTS = BK / 8;                      // 16-byte slots per row
  constexpr int RPB = 128 / BK;                      // rows per 256-byte bank span
  constexpr int RPP = 512 / BK;                      // rows per 1-KiB DMA piece
  constexpr int PPW = RW / RPP;                      // pieces per wave, plane and operand
  constexpr int PL = BM * BK;                        // elements per plane tile
  __shared__ __attribute__((aligned(16))) unsigned short smem[ST * NP * PL * 2 + 12 * BM];
  unsigned short* const As = smem;                   // [buf][plane][BM][BK]
  unsigned short* const Bs = smem + ST * NP * PL;
  long* const a_off = reinterpret_cast<long*>(smem + 2 * ST * NP * PL);
  long* const c_off = a_off + BM;
  long* const m_off = c_off + BM;

  const int split = blockIdx.x / (p.chunk * 8);
  const int bid = blockIdx.x - split * (p.chunk * 8);
  // block b runs on XCD b % 8 and every XCD has its own L2: an XCD owns a rectangle of the tile grid (chosen by the host to
  // minimise the operand bytes the eight L2s pull in together) and walks it row tiles fastest, so that the workgroups
  // running side by side share a filter panel
  const int xcd = bid & 7, local = bid >> 3;
  const int ln = local / p.tm_per, lm = local - ln * p.tm_per;
  const int tile_m = (xcd % p.gm) * p.tm_per + lm, tile_n = (xcd / p.gm) * p.tn_per + ln;
  if (local >= p.chunk || tile_m >= p.tiles_m || tile_n >= p.tiles_n) return;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  for (int t = tid; t < BM; t += 64 * NW) {
    int m = m0 + t;
    bool valid = m < p.M;
    int mm = valid ? m : p.M - 1;
    a_off[t] = p.amap.off(mm);
    c_off[t] = valid ? p.cmap.off(mm) : -1;
    m_off[t] = (p.mask || p.mask_b) ? p.mmap.off(mm) : 0;
  }
  __syncthreads();

  // physical 16-byte slot s of row r holds source slot s ^ ((r / RPB) % SLOTS)
  const int prow = lane / SLOTS, pslot = lane % SLOTS;
  const __bf16* asrc[PPW];
  const __bf16* bsrc[PPW];
  int slot8[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int r = wave * RW + i * RPP + prow;
    asrc[i] = p.A + a_off[r];
    slot8[i] = (pslot ^ ((r / RPB) % SLOTS)) * 8;
    bsrc[i] = p.B + (p.rows_per_bin > 0 ? (long)((m0 / p.rows_per_bin) >> p.b_bin_shift) * p.b_bin : 0L) + (long)min(n0 + r, p.Np - 1) * p.Kp;
  }
  const int ktail = p.Kvalid - 8;
  constexpr int N_DMA = 2 * NP * PPW;
  const __bf16* aptr[NP][PPW];
  const __bf16* bptr[NP][PPW];
#pragma unroll
  for (int pl = 0; pl < NP; ++pl)
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      aptr[pl][i] = asrc[i] + pl * p.a_plane + slot8[i];
      bptr[pl][i] = bsrc[i] + pl * p.b_plane + slot8[i];
    }
  auto dma_piece = [&](int pc, int k0, int buf) {        // pc -> (operand, plane, i)
    const int op = pc / (NP * PPW), pl = (pc / PPW) % NP, i = pc % PPW;
    if (op == 0) {
      const __bf16* g = FAST ? aptr[pl][i] + k0 : asrc[i] + pl * p.a_plane + min(k0 + slot8[i], ktail);
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(As + (buf * NP + pl) * PL + (wave * RW + i * RPP) * BK), 16, 0, 0);
    } else {
      const __bf16* g = FAST ? bptr[pl][i] + k0 : bsrc[i] + pl * p.b_plane + min(k0 + slot8[i], p.Kp - 8);
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Bs + (buf * NP + pl) * PL + (wave * RW + i * RPP) * BK), 16, 0, 0);
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

  // Reduction order: tap-inner for convolutions (taps > 1: all taps of a 32-deep channel chunk, then the next chunk),
  // plain for one-tap layers and filter gradients.  A stage is (tap t, chunk c) at reduction index k0 = t * cp + c * BK;
  // the cursors advance without branches (the loop body must stay free of them, see `stage`).
  const bool tap_inner = p.taps > 1;
  const int len = tap_inner ? p.cp : p.Kvalid;       // reduction length per tap
  const int chunks = (len + BK - 1) / BK;
  // reduction split (filter gradients of the small layers, back-prop through L8): this workgroup owns a
  // contiguous range of chunks (tap-inner convolutions split over channel chunks, every split walks all taps)
  const int per_split = (chunks + p.splits - 1) / p.splits;
  const int unit0 = split * per_split;
  const int nk = max(min(per_split, chunks - unit0), 0) * p.taps;
  const int wrap_inc = BK - (p.taps - 1) * p.cp;     // k0 step from the last tap of a chunk to the first of the next
  struct Cursor { int t, c, k0; };
  Cursor cc{0, unit0, unit0 * BK};                   // compute cursor
  Cursor ic = cc;                                    // DMA issue cursor (ST-1 stages ahead)
  auto tile_ks = [&](const Cursor& q) {
    const int valid = len - q.c * BK;
    return valid >= BK ? KS : (valid + 15) / 16;
  };
  auto advance = [&](Cursor& q) {
    const int t1 = q.t + 1;
    const bool wrap = t1 == p.taps;
    q.t = wrap ? 0 : t1;
    q.c += wrap ? 1 : 0;
    q.k0 += wrap ? wrap_inc : p.cp;
  };
  // every wave issues N_DMA pieces per stage and they retire in order, so "my stage kt+1 has landed" is
  // vmcnt <= (ST-2) * N_DMA while ST-1 stages are in flight; in the tail (nothing new issued) wait for all
  auto stage_sync = [&](bool full_ring) {
    if (ST > 2 && full_ring) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((ST - 2) * N_DMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  };
  static_assert((ST - 2) * N_DMA < 64, "vmcnt is a 6-bit counter");
#pragma unroll
  for (int sgi = 0; sgi < ST - 1; ++sgi) {
    if (sgi < nk) {
#pragma unroll
      for (int pc = 0; pc < N_DMA; ++pc) dma_piece(pc, ic.k0, sgi);
      advance(ic);
    }
  }
  stage_sync(nk >= ST - 1);

  auto mfma_terms = [&](bf16x8 (&af)[KS][NP][MT], bf16x8 (&bf)[KS][NP][NT], int ks) {
    constexpr int TERMS = NP == 3 ? 6 : 1;
#pragma unroll
    for (int t = 0; t < TERMS; ++t) {               // smallest terms first
      constexpr int TA[6] = {NP - 1, 0, NP / 2, NP / 2, 0, 0}, TB[6] = {0, NP - 1, NP / 2, 0, NP / 2, 0};
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ks][TB[t]][n], af[ks][TA[t]][i], acc[i][n], 0, 0, 0);
    }
  };
  auto phase_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto wait_stage = [&](bool full_ring) {          // my pieces of the next stage have landed
    if (full_ring) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * N_DMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  static_assert(!PP || (ST >= 3 && NW % 2 == 0), "ping-pong needs a ring of 3 and an even number of waves");

  // One stage of the reduction, instantiated per ring slot (CUR: being consumed; the one being filled is CUR - 1) so
  // that every LDS address is a register plus an immediate, and per "a successor stage certainly exists" (MORE_CT):
  // the steady-state loop below is then free of scalar branches -- with 32-cycle bf16 MFMAs, 16 per stage and wave,
  // the ring arithmetic and the `more` / group tests of a run-time loop body were a visible share of every stage.
  // PP (ping-pong): the waves form two groups (one wave of each per SIMD; GRP) that run half a stage out of phase:
  // while one feeds the matrix pipe (at raised priority) the other reads its fragments of the next stage and issues DMA.
  auto stage = [&](auto cur_c, auto more_c, auto grp_c, int kt) {
    constexpr int CUR = decltype(cur_c)::value, FILL = (CUR + ST - 1) % ST, GRP = decltype(grp_c)::value;
    constexpr bool MORE_CT = decltype(more_c)::value;
    const int nks = FAST ? KS : tile_ks(cc);
    if (!FAST) advance(cc);
    const bool more = MORE_CT || kt + ST - 1 < nk;
    const int nk0 = ic.k0;
    if (more) advance(ic);
    const unsigned short* as = As + CUR * NP * PL;
    const unsigned short* bs = Bs + CUR * NP * PL;
    bf16x8 af[KS][NP][MT], bf[KS][NP][NT];
    auto read_frags = [&](int ks) {
#pragma unroll
      for (int pl = NP - 1; pl >= 0; --pl) {         // low planes first: their MFMAs are issued first
#pragma unroll
        for (int i = 0; i < MT; ++i) af[ks][pl][i] = *reinterpret_cast<const bf16x8*>(as + pl * PL + a_frag[ks] + i * 32 * BK);
#pragma unroll
        for (int n = 0; n < NT; ++n) bf[ks][pl][n] = *reinterpret_cast<const bf16x8*>(bs + pl * PL + b_frag[ks] + n * 32 * BK);
      }
    };
    if constexpr (PP) {
      // ---- read phase (the other group is in its MFMA phase)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) read_frags(ks);
      if (more) {                                    // slot FILL was last read one phase ago by group 1
#pragma unroll
        for (int pc = 0; pc < N_DMA; ++pc) dma_piece(pc, nk0, FILL);
      }
      if (GRP == 1) wait_stage(more);                // this barrier is group 0's end-of-stage barrier
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      phase_barrier();
      // ---- MFMA phase: nothing but the matrix instructions (the other group reads / issues DMA)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        if (FAST || ks < nks) mfma_terms(af, bf, ks);
      __builtin_amdgcn_s_setprio(0);
      if (GRP == 0) wait_stage(more);
      phase_barrier();
    } else {
      read_frags(0);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (more) {
#pragma unroll
          for (int pc = ks * (N_DMA / KS); pc < (ks + 1) * (N_DMA / KS); ++pc) dma_piece(pc, nk0, FILL);
        }
        if (ks + 1 < KS) read_frags(ks + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (FAST || ks < nks) mfma_terms(af, bf, ks);
        __builtin_amdgcn_sched_barrier(0);
      }
      stage_sync(more);
    }
  };
  if constexpr (SCH == 1) {
    static_assert(!PP && NP == 1 && (KS == 2 || KS == 4) && MT % 2 == 0, "schedule 1: one plane, two or four k-steps per stage");
    bf16x8 fa[2][MT], fb[2][NT];                       // fragments of even / odd k-steps (double buffer, also across stages)
    auto reads = [&](auto slot_c, auto ks_c) {
      constexpr int SL = decltype(slot_c)::value, KI = decltype(ks_c)::value, BUF = KI & 1;
      const unsigned short* as = As + SL * PL;
      const unsigned short* bs = Bs + SL * PL;
      // in the order the MFMAs want them (LDS returns in order, so the first MFMA waits for two reads, not for all)
      fb[BUF][0] = *reinterpret_cast<const bf16x8*>(bs + b_frag[KI]);
      fa[BUF][0] = *reinterpret_cast<const bf16x8*>(as + a_frag[KI]);
#pragma unroll
      for (int n = 1; n < NT; ++n) fb[BUF][n] = *reinterpret_cast<const bf16x8*>(bs + b_frag[KI] + n * 32 * BK);
#pragma unroll
      for (int i = 1; i < MT; ++i) fa[BUF][i] = *reinterpret_cast<const bf16x8*>(as + a_frag[KI] + i * 32 * BK);
    };
    auto mfmas = [&](auto ks_c, int i0, int i1) {
      constexpr int BUF = decltype(ks_c)::value & 1;
#pragma unroll
      for (int i = i0; i < i1; ++i)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[BUF][n], fa[BUF][i], acc[i][n], 0, 0, 0);
    };
    // !FAST here means: every stage is whole EXCEPT possibly the last one (one-tap layers whose reduction length is a
    // multiple of 16 but not of BK, e.g. 2016 channels with 64-deep stages).  That stage runs `nks_last` k-steps and its DMA
    // sources are clamped into the row (what lies behind the valid range is never multiplied).
    const int nks_last = (FAST || unit0 + nk != chunks) ? KS : (len - (chunks - 1) * BK + 15) / 16;   // (!FAST: one tap)
    const int kp8 = p.Kp - 8;
    auto dma_piece1 = [&](int pc, int k0, int buf, bool clamp) {
      const int op = pc / PPW, i = pc % PPW;
      if (op == 0) {
        const __bf16* g = aptr[0][i] + (clamp ? min(k0, ktail - slot8[i]) : k0);
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(As + buf * PL + (wave * RW + i * RPP) * BK), 16, 0, 0);
      } else {
        const __bf16* g = bptr[0][i] + (clamp ? min(k0, kp8 - slot8[i]) : k0);
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Bs + buf * PL + (wave * RW + i * RPP) * BK), 16, 0, 0);
      }
    };
    constexpr int DPK = (N_DMA + KS - 2) / (KS - 1);   // DMA pieces per k-step, spread over the first KS - 1 k-steps
    auto stage1 = [&](auto cur_c, auto more_c, int kt) {
      constexpr int CUR = decltype(cur_c)::value, FILL = (CUR + ST - 1) % ST, NEXT = (CUR + 1) % ST;
      constexpr bool MORE_CT = decltype(more_c)::value;
      const bool more = MORE_CT || kt + ST - 1 < nk;   // a stage ST - 1 ahead exists: stage it into the slot freed last
      const bool next = MORE_CT || kt + 1 < nk;        // a stage kt + 1 exists: its first fragments are read in this one
      const int nks = (MORE_CT || FAST || kt + 1 < nk) ? KS : nks_last;
      const bool clamp = !FAST && !MORE_CT && kt + ST == nk;    // the stage being staged is the (short) last one
      const int nk0 = ic.k0;
      if (more) advance(ic);
      // ---- k-steps 0 .. KS - 2: the fragments of the k-step are in registers; between its 16 MFMAs go the 8 fragment
      // reads of the next k-step and a share of the DMA pieces of the stage ST - 1 ahead
      static_for<KS - 1>([&](auto ks_c) {
        constexpr int KI = decltype(ks_c)::value;
        reads(std::integral_constant<int, CUR>{}, std::integral_constant<int, KI + 1>{});
        constexpr int P0 = KI * DPK, P1 = (KI + 1) * DPK < N_DMA ? (KI + 1) * DPK : N_DMA;
        if (more) {
#pragma unroll
          for (int pc = P0; pc < P1; ++pc) dma_piece1(pc, nk0, FILL, clamp);
        }
        if (KI < nks) mfmas(ks_c, 0, MT);
#pragma unroll
        for (int k = 0; k < MT + NT; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA ...
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // ... one fragment read in its shadow
        }
#pragma unroll
        for (int k = 0; k < P1 - P0; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);   // address arithmetic / M0 of the piece
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // one DMA piece
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      // ---- last k-step, first half
      using KL = std::integral_constant<int, KS - 1>;
      if (KS - 1 < nks) mfmas(KL{}, 0, MT / 2);
      __builtin_amdgcn_sched_barrier(0);
      if (next) {
        // stage kt + 1 complete in LDS: my own pieces by the counted wait, everybody else's by the barrier -- which also
        // says that no wave still reads the slot the next stage's DMA (issued after this point) will overwrite
        if (more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((ST - 2) * N_DMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- last k-step, second half, with the first fragments of the next stage read underneath: two reads behind each of
      // the first four MFMAs, so that the last read is four MFMAs (~130 cycles) old when the next stage opens
      if (next) reads(std::integral_constant<int, NEXT>{}, std::integral_constant<int, 0>{});
      if (KS - 1 < nks) mfmas(KL{}, MT / 2, MT);
#pragma unroll
      for (int k = 0; k < (MT + NT) / 2; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    if (nk > 0) reads(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    int kt = 0;
    for (; kt + 2 * ST - 2 < nk; kt += ST)
      static_for<ST>([&](auto s_c) { stage1(s_c, std::true_type{}, kt + decltype(s_c)::value); });
    static_for<2 * ST - 2>([&](auto r_c) {
      constexpr int R = decltype(r_c)::value;
      if (kt + R < nk) stage1(std::integral_constant<int, R % ST>{}, std::false_type{}, kt + R);
    });
  }
  auto run = [&](auto grp_c) {
    constexpr int GRP = decltype(grp_c)::value;
    if (PP && GRP == 1) phase_barrier();             // group 1 runs one phase behind
    int kt = 0;
    for (; kt + 2 * ST - 2 < nk; kt += ST)           // every stage of the trip has a successor ST - 1 ahead
      static_for<ST>([&](auto s_c) { stage(s_c, std::true_type{}, grp_c, kt + decltype(s_c)::value); });
    // the last (up to 2 ST - 2) stages: ring slots go on round-robin, `more` is tested
    static_for<2 * ST - 2>([&](auto r_c) {
      constexpr int R = decltype(r_c)::value;
      if (kt + R < nk) stage(std::integral_constant<int, R % ST>{}, std::false_type{}, grp_c, kt + R);
    });
    if (PP && GRP == 0) phase_barrier();             // every wave executes the same number of barriers
  };
  if constexpr (SCH == 1) {
    (void)run;
  } else if constexpr (PP) {
    if (wave / (NW / 2) == 0) run(std::integral_constant<int, 0>{});   // waves i and i + NW/2 share a SIMD
    else run(std::integral_constant<int, 1>{});
  } else {
    run(std::integral_constant<int, 0>{});
  }

  // epilogue.  The MFMAs above were issued with the operands swapped (filter fragment first), so the
  // accumulator holds the TRANSPOSED 32x32 sub-tile: lane&31 = output row m, register r = output column
  // (r&3) + 8*(r>>2) + 4*h.  Every lane therefore owns runs of 4 consecutive channels of one row and the
  // epilogue moves 8-byte (bf16x4) / 16-byte (fp32x4) pieces instead of single elements.
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = wm * WTM + i * 32 + l31;
    const long co = c_off[row];
    const long mo = m_off[row];                          // valid for every tile row (clamped)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      // the ReLU mask of the 4 runs first: loads interleaved with the stores below would be serialised
      // (for all the compiler knows the output may alias the mask source)
      bool keep[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = n0 + wn * WTN + n * 32 + 8 * g + 4 * h;
        const int colc = col < p.n_store ? col : 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) keep[4 * g + e] = true;
        if (p.mask) {
          const f32x4 mv = *reinterpret_cast<const f32x4*>(p.mask + mo + colc);
#pragma unroll
          for (int e = 0; e < 4; ++e) keep[4 * g + e] = mv[e] > 0.f;
        }
        if (p.mask_b) {
          const bf16x4 mv = *reinterpret_cast<const bf16x4*>(p.mask_b + mo + colc);
#pragma unroll
          for (int e = 0; e < 4; ++e) keep[4 * g + e] = (float)mv[e] > 0.f;
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = n0 + wn * WTN + n * 32 + 8 * g + 4 * h;
        if (co < 0 || col >= p.n_store) continue;         // n_store is a multiple of 16: runs are all-or-nothing
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][n][4 * g + e];
        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (p.relu) v[e] = fmaxf(v[e], 0.f);
          v[e] = keep[4 * g + e] ? v[e] : 0.f;
        }
        if (p.C) *reinterpret_cast<f32x4*>(p.C + (long)split * p.slab_stride + co + col) = v;
        if (p.Cp) {                                       // the consumer's operand, written once
          if constexpr (NP == 1) {
            *reinterpret_cast<bf16x4*>(p.Cp + co + col) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
          } else {
            bf16x4 ph, pm, pl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              __bf16 sh, sm, sl;
              split3(v[e], sh, sm, sl);
              ph[e] = sh; pm[e] = sm; pl[e] = sl;
            }
            *reinterpret_cast<bf16x4*>(p.Cp + co + col) = ph;
            *reinterpret_cast<bf16x4*>(p.Cp + p.c_plane + co + col) = pm;
            *reinterpret_cast<bf16x4*>(p.Cp + 2 * p.c_plane + co + col) = pl;
          }
        }
      }
    }
  }
}

int npad_of(int cout) { return cout <= 32 ? 32 : (cout <= 64 ? 64 : (int)st::round_up(cout, 128)); }

// out[i] = sum_s slabs[s][i]
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ slabs, int n_slabs, size_t n4,
                                                       float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 a = reinterpret_cast<const f32x4*>(slabs)[i];
    for (int s = 1; s < n_slabs; ++s) a += reinterpret_cast<const f32x4*>(slabs + (size_t)s * n4 * 4)[i];
    reinterpret_cast<f32x4*>(out)[i] = a;
  }
}

// second half of a split convolution: v = [mask] relu?(sum_s slabs[s] + bias) -> the padded NWC output as NP
// bf16 planes and/or fp32
template <int NP>
__global__ __launch_bounds__(256) void slab_epilogue_kernel(const float* __restrict__ slabs, int n_slabs, long slab_stride,
                                                            int M, int Np, int n_store, const float* __restrict__ bias,
                                                            int relu, const float* __restrict__ mask_f,
                                                            const __bf16* __restrict__ mask_b, RowMapB mmap, RowMapB cmap,
                                                            __bf16* __restrict__ out, size_t c_plane,
                                                            float* __restrict__ out_f) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  const int q = n_store / 4;                                     // n_store is a multiple of 16
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < (long)M * q; idx += (long)gridDim.x * 256) {
    const int m = (int)(idx / q), c = (int)(idx - (long)m * q) * 4;
    f32x4 a = *reinterpret_cast<const f32x4*>(slabs + (long)m * Np + c);
    for (int s = 1; s < n_slabs; ++s) a += *reinterpret_cast<const f32x4*>(slabs + s * slab_stride + (long)m * Np + c);
    if (bias) a += *reinterpret_cast<const f32x4*>(bias + c);
    f32x4 keep{1.f, 1.f, 1.f, 1.f};
    if (mask_f) keep = *reinterpret_cast<const f32x4*>(mask_f + mmap.off(m) + c);
    if (mask_b) {
      const bf16x4 kb = *reinterpret_cast<const bf16x4*>(mask_b + mmap.off(m) + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) keep[e] = (float)kb[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = relu ? fmaxf(a[e], 0.f) : a[e];
      a[e] = keep[e] > 0.f ? v : 0.f;
    }
    const long o = cmap.off(m) + c;
    if (out_f) *reinterpret_cast<f32x4*>(out_f + o) = a;
    if (out) {
      if constexpr (NP == 1) {
        *reinterpret_cast<bf16x4*>(out + o) = bf16x4{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3]};
      } else {
        bf16x4 ph, pm, pl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          __bf16 sh, sm, sl;
          split3(a[e], sh, sm, sl);
          ph[e] = sh; pm[e] = sm; pl[e] = sl;
        }
        *reinterpret_cast<bf16x4*>(out + o) = ph;
        *reinterpret_cast<bf16x4*>(out + c_plane + o) = pm;
        *reinterpret_cast<bf16x4*>(out + 2 * c_plane + o) = pl;
      }
    }
  }
}

// bias gradient from the reduction-major copy of dz: dbias[n] = sum_r dzt[n][r] (fp32 accumulation)
__global__ __launch_bounds__(256) void row_sum_bf16_kernel(const __bf16* __restrict__ dzt, long red, long pitch,
                                                           float* __restrict__ out) {
  __shared__ float part[256];
  const __bf16* row = dzt + (size_t)blockIdx.x * pitch;
  float acc = 0.f;
  for (long r = (long)threadIdx.x * 8; r < red; r += 256 * 8) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(row + r);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += (float)v[e];
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = part[0];
}

// Tile / pipeline choice (measured on config-2 shapes):
//   wide problems  -> 256 x 256 tile, 8 waves, LDS ring + ping-pong wave groups
//                     (NP = 3: 16-deep stages, ring of 3; NP = 1: 32-deep stages, ring of 4)
//   everything else -> 128 x 128 tile, 4 waves (NP = 3: 32-deep stages, 2 slots; NP = 1: 64-deep, ring of 4)
// st_set_tuning("bf16_tile", 128) forces the small tile (perf experiments).
template <int NP>
int launch_gemm(X6Params& p, hipStream_t s) {
  const int forced_tile = st::tuning(st::TUNE_BF16_TILE);
  if (p.splits < 1) p.splits = 1;
  const bool fits256 = NP == 1 ? p.Np >= 256 : p.Np % 256 == 0;
  const bool wide = (long)st::ceil_div(p.M, 256) * st::ceil_div(p.Np, 256) * p.splits >= 192;
  // The 250-channel layers (252 tiles of 128 x 128, ONE 128 KB workgroup per CU, 23-26 us for 14.4 GFLOP) were tried in round 3
  // as 64 x 64 tiles of two waves with 2-5 workgroups per CU (33-35 us) and in the one-wave-per-SIMD schedule (SCH = 1: 23 us
  // alone, no change of the step): the 128 x 128 tile of four waves stays.
  // (stacked per-bin products: a tile must lie inside one bin)
  const int BT = (forced_tile != 128 && fits256 && wide && (p.rows_per_bin <= 0 || p.rows_per_bin % 256 == 0)) ? 256 : 128;
  p.tiles_m = st::ceil_div(p.M, BT);
  p.tiles_n = st::ceil_div(p.Np, BT);
  {
    // Operand bytes the eight XCD-private L2s pull in together: the activation rows of an XCD's M range once per column
    // group, the filter panel of its N range once per row group (a convolution's taps re-read the same rows: unique bytes).
    // L9 at config 2 (16 032 x 2 016 x 2 048, bf16): one column panel per XCD reads the 65 MB of activations eight times
    // (PMC round 3: 345 MB per launch at 2.1 TB/s, L2 hit rate 0.89); row bands read them once and the 8 MB of filters
    // eight times.
    const double a_bytes = (double)p.M * (p.taps > 1 ? p.cp : p.Kvalid) * 2.0 * NP, b_bytes = (double)p.Kp * p.Np * 2.0 * NP;
    const int forced_gm = st::tuning(st::TUNE_XCD_GM);
    double best = 0.0;
    p.gm = 1;
    for (int gm = 1; gm <= 8; gm *= 2) {
      const int gn = 8 / gm;
      if (gm > p.tiles_m || gn > p.tiles_n) continue;
      const int slots = st::ceil_div(p.tiles_m, gm) * st::ceil_div(p.tiles_n, gn) * 8;
      const double cost = (a_bytes * gn + b_bytes * gm) * ((double)slots / (p.tiles_m * p.tiles_n));   // idle slots cost time
      if (best == 0.0 || cost < best || gm == forced_gm) { best = gm == forced_gm ? -1.0 : cost; p.gm = gm; }
      if (gm == forced_gm) break;
    }
    if (8 / p.gm > p.tiles_n) p.gm = 8;                       // fewer than 8/gm column panels: stack the XCDs along M
    if (p.gm > p.tiles_m && 8 / p.gm <= p.tiles_n) p.gm = 1;
    if (p.rows_per_bin > 0 && p.tiles_m >= 8) p.gm = 8;      // per-bin filter operands: an XCD's L2 should see few bins, each whole
    p.tm_per = st::ceil_div(p.tiles_m, p.gm);
    p.tn_per = st::ceil_div(p.tiles_n, 8 / p.gm);
    p.chunk = p.tm_per * p.tn_per;
  }
  const dim3 grid(p.chunk * 8 * p.splits);
  if (p.rows_per_bin > 0)
    st::trace("gemm_nn_bf16<%d,NP=%d> batched bins=%d M=%d Np=%d Kp=%d xcd=%dx%d gflop=%.3f", BT, NP, p.M / p.rows_per_bin, p.rows_per_bin,
              p.Np, p.Kvalid, p.gm, 8 / p.gm, 2e-9 * p.tiles_m * BT * (double)(p.tiles_n * BT) * p.Kvalid * (NP == 3 ? 6 : 1));
  else
    st::trace("gemm_nn_bf16<%d,NP=%d> splits=%d M=%d Np=%d Kp=%d taps=%d sched=%d xcd=%dx%d gflop=%.3f", BT, NP, p.splits, p.M, p.Np,
              p.Kp, p.taps, st::tuning(st::TUNE_BF16_SCHED), p.gm, 8 / p.gm,
              2e-9 * p.tiles_m * BT * (double)(p.tiles_n * BT) * p.Kp * (NP == 3 ? 6 : 1));
  st::LaunchTimer timer(s);
  const auto whole = [&](int bk) { return (p.taps > 1 ? p.cp % bk : p.Kvalid % bk) == 0; };   // FAST eligibility
#define ST_LAUNCH(T, W1, W2, K, N, R, P)                                                                       \
  do {                                                                                                         \
    if (whole(K)) st::launch_timed(timer, gemm_nn_bf16_kernel<T, W1, W2, K, N, R, P, true>, grid, dim3(64 * W1 * W2), s, p);   \
    else st::launch_timed(timer, gemm_nn_bf16_kernel<T, W1, W2, K, N, R, P, false>, grid, dim3(64 * W1 * W2), s, p);          \
  } while (0)
  // Schedule of the 256 x 256 bf16-activation kernel (st_set_tuning("bf16_sched", v) selects one; 0 = the policy = 1):
  //   1  eight waves in ping-pong groups, 32-deep stages, ring of 4 (rounds 1-2) -- still the fastest (round 3, L8 forward /
  //      back-prop / filter gradient: 0.462 / 0.419 / 0.537 ms)
  //   2 / 3  one wave per SIMD, software pipelined (SCH = 1), 32-deep stages, ring of 4 / 3: 0.489 / 0.419 / 0.605 ms
  //   4  SCH = 1 with 64-deep stages in a ring of 2 (a DMA row is a full 128-byte line: a third fewer L2 requests,
  //      PMC 78 M -> 55 M per launch): 0.464 / 0.417 / 0.572 ms; L9 (short last stage) 0.175 vs 0.138
  // None of the three moves the launch time: the limit is not instruction issue (scripts/ubench/gemm_issue.hip: the SCH = 1
  // stage with every operand byte coming from L2 runs at 1.93 PFLOP/s on constant data, 1.49 on random data -- the board's
  // power limit -- and 0.86 when every byte comes from HBM); see DESIGN 4.3.
  const int sched = st::tuning(st::TUNE_BF16_SCHED);
  const bool one_tap_tail = p.taps == 1 && p.Kvalid % 16 == 0;           // only the last stage may be short
  if constexpr (NP == 3) {
    if (BT == 256) ST_LAUNCH(256, 2, 4, 16, 3, 3, true);
    else ST_LAUNCH(128, 2, 2, 32, 3, 2, false);
  } else {
    if (BT == 256 && sched == 4 && (whole(64) || one_tap_tail)) {
      if (whole(64)) st::launch_timed(timer, gemm_nn_bf16_kernel<256, 2, 2, 64, 1, 2, false, true, 1>, grid, dim3(256), s, p);
      else st::launch_timed(timer, gemm_nn_bf16_kernel<256, 2, 2, 64, 1, 2, false, false, 1>, grid, dim3(256), s, p);
    } else if (BT == 256 && (sched == 2 || sched == 3) && whole(32)) {
      if (sched == 3) st::launch_timed(timer, gemm_nn_bf16_kernel<256, 2, 2, 32, 1, 3, false, true, 1>, grid, dim3(256), s, p);
      else st::launch_timed(timer, gemm_nn_bf16_kernel<256, 2, 2, 32, 1, 4, false, true, 1>, grid, dim3(256), s, p);
    }
    else if (BT == 256) ST_LAUNCH(256, 2, 4, 32, 1, 4, true);
    else ST_LAUNCH(128, 2, 2, 64, 1, 4, false);
  }
#undef ST_LAUNCH
  return st::check_launch("gemm_nn_bf16");
}

RowMapB map_of(const st_tensor3& t, int first_row, int frame_stride, int frames) {
  RowMapB m{};
  m.frames = frames;
  m.row_stride = frame_stride * t.c_pitch;
  m.batch_stride = (long)t.t_pitch * t.c_pitch;
  m.row0 = (long)first_row * t.c_pitch;
  return m;
}

// ---- shared bodies of the forward / back-prop-to-input / filter-gradient entry points ------------------
template <int NP>
int conv_fwd(const st_tensor3* x, const void* x_planes, const void* w_planes, const float* bias, int width, int stride,
             int pad_left, int relu, const st_tensor3* y, float* y_f32, void* y_planes, hipStream_t s, int splits = 1,
             float* slabs = nullptr) {
  X6Params p{};
  p.A = reinterpret_cast<const __bf16*>(x_planes);
  p.a_plane = (size_t)x->batch * x->t_pitch * x->c_pitch;
  p.amap = map_of(*x, x->halo - pad_left, stride, y->frames);
  p.Np = npad_of(y->channels);
  p.Kvalid = width * x->c_pitch;
  p.Kp = (int)st::round_up(p.Kvalid, 32);
  p.B = reinterpret_cast<const __bf16*>(w_planes);
  p.b_plane = (size_t)p.Np * p.Kp;
  p.C = y_f32;
  p.cmap = map_of(*y, y->halo, 1, y->frames);
  p.Cp = reinterpret_cast<__bf16*>(y_planes);
  p.c_plane = (size_t)y->batch * y->t_pitch * y->c_pitch;
  p.bias = bias;
  p.M = y->batch * y->frames;
  p.n_store = std::min(y->c_pitch, p.Np);
  p.relu = relu;
  p.taps = width;
  p.cp = x->c_pitch;
  if (splits <= 1) return launch_gemm<NP>(p, s);
  // few output rows: fp32 partial sums [split][M][Np], then one pass with bias + ReLU into the output tensor(s)
  const RowMapB out_map = p.cmap;
  __bf16* out = p.Cp;
  float* out_f = p.C;
  p.C = slabs; p.Cp = nullptr; p.bias = nullptr; p.relu = 0;
  p.cmap = RowMapB{};
  p.cmap.frames = p.M;
  p.cmap.row_stride = p.Np;
  p.splits = splits;
  p.slab_stride = (long)p.M * p.Np;
  if (int e = launch_gemm<NP>(p, s)) return e;
  const long work = (long)p.M * (p.n_store / 4);
  hipLaunchKernelGGL(slab_epilogue_kernel<NP>, dim3((unsigned)std::min<long>((work + 255) / 256, 4096)), dim3(256), 0, s,
                     slabs, splits, p.slab_stride, p.M, p.Np, p.n_store, bias, relu, (const float*)nullptr,
                     (const __bf16*)nullptr, RowMapB{}, out_map, out, p.c_plane, out_f);
  return st::check_launch("slab_epilogue");
}

// Forward pass on few output rows (single-utterance inference: 16 tiles on 256 CUs): slices of the reduction on
// the idle CUs.  Tap-inner convolutions split over 64-channel chunks (every slice walks all taps), 1x1 layers
// over slices of >= 4 stages.
int fwd_splits(const st_tensor3& x, const st_tensor3& y, int width) {
  const long tiles = (long)st::ceil_div(y.batch * y.frames, 128) * st::ceil_div(npad_of(y.channels), 128);
  if (tiles >= 128 || npad_of(y.channels) % 128) return 1;
  const int chunks = st::ceil_div(x.c_pitch, 64);
  const long slices = width > 1 ? chunks : chunks / 4;
  return (int)std::max(1L, std::min<long>(256 / tiles, slices));
}

// channel-chunk split of a tap-inner convolution whose tile grid cannot fill the chip (back-prop through L8:
// 252 tiles, 1000 stages each): 1 = no split
int bwd_data_splits(const st_tensor3& dz, const st_tensor3& dx, int width) {
  const long tiles = (long)st::ceil_div(dx.batch * dx.frames, 128) * st::ceil_div(npad_of(dx.channels), 128);
  const int chunks = st::ceil_div(dz.c_pitch, 64);
  if (tiles >= 384 || (long)chunks * width < 128) return 1;
  return (int)std::max(1L, std::min<long>(st::ceil_div(768, (int)tiles), chunks / 4));
}

template <int NP>
int conv_bwd_data(const st_tensor3* dz, const void* dz_planes, const void* wt_planes, int width, int pad_left,
                  const st_tensor3* act, const float* act_f32, const void* act_bf16, const st_tensor3* dx,
                  float* dx_f32, void* dx_planes, hipStream_t s, int splits = 1, float* slabs = nullptr) {
  const int lead = width - 1 - pad_left;
  X6Params p{};
  p.A = reinterpret_cast<const __bf16*>(dz_planes);
  p.a_plane = (size_t)dz->batch * dz->t_pitch * dz->c_pitch;
  p.amap = map_of(*dz, dz->halo - lead, 1, dx->frames);
  p.Np = npad_of(dx->channels);
  p.Kvalid = width * dz->c_pitch;
  p.Kp = (int)st::round_up(p.Kvalid, 32);
  p.B = reinterpret_cast<const __bf16*>(wt_planes);
  p.b_plane = (size_t)p.Np * p.Kp;
  p.C = dx_f32;
  p.cmap = map_of(*dx, dx->halo, 1, dx->frames);
  p.Cp = reinterpret_cast<__bf16*>(dx_planes);
  p.c_plane = (size_t)dx->batch * dx->t_pitch * dx->c_pitch;
  if (act) {
    p.mask = act_f32;
    p.mask_b = reinterpret_cast<const __bf16*>(act_bf16);
    p.mmap = map_of(*act, act->halo, 1, act->frames);
  }
  p.M = dx->batch * dx->frames;
  p.n_store = std::min(dx->c_pitch, p.Np);
  p.taps = width;
  p.cp = dz->c_pitch;
  if (splits <= 1) return launch_gemm<NP>(p, s);
  // split: fp32 partial sums [split][M][Np], then one pass that sums, masks and writes the output tensor(s)
  const RowMapB out_map = p.cmap, mask_map = p.mmap;
  const __bf16* mask_b = p.mask_b;
  const float* mask_f = p.mask;
  __bf16* out = p.Cp;
  float* out_f = p.C;
  p.C = slabs; p.Cp = nullptr; p.mask = nullptr; p.mask_b = nullptr;
  p.cmap = RowMapB{};
  p.cmap.frames = p.M;
  p.cmap.row_stride = p.Np;
  p.splits = splits;
  p.slab_stride = (long)p.M * p.Np;
  if (int e = launch_gemm<NP>(p, s)) return e;
  const long work = (long)p.M * (p.n_store / 4);
  hipLaunchKernelGGL(slab_epilogue_kernel<NP>, dim3((unsigned)std::min<long>((work + 255) / 256, 4096)), dim3(256), 0, s,
                     slabs, splits, p.slab_stride, p.M, p.Np, p.n_store, (const float*)nullptr, 0, mask_f, mask_b, mask_map,
                     out_map, out, p.c_plane, out_f);
  return st::check_launch("slab_epilogue");
}

// dF[(w,c)][n] = sum_r XT[c][r + shift(w)] * dZT[n][r], r = b * tq + t; `phases` = stride of the layer (1 or 2):
// tap w reads phase plane (w & (phases-1)) at shift w / phases
template <int NP>
int conv_bwd_filter(const void* xt_planes, size_t xt_plane, long xt_phase_stride, int phases, const void* dzt_planes,
                    int batch, int tq, int width, int cin_pitch, int x_first_row, int cout, float* dpacked, int splits,
                    float* slabs, hipStream_t s, long pitch = 0) {
  X6Params p{};
  const long red = (long)batch * tq;                   // reduction length
  if (pitch == 0) pitch = red;                         // elements between consecutive rows of the transposed operands
  p.A = reinterpret_cast<const __bf16*>(xt_planes);
  p.a_plane = xt_plane;
  p.amap.frames = cin_pitch;                           // output row k = w * cin_pitch + c
  p.amap.batch_stride = 1;                             // a tap shifts the window by one (phase) frame
  p.amap.row_stride = (int)pitch;                      // channel c selects the plane row
  p.amap.row0 = x_first_row;
  p.amap.phase_shift = phases == 2 ? 1 : 0;
  p.amap.phase_mask = phases - 1;
  p.amap.phase_stride = xt_phase_stride;
  p.Np = npad_of(cout);
  p.Kvalid = (int)red;
  p.Kp = (int)pitch;
  p.B = reinterpret_cast<const __bf16*>(dzt_planes);
  p.b_plane = (size_t)p.Np * pitch;
  p.M = width * cin_pitch;
  p.cmap.frames = p.M;
  p.cmap.row_stride = p.Np;
  p.n_store = p.Np;
  p.taps = 1;
  p.cp = cin_pitch;
  p.splits = splits;
  p.slab_stride = (long)p.M * p.Np;
  p.C = splits > 1 ? slabs : dpacked;
  if (int e = launch_gemm<NP>(p, s)) return e;
  if (splits > 1) {
    const size_t n4 = (size_t)p.M * p.Np / 4;
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 2048)), dim3(256), 0, s, slabs,
                       splits, n4, dpacked);
    return st::check_launch("slab_sum");
  }
  return ST_OK;
}

// geometry of the filter-gradient workspace of the bf16-activation path
struct WgradPlan {
  int tq, phases, splits, n_pad;
  long red, pitch;
  size_t xt_phase_elems, xt_bytes, dzt_bytes, slab_bytes;
};
WgradPlan wgrad_plan(const st_tensor3& x, const st_tensor3& dz, int width, int stride, int x_first_row) {
  WgradPlan w{};
  w.phases = stride;
  const int x_rows = st::ceil_div(x.t_pitch - x_first_row, stride);
  w.tq = (int)st::round_up(std::max(x_rows, dz.frames), 32);
  w.red = (long)x.batch * w.tq;
  w.n_pad = npad_of(dz.channels);
  w.pitch = w.red;      // a padded row pitch (64..520 elements, against power-of-two strides) measured 4-14 % slower
  w.xt_phase_elems = (size_t)x.c_pitch * w.pitch + 4096;         // slack: the last taps read past the last row
  w.xt_bytes = st::round_up(w.xt_phase_elems * stride * 2, 256);
  w.dzt_bytes = st::round_up((size_t)w.n_pad * w.pitch * 2, 256);
  const long M = (long)width * x.c_pitch;
  const long tiles = st::ceil_div((int)M, 128) * (long)st::ceil_div(w.n_pad, 128);
  const int stages = (int)((w.red + 63) / 64);
  const int forced = st::tuning(st::TUNE_BF16_WGRAD_SPLITS);
  // one 128 KB workgroup per CU: more than ~7/8 of the 256 CUs' worth of (tile, split) pairs starts a second round on some
  // XCD (a split's tiles are dealt to the XCDs in rectangles, 28 tiles -> 32 slots), and every split adds a slab to sum.
  // Round 3, filter gradient of a 250-channel layer incl. transposes and slab sum: 19 splits (the old "fill 512") 67 us,
  // 12 -> 60, 7 -> 51, 4 -> 65; first layer 8 -> 95, 4 -> 73; last layer 32 -> 55, 12 -> 46.
  const int target = st::tuning(st::TUNE_BF16_WGRAD_TARGET) > 0 ? st::tuning(st::TUNE_BF16_WGRAD_TARGET) : 224;
  w.splits = forced ? forced : tiles >= 192 ? 1 : (int)std::max(1L, std::min<long>(target / tiles, stages / 8));
  w.slab_bytes = w.splits > 1 ? (size_t)w.splits * M * w.n_pad * 4 : 0;
  return w;
}

// ---- stacked per-bin products of the frequency-domain layers (conv_fft.hip) on this kernel -------------------------
// C[b] = A[b] * Bt[b]^T for b < bins: A planes [bins][rows][lda] (k contiguous, rows a multiple of 128), Bt planes
// [bins][n][ldb] (k contiguous), C fp32 [bins][rows][ldc].  `a_rows_apart` / `a_bin`: element distance between consecutive
// rows of A and between the bins' first rows (a plain stack: lda and rows * lda; the transposed spectra of the lag products:
// row = channel, `pitch` apart, bins `rows_of_reduction` apart inside a row -- see st::transpose_bf16_bins).
template <int NP>
static int gemm_bins(const void* a_planes, size_t a_plane, long a_rows_apart, long a_bin, const void* bt_planes, size_t b_plane,
                     long ldb, long b_bin, float* c, long ldc, int rows, int k, int n, int bins, hipStream_t s, int b_bin_shift) {
  X6Params p{};
  p.A = reinterpret_cast<const __bf16*>(a_planes);
  p.a_plane = a_plane;
  p.amap.frames = rows;                       // m = bin * rows + r
  p.amap.row_stride = (int)a_rows_apart;
  p.amap.batch_stride = a_bin;
  p.B = reinterpret_cast<const __bf16*>(bt_planes);
  p.b_plane = b_plane;
  p.C = c;
  p.cmap.frames = rows * bins;                // plain stack of the output matrices
  p.cmap.row_stride = (int)ldc;
  p.M = rows * bins;
  p.Np = n;
  p.Kvalid = k;
  p.Kp = (int)ldb;
  p.n_store = n;
  p.taps = 1;
  p.cp = k;
  p.rows_per_bin = rows;
  p.b_bin = b_bin;
  p.b_bin_shift = b_bin_shift;
  return launch_gemm<NP>(p, s);
}

}  // namespace

int st::gemm_bf16_bins(int planes, const void* a_planes, size_t a_plane, long a_rows_apart, long a_bin, const void* bt_planes,
                       size_t b_plane, long ldb, long b_bin, float* c, long ldc, int rows, int k, int n, int bins, hipStream_t s,
                       int b_bin_shift) {
  if (!(a_planes && bt_planes && c && rows > 0 && rows % 128 == 0 && k > 0 && k % 32 == 0 && n > 0 && n % 128 == 0 && bins > 0 &&
        (planes == 1 || planes == 3) && b_bin_shift >= 0 && b_bin_shift <= 1 && a_rows_apart % 8 == 0 && a_bin % 8 == 0 && ldb % 8 == 0 && b_bin % 8 == 0 && ldc % 4 == 0)) {
    st::set_error("gemm_bf16_bins: bad shape rows=%d k=%d n=%d bins=%d planes=%d", rows, k, n, bins, planes);
    return ST_EINVAL;
  }
  return planes == 3 ? gemm_bins<3>(a_planes, a_plane, a_rows_apart, a_bin, bt_planes, b_plane, ldb, b_bin, c, ldc, rows, k, n, bins, s, b_bin_shift)
                     : gemm_bins<1>(a_planes, a_plane, a_rows_apart, a_bin, bt_planes, b_plane, ldb, b_bin, c, ldc, rows, k, n, bins, s, b_bin_shift);
}

// dst[c][bin * rows + r] = src[bin][r][c] (2-byte elements, one plane): the reduction-major copies of the spectra that the lag
// products of the filter gradient read -- row c of dst holds all bins behind each other, `bins * rows` elements per row
int st::transpose_bf16_bins(const void* src, void* dst, int bins, int rows, int cols, hipStream_t s) {
  if (!(src && dst && bins > 0 && rows > 0 && rows % 64 == 0 && cols > 0 && cols % 64 == 0)) {
    st::set_error("transpose_bf16_bins: bad shape bins=%d rows=%d cols=%d", bins, rows, cols);
    return ST_EINVAL;
  }
  TransposeJobs js{};
  js.n = 1;
  js.tq = rows;
  js.pitch = (long)bins * rows;
  js.job[0].src = reinterpret_cast<const unsigned short*>(src);
  js.job[0].dst = reinterpret_cast<unsigned short*>(dst);
  js.job[0].zero_tail = nullptr;
  js.job[0].rows = rows;
  js.job[0].row0 = 0;
  js.job[0].step = 1;
  js.job[0].t_pitch = rows;
  js.job[0].c_pitch = cols;
  js.job[0].c_rows = cols;
  js.job[0].y_tiles = cols / 64;
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3(rows / 64, cols / 64, bins), dim3(256), 0, s, js);
  return st::check_launch("transpose_bf16_bins");
}

// The same copies for the lag products in their split form (conv_fft.hip: Re Q = S2^T Z2, Im Q = S'2^T Z2 over the half-length
// row views of the spectra [re | im], cols = 2 * half): the reduction index of a bin runs over (part, r) -- real parts, then
// imaginary parts.  forms = 1 (the dz spectra):  dst[c][(bin * 2 + part) * rows + r] = src[bin][r][part * half + c];
// forms = 2 (the input spectra, both operands behind each other: batch 2 bin + j of the stacked product):
//   dst[c][((bin * 2 + j) * 2 + part) * rows + r] = j == 0 ? S[part]  :  (part == 0 ? S[1] : -S[0])      (S' = [S_i | -S_r]).
int st::transpose_bf16_bins_split(const void* src, void* dst, int bins, int rows, int cols, int forms, hipStream_t s) {
  if (!(src && dst && bins > 0 && rows > 0 && rows % 64 == 0 && cols > 0 && cols % 128 == 0 && (forms == 1 || forms == 2))) {
    st::set_error("transpose_bf16_bins_split: bad shape bins=%d rows=%d cols=%d forms=%d", bins, rows, cols, forms);
    return ST_EINVAL;
  }
  const int half = cols / 2;
  TransposeJobs js{};
  js.n = 2 * forms;
  js.tq = 2 * forms * rows;
  js.pitch = (long)bins * js.tq;
  for (int j = 0; j < forms; ++j)
    for (int part = 0; part < 2; ++part) {
      TransposeJob& jb = js.job[j * 2 + part];
      const int from = j == 0 ? part : 1 - part;            // which half of the source row
      jb.src = reinterpret_cast<const unsigned short*>(src) + from * half;
      jb.dst = reinterpret_cast<unsigned short*>(dst) + (size_t)(j * 2 + part) * rows;
      jb.zero_tail = nullptr;
      jb.rows = rows;
      jb.row0 = 0;
      jb.step = 1;
      jb.t_pitch = rows;
      jb.c_pitch = cols;                                     // (the source row pitch; only `half` columns of it are this job's)
      jb.c_rows = half;
      jb.y_tiles = half / 64;
      jb.flip = (j == 1 && part == 1) ? 0x8000 : 0;
    }
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3(rows / 64, js.n * (half / 64), bins), dim3(256), 0, s, js);
  return st::check_launch("transpose_bf16_bins_split");
}

namespace {
}  // namespace

extern "C" {

// ======== bf16x6 (experimental, opt-in) =================================================================
// planes: 3 * n bf16 elements (h | m | l)
int st_exp_split3_bf16(const float* src, size_t n, void* planes, void* stream) {
  ST_REQUIRE(src && planes && n % 4 == 0, "split3: bad args");
  const int blocks = (int)std::min<size_t>((n / 4 + 255) / 256, 4096);
  hipLaunchKernelGGL(split_kernel<3>, dim3(blocks), dim3(256), 0, st::as_stream(stream), src, n / 4,
                     reinterpret_cast<__bf16*>(planes));
  return st::check_launch("split3");
}

// packed [k_pad][n_pad] fp32 -> planes 3 x [n_pad][k_pad] bf16
int st_exp_split3_transpose_bf16(const float* packed, int k_pad, int n_pad, void* planes, void* stream) {
  ST_REQUIRE(packed && planes && k_pad % 32 == 0 && n_pad % 32 == 0, "split3_transpose: bad args");
  hipLaunchKernelGGL(split_transpose_kernel<3>, dim3(k_pad / 32, n_pad / 32), dim3(256), 0, st::as_stream(stream),
                     packed, k_pad, n_pad, reinterpret_cast<__bf16*>(planes));
  return st::check_launch("split3_transpose");
}

// forward conv on the bf16x6 path; x gives the geometry of the three activation planes;
// y_planes (nullable) receives the three planes of the output for the next layer
int st_exp_conv1d_fwd_bf16x6(const st_tensor3* x, const void* x_planes, const void* w_planes, const float* bias,
                             int width, int stride, int pad_left, int relu, const st_tensor3* y, void* y_planes,
                             void* stream) {
  ST_REQUIRE(x && y && x_planes && w_planes && y->base, "conv bf16x6: null argument");
  ST_REQUIRE(x->halo >= pad_left && y->frames == st::ceil_div(x->frames, stride) && x->c_pitch % 16 == 0,
             "conv bf16x6: bad geometry");
  ST_REQUIRE(npad_of(y->channels) % 128 == 0, "conv bf16x6: n_pad must be a multiple of 128");
  return conv_fwd<3>(x, x_planes, w_planes, bias, width, stride, pad_left, relu, y, y->base, y_planes,
                     st::as_stream(stream));
}

// back-prop to the layer input on the bf16x6 path (stride-1 layers): dz planes x planes of the
// flipped/transposed filter operand; act (nullable) is the ReLU mask source
size_t st_exp_conv1d_bwd_data_bf16x6_ws(const st_tensor3* dz, const st_tensor3* dx, int width) {
  if (!dz || !dx) return 0;
  const int splits = bwd_data_splits(*dz, *dx, width);
  return splits > 1 ? (size_t)splits * dx->batch * dx->frames * npad_of(dx->channels) * sizeof(float) : 0;
}

int st_exp_conv1d_bwd_data_bf16x6(const st_tensor3* dz, const void* dz_planes, const void* wt_planes, int width,
                                  int pad_left, const st_tensor3* act, const st_tensor3* dx, void* dx_planes,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  ST_REQUIRE(dz && dx && dz_planes && wt_planes && dx->base, "conv bwd bf16x6: null argument");
  const int lead = width - 1 - pad_left;
  ST_REQUIRE(lead >= 0 && dz->halo >= lead && dz->frames == dx->frames && dz->batch == dx->batch, "conv bwd bf16x6: bad geometry");
  ST_REQUIRE(npad_of(dx->channels) % 128 == 0, "conv bwd bf16x6: n_pad must be a multiple of 128");
  // few output tiles and a long reduction (back-prop through L8): split over channel chunks when a workspace is given
  int splits = bwd_data_splits(*dz, *dx, width);
  if (!workspace || workspace_bytes < st_exp_conv1d_bwd_data_bf16x6_ws(dz, dx, width)) splits = 1;
  return conv_bwd_data<3>(dz, dz_planes, wt_planes, width, pad_left, act, act ? act->base : nullptr, nullptr, dx,
                          dx->base, dx_planes, st::as_stream(stream), splits, reinterpret_cast<float*>(workspace));
}

// planes [c_rows][batch * tq] of a padded tensor, rows [row0, row0 + rows) of every utterance
int st_exp_transpose_split3_bf16(const st_tensor3* t, int row0, int rows, int tq, size_t plane_elems, void* planes,
                                 void* stream) {
  ST_REQUIRE(t && t->base && planes && rows > 0 && row0 >= 0 && row0 + rows <= t->t_pitch && tq >= rows &&
                 plane_elems >= (size_t)t->c_pitch * t->batch * tq, "transpose_split3: bad args");
  dim3 grid(st::ceil_div(rows, 32), st::ceil_div(t->c_pitch, 32), t->batch);
  hipLaunchKernelGGL(transpose_split_kernel<3>, grid, dim3(256), 0, st::as_stream(stream), t->base, rows, row0,
                     t->t_pitch, t->c_pitch, tq, plane_elems, reinterpret_cast<__bf16*>(planes));
  return st::check_launch("transpose_split3");
}

// filter gradient on the bf16x6 path (stride-1 layers): dF[(w,c)][n] = sum_r XT[c][r + w + lead] * dZT[n][r],
// r = b * tq + t.  xt_planes: [x.c_pitch][batch*tq (+ slack)], dzt_planes: [n_pad][batch*tq].
int st_exp_conv1d_bwd_filter_bf16x6(const void* xt_planes, const void* dzt_planes, int batch, int tq, int width,
                                    int cin_pitch, int x_first_row, int cout, float* dpacked, void* stream) {
  ST_REQUIRE(xt_planes && dzt_planes && dpacked && tq % 32 == 0 && cin_pitch % 16 == 0, "bwd_filter bf16x6: bad args");
  const long red = (long)batch * tq;
  ST_REQUIRE(npad_of(cout) % 128 == 0 && red < (1L << 31), "bwd_filter bf16x6: unsupported shape");
  return conv_bwd_filter<3>(xt_planes, (size_t)cin_pitch * red + 4096, 0, 1, dzt_planes, batch, tq, width, cin_pitch,
                            x_first_row, cout, dpacked, 1, nullptr, st::as_stream(stream));
}

// ======== bf16 activations (BASELINE config 4; declared in include/speecht_hip.h) ======================
int st_cast_bf16(const float* src, size_t n, void* dst, void* stream) {
  ST_REQUIRE(src && dst && n % 4 == 0, "st_cast_bf16: null argument or n not a multiple of 4");
  if (n == 0) return ST_OK;
  const int blocks = (int)std::min<size_t>((n / 4 + 255) / 256, 4096);
  hipLaunchKernelGGL(split_kernel<1>, dim3(blocks), dim3(256), 0, st::as_stream(stream), src, n / 4,
                     reinterpret_cast<__bf16*>(dst));
  return st::check_launch("cast_bf16");
}

int st_filters_bf16(const float* packed, int k_pad, int n_pad, void* wt, void* stream) {
  ST_REQUIRE(packed && wt && k_pad > 0 && n_pad > 0 && k_pad % 32 == 0 && n_pad % 32 == 0, "st_filters_bf16: bad args");
  hipLaunchKernelGGL(split_transpose_kernel<1>, dim3(k_pad / 32, n_pad / 32), dim3(256), 0, st::as_stream(stream),
                     packed, k_pad, n_pad, reinterpret_cast<__bf16*>(wt));
  return st::check_launch("filters_bf16");
}

int st_filters_bwd_bf16(const float* packed, int width, int cin, int cout, int cin_pitch, int cout_pitch, void* wtt,
                        void* stream) {
  ST_REQUIRE(packed && wtt && width >= 1 && cin >= 1 && cout >= 1 && cin_pitch >= cin && cout_pitch >= cout &&
                 cout_pitch % 16 == 0 && cin_pitch % 16 == 0,
             "st_filters_bwd_bf16: bad args");
  const int n_pad = npad_of(cout);                                   // column pitch of the packed filters
  const int kt_pad = (int)st::round_up((size_t)width * cout_pitch, 32);
  const long total = (long)width * cin * (cout_pitch / 4);
  hipLaunchKernelGGL(filters_bwd_bf16_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0,
                     st::as_stream(stream), packed, width, cin, cin_pitch, cout_pitch, n_pad, kt_pad,
                     reinterpret_cast<__bf16*>(wtt));
  return st::check_launch("filters_bwd_bf16");
}

size_t st_conv1d_fwd_bf16_ws(const st_tensor3* x, const st_tensor3* y, int width) {
  if (!x || !y) return 0;
  const int splits = fwd_splits(*x, *y, width);
  return splits > 1 ? (size_t)splits * y->batch * y->frames * npad_of(y->channels) * sizeof(float) : 0;
}

int st_conv1d_nwc_fwd_ws_bf16(const st_tensor3* x, const void* x_bf16, const void* wt_bf16, const float* bias, int width,
                              int stride, int pad_left, int relu, const st_tensor3* y, void* y_bf16, float* y_f32,
                              void* workspace, size_t workspace_bytes, void* stream) {
  ST_REQUIRE(x && y && x_bf16 && wt_bf16 && (y_bf16 || y_f32), "conv fwd bf16: null argument");
  ST_REQUIRE(width >= 1 && stride >= 1 && x->halo >= pad_left && y->frames == st::ceil_div(x->frames, stride) &&
                 x->batch == y->batch && x->c_pitch % 16 == 0 && y->c_pitch % 16 == 0,
             "conv fwd bf16: bad geometry");
  ST_REQUIRE(x->halo - pad_left + (y->frames - 1) * stride + width <= x->t_pitch, "conv fwd bf16: right halo too small");
  if (stride == 1 && y_bf16 && !y_f32 && st::conv_taps_bf16_eligible(*x, *y, width, pad_left, nullptr))
    return st::conv_taps_bf16(*x, x_bf16, wt_bf16, bias, width, pad_left, relu, nullptr, nullptr, *y, y_bf16, st::as_stream(stream));
  int splits = fwd_splits(*x, *y, width);
  if (!workspace || workspace_bytes < st_conv1d_fwd_bf16_ws(x, y, width)) splits = 1;
  return conv_fwd<1>(x, x_bf16, wt_bf16, bias, width, stride, pad_left, relu, y, y_f32, y_bf16, st::as_stream(stream),
                     splits, reinterpret_cast<float*>(workspace));
}

int st_conv1d_nwc_fwd_bf16(const st_tensor3* x, const void* x_bf16, const void* wt_bf16, const float* bias, int width,
                           int stride, int pad_left, int relu, const st_tensor3* y, void* y_bf16, float* y_f32,
                           void* stream) {
  return st_conv1d_nwc_fwd_ws_bf16(x, x_bf16, wt_bf16, bias, width, stride, pad_left, relu, y, y_bf16, y_f32, nullptr, 0,
                                   stream);
}

size_t st_conv1d_bwd_data_bf16_ws(const st_tensor3* dz, const st_tensor3* dx, int width) {
  if (!dz || !dx) return 0;
  const int splits = bwd_data_splits(*dz, *dx, width);
  return splits > 1 ? (size_t)splits * dx->batch * dx->frames * npad_of(dx->channels) * sizeof(float) : 0;
}

int st_conv1d_nwc_bwd_data_bf16(const st_tensor3* dz, const void* dz_bf16, const void* wtt_bf16, int width, int pad_left,
                                const st_tensor3* act, const void* act_bf16, const st_tensor3* dx, void* dx_bf16,
                                void* workspace, size_t workspace_bytes, void* stream) {
  ST_REQUIRE(dz && dx && dz_bf16 && wtt_bf16 && dx_bf16 && (!act || act_bf16), "conv bwd-data bf16: null argument");
  const int lead = width - 1 - pad_left;
  ST_REQUIRE(lead >= 0 && dz->halo >= lead && dz->frames == dx->frames && dz->batch == dx->batch &&
                 dz->t_pitch >= dz->halo - lead + dz->frames + width - 1 && dz->c_pitch % 16 == 0,
             "conv bwd-data bf16: bad geometry (stride-1 layers only)");
  ST_REQUIRE(!act || (act->frames == dx->frames && act->batch == dx->batch && act->c_pitch >= dx->c_pitch),
             "conv bwd-data bf16: mask geometry");
  if (st::conv_taps_bf16_eligible(*dz, *dx, width, lead, act))
    return st::conv_taps_bf16(*dz, dz_bf16, wtt_bf16, nullptr, width, lead, 0, act, act_bf16, *dx, dx_bf16, st::as_stream(stream));
  const int splits = bwd_data_splits(*dz, *dx, width);
  const size_t need = st_conv1d_bwd_data_bf16_ws(dz, dx, width);
  if (need && (!workspace || workspace_bytes < need)) {
    st::set_error("conv bwd-data bf16: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
    return ST_EWORKSPACE;
  }
  return conv_bwd_data<1>(dz, dz_bf16, wtt_bf16, width, pad_left, act, nullptr, act_bf16, dx, nullptr, dx_bf16,
                          st::as_stream(stream), splits, reinterpret_cast<float*>(workspace));
}

size_t st_conv1d_bwd_filter_bf16_ws(const st_tensor3* x, const st_tensor3* dz, int width, int stride, int pad_left) {
  if (!x || !dz || stride < 1 || stride > 2 || x->halo < pad_left) return 0;
  const WgradPlan w = wgrad_plan(*x, *dz, width, stride, x->halo - pad_left);
  return w.xt_bytes + w.dzt_bytes + w.slab_bytes;
}

int st_conv1d_nwc_bwd_filter_bf16(const st_tensor3* x, const void* x_bf16, const st_tensor3* dz, const void* dz_bf16,
                                  int width, int stride, int pad_left, float* dpacked, float* dbias, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  ST_REQUIRE(x && dz && x_bf16 && dz_bf16 && dpacked && dbias, "conv bwd-filter bf16: null argument");
  ST_REQUIRE((stride == 1 || stride == 2) && x->halo >= pad_left && x->batch == dz->batch &&
                 dz->frames == st::ceil_div(x->frames, stride) && x->c_pitch % 16 == 0,
             "conv bwd-filter bf16: bad geometry (stride 1 or 2)");
  const int first = x->halo - pad_left;
  const WgradPlan w = wgrad_plan(*x, *dz, width, stride, first);
  ST_REQUIRE(w.red < (1L << 31), "conv bwd-filter bf16: reduction too long");
  const size_t need = w.xt_bytes + w.dzt_bytes + w.slab_bytes;
  if (!workspace || workspace_bytes < need) {
    st::set_error("conv bwd-filter bf16: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
    return ST_EWORKSPACE;
  }
  hipStream_t s = st::as_stream(stream);
  unsigned short* xt = reinterpret_cast<unsigned short*>(workspace);
  unsigned short* dzt = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(workspace) + w.xt_bytes);
  float* slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + w.xt_bytes + w.dzt_bytes);
  // reduction-major copies (every element of [c][b*tq + j] is written: zeros where the source has no row), one launch
  TransposeJobs js{};
  js.tq = w.tq;
  js.pitch = w.pitch;
  int y_tiles = 0;
  for (int ph = 0; ph < stride; ++ph) {
    TransposeJob& j = js.job[js.n++];
    j.src = reinterpret_cast<const unsigned short*>(x_bf16);
    j.dst = xt + ph * w.xt_phase_elems;
    j.zero_tail = j.dst + (size_t)x->c_pitch * w.pitch;
    j.rows = st::ceil_div(x->t_pitch - first - ph, stride);
    j.row0 = first + ph; j.step = stride; j.t_pitch = x->t_pitch; j.c_pitch = x->c_pitch; j.c_rows = x->c_pitch;
    j.y_tiles = st::ceil_div(x->c_pitch, 64);
    y_tiles += j.y_tiles;
  }
  {
    TransposeJob& j = js.job[js.n++];
    j.src = reinterpret_cast<const unsigned short*>(dz_bf16);
    j.dst = dzt;
    j.rows = dz->frames; j.row0 = dz->halo; j.step = 1; j.t_pitch = dz->t_pitch; j.c_pitch = dz->c_pitch; j.c_rows = w.n_pad;
    j.y_tiles = st::ceil_div(w.n_pad, 64);
    y_tiles += j.y_tiles;
  }
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3(st::ceil_div(w.tq, 64), y_tiles, x->batch), dim3(256), 0, s, js);
  if (int e = st::check_launch("transpose_bf16")) return e;
  if (int e = conv_bwd_filter<1>(xt, 0, (long)w.xt_phase_elems, stride, dzt, x->batch, w.tq, width, x->c_pitch, 0,
                                 dz->channels, dpacked, w.splits, slabs, s, w.pitch))
    return e;
  hipLaunchKernelGGL(row_sum_bf16_kernel, dim3(w.n_pad), dim3(256), 0, s, reinterpret_cast<const __bf16*>(dzt), w.red, w.pitch,
                     dbias);
  return st::check_launch("row_sum_bf16");
}

}  // extern "C"
