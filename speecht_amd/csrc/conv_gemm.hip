// Wav2Letter convolution stack on gfx950: implicit-GEMM conv1d (NWC, SAME) forward,
// back-prop to the input, and back-prop to the filters, all on the exact-f32 MFMA
// (v_mfma_f32_32x32x2_f32, 157 TF peak = the f32 vector peak, bitwise an fmaf chain).
//
// Replaces the TF call sites speech_model.py:155 (tf.nn.conv1d), :173 (bias_add), :177 (relu)
// and their gradients (optimizer.compute_gradients, speech_model.py:78).
//
// Layout idea: with activations stored (batch, time, channel) and zero halo
// rows around each utterance, the im2col row of output frame t is the contiguous span
// x[b, t*stride - pad_left .. +W) -- W*c_pitch floats -- and filters[W][Cin][Cout] reshaped
// to [W*c_pitch][n_pad] is already the row-major B operand.  So conv == GEMM whose A rows
// overlap in memory; nothing is ever materialised.  Back-prop to the input is the same kernel
// run over dz with the flipped/transposed filter operand; back-prop to the filters is
// A^T * dz reduced over all (b, t) rows, split over row ranges into slabs that a second
// kernel sums (deterministic, no atomics).  Staging is LDS-DMA everywhere (global_load_lds).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>

#include <type_traits>

#include "st_common.h"
#include "streamk_map.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;            // reduction depth of one LDS stage
constexpr int APITCH = BK + 4;    // 36 dwords: ds_read_b128 of 16 rows hits 64 distinct banks
constexpr int NTHREADS = 256;     // 4 waves, one per SIMD

struct RowMap {   // flat output row m = b * frames + t  ->  float offset of that row
  int frames;
  int row_stride;
  long batch_stride;
  long row0;
  __device__ __forceinline__ long off(int m) const {
    int b = m / frames;
    int t = m - b * frames;
    return (long)b * batch_stride + row0 + (long)t * row_stride;
  }
};

struct NNParams {
  const float* A;  RowMap amap;      // implicit im2col operand
  const float* Bm; int Np;           // packed [Kp][Np]
  float* C;        RowMap cmap;
  const float* mask; RowMap mmap;    // EPI 1: relu mask source (may be null)
  const float* bias;                 // EPI 0
  int M, Kvalid, Kp, n_store, relu;
  int taps, cp;                      // filter width and channel pitch of A (k = tap * cp + channel)
  int tiles_m, tiles_n, chunk;       // XCD-aware tile order: 8 XCDs as a gm x gn grid over the tile grid,
  int gm, tm_per, tn_per;            // each XCD owns tm_per x tn_per tiles (chunk = tm_per * tn_per)
  int splits, steps_per_split;       // split-K over blockIdx.y: raw partial tiles go to `slab`
  float* slab;                       // [splits][M][Np]
  float* colsum;                     // EPI 1, optional: column sums of the stored tile rows, one row of Np floats per
  int colsum_rows;                   // (tile_m, wave row): [colsum_rows][Np] -- the bias gradient of the layer below
  int batches;                       // > 0: `batches` independent GEMMs of the same shape (the frequency bins of
  long a_batch, b_batch, c_batch;    // csrc/conv_fft.hip), operand strides in floats; workgroup -> (bin, tile) below
  long bt_ld; int bt_rows;           // BT kernels: Bm is B TRANSPOSED, [bt_rows][bt_ld] with the reduction index contiguous
};

// ------------------------------------------------------------------------------------
// C[M, n_store] = epilogue(A[M, Kp] * B[Kp, Np]).  256 threads = WMW x WNW waves, each wave
// owns a (BM/WMW) x (BN/WNW) block of 32x32 MFMA tiles.
//
// Staging is LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write instructions, and
// the copy of k-tile kt+1 is in flight under all 64 MFMAs of tile kt (one barrier per tile, which
// also drains the DMA).  DMA destinations are lane-linear, so the bank-conflict fix for the
// im2col operand is an XOR swizzle applied on the SOURCE address: LDS row m, 16-byte slot p holds
// source slot p ^ ((m>>1)&7); the fragment ds_read_b128 applies the same XOR.  The filter operand
// stays linear [k][n]; a lane fetches NT adjacent columns with one read, so n-tile nt of a wave
// covers columns {NT*lane + nt} -- the epilogue stores NT adjacent floats per lane.
// All LDS lives in ONE array (a second __shared__ object makes hipcc drain the DMA queue before
// every fragment read).
// ------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA with scalar base + 32-bit per-lane byte offset (no VALU address math per piece).
// hipcc does not model the M0 write or the outstanding load: the issuing wave drains with
// dma_wait_all() before the barrier that publishes the data.
__device__ __forceinline__ void dma16_sv(unsigned lds_byte_addr, const void* sbase, unsigned voff_bytes) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_byte_addr), "v"(voff_bytes), "s"(sbase) : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const float*)p;
}

template <int N> struct FVec;
template <> struct FVec<1> { typedef float type; };
template <> struct FVec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct FVec<4> { typedef float type __attribute__((ext_vector_type(4))); };
template <int N> __device__ __forceinline__ float vget(const typename FVec<N>::type& v, int i) { return v[i]; }
template <> __device__ __forceinline__ float vget<1>(const float& v, int) { return v; }
template <int N> __device__ __forceinline__ void vset(typename FVec<N>::type& v, int i, float x) { v[i] = x; }
template <> __device__ __forceinline__ void vset<1>(float& v, int, float x) { v = x; }

// FAST: every k-tile is whole (channel pitch and reduction length multiples of 32), so the DMA source of a
// piece is a per-lane pointer plus the uniform k0 -- no clamps, one 64-bit add per piece.
// BT (needs FAST, one tap): the filter operand is given TRANSPOSED, Bt[n][k] with k contiguous -- back-prop to the input of a
// 1-tap layer reads the layer's own packed filters [cin][cout] (dx = dz W^T), back-prop of a frequency-domain layer the
// FORWARD filter spectra (gbwd = gfwd^T, conv_fft.hip): no flipped / transposed copies, no second set of spectra.  The
// B stage is then staged and read exactly like the A stage ([BN][32], XOR swizzle on the source, ds_read_b128 fragments of 4
// consecutive k); a lane's NT columns are 32 apart (n * 32 + lane) instead of adjacent, so that the 32 fragment rows of a
// read are consecutive LDS rows (conflict-free like A's), and the epilogue addresses its columns one by one.
template <int BM, int BN, int WMW, int WNW, int EPI, bool FAST = false, bool BT = false>
__global__ __launch_bounds__(NTHREADS) void gemm_nn_kernel(NNParams p) {
  constexpr bool SGB = true;
  static_assert(!BT || FAST, "the transposed-operand variant exists for whole k-tiles only");
  constexpr int WTM = BM / WMW, WTN = BN / WNW;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int A_DMA = BM / 32;            // DMA instructions per wave for the [BM][32] A stage
  constexpr int B_DMA = BN / 32;            // ... for the [32][BN] B stage
  constexpr int B_LPR = BN / 4;             // lanes per B row
  constexpr int A_SZ = BM * BK, B_SZ = BK * BN;
  static_assert(WMW * WNW == 4 && MT >= 1 && (NT == 1 || NT == 2 || NT == 4) && A_DMA >= 1 && B_DMA >= 1, "tile config");
  typedef typename FVec<NT>::type bvec;

  __shared__ __attribute__((aligned(16))) float smem[2 * A_SZ + 2 * B_SZ + 6 * BM];
  float* const As = smem;
  float* const Bs = smem + 2 * A_SZ;
  long* const a_off = reinterpret_cast<long*>(smem + 2 * A_SZ + 2 * B_SZ);
  long* const c_off = a_off + BM;
  long* const m_off = c_off + BM;

  // XCD-aware order: block b runs on XCD b%8, and every XCD has its own L2.  The 8 XCDs are laid over the
  // tile grid as gm x gn rectangles (chosen by the host to minimise  A-bytes * gn + B-bytes * gm, i.e. how
  // often each operand is fetched into some L2); inside its rectangle an XCD walks panel-major so that
  // the CUs sharing the L2 stream the same filter panel together.
  const int bid = blockIdx.x;
  const int xcd = bid & 7, local = bid >> 3;
  int tile_m, tile_n;
  const float* __restrict__ Abase = p.A;
  const float* __restrict__ Bbase = p.Bm;
  float* __restrict__ Cbase = p.C;
  if (p.batches > 0) {
    // Batched mode: bin = 8 * set + xcd -- all tiles of one bin run on ONE XCD, whose L2 then holds that bin's
    // operands (a bin's filter matrix is 8 MB: spread over the XCDs every L2 would stream all of them).
    // Row tiles fastest, so the workgroups sharing a filter panel sit next to each other.
    const int per_bin = p.tiles_m * p.tiles_n;
    const int set = local / per_bin, t = local - set * per_bin;
    const int bin = set * 8 + xcd;
    if (bin >= p.batches) return;
    tile_n = t / p.tiles_m;
    tile_m = t - tile_n * p.tiles_m;
    Abase += (long)bin * p.a_batch;
    Bbase += (long)bin * p.b_batch;
    Cbase += (long)bin * p.c_batch;
  } else {
    const int xm = xcd % p.gm, xn = xcd / p.gm;
    const int ln = local / p.tm_per, lm = local - ln * p.tm_per;
    tile_m = xm * p.tm_per + lm;
    tile_n = xn * p.tn_per + ln;
    if (local >= p.chunk || tile_m >= p.tiles_m || tile_n >= p.tiles_n) return;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave / WNW, wn = wave % WNW;

  if (tid < BM) {
    int m = m0 + tid;
    bool valid = m < p.M;
    int mm = valid ? m : p.M - 1;
    a_off[tid] = p.amap.off(mm);
    c_off[tid] = valid ? p.cmap.off(mm) : -1;
    if (EPI == 1) m_off[tid] = p.mask ? p.mmap.off(mm) : 0;
  }
  __syncthreads();

  // DMA sources.  A: instruction i of this wave covers rows (wave*A_DMA + i)*8 .. +8, lane -> (row,
  // physical slot); B: instruction i covers 64/B_LPR rows of k.
  const float* asrc[A_DMA];
  int aslot4[A_DMA];
#pragma unroll
  for (int i = 0; i < A_DMA; ++i) {
    const int row = (wave * A_DMA + i) * 8 + (lane >> 3);
    asrc[i] = Abase + a_off[row];
    aslot4[i] = (((lane & 7) ^ ((row >> 1) & 7))) * 4;
  }
  const float* bsrc[B_DMA];
#pragma unroll
  for (int i = 0; i < B_DMA; ++i) {
    bsrc[i] = Bbase + n0 + (lane % B_LPR) * 4;
  }
  const int ktail = p.Kvalid - 4;   // last float4 inside the valid reduction range

  const int kplast = p.Kp - 1;
  constexpr int N_DMA = A_DMA + B_DMA;
  // DMA piece `pc` (compile-time) of the tile starting at reduction index k0: pieces [0, A_DMA) are
  // 8-row groups of the im2col operand, the rest 1-KiB groups of filter rows.
  // (The scalar-base inline-asm DMA of the filter-gradient kernel measured SLOWER here, 123 vs 128
  // TF/s: the per-piece address math is already only a min + 64-bit add, and the asm statements
  // block hipcc's own interleave.)
  const float* aptr[A_DMA];
  const float* bptr[B_DMA];
#pragma unroll
  for (int i = 0; i < A_DMA; ++i) aptr[i] = asrc[i] + aslot4[i];
#pragma unroll
  for (int i = 0; i < B_DMA; ++i) {
    if (BT) {            // rows of Bt like rows of A: instruction i covers tile rows (wave * B_DMA + i) * 8 .. +8 (clamped into Bt)
      const int row = (wave * B_DMA + i) * 8 + (lane >> 3);
      bptr[i] = Bbase + (long)min(n0 + row, p.bt_rows - 1) * p.bt_ld + (((lane & 7) ^ ((row >> 1) & 7))) * 4;
    } else {
      bptr[i] = bsrc[i] + (long)((wave * B_DMA + i) * (64 / B_LPR) + lane / B_LPR) * p.Np;
    }
  }
  auto dma_piece = [&](int pc, int k0, int buf) {
    if (pc < A_DMA) {
      const int i = pc < A_DMA ? pc : 0;
      // reduction tail: clamp so the read stays inside the row span (values there are unused)
      const float* g = FAST ? aptr[i] + k0 : asrc[i] + min(k0 + aslot4[i], ktail);
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(As + buf * A_SZ + (wave * A_DMA + pc) * 256), 16, 0, 0);
    } else {
      const int i = pc - A_DMA < B_DMA ? pc - A_DMA : 0;
      const int krow = min(k0 + (wave * B_DMA + i) * (64 / B_LPR) + lane / B_LPR, kplast);
      const float* g = BT ? bptr[i] + k0 : (FAST ? bptr[i] + (long)k0 * p.Np : bsrc[i] + (long)krow * p.Np);
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Bs + buf * B_SZ + (wave * B_DMA + i) * 256), 16, 0, 0);
    }
  };
  auto dma = [&](int k0, int buf) {
#pragma unroll
    for (int pc = 0; pc < N_DMA; ++pc) dma_piece(pc, k0, buf);
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses: A row = wm*WTM + mt*32 + l31, logical slot 2q+h, XOR-swizzled
  int a_frag[4];
  {
    const int row = wm * WTM + l31;
    const int sw = (row >> 1) & 7;
#pragma unroll
    for (int q = 0; q < 4; ++q) a_frag[q] = row * BK + (((2 * q + h) ^ sw) * 4);
  }
  const int b_frag = (4 * h) * BN + wn * WTN + NT * l31;
  int bt_frag[4];        // BT: fragment of column tile n = bt_frag[q] + n * 32 * BK (row = wn * WTN + n * 32 + l31; same swizzle for every n)
  {
    const int row = wn * WTN + l31;
    const int sw = (row >> 1) & 7;
#pragma unroll
    for (int q = 0; q < 4; ++q) bt_frag[q] = row * BK + (((2 * q + h) ^ sw) * 4);
  }

  // Reduction order.  k = tap * cp + channel.  With more than one tap the k-tiles are walked
  // channel-chunk OUTER, tap INNER: consecutive tiles then read the same 32-channel column of input
  // rows shifted by one frame, so 127 of the 128 staged rows were fetched by the previous tile and
  // the DMA hits L1/L2 instead of re-streaming the receptive window from HBM/MALL for every tap.
  // A chunk (or the flat tail tile) holding only 16 valid channels runs half the MFMA quads.
  const bool tap_inner = p.taps > 1;
  const int chunks = (p.cp + BK - 1) / BK;
  const int nk_total = tap_inner ? chunks * p.taps : p.Kp / BK;
  // split-K: this workgroup reduces k-tiles [s0, s0 + nk) and writes a raw partial tile
  const int s0 = p.splits > 1 ? blockIdx.y * p.steps_per_split : 0;
  const int nk = p.splits > 1 ? min(p.steps_per_split, nk_total - s0) : nk_total;
  int tap = tap_inner ? s0 % p.taps : 0;          // position of the tile being COMPUTED
  int chunk = tap_inner ? s0 / p.taps : s0;
  auto tile_k0 = [&](int t, int c) { return tap_inner ? t * p.cp + c * BK : c * BK; };
  auto tile_nq = [&](int t, int c) {
    const int valid = tap_inner ? p.cp - c * BK : p.Kvalid - c * BK;
    return valid >= BK ? 4 : 2;
  };
  dma(tile_k0(tap, chunk), 0);
  __syncthreads();

  // One k-tile.  CUR (LDS slot) and MORE (is there a next tile to stage) are compile-time: in the FAST
  // variant the steady-state loop below is then one basic block per pair of tiles -- no uniform branches
  // cutting hipcc's MFMA / ds_read / DMA interleave, LDS addresses folded into instruction offsets.
  auto stage = [&](auto cur_c, auto more_c) {
    constexpr int CUR = decltype(cur_c)::value;
    constexpr bool MORE = decltype(more_c)::value;
    const int nq = FAST ? 4 : tile_nq(tap, chunk);
    // next tile
    int ntap = tap, nchunk = chunk;
    if (tap_inner) { if (++ntap == p.taps) { ntap = 0; ++nchunk; } } else { ++nchunk; }
    const int nk0 = tile_k0(ntap, nchunk);
    tap = ntap; chunk = nchunk;
    const float* as = As + CUR * A_SZ;
    const float* bs = Bs + CUR * B_SZ + (BT ? 0 : b_frag);
    // Software-pipelined fragment reads: the reads of k-quad q+1 are issued BEFORE the 16 MFMAs of
    // quad q (sched_barrier pins the order; hipcc otherwise sinks the reads behind the MFMAs to
    // save registers and then stalls on LDS latency four times per tile).
    f32x4 af[4][MT];
    bvec bf[4][4];
    f32x4 bft[4][NT];
    auto read_frags = [&](int q) {
#pragma unroll
      for (int i = 0; i < MT; ++i) af[q][i] = *reinterpret_cast<const f32x4*>(as + a_frag[q] + i * 32 * BK);
      if (BT) {
#pragma unroll
        for (int n = 0; n < NT; ++n) bft[q][n] = *reinterpret_cast<const f32x4*>(bs + bt_frag[q] + n * 32 * BK);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[q][j] = *reinterpret_cast<const bvec*>(bs + (8 * q + j) * BN);
      }
    };
    read_frags(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // The DMA of the next tile is issued in slices between the MFMA quads: a burst of 8 DMA
      // instructions stalls the wave's issue long enough to drain the matrix pipe, two (~60 cycles
      // each) hide in the shadow of the MFMAs in flight.  (One per 4 MFMAs with more sched_barriers
      // measured slower: the pinning then also blocks hipcc's own ds_read/MFMA interleave.)
      if (MORE) {
#pragma unroll
        for (int pc = q; pc < N_DMA; pc += 4) dma_piece(pc, nk0, CUR ^ 1);
      }
      if (q < 3) read_frags(q + 1);
      if (q < nq) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q][i][j], BT ? bft[q][n][j] : vget<NT>(bf[q][j], n), acc[i][n], 0, 0, 0);
      }
      if (SGB) {
        // interleave request: one ds_read behind each of the first MFMAs, then the address VALU and
        // one DMA behind later MFMAs (each MFMA shadows ~64 issue cycles)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x006, 8, 0);
          __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                         // also drains this wave's DMA (vmcnt) before anyone reads it
  };
  using std::integral_constant;
  using std::true_type;
  using std::false_type;
  int kt = 0;
  for (; kt + 2 < nk; kt += 2) {             // steady state: both tiles have a successor
    stage(integral_constant<int, 0>{}, true_type{});
    stage(integral_constant<int, 1>{}, true_type{});
  }
  if (kt + 2 == nk) {
    stage(integral_constant<int, 0>{}, true_type{});
    stage(integral_constant<int, 1>{}, false_type{});
  } else if (kt + 1 == nk) {
    stage(integral_constant<int, 0>{}, false_type{});
  }

  if constexpr (BT) {
    // epilogue of the transposed-operand variant: column tile n of this lane is column n * 32 + l31 of the wave's block
    // (one float per lane and row: 128-byte runs per half-wave); same arithmetic as below, column by column
    const int colb = n0 + wn * WTN + l31;
    if (p.splits > 1) {
      float* slab = p.slab + (long)blockIdx.y * p.M * p.Np;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m < p.M) {
#pragma unroll
            for (int n = 0; n < NT; ++n) slab[(long)m * p.Np + colb + n * 32] = acc[i][n][r];
          }
        }
      return;
    }
    float csum[NT], bv[NT];
    bool ok[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      ok[n] = colb + n * 32 < p.n_store;
      csum[n] = 0.f;
      bv[n] = (EPI == 0 && p.bias && ok[n]) ? p.bias[colb + n * 32] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float mk[16][NT];
      if (EPI == 1 && p.mask) {                    // the ReLU mask of the 16 rows first, branch-free (see below)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
          for (int n = 0; n < NT; ++n) mk[r][n] = p.mask[m_off[row] + (ok[n] ? colb + n * 32 : 0)];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const long co = c_off[row];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          float v = acc[i][n][r];
          if (EPI == 0) {
            v += bv[n];
            if (p.relu) v = fmaxf(v, 0.f);
          } else if (p.mask) {
            v = mk[r][n] > 0.f ? v : 0.f;
          }
          if (co >= 0 && ok[n]) {
            if (EPI == 1) csum[n] += v;
            Cbase[co + colb + n * 32] = v;
          }
        }
      }
    }
    if (EPI == 1 && p.colsum) {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float tot = csum[n] + __shfl_xor(csum[n], 32, 64);
        if (h == 0) p.colsum[(long)(tile_m * WMW + wm) * p.Np + colb + n * 32] = tot;
      }
    }
    return;
  }
  // epilogue: C/D layout of 32x32 MFMA: tile column = lane&31 (-> output column NT*l31 + nt),
  // row = (r&3) + 8*(r>>2) + 4*(lane>>5); every lane stores NT adjacent floats per row.
  const int col0 = n0 + wn * WTN + NT * l31;
  if (p.splits > 1) {                            // raw partial tile; splitk_epilogue_kernel finishes
    float* slab = p.slab + (long)blockIdx.y * p.M * p.Np;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < p.M) {
          bvec out;
#pragma unroll
          for (int n = 0; n < NT; ++n) vset<NT>(out, n, acc[i][n][r]);
          *reinterpret_cast<bvec*>(slab + (long)m * p.Np + col0) = out;
        }
      }
    return;
  }
  const bool col_ok = col0 < p.n_store;          // n_store is a multiple of 16: all NT columns in or out
  float csum[NT];                                // EPI 1 + p.colsum: this lane's share of the column sums
#pragma unroll
  for (int n = 0; n < NT; ++n) csum[n] = 0.f;
  bvec bv;
#pragma unroll
  for (int n = 0; n < NT; ++n) vset<NT>(bv, n, (EPI == 0 && p.bias && col_ok) ? p.bias[col0 + n] : 0.f);
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    // Back-prop: gather the ReLU mask of the 16 rows first, branch-free (m_off is valid for every tile row).
    // Loads interleaved with the stores below are serialised by the compiler -- for all it knows the output
    // aliases the mask source -- and each then exposes a full memory latency.
    bvec mk[16];
    if (EPI == 1 && p.mask) {
      const int colc = col_ok ? col0 : 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        mk[r] = *reinterpret_cast<const bvec*>(p.mask + m_off[row] + colc);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const long co = c_off[row];
      if (co >= 0 && col_ok) {
        bvec out;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          float v = acc[i][n][r];
          if (EPI == 0) {
            v += vget<NT>(bv, n);
            if (p.relu) v = fmaxf(v, 0.f);
          } else if (p.mask) {
            v = vget<NT>(mk[r], n) > 0.f ? v : 0.f;
          }
          if (EPI == 1) csum[n] += v;
          vset<NT>(out, n, v);
        }
        *reinterpret_cast<bvec*>(Cbase + co + col0) = out;
      }
    }
  }
  if (EPI == 1 && p.colsum) {
    // the gradient tile just stored is also the operand of the bias gradient of the layer below: its column sums
    // over this wave's rows (lanes l and l + 32 hold the same columns, other rows), fixed order, no atomics
    bvec out;
#pragma unroll
    for (int n = 0; n < NT; ++n) vset<NT>(out, n, csum[n] + __shfl_xor(csum[n], 32, 64));
    if (h == 0) *reinterpret_cast<bvec*>(p.colsum + (long)(tile_m * WMW + wm) * p.Np + col0) = out;
  }
}

// ------------------------------------------------------------------------------------
// Per-bin products of the frequency-domain layers as ONE persistent launch ("stream-K"): `bins` independent GEMMs
// C[b] = A[b] * B[b] of a shape whose tile count loads the CUs unevenly -- the seven 7-tap layers: 36 bins x 16 tiles of
// 64 x 128 = 576 workgroups, three on 64 CUs and two on the other 192: the launch costs three tiles for 2.25 tiles of work
// per CU (measured rounds 2-3: 32 bins 46 us, 36 bins 65 us; a K-split of the last bins only, round 3, bought nothing because
// its slices started when the full tiles left).  Here the (bin, tile, k-tile) unit list is dealt from t = 0 in equal
// contiguous runs to 8 x 64 (or 8 x 96) workgroups (csrc/streamk_map.h): every workgroup walks the same number of k-tiles --
// the end of one tile, whole tiles, the start of the next.  A tile cut into pieces is finished by the workgroup that holds
// its START (the last piece of its run, computed late); the workgroups holding the rest computed theirs first thing and
// published them long before.  Hand-off per MI355X_MICROARCH.md (R1: write-through `sc1` payload, every storing wave drains
// `vmcnt`, one relaxed agent-scope flag; the reader polls that one word and reads the payload with `sc1` loads) --
// independent of where the workgroups run; the sum is head + next + next ..., a fixed order.
// Flags carry the launch's epoch (a host counter, never 0) and are put back to 0 by the reader: the scratch may hold ANY
// content before the first call (a flag equals the epoch by accident with probability 2^-32), and a launch replayed from a
// graph -- its epoch frozen -- finds the zeros its previous replay left.  The reader waits for workgroups with HIGHER block
// ids on the same XCD's dispatcher (blockIdx.x + 8, + 16), which that dispatcher starts no later than the reader's own
// successors; the poll is bounded all the same (a lost producer must not hang the GPU; the timeout word counts).
// Inner loop: the FAST stage of gemm_nn_kernel (whole k-tiles, LDS-DMA with source-side XOR swizzle, DMA slices between
// MFMA quads); no row-offset tables (rows are plain multiples of lda).
// ------------------------------------------------------------------------------------
// Lost hand-offs since the library was loaded (module-scope, zero at load): a reader whose poll ran out counts here AND poisons its
// tile with NaN instead of adding whatever the producer's slot holds; the host reads the word with the step's other status words
// (st_streamk_lost_ptr / st_streamk_lost_count; engine.fetch_losses raises on a non-zero count).
__device__ unsigned g_sk_lost = 0;

struct BinsParams {
  const float* A; const float* B; float* C;
  long lda, ldb, ldc, a_batch, b_batch, c_batch;
  int tiles_m, tiles_n;               // per bin
  st::SkPlan plan;
  unsigned* ctrl;                     // st::SK_CTRL_WORDS control words (flags, timeout count)
  float* partial;                     // [8 * wgs_per_xcd][BM * BN]
  unsigned epoch;                     // != 0
  unsigned pub_epoch;                 // what producers publish: == epoch (the "streamk_test_drop" knob: another value -- every hand-off is lost)
};

// BT: B given transposed ([N][ldb], the reduction index contiguous) -- back-prop to the input reads the FORWARD filter spectra
// (gbwd = gfwd^T); staged and read like the A operand, a lane's two columns 32 apart (see gemm_nn_kernel).
template <int BM, int BN, int WMW, int WNW, int MINW, bool BT>
__global__ __launch_bounds__(NTHREADS, MINW) void gemm_nn_bins_kernel(BinsParams p) {
  constexpr int WTM = BM / WMW, WTN = BN / WNW;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int A_DMA = BM / 32, B_DMA = BN / 32, B_LPR = BN / 4;
  constexpr int A_SZ = BM * BK, B_SZ = BK * BN;
  constexpr int N_DMA = A_DMA + B_DMA;
  static_assert(WMW * WNW == 4 && MT >= 1 && NT == 2 && A_DMA >= 1 && B_DMA >= 1, "tile config");
  typedef typename FVec<NT>::type bvec;
  typedef unsigned long long u64;

  // (ONE LDS object: with a second __shared__ variable the LDS lowering tags every access with alias scopes, the waitcnt pass
  // then sees fragment reads that may alias the LDS-DMA in flight and drains it -- s_waitcnt vmcnt(0) in the stage loop, 54 -> 77 us
  // per launch, measured round 5.  The sticky flag lives in four floats behind the stages.)
  __shared__ __attribute__((aligned(16))) float smem[2 * A_SZ + 2 * B_SZ + 4];
  float* const As = smem;
  float* const Bs = smem + 2 * A_SZ;
  volatile int* const sk_lost_p = reinterpret_cast<volatile int*>(smem + 2 * A_SZ + 2 * B_SZ);   // sticky: a hand-off timed out

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave / WNW, wn = wave % WNW;
  if (tid == 0) *sk_lost_p = 0;       // (read only behind the barrier that follows a poll)

  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int slot = blockIdx.x;                               // this workgroup's partial tile and flag
  st::SkCursor cur;
  if (!st::sk_begin(p.plan, xcd, local, cur)) return;
  const int per_bin = p.tiles_m * p.tiles_n;

  // lane-constant parts of the DMA sources (see gemm_nn_kernel): A instruction i of this wave covers rows
  // (wave * A_DMA + i) * 8 .. +8, lane -> (row, swizzled 16-byte slot); B instruction i covers 64 / B_LPR rows of k
  int a_lane[A_DMA];
#pragma unroll
  for (int i = 0; i < A_DMA; ++i) {
    const int row = (wave * A_DMA + i) * 8 + (lane >> 3);
    a_lane[i] = row * (int)p.lda + (((lane & 7) ^ ((row >> 1) & 7))) * 4;           // (BM rows of a bin: fits an int)
  }
  int b_lane[B_DMA];
#pragma unroll
  for (int i = 0; i < B_DMA; ++i) {
    if (BT) {
      const int row = (wave * B_DMA + i) * 8 + (lane >> 3);
      b_lane[i] = row * (int)p.ldb + (((lane & 7) ^ ((row >> 1) & 7))) * 4;                             // (BN rows of Bt)
    } else {
      b_lane[i] = ((wave * B_DMA + i) * (64 / B_LPR) + lane / B_LPR) * (int)p.ldb + (lane % B_LPR) * 4;   // (32 rows of B)
    }
  }
  int bt_frag[4];
  {
    const int row = wn * WTN + l31;
    const int sw = (row >> 1) & 7;
#pragma unroll
    for (int q = 0; q < 4; ++q) bt_frag[q] = row * BK + (((2 * q + h) ^ sw) * 4);
  }
  int a_frag[4];
  {
    const int row = wm * WTM + l31;
    const int sw = (row >> 1) & 7;
#pragma unroll
    for (int q = 0; q < 4; ++q) a_frag[q] = row * BK + (((2 * q + h) ^ sw) * 4);
  }
  const int b_frag = (4 * h) * BN + wn * WTN + NT * l31;
  const int tcol = wn * WTN + NT * l31;                      // this lane's first column inside the tile (the partial tiles'
                                                             // 8-byte slots keep this address also when the columns are 32 apart)

  while (cur.u < cur.u_end) {
    const st::SkPiece pc = st::sk_piece(p.plan, cur);
    cur.u += pc.kt1 - pc.kt0;
    const int bin = pc.tile / per_bin, t = pc.tile - bin * per_bin;
    const int tile_n = t / p.tiles_m, tile_m = t - tile_n * p.tiles_m;   // row tiles fastest: neighbours share a filter panel
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const float* __restrict__ Ab = p.A + (long)bin * p.a_batch + (long)m0 * p.lda;
    const float* __restrict__ Bb = p.B + (long)bin * p.b_batch + (BT ? (long)n0 * p.ldb : (long)n0);
    const float* aptr[A_DMA];
    const float* bptr[B_DMA];
#pragma unroll
    for (int i = 0; i < A_DMA; ++i) aptr[i] = Ab + a_lane[i];
#pragma unroll
    for (int i = 0; i < B_DMA; ++i) bptr[i] = Bb + b_lane[i];
    auto dma_piece = [&](int pcx, int k0, int buf) {
      if (pcx < A_DMA) {
        const int i = pcx < A_DMA ? pcx : 0;
        __builtin_amdgcn_global_load_lds((gptr_t)(aptr[i] + k0), (lptr_t)(As + buf * A_SZ + (wave * A_DMA + i) * 256), 16, 0, 0);
      } else {
        const int i = pcx - A_DMA < B_DMA ? pcx - A_DMA : 0;
        __builtin_amdgcn_global_load_lds((gptr_t)(BT ? bptr[i] + k0 : bptr[i] + (long)k0 * p.ldb), (lptr_t)(Bs + buf * B_SZ + (wave * B_DMA + i) * 256), 16, 0, 0);
      }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int k0 = pc.kt0 * BK;                                    // reduction index of the tile being COMPUTED
#pragma unroll
    for (int pcx = 0; pcx < N_DMA; ++pcx) dma_piece(pcx, k0, 0);
    __syncthreads();                                         // (drains the DMA; the previous piece's stores too)

    auto stage = [&](auto cur_c, auto more_c) {
      constexpr int CUR = decltype(cur_c)::value;
      constexpr bool MORE = decltype(more_c)::value;
      const int nk0 = k0 + BK;
      k0 = nk0;
      const float* as = As + CUR * A_SZ;
      const float* bs = Bs + CUR * B_SZ + (BT ? 0 : b_frag);
      f32x4 af[4][MT];
      bvec bf[4][4];
      f32x4 bft[4][NT];
      auto read_frags = [&](int q) {
#pragma unroll
        for (int i = 0; i < MT; ++i) af[q][i] = *reinterpret_cast<const f32x4*>(as + a_frag[q] + i * 32 * BK);
        if (BT) {
#pragma unroll
          for (int n = 0; n < NT; ++n) bft[q][n] = *reinterpret_cast<const f32x4*>(bs + bt_frag[q] + n * 32 * BK);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) bf[q][j] = *reinterpret_cast<const bvec*>(bs + (8 * q + j) * BN);
        }
      };
      read_frags(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (MORE) {
#pragma unroll
          for (int pcx = q; pcx < N_DMA; pcx += 4) dma_piece(pcx, nk0, CUR ^ 1);
        }
        if (q < 3) read_frags(q + 1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q][i][j], BT ? bft[q][n][j] : vget<NT>(bf[q][j], n), acc[i][n], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x006, 8, 0);
          __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    };
    using std::integral_constant;
    using std::true_type;
    using std::false_type;
    const int nk = pc.kt1 - pc.kt0;
    int kt = 0;
    for (; kt + 2 < nk; kt += 2) {
      stage(integral_constant<int, 0>{}, true_type{});
      stage(integral_constant<int, 1>{}, true_type{});
    }
    if (kt + 2 == nk) {
      stage(integral_constant<int, 0>{}, true_type{});
      stage(integral_constant<int, 1>{}, false_type{});
    } else if (kt + 1 == nk) {
      stage(integral_constant<int, 0>{}, false_type{});
    }

    if (pc.kt0 > 0) {
      // a piece that does not hold the tile's start (always the first piece of this run): publish the partial tile --
      // write-through stores, drained by every storing wave -- then the flag
      float* const mine = p.partial + (long)slot * (BM * BN);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const u64 bits = (u64)__float_as_uint(acc[i][0][r]) | ((u64)__float_as_uint(acc[i][1][r]) << 32);
          __hip_atomic_store(reinterpret_cast<u64*>(mine + row * BN + tcol), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(p.ctrl + st::SK_FLAGS + slot, p.pub_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      continue;
    }
    // The piece holds the tile's start.  If it is not the whole tile, the rest are the FIRST pieces of the next runs of this
    // label (blocks slot + 8, + 16, ...), computed at the start of the launch: add them in that order.
    int covered = pc.kt1, src = slot;
    while (covered < p.plan.nk) {
      src += 8;
      if (tid == 0) {
        unsigned* flag = p.ctrl + st::SK_FLAGS + src;
        unsigned spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.epoch) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > (1u << 21)) {
            __hip_atomic_fetch_add(p.ctrl + st::SK_TIMEOUTS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&g_sk_lost, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *sk_lost_p = 1;
            break;
          }
        }
        __hip_atomic_store(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // taken back: zero again for the next call
      }
      __syncthreads();
      // an unpublished partial must not be consumed silently: the tile becomes NaN (and the count says why)
      const float poison = *sk_lost_p ? __uint_as_float(0x7fc00000u) : 0.f;
      const float* const theirs = p.partial + (long)src * (BM * BN);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        u64 part[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          part[r] = __hip_atomic_load(reinterpret_cast<const u64*>(theirs + row * BN + tcol), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc[i][0][r] += __uint_as_float((unsigned)part[r]) + poison;
          acc[i][1][r] += __uint_as_float((unsigned)(part[r] >> 32)) + poison;
        }
      }
      covered += min(p.plan.upw, p.plan.nk - covered);
    }
    float* __restrict__ Cb = p.C + (long)bin * p.c_batch + (long)m0 * p.ldc + n0 + (BT ? wn * WTN + l31 : tcol);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (BT) {                            // this lane's columns are 32 apart
#pragma unroll
          for (int n = 0; n < NT; ++n) Cb[(long)row * p.ldc + n * 32] = acc[i][n][r];
        } else {
          bvec out;
#pragma unroll
          for (int n = 0; n < NT; ++n) vset<NT>(out, n, acc[i][n][r]);
          *reinterpret_cast<bvec*>(Cb + (long)row * p.ldc) = out;
        }
      }
  }
}

// ------------------------------------------------------------------------------------
// A bin's COMPLEX product in three real products (Gauss), one launch, one pass over the operands (round 6).
// (a + ib)(c + id) = (k1 - k3) + i (k1 + k2) with k1 = c (a + b), k2 = a (d - c), k3 = b (c + d): as the real embedding
// [re | im] x [[Gr, -Gi], [Gi, Gr]] the 32-tap layer's per-bin products execute four real products of the half-size operands
// (51.5 GFLOP per pass at config 2); with the sums formed where the spectra are produced (transforms: a + b; filter spectra:
// d - c and -(c + d), signs included) they are three (38.7 GFLOP).  The shared product k1 is computed ONCE per output tile:
//     phase 0:  acc  = A0 . B0                 (k1)           acc2 = acc
//     phase 1:  acc  += A1 . B1                (-> real part)
//     phase 2:  acc2 += A2 . B2                (-> imaginary part)
// one K-loop of 3 K / 32 stages over two accumulator sets (128 of the wave's registers), the tile's real and imaginary parts
// stored `c_off2` columns apart -- the [re | im] rows the inverse transforms read.  A_p = A + a_off[p] (column offsets inside a
// spectra row), B_p = B + b_off[p] (element offsets of a bin's three filter planes).  Stage, swizzle and scheduling are the FAST
// stage of gemm_nn_kernel; BT reads the planes transposed in place (back-prop to the input).  Whole k-tiles, K / 32 even.
// ksplit = 2 (blockIdx.y): a product with few output tiles and a long reduction -- back-prop to the input of the 32-tap layer:
// 256 x 256 outputs per bin over 3 x 2048 -- gives every workgroup half of each phase's reduction and ADDS its tile into a
// zeroed C with float atomics: exactly two addends per element onto +0, so the sum is the same whichever arrives first
// (a + b == b + a) and the result stays bit-reproducible.
struct G3Params {
  const float* A; long lda, a_batch; long a_off[3];
  const float* B; long ldb, b_batch; long b_off[3];   // ldb: floats between rows of B (BT: between rows of B^T)
  float* C; long ldc, c_batch, c_off2;
  int M, K, N;                                        // K: reduction length of ONE phase (of one split of it)
  int tiles_m, tiles_n, batches, ksplit;
};

// rows [0, M) of every bin, columns [0, N) and [c_off2, c_off2 + N): zero (four rows per workgroup, N a multiple of 128)
__global__ __launch_bounds__(256) void g3_zero_out_kernel(float* __restrict__ C, long ldc, long c_batch, long c_off2, int M, int N) {
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  float* row = C + (long)blockIdx.y * c_batch + (long)m * ldc;
  for (int c = (threadIdx.x & 63) * 4; c < N; c += 256) {
    *reinterpret_cast<f32x4*>(row + c) = f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(row + c_off2 + c) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

template <int BM, bool BT>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_nn_g3_kernel(G3Params p) {
  constexpr int BN = 128, WMW = 2, WNW = 2;
  constexpr int WTM = BM / WMW, WTN = BN / WNW;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int A_DMA = BM / 32, B_DMA = BN / 32, B_LPR = BN / 4;
  constexpr int A_SZ = BM * BK, B_SZ = BK * BN;
  constexpr int N_DMA = A_DMA + B_DMA;
  static_assert(MT >= 1 && NT == 2, "tile config");
  typedef typename FVec<NT>::type bvec;

  __shared__ __attribute__((aligned(16))) float smem[2 * A_SZ + 2 * B_SZ];
  float* const As = smem;
  float* const Bs = smem + 2 * A_SZ;

  // one bin per XCD at a time (its operands stay in that L2), row tiles fastest (see gemm_nn_kernel's batched mode)
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int per_bin = p.tiles_m * p.tiles_n;
  const int set = local / per_bin, t = local - set * per_bin;
  const int bin = set * 8 + xcd;
  if (bin >= p.batches) return;
  const int tile_n = t / p.tiles_m, tile_m = t - tile_n * p.tiles_m;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave / WNW, wn = wave % WNW;

  // (a split of the reduction: this workgroup's share of every phase starts blockIdx.y * K further in)
  const float* __restrict__ Ab = p.A + (long)bin * p.a_batch + (long)blockIdx.y * p.K;
  const float* __restrict__ Bb = p.B + (long)bin * p.b_batch + (long)blockIdx.y * p.K * (BT ? 1L : p.ldb);
  const float* aptr[A_DMA];
  const float* bptr[B_DMA];
#pragma unroll
  for (int i = 0; i < A_DMA; ++i) {
    const int row = (wave * A_DMA + i) * 8 + (lane >> 3);
    aptr[i] = Ab + (long)min(m0 + row, p.M - 1) * p.lda + (((lane & 7) ^ ((row >> 1) & 7))) * 4;   // rows past M: re-read, never stored
  }
#pragma unroll
  for (int i = 0; i < B_DMA; ++i) {
    if (BT) {
      const int row = (wave * B_DMA + i) * 8 + (lane >> 3);
      bptr[i] = Bb + (long)(n0 + row) * p.ldb + (((lane & 7) ^ ((row >> 1) & 7))) * 4;
    } else {
      bptr[i] = Bb + (long)((wave * B_DMA + i) * (64 / B_LPR) + lane / B_LPR) * p.ldb + n0 + (lane % B_LPR) * 4;
    }
  }
  // (ao, bo): element offsets of a stage -- phase offset + position inside the phase
  auto dma_piece = [&](int pcx, long ao, long bo, int buf) {
    if (pcx < A_DMA) {
      const int i = pcx < A_DMA ? pcx : 0;
      __builtin_amdgcn_global_load_lds((gptr_t)(aptr[i] + ao), (lptr_t)(As + buf * A_SZ + (wave * A_DMA + i) * 256), 16, 0, 0);
    } else {
      const int i = pcx - A_DMA < B_DMA ? pcx - A_DMA : 0;
      __builtin_amdgcn_global_load_lds((gptr_t)(bptr[i] + bo), (lptr_t)(Bs + buf * B_SZ + (wave * B_DMA + i) * 256), 16, 0, 0);
    }
  };
  const long bstep = BT ? (long)BK : (long)BK * p.ldb;         // a k-tile further inside a plane

  int a_frag[4], bt_frag[4];
  {
    const int row = wm * WTM + l31, sw = (row >> 1) & 7;
#pragma unroll
    for (int q = 0; q < 4; ++q) a_frag[q] = row * BK + (((2 * q + h) ^ sw) * 4);
  }
  {
    const int row = wn * WTN + l31, sw = (row >> 1) & 7;
#pragma unroll
    for (int q = 0; q < 4; ++q) bt_frag[q] = row * BK + (((2 * q + h) ^ sw) * 4);
  }
  const int b_frag = (4 * h) * BN + wn * WTN + NT * l31;

  f32x16 acc[MT][NT], acc2[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int pcx = 0; pcx < N_DMA; ++pcx) dma_piece(pcx, p.a_off[0], p.b_off[0], 0);
  __syncthreads();

  // one k-tile into `tgt`; (nao, nbo): offsets of the tile to stage meanwhile
  auto stage = [&](auto cur_c, auto more_c, f32x16 (&tgt)[MT][NT], long nao, long nbo) __attribute__((always_inline)) {
    constexpr int CUR = decltype(cur_c)::value;
    constexpr bool MORE = decltype(more_c)::value;
    const float* as = As + CUR * A_SZ;
    const float* bs = Bs + CUR * B_SZ + (BT ? 0 : b_frag);
    f32x4 af[4][MT];
    bvec bf[4][4];
    f32x4 bft[4][NT];
    auto read_frags = [&](int q) {
#pragma unroll
      for (int i = 0; i < MT; ++i) af[q][i] = *reinterpret_cast<const f32x4*>(as + a_frag[q] + i * 32 * BK);
      if (BT) {
#pragma unroll
        for (int n = 0; n < NT; ++n) bft[q][n] = *reinterpret_cast<const f32x4*>(bs + bt_frag[q] + n * 32 * BK);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[q][j] = *reinterpret_cast<const bvec*>(bs + (8 * q + j) * BN);
      }
    };
    read_frags(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (MORE) {
#pragma unroll
        for (int pcx = q; pcx < N_DMA; pcx += 4) dma_piece(pcx, nao, nbo, CUR ^ 1);
      }
      if (q < 3) read_frags(q + 1);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            tgt[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q][i][j], BT ? bft[q][n][j] : vget<NT>(bf[q][j], n), tgt[i][n], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x006, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                         // also drains this wave's DMA (vmcnt) before anyone reads it
  };
  using std::integral_constant;
  using std::true_type;
  using std::false_type;
  const int spp = p.K / BK;                  // stages per phase (even)
  // a phase: pairs of stages; the tile staged under the last stage of phases 0 and 1 is the first of the next phase
  auto phase = [&](f32x16 (&tgt)[MT][NT], long ao, long bo, long next_ao, long next_bo, auto last_c) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_c)::value;
    for (int kt = 0; kt + 2 < spp; kt += 2) {
      stage(integral_constant<int, 0>{}, true_type{}, tgt, ao + (kt + 1) * BK, bo + (kt + 1) * bstep);
      stage(integral_constant<int, 1>{}, true_type{}, tgt, ao + (kt + 2) * BK, bo + (kt + 2) * bstep);
    }
    stage(integral_constant<int, 0>{}, true_type{}, tgt, ao + (spp - 1) * BK, bo + (spp - 1) * bstep);
    if (LAST) stage(integral_constant<int, 1>{}, false_type{}, tgt, 0L, 0L);
    else stage(integral_constant<int, 1>{}, true_type{}, tgt, next_ao, next_bo);
  };
  phase(acc, p.a_off[0], p.b_off[0], p.a_off[1], p.b_off[1], false_type{});
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc2[i][j] = acc[i][j];
  phase(acc, p.a_off[1], p.b_off[1], p.a_off[2], p.b_off[2], false_type{});
  phase(acc2, p.a_off[2], p.b_off[2], 0L, 0L, true_type{});

  float* __restrict__ Cb = p.C + (long)bin * p.c_batch + n0 + wn * WTN + (BT ? l31 : NT * l31);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (m < p.M) {
        float* row = Cb + (long)m * p.ldc;
        if (p.ksplit > 1) {
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const int cn = BT ? n * 32 : n;
            __hip_atomic_fetch_add(row + cn, acc[i][n][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(row + p.c_off2 + cn, acc2[i][n][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        } else if (BT) {                     // this lane's columns are 32 apart
#pragma unroll
          for (int n = 0; n < NT; ++n) { row[n * 32] = acc[i][n][r]; row[p.c_off2 + n * 32] = acc2[i][n][r]; }
        } else {
          bvec o1, o2;
#pragma unroll
          for (int n = 0; n < NT; ++n) { vset<NT>(o1, n, acc[i][n][r]); vset<NT>(o2, n, acc2[i][n][r]); }
          *reinterpret_cast<bvec*>(row) = o1;
          *reinterpret_cast<bvec*>(row + p.c_off2) = o2;
        }
      }
    }
}

// ------------------------------------------------------------------------------------
// Filter gradient: out[split][k][n] = sum over the split's rows m of A[m][k] * Z[m][n].
// Both operands are reduction-major in memory (a row of A / Z is one value of the reduction
// index m) and are staged as they are, linear [m][cols], by LDS-DMA.  A lane fetches MT (NT)
// adjacent columns of row m with one ds_read_b64, so MFMA tile i of a wave covers output rows
// {MT*lane + i}; the epilogue undoes that permutation.  Tile 128(k) x BN(n), 32 rows of m per
// stage, the row range of a split walked with a scalar base per DMA piece (no per-row division).
// ------------------------------------------------------------------------------------
struct TNParams {
  const float* A; RowMap amap;
  const float* Z; RowMap zmap;
  float* out;            // [splits][Kp][Np]
  int M, Kvalid, Kp, Np, z_cols;   // z_cols = readable floats per Z row (its c_pitch)
  int rows_per_split;    // multiple of 32
  int tiles_k, tiles_n;
  int amap_batches;      // utterances (rows never advance past the last one)
  int adv_b, adv_t;      // 32 rows = adv_b utterances + adv_t frames
  long a_batch, z_batch, o_batch;   // blockIdx.z: independent products of the same shape (csrc/conv_fft.hip), float strides
  int z_shift;                      // ... product b reads Z of batch b >> z_shift (the real and imaginary lag products of a bin share Z)
};

__device__ __attribute__((aligned(16))) float g_zero_row[4] = {0.f, 0.f, 0.f, 0.f};   // DMA source of rows past a split's end

constexpr int TN_THREADS = NTHREADS;

template <int BN, int WKW, int WNW>
__global__ __launch_bounds__(TN_THREADS) void gemm_tn_kernel(TNParams p) {
  constexpr int BKO = 128;
  constexpr int BMR = 32;
  constexpr int WTK = BKO / WKW, WTN = BN / WNW;   // wave tile
  constexpr int MT = WTK / 32, NT = WTN / 32;
  constexpr int A_DMA = BKO / 32, Z_DMA = BN / 32; // DMA instructions per wave and stage
  constexpr int A_LPR = BKO / 4, Z_LPR = BN / 4;   // lanes per stage row
  constexpr int A_SZ = BMR * BKO, Z_SZ = BMR * BN;
  constexpr int N_DMA = A_DMA + Z_DMA;
  static_assert(WKW * WNW == 4 && (MT == 1 || MT == 2 || MT == 4) && (NT == 1 || NT == 2 || NT == 4), "tile config");
  typedef typename FVec<MT>::type avec;
  typedef typename FVec<NT>::type zvec;

  // LDS-DMA staged, linear [m][cols] images (one array: see gemm_nn_kernel).  A lane fetches MT
  // (resp. NT) adjacent columns of reduction row m with one read, so MFMA tile i of a wave covers
  // output rows {MT*lane + i}: a permutation the epilogue undoes with MT-strided row addresses.
  __shared__ __attribute__((aligned(16))) float smem[2 * A_SZ + 2 * Z_SZ];
  float* const As = smem;
  float* const Zs = smem + 2 * A_SZ;

  // XCD-aware order (see gemm_nn_kernel): 8x8 super-tiles so the CUs behind one L2 share operands
  int tile_k, tile_n;
  {
    const int bid = blockIdx.x;
    const int total = p.tiles_k * p.tiles_n;
    if ((p.tiles_k & 7) == 0 && (p.tiles_n & 7) == 0) {
      const int idx = (bid & 7) * (total >> 3) + (bid >> 3);
      const int sb = idx >> 6, within = idx & 63;
      const int sbn = p.tiles_n >> 3;
      tile_k = (sb / sbn) * 8 + (within >> 3);
      tile_n = (sb % sbn) * 8 + (within & 7);
    } else {
      tile_n = bid % p.tiles_n;
      tile_k = bid / p.tiles_n;
    }
  }
  const int k0 = tile_k * BKO, n0 = tile_n * BN;
  const int split = blockIdx.y;
  const float* __restrict__ Ab = p.A + (long)blockIdx.z * p.a_batch;
  const float* __restrict__ Zb = p.Z + (long)(blockIdx.z >> p.z_shift) * p.z_batch;
  const int m_begin = split * p.rows_per_split;
  const int m_end = min(p.M, m_begin + p.rows_per_split);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wk = wave / WNW, wn = wave % WNW;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nstages = (m_end - m_begin + BMR - 1) / BMR;

  // ---- staging: every wave issues 1/4 of the stage's DMA pieces, in slices between MFMA quads ----
  // A piece = 64/A_LPR rows of the A stage (1 KiB), likewise for Z.  In a PLAIN stage (all 32 rows
  // inside one utterance and inside the split -- all but ~1 in 15) a piece is addressed by a scalar
  // base + a constant per-lane byte offset: no VALU between the DMA instructions, which matters
  // because any VALU burst in an MFMA-bound wave drains the matrix pipe.  Boundary / last stages
  // take the general per-lane path.
  constexpr int A_PIECES = BMR * A_LPR / 64, Z_PIECES = BMR * Z_LPR / 64;
  constexpr int A_PW = (A_PIECES + 3) / 4, Z_PW = (Z_PIECES + 3) / 4;       // pieces per wave
  const int acol = min(k0 + (lane % A_LPR) * 4, p.Kvalid - 4);   // clamped into the row; masked at the store
  const int zcol = min(n0 + (lane % Z_LPR) * 4, p.z_cols - 4);
  const int arow = lane / A_LPR, zrow = lane / Z_LPR;
  const long a_jump = p.amap.batch_stride - (long)p.amap.frames * p.amap.row_stride;
  const long z_jump = p.zmap.batch_stride - (long)p.zmap.frames * p.zmap.row_stride;
  const bool tiny = p.amap.frames < BMR;          // more than one utterance boundary per stage possible
  const unsigned a_voff = (unsigned)((arow * p.amap.row_stride + acol) * 4);
  const unsigned z_voff = (unsigned)((zrow * p.zmap.row_stride + zcol) * 4);
  const unsigned lds_a = lds_addr(As), lds_z = lds_addr(Zs);
  const long a_step = (long)(64 / A_LPR) * p.amap.row_stride * 4, z_step = (long)(64 / Z_LPR) * p.zmap.row_stride * 4;
  int sb = min(m_begin, p.M - 1) / p.amap.frames;  // (utterance, frame) of the first row of the stage being STAGED
  int stt = min(m_begin, p.M - 1) - sb * p.amap.frames;

  // slice `sl` (0..3) of the stage starting at row mb -> LDS buffer buf
  auto issue_slice = [&](int sl, int mb, int buf) {
    const bool plain = !tiny && stt + BMR <= p.amap.frames && mb + BMR <= m_end;
    if (plain) {
      const char* sa = reinterpret_cast<const char*>(Ab + ((long)sb * p.amap.batch_stride + p.amap.row0 + (long)stt * p.amap.row_stride));
      const char* sz = reinterpret_cast<const char*>(Zb + ((long)sb * p.zmap.batch_stride + p.zmap.row0 + (long)stt * p.zmap.row_stride));
#pragma unroll
      for (int i = sl; i < A_PW; i += 4) {
        const int pc = wave * A_PW + i;
        if (pc < A_PIECES) dma16_sv(lds_a + (buf * A_SZ + pc * 256) * 4, sa + pc * a_step, a_voff);
      }
#pragma unroll
      for (int i = sl; i < Z_PW; i += 4) {
        const int pc = wave * Z_PW + i;
        if (pc < Z_PIECES) dma16_sv(lds_z + (buf * Z_SZ + pc * 256) * 4, sz + pc * z_step, z_voff);
      }
    } else {
#pragma unroll
      for (int i = sl; i < A_PW + Z_PW; i += 4) {
        const bool isA = i < A_PW;
        const int pc = isA ? wave * A_PW + i : wave * Z_PW + (i - A_PW);
        if (pc >= (isA ? A_PIECES : Z_PIECES)) continue;
        const int r = isA ? pc * (64 / A_LPR) + arow : pc * (64 / Z_LPR) + zrow;
        const int m = min(mb + r, p.M - 1);
        const float* g = isA ? Ab + p.amap.off(m) + acol : Zb + p.zmap.off(m) + zcol;
        if (mb + r >= m_end) g = g_zero_row;            // rows past the split end contribute zero
        float* dst = isA ? As + buf * A_SZ + pc * 256 : Zs + buf * Z_SZ + pc * 256;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)dst, 16, 0, 0);
      }
    }
  };
  auto stage_advance = [&]() {
    stt += p.adv_t; sb += p.adv_b;
    if (stt >= p.amap.frames) { stt -= p.amap.frames; ++sb; }
  };

  if (nstages > 0) {
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) issue_slice(sl, m_begin, 0);
    stage_advance();
  }
  dma_wait_all();
  __syncthreads();

  const int a_frag = (4 * h) * BKO + wk * WTK + MT * l31;
  const int z_frag = (4 * h) * BN + wn * WTN + NT * l31;
  for (int st = 0; st < nstages; ++st) {
    const int cur = st & 1;
    const bool more = st + 1 < nstages;
    const int mb_next = m_begin + (st + 1) * BMR;
    const float* as = As + cur * A_SZ + a_frag;
    const float* zs = Zs + cur * Z_SZ + z_frag;
    avec af[4][4];
    zvec zf[4][4];
    auto read_frags = [&](int q) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        af[q][j] = *reinterpret_cast<const avec*>(as + (8 * q + j) * BKO);
        zf[q][j] = *reinterpret_cast<const zvec*>(zs + (8 * q + j) * BN);
      }
    };
    read_frags(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (more) issue_slice(q, mb_next, cur ^ 1);
      if (q < 3) read_frags(q + 1);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(vget<MT>(af[q][j], i), vget<NT>(zf[q][j], n), acc[i][n], 0, 0, 0);
      // interleave request: one fragment read behind each of the first eight MFMAs of the quad
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) stage_advance();
    dma_wait_all();
    __syncthreads();
  }

  float* out = p.out + (long)split * p.Kp * p.Np + (long)blockIdx.z * p.o_batch;
  const int col0 = n0 + wn * WTN + NT * l31;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = k0 + wk * WTK + MT * ((r & 3) + 8 * (r >> 2) + 4 * h) + i;
      if (k < p.Kp && col0 < p.Np) {
        zvec o;
#pragma unroll
        for (int n = 0; n < NT; ++n)
          vset<NT>(o, n, (k < p.Kvalid && col0 + n < p.z_cols) ? acc[i][n][r] : 0.f);
        *reinterpret_cast<zvec*>(out + (long)k * p.Np + col0) = o;
      }
    }
}

// The lag products of a bin in three real products (see gemm_nn_g3_kernel): Re Q = S_r^T Z_r + S_i^T Z_i and
// Im Q = S_i^T Z_r - S_r^T Z_i from  k1 = S_r^T (Z_r + Z_i),  k2 = (S_i - S_r)^T Z_i,  k3 = (S_r + S_i)^T Z_r:  Re = k1 + k2,
// Im = k3 - k1.  One workgroup per 128 x 128 output tile of a bin walks the bin's M rows three times with other column
// offsets (A_p = A + a_off[p], Z_p = Z + z_off[p]):
//     phase 0: acc = A0^T Z0 (k1), acc2 = -acc;   phase 1: acc += A1^T Z1;   phase 2: acc2 += A2^T Z2
// and stores acc to out[bin][0], acc2 to out[bin][1] ([K][N] each).  Plain row-major operands (row m at m * ld), M a multiple
// of 32, M / 32 stages per phase; the stage is gemm_tn_kernel's (linear LDS-DMA images, scalar base + per-lane offset).
struct TN3Params {
  const float* A; long lda, a_batch; long a_off[3];
  const float* Z; long ldz, z_batch; long z_off[3];
  float* out; long o_batch, o_part;                  // o_part: floats between the real and the imaginary product of a bin
  int M, K, N, tiles_k, tiles_n, batches;
};

__global__ __launch_bounds__(TN_THREADS, 2) void gemm_tn_g3_kernel(TN3Params p) {
  constexpr int BKO = 128, BN = 128, BMR = 32, WKW = 2, WNW = 2;
  constexpr int WTK = BKO / WKW, WTN = BN / WNW;
  constexpr int MT = WTK / 32, NT = WTN / 32;
  constexpr int A_LPR = BKO / 4, Z_LPR = BN / 4;
  constexpr int A_SZ = BMR * BKO, Z_SZ = BMR * BN;
  constexpr int A_PIECES = BMR * A_LPR / 64, Z_PIECES = BMR * Z_LPR / 64;   // 16 + 16 one-KiB pieces per stage
  constexpr int A_PW = A_PIECES / 4, Z_PW = Z_PIECES / 4;                   // pieces per wave
  typedef typename FVec<MT>::type avec;
  typedef typename FVec<NT>::type zvec;
  __shared__ __attribute__((aligned(16))) float smem[2 * A_SZ + 2 * Z_SZ];
  float* const As = smem;
  float* const Zs = smem + 2 * A_SZ;

  // all tiles of a bin on one XCD (blockIdx.x = 8 * (set * tiles + tile) + xcd, bin = 8 * set + xcd): the bin's two operands
  // are read by its 32 tiles out of that L2
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int per_bin = p.tiles_k * p.tiles_n;
  const int set = local / per_bin, t = local - set * per_bin;
  const int bin = set * 8 + xcd;
  if (bin >= p.batches) return;
  const int tile_n = t / p.tiles_k, tile_k = t - tile_n * p.tiles_k;
  const int k0 = tile_k * BKO, n0 = tile_n * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wk = wave / WNW, wn = wave % WNW;

  const char* Ab = reinterpret_cast<const char*>(p.A + (long)bin * p.a_batch + k0);
  const char* Zb = reinterpret_cast<const char*>(p.Z + (long)bin * p.z_batch + n0);
  const unsigned a_voff = (unsigned)(((lane / A_LPR) * p.lda + (lane % A_LPR) * 4) * 4);
  const unsigned z_voff = (unsigned)(((lane / Z_LPR) * p.ldz + (lane % Z_LPR) * 4) * 4);
  const unsigned lds_a = lds_addr(As), lds_z = lds_addr(Zs);
  const long a_step = (long)(64 / A_LPR) * p.lda * 4, z_step = (long)(64 / Z_LPR) * p.ldz * 4;   // bytes between pieces
  const long a_stage = (long)BMR * p.lda * 4, z_stage = (long)BMR * p.ldz * 4;                     // bytes between stages

  // slice `sl` (0..3) of the stage whose first row lies at (sa, sz) -> LDS buffer buf
  auto issue_slice = [&](int sl, const char* sa, const char* sz, int buf) {
#pragma unroll
    for (int i = sl; i < A_PW; i += 4) {
      const int pc = wave * A_PW + i;
      dma16_sv(lds_a + (buf * A_SZ + pc * 256) * 4, sa + pc * a_step, a_voff);
    }
#pragma unroll
    for (int i = sl; i < Z_PW; i += 4) {
      const int pc = wave * Z_PW + i;
      dma16_sv(lds_z + (buf * Z_SZ + pc * 256) * 4, sz + pc * z_step, z_voff);
    }
  };

  f32x16 acc[MT][NT], acc2[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int spp = p.M / BMR;                         // stages per phase
  const int a_frag = (4 * h) * BKO + wk * WTK + MT * l31;
  const int z_frag = (4 * h) * BN + wn * WTN + NT * l31;
  int gst = 0;                                       // stages done so far, over the three phases (LDS buffer = gst & 1)
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) issue_slice(sl, Ab + p.a_off[0] * 4, Zb + p.z_off[0] * 4, 0);
  dma_wait_all();
  __syncthreads();

  auto phase = [&](f32x16 (&tgt)[MT][NT], int ph) __attribute__((always_inline)) {
    const char* sa = Ab + p.a_off[ph] * 4;
    const char* sz = Zb + p.z_off[ph] * 4;
    const char* na = ph < 2 ? Ab + p.a_off[ph < 2 ? ph + 1 : 2] * 4 : nullptr;     // first stage of the next phase
    const char* nz = ph < 2 ? Zb + p.z_off[ph < 2 ? ph + 1 : 2] * 4 : nullptr;
    for (int st = 0; st < spp; ++st, ++gst) {
      const int cur = gst & 1;
      const bool last = st + 1 == spp;
      const bool more = !last || ph < 2;
      const char* nsa = last ? na : sa + (long)(st + 1) * a_stage;
      const char* nsz = last ? nz : sz + (long)(st + 1) * z_stage;
      const float* as = As + cur * A_SZ + a_frag;
      const float* zs = Zs + cur * Z_SZ + z_frag;
      avec af[4][4];
      zvec zf[4][4];
      auto read_frags = [&](int q) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          af[q][j] = *reinterpret_cast<const avec*>(as + (8 * q + j) * BKO);
          zf[q][j] = *reinterpret_cast<const zvec*>(zs + (8 * q + j) * BN);
        }
      };
      read_frags(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (more) issue_slice(q, nsa, nsz, cur ^ 1);
        if (q < 3) read_frags(q + 1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              tgt[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(vget<MT>(af[q][j], i), vget<NT>(zf[q][j], n), tgt[i][n], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      dma_wait_all();
      __syncthreads();
    }
  };
  phase(acc, 0);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc2[i][j] = -acc[i][j];
  phase(acc, 1);
  phase(acc2, 2);

  float* out = p.out + (long)bin * p.o_batch;
  const int col0 = n0 + wn * WTN + NT * l31;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = k0 + wk * WTK + MT * ((r & 3) + 8 * (r >> 2) + 4 * h) + i;
      zvec o1, o2;
#pragma unroll
      for (int n = 0; n < NT; ++n) { vset<NT>(o1, n, acc[i][n][r]); vset<NT>(o2, n, acc2[i][n][r]); }
      *reinterpret_cast<zvec*>(out + (long)k * p.N + col0) = o1;
      *reinterpret_cast<zvec*>(out + p.o_part + (long)k * p.N + col0) = o2;
    }
}

// dst[i] = sum_s slabs[s][i]
__global__ void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dst,
                                   long n4, int splits) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    f32x4 s = reinterpret_cast<const f32x4*>(slabs)[i];
    for (int k0 = 1; k0 < splits; k0 += 8) {                     // eight slabs' loads in flight, added in slab order
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = reinterpret_cast<const f32x4*>(slabs)[i + (long)min(k0 + j, splits - 1) * n4];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k0 + j < splits) s += v[j];
    }
    reinterpret_cast<f32x4*>(dst)[i] = s;
  }
}

// column sums of dz (the bias gradient): partial[(b, chunk)][c] over 256-frame chunks of every
// utterance, then summed in fixed order by a second pass.  HBM-bound streaming read: a block covers
// 128 columns with 16-byte loads, 8 rows in flight per pass.
constexpr int COLSUM_ROWS = 256;
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ Z, long batch_stride, long row0,
                                                             int row_stride, int frames, int cols, int c_pitch,
                                                             float* __restrict__ partial, int np) {
  const int c4 = blockIdx.x * 128 + (threadIdx.x & 31) * 4;
  const int rl = threadIdx.x >> 5;
  const int chunks = gridDim.y;
  const int t_lo = blockIdx.y * COLSUM_ROWS, t_hi = min(frames, t_lo + COLSUM_ROWS);
  const float* base = Z + (long)blockIdx.z * batch_stride + row0 + c4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c4 < c_pitch)
    for (int t = t_lo + rl; t < t_hi; t += 8) acc += *reinterpret_cast<const f32x4*>(base + (long)t * row_stride);
  __shared__ f32x4 red[8][32];
  red[rl][threadIdx.x & 31] = acc;
  __syncthreads();
  if (rl == 0 && c4 < np) {
    f32x4 s = red[0][threadIdx.x];
#pragma unroll
    for (int k = 1; k < 8; ++k) s += red[k][threadIdx.x];
#pragma unroll
    for (int e = 0; e < 4; ++e) if (c4 + e >= cols) s[e] = 0.f;
    *reinterpret_cast<f32x4*>(partial + ((long)blockIdx.z * chunks + blockIdx.y) * np + c4) = s;
  }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int chunks, int np,
                                                           float* __restrict__ dbias) {
  // 32 columns x 8 chunk-lanes per block; lane r sums chunks r, r+8, ... (independent loads in
  // flight), then a fixed-order LDS reduce -- the serial 64-load chain of a 1-thread-per-column
  // version cost 15 us per layer.
  __shared__ float red[8][33];
  const int cl = threadIdx.x & 31, r = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s = 0.f;
  if (c < np) {
    // eight loads in flight, added in chunk order: a loop of one load per ordered add is a latency chain (round 5: the 252
    // partial rows of the 2000 x 2000 layer took 15-17 us this way, twice per step on the compute stream)
    for (int k0 = r; k0 < chunks; k0 += 64) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = partial[(long)min(k0 + 8 * j, chunks - 1) * np + c];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k0 + 8 * j < chunks) s += v[j];
    }
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

// ---- filter layout kernels ---------------------------------------------------------------
__global__ void pack_filters_kernel(const float* __restrict__ f, int W, int cin, int cout, int cp,
                                    int Np, float* __restrict__ packed, int unpack) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)W * cin * cout;
  if (i >= total) return;
  int o = i % cout;
  long r = i / cout;
  int c = r % cin, w = r / cin;
  long pi = ((long)w * cp + c) * Np + o;
  if (unpack) const_cast<float*>(f)[i] = packed[pi];
  else packed[pi] = f[i];
}

// out[(w' * cop + o) * NpT + c] = in[((W-1-w') * cip + c) * Np + o]   (32x32 LDS transpose).
// The grid covers the WHOLE padded output [kt_pad][NpT] and writes zeros into every padding
// element, so no separate clear pass is needed per step.
__global__ __launch_bounds__(256) void flip_transpose_kernel(const float* __restrict__ in, int W, int cin,
                                                             int cout, int cip, int Np, int cop, int NpT,
                                                             int kt_pad, float* __restrict__ out) {
  __shared__ float tile[32][33];
  const int w = blockIdx.z;                 // taps 0..W-1, and W = the k-padding rows past W*cop
  const int c0 = blockIdx.y * 32, o0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  if (w < W) {
    for (int r = ty; r < 32; r += 8) {
      int c = c0 + r, o = o0 + tx;
      tile[r][tx] = (c < cin && o < cout) ? in[((long)(W - 1 - w) * cip + c) * Np + o] : 0.f;
    }
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int o = o0 + r, c = c0 + tx;
    const long krow = (long)w * cop + o;
    if (c < NpT && krow < kt_pad && (w == W || o < cop)) out[krow * NpT + c] = (w < W) ? tile[tx][r] : 0.f;
  }
}

// zero the halo rows (in front of frame 0 and behind the last frame) of every utterance
__global__ __launch_bounds__(256) void zero_halos_kernel(float* base, int frames, int halo, int t_pitch, int c_pitch) {
  const int halo_rows = t_pitch - frames;
  const long row4 = c_pitch / 4;
  const long total = (long)halo_rows * row4;
  float* utt = base + (long)blockIdx.y * t_pitch * c_pitch;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int r = (int)(i / row4);
    const int c4 = (int)(i - r * row4) * 4;
    if (r >= halo) r += frames;                       // rows behind the interior
    *reinterpret_cast<f32x4*>(utt + (long)r * c_pitch + c4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// zero a table of byte ranges, one workgroup per range (st_zero_regions): addresses and sizes are multiples of 16
struct ZeroRegion { unsigned long long address, bytes; };
__global__ __launch_bounds__(256) void zero_regions_kernel(const ZeroRegion* __restrict__ regions) {
  const ZeroRegion r = regions[blockIdx.x];
  f32x4* p = reinterpret_cast<f32x4*>(r.address);
  const unsigned long long n16 = r.bytes / 16;
  for (unsigned long long i = threadIdx.x; i < n16; i += 256) p[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

__global__ void fill_kernel(float* dst, float v, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = v;
}

int npad_of(int cout) { return cout <= 32 ? 32 : (cout <= 64 ? 64 : (int)st::round_up(cout, 128)); }

RowMap make_map(const st_tensor3& t, int first_row, int frame_stride, int frames) {
  RowMap m;
  m.frames = frames;
  m.row_stride = frame_stride * t.c_pitch;
  m.batch_stride = (long)t.t_pitch * t.c_pitch;
  m.row0 = (long)first_row * t.c_pitch;
  return m;
}

// C[m, :] = epilogue(sum_s slab[s][m][:]) for split-K launches (fixed summation order).
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(NNParams p, int epi) {
  const int cols4 = p.n_store / 4;
  const long total = (long)p.M * cols4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int m = (int)(idx / cols4), c4 = (int)(idx % cols4) * 4;
    const float* src = p.slab + (long)m * p.Np + c4;
    f32x4 v = *reinterpret_cast<const f32x4*>(src);
    for (int sidx = 1; sidx < p.splits; ++sidx) v += *reinterpret_cast<const f32x4*>(src + (long)sidx * p.M * p.Np);
    if (epi == 0) {
      if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + c4);
      if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
    } else if (p.mask) {
      f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask + p.mmap.off(m) + c4);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = mk[e] > 0.f ? v[e] : 0.f;
    }
    *reinterpret_cast<f32x4*>(p.C + p.cmap.off(m) + c4) = v;
  }
}

// Split-K policy for C = A * B with few output tiles and a long reduction (back-prop through L8: 252 tiles of
// 128x128 on 256 CUs, K = 64 512).  With about one workgroup per CU every SIMD holds a single wave and
// nothing overlaps its barriers; two K-halves give two co-resident workgroups per CU and, with the lean loop
// of today, 3.99 -> 3.75 ms (128.7 -> 136.8 TFLOP/s) including the slab epilogue; 3 splits leave a ragged round
// (4.74 ms), 4 splits 3.87 ms.  (The first version of this kernel measured no gain: 4.31 vs 4.32 ms.)
int nn_splits(int M, int Np, int Kp) {
  const int forced = st::tuning(st::TUNE_GEMM_SPLITS);
  if (Np % 128) return 1;
  const long tiles128 = (long)st::ceil_div(M, 128) * (Np / 128);
  const int nk = Kp / BK;
  int splits = forced ? forced
                      : (tiles128 < 64 && nk >= 256 ? (int)std::min<long>(8, 256 / tiles128)
                                                     : (tiles128 > 128 && tiles128 <= 320 && nk >= 512 ? 2 : 1));
  while (splits > 1 && nk / splits < 64) --splits;
  return splits;
}

// Forward pass on few output rows (one 2 s utterance: M = 101, so 16 tiles of 128x128 on 256 CUs, each walking
// the whole reduction alone -- L8's 250 k-tiles are a 0.43 ms serial MFMA chain).  The reduction is cut into
// slices of >= 4 k-tiles until about one workgroup per CU exists; the raw partial tiles go to slabs that
// splitk_epilogue_kernel sums in a fixed order before bias + ReLU.  Measured, forward + greedy decode of one
// 2 s utterance: 1.18 ms unsplit, 0.355 ms with this policy (>= 8 tiles per slice 0.396, >= 2: 0.398, twice
// the workgroups 0.363).
int fwd_splits(int M, int Np, int nk) {
  const int forced = st::tuning(st::TUNE_FWD_SPLITS);
  // The 29-class output layer at full batch (L10: 16032 x 2016 x 32) is an HBM stream of the activations with
  // one 128-row tile per workgroup: 126 workgroups leave half the CUs without any and every workgroup walks 1 MB
  // alone (72 us for 130 MB).  Four slices of the reduction put ~2 workgroups on every CU.
  // (round 6: shorter batches too -- 32 x 2 s is 3 232 rows = 26 workgroups, 58 us unsplit; about two workgroups per CU, at most
  // eight slices of at least four k-tiles)
  if (Np == 32 && !forced) {
    if (M < 1024 || nk < 32) return 1;
    const int tiles_m = st::ceil_div(M, 128);
    return std::max(1, std::min(std::min(8, nk / 4), (512 + tiles_m / 2) / tiles_m));
  }
  if (Np % 128) return 1;
  const long tiles128 = (long)st::ceil_div(M, 128) * (Np / 128);
  if (forced) return std::max(1, std::min(forced, nk));
  if (tiles128 >= 128) return 1;
  return (int)std::max<long>(1, std::min<long>(256 / tiles128, nk / 4));
}

template <int BM, int BN, int WMW, int WNW, bool FAST = false, bool BT = false>
void launch_nn(NNParams& p, int epi, hipStream_t s) {
  p.tiles_m = st::ceil_div(p.M, BM);
  p.tiles_n = p.Np / BN;
  {
    // operand bytes an XCD's L2 has to pull in: the activation rows of its M range (once per N column group)
    // and the filter panel of its N range (once per M row group)
    const double a_bytes = (double)p.M * p.cp * 4.0, b_bytes = (double)p.Kp * p.Np * 4.0;
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
    if (8 / p.gm > p.tiles_n) p.gm = 8;                       // fewer than 8/gm filter panels: stack the XCDs along M
    p.tm_per = st::ceil_div(p.tiles_m, p.gm);
    p.tn_per = st::ceil_div(p.tiles_n, 8 / p.gm);
    p.chunk = p.tm_per * p.tn_per;
  }
  p.colsum_rows = p.tiles_m * WMW;
  if (p.batches > 0) p.chunk = st::ceil_div(p.batches, 8) * p.tiles_m * p.tiles_n;
  dim3 grid(p.chunk * 8, p.splits > 1 ? p.splits : 1), block(NTHREADS);
  const double gflop = 2e-9 * p.tiles_m * BM * (double)p.Np * p.Kp * std::max(1, p.batches);      // executed, padding included
  if (p.batches > 0)
    st::trace("gemm_nn<%d,%d,%d,%d,%s> batched bins=%d M=%d Np=%d Kp=%d gflop=%.3f", BM, BN, WMW, WNW, FAST ? (BT ? "fast-bt" : "fast") : "clamped",
              p.batches, p.M, p.Np, p.Kp, gflop);
  else
    st::trace("gemm_nn<%d,%d,%d,%d,%s> epi=%d splits=%d M=%d Np=%d Kp=%d taps=%d xcd=%dx%d gflop=%.3f", BM, BN, WMW, WNW,
              FAST ? (BT ? "fast-bt" : "fast") : "clamped", epi, p.splits > 1 ? p.splits : 1, p.M, p.Np, p.Kp, p.taps, p.gm, 8 / p.gm, gflop);
  {
    st::LaunchTimer timer(s);
    if (epi == 0) st::launch_timed(timer, gemm_nn_kernel<BM, BN, WMW, WNW, 0, FAST, BT>, grid, block, s, p);
    else st::launch_timed(timer, gemm_nn_kernel<BM, BN, WMW, WNW, 1, FAST, BT>, grid, block, s, p);
  }
  if (p.splits > 1) {
    const long quads = (long)p.M * (p.n_store / 4);
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)std::min<long>((quads + 255) / 256, 4096)), dim3(256), 0, s,
                       p, epi);
  }
}

int run_nn(NNParams& p, int epi, hipStream_t s) {
  const int force = st::tuning(st::TUNE_GEMM_TILE);   // perf experiments (st_set_tuning)
  if (p.bt_ld > 0) {
    // the filter operand transposed (k contiguous): whole k-tiles, one tap, 128-column tiles only
    if (!(p.Np % 128 == 0 && p.taps == 1 && p.cp % 32 == 0 && p.Kvalid % 32 == 0 && p.Kvalid == p.Kp && p.bt_ld % 4 == 0 && p.bt_rows > 0)) {
      st::set_error("gemm_nn: the transposed-operand form needs n_pad %% 128 == 0, one tap and a reduction length that is a multiple of 32");
      return ST_EINVAL;
    }
    const long tiles128 = (long)st::ceil_div(p.M, 128) * (p.Np / 128) * std::max(1, p.batches);
    if ((p.batches > 0 && (tiles128 < 512 || p.M <= 64)) || (p.batches <= 0 && tiles128 < 192 && p.splits <= 1)) launch_nn<64, 128, 2, 2, true, true>(p, epi, s);
    else launch_nn<128, 128, 2, 2, true, true>(p, epi, s);
    return st::check_launch("gemm_nn (bt)");
  }
  if (p.Np % 128 == 0) {
    long tiles128 = (long)st::ceil_div(p.M, 128) * (p.Np / 128) * std::max(1, p.batches);
    if (force == 1) launch_nn<64, 128, 2, 2>(p, epi, s);
    else if (force == 2) launch_nn<128, 128, 2, 2>(p, epi, s);
    else if (force == 3) launch_nn<128, 64, 2, 2>(p, epi, s);
    else if (p.batches > 0 && (tiles128 < 512 || p.M <= 64)) {                         // per-bin products with few column tiles (or one half tile of rows)
      if (p.cp % 32 == 0 && p.Kvalid % 32 == 0) launch_nn<64, 128, 2, 2, true>(p, epi, s);    // (back-prop: 8 per bin): twice the
      else launch_nn<64, 128, 2, 2>(p, epi, s);                                               // workgroups, +10 %
    }
    else if ((tiles128 >= 192 || p.splits > 1) && p.cp % 32 == 0 && p.Kvalid % 32 == 0 && !st::tuning(st::TUNE_NO_FAST))
      launch_nn<128, 128, 2, 2, true>(p, epi, s);                                      // whole k-tiles only: unclamped DMA addresses
    else if (tiles128 >= 192 || p.splits > 1) launch_nn<128, 128, 2, 2>(p, epi, s);   // >= 3/4 of the CUs busy
    else launch_nn<64, 128, 2, 2>(p, epi, s);
  } else if (p.Np == 64) {
    launch_nn<128, 64, 2, 2>(p, epi, s);
  } else if (p.Np == 32) {
    launch_nn<128, 32, 4, 1>(p, epi, s);
  } else {
    st::set_error("unsupported packed width n_pad=%d", p.Np);
    return ST_EINVAL;
  }
  return st::check_launch("gemm_nn");
}

}  // namespace

// Plain batched C[b] = A[b] * B[b] (row-major fp32, no epilogue) on the convolution GEMM kernel: A [M][lda] with
// K <= lda readable floats per row, B [K][N] with N a multiple of 128, C [M][ldc]; K a multiple of 32.
// With a workspace (st::SK_WS_FLOATS floats, control words zero) a launch whose 64 x 128 tiles would leave the last round of
// workgroups ragged runs as the persistent stream-K kernel instead (gemm_nn_bins_kernel).
int st::gemm_nn_batched(const float* A, long lda, long a_batch, const float* B, long b_batch, float* C, long ldc,
                        long c_batch, int M, int K, int N, int batches, hipStream_t s, float* sk_ws, bool b_transposed) {
  if (!(A && B && C && M > 0 && K > 0 && K % 32 == 0 && N % 128 == 0 && batches > 0 && lda % 4 == 0 && ldc % 4 == 0)) {
    st::set_error("gemm_nn_batched: bad shape M=%d K=%d N=%d", M, K, N);
    return ST_EINVAL;
  }
  // stream-K policy.  The plain launch of 64 x 128 tiles keeps three workgroups per CU resident (768 slots): up to 768
  // tiles run as one round whose length is the FULLEST CU's.  Worth replacing when that CU holds much more than the average:
  // 36 bins x 16 tiles = 576 -> 3 against 2.25 (66 -> 56 us measured, round 4); not for the first layer's 720 (3 against
  // 2.8: 52 us either way) nor the 32-tap layer's back-prop (768 = 3 each: the persistent form with two workgroups per CU
  // measured slower, 503 against 486 us); launches of several rounds stay as they are.
  // st_set_tuning("streamk", 1) forces it wherever the shape allows, 2 turns it off; "streamk_slots" = workgroups per XCD
  // (64 or 96; 0 = the policy's).
  const int knob = st::tuning(st::TUNE_STREAMK);
  if (sk_ws && knob != 2 && M % 64 == 0 && a_batch % 4 == 0 && b_batch % 4 == 0 && (long)64 * lda < (1L << 30) && (long)32 * N < (1L << 30)) {
    const long tiles = (long)batches * (M / 64) * (N / 128);
    const double per_cu = (double)tiles / 256.0;
    const bool uneven = tiles > 256 && tiles <= 768 && std::ceil(per_cu) / per_cu > 1.2;
    int slots = st::tuning(st::TUNE_STREAMK_SLOTS);
    if (slots != 64 && slots != 96) slots = 64;
    st::SkPlan plan;
    if ((knob == 1 || uneven) && tiles < (1L << 24) && st::sk_make_plan((int)tiles, K / BK, slots, plan)) {
      static std::atomic<unsigned> g_epoch{0};
      unsigned epoch = g_epoch.fetch_add(1, std::memory_order_relaxed) + 1;
      if (epoch == 0) epoch = g_epoch.fetch_add(1, std::memory_order_relaxed) + 1;
      BinsParams q{};
      q.A = A; q.B = B; q.C = C;
      q.lda = lda; q.ldb = b_transposed ? K : N; q.ldc = ldc;
      q.a_batch = a_batch; q.b_batch = b_batch; q.c_batch = c_batch;
      q.tiles_m = M / 64; q.tiles_n = N / 128;
      q.plan = plan;
      q.ctrl = reinterpret_cast<unsigned*>(sk_ws);
      q.partial = sk_ws + st::SK_CTRL_WORDS;
      q.epoch = epoch;
      q.pub_epoch = st::tuning(st::TUNE_STREAMK_TEST_DROP) ? (epoch ^ 0x80000000u) : epoch;   // (test hook: producers that never arrive)
      st::trace("gemm_nn_bins<64,128,2,2%s> batched bins=%d M=%d Np=%d Kp=%d streamk wgs=%d upw=%d gflop=%.3f", b_transposed ? ",bt" : "",
                batches, M, N, K, 8 * plan.wgs_per_xcd, plan.upw, 2e-9 * M * (double)N * K * batches);
      {
        st::LaunchTimer timer(s);
        const dim3 grid(8 * plan.wgs_per_xcd), block(NTHREADS);
        if (b_transposed) {
          if (plan.wgs_per_xcd > 64) st::launch_timed(timer, gemm_nn_bins_kernel<64, 128, 2, 2, 3, true>, grid, block, s, q);
          else st::launch_timed(timer, gemm_nn_bins_kernel<64, 128, 2, 2, 2, true>, grid, block, s, q);
        } else {
          if (plan.wgs_per_xcd > 64) st::launch_timed(timer, gemm_nn_bins_kernel<64, 128, 2, 2, 3, false>, grid, block, s, q);
          else st::launch_timed(timer, gemm_nn_bins_kernel<64, 128, 2, 2, 2, false>, grid, block, s, q);
        }
      }
      return st::check_launch("gemm_nn_bins");
    }
  }
  // Rows that fill whole 128-row tiles but for a last HALF tile (the planes of a bin are padded to 64 rows: 32 utterances of 9
  // blocks are 288 -> 320 rows): the 128-row kernel would walk 384.  The whole tiles go to it, the last 64 rows to the 64-row
  // kernel as a launch of their own (same streams of operands, plain row-major matrices: only the row offset differs).
  // Measured on the 32-tap layer at 32 x 11 s (rows 320): forward 625 us / back-prop 762 us in one launch of three row tiles.
  // Only the wide layer's products (K x N of 512 x 4096 or 4096 x 512 per bin: 100+ us of matrix work in the 64 rows saved); the
  // narrow layers' launches are short and would pay a second launch each.  Same box, no_row_split 1 -> 0: the step at 32 x 6 s
  // 5.26 -> 5.18 ms, 11 s 8.72 -> 8.52, 12 s 9.13 -> 8.92; bucketed training +0.7 % (profiles/r6_row_split_ab.txt).
  if (M % 128 == 64 && M > 128 && (long)K * N >= (1L << 21) && (long)st::ceil_div(M, 128) * (N / 128) * batches >= 512 &&
      st::tuning(st::TUNE_GEMM_TILE) == 0 && st::tuning(st::TUNE_NO_ROW_SPLIT) == 0) {
    if (int e = st::gemm_nn_batched(A, lda, a_batch, B, b_batch, C, ldc, c_batch, M - 64, K, N, batches, s, nullptr, b_transposed)) return e;
    return st::gemm_nn_batched(A + (long)(M - 64) * lda, lda, a_batch, B, b_batch, C + (long)(M - 64) * ldc, ldc, c_batch, 64, K, N, batches, s,
                               nullptr, b_transposed);
  }
  NNParams p{};
  p.A = A;
  p.amap.frames = M; p.amap.row_stride = (int)lda; p.amap.batch_stride = 0; p.amap.row0 = 0;
  p.Bm = B;
  p.Np = N;
  p.C = C;
  p.cmap.frames = M; p.cmap.row_stride = (int)ldc; p.cmap.batch_stride = 0; p.cmap.row0 = 0;
  p.M = M;
  p.Kvalid = K;
  p.Kp = K;
  p.n_store = N;
  p.taps = 1;
  p.cp = K;
  p.batches = batches;
  p.a_batch = a_batch; p.b_batch = b_batch; p.c_batch = c_batch;
  if (b_transposed) { p.bt_ld = K; p.bt_rows = N; }
  return run_nn(p, 0, s);
}

// Plain batched out[b] = A[b]^T * Z[b] on the filter-gradient kernel: A [M][lda] (K <= lda columns used), Z [M][ldz]
// (N columns), out [K][N]; the reduction runs over the M rows.  K a multiple of 128, N of 128, M of 32.
int st::gemm_tn_batched(const float* A, long lda, long a_batch, const float* Z, long ldz, long z_batch, float* out,
                        long o_batch, int M, int K, int N, int batches, hipStream_t s, int z_batch_shift) {
  if (!(A && Z && out && M > 0 && M % 32 == 0 && K % 128 == 0 && N % 128 == 0 && batches > 0)) {
    st::set_error("gemm_tn_batched: bad shape M=%d K=%d N=%d", M, K, N);
    return ST_EINVAL;
  }
  TNParams p{};
  p.A = A;
  p.amap.frames = M; p.amap.row_stride = (int)lda; p.amap.batch_stride = 0; p.amap.row0 = 0;
  p.Z = Z;
  p.zmap.frames = M; p.zmap.row_stride = (int)ldz; p.zmap.batch_stride = 0; p.zmap.row0 = 0;
  p.M = M;
  p.Kvalid = K;
  p.Kp = K;
  p.Np = N;
  p.z_cols = N;
  p.rows_per_split = M;
  p.out = out;
  p.tiles_k = K / 128;
  p.tiles_n = N / 128;
  p.amap_batches = 1;
  p.adv_b = 32 / M;
  p.adv_t = 32 % M;
  p.a_batch = a_batch; p.z_batch = z_batch; p.o_batch = o_batch;
  p.z_shift = z_batch_shift;
  st::trace("gemm_tn<128> batched bins=%d M=%d Kp=%d Np=%d gflop=%.3f", batches, M, K, N, 2e-9 * M * (double)K * N * batches);
  {
    st::LaunchTimer timer(s);
    st::launch_timed(timer, gemm_tn_kernel<128, 2, 2>, dim3(p.tiles_k * p.tiles_n, 1, batches), dim3(TN_THREADS), s, p);
  }
  return st::check_launch("gemm_tn_batched");
}

// A bin's complex product in three real products (gemm_nn_g3_kernel): C[bin][m][n] = sum_k A0 B0 + A1 B1, C[bin][m][c_off2 + n] =
// sum_k A0 B0 + A2 B2 with A_p = A + a_off[p] (K readable floats of row m from there), B_p = B + b_off[p] -- [K][ldb] row-major, or
// with b_transposed [N][ldb] holding B_p^T.  K a multiple of 64, N of 128.  A last half tile of rows (M % 128 == 64) runs on 64-row
// tiles as a launch of its own, like gemm_nn_batched's.
int st::gemm_nn_g3_batched(const float* A, long lda, long a_batch, const long a_off[3], const float* B, long ldb, long b_batch,
                           const long b_off[3], float* C, long ldc, long c_batch, long c_off2, int M, int K, int N, int batches,
                           hipStream_t s, bool b_transposed) {
  if (!(A && B && C && M > 0 && K > 0 && K % 64 == 0 && N % 128 == 0 && batches > 0 && lda % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 &&
        c_off2 % 4 == 0 && a_off[0] % 4 == 0 && a_off[1] % 4 == 0 && a_off[2] % 4 == 0 && b_off[0] % 4 == 0 && b_off[1] % 4 == 0 && b_off[2] % 4 == 0)) {
    st::set_error("gemm_nn_g3_batched: bad shape M=%d K=%d N=%d", M, K, N);
    return ST_EINVAL;
  }
  if (M % 128 == 64 && M > 128 && (long)batches * st::ceil_div(M, 128) * (N / 128) >= 512 && st::tuning(st::TUNE_G3_TILE) == 0 &&
      st::tuning(st::TUNE_NO_ROW_SPLIT) == 0) {                    // (a launch that takes 64-row tiles anyway stays whole)
    if (int e = st::gemm_nn_g3_batched(A, lda, a_batch, a_off, B, ldb, b_batch, b_off, C, ldc, c_batch, c_off2, M - 64, K, N, batches, s, b_transposed)) return e;
    return st::gemm_nn_g3_batched(A + (long)(M - 64) * lda, lda, a_batch, a_off, B, ldb, b_batch, b_off, C + (long)(M - 64) * ldc, ldc, c_batch, c_off2,
                                  64, K, N, batches, s, b_transposed);
  }
  G3Params p{};
  p.A = A; p.lda = lda; p.a_batch = a_batch;
  p.B = B; p.ldb = ldb; p.b_batch = b_batch;
  for (int i = 0; i < 3; ++i) { p.a_off[i] = a_off[i]; p.b_off[i] = b_off[i]; }
  p.C = C; p.ldc = ldc; p.c_batch = c_batch; p.c_off2 = c_off2;
  p.M = M; p.N = N; p.batches = batches;
  // 128-row tiles while they give every CU its two workgroups; else 64-row tiles (three resident per CU), and if those are still
  // fewer than two per CU the reduction is split in two (the 32-tap layer's back-prop at config 2: 4 x 48 = 192 tiles of 128 rows
  // -> 768 workgroups of 64 rows x half the reduction, three per CU; measured 423 -> see DESIGN)
  const long wgs128 = (long)batches * st::ceil_div(M, 128) * (N / 128), wgs64 = (long)batches * st::ceil_div(M, 64) * (N / 128);
  const int tile = st::tuning(st::TUNE_G3_TILE);                    // 1: 64-row tiles, 2: 128-row tiles, 4: 64 rows + split reduction
  const bool half_rows = tile == 2 ? false : (tile == 1 || tile == 4 || M <= 64 || wgs128 < 512);
  p.ksplit = (half_rows && K % 128 == 0 && (tile == 4 || (tile == 0 && wgs64 < 512))) ? 2 : 1;
  p.K = K / p.ksplit;
  p.tiles_m = st::ceil_div(M, half_rows ? 64 : 128);
  p.tiles_n = N / 128;
  if (p.ksplit > 1)          // the tiles are ADDED into C: both column ranges of the M rows of every bin zeroed first
    hipLaunchKernelGGL(g3_zero_out_kernel, dim3(st::ceil_div(M, 4), batches), dim3(256), 0, s, C, ldc, c_batch, c_off2, M, N);
  const dim3 grid(st::ceil_div(batches, 8) * p.tiles_m * p.tiles_n * 8, p.ksplit), block(NTHREADS);
  st::trace("gemm_nn_g3<%d%s> batched bins=%d M=%d Np=%d Kp=%d ksplit=%d gflop=%.3f", half_rows ? 64 : 128, b_transposed ? ",bt" : "", batches, M, N, K,
            p.ksplit, 2e-9 * p.tiles_m * (half_rows ? 64 : 128) * (double)N * 3 * K * batches);
  st::LaunchTimer timer(s);
  if (half_rows) {
    if (b_transposed) st::launch_timed(timer, gemm_nn_g3_kernel<64, true>, grid, block, s, p);
    else st::launch_timed(timer, gemm_nn_g3_kernel<64, false>, grid, block, s, p);
  } else {
    if (b_transposed) st::launch_timed(timer, gemm_nn_g3_kernel<128, true>, grid, block, s, p);
    else st::launch_timed(timer, gemm_nn_g3_kernel<128, false>, grid, block, s, p);
  }
  return st::check_launch("gemm_nn_g3");
}

// The lag products of a bin in three real products (gemm_tn_g3_kernel): out[bin][0] = A0^T Z0 + A1^T Z1, out[bin][1] = A2^T Z2 - A0^T Z0
// ([K][N] each, o_part floats apart), the reduction over the M rows; A_p = A + a_off[p], Z_p = Z + z_off[p].  M a multiple of 32,
// K and N of 128.
int st::gemm_tn_g3_batched(const float* A, long lda, long a_batch, const long a_off[3], const float* Z, long ldz, long z_batch,
                           const long z_off[3], float* out, long o_batch, long o_part, int M, int K, int N, int batches, hipStream_t s) {
  if (!(A && Z && out && M > 0 && M % 32 == 0 && K % 128 == 0 && N % 128 == 0 && batches > 0 && lda % 4 == 0 && ldz % 4 == 0 &&
        (long)32 * lda < (1L << 29) && (long)32 * ldz < (1L << 29))) {
    st::set_error("gemm_tn_g3_batched: bad shape M=%d K=%d N=%d", M, K, N);
    return ST_EINVAL;
  }
  TN3Params p{};
  p.A = A; p.lda = lda; p.a_batch = a_batch;
  p.Z = Z; p.ldz = ldz; p.z_batch = z_batch;
  for (int i = 0; i < 3; ++i) { p.a_off[i] = a_off[i]; p.z_off[i] = z_off[i]; }
  p.out = out; p.o_batch = o_batch; p.o_part = o_part;
  p.M = M; p.K = K; p.N = N; p.tiles_k = K / 128; p.tiles_n = N / 128; p.batches = batches;
  st::trace("gemm_tn_g3<128> batched bins=%d M=%d Kp=%d Np=%d gflop=%.3f", batches, M, K, N, 2e-9 * 3 * M * (double)K * N * batches);
  st::LaunchTimer timer(s);
  st::launch_timed(timer, gemm_tn_g3_kernel, dim3(st::ceil_div(batches, 8) * p.tiles_k * p.tiles_n * 8), dim3(TN_THREADS), s, p);
  return st::check_launch("gemm_tn_g3_batched");
}

namespace {

bool tensor_ok(const st_tensor3* t) {
  return t && t->base && t->batch > 0 && t->frames > 0 && t->channels > 0 && t->halo >= 0 &&
         t->c_pitch % 16 == 0 && t->c_pitch >= t->channels && t->t_pitch >= t->halo + t->frames;
}

}  // namespace

static int bwd_data_impl(const st_tensor3* dz, const float* packed_t, bool forward_filters, int width, int pad_left,
                         const st_tensor3* act, const st_tensor3* dx, float* dbias_dx, void* workspace,
                         size_t workspace_bytes, void* stream);

extern "C" {

int st_streamk_lost_ptr(void** device_word) {
  ST_REQUIRE(device_word, "st_streamk_lost_ptr: null argument");
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_sk_lost)) != hipSuccess || !p) {
    st::set_error("st_streamk_lost_ptr: hipGetSymbolAddress failed");
    return ST_ELAUNCH;
  }
  *device_word = p;
  return ST_OK;
}

int st_streamk_lost_fetch_async(unsigned* host_word, void* stream) {
  ST_REQUIRE(host_word, "st_streamk_lost_fetch_async: null argument");
  if (hipMemcpyFromSymbolAsync(host_word, HIP_SYMBOL(g_sk_lost), sizeof(unsigned), 0, hipMemcpyDeviceToHost, st::as_stream(stream)) !=
      hipSuccess) {
    st::set_error("st_streamk_lost_fetch_async: hipMemcpyFromSymbolAsync failed");
    return ST_ELAUNCH;
  }
  return ST_OK;
}

int st_streamk_lost_count(unsigned* count) {
  ST_REQUIRE(count, "st_streamk_lost_count: null argument");
  if (hipMemcpyFromSymbol(count, HIP_SYMBOL(g_sk_lost), sizeof(unsigned), 0, hipMemcpyDeviceToHost) != hipSuccess) {
    st::set_error("st_streamk_lost_count: hipMemcpyFromSymbol failed");
    return ST_ELAUNCH;
  }
  return ST_OK;
}

int st_packed_dims(int width, int cin_pitch, int cout, int* k_valid, int* k_pad, int* n_pad) {
  ST_REQUIRE(width > 0 && cin_pitch > 0 && cin_pitch % 16 == 0 && cout > 0, "st_packed_dims: bad shape");
  int kv = width * cin_pitch;
  if (k_valid) *k_valid = kv;
  if (k_pad) *k_pad = (int)st::round_up(kv, BK);
  if (n_pad) *n_pad = npad_of(cout);
  return ST_OK;
}

int st_fill_f32(float* dst, float value, size_t n, void* stream) {
  if (n == 0) return ST_OK;
  ST_REQUIRE(dst, "st_fill_f32: null");
  int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, st::as_stream(stream), dst, value, n);
  return st::check_launch("fill");
}

int st_zero_regions(const void* regions_device, int n_regions, void* stream) {
  ST_REQUIRE(regions_device && n_regions >= 0 && ((uintptr_t)regions_device & 15) == 0, "st_zero_regions: bad table");
  if (n_regions == 0) return ST_OK;
  hipLaunchKernelGGL(zero_regions_kernel, dim3(n_regions), dim3(256), 0, st::as_stream(stream),
                     reinterpret_cast<const ZeroRegion*>(regions_device));
  return st::check_launch("zero_regions");
}

int st_zero_halos_f32(const st_tensor3* t, void* stream) {
  ST_REQUIRE(tensor_ok(t), "st_zero_halos_f32: bad tensor descriptor");
  const int halo_rows = t->t_pitch - t->frames;
  if (halo_rows == 0) return ST_OK;
  const long total = (long)halo_rows * (t->c_pitch / 4);
  const int bx = (int)std::min<long>((total + 255) / 256, 64);
  hipLaunchKernelGGL(zero_halos_kernel, dim3(bx, t->batch), dim3(256), 0, st::as_stream(stream), t->base, t->frames,
                     t->halo, t->t_pitch, t->c_pitch);
  return st::check_launch("zero_halos");
}

int st_pack_filters_f32(const float* filters, int width, int cin, int cout, int cin_pitch,
                        float* packed, void* stream) {
  ST_REQUIRE(filters && packed && cin <= cin_pitch, "st_pack_filters_f32: bad args");
  int kv, kp, np;
  if (int e = st_packed_dims(width, cin_pitch, cout, &kv, &kp, &np)) return e;
  if (int e = st_fill_f32(packed, 0.f, (size_t)kp * np, stream)) return e;
  long total = (long)width * cin * cout;
  hipLaunchKernelGGL(pack_filters_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     st::as_stream(stream), filters, width, cin, cout, cin_pitch, np, packed, 0);
  return st::check_launch("pack_filters");
}

int st_unpack_filters_f32(const float* packed, int width, int cin, int cout, int cin_pitch,
                          float* filters, void* stream) {
  ST_REQUIRE(filters && packed && cin <= cin_pitch, "st_unpack_filters_f32: bad args");
  int kv, kp, np;
  if (int e = st_packed_dims(width, cin_pitch, cout, &kv, &kp, &np)) return e;
  long total = (long)width * cin * cout;
  hipLaunchKernelGGL(pack_filters_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     st::as_stream(stream), filters, width, cin, cout, cin_pitch, np,
                     const_cast<float*>(packed), 1);
  return st::check_launch("unpack_filters");
}

int st_filters_flip_transpose_f32(const float* packed, int width, int cin, int cout, int cin_pitch,
                                  int cout_pitch, float* packed_t, void* stream) {
  ST_REQUIRE(packed && packed_t && cin <= cin_pitch && cout <= cout_pitch && cout_pitch % 16 == 0,
             "st_filters_flip_transpose_f32: bad args");
  int np = npad_of(cout);
  int kvt, kpt, npt;
  if (int e = st_packed_dims(width, cout_pitch, cin, &kvt, &kpt, &npt)) return e;
  const int pad_rows = kpt - width * cout_pitch;          // < 32 rows of k-padding
  dim3 grid(st::ceil_div(std::max(cout_pitch, pad_rows), 32), st::ceil_div(npt, 32), width + (pad_rows > 0 ? 1 : 0));
  hipLaunchKernelGGL(flip_transpose_kernel, grid, dim3(256), 0, st::as_stream(stream), packed, width,
                     cin, cout, cin_pitch, np, cout_pitch, npt, kpt, packed_t);
  return st::check_launch("flip_transpose");
}

size_t st_conv1d_fwd_ws(const st_tensor3* x, const st_tensor3* y, int width) {
  if (!x || !y) return 0;
  const int np = npad_of(y->channels);
  const int M = y->batch * y->frames;
  const int nk = width > 1 ? st::ceil_div(x->c_pitch, BK) * width : st::ceil_div(x->c_pitch, BK);
  const int splits = fwd_splits(M, np, nk);
  return splits > 1 ? (size_t)splits * M * np * sizeof(float) : 0;
}

int st_conv1d_nwc_fwd_f32(const st_tensor3* x, const float* packed, const float* bias, int width,
                          int stride, int pad_left, int relu, const st_tensor3* y, void* stream) {
  return st_conv1d_nwc_fwd_ws_f32(x, packed, bias, width, stride, pad_left, relu, y, nullptr, 0, stream);
}

int st_conv1d_nwc_fwd_ws_f32(const st_tensor3* x, const float* packed, const float* bias, int width,
                             int stride, int pad_left, int relu, const st_tensor3* y, void* workspace,
                             size_t workspace_bytes, void* stream) {
  ST_REQUIRE(tensor_ok(x) && tensor_ok(y) && packed, "conv fwd: bad tensor descriptor");
  ST_REQUIRE(width > 0 && stride > 0 && pad_left >= 0 && x->batch == y->batch, "conv fwd: bad shape");
  ST_REQUIRE(y->frames == st::ceil_div(x->frames, stride), "conv fwd: y.frames != ceil(x.frames/stride)");
  ST_REQUIRE(x->halo >= pad_left, "conv fwd: x.halo %d < pad_left %d", x->halo, pad_left);
  ST_REQUIRE((y->frames - 1) * stride + width - pad_left <= x->t_pitch - x->halo,
             "conv fwd: trailing halo of x too small");
  NNParams p{};
  p.A = x->base;
  p.amap = make_map(*x, x->halo - pad_left, stride, y->frames);
  p.Bm = packed;
  p.Np = npad_of(y->channels);
  p.C = y->base;
  p.cmap = make_map(*y, y->halo, 1, y->frames);
  p.bias = bias;
  p.M = y->batch * y->frames;
  p.Kvalid = width * x->c_pitch;
  p.Kp = (int)st::round_up(p.Kvalid, BK);
  p.n_store = std::min(y->c_pitch, p.Np);
  p.relu = relu;
  p.taps = width;
  p.cp = x->c_pitch;
  // few output rows (live / single-utterance inference): spread the reduction over the idle CUs
  const int nk = p.taps > 1 ? st::ceil_div(p.cp, BK) * p.taps : p.Kp / BK;
  const int splits = fwd_splits(p.M, p.Np, nk);
  if (splits > 1 && workspace && workspace_bytes >= (size_t)splits * p.M * p.Np * sizeof(float)) {
    p.steps_per_split = st::ceil_div(nk, splits);
    p.splits = st::ceil_div(nk, p.steps_per_split);
    p.slab = reinterpret_cast<float*>(workspace);
  }
  return run_nn(p, 0, st::as_stream(stream));
}

size_t st_conv1d_bwd_data_ws(const st_tensor3* dz, const st_tensor3* dx, int width) {
  if (!dz || !dx) return 0;
  const int np = npad_of(dx->channels), kp = (int)st::round_up((size_t)width * dz->c_pitch, BK);
  const int M = dx->batch * dx->frames;
  const int splits = nn_splits(M, np, kp);
  return splits > 1 ? (size_t)splits * M * np * sizeof(float) : 0;
}

int st_conv1d_nwc_bwd_data_f32(const st_tensor3* dz, const float* packed_t, int width, int pad_left,
                               const st_tensor3* act, const st_tensor3* dx, void* workspace,
                               size_t workspace_bytes, void* stream) {
  return st_conv1d_nwc_bwd_data_bias_f32(dz, packed_t, width, pad_left, act, dx, nullptr, workspace, workspace_bytes, stream);
}

size_t st_conv1d_bwd_data_bias_ws(const st_tensor3* dz, const st_tensor3* dx, int width) {
  if (!dz || !dx) return 0;
  const int np = npad_of(dx->channels);
  const size_t rows = (size_t)std::max(st::ceil_div(dx->batch * dx->frames, 64) * 4, dx->batch * st::ceil_div(dx->frames, COLSUM_ROWS));
  return st::round_up(st_conv1d_bwd_data_ws(dz, dx, width), 256) + rows * np * sizeof(float) + 256;
}

int st_conv1d_nwc_bwd_data_bias_f32(const st_tensor3* dz, const float* packed_t, int width, int pad_left,
                                    const st_tensor3* act, const st_tensor3* dx, float* dbias_dx, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  return bwd_data_impl(dz, packed_t, false, width, pad_left, act, dx, dbias_dx, workspace, workspace_bytes, stream);
}

int st_conv1d_1tap_bwd_data_bias_f32(const st_tensor3* dz, const float* packed, const st_tensor3* act, const st_tensor3* dx,
                                     float* dbias_dx, void* workspace, size_t workspace_bytes, void* stream) {
  ST_REQUIRE(tensor_ok(dz) && tensor_ok(dx) && dz->c_pitch % 32 == 0 && npad_of(dx->channels) % 128 == 0 &&
                 npad_of(dz->channels) >= dz->c_pitch,
             "conv 1-tap bwd_data: needs a channel pitch of dz that is a multiple of 32 and an input width that packs to 128s");
  return bwd_data_impl(dz, packed, true, 1, 0, act, dx, dbias_dx, workspace, workspace_bytes, stream);
}

}  // extern "C"

// back-prop to the input: dz (*) flipped / transposed filters (packed_t), or -- one tap, `forward_filters` -- dz W^T with the
// layer's own packed filters [cin_pitch][n_pad(cout)] read as a transposed operand
static int bwd_data_impl(const st_tensor3* dz, const float* packed_t, bool forward_filters, int width, int pad_left,
                         const st_tensor3* act, const st_tensor3* dx, float* dbias_dx, void* workspace,
                         size_t workspace_bytes, void* stream) {
  ST_REQUIRE(tensor_ok(dz) && tensor_ok(dx) && packed_t, "conv bwd_data: bad tensor descriptor");
  ST_REQUIRE(dz->batch == dx->batch && dz->frames == dx->frames, "conv bwd_data: stride-1 layers only");
  const int lead = width - 1 - pad_left;   // zero rows needed in front of dz frame 0
  ST_REQUIRE(lead >= 0 && dz->halo >= lead, "conv bwd_data: dz.halo %d < %d", dz->halo, lead);
  ST_REQUIRE(dz->frames + pad_left <= dz->t_pitch - dz->halo, "conv bwd_data: trailing halo of dz too small");
  if (act) ST_REQUIRE(tensor_ok(act) && act->batch == dx->batch && act->frames == dx->frames &&
                      act->c_pitch >= std::min(dx->c_pitch, npad_of(dx->channels)),
                      "conv bwd_data: mask tensor mismatch");
  NNParams p{};
  p.A = dz->base;
  p.amap = make_map(*dz, dz->halo - lead, 1, dx->frames);
  p.Bm = packed_t;
  p.Np = npad_of(dx->channels);
  p.C = dx->base;
  p.cmap = make_map(*dx, dx->halo, 1, dx->frames);
  if (act) {
    p.mask = act->base;
    p.mmap = make_map(*act, act->halo, 1, act->frames);
  }
  p.M = dx->batch * dx->frames;
  p.Kvalid = width * dz->c_pitch;
  p.Kp = (int)st::round_up(p.Kvalid, BK);
  p.n_store = std::min(dx->c_pitch, p.Np);
  p.taps = width;
  p.cp = dz->c_pitch;
  if (forward_filters) {                       // Bt[n = input channel][k = output channel]: rows of the forward operand
    p.bt_ld = npad_of(dz->channels);
    p.bt_rows = dx->c_pitch;
  }
  // long reductions on few output tiles (L8: K = 64000, N = 256) are split over K; needs the workspace
  const int splits = nn_splits(p.M, p.Np, p.Kp);
  if (splits > 1 && workspace && workspace_bytes >= st_conv1d_bwd_data_ws(dz, dx, width)) {
    const int nk = p.taps > 1 ? st::ceil_div(p.cp, BK) * p.taps : p.Kp / BK;
    p.splits = splits;
    p.steps_per_split = st::ceil_div(nk, splits);
    p.splits = st::ceil_div(nk, p.steps_per_split);
    p.slab = reinterpret_cast<float*>(workspace);
  }
  if (!dbias_dx) return run_nn(p, 1, st::as_stream(stream));
  // bias gradient of the layer below = column sums of dx.  One-pass launches collect them in the epilogue of the
  // kernel that writes dx (no second read of up to 129 MB); split-K launches finish in splitk_epilogue_kernel, so
  // there the two-level column sum reads dx back.
  ST_REQUIRE(workspace && workspace_bytes >= st_conv1d_bwd_data_bias_ws(dz, dx, width), "conv bwd_data: workspace too small for the bias gradient");
  hipStream_t s = st::as_stream(stream);
  float* partial = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + st::round_up(st_conv1d_bwd_data_ws(dz, dx, width), 256));
  if (p.splits <= 1) p.colsum = partial;
  if (int e = run_nn(p, 1, s)) return e;
  if (p.splits <= 1) {
    hipLaunchKernelGGL(colsum_final_kernel, dim3(st::ceil_div(p.Np, 32)), dim3(256), 0, s, partial, p.colsum_rows, p.Np, dbias_dx);
  } else {
    const int chunks = st::ceil_div(dx->frames, COLSUM_ROWS);
    const RowMap xmap = make_map(*dx, dx->halo, 1, dx->frames);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(st::ceil_div(p.Np, 128), chunks, dx->batch), dim3(256), 0, s, dx->base,
                       xmap.batch_stride, xmap.row0, xmap.row_stride, dx->frames, dx->channels, dx->c_pitch, partial, p.Np);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(st::ceil_div(p.Np, 32)), dim3(256), 0, s, partial, chunks * dx->batch, p.Np,
                       dbias_dx);
  }
  return st::check_launch("bwd_data bias gradient");
}

extern "C" {

static int bwd_filter_splits(int M, int kp, int np) {
  // Row-range splits of the M reduction.  The kernel runs 2 workgroups per CU (512 slots): pick the
  // split count whose total workgroup count fills whole rounds of 512 (1 to 4 rounds, the fewest rounds that
  // fill best: every extra split writes and re-reads another partial tile), each split keeping at least 256
  // rows.  Measured on L0/L1/L9: a ragged last round costs 10-20 %; L1 with 18 instead of 36 splits 0.149 ->
  // 0.140 ms.
  const int tiles = st::ceil_div(kp, 128) * (np / (np % 128 == 0 ? 128 : np));
  const int max_splits = std::max(1, M / 256);
  if (tiles >= 512) return 1;
  int best = 1;
  double best_eff = 0.0;
  for (int rounds = 1; rounds <= 4; ++rounds) {
    const int sp = std::min(max_splits, std::max(1, rounds * 512 / tiles));
    const int wgs = tiles * sp;
    const double eff = (double)wgs / (st::ceil_div(wgs, 512) * 512.0);
    if (eff > best_eff + 0.01) { best_eff = eff; best = sp; }
  }
  return best;
}

size_t st_bias_grad_ws(const st_tensor3* dz) {
  if (!dz) return 0;
  return (size_t)dz->batch * st::ceil_div(dz->frames, COLSUM_ROWS) * npad_of(dz->channels) * sizeof(float) + 256;
}

int st_bias_grad_f32(const st_tensor3* dz, float* dbias, void* workspace, size_t workspace_bytes, void* stream) {
  ST_REQUIRE(tensor_ok(dz) && dbias && workspace && workspace_bytes >= st_bias_grad_ws(dz), "bias_grad: bad args");
  hipStream_t s = st::as_stream(stream);
  const int np = npad_of(dz->channels);
  const int chunks = st::ceil_div(dz->frames, COLSUM_ROWS);
  float* partial = reinterpret_cast<float*>(workspace);
  const RowMap zmap = make_map(*dz, dz->halo, 1, dz->frames);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(st::ceil_div(np, 128), chunks, dz->batch), dim3(256), 0, s, dz->base,
                     zmap.batch_stride, zmap.row0, zmap.row_stride, dz->frames, dz->channels, dz->c_pitch, partial, np);
  hipLaunchKernelGGL(colsum_final_kernel, dim3(st::ceil_div(np, 32)), dim3(256), 0, s, partial, chunks * dz->batch, np, dbias);
  return st::check_launch("bias_grad");
}

size_t st_conv1d_bwd_filter_ws(const st_tensor3* x, const st_tensor3* dz, int width) {
  if (!x || !dz) return 0;
  int kp = (int)st::round_up((size_t)width * x->c_pitch, BK), np = npad_of(dz->channels);
  int M = dz->batch * dz->frames;
  int splits = bwd_filter_splits(M, kp, np);
  size_t slabs = splits > 1 ? (size_t)splits * kp * np * sizeof(float) : 0;
  size_t colsum = (size_t)dz->batch * st::ceil_div(dz->frames, COLSUM_ROWS) * np * sizeof(float);
  return slabs + colsum + 256;
}

int st_conv1d_nwc_bwd_filter_f32(const st_tensor3* x, const st_tensor3* dz, int width, int stride,
                                 int pad_left, float* dpacked, float* dbias, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  ST_REQUIRE(tensor_ok(x) && tensor_ok(dz) && dpacked, "conv bwd_filter: bad tensor descriptor");
  ST_REQUIRE(x->batch == dz->batch && dz->frames == st::ceil_div(x->frames, stride), "conv bwd_filter: bad shape");
  ST_REQUIRE(x->halo >= pad_left && (dz->frames - 1) * stride + width - pad_left <= x->t_pitch - x->halo,
             "conv bwd_filter: halo of x too small");
  ST_REQUIRE(workspace_bytes >= st_conv1d_bwd_filter_ws(x, dz, width) && workspace, "conv bwd_filter: workspace too small");
  hipStream_t s = st::as_stream(stream);
  TNParams p{};
  p.A = x->base;
  p.amap = make_map(*x, x->halo - pad_left, stride, dz->frames);
  p.Z = dz->base;
  p.zmap = make_map(*dz, dz->halo, 1, dz->frames);
  p.M = dz->batch * dz->frames;
  p.Kvalid = width * x->c_pitch;
  p.Kp = (int)st::round_up(p.Kvalid, BK);
  p.Np = npad_of(dz->channels);
  p.z_cols = std::min(dz->c_pitch, p.Np);
  const int splits = bwd_filter_splits(p.M, p.Kp, p.Np);
  p.rows_per_split = (int)st::round_up(st::ceil_div(p.M, splits), 32);
  const int used = st::ceil_div(p.M, p.rows_per_split);
  float* slabs = reinterpret_cast<float*>(workspace);
  p.out = used > 1 ? slabs : dpacked;
  p.tiles_k = st::ceil_div(p.Kp, 128);
  p.amap_batches = dz->batch;
  p.adv_b = 32 / dz->frames;
  p.adv_t = 32 % dz->frames;
  st::trace("gemm_tn<%d> slabs=%d rows_per_slab=%d M=%d Kp=%d Np=%d gflop=%.3f", p.Np % 128 == 0 ? 128 : p.Np, used,
            p.rows_per_split, p.M, p.Kp, p.Np, 2e-9 * st::round_up(p.M, 32) * (double)(p.tiles_k * 128) * p.Np);
  {
    st::LaunchTimer timer(s);                // the product kernel alone (what rocprofv3 lists under this symbol)
    if (p.Np % 128 == 0) {
      p.tiles_n = p.Np / 128;
      st::launch_timed(timer, gemm_tn_kernel<128, 2, 2>, dim3(p.tiles_k * p.tiles_n, used), dim3(TN_THREADS), s, p);
    } else if (p.Np == 64) {
      p.tiles_n = 1;
      st::launch_timed(timer, gemm_tn_kernel<64, 2, 2>, dim3(p.tiles_k, used), dim3(TN_THREADS), s, p);
    } else if (p.Np == 32) {
      p.tiles_n = 1;
      st::launch_timed(timer, gemm_tn_kernel<32, 4, 1>, dim3(p.tiles_k, used), dim3(TN_THREADS), s, p);
    } else {
      st::set_error("conv bwd_filter: unsupported n_pad=%d", p.Np);
      return ST_EINVAL;
    }
  }
  if (int e = st::check_launch("gemm_tn")) return e;
  if (used > 1) {
    long n4 = (long)p.Kp * p.Np / 4;
    int blocks = (int)std::min<long>((n4 + 255) / 256, 2048);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, s, slabs, dpacked, n4, used);
    if (int e = st::check_launch("slab_reduce")) return e;
  }
  if (dbias) {
    size_t slab_bytes = splits > 1 ? (size_t)splits * p.Kp * p.Np * sizeof(float) : 0;
    float* partial = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + st::round_up(slab_bytes, 256));
    const int chunks = st::ceil_div(dz->frames, COLSUM_ROWS);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(st::ceil_div(p.Np, 128), chunks, dz->batch), dim3(256), 0, s,
                       dz->base, p.zmap.batch_stride, p.zmap.row0, p.zmap.row_stride, dz->frames, dz->channels,
                       dz->c_pitch, partial, p.Np);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(st::ceil_div(p.Np, 32)), dim3(256), 0, s, partial,
                       chunks * dz->batch, p.Np, dbias);
    if (int e = st::check_launch("colsum")) return e;
  }
  return ST_OK;
}

}  // extern "C"
