// Filter gradient of a stride-1 layer with bf16 activations, straight from the padded NWC planes (BASELINE config 4 arithmetic):
//
//   dF[w * cp + c][n] = sum_q X[q + w + xoff][c] * dZ[q][n]          (speech_model.py:78, gradient of tf.nn.conv1d wrt filters)
//
// q runs over the FLAT rows of the gradient tensor -- utterances one behind the other, halo rows included: they are zero, and for
// a stride-1 'SAME' layer the input and the gradient tensor have the same frame pitch (T + W - 1), so one shift per tap maps a
// gradient row to its input row for every utterance at once.  Both operands are reduction-MAJOR in memory (a row is one value of
// the reduction index), but v_mfma_f32_32x32x16_bf16 wants eight consecutive reduction values per lane.  Rounds 1-4 therefore made
// reduction-minor copies first (transpose_bf16_kernel: 13 launches and 0.38 ms of pure layout work per step).
// gfx950 has the instruction for exactly this: ds_read_b64_tr_b16 reads, per 16-lane group, sixteen 8-byte pieces and hands lane
// i the i-th 16-bit COLUMN of the 4 x 16 matrix they form (measured with scripts/ubench/tr16_probe.hip: result j of lane i is
// element i % 4 of the piece lane 4 j + i / 4 addressed).  With lane 4 j + p pointing at channels [4 p, 4 p + 4) of row j, lane i
// receives rows 0..3 of channel i -- four consecutive reduction values -- so the tiles are staged as they lie in HBM (LDS-DMA,
// 256-byte rows) and transposed on their way into the registers: two reads per 32 x 16 fragment, no copy, no extra pass.
//
// Tile 128 (input channels of one tap) x 128 (output channels), four waves of 64 x 64, 32 reduction rows per stage in a ring of
// four 16 KB stages (two workgroups per CU) -- eight when the grid gives a CU one workgroup anyway (seven stages in flight hide
// more of the fabric latency) --, counted vmcnt + one barrier per stage, fragments read one stage ahead of their MFMAs.  LDS image of a stage: [32 rows][16 chunks of
// 16 bytes], physical chunk = chunk ^ (4 * (row & 3)) applied on the SOURCE side of the DMA: the four rows a transposing read
// touches land in four different quarter-rows, so the 32 lanes of a half-wave cover all 64 banks exactly once.
// The reduction is cut into `splits` contiguous runs of stages whose fp32 slabs wgrad_tr_finish_kernel sums in a fixed order
// together with the bias gradient (column sums of dZ, colsum_bf16_partial_kernel): deterministic, no atomics.
//
// Contract with the caller (st_conv1d_nwc_bwd_filter_tr_bf16): both planes are READABLE and ZERO for 40 rows behind their last
// row (the last stage and the last taps read past the end; what they read meets zero gradient rows).
#include <algorithm>
#include <type_traits>

#include "st_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int TM = 128, TN = 128, TK = 32;         // tile: input channels x output channels x reduction rows per stage
constexpr int STAGE_ELEMS = TK * 128;              // bf16 elements of one operand's stage (32 rows x 256 bytes)
constexpr int SLACK_ROWS = 40;

struct TrParams {
  const unsigned short* X;       // bf16 plane of the layer input, flat rows of x_cp elements
  const unsigned short* Z;       // bf16 plane of the gradient wrt the layer output, flat rows of z_cp elements
  float* out;                    // [splits][width * x_cp][n_pad]
  long x_row0, z_row0;           // flat rows of tap 0 / of the gradient at reduction index 0
  long slab_stride;
  long lag_im_offset;            // LAG: floats from a bin's real product to its imaginary one
  int x_cp, z_cp, n_pad, width;
  int stages, stages_per_split, splits;
  int mtiles_per_tap, tiles_m, tiles_n;
  int map, map_a, map_b;         // workgroup -> (split, tile) order, see the kernel
  float* bias_direct;            // non-null: the bias gradient rides on row x_cp - 1 of tap 0 (see `bias_tile`) and, unsplit, is stored here
  int bias_in_tile;
};

// ds_read_b64_tr_b16 as inline assembly, OFF = immediate byte offset.  (Through __builtin_amdgcn_ds_read_tr16_b64 the compiler sees
// an LDS read that may alias the LDS-DMA in flight and puts s_waitcnt vmcnt(0) in front of the first read of every stage -- the
// ring then holds one stage, not three.  Here the data dependence is stated by hand: `lds_wait` takes the eight results of a
// k-step as in/out operands, so nothing that uses them can be scheduled in front of the wait.)
typedef unsigned long long u64;
template <int OFF>
__device__ __forceinline__ u64 lds_read_tr16(unsigned addr) {
  u64 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ void lds_wait(u64 (&r)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
               :
               : "memory");
}
// four consecutive reduction rows of one channel, (e0, e1, e2, e3) -> (e1, -e0, e3, -e2) when `on` (the rotated spectra operand)
__device__ __forceinline__ u64 rotate_rows(u64 v, bool on) {
  const unsigned sh = on ? 16u : 0u, flip = on ? 0x80000000u : 0u;
  const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
  const unsigned lo2 = __builtin_amdgcn_alignbit(lo, lo, sh) ^ flip, hi2 = __builtin_amdgcn_alignbit(hi, hi, sh) ^ flip;
  return (u64)lo2 | ((u64)hi2 << 32);
}
__device__ __forceinline__ bf16x8 frag_of(u64 lo, u64 hi) {
  typedef u64 u64x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(bf16x8, u64x2{lo, hi});
}

// LAG: the lag products of a frequency-domain layer's filter gradient (conv_fft.hip, st_conv1d_nwc_bwd_filter_fft_planes with one
// bf16 plane): a "split" is a frequency bin -- its 2 * rows_pad half-rows (a spectra row [re | im] read as two rows of half
// length) are the reduction -- and blockIdx.y picks the real or the imaginary product of the bin:
//   Re Q = S2^T Z2,   Im Q = S'2^T Z2  with  S' = [S_i | -S_r],  i.e. half-row 2 r of S'2 is half-row 2 r + 1 of S2 and half-row
// 2 r + 1 is MINUS half-row 2 r: inside a fragment of four consecutive reduction rows the pairs change places and the odd ones
// change sign -- a 16-bit rotation and a sign-bit flip of each 32-bit word, exact in bf16.  Rounds 4's reduction-minor copies of
// both spectra (transpose_bf16_bins_split, 52 us per step) and the rotated copy they formed are not made any more.
template <bool LAG, int ST>
__global__ __launch_bounds__(256, ST <= 4 ? 2 : 1) void wgrad_tr_bf16_kernel(TrParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned short smem[2 * ST * STAGE_ELEMS];
  unsigned short* const As = smem;                              // [ST][32][128]
  unsigned short* const Bs = smem + ST * STAGE_ELEMS;

  // Workgroup -> (split, tile, real / imaginary).  Workgroups go to the eight XCDs in turn (id & 7), each XCD has its own L2: what
  // shares operand rows must share an XCD, or every L2 fetches the same rows over the fabric (measured with the plain order
  // id = split * tiles + tile: 100 MB per launch for a 7-tap layer's 16.6 MB of operands, 640 for the 2000 x 2000 layer's 131).
  //   map 1 (splits a multiple of 8): an XCD owns whole splits -- every row range is fetched by one L2;
  //   map 2 (8 a multiple of splits): the XCDs of a split take contiguous runs of its tiles;
  //   map 3 (LAG): the workgroups that read one (bin, column tile) panel of the gradient spectra -- both row tiles, real and
  //          imaginary product -- follow each other on one XCD;
  //   map 0: the plain order.
  const int tiles = p.tiles_m * p.tiles_n;
  int split, tile;
  bool rot = false;
  {
    const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
    if (p.map == 1) {
      split = xcd * p.map_a + j % p.map_a;                       // map_a = splits / 8
      tile = j / p.map_a;
    } else if (p.map == 2) {
      split = xcd / p.map_a;                                     // map_a = 8 / splits XCDs per split, map_b tiles each
      tile = (xcd % p.map_a) * p.map_b + j;
      if (j >= p.map_b || tile >= tiles) return;
    } else if (p.map == 3) {
      const int members = 2 * p.tiles_m, g = (j / members) * 8 + xcd, m = j % members;      // g = bin * tiles_n + column tile
      if (g >= p.splits * p.tiles_n) return;
      split = g / p.tiles_n;
      tile = (g - split * p.tiles_n) * p.tiles_m + (m >> 1);
      rot = LAG && (m & 1);                                      // the imaginary product
    } else {
      split = id / tiles;
      tile = id - split * tiles;
      rot = LAG && blockIdx.y == 1;
    }
  }
  const int tn = tile / p.tiles_m, tm = tile - tn * p.tiles_m;   // row tiles fastest: neighbours share the gradient panel
  const int w = tm / p.mtiles_per_tap, c0 = (tm - w * p.mtiles_per_tap) * TM, n0 = tn * TN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // The bias gradient is the column sum of the gradient plane: the product of a row of ONES with it.  The last channel of the
  // channel pitch is padding (x_cp > channels, x_cp a multiple of 128: asked for by the caller) -- its row of the filter gradient is
  // zero by construction and nobody's -- so the wave that owns row x_cp - 1 of tap 0 feeds ones instead of that channel's zeros:
  // the column sums come out of the same MFMAs, in the same slabs, and the separate pass over the plane (7.5 us per layer) is gone.
  const bool bias_wave = !LAG && p.bias_in_tile && w == 0 && c0 + TM == p.x_cp && wm == 1;
  const int kt_begin = split * p.stages_per_split;
  const int nk = min(p.stages_per_split, p.stages - kt_begin);
  if (nk <= 0) return;

  // DMA: wave v stages rows [8 v, 8 v + 8) of both operands, two 1 KB pieces (4 rows x 256 bytes) each; lane i of a piece writes
  // physical chunk i & 15 of row i >> 4 and fetches source chunk (i & 15) ^ (4 * (i >> 4))
  const int prow = lane >> 4, pchunk = (lane & 15) ^ (4 * prow);
  const unsigned short* asrc = p.X + (p.x_row0 + w + (long)kt_begin * TK + wave * 8 + prow) * p.x_cp + c0 + pchunk * 8;
  const unsigned short* bsrc = p.Z + (p.z_row0 + (long)kt_begin * TK + wave * 8 + prow) * p.z_cp + n0 + pchunk * 8;
  const long a_piece = 4L * p.x_cp, b_piece = 4L * p.z_cp, a_stage = (long)TK * p.x_cp, b_stage = (long)TK * p.z_cp;
  auto issue = [&](int kt) {
    const int buf = kt % ST;
    const unsigned short* a = asrc + (long)kt * a_stage;
    const unsigned short* b = bsrc + (long)kt * b_stage;
    unsigned short* la = As + buf * STAGE_ELEMS + wave * 8 * 128;
    unsigned short* lb = Bs + buf * STAGE_ELEMS + wave * 8 * 128;
    __builtin_amdgcn_global_load_lds((gptr_t)a, (lptr_t)la, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(a + a_piece), (lptr_t)(la + 4 * 128), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)b, (lptr_t)lb, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(b + b_piece), (lptr_t)(lb + 4 * 128), 16, 0, 0);
  };

  // fragment addresses (bytes inside a stage): 16-lane group g = lane >> 4 reads rows 8 (g >> 1) + 4 h + (l >> 2), l = lane & 15,
  // channels base + 16 (g & 1) + 4 (l & 3) .. + 3: chunk = base / 8 + 2 (g & 1) + ((l & 3) >> 1), half-chunk l & 1
  const int l16 = lane & 15, g = lane >> 4;
  const int frow = 8 * (g >> 1) + (l16 >> 2);                    // + 4 h + 16 ks
  const int fchunk = 2 * (g & 1) + ((l16 & 3) >> 1);            // + tile base chunk
  const int fswz = 4 * (l16 >> 2);                               // (row & 3) of the rows this lane addresses
  unsigned a_at[2], b_at[2];                                     // LDS byte addresses inside ring slot 0
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ca = ((wm * 64 + i * 32) >> 3) + fchunk, cb = ((wn * 64 + i * 32) >> 3) + fchunk;
    a_at[i] = (unsigned)(size_t)As + (unsigned)(frow * 256 + ((ca ^ fswz) << 4) + (l16 & 1) * 8);
    b_at[i] = (unsigned)(size_t)Bs + (unsigned)(frow * 256 + ((cb ^ fswz) << 4) + (l16 & 1) * 8);
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // One stage BEHIND: the sixteen transposing reads of stage kt are issued after its barrier and return while the eight MFMAs
  // of stage kt - 1 run from the registers filled an iteration earlier (a workgroup alone on its CU used to wait for the first
  // k-step's reads with nothing to do: 0.32 us per stage for 0.11 us of MFMAs).  Ring slot kt - 1 is free for the DMA of stage
  // kt + ST - 1 all the same: its reads were waited for (lds_wait) before this stage's barrier.
  // A wave issues in order: sixteen reads in a row hold the MFMAs behind them back for as long as the LDS takes to accept them, so
  // a stage is issued as eight groups -- one MFMA of the stage in hand, up to three fragment reads of the next stage in its
  // 32-cycle shadow, none under the last two (they cover the latency of the last read) -- with a scheduling fence after each.
  // k-step 0 of a stage: rows 0..15 (reads at +0 and +4 rows); k-step 1: +16 and +20 rows.
  auto step = [&](auto do_mul, auto do_read, u64 (&p0)[8], u64 (&p1)[8], u64 (&n0)[8], u64 (&n1)[8], int kt) {
    constexpr bool M = decltype(do_mul)::value, R = decltype(do_read)::value;
    const unsigned so = (unsigned)((kt % ST) * STAGE_ELEMS * 2);
    const unsigned a0 = a_at[0] + so, a1 = a_at[1] + so, b0 = b_at[0] + so, b1 = b_at[1] + so;
    bf16x8 fa0{}, fa1{}, fb0{}, fb1{};
    auto frags = [&](u64 (&r)[8]) {
      if (LAG) {
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = rotate_rows(r[q], rot);
      }
      if (!LAG && bias_wave && (lane & 31) == 31) r[2] = r[3] = 0x3F803F803F803F80ull;      // channel c0 + 127: four bf16 ones
      fa0 = frag_of(r[0], r[1]), fa1 = frag_of(r[2], r[3]), fb0 = frag_of(r[4], r[5]), fb1 = frag_of(r[6], r[7]);
    };
    if (M) frags(p0);
    if (M) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb0, acc[0][0], 0, 0, 0);
    if (R) n0[0] = lds_read_tr16<0>(a0), n0[1] = lds_read_tr16<1024>(a0), n0[2] = lds_read_tr16<0>(a1);
    __builtin_amdgcn_sched_barrier(0);
    if (M) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb1, acc[0][1], 0, 0, 0);
    if (R) n0[3] = lds_read_tr16<1024>(a1), n0[4] = lds_read_tr16<0>(b0), n0[5] = lds_read_tr16<1024>(b0);
    __builtin_amdgcn_sched_barrier(0);
    if (M) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb0, acc[1][0], 0, 0, 0);
    if (R) n0[6] = lds_read_tr16<0>(b1), n0[7] = lds_read_tr16<1024>(b1), n1[0] = lds_read_tr16<4096>(a0);
    __builtin_amdgcn_sched_barrier(0);
    if (M) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb1, acc[1][1], 0, 0, 0);
    if (R) n1[1] = lds_read_tr16<5120>(a0), n1[2] = lds_read_tr16<4096>(a1), n1[3] = lds_read_tr16<5120>(a1);
    __builtin_amdgcn_sched_barrier(0);
    if (M) frags(p1);
    if (M) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb0, acc[0][0], 0, 0, 0);
    if (R) n1[4] = lds_read_tr16<4096>(b0), n1[5] = lds_read_tr16<5120>(b0);
    __builtin_amdgcn_sched_barrier(0);
    if (M) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb1, acc[0][1], 0, 0, 0);
    if (R) n1[6] = lds_read_tr16<4096>(b1), n1[7] = lds_read_tr16<5120>(b1);
    __builtin_amdgcn_sched_barrier(0);
    if (M) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb0, acc[1][0], 0, 0, 0);
    if (M) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb1, acc[1][1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  const std::true_type yes{};
  const std::false_type no{};
  auto land = [&](int kt) {
    // this wave's pieces of stage kt have landed when at most the younger stages' (4 instructions each) are outstanding
    // (a bare s_barrier: __syncthreads() is a fence too and would drain every DMA in flight -- vmcnt(0) -- each stage)
    if (kt + ST - 1 <= nk) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (ST - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (kt + ST - 1 < nk) issue(kt + ST - 1);                    // into the slot of stage kt - 1: everyone is past reading it
  };

  for (int kt = 0; kt < ST - 1 && kt < nk; ++kt) issue(kt);
  u64 e0[8], e1[8], o0[8], o1[8];                                // fragments of the even / the odd stages
  land(0);
  step(no, yes, o0, o1, e0, e1, 0);
  lds_wait(e0);
  lds_wait(e1);
  int kt = 1;
  for (; kt + 1 < nk; kt += 2) {
    land(kt);
    step(yes, yes, e0, e1, o0, o1, kt);
    lds_wait(o0);
    lds_wait(o1);
    land(kt + 1);
    step(yes, yes, o0, o1, e0, e1, kt + 1);
    lds_wait(e0);
    lds_wait(e1);
  }
  if (kt < nk) {                                                 // an even number of stages: one odd stage is left
    land(kt);
    step(yes, yes, e0, e1, o0, o1, kt);
    lds_wait(o0);
    lds_wait(o1);
    step(yes, no, o0, o1, e0, e1, 0);
  } else {
    step(yes, no, e0, e1, o0, o1, 0);
  }

  // out[split][w * cp + c][n]: accumulator r of a 32 x 32 tile is row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31
  float* const out = p.out + (long)split * p.slab_stride + (rot ? p.lag_im_offset : 0L);
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        // (columns between the gradient's channel pitch and n_pad met the NEXT row's bytes: they are padding, exactly zero)
        if (c < p.x_cp && n < p.n_pad) {
          float v = n < p.z_cp ? acc[i][j][r] : 0.f;
          if (!LAG && bias_wave && c == p.x_cp - 1 && p.bias_direct) {       // unsplit: straight to the bias gradient, the row stays zero
            p.bias_direct[n] = v;
            v = 0.f;
          }
          out[((long)w * p.x_cp + c) * p.n_pad + n] = v;
        }
      }
    }
}

// column sums of a bf16 plane [rows][cp] over row chunks: part[chunk][c] (fp32).  A workgroup covers 256 columns -- 32 threads x 8
// columns = one 512-byte piece of a row -- with 8 row lanes, four rows in flight per lane
__global__ __launch_bounds__(256) void colsum_bf16_partial_kernel(const unsigned short* __restrict__ z, long rows, int cp, int n_pad,
                                                                  float* __restrict__ part) {
  __shared__ float red[8][257];
  const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 256 + cq * 8;
  const long per = (rows + gridDim.y - 1) / gridDim.y;
  const long r0 = (long)blockIdx.y * per, r1 = min(rows, r0 + per);
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < cp) {
    long r = r0 + rl;
    for (; r + 24 < r1; r += 32) {
      bf16x8 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const bf16x8*>(z + (r + 8 * k) * cp + c);
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += (float)v[k][e];
    }
    for (; r < r1; r += 8) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(z + r * cp + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cq * 8 + e] = s[e];
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x];          // fixed order
  const int cc = blockIdx.x * 256 + threadIdx.x;
  if (cc < n_pad) part[(long)blockIdx.y * n_pad + cc] = cc < cp ? t : 0.f;
}

// dpacked = sum of the slabs (fixed order); dbias = sum of the column-sum partials: the first n_pad / 32 workgroups take 32 columns
// each, eight lanes per column summing every eighth chunk, then the eight in lane order (a fixed tree: deterministic)
__global__ __launch_bounds__(256) void wgrad_tr_finish_kernel(const float* __restrict__ slabs, int n_slabs, size_t n4,
                                                              float* __restrict__ dpacked, const float* __restrict__ part, int chunks,
                                                              int n_pad, float* __restrict__ dbias, size_t bias_row4) {
  __shared__ float red[8][33];
  if (part && (int)blockIdx.x * 32 < n_pad) {
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), r = threadIdx.x >> 5;
    float t = 0.f;
    if (c < n_pad) {
      for (int k0 = r; k0 < chunks; k0 += 8 * 8) {               // eight loads in flight, added in chunk order
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = part[(long)min(k0 + 8 * k, chunks - 1) * n_pad + c];
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k0 + 8 * k < chunks) t += v[k];
      }
    }
    red[r][threadIdx.x & 31] = t;
    __syncthreads();
    if (r == 0 && c < n_pad) {
      float a = red[0][threadIdx.x];
#pragma unroll
      for (int k = 1; k < 8; ++k) a += red[k][threadIdx.x];
      dbias[c] = a;
    }
  }
  if (slabs) {
    // eight slabs' loads in flight at a time, added in slab order (a loop of one dependent load per slab was latency-bound: 21 us
    // for the 16 MB of a 250-channel layer's nine slabs)
    const f32x4* const src = reinterpret_cast<const f32x4*>(slabs);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      for (int s0 = 0; s0 < n_slabs; s0 += 8) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[(size_t)min(s0 + k, n_slabs - 1) * n4 + i];
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (s0 + k < n_slabs) a += v[k];
      }
      if (bias_row4 && i >= bias_row4 && i < bias_row4 + (size_t)n_pad / 4) {    // row x_cp - 1 of tap 0 carried the column sums
        reinterpret_cast<f32x4*>(dbias)[i - bias_row4] = a;
        a = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      reinterpret_cast<f32x4*>(dpacked)[i] = a;
    }
  }
}

int npad_of(int cout) { return cout <= 32 ? 32 : (cout <= 64 ? 64 : (int)st::round_up(cout, 128)); }

struct TrPlan {
  long z_row0, x_row0, rows;
  int stages, splits, stages_per_split, n_pad, mtiles_per_tap, tiles_m, tiles_n, chunks;
  size_t slab_bytes, part_bytes;
};

TrPlan tr_plan(const st_tensor3& x, const st_tensor3& dz, int width, int pad_left) {
  TrPlan t{};
  t.n_pad = npad_of(dz.channels);
  t.z_row0 = dz.halo;                                                     // the first real gradient row
  t.x_row0 = (long)dz.halo + (x.halo - pad_left) - dz.halo;               // its input row for tap 0
  t.rows = (long)(dz.batch - 1) * dz.t_pitch + dz.frames;                 // up to the last real gradient row
  t.stages = (int)((t.rows + TK - 1) / TK);
  t.mtiles_per_tap = st::ceil_div(x.c_pitch, TM);
  t.tiles_m = width * t.mtiles_per_tap;
  t.tiles_n = st::ceil_div(t.n_pad, TN);
  const long tiles = (long)t.tiles_m * t.tiles_n;
  // Splits (measured round 5 on the config-2 shapes, bf16 step): every split adds a slab to write and sum -- the 250-channel
  // layers' 28 tiles in 9 runs of 57 stages (256 pairs) and in 18 (512 pairs) give the same step, 2.49 ms; 768 pairs 2.58, 1 024
  // 2.62 -- but a workgroup alone on its CU is latency-bound (the 2000 x 2000 layer's 256 tiles unsplit: 0.32 us per stage for 0.11
  // us of MFMAs, 163 us; in two halves, two workgroups per CU, 146 us): about 256 pairs, and two halves when the tile grid
  // alone gives about one workgroup per CU and the reduction is long; at least 8 stages per split.
  const int forced = st::tuning(st::TUNE_BF16_WGRAD_SPLITS);
  const int target = st::tuning(st::TUNE_BF16_WGRAD_TARGET) > 0 ? st::tuning(st::TUNE_BF16_WGRAD_TARGET) : 256;
  int splits = forced ? forced : (int)std::max(1L, std::min<long>(target / std::max(1L, tiles), t.stages / 8));
  if (!forced && splits == 1 && tiles >= 128 && tiles <= 320 && t.stages >= 64) splits = 2;
  // (a power of two, so that splits and XCDs divide each other: whole splits per XCD or whole XCDs per split, see the kernel)
  if (!forced && !st::tuning(st::TUNE_BF16_WGRAD_PLAIN_ORDER))
    while (splits & (splits - 1)) splits &= splits - 1;
  splits = std::max(1, std::min(splits, t.stages));
  t.stages_per_split = st::ceil_div(t.stages, splits);
  t.splits = st::ceil_div(t.stages, t.stages_per_split);
  t.slab_bytes = t.splits > 1 ? st::round_up((size_t)t.splits * width * x.c_pitch * t.n_pad * 4, 256) : 0;
  t.chunks = std::max(64, std::min(256, 512 / st::ceil_div(t.n_pad, 256)));     // row chunks of the column sums: about 512 workgroups
  t.part_bytes = st::round_up((size_t)t.chunks * t.n_pad * 4, 256);
  return t;
}

bool tr_eligible(const st_tensor3* x, const st_tensor3* dz, int width, int stride, int pad_left) {
  if (!(x && dz && stride == 1 && width >= 1 && x->t_pitch == dz->t_pitch && x->batch == dz->batch && x->frames == dz->frames &&
        x->halo >= pad_left && x->c_pitch % 8 == 0 && dz->c_pitch % 8 == 0 && dz->c_pitch >= dz->channels &&
        (long)(x->halo - pad_left) + width - 1 <= (long)x->t_pitch))
    return false;
  // The kernel walks WHOLE 32-row stages over the flat rows and whole 128-column tiles over a row: its last stage, the taps'
  // row shifts and a tile wider than what is left of the channel pitch all read past the last row of the planes.  The caller's
  // contract is SLACK_ROWS readable zero rows behind each plane (st_conv1d_bwd_filter_tr_bf16_slack_rows): a geometry whose
  // furthest read would leave that slack is refused here (workspace query 0, launch ST_EINVAL) instead of reading beyond it.
  auto cdiv = [](long a, long b) { return (a + b - 1) / b; };
  const long plane_rows = (long)x->batch * x->t_pitch;
  const long rows = (long)(dz->batch - 1) * dz->t_pitch + dz->frames;
  const long staged = cdiv(rows, TK) * TK;                                                 // rows the stages cover
  const int n_pad = npad_of(dz->channels);
  const long x_over = cdiv(x->c_pitch, TM) * TM - x->c_pitch, z_over = cdiv(n_pad, TN) * TN - dz->c_pitch;
  const long x_spill = x_over > 0 ? cdiv(x_over, x->c_pitch) : 0, z_spill = z_over > 0 ? cdiv(z_over, dz->c_pitch) : 0;
  const long x_last = (long)(x->halo - pad_left) + (width - 1) + staged - 1 + x_spill;     // furthest input row read
  const long z_last = (long)dz->halo + staged - 1 + z_spill;                               // furthest gradient row read
  return x_last < plane_rows + SLACK_ROWS && z_last < plane_rows + SLACK_ROWS;
}

}  // namespace

// Q[2 b + j] = (j == 0 ? S2 : S'2)[b]^T Z2[b] for b < bins: S spectra [bins][rows][2 * half] and Z spectra [bins][rows][2 * npo] as one
// bf16 plane each, Q fp32 [2 * bins][half][npo]; rows a multiple of 16, half and npo multiples of 128 (no read runs past a bin)
int st::lag_products_tr_bf16(const void* s_plane, const void* z_plane, int bins, int rows, int half, int npo, float* q, hipStream_t s) {
  if (!(s_plane && z_plane && q && bins > 0 && rows > 0 && (2 * rows) % TK == 0 && half % TM == 0 && npo % TN == 0)) {
    st::set_error("lag_products_tr_bf16: bad shape bins=%d rows=%d half=%d npo=%d", bins, rows, half, npo);
    return ST_EINVAL;
  }
  TrParams p{};
  p.X = reinterpret_cast<const unsigned short*>(s_plane);
  p.Z = reinterpret_cast<const unsigned short*>(z_plane);
  p.out = q;
  p.x_row0 = 0; p.z_row0 = 0;
  p.slab_stride = 2L * half * npo;                   // bin b: Q[2 b] (real), then Q[2 b + 1] (imaginary)
  p.lag_im_offset = (long)half * npo;
  p.x_cp = half; p.z_cp = npo; p.n_pad = npo; p.width = 1;
  p.stages_per_split = 2 * rows / TK; p.splits = bins; p.stages = bins * p.stages_per_split;
  p.mtiles_per_tap = half / TM; p.tiles_m = p.mtiles_per_tap; p.tiles_n = npo / TN;
  st::trace("wgrad_tr_bf16<128,128,32,lag> batched bins=%d M=%d Np=%d Kp=%d gflop=%.3f", 2 * bins, half, npo, 2 * rows,
            2e-9 * 2.0 * bins * half * (double)npo * 2.0 * rows);
  {
    st::LaunchTimer timer(s);
    if (st::tuning(st::TUNE_BF16_WGRAD_PLAIN_ORDER)) {
      st::launch_timed(timer, wgrad_tr_bf16_kernel<true, 4>, dim3((unsigned)(bins * p.tiles_m * p.tiles_n), 2), dim3(256), s, p);
    } else {
      p.map = 3;
      const unsigned groups = (unsigned)(bins * p.tiles_n), members = 2u * p.tiles_m;
      st::launch_timed(timer, wgrad_tr_bf16_kernel<true, 4>, dim3(8u * st::ceil_div(groups, 8u) * members), dim3(256), s, p);
    }
  }
  return st::check_launch("lag_products_tr_bf16");
}

extern "C" {

int st_conv1d_bwd_filter_tr_bf16_slack_rows(void) { return SLACK_ROWS; }

size_t st_conv1d_bwd_filter_tr_bf16_ws(const st_tensor3* x, const st_tensor3* dz, int width, int stride, int pad_left) {
  if (!tr_eligible(x, dz, width, stride, pad_left)) return 0;
  const TrPlan t = tr_plan(*x, *dz, width, pad_left);
  return t.slab_bytes + t.part_bytes + 256;
}

int st_conv1d_nwc_bwd_filter_tr_bf16(const st_tensor3* x, const void* x_bf16, const st_tensor3* dz, const void* dz_bf16, int width,
                                     int stride, int pad_left, float* dpacked, float* dbias, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  ST_REQUIRE(x && dz && x_bf16 && dz_bf16 && dpacked && dbias, "conv bwd-filter tr bf16: null argument");
  ST_REQUIRE(tr_eligible(x, dz, width, stride, pad_left),
             "conv bwd-filter tr bf16: stride-1 layers whose input and gradient tensors share one frame pitch only");
  const TrPlan t = tr_plan(*x, *dz, width, pad_left);
  const size_t need = t.slab_bytes + t.part_bytes + 256;
  if (!workspace || workspace_bytes < need) {
    st::set_error("conv bwd-filter tr bf16: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
    return ST_EWORKSPACE;
  }
  hipStream_t s = st::as_stream(stream);
  float* slabs = reinterpret_cast<float*>(workspace);
  float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + t.slab_bytes);
  TrParams p{};
  p.X = reinterpret_cast<const unsigned short*>(x_bf16);
  p.Z = reinterpret_cast<const unsigned short*>(dz_bf16);
  p.out = t.splits > 1 ? slabs : dpacked;
  p.x_row0 = t.x_row0; p.z_row0 = t.z_row0;
  p.slab_stride = (long)width * x->c_pitch * t.n_pad;
  p.x_cp = x->c_pitch; p.z_cp = dz->c_pitch; p.n_pad = t.n_pad; p.width = width;
  p.stages = t.stages; p.stages_per_split = t.stages_per_split; p.splits = t.splits;
  p.mtiles_per_tap = t.mtiles_per_tap; p.tiles_m = t.tiles_m; p.tiles_n = t.tiles_n;
  // the bias gradient from the same MFMAs (see the kernel) when the input's channel pitch has a padding channel to carry it
  const bool in_tile = x->c_pitch > x->channels && x->c_pitch % TM == 0 && st::tuning(st::TUNE_BF16_WGRAD_BIAS_PASS) == 0;
  p.bias_in_tile = in_tile ? 1 : 0;
  p.bias_direct = in_tile && t.splits == 1 ? dbias : nullptr;
  st::trace("wgrad_tr_bf16<128,128,32> M=%d Np=%d rows=%ld stages=%d splits=%d gflop=%.3f", width * x->c_pitch, t.n_pad, t.rows, t.stages,
            t.splits, 2e-9 * (double)t.stages * TK * t.tiles_m * TM * t.tiles_n * TN);
  {
    st::LaunchTimer timer(s);
    // ring depth: a grid of about one workgroup per CU cannot hide the fabric latency behind a second workgroup: eight 16 KB
    // stages (seven in flight) instead of four, the whole LDS for the one workgroup a CU gets anyway
    const int tiles = t.tiles_m * t.tiles_n;
    unsigned grid = (unsigned)(t.splits * tiles);
    const int ring = st::tuning(st::TUNE_BF16_WGRAD_RING) ? st::tuning(st::TUNE_BF16_WGRAD_RING) : (grid <= 320 ? 8 : 4);
    if (!st::tuning(st::TUNE_BF16_WGRAD_PLAIN_ORDER)) {
      if (t.splits % 8 == 0) {
        p.map = 1; p.map_a = t.splits / 8;
      } else if (8 % t.splits == 0) {
        p.map = 2; p.map_a = 8 / t.splits; p.map_b = st::ceil_div(tiles, p.map_a);
        grid = 8u * p.map_b;
      }
    }
    if (ring == 8) st::launch_timed(timer, wgrad_tr_bf16_kernel<false, 8>, dim3(grid), dim3(256), s, p);
    else st::launch_timed(timer, wgrad_tr_bf16_kernel<false, 4>, dim3(grid), dim3(256), s, p);
  }
  if (int e = st::check_launch("wgrad_tr_bf16")) return e;
  if (!in_tile) {
    const long z_rows = (long)dz->batch * dz->t_pitch;
    hipLaunchKernelGGL(colsum_bf16_partial_kernel, dim3(st::ceil_div(t.n_pad, 256), t.chunks), dim3(256), 0, s, p.Z, z_rows,
                       dz->c_pitch, t.n_pad, part);
  } else if (t.splits == 1) {
    return 0;                                                    // filter and bias gradient are both where they belong
  }
  const size_t n4 = (size_t)width * x->c_pitch * t.n_pad / 4;
  const unsigned bias_blocks = in_tile ? 1u : (unsigned)st::ceil_div(t.n_pad, 32);
  const unsigned sum_blocks = t.splits > 1 ? (unsigned)std::min<size_t>((n4 + 255) / 256, 2048) : 0u;
  hipLaunchKernelGGL(wgrad_tr_finish_kernel, dim3(std::max(bias_blocks, sum_blocks)), dim3(256), 0, s,
                     t.splits > 1 ? slabs : (const float*)nullptr, t.splits, n4, dpacked, in_tile ? (const float*)nullptr : part, t.chunks,
                     t.n_pad, dbias, in_tile ? (size_t)(x->c_pitch - 1) * t.n_pad / 4 : (size_t)0);
  return st::check_launch("wgrad_tr_finish");
}

}  // extern "C"
