// W-tap stride-1 convolution with bf16 activations on 256-channel rows (BASELINE config 4 arithmetic; the seven 250 -> 250, 7-tap
// layers of the model, forward pass and back-prop to the input):
//
//   Y[q][n] = sum_w sum_c A[q + w + a_row0][c] * B[n][w * 256 + c]          (speech_model.py:285-288, tf.nn.conv1d 'SAME')
//
// q runs over the FLAT rows of the output plane, utterances one behind the other (input and output have the same frame pitch, so
// one shift per tap maps an output row to its input row for every utterance at once); rows between utterances are computed and
// not stored.  The general kernel (conv_bf16.hip, gemm_nn_bf16_kernel<128, 2, 2, 64, 1, 4>) treats the taps as seven
// independent reduction slices: every tap stages its own shifted copy of the same 128 input rows, 458 KB of input rows and 458 KB
// of filter rows per tile through a ring that holds three stages in flight -- and that, not the matrix pipe (22 % busy), sets its
// 24 us: the waves are parked on the ring 44 % of the time (profiles/r5_pmc_shapes_bf16.txt).  Here the 134 input rows a tile
// needs are staged ONCE (68 KB, all taps read them at a row offset) and the rest of the LDS is a ring for the filter rows alone:
// half the bytes per tile through the L2 -> LDS path, and five 64-deep filter stages (four in flight = 256 reduction steps,
// against 192) in the 80 KB left.
//
// Tile 128 output rows x 128 output channels, four waves of 64 x 64 (v_mfma_f32_32x32x16_bf16, filter fragment as the first
// operand so that a lane's accumulators are runs of four consecutive channels of one row), one workgroup per CU.
// LDS image of the input panel: [136 rows][32 chunks of 16 bytes], physical chunk = chunk ^ (row & 15); of a filter stage:
// [128 channels][8 chunks], physical chunk = chunk ^ ((channel >> 1) & 7): both swizzles are applied on the SOURCE side of the
// LDS-DMA, and make the sixteen lanes ds_read_b128 serves per cycle cover all 64 banks for any tap offset.
// The fragments of a stage are read one stage AHEAD of the MFMAs that consume them (two register sets), counted vmcnt and one
// bare s_barrier per stage as in wgrad_tr_bf16.hip.
#include <algorithm>
#include <type_traits>

#include "st_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int TM = 128, TN = 128, TK = 64, CP = 256;
constexpr int ST = 5;                                // filter ring depth
constexpr int A_ROWS = 136;                          // TM + 8 taps at most (rounded to whole 1 KB DMA pieces of two rows)
constexpr int A_BYTES = A_ROWS * CP * 2;             // 69 632
constexpr int B_STAGE = TN * TK * 2;                 // 16 384
constexpr int MAX_TAPS = A_ROWS - TM + 1;            // 9

struct TapsParams {
  const unsigned short* A;       // bf16 plane of the operand, flat rows of 256 elements
  const unsigned short* B;       // filters [Np][Kp], Kp = width * 256, reduction index w * 256 + c
  unsigned short* C;             // bf16 plane of the result, flat rows of c_cp elements
  const float* bias;             // nullable
  const unsigned short* mask;    // nullable: bf16 plane, the result is kept where mask > 0 (ReLU of the layer below)
  long a_row0, a_rows;           // flat operand row of (q = 0, tap 0); rows of the operand plane (the panel's DMA is clamped to it)
  long c_row0, m_row0;           // flat result / mask row of q = 0
  long Q;                        // flat rows to cover
  int t_pitch, frames;           // row q is a frame iff q % t_pitch < frames
  int Kp, width, c_cp, m_cp, n_store, relu;
  int tiles_m, tiles_n;
};

__device__ __forceinline__ u32x4 lds_read128(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
// the sixteen fragments of a stage as in/out operands: nothing that uses them is scheduled in front of the wait
__device__ __forceinline__ void lds_wait(u32x4 (&a)[8], u32x4 (&b)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(b[0]),
                 "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7])
               :
               : "memory");
}

__global__ __launch_bounds__(256, 1) void conv_taps_bf16_kernel(TapsParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned short smem[(A_BYTES + ST * B_STAGE) / 2];

  // the eight XCDs take the row tiles in turn; the column tiles of a row tile follow each other on ONE XCD (they share the panel)
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int tn = idx % p.tiles_n, tm = (idx / p.tiles_n) * 8 + xcd;
  if (tm >= p.tiles_m) return;
  const long q0 = (long)tm * TM;
  const int n0 = tn * TN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
  const int nk = p.width * (CP / TK);

  // the input panel: 68 pieces of two rows, wave v takes pieces v, v + 4, ...; lane i writes physical chunk i & 31 of row i >> 5
  {
    const int prow = lane >> 5, pch = lane & 31;
    for (int piece = wave; piece < A_ROWS / 2; piece += 4) {
      const int row = 2 * piece + prow;
      const long g = min(max(p.a_row0 + q0 + row, 0L), p.a_rows - 1);
      const unsigned short* src = p.A + g * CP + ((pch ^ (row & 15)) << 3);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + piece * 512), 16, 0, 0);
    }
  }
  // a filter stage: 16 pieces of eight channels x 128 bytes, wave v takes pieces 4 v .. 4 v + 3; lane i writes physical chunk
  // i & 7 of channel i >> 3
  const unsigned short* bsrc[4];
#pragma unroll
  for (int pi = 0; pi < 4; ++pi) {
    const int n = (wave * 4 + pi) * 8 + (lane >> 3);
    bsrc[pi] = p.B + (long)(n0 + n) * p.Kp + (((lane & 7) ^ ((n >> 1) & 7)) << 3);
  }
  auto issue = [&](int kt, int slot) {
#pragma unroll
    for (int pi = 0; pi < 4; ++pi)
      __builtin_amdgcn_global_load_lds((gptr_t)(bsrc[pi] + (long)kt * TK),
                                       (lptr_t)(smem + (A_BYTES + slot * B_STAGE) / 2 + (wave * 4 + pi) * 512), 16, 0, 0);
  };

  // fragment addresses.  Panel: row = 64 wm + 32 i + (lane & 31) + tap, chunk = 8 kc + 2 kk + h; filter stage: channel
  // 64 wn + 32 j + (lane & 31), chunk 2 kk + h
  const unsigned lds0 = (unsigned)(size_t)smem;
  int a_row[2];
  unsigned b_at[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    a_row[i] = wm * 64 + i * 32 + l31;
    const int n = wn * 64 + i * 32 + l31;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      b_at[i][kk] = lds0 + A_BYTES + (unsigned)(n * 128 + ((((2 * kk + h) ^ ((n >> 1) & 7))) << 4));
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // One wave per SIMD issues in order: sixteen ds_read_b128 in a row hold the MFMAs behind them back for as long as the LDS takes
  // to accept them (measured: reads and MFMAs cost the SUM of their times, 5.5 + 7 us of 25), so a stage is issued as sixteen
  // pairs -- one MFMA of the stage in hand (registers filled an iteration ago), one fragment read of the next stage in its
  // 32-cycle shadow -- with a scheduling fence after each pair.  DO_MUL / DO_READ: the first stage only reads, the last only
  // multiplies.
  auto step = [&](auto do_mul, auto do_read, u32x4 (&pa)[8], u32x4 (&pb)[8], u32x4 (&na)[8], u32x4 (&nb)[8], int kt, int slot) {
    const int w = kt >> 2, kc = kt & 3;
    const unsigned so = (unsigned)(slot * B_STAGE);
    unsigned abase[2], ax[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = a_row[i] + w;
      abase[i] = lds0 + (unsigned)(r * (CP * 2)) ;
      ax[i] = (unsigned)((r & 15) << 4);
    }
    const unsigned c0 = (unsigned)((kc * 8 + h) << 4);
#pragma unroll
    for (int idx = 0; idx < 16; ++idx) {
      if (decltype(do_mul)::value) {
        const int kk = idx >> 2, i = (idx >> 1) & 1, j = idx & 1;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pb[j * 4 + kk]),
                                                            __builtin_bit_cast(bf16x8, pa[i * 4 + kk]), acc[i][j], 0, 0, 0);
      }
      if (decltype(do_read)::value) {
        // sixteen reads under the first twelve MFMAs (two each under the first four), none under the last four: their 128 cycles
        // cover the latency of the last read.  Fragments of k-step 0 first (both operands), then 1, 2, 3 -- the order the next
        // stage's MFMAs want them in
        const int first = idx < 4 ? 2 * idx : idx + 4, count = idx < 4 ? 2 : (idx < 12 ? 1 : 0);
#pragma unroll
        for (int r = first; r < first + count; ++r) {
          const int kk = r >> 2, o = r & 3;
          if (o < 2) na[o * 4 + kk] = lds_read128(abase[o] + ((c0 + kk * 32) ^ ax[o]));
          else nb[(o - 2) * 4 + kk] = lds_read128(b_at[o - 2][kk] + so);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  const std::true_type yes{};
  const std::false_type no{};
  auto land = [&](int kt, int slot) {
    // this wave's pieces of stage kt (and, older, of the panel) have landed when at most the younger stages' are outstanding;
    // a bare s_barrier (__syncthreads() is a fence too and would drain every DMA in flight)
    if (kt + ST - 1 <= nk) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (ST - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    // into the slot of stage kt - 1: its fragment reads were waited for before this barrier
    if (kt + ST - 1 < nk) issue(kt + ST - 1, slot == 0 ? ST - 1 : slot - 1);
  };

  for (int kt = 0; kt < ST - 1 && kt < nk; ++kt) issue(kt, kt);
  u32x4 ea[8], eb[8], oa[8], ob[8];                              // fragments of the even / the odd stages
  int slot = 0;
  land(0, slot);
  step(no, yes, oa, ob, ea, eb, 0, slot);
  lds_wait(ea, eb);
  int kt = 1;
  for (; kt + 1 < nk; kt += 2) {
    slot = slot == ST - 1 ? 0 : slot + 1;
    land(kt, slot);
    step(yes, yes, ea, eb, oa, ob, kt, slot);
    lds_wait(oa, ob);
    slot = slot == ST - 1 ? 0 : slot + 1;
    land(kt + 1, slot);
    step(yes, yes, oa, ob, ea, eb, kt + 1, slot);
    lds_wait(ea, eb);
  }
  if (kt < nk) {                                                 // an even number of stages: one odd stage is left
    slot = slot == ST - 1 ? 0 : slot + 1;
    land(kt, slot);
    step(yes, yes, ea, eb, oa, ob, kt, slot);
    lds_wait(oa, ob);
    step(yes, no, oa, ob, ea, eb, 0, 0);
  } else {
    step(yes, no, ea, eb, oa, ob, 0, 0);
  }

  // epilogue: lane & 31 = output row, accumulator r = output channel (r & 3) + 8 (r >> 2) + 4 h of the 32 x 32 sub-tile.  A lane
  // owns runs of four channels of one row; stored from the registers they are 16-byte pieces in 32 different rows per
  // instruction, half a million partial-line writes per layer (measured: 7 of the kernel's 25 us).  The tile goes through the
  // LDS instead ([128 rows][272 bytes]: the 16 lanes ds_write_b64 serves per cycle land 4 banks apart) and leaves as whole
  // 256-byte row segments, 16 bytes per lane; the ReLU mask of back-prop is read the same way.
  asm volatile("s_barrier" ::: "memory");                        // every wave is through with the panel and the ring
  constexpr int SP = 272;
  unsigned char* const stage = reinterpret_cast<unsigned char*>(smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = wn * 64 + j * 32 + 8 * g + 4 * h;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
        if (p.bias && n0 + col < p.n_store) v += *reinterpret_cast<const f32x4*>(p.bias + n0 + col);
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        *reinterpret_cast<bf16x4*>(stage + (wm * 64 + i * 32 + l31) * SP + col * 2) =
            bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
      }
  __syncthreads();
  const int ch = tid & 15;
  const int col = n0 + ch * 8;
  if (col >= p.n_store) return;                                  // n_store is a multiple of 16: chunks are all-or-nothing
#pragma unroll
  for (int pass = 0; pass < TM / 16; ++pass) {
    const int r = pass * 16 + (tid >> 4);
    const long q = q0 + r;
    if (q >= p.Q || (int)(q % p.t_pitch) >= p.frames) continue;  // a row between two utterances
    bf16x8 v = *reinterpret_cast<const bf16x8*>(stage + r * SP + ch * 16);
    if (p.mask) {
      const bf16x8 m = *reinterpret_cast<const bf16x8*>(p.mask + (p.m_row0 + q) * p.m_cp + col);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (float)m[e] > 0.f ? v[e] : (__bf16)0.f;
    }
    *reinterpret_cast<bf16x8*>(p.C + (p.c_row0 + q) * p.c_cp + col) = v;
  }
}

int npad_of(int cout) { return cout <= 32 ? 32 : (cout <= 64 ? 64 : (int)st::round_up(cout, 128)); }

}  // namespace

// operand `a` (plane a_plane) -> result `y` (plane y_plane): tap w of output frame t reads operand frame t + w - lead.
bool st::conv_taps_bf16_eligible(const st_tensor3& a, const st_tensor3& y, int width, int lead, const st_tensor3* act) {
  if (st::tuning(st::TUNE_BF16_TAPS_PANEL) == 1) return false;
  const int np = npad_of(y.channels);
  if (a.c_pitch != CP || width < 2 || width > MAX_TAPS || np % TN || y.c_pitch % 16) return false;
  if (a.t_pitch != y.t_pitch || a.batch != y.batch || a.frames != y.frames || a.halo < lead) return false;
  if (act && (act->t_pitch != y.t_pitch || act->batch != y.batch || act->c_pitch < std::min(y.c_pitch, np))) return false;
  const long Q = (long)(y.batch - 1) * y.t_pitch + y.frames;
  // few output rows (single-utterance inference) keep the general kernel's reduction split
  return st::ceil_div(Q, (long)TM) * (np / TN) >= 128;
}

int st::conv_taps_bf16(const st_tensor3& a, const void* a_plane, const void* filters, const float* bias, int width, int lead, int relu,
                       const st_tensor3* act, const void* act_plane, const st_tensor3& y, void* y_plane, hipStream_t s) {
  TapsParams p{};
  p.A = reinterpret_cast<const unsigned short*>(a_plane);
  p.B = reinterpret_cast<const unsigned short*>(filters);
  p.C = reinterpret_cast<unsigned short*>(y_plane);
  p.bias = bias;
  p.mask = act ? reinterpret_cast<const unsigned short*>(act_plane) : nullptr;
  p.a_row0 = a.halo - lead;
  p.a_rows = (long)a.batch * a.t_pitch;
  p.c_row0 = y.halo;
  p.m_row0 = act ? act->halo : 0;
  p.Q = (long)(y.batch - 1) * y.t_pitch + y.frames;
  p.t_pitch = y.t_pitch;
  p.frames = y.frames;
  p.Kp = width * CP;
  p.width = width;
  p.c_cp = y.c_pitch;
  p.m_cp = act ? act->c_pitch : 0;
  const int np = npad_of(y.channels);
  p.n_store = std::min(y.c_pitch, np);
  p.relu = relu;
  p.tiles_m = (int)st::ceil_div(p.Q, (long)TM);
  p.tiles_n = np / TN;
  st::trace("conv_taps_bf16<128,128,64,panel> M=%ld Np=%d Kp=%d taps=%d gflop=%.3f", p.Q, np, p.Kp, width,
            2e-9 * (double)p.tiles_m * TM * np * p.Kp);
  {
    st::LaunchTimer timer(s);
    st::launch_timed(timer, conv_taps_bf16_kernel, dim3((unsigned)(8 * st::ceil_div(p.tiles_m, 8) * p.tiles_n)), dim3(256), s, p);
  }
  return st::check_launch("conv_taps_bf16");
}
