// CTC loss + gradient and greedy decoding on gfx950.
//
// Replaces tf.nn.ctc_loss (speech_model.py:74; CPU-only op in TF1) and
// tf.nn.ctc_greedy_decoder(merge_repeated=True) (speech_model.py:113-115).
// Semantics follow TF: blank = C-1, beta excludes the emission at t,
// loss = -log p(l|x) unnormalised, gradient wrt the *logits*.  Arithmetic: the lattice is NOT kept in log space
// (TF's, and rounds 1-2's here) but as mantissa * 2^exponent with an integer exponent per state ("scaled arithmetic"
// below): same range, fewer and cheaper instructions on the sequential chain, and tighter than fp32 logs.
//
// Structure:
//  1. ctc_logsoftmax: one thread per (b, t) row of <= 32 classes.
//  2. ctc_alpha_beta<KPL>: the sequential part.  One WAVE per (utterance, direction): the
//     2L+1 lattice states are dealt KPL-contiguous per lane, so a whole time step is register
//     arithmetic plus four cross-lane shifts (DPP) -- no LDS round trip, no barrier on the 500-1500
//     step critical path.  Emission factors are staged through LDS in 64-frame chunks, prefetched
//     one chunk ahead.  alpha and beta run concurrently on different CUs.
//  3. ctc_grad: fully parallel over (b, t): occupancy per class from alpha+beta with a fixed
//     summation order (per-class position lists), so results are run-to-run deterministic.
#include <algorithm>

#include "st_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int CP = 32;        // class pitch of the log-softmax scratch
constexpr int TC = 64;        // frames per LDS emission chunk
#define NEG_INF (-__builtin_inff())

struct RowMap2 {   // (b, t) -> float offset
  long batch_stride;
  long row0;
  int row_stride;
  __device__ __forceinline__ long off(int b, int t) const {
    return (long)b * batch_stride + row0 + (long)t * row_stride;
  }
};

// ---- scaled arithmetic --------------------------------------------------------------------------
// A lattice value is m * 2^e: m a float in [0.5, 1) (0: no path reaches the state), e an int of its own PER STATE.
// The recursion then is three v_ldexp_f32 (align to the largest exponent), two adds, one multiply by the emission and
// v_frexp_mant / v_frexp_exp -- no v_exp_f32 / v_log_f32 (8.8 cycles of issue against 4.9 for everything else,
// scripts/ubench/valu_rates.hip: 20 of the 130 instructions of a frame of the log-domain recursion of rounds 1-2) on the
// 500-1500 step chain, no re-centring of columns, no double-precision offsets: 108 instructions per frame for five states
// per lane, 160 -> 117 us for T' = 501.  It is also the more accurate form: every step rounds a 24-bit mantissa (relative 6e-8), where
// a log2 value of magnitude 1000 has an ulp of 6e-5; and unlike a column-wide scale (tried in round 2: 3.7 nats off on a
// 1 377-frame utterance, states 2^186 below the column maximum still carry the alignments that finish) a per-state
// exponent cannot underflow.
constexpr int EZ = -(1 << 28);           // exponent of a zero state: below anything a path reaches, differences stay in int range
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

// cross-lane moves on the VALU (DPP), no LDS round trip; lanes without a source keep `fill`
template <int CTRL>
__device__ __forceinline__ int dpp_movei(int v, int fill) {
  return __builtin_amdgcn_update_dpp(fill, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp_movef(float v, float fill) {
  return __builtin_bit_cast(float, dpp_movei<CTRL>(__builtin_bit_cast(int, v), __builtin_bit_cast(int, fill)));
}
// (by value: __builtin_bit_cast applied directly to a vector ELEMENT reads element 0 -- clang takes the vector's address)
__device__ __forceinline__ int ibits(float x) { return __builtin_bit_cast(int, x); }
constexpr int SHR1 = 0x138, SHL1 = 0x130;   // wave_shr:1 (value of the lane below), wave_shl:1 (of the lane above)

// m * 2^e of the three aligned terms; zero terms (m = 0, any e) drop out, all zero -> 0 * 2^EZ
__device__ __forceinline__ float aligned_sum3(float m0, int e0, float m1, int e1, float m2, int e2, int& E) {
  E = max(max(e0, e1), e2);
  return __builtin_ldexpf(m0, e0 - E) + __builtin_ldexpf(m1, e1 - E) + __builtin_ldexpf(m2, e2 - E);
}

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// log2-softmax of every (b, t) row into the scratch [B*T][32], and the same numbers as emission factors for the recursion:
// emis[row][c] = (m, e) with 2^logy = m * 2^e, m in [1, 2).
// One lane per (row, class): two rows per wavefront, reductions over the 32 lanes of a row by DPP-free shuffles.  The
// arithmetic is DOUBLE: log2 y has to be right to ~1e-8 ABSOLUTE, not to a float's 6e-8 relative -- on a fresh network every
// frame's distribution is nearly the same (all logits ~0, log2 y ~ -4.86), so the float rounding of log2 y (2.4e-7) had the
// same sign on all 501 frames and the loss came out 1e-4 low (measured round 4, scripts/diag_ctc_loss.py: -9.9e-5 on a loss
// of 1 245; the recursion itself adds ~1e-6).  What remains is the float mantissa of the factor (3e-8 relative per frame).
__global__ __launch_bounds__(256) void ctc_logsoftmax_kernel(const float* __restrict__ logits, RowMap2 map,
                                                             int B, int T, int C, float* __restrict__ logy,
                                                             float* __restrict__ emis) {
  const int lane = threadIdx.x & 63, c = lane & 31;
  const int i = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
  const bool row_ok = i < B * T;
  const int ii = row_ok ? i : B * T - 1;
  const int b = ii / T, t = ii - b * T;
  const float v = c < C ? logits[map.off(b, t) + c] : NEG_INF;
  float m = v;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  double e = c < C ? exp((double)v - (double)m) : 0.0;
  double s = e;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const double lz = (double)m + log(s);
  const double l2 = c < C ? ((double)v - lz) * 1.4426950408889634074 : 0.0;
  // a class 2^-30000 below the frame's best cannot be emitted (a logit of -inf, in practice): factor 0.  (The bound also
  // keeps every exponent sum of a 12 000-frame utterance inside int range.)
  const bool dead = !(l2 > -30000.0);
  const double fl = floor(l2);
  float mant = dead ? 0.f : (float)exp2(l2 - fl);
  int ex = dead ? 0 : (int)fl;
  if (mant >= 2.f) { mant = 1.f; ++ex; }                   // (2^0.99999999 rounds to 2.0f)
  if (row_ok) {
    logy[(long)i * CP + c] = (float)l2;
    *reinterpret_cast<f32x2*>(emis + ((long)i * CP + c) * 2) = f32x2{mant, __builtin_bit_cast(float, ex)};
  }
}

// One wave per (utterance, direction).  blockIdx.y: 0 = alpha, 1 = beta.
// Storage is lane-major, one (mantissa, exponent bits) record of 8 bytes per state: state u = lane*KPL + j at frame t is
// record (t*KPL + j)*64 + lane.  A state no path reaches has mantissa 0 and exponent EZ, far below every live one, so it
// never wins the alignment.
// States u >= U: beta's stay 0 by themselves (they only receive from states above them), alpha's hold numbers nobody reads.
template <int KPL>
__global__ __launch_bounds__(64) void ctc_alpha_beta_kernel(const float* __restrict__ emis, int T, int C,
                                                            const int* __restrict__ label_ids,
                                                            const int* __restrict__ label_off,
                                                            const int* __restrict__ seq_lens,
                                                            float* __restrict__ alpha, float* __restrict__ beta,
                                                            int* __restrict__ status) {
  constexpr int UP = KPL * 64;
  constexpr int EC = CP * 2;                // floats per emission row (mantissa, exponent bits)
  __shared__ __attribute__((aligned(16))) float E[2][TC * EC];
  const int b = blockIdx.x;
  const bool is_beta = blockIdx.y != 0;
  const int lane = threadIdx.x;
  const int blank = C - 1;
  const int* lab = label_ids + label_off[b];
  const int L = label_off[b + 1] - label_off[b];
  const int U = 2 * L + 1;
  const int Tb = seq_lens[b];

  // "Not enough time for target transition sequence": L + #adjacent repeats must fit in Tb
  int rep = 0;
  for (int i = 1 + lane; i < L; i += 64) rep += lab[i] == lab[i - 1];
  rep = (int)st::wave_sum((float)rep);
  const bool bad = Tb < 0 || Tb > T || L + rep > Tb || U > UP;
  if (bad) {
    if (!is_beta && lane == 0) status[b] = 1;
    return;
  }
  if (!is_beta && lane == 0) status[b] = 0;
  if (Tb == 0) return;      // no frames and (checked above) an empty label: p = 1, nothing to recurse over

  int coff[KPL];          // emission of each state's class (LDS float offset inside an emission row)
  bool valid[KPL], skip[KPL];
#pragma unroll
  for (int j = 0; j < KPL; ++j) {
    const int u = lane * KPL + j;
    valid[j] = u < U;
    const bool odd = (u & 1) && valid[j];
    const int li = (u - 1) >> 1;
    coff[j] = 2 * (odd ? lab[li] : blank);
    if (!is_beta) skip[j] = odd && u >= 3 && lab[li] != lab[li - 1];          // may arrive from u-2
    else skip[j] = odd && u + 2 < U && lab[li + 1] != lab[li];                  // may leave to u+2
  }

  const float* ly = emis + (long)b * T * EC;
  f32x2* dst = reinterpret_cast<f32x2*>(is_beta ? beta : alpha) + (long)b * T * UP + lane;
  auto put = [&](int t, const float (&m)[KPL], const int (&e)[KPL]) {
    f32x2* o = dst + (long)t * UP;
#pragma unroll
    for (int j = 0; j < KPL; ++j) o[j * 64] = f32x2{m[j], __builtin_bit_cast(float, e[j])};
  };

  // stage one 64-frame chunk of emissions [chunk*TC, +TC) into E[buf]; rows past T read row T-1
  constexpr int STG = TC * EC / 4 / 64;
  f32x4 stage[STG];
  auto chunk_load = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < STG; ++i) {
      int f = lane + 64 * i;                 // float4 index inside the chunk
      int t = min(chunk * TC + f / (EC / 4), T - 1);
      stage[i] = *reinterpret_cast<const f32x4*>(ly + (long)t * EC + (f % (EC / 4)) * 4);
    }
  };
  auto chunk_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < STG; ++i)
      *reinterpret_cast<f32x4*>(&E[buf][(lane + 64 * i) * 4]) = stage[i];
  };
  float sm[KPL];   // mantissa of alpha_t(u) resp. beta_t(u)
  int se[KPL];     // its exponent
  // One step of either recursion for the KPL states of a lane: out = (a + b + [skip] c) * emission, brought back to [0.5, 1);
  // sum_m / sum_e (optional) receive the aligned sum before the emission.
  // Measured (scripts/bench_ctc.py, B = 32, T' = 501, 150 labels, the three kernels alone: 155 us; this one 117): normalising
  // only every 8th frame (the mantissa can only grow in between, by < 6x per frame) removes 15 of the 108 instructions of a
  // frame and is SLOWER (172 us: two bodies, a branch per frame); writing the step stage by stage over the states with
  // scheduler fences, so that no instruction depends on its predecessor (a wave alone on its SIMD issues a dependent
  // instruction after 8.3 cycles, an independent one after 4.9: scripts/ubench/valu_rates.hip), changes nothing (157): the
  // compiler's order already runs at 4.2 cycles per instruction.  Ablations: no lattice stores -8 us, no LDS emission reads -18.
  auto step = [&](const float (&am)[KPL], const int (&ae)[KPL], const float (&bm_)[KPL], const int (&be_)[KPL],
                  const float (&cm)[KPL], const int (&ce)[KPL], const f32x2 (&em)[KPL], float (&om)[KPL], int (&oe)[KPL],
                  float* sum_m = nullptr, int* sum_e = nullptr) {
    float nm[KPL];
    int ne[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      int big;
      const float sum = aligned_sum3(am[j], ae[j], bm_[j], be_[j], cm[j], skip[j] ? ce[j] : EZ, big);   // (2^(EZ - big) = 0)
      if (sum_m) { sum_m[j] = sum; sum_e[j] = big; }
      const float v = sum * em[j][0];
      nm[j] = __builtin_amdgcn_frexp_mantf(v);
      // a state without a path (all predecessors zero, or a dead emission: a logit of -inf) goes back to EZ: a zero
      // mantissa under a live-scale exponent would win the next alignment and flush live neighbours more than 126 binades
      // below it (round 3 kept `big + ...` here: harmless for finite logits, wrong for masked classes)
      ne[j] = v > 0.f ? big + ibits(em[j][1]) + __builtin_amdgcn_frexp_expf(v) : EZ;
    }
#pragma unroll
    for (int j = 0; j < KPL; ++j) { om[j] = nm[j]; oe[j] = ne[j]; }
  };
  if (!is_beta) {
    // ---- alpha: t ascending ------------------------------------------------------------
    chunk_load(0);
    chunk_store(0);
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      const int u = lane * KPL + j;
      const f32x2 q = *reinterpret_cast<const f32x2*>(&E[0][coff[j]]);
      const bool on = u < 2 && valid[j] && q[0] > 0.f;
      sm[j] = on ? 0.5f * q[0] : 0.f;
      se[j] = on ? ibits(q[1]) + 1 : EZ;
    }
    put(0, sm, se);
    // one frame: alpha_t(u) = (alpha_{t-1}(u) + alpha_{t-1}(u-1) + [skip] alpha_{t-1}(u-2)) * y_t(u)
    auto frame = [&](int t, const float* e) {
      f32x2 em[KPL];
#pragma unroll
      for (int j = 0; j < KPL; ++j) em[j] = *reinterpret_cast<const f32x2*>(e + coff[j]);
      const float up1m = dpp_movef<SHR1>(sm[KPL - 1], 0.f);
      const int up1e = dpp_movei<SHR1>(se[KPL - 1], EZ);
      const float up2m = KPL >= 2 ? dpp_movef<SHR1>(sm[KPL >= 2 ? KPL - 2 : 0], 0.f) : dpp_movef<SHR1>(up1m, 0.f);
      const int up2e = KPL >= 2 ? dpp_movei<SHR1>(se[KPL >= 2 ? KPL - 2 : 0], EZ) : dpp_movei<SHR1>(up1e, EZ);
      float m1[KPL], m2[KPL];
      int e1[KPL], e2[KPL];
#pragma unroll
      for (int j = 0; j < KPL; ++j) {
        m1[j] = j >= 1 ? sm[j >= 1 ? j - 1 : 0] : up1m;
        e1[j] = j >= 1 ? se[j >= 1 ? j - 1 : 0] : up1e;
        m2[j] = j >= 2 ? sm[j >= 2 ? j - 2 : 0] : (j == 1 ? up1m : up2m);
        e2[j] = j >= 2 ? se[j >= 2 ? j - 2 : 0] : (j == 1 ? up1e : up2e);
      }
      step(sm, se, m1, e1, m2, e2, em, sm, se);
      put(t, sm, se);
    };
    const int nchunks = (Tb + TC - 1) / TC;
    for (int ch = 0; ch < nchunks; ++ch) {
      const int buf = ch & 1;
      if (ch + 1 < nchunks) chunk_load(ch + 1);
      const int t_lo = max(1, ch * TC), t_hi = min(Tb, (ch + 1) * TC);
      for (int t = t_lo; t < t_hi; ++t) {
        const float* e = &E[buf][(t - ch * TC) * EC];
        frame(t, e);
      }
      if (ch + 1 < nchunks) chunk_store(buf ^ 1);
    }
  } else {
    // ---- beta: t descending.  Carried: G_t(u) = beta_t(u) * y_t(u) (beta itself excludes the emission at t, and is what is
    // stored): beta_t(u) = G_{t+1}(u) + G_{t+1}(u+1) + [skip] G_{t+1}(u+2).  The emission is then needed at the END of a
    // step, as in alpha, and its LDS read has the whole step to arrive. ---------------------------------------------------
    const int last = (Tb - 1) / TC;
    chunk_load(last);
    chunk_store(last & 1);
    {
      const float* e = &E[last & 1][(Tb - 1 - last * TC) * EC];
#pragma unroll
      for (int j = 0; j < KPL; ++j) {
        const int u = lane * KPL + j;
        const bool on = valid[j] && u >= U - 2;
        sm[j] = on ? 0.5f : 0.f;               // 1 = 0.5 * 2^1
        se[j] = on ? 1 : EZ;
      }
      put(Tb - 1, sm, se);
#pragma unroll
      for (int j = 0; j < KPL; ++j) {
        const f32x2 q = *reinterpret_cast<const f32x2*>(e + coff[j]);
        sm[j] *= 0.5f * q[0];                  // stays in [0.5, 1) (or 0)
        se[j] = sm[j] > 0.f ? se[j] + ibits(q[1]) + 1 : EZ;
      }
    }
    auto frame = [&](int t, const float* e) {
      f32x2 em[KPL];
#pragma unroll
      for (int j = 0; j < KPL; ++j) em[j] = *reinterpret_cast<const f32x2*>(e + coff[j]);
      const float dn1m = dpp_movef<SHL1>(sm[0], 0.f);
      const int dn1e = dpp_movei<SHL1>(se[0], EZ);
      const float dn2m = KPL >= 2 ? dpp_movef<SHL1>(sm[KPL >= 2 ? 1 : 0], 0.f) : dpp_movef<SHL1>(dn1m, 0.f);
      const int dn2e = KPL >= 2 ? dpp_movei<SHL1>(se[KPL >= 2 ? 1 : 0], EZ) : dpp_movei<SHL1>(dn1e, EZ);
      float m1[KPL], m2[KPL], bm[KPL];
      int e1[KPL], e2[KPL], be[KPL];
#pragma unroll
      for (int j = 0; j < KPL; ++j) {
        m1[j] = j + 1 < KPL ? sm[j + 1 < KPL ? j + 1 : 0] : dn1m;
        e1[j] = j + 1 < KPL ? se[j + 1 < KPL ? j + 1 : 0] : dn1e;
        m2[j] = j + 2 < KPL ? sm[j + 2 < KPL ? j + 2 : 0] : (j + 2 == KPL ? dn1m : dn2m);
        e2[j] = j + 2 < KPL ? se[j + 2 < KPL ? j + 2 : 0] : (j + 2 == KPL ? dn1e : dn2e);
      }
      step(sm, se, m1, e1, m2, e2, em, sm, se, bm, be);       // beta_t(u) itself (before the emission) is what is stored
      put(t, bm, be);
    };
    for (int ch = last; ch >= 0; --ch) {
      const int buf = ch & 1;
      if (ch > 0) chunk_load(ch - 1);
      const int t_hi = min(Tb - 2, ch * TC + TC - 1), t_lo = ch * TC;
      for (int t = t_hi; t >= t_lo; --t) {
        const float* e = &E[buf][(t - ch * TC) * EC];
        frame(t, e);
      }
      if (ch > 0) chunk_store(buf ^ 1);
    }
  }
}

// grad[b,t,c] = scale * (y_t(c) - sum_{u: l'_u = c} alpha_t(u) beta_t(u) / p)
constexpr int GF = 16;   // frames per block (4 waves x 4)
template <int KPL>
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ logy, const float* __restrict__ alpha,
                                                       const float* __restrict__ beta, int T, int C,
                                                       const int* __restrict__ label_ids,
                                                       const int* __restrict__ label_off,
                                                       const int* __restrict__ seq_lens,
                                                       const int* __restrict__ status, float scale,
                                                       float* __restrict__ grad, RowMap2 gmap, int gcols,
                                                       float* __restrict__ loss, float* __restrict__ loss_lo, int lmax) {
  extern __shared__ __attribute__((aligned(16))) int smem[];
  int* pos_off = smem;                 // [32]
  int* pos_list = smem + 32;           // [lmax]
  float* wbuf = reinterpret_cast<float*>(smem + 32 + lmax);   // [4][lmax]
  constexpr int UP = KPL * 64;
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int* lab = label_ids + label_off[b];
  const int L = label_off[b + 1] - label_off[b];
  const int U = 2 * L + 1;
  const int Tb = seq_lens[b];
  const bool bad = status[b] != 0;
  const int blank = C - 1;
  auto sidx = [&](int u) { return (u % KPL) * 64 + u / KPL; };     // lane-major state index

  if (wave == 0) {
    // per-class position lists in increasing position order (fixed summation order)
    int cnt = 0;
    if (lane < blank) for (int i = 0; i < L; ++i) cnt += lab[i] == lane;
    int inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
    int start = inc - cnt;
    if (lane < 32) pos_off[lane] = lane < blank ? start : L;
    if (lane < blank) { int w = start; for (int i = 0; i < L; ++i) if (lab[i] == lane) pos_list[w++] = i; }
  }
  __syncthreads();

  // p(l|x) = alpha_{Tb-1}(U-1) + alpha_{Tb-1}(U-2) = pm * 2^pe; stays 1 for an utterance without frames
  int pe = 0;
  double logp2 = 0.0;
  if (!bad && Tb > 0) {
    const f32x2* al = reinterpret_cast<const f32x2*>(alpha) + ((long)b * T + (Tb - 1)) * UP;
    const f32x2 a1 = al[sidx(U - 1)], a2 = U > 1 ? al[sidx(U - 2)] : f32x2{0.f, __builtin_bit_cast(float, EZ)};
    const float sum = aligned_sum3(a1[0], ibits(a1[1]), a2[0], ibits(a2[1]), 0.f, EZ, pe);
    logp2 = sum > 0.f ? (double)pe + log2((double)sum) : -__builtin_inf();
  }
  if (blockIdx.x == 0 && tid == 0) {
    // -log p is known here far better than a float of its magnitude can hold (the lattice rounds 6e-8 RELATIVE per step; one
    // fp32 ulp of a loss of 1 239 is 1.2e-4): the double goes out as a (hi, lo) float pair, hi = the fp32 loss TF would return
    const double nll = -logp2 * 0.69314718055994530942;
    const float hi = bad ? __builtin_inff() : (float)nll;
    loss[b] = hi;
    if (loss_lo) loss_lo[b] = (bad || !(hi < __builtin_inff())) ? 0.f : (float)(nll - (double)hi);
  }

  float* wb = wbuf + wave * lmax;
  for (int k = 0; k < GF / 4; ++k) {
    const int t = blockIdx.x * GF + k * 4 + wave;
    const bool live = !bad && t < Tb && t < T;
    float blank_sum = 0.f, occ_scale = 1.f;
    if (live) {
      const f32x2* al = reinterpret_cast<const f32x2*>(alpha) + ((long)b * T + t) * UP;
      const f32x2* be = reinterpret_cast<const f32x2*>(beta) + ((long)b * T + t) * UP;
      float label_sum = 0.f;
      // in storage order (state lane*KPL + j is record j*64 + lane: 512 contiguous bytes per j), all loads of the frame issued
      // before the first is used
      f32x2 av[KPL], bv[KPL];
#pragma unroll
      for (int j = 0; j < KPL; ++j) { av[j] = al[j * 64 + lane]; bv[j] = be[j * 64 + lane]; }
#pragma unroll
      for (int j = 0; j < KPL; ++j) {
        const int u = lane * KPL + j;
        // alpha_t(u) beta_t(u) / 2^pe: of the order of 1 where it matters, 0 for states no path passes through
        float w = __builtin_ldexpf(av[j][0] * bv[j][0], ibits(av[j][1]) + ibits(bv[j][1]) - pe);
        w = u < U ? w : 0.f;                                       // (alpha's records beyond U hold numbers nobody may read)
        if (u & 1) { if (u < U) wb[u >> 1] = w; label_sum += w; } else blank_sum += w;
      }
      blank_sum = st::wave_sum(blank_sum);
      // sum_u alpha_t(u) beta_t(u) == p(l|x) for EVERY t: normalising by the frame's own total removes the rounding the two
      // recursions accumulated on their way to this frame (a common factor of all its states)
      const float total = blank_sum + st::wave_sum(label_sum);
      occ_scale = total > 0.f ? 1.f / total : 1.f;
      blank_sum *= occ_scale;
    }
    __syncthreads();
    if (t < T && lane < gcols) {
      float g = 0.f;
      if (live && lane < C) {
        float occ = blank_sum;
        if (lane < blank) {
          occ = 0.f;
          for (int i = pos_off[lane]; i < pos_off[lane + 1]; ++i) occ += wb[pos_list[i]];
          occ *= occ_scale;
        }
        g = (ex2(logy[((long)b * T + t) * CP + lane]) - occ) * scale;
      }
      grad[gmap.off(b, t) + lane] = g;
    }
    __syncthreads();
  }
}

// greedy decode: argmax per frame, collapse repeats, drop blanks; one block per utterance
__global__ __launch_bounds__(256) void ctc_greedy_kernel(const float* __restrict__ logits, RowMap2 map, int T, int C,
                                                         const int* __restrict__ seq_lens, int merge_repeated,
                                                         int* __restrict__ ids, int max_out,
                                                         int* __restrict__ out_lens, float* __restrict__ neg_sum) {
  extern __shared__ int ksm[];          // [T] argmax per frame, then scan scratch [256]
  int* kbuf = ksm;
  int* scan = ksm + T;
  float* fsum = reinterpret_cast<float*>(scan + 256);
  const int b = blockIdx.x, tid = threadIdx.x;
  const int Tb = min(seq_lens[b], T);
  const int blank = C - 1;
  float msum = 0.f;
  for (int t = tid; t < Tb; t += 256) {
    const float* row = logits + map.off(b, t);
    float best = row[0];
    int bi = 0;
    for (int c = 1; c < C; ++c) { float v = row[c]; if (v > best) { best = v; bi = c; } }   // first max wins
    kbuf[t] = bi;
    msum += best;
  }
  __syncthreads();
  // contiguous segment per thread so positions stay ordered
  const int seg = (Tb + 255) / 256;
  const int lo = min(tid * seg, Tb), hi = min(lo + seg, Tb);
  int cnt = 0;
  for (int t = lo; t < hi; ++t) {
    int k = kbuf[t];
    cnt += (k != blank) && !(merge_repeated && t > 0 && k == kbuf[t - 1]);
  }
  scan[tid] = cnt;
  fsum[tid] = msum;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    int v = tid >= o ? scan[tid - o] : 0;
    float f = tid + o < 256 ? fsum[tid + o] : 0.f;
    __syncthreads();
    scan[tid] += v;
    fsum[tid] += f;
    __syncthreads();
  }
  int w = scan[tid] - cnt;
  for (int t = lo; t < hi; ++t) {
    int k = kbuf[t];
    if ((k != blank) && !(merge_repeated && t > 0 && k == kbuf[t - 1])) {
      if (w < max_out) ids[(long)b * max_out + w] = k;
      ++w;
    }
  }
  if (tid == 255) out_lens[b] = scan[255];
  if (tid == 0) neg_sum[b] = -fsum[0];
}

int pick_kpl(int max_label_len) {
  static const int opts[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16};
  const int U = 2 * max_label_len + 1;
  for (int k : opts) if (k * 64 >= U) return k;
  return -1;
}

RowMap2 make_map2(const st_tensor3& t) {
  RowMap2 m;
  m.batch_stride = (long)t.t_pitch * t.c_pitch;
  m.row0 = (long)t.halo * t.c_pitch;
  m.row_stride = t.c_pitch;
  return m;
}

template <int KPL>
void launch_ab(int B, hipStream_t s, const float* emis, int T, int C, const int* ids, const int* off,
               const int* lens, float* alpha, float* beta, int* status) {
  hipLaunchKernelGGL((ctc_alpha_beta_kernel<KPL>), dim3(B, 2), dim3(64), 0, s, emis, T, C, ids, off, lens,
                     alpha, beta, status);
}

}  // namespace

extern "C" {

size_t st_ctc_ws(int batch, int frames, int max_label_len) {
  int kpl = pick_kpl(std::max(max_label_len, 0));
  if (kpl < 0 || batch <= 0 || frames <= 0) return 0;
  size_t rows = (size_t)batch * frames;
  // log2-softmax [rows][32] | emission factors [rows][32][2] | alpha, beta: (mantissa, exponent) records [rows][kpl*64][2]
  return rows * CP * sizeof(float) * 3 + 4 * rows * kpl * 64 * sizeof(float) + 512;
}

int st_ctc_loss_grad_f32(const st_tensor3* logits, const int32_t* label_ids, const int32_t* label_offsets,
                         const int32_t* seq_lens, int max_label_len, float grad_scale, float* loss,
                         const st_tensor3* grad, int32_t* status, void* workspace, size_t workspace_bytes,
                         void* stream) {
  return st_ctc_loss_grad_hilo_f32(logits, label_ids, label_offsets, seq_lens, max_label_len, grad_scale, loss, nullptr, grad, status,
                                   workspace, workspace_bytes, stream);
}

int st_ctc_loss_grad_hilo_f32(const st_tensor3* logits, const int32_t* label_ids, const int32_t* label_offsets,
                              const int32_t* seq_lens, int max_label_len, float grad_scale, float* loss, float* loss_lo,
                              const st_tensor3* grad, int32_t* status, void* workspace, size_t workspace_bytes,
                              void* stream) {
  ST_REQUIRE(logits && logits->base && grad && grad->base && label_ids && label_offsets && seq_lens && loss &&
                 status && workspace, "ctc: null argument");
  ST_REQUIRE(logits->channels >= 2 && logits->channels <= CP && logits->c_pitch >= CP && logits->c_pitch % 4 == 0,
             "ctc: num_classes must be 2..32 with c_pitch >= 32");
  ST_REQUIRE(grad->batch == logits->batch && grad->frames == logits->frames && grad->c_pitch >= logits->channels,
             "ctc: grad tensor mismatch");
  const int kpl = pick_kpl(max_label_len);
  ST_REQUIRE(kpl > 0, "ctc: label length %d exceeds 511", max_label_len);
  ST_REQUIRE(workspace_bytes >= st_ctc_ws(logits->batch, logits->frames, max_label_len), "ctc: workspace too small");
  hipStream_t s = st::as_stream(stream);
  const int B = logits->batch, T = logits->frames, C = logits->channels;
  const size_t rows = (size_t)B * T;
  float* logy = reinterpret_cast<float*>(workspace);
  float* emis = logy + rows * CP;
  float* alpha = emis + rows * CP * 2;             // (mantissa, exponent) records
  float* beta = alpha + rows * kpl * 64 * 2;
  hipLaunchKernelGGL(ctc_logsoftmax_kernel, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, s, logits->base,
                     make_map2(*logits), B, T, C, logy, emis);
  switch (kpl) {
#define ST_AB(K) case K: launch_ab<K>(B, s, emis, T, C, label_ids, label_offsets, seq_lens, alpha, beta, status); break;
    ST_AB(1) ST_AB(2) ST_AB(3) ST_AB(4) ST_AB(5) ST_AB(6) ST_AB(8) ST_AB(10) ST_AB(12) ST_AB(16)
#undef ST_AB
  }
  if (int e = st::check_launch("ctc_alpha_beta")) return e;
  const int lmax = std::max(1, kpl * 32);
  const size_t shm = (32 + (size_t)lmax * 5) * sizeof(int);
  switch (kpl) {
#define ST_GR(K) case K: hipLaunchKernelGGL(ctc_grad_kernel<K>, dim3(st::ceil_div(T, GF), B), dim3(256), shm, s, logy, alpha, beta, \
                                            T, C, label_ids, label_offsets, seq_lens, status, grad_scale, grad->base,            \
                                            make_map2(*grad), std::min(grad->c_pitch, CP), loss, loss_lo, lmax); break;
    ST_GR(1) ST_GR(2) ST_GR(3) ST_GR(4) ST_GR(5) ST_GR(6) ST_GR(8) ST_GR(10) ST_GR(12) ST_GR(16)
#undef ST_GR
  }
  return st::check_launch("ctc_grad");
}

int st_ctc_greedy_decode(const st_tensor3* logits, const int32_t* seq_lens, int merge_repeated, int32_t* ids,
                         int max_out, int32_t* out_lens, float* neg_sum_logits, void* stream) {
  ST_REQUIRE(logits && logits->base && seq_lens && ids && out_lens && neg_sum_logits, "greedy: null argument");
  ST_REQUIRE(logits->channels >= 2 && max_out >= 1, "greedy: bad shape");
  ST_REQUIRE(logits->frames <= 12000, "greedy: more than 12000 frames per utterance not supported");
  const size_t shm = ((size_t)logits->frames + 512) * sizeof(int);
  hipLaunchKernelGGL(ctc_greedy_kernel, dim3(logits->batch), dim3(256), shm, st::as_stream(stream), logits->base,
                     make_map2(*logits), logits->frames, logits->channels, seq_lens, merge_repeated, ids, max_out,
                     out_lens, neg_sum_logits);
  return st::check_launch("ctc_greedy");
}

}  // extern "C"
