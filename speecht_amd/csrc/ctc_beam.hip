// LM-free CTC prefix beam search (top path) on gfx950 -- SURVEY 8(f) item 3, BASELINE config 5.
//
// The reference reaches a beam search only through its KenLM TensorFlow fork (speech_model.py:101-111);
// this kernel follows the stock tf.nn.ctc_beam_search_decoder recursion without a scorer, with the
// candidate order of oracle/w2l_oracle.py::ctc_beam_search_decode (total desc, slot*C + c asc).
//
// Mapping: ONE wavefront per utterance (the recursion is sequential in time, utterances are the
// parallel axis), so a frame is a chain of dependent steps on one wave and its length IS the decode time
// (round 2: 13 300 cycles per frame, measured per phase with s_memtime; round 3: the chain below).
// What does not depend on the beam is taken off the chain: the log-softmax of every frame is computed by a
// fully parallel kernel first (same wave reductions, bit-identical values) and read one frame ahead.
// Everything that lives across frames -- the beam entries -- sits in LDS, double buffered; a frame is:
// parent matching (W^2 hash compares spread over the lanes, all LDS reads of the four passes issued before the
// first compare), the stay candidates, one sortable 32-bit score per candidate (W*C of them, lane-strided, in
// registers; branch-free, three LDS reads each), and a wave-parallel selection of the W best: the W-th largest
// LANE maximum -- found by a 32-step radix select on ballots, not by ranking the lanes -- bounds the W-th largest
// candidate from below, the few candidates at or above it are compacted into LDS and ranked by counting
// (sequential maximum rounds remain for the rare case of more than 64 survivors).  Prefix identity is a
// 64-bit mixed hash + length (the trie TF keeps in host memory would be a pointer chase per candidate);
// the emitted labels are recorded as (parent node, label) pairs in a per-utterance pool whose slot is a
// pure function of (frame, rank), so there are no atomics and the result is deterministic.
#include <math.h>

#include "st_common.h"

namespace {

constexpr int kMaxBeam = 128;          // beams of 65..128 (the reference's operating point is 100) run the WIDE instantiation
constexpr int kMaxClasses = 32;
constexpr unsigned long long kRootHash = 0x243F6A8885A308D3ull;

struct RowMap {   // (b, t) -> float offset into a padded NWC tensor
  long batch_stride;
  long row0;
  int row_stride;
  __device__ __forceinline__ long off(int b, int t) const { return (long)b * batch_stride + row0 + (long)t * row_stride; }
};

template <int MAXB>
struct BeamSet {   // structure of arrays: lane r reads/writes entry r without bank conflicts
  unsigned long long hash[MAXB];
  unsigned long long parent_hash[MAXB];
  int len[MAXB];
  int last[MAXB];
  int node[MAXB];
  float pb[MAXB];
  float pl[MAXB];
  float total[MAXB];
};

__device__ __forceinline__ unsigned long long child_hash(unsigned long long h, int c) {
  unsigned long long x = h + 0x9E3779B97F4A7C15ull * (unsigned long long)(c + 1);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__device__ __forceinline__ float lse(float a, float b) {
  // hardware exp2 / log2 (1 ulp): the two calls sit on the per-frame dependent chain of the search, and the libm forms
  // (range reduction + polynomial, ~40 instructions each) were a fifth of the `stay` phase; the result differs from them
  // by < 2e-7, far inside the 1e-4 the scores are held to
  float hi = fmaxf(a, b), lo = fminf(a, b);
  return hi == -INFINITY ? -INFINITY : hi + __logf(1.f + __expf(lo - hi));
}

// ---- wave64 reductions on the DPP network (row shifts + row broadcasts; no LDS round trips) --------
template <int CTRL, int ROW_MASK, bool ZERO_INVALID>
__device__ __forceinline__ int dpp_i32(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, ZERO_INVALID);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#define ST_STEP(CTRL) v = max(v, (unsigned)dpp_i32<CTRL, 0xf, false>((int)v, (int)v))   /* invalid source lane -> own value */
  ST_STEP(0x111); ST_STEP(0x112); ST_STEP(0x114); ST_STEP(0x118);   // row_shr 1, 2, 4, 8: lane 15 of a row = row maximum
  ST_STEP(0x142); ST_STEP(0x143);                                   // row_bcast 15, 31: lane 63 = wave maximum
#undef ST_STEP
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#define ST_STEP(CTRL) v = min(v, (unsigned)dpp_i32<CTRL, 0xf, false>((int)v, (int)v))
  ST_STEP(0x111); ST_STEP(0x112); ST_STEP(0x114); ST_STEP(0x118); ST_STEP(0x142); ST_STEP(0x143);
#undef ST_STEP
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ float wave_max_f32(float v) {
#define ST_STEP(CTRL) v = fmaxf(v, __int_as_float(dpp_i32<CTRL, 0xf, false>(__float_as_int(v), __float_as_int(v))))
  ST_STEP(0x111); ST_STEP(0x112); ST_STEP(0x114); ST_STEP(0x118); ST_STEP(0x142); ST_STEP(0x143);
#undef ST_STEP
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_sum_f32(float v) {
#define ST_STEP(CTRL, MASK) v += __int_as_float(dpp_i32<CTRL, MASK, true>(0, __float_as_int(v)))
  ST_STEP(0x111, 0xf); ST_STEP(0x112, 0xf); ST_STEP(0x114, 0xf); ST_STEP(0x118, 0xf);   // scan inside each row
  ST_STEP(0x142, 0xa); ST_STEP(0x143, 0xc);                                            // fold rows into lane 63
#undef ST_STEP
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// inclusive prefix sum over the wave (lane 63 holds the total)
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
#define ST_STEP(CTRL, MASK) v += dpp_i32<CTRL, MASK, true>(0, v)
  ST_STEP(0x111, 0xf); ST_STEP(0x112, 0xf); ST_STEP(0x114, 0xf); ST_STEP(0x118, 0xf);   // scan inside each row of 16
  ST_STEP(0x142, 0xa); ST_STEP(0x143, 0xc);                                            // carry the rows' totals up
#undef ST_STEP
  return v;
}

// monotone map float -> unsigned (larger float, larger integer) and back
__device__ __forceinline__ unsigned order_bits(float v) {
  const unsigned bits = (unsigned)__float_as_int(v);
  return bits ^ ((bits >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
constexpr unsigned kOrdNegInf = 0x007FFFFFu;      // order_bits(-inf)
__device__ __forceinline__ float order_value(unsigned ord) {
  return __int_as_float((int)(ord ^ ((ord >> 31) ? 0x80000000u : 0xFFFFFFFFu)));
}

// log-softmax of every frame, one wavefront per frame: out[(b * T + t) * 32 + c] (0 for c >= C).  The same wave reductions
// in the same order as the search kernel used when it did this inside its frame loop, so the values are bit-identical;
// frames at or beyond an utterance's length are skipped.
// transform 1: the decoder's input is tf.log(tf.nn.softmax(logits) + 1e-8) / log(10) -- what the reference hands its beam search
// (speech_model.py:102) -- and the decoder then normalises THAT per frame like any other input (ctc_beam_search.h Step).
__global__ __launch_bounds__(256) void logsoftmax_rows_kernel(const float* __restrict__ logits, RowMap map, int T, int C,
                                                              const int* __restrict__ seq_lens, int transform, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, b = blockIdx.y;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= min(seq_lens[b], T)) return;
  const float x = lane < C ? logits[map.off(b, t) + lane] : -INFINITY;
  const float m = wave_max_f32(x);
  const float z = wave_sum_f32(lane < C ? expf(x - m) : 0.f);
  if (transform == 1) {
    // in double (off the search's chain, one wave per frame): the transform flattens the distribution (a temperature of ln 10),
    // near-ties at the beam boundary get closer, and the stored row should be the fp32 rounding of the exact value
    auto wsum = [](double v) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      return v;
    };
    const double e = lane < C ? exp((double)x - (double)m) : 0.0;
    const double zz = wsum(e);
    const double xt = lane < C ? log10(e / zz + 1e-8) : -INFINITY;
    const float mt = wave_max_f32((float)xt);                       // any value near the maximum centres the sum
    const double z2 = wsum(lane < C ? exp(xt - (double)mt) : 0.0);
    if (lane < kMaxClasses) out[((long)b * T + t) * kMaxClasses + lane] = lane < C ? (float)(xt - (double)mt - log(z2)) : 0.f;
    return;
  }
  if (lane < kMaxClasses) out[((long)b * T + t) * kMaxClasses + lane] = lane < C ? x - m - logf(z) : 0.f;
}

// The search kernel runs ONE wavefront per workgroup, and a wave's LDS operations execute in order: between its phases it
// needs no s_barrier, only the compiler kept from moving LDS accesses across the phase boundary and the LDS queue drained.
// __syncthreads() would also wait for vmcnt(0) -- i.e. for the next frame's log-probabilities, whose load is meant to stay
// in flight under this frame's work.
#define ST_WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// CPL = candidates per lane (beam_width * C <= 64 * CPL).  MAXB = 64: candidate k = lane + 64 j is (slot, class) = (k / C, k % C).
// MAXB = 128 (WIDE, beams of 65..128; CPL = 64): the classes take a pitch of 32 -- k = slot * 32 + class, the same order between live
// candidates -- so a lane's class is lane & 31 for all its candidates and its slots are (lane >> 5) + 2 j: no division, no index
// arrays; two beam entries per lane in the per-entry phases; the selection bound is the exact W-th largest score (a bitwise binary
// search over the sortable scores: 32 wave-parallel counts) instead of the lane / pair / quad maxima, which bound only 64.
template <int CPL, int MAXB>
__global__ __launch_bounds__(64) void ctc_beam_kernel(const float* __restrict__ lp_all, int T, int C,
                                                      const int* __restrict__ seq_lens, int W,
                                                      int2* __restrict__ node_pool, long pool_stride,
                                                      int* __restrict__ ids, int max_out,
                                                      int* __restrict__ out_lens, float* __restrict__ out_logp) {
  constexpr bool WIDE = MAXB > 64;
  constexpr int SURV = WIDE ? 256 : 128;             // capacity of the fast selection; more survivors take the sequential rounds
  static_assert(MAXB == 64 || (MAXB == 128 && CPL == 64), "wide beams: 128 entries x 32 class slots = 64 candidates per lane");
  __shared__ BeamSet<MAXB> sets[2];
  __shared__ float lp_s[kMaxClasses];
  // per beam entry, for the scoring pass: {total, p_blank of the previous frame, total of the stay candidate, last label}
  __shared__ __attribute__((aligned(16))) float4 slot_s[MAXB];
  __shared__ float stay_pb[MAXB], stay_pl[MAXB];
  __shared__ int parent_of[MAXB];
  __shared__ unsigned dead[MAXB];
  __shared__ int sel_k[MAXB];
  __shared__ __attribute__((aligned(16))) unsigned long long surv[SURV + 16];   // compacted survivor keys of the fast selection (+ zero pad)
  __shared__ float sel_v[MAXB];

  const int b = blockIdx.x, lane = threadIdx.x;
  const int Tb = min(seq_lens[b], T);
  const int blank = C - 1;
  int2* nodes = node_pool + (long)b * pool_stride;

  // (slot, class) of this lane's candidates k = lane + 64 j: fixed for the whole utterance
  constexpr int NIDX = WIDE ? 1 : CPL;               // (wide beams compute them on the fly: slot (lane >> 5) + 2 j, class lane & 31)
  int cslot[NIDX], ccls[NIDX];
#pragma unroll
  for (int j = 0; j < NIDX; ++j) {
    int k = lane + 64 * j;
    cslot[j] = k / C;
    ccls[j] = k - cslot[j] * C;
  }
  const int wslot0 = lane >> 5, wcls = lane & 31;

  int cur = 0, nb = 1;
  double offset = 0.0;                   // log-probability of the current best entry (wave-uniform)
  if (lane == 0) {
    BeamSet<MAXB>& s = sets[0];
    s.hash[0] = kRootHash; s.parent_hash[0] = 0; s.len[0] = 0; s.last[0] = -1; s.node[0] = 0;
    s.pb[0] = 0.f; s.pl[0] = -INFINITY; s.total[0] = 0.f;
  }
  __syncthreads();

  // log-softmax rows of this utterance, written by logsoftmax_rows_kernel: [t][32]
  const float* __restrict__ lp_rows = lp_all + (long)b * T * kMaxClasses;
  float lp_next = (lane < kMaxClasses && Tb > 0) ? lp_rows[lane] : 0.f;
  for (int t = 0; t < Tb; ++t) {
    const BeamSet<MAXB>& S = sets[cur];
    BeamSet<MAXB>& N = sets[cur ^ 1];
    // (1) this frame's log-probabilities into LDS; the next frame's row is already on its way
    if (lane < kMaxClasses) lp_s[lane] = lp_next;
    if (t + 1 < Tb && lane < kMaxClasses) lp_next = lp_rows[(long)(t + 1) * kMaxClasses + lane];
#pragma unroll
    for (int e = lane; e < MAXB; e += 64) {
      parent_of[e] = -1;
      dead[e] = 0u;
    }
    ST_WAVE_SYNC();
    // (2) which entries have their parent prefix in the beam?  (e, p) pairs spread over the lanes; the LDS reads of all
    // passes are issued before the first compare (a pass at a time each one waits a full LDS round trip)
    if constexpr (WIDE) {
      // two entries per lane against every candidate parent p in turn (broadcast reads of p's fields); prefix identity is
      // unique in the beam, so an entry matches at most one p
      int len_e[2], last_e[2];
      unsigned long long he[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int e = lane + 64 * i;
        const bool in = e < nb;
        len_e[i] = in ? S.len[e] : -7;
        he[i] = S.parent_hash[in ? e : 0];
        last_e[i] = S.last[in ? e : 0];
      }
#pragma unroll 4
      for (int p = 0; p < nb; ++p) {
        const int len_p = S.len[p];
        const unsigned long long hp = S.hash[p];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (len_p + 1 == len_e[i] && hp == he[i]) {
            parent_of[lane + 64 * i] = p;
            atomicOr(&dead[p], 1u << last_e[i]);
          }
        }
      }
    } else {
      int lg = 32 - __clz(max(nb - 1, 1));
      if (nb <= 1) lg = 0;
      const int wp = 1 << lg, pairs = nb << lg;
      constexpr int PASSES = 4;                      // 64 * 4 = 16 x 16 pairs; wider beams loop
      if (wp <= 64 && pairs <= 64 * PASSES) {
        // p = lane & (wp - 1) for every pass: the candidate parent's fields are read once
        const int p = lane & (wp - 1);
        const bool p_in = p < nb;
        const int len_p = S.len[p_in ? p : 0];
        const unsigned long long hp = S.hash[p_in ? p : 0];
        int e_[PASSES], len_e[PASSES], last_e[PASSES];
        unsigned long long he[PASSES];
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
          const int idx = lane + 64 * i;
          const bool in = idx < pairs && p_in;
          e_[i] = in ? idx >> lg : 0;
          len_e[i] = in ? S.len[e_[i]] : -7;
          he[i] = S.parent_hash[e_[i]];
          last_e[i] = S.last[e_[i]];
        }
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
          if (len_p + 1 == len_e[i] && hp == he[i]) {
            parent_of[e_[i]] = p;
            atomicOr(&dead[p], 1u << last_e[i]);         // child (p, last[e]) already exists: merged into e's stay
          }
        }
      } else {
        for (int idx = lane; idx < pairs; idx += 64) {
          const int e = idx >> lg, p = idx & (wp - 1);
          if (p < nb && S.len[p] + 1 == S.len[e] && S.hash[p] == S.parent_hash[e]) {
            parent_of[e] = p;
            atomicOr(&dead[p], 1u << S.last[e]);
          }
        }
      }
    }
    ST_WAVE_SYNC();
    // (2b) the stay candidate of entry `lane` (wide beams: and of entry lane + 64)
#pragma unroll
    for (int e = lane; e < MAXB; e += 64) {
      if (e >= nb) break;
      // every LDS read up front (own fields, then the parent's through a clamped index), the arithmetic behind them
      const float tot = S.total[e], pl0 = S.pl[e], pb0 = S.pb[e];
      const int len0 = S.len[e], last0 = S.last[e], p = parent_of[e];
      const float lp_blank = lp_s[blank], lp_last = lp_s[max(last0, 0)];
      const int pc = max(p, 0);
      const int plen = S.len[pc], plast = S.last[pc];
      const float ppb = S.pb[pc], ptot = S.total[pc];
      const float npb = tot + lp_blank;
      float npl = -INFINITY;
      if (len0 > 0) {
        float mass = pl0;
        if (p >= 0) mass = lse(mass, (plen > 0 && plast == last0) ? ppb : ptot);
        npl = mass + lp_last;
      }
      stay_pb[e] = npb;
      stay_pl[e] = npl;
      slot_s[e] = make_float4(tot, pb0, lse(npb, npl), __int_as_float(last0));
    }
    ST_WAVE_SYNC();
    // (3) one sortable 32-bit score per candidate, in registers (candidate index k = lane + 64 j is implicit)
    unsigned ord[CPL];
    unsigned best_ord = 0u;                               // 0 is below every real candidate
    int best_j = 0;
    if constexpr (!WIDE) {
      // all reads first (three per candidate, none behind a branch), then the arithmetic
      float4 info[CPL];
      unsigned dd[CPL];
      float lpc[CPL];
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const int sl = cslot[j] < nb ? cslot[j] : 0;
        info[j] = slot_s[sl];
        dd[j] = dead[sl];
        lpc[j] = lp_s[ccls[j]];
      }
#pragma unroll
      for (int j = CPL - 1; j >= 0; --j) {                // descending j + ">=": the lowest j wins a tie
        const int c = ccls[j];
        const float child = ((dd[j] >> c) & 1u) ? -INFINITY : ((__float_as_int(info[j].w) == c) ? info[j].y : info[j].x) + lpc[j];
        const float v = c == blank ? info[j].z : child;
        const unsigned oj = cslot[j] < nb ? order_bits(v) : 0u;
        ord[j] = oj;
        if (oj >= best_ord) { best_ord = oj; best_j = j; }
      }
    } else {
      // one class per lane (the 32 lanes of a half read the same entry: broadcast reads), eight entries at a time
      const int c = wcls;
      const float lpc = lp_s[c];
      const bool cls_live = c < C;
#pragma unroll
      for (int j0 = 0; j0 < CPL; j0 += 8) {
        float4 info[8];
        unsigned dd[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const int sl = wslot0 + 2 * (j0 + jj);
          info[jj] = slot_s[sl < nb ? sl : 0];
          dd[jj] = dead[sl < nb ? sl : 0];
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const int sl = wslot0 + 2 * (j0 + jj);
          const float child = ((dd[jj] >> c) & 1u) ? -INFINITY : ((__float_as_int(info[jj].w) == c) ? info[jj].y : info[jj].x) + lpc;
          const float v = c == blank ? info[jj].z : child;
          ord[j0 + jj] = (sl < nb && cls_live) ? order_bits(v) : 0u;
        }
      }
#pragma unroll
      for (int j = CPL - 1; j >= 0; --j)                  // descending j + ">=": the lowest j wins a tie
        if (ord[j] >= best_ord) { best_ord = ord[j]; best_j = j; }
    }
    // (4) selection of the W best candidates, best first.
    // Fast path (threshold + rank): a cheap lower bound of the W-th largest candidate; only candidates at or above it
    // ("survivors", a few dozen) can be selected; they are compacted into LDS and every survivor finds its rank by counting
    // the survivors ahead of it (keys are unique: score, then smaller candidate index).  All of it is wave-parallel; the
    // sequential rounds below remain as the fallback for more than 128 survivors.
    int n_new = 0;
    bool selected = false;
    {
      // A lower bound of the W-th largest candidate that costs ten instructions: the lanes form 64 / 4 = 16 quads, every quad
      // maximum is a candidate, so at least 16 candidates are >= the SMALLEST quad maximum; for beams of 17 - 32 the 32 lane
      // pairs take the quads' place, for 33 - 64 the lanes themselves (round 4: those widths took the sequential path).  It
      // is looser than the W-th largest lane maximum of round 2 (flat posteriors: ~48 survivors instead of ~20 at beam 16)
      // but that bound cost a ranking of the 64 lanes -- 3 800 of the frame's 13 300 cycles as 64 v_readlane steps, 1 700 as
      // a 32-step radix select on ballots, more again as 16 four-key LDS reads -- and the survivors are ranked by counting
      // anyway; more than 128 survivors fall through to the sequential rounds.
      unsigned thr;
      if constexpr (!WIDE) {
        unsigned g = best_ord;
        if (W <= 32) g = max(g, (unsigned)dpp_i32<0xB1, 0xf, false>((int)g, (int)g));        // quad_perm [1, 0, 3, 2]: pairs
        if (W <= 16) g = max(g, (unsigned)dpp_i32<0x4E, 0xf, false>((int)g, (int)g));        // quad_perm [2, 3, 0, 1]: quads
        // (a group without a live candidate: every live candidate survives)
        thr = max(wave_min_u32(g), kOrdNegInf + 1u);
      } else {
        // beams wider than the wave: the EXACT W-th largest score, bit by bit from the top -- the largest value v with at least W
        // candidates >= v (dead candidates score 0 or order_bits(-inf): fewer than W live ones leave v below every live score)
        unsigned prefix = 0u;
        for (int bit = 31; bit >= 0; --bit) {
          const unsigned trial = prefix | (1u << bit);
          int cnt = 0;
#pragma unroll
          for (int j = 0; j < CPL; ++j) cnt += ord[j] >= trial ? 1 : 0;
          cnt = __builtin_amdgcn_readlane(wave_incl_scan_i32(cnt), 63);
          if (cnt >= W) prefix = trial;
        }
        thr = max(prefix, kOrdNegInf + 1u);
      }
      // compaction of the survivors into LDS: a lane's survivors sit behind those of the lanes below it (one wave prefix
      // sum instead of a ballot per candidate row; the keys carry the candidate index, so the order does not matter)
      int mine_n = 0;
#pragma unroll
      for (int j = 0; j < CPL; ++j) mine_n += (ord[j] >= thr && ord[j] > kOrdNegInf) ? 1 : 0;
      const int incl = wave_incl_scan_i32(mine_n);
      const int base = __builtin_amdgcn_readlane(incl, 63);
      if (base <= SURV) {
        int pos = incl - mine_n;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          if (ord[j] >= thr && ord[j] > kOrdNegInf)
            surv[pos++] = ((unsigned long long)ord[j] << 32) | (0xFFFFFFFFu - (unsigned)(lane + 64 * j));
        }
        selected = true;
        n_new = min(base, W);
        if (lane < 16) surv[min(base + lane, SURV + 15)] = 0ull; // pad: the counting loop reads up to sixteen keys past the end
        ST_WAVE_SYNC();
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        // every survivor finds its rank by counting the survivors ahead of it (keys are unique); lane l ranks survivors
        // l and l + 64
        const bool has0 = lane < base, has1 = lane + 64 < base;
        const unsigned long long mine0 = surv[has0 ? lane : 0], mine1 = surv[has1 ? lane + 64 : 0];
        int rank0 = 0, rank1 = 0;
        if constexpr (WIDE) {
          // lane l ranks survivors l, l + 64, l + 128, l + 192 against all of them, two keys per (broadcast) read
          const bool has2 = lane + 128 < base, has3 = lane + 192 < base;
          const unsigned long long mine2 = surv[has2 ? lane + 128 : 0], mine3 = surv[has3 ? lane + 192 : 0];
          int rank2 = 0, rank3 = 0;
          for (int t2 = 0; t2 < base; t2 += 2) {
            const u64x2 a = *reinterpret_cast<const u64x2*>(&surv[t2]);
            rank0 += (a[0] > mine0) + (a[1] > mine0);
            rank1 += (a[0] > mine1) + (a[1] > mine1);
            rank2 += (a[0] > mine2) + (a[1] > mine2);
            rank3 += (a[0] > mine3) + (a[1] > mine3);
          }
          if (has2 && rank2 < W) {
            sel_k[rank2] = (int)(0xFFFFFFFFu - (unsigned)mine2);
            sel_v[rank2] = order_value((unsigned)(mine2 >> 32));
          }
          if (has3 && rank3 < W) {
            sel_k[rank3] = (int)(0xFFFFFFFFu - (unsigned)mine3);
            sel_v[rank3] = order_value((unsigned)(mine3 >> 32));
          }
        } else if (base <= 64) {                                // (wave-uniform)
          // eight keys per trip, the next trip's reads issued before this trip's compares (the trip count is small and
          // every trip would otherwise wait out a full LDS round trip)
          u64x2 q[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) q[k] = *reinterpret_cast<const u64x2*>(&surv[2 * k]);
          for (int t2 = 0; t2 < base; t2 += 8) {
            u64x2 n[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) n[k] = *reinterpret_cast<const u64x2*>(&surv[min(t2 + 8, SURV) + 2 * k]);
#pragma unroll
            for (int k = 0; k < 4; ++k) rank0 += (q[k][0] > mine0) + (q[k][1] > mine0);
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = n[k];
          }
        } else {
          for (int t2 = 0; t2 < base; t2 += 4) {
            const u64x2 a = *reinterpret_cast<const u64x2*>(&surv[t2]), c2 = *reinterpret_cast<const u64x2*>(&surv[t2 + 2]);
            rank0 += (a[0] > mine0) + (a[1] > mine0) + (c2[0] > mine0) + (c2[1] > mine0);
            rank1 += (a[0] > mine1) + (a[1] > mine1) + (c2[0] > mine1) + (c2[1] > mine1);
          }
        }
        if (has0 && rank0 < W) {
          sel_k[rank0] = (int)(0xFFFFFFFFu - (unsigned)mine0);
          sel_v[rank0] = order_value((unsigned)(mine0 >> 32));
        }
        if (has1 && rank1 < W) {
          sel_k[rank1] = (int)(0xFFFFFFFFu - (unsigned)mine1);
          sel_v[rank1] = order_value((unsigned)(mine1 >> 32));
        }
      }
    }
    // Fallback: up to W rounds -- wave maximum of the scores on the DPP network, the winner is the tied lane
    // with the smallest candidate index (one lane almost always: ballot + find-first; a real tie takes a second
    // reduction); only the winner's lane retires its candidate and refreshes its local best
    for (int r = 0; !selected && r < W; ++r) {
      const unsigned top = wave_max_u32(best_ord);
      if (top <= kOrdNegInf) break;                        // nothing with non-zero probability left
      const unsigned long long tied = __ballot(best_ord == top);
      int owner = __builtin_ctzll(tied);
      if (__builtin_popcountll(tied) > 1) {
        const unsigned kmin = wave_min_u32(best_ord == top ? (unsigned)(lane + 64 * best_j) : 0xFFFFFFFFu);
        owner = (int)(kmin & 63u);
      }
      const int jw = __builtin_amdgcn_readlane(best_j, owner);
      if (lane == 0) { sel_k[r] = owner + 64 * jw; sel_v[r] = order_value(top); }
      n_new = r + 1;
      if (lane == owner) {
        best_ord = 0u;
        best_j = 0;
#pragma unroll
        for (int j = CPL - 1; j >= 0; --j) {
          if (j == jw) ord[j] = 0u;
          if (ord[j] >= best_ord) { best_ord = ord[j]; best_j = j; }
        }
      }
    }
    ST_WAVE_SYNC();
    // (5) materialise the surviving entries, best first.  Scores are kept RELATIVE to the best entry (its
    // total becomes 0) with the running offset in double: after 1500 frames the absolute log-probabilities
    // are O(-3000), where fp32 resolves only 2e-4 and near-ties at the beam boundary would be decided by
    // rounding; relative scores stay O(10) for the whole utterance.
    const float top = n_new > 0 ? sel_v[0] : 0.f;
    offset += (double)top;
#pragma unroll
    for (int e = lane; e < MAXB; e += 64) {
      if (e >= n_new) break;
      const int k = sel_k[e];
      const float v = sel_v[e];
      const int slot = WIDE ? k >> 5 : k / C, c = WIDE ? k & 31 : k - slot * C;
      // the parent entry's fields, read before the stay / child decision
      const unsigned long long h0 = S.hash[slot], ph0 = S.parent_hash[slot];
      const int len0 = S.len[slot], last0 = S.last[slot], node0 = S.node[slot];
      const float spb = stay_pb[slot], spl = stay_pl[slot];
      const bool stay = c == blank;
      const int id = 1 + t * W + e;
      if (!stay) nodes[id] = make_int2(node0, c);
      N.hash[e] = stay ? h0 : child_hash(h0, c);
      N.parent_hash[e] = stay ? ph0 : h0;
      N.len[e] = stay ? len0 : len0 + 1;
      N.last[e] = stay ? last0 : c;
      N.node[e] = stay ? node0 : id;
      N.pb[e] = stay ? spb - top : -INFINITY;
      N.pl[e] = stay ? spl - top : v - top;
      N.total[e] = v - top;
    }
    nb = n_new;
    cur ^= 1;
    ST_WAVE_SYNC();
  }

  __syncthreads();
  // top path: entry 0 of the final set; walk the node chain backwards
  if (lane == 0) {
    const BeamSet<MAXB>& S = sets[cur];
    int n = min(S.len[0], max_out);
    out_lens[b] = S.len[0];
    out_logp[b] = (float)((double)S.total[0] + offset);
    int id = S.node[0];
    for (int i = S.len[0] - 1; i >= 0; --i) {
      int2 nd = nodes[id];
      if (i < n) ids[(long)b * max_out + i] = nd.y;
      id = nd.x;
    }
  }
}

}  // namespace

extern "C" {

static size_t beam_pool_bytes(int batch, int frames, int beam_width) {
  return st::round_up((size_t)batch * ((size_t)frames * beam_width + 1) * sizeof(int2), 256);
}

// [node pool: (parent, label) pairs | log-softmax rows [batch][frames][32]]
size_t st_ctc_beam_ws(int batch, int frames, int beam_width) {
  if (batch <= 0 || frames < 0 || beam_width <= 0) return 0;
  return beam_pool_bytes(batch, frames, beam_width) + (size_t)batch * frames * kMaxClasses * sizeof(float);
}

int st_ctc_beam_search_decode(const st_tensor3* logits, const int32_t* seq_lens, int beam_width, int32_t* ids,
                              int max_out, int32_t* out_lens, float* log_prob, void* workspace,
                              size_t workspace_bytes, void* stream) {
  return st_ctc_beam_search_decode_ex(logits, seq_lens, beam_width, 0, ids, max_out, out_lens, log_prob, workspace, workspace_bytes, stream);
}

int st_ctc_beam_search_decode_ex(const st_tensor3* logits, const int32_t* seq_lens, int beam_width, int input_transform, int32_t* ids,
                                 int max_out, int32_t* out_lens, float* log_prob, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  ST_REQUIRE(logits && logits->base && seq_lens && ids && out_lens && log_prob, "beam search: null argument");
  ST_REQUIRE(input_transform == 0 || input_transform == 1, "beam search: input_transform 0 (logits) or 1 (log10(softmax + 1e-8)), got %d",
             input_transform);
  ST_REQUIRE(logits->channels >= 2 && logits->channels <= kMaxClasses, "beam search: 2..%d classes supported, got %d",
             kMaxClasses, logits->channels);
  ST_REQUIRE(beam_width >= 1 && beam_width <= kMaxBeam, "beam search: beam width 1..%d supported, got %d", kMaxBeam,
             beam_width);
  ST_REQUIRE(max_out >= 1, "beam search: max_out must be positive");
  size_t need = st_ctc_beam_ws(logits->batch, logits->frames, beam_width);
  if (!workspace || workspace_bytes < need) {
    st::set_error("beam search: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
    return ST_EWORKSPACE;
  }
  if (logits->batch == 0) return ST_OK;
  RowMap map{(long)logits->t_pitch * logits->c_pitch, (long)logits->halo * logits->c_pitch, logits->c_pitch};
  float* lp_rows = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + beam_pool_bytes(logits->batch, logits->frames, beam_width));
  hipLaunchKernelGGL(logsoftmax_rows_kernel, dim3(st::ceil_div(logits->frames, 4), logits->batch), dim3(256), 0, st::as_stream(stream),
                     logits->base, map, logits->frames, logits->channels, seq_lens, input_transform, lp_rows);
  const int per_lane = st::ceil_div(beam_width * logits->channels, 64);
  st::trace("ctc_beam<%s> beam=%d transform=%d", beam_width > 64 ? "wide" : "wave", beam_width, input_transform);
#define ST_LAUNCH_BEAM(CPL, MAXB)                                                                                    \
  hipLaunchKernelGGL((ctc_beam_kernel<CPL, MAXB>), dim3(logits->batch), dim3(64), 0, st::as_stream(stream), lp_rows,  \
                     logits->frames, logits->channels, seq_lens, beam_width,                                         \
                     reinterpret_cast<int2*>(workspace), (long)logits->frames * beam_width + 1, ids, max_out,        \
                     out_lens, log_prob)
  if (beam_width > 64) ST_LAUNCH_BEAM(64, 128);
  else if (per_lane <= 4) ST_LAUNCH_BEAM(4, 64);
  else if (per_lane <= 8) ST_LAUNCH_BEAM(8, 64);
  else if (per_lane <= 16) ST_LAUNCH_BEAM(16, 64);
  else ST_LAUNCH_BEAM(32, 64);
#undef ST_LAUNCH_BEAM
  return st::check_launch("ctc_beam_search");
}

}  // extern "C"
