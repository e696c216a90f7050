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

constexpr int kMaxBeam = 64;
constexpr int kMaxClasses = 32;
constexpr unsigned long long kRootHash = 0x243F6A8885A308D3ull;

struct RowMap {   // (b, t) -> float offset into a padded NWC tensor
  long batch_stride;
  long row0;
  int row_stride;
  __device__ __forceinline__ long off(int b, int t) const { return (long)b * batch_stride + row0 + (long)t * row_stride; }
};

struct BeamSet {   // structure of arrays: lane r reads/writes entry r without bank conflicts
  unsigned long long hash[kMaxBeam];
  unsigned long long parent_hash[kMaxBeam];
  int len[kMaxBeam];
  int last[kMaxBeam];
  int node[kMaxBeam];
  float pb[kMaxBeam];
  float pl[kMaxBeam];
  float total[kMaxBeam];
};

__device__ __forceinline__ unsigned long long child_hash(unsigned long long h, int c) {
  unsigned long long x = h + 0x9E3779B97F4A7C15ull * (unsigned long long)(c + 1);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__device__ __forceinline__ float lse(float a, float b) {
  float hi = fmaxf(a, b), lo = fminf(a, b);
  return hi == -INFINITY ? -INFINITY : hi + log1pf(expf(lo - hi));
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
__global__ __launch_bounds__(256) void logsoftmax_rows_kernel(const float* __restrict__ logits, RowMap map, int T, int C,
                                                              const int* __restrict__ seq_lens, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, b = blockIdx.y;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= min(seq_lens[b], T)) return;
  const float x = lane < C ? logits[map.off(b, t) + lane] : -INFINITY;
  const float m = wave_max_f32(x);
  const float z = wave_sum_f32(lane < C ? expf(x - m) : 0.f);
  if (lane < kMaxClasses) out[((long)b * T + t) * kMaxClasses + lane] = lane < C ? x - m - logf(z) : 0.f;
}

// CPL = candidates per lane (beam_width * C <= 64 * CPL)
template <int CPL>
__global__ __launch_bounds__(64) void ctc_beam_kernel(const float* __restrict__ lp_all, int T, int C,
                                                      const int* __restrict__ seq_lens, int W,
                                                      int2* __restrict__ node_pool, long pool_stride,
                                                      int* __restrict__ ids, int max_out,
                                                      int* __restrict__ out_lens, float* __restrict__ out_logp) {
  __shared__ BeamSet sets[2];
  __shared__ float lp_s[kMaxClasses];
  // per beam entry, for the scoring pass: {total, p_blank of the previous frame, total of the stay candidate, last label}
  __shared__ __attribute__((aligned(16))) float4 slot_s[kMaxBeam];
  __shared__ float stay_pb[kMaxBeam], stay_pl[kMaxBeam];
  __shared__ int parent_of[kMaxBeam];
  __shared__ unsigned dead[kMaxBeam];
  __shared__ int sel_k[kMaxBeam];
  __shared__ unsigned long long surv[64 + 4];     // compacted survivor keys of the fast selection (+ zero pad)
  __shared__ float sel_v[kMaxBeam];

  const int b = blockIdx.x, lane = threadIdx.x;
  const int Tb = min(seq_lens[b], T);
  const int blank = C - 1;
  int2* nodes = node_pool + (long)b * pool_stride;

  // (slot, class) of this lane's candidates k = lane + 64 j: fixed for the whole utterance
  int cslot[CPL], ccls[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    int k = lane + 64 * j;
    cslot[j] = k / C;
    ccls[j] = k - cslot[j] * C;
  }

  int cur = 0, nb = 1;
  double offset = 0.0;                   // log-probability of the current best entry (wave-uniform)
  if (lane == 0) {
    BeamSet& s = sets[0];
    s.hash[0] = kRootHash; s.parent_hash[0] = 0; s.len[0] = 0; s.last[0] = -1; s.node[0] = 0;
    s.pb[0] = 0.f; s.pl[0] = -INFINITY; s.total[0] = 0.f;
  }
  __syncthreads();

  // log-softmax rows of this utterance, written by logsoftmax_rows_kernel: [t][32]
  const float* __restrict__ lp_rows = lp_all + (long)b * T * kMaxClasses;
  float lp_next = (lane < kMaxClasses && Tb > 0) ? lp_rows[lane] : 0.f;
  for (int t = 0; t < Tb; ++t) {
    const BeamSet& S = sets[cur];
    BeamSet& N = sets[cur ^ 1];
    // (1) this frame's log-probabilities into LDS; the next frame's row is already on its way
    if (lane < kMaxClasses) lp_s[lane] = lp_next;
    if (t + 1 < Tb && lane < kMaxClasses) lp_next = lp_rows[(long)(t + 1) * kMaxClasses + lane];
    parent_of[lane] = -1;
    dead[lane] = 0u;
    __syncthreads();
    // (2) which entries have their parent prefix in the beam?  (e, p) pairs spread over the lanes; the LDS reads of all
    // passes are issued before the first compare (a pass at a time each one waits a full LDS round trip)
    {
      int lg = 32 - __clz(max(nb - 1, 1));
      if (nb <= 1) lg = 0;
      const int wp = 1 << lg, pairs = nb << lg;
      constexpr int PASSES = 4;                      // 64 * 4 = 16 x 16 pairs; wider beams loop
      for (int idx0 = 0; idx0 < pairs; idx0 += 64 * PASSES) {
        int e_[PASSES], p_[PASSES], len_e[PASSES], len_p[PASSES], last_e[PASSES];
        unsigned long long hp[PASSES], he[PASSES];
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
          const int idx = idx0 + lane + 64 * i;
          const bool in = idx < pairs && (idx & (wp - 1)) < nb;
          e_[i] = in ? idx >> lg : 0;
          p_[i] = in ? idx & (wp - 1) : 0;
          len_e[i] = in ? S.len[e_[i]] : -7;
          len_p[i] = S.len[p_[i]];
          hp[i] = S.hash[p_[i]];
          he[i] = S.parent_hash[e_[i]];
          last_e[i] = S.last[e_[i]];
        }
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
          if (len_p[i] + 1 == len_e[i] && hp[i] == he[i]) {
            parent_of[e_[i]] = p_[i];
            atomicOr(&dead[p_[i]], 1u << last_e[i]);     // child (p, last[e]) already exists: merged into e's stay
          }
        }
      }
    }
    __syncthreads();
    // (2b) the stay candidate of entry `lane`
    if (lane < nb) {
      float tot = S.total[lane];
      float npb = tot + lp_s[blank];
      float npl = -INFINITY;
      if (S.len[lane] > 0) {
        float mass = S.pl[lane];
        int p = parent_of[lane];
        if (p >= 0) mass = lse(mass, (S.len[p] > 0 && S.last[p] == S.last[lane]) ? S.pb[p] : S.total[p]);
        npl = mass + lp_s[S.last[lane]];
      }
      stay_pb[lane] = npb;
      stay_pl[lane] = npl;
      slot_s[lane] = make_float4(tot, S.pb[lane], lse(npb, npl), __int_as_float(S.last[lane]));
    }
    __syncthreads();
    // (3) one sortable 32-bit score per candidate, in registers (candidate index k = lane + 64 j is implicit)
    unsigned ord[CPL];
    unsigned best_ord = 0u;                               // 0 is below every real candidate
    int best_j = 0;
    {
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
    }
    // (4) selection of the W best candidates, best first.
    // Fast path (threshold + rank): the W-th largest LANE maximum is a lower bound of the W-th largest
    // candidate, so only candidates at or above it ("survivors", a few dozen at most in practice) can be
    // selected; they are compacted into LDS in candidate order and every survivor finds its rank by counting the
    // survivors ahead of it (keys are unique: score, then smaller candidate index).  All of it is wave-parallel;
    // the sequential rounds below remain as the fallback for more than 64 survivors (flat posteriors, wide beams).
    int n_new = 0;
    bool selected = false;
    if (CPL <= 16) {
      const int live_lanes = __builtin_popcountll(__ballot(best_ord > kOrdNegInf));
      unsigned thr = kOrdNegInf + 1u;                     // fewer than W live lanes: every live candidate survives
      if (live_lanes >= W) {
        // the W-th largest lane maximum = the largest v with at least W lane maxima >= v: a radix select, one ballot and
        // a population count per bit (ranking the 64 lanes against each other took 3 800 of the 13 300 cycles of a frame)
        unsigned prefix = 0u;
#pragma unroll
        for (int bit = 31; bit >= 0; --bit) {
          const unsigned trial = prefix | (1u << bit);
          if (__builtin_popcountll(__ballot(best_ord >= trial)) >= W) prefix = trial;
        }
        thr = prefix;
      }
      int base = 0;
#pragma unroll
      for (int j = 0; j < CPL; ++j) {                     // compaction in candidate order k = lane + 64 j
        const bool keep = ord[j] >= thr && ord[j] > kOrdNegInf;
        const unsigned long long m = __ballot(keep);
        const int pos = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (keep && pos < 64) surv[pos] = ((unsigned long long)ord[j] << 32) | (0xFFFFFFFFu - (unsigned)(lane + 64 * j));
        base += __builtin_popcountll(m);
      }
      if (base <= 64) {
        selected = true;
        n_new = min(base, W);
        if (lane < 4) surv[min(base, 64) + lane] = 0ull;       // pad: the counting loop reads four keys at a time
        __syncthreads();
        if (lane < base) {
          const unsigned long long mine = surv[lane];
          int rank = 0;
          for (int t2 = 0; t2 < base; t2 += 4) {
            const unsigned long long k0 = surv[t2], k1 = surv[t2 + 1], k2 = surv[t2 + 2], k3 = surv[t2 + 3];
            rank += (k0 > mine) + (k1 > mine) + (k2 > mine) + (k3 > mine);
          }
          if (rank < W) {
            sel_k[rank] = (int)(0xFFFFFFFFu - (unsigned)mine);
            sel_v[rank] = order_value((unsigned)(mine >> 32));
          }
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
    __syncthreads();
    // (5) materialise the surviving entries, best first.  Scores are kept RELATIVE to the best entry (its
    // total becomes 0) with the running offset in double: after 1500 frames the absolute log-probabilities
    // are O(-3000), where fp32 resolves only 2e-4 and near-ties at the beam boundary would be decided by
    // rounding; relative scores stay O(10) for the whole utterance.
    const float top = n_new > 0 ? sel_v[0] : 0.f;
    offset += (double)top;
    if (lane < n_new) {
      int k = sel_k[lane];
      int slot = k / C, c = k - slot * C;
      if (c == blank) {
        N.hash[lane] = S.hash[slot]; N.parent_hash[lane] = S.parent_hash[slot];
        N.len[lane] = S.len[slot]; N.last[lane] = S.last[slot]; N.node[lane] = S.node[slot];
        N.pb[lane] = stay_pb[slot] - top; N.pl[lane] = stay_pl[slot] - top;
      } else {
        int id = 1 + t * W + lane;
        nodes[id] = make_int2(S.node[slot], c);
        N.hash[lane] = child_hash(S.hash[slot], c); N.parent_hash[lane] = S.hash[slot];
        N.len[lane] = S.len[slot] + 1; N.last[lane] = c; N.node[lane] = id;
        N.pb[lane] = -INFINITY; N.pl[lane] = sel_v[lane] - top;
      }
      N.total[lane] = sel_v[lane] - top;
    }
    nb = n_new;
    cur ^= 1;
    __syncthreads();
  }

  // top path: entry 0 of the final set; walk the node chain backwards
  if (lane == 0) {
    const BeamSet& S = sets[cur];
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
  ST_REQUIRE(logits && logits->base && seq_lens && ids && out_lens && log_prob, "beam search: null argument");
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
                     logits->base, map, logits->frames, logits->channels, seq_lens, lp_rows);
  const int per_lane = st::ceil_div(beam_width * logits->channels, 64);
#define ST_LAUNCH_BEAM(CPL)                                                                                          \
  hipLaunchKernelGGL(ctc_beam_kernel<CPL>, dim3(logits->batch), dim3(64), 0, st::as_stream(stream), lp_rows,         \
                     logits->frames, logits->channels, seq_lens, beam_width,                                         \
                     reinterpret_cast<int2*>(workspace), (long)logits->frames * beam_width + 1, ids, max_out,        \
                     out_lens, log_prob)
  if (per_lane <= 4) ST_LAUNCH_BEAM(4);
  else if (per_lane <= 8) ST_LAUNCH_BEAM(8);
  else if (per_lane <= 16) ST_LAUNCH_BEAM(16);
  else ST_LAUNCH_BEAM(32);
#undef ST_LAUNCH_BEAM
  return st::check_launch("ctc_beam_search");
}

}  // extern "C"
