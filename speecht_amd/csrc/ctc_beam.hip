// LM-free CTC prefix beam search (top path) on gfx950 -- SURVEY 8(f) item 3, BASELINE config 5.
//
// The reference reaches a beam search only through its KenLM TensorFlow fork (speech_model.py:101-111);
// this kernel follows the stock tf.nn.ctc_beam_search_decoder recursion without a scorer, with the
// candidate order of oracle/w2l_oracle.py::ctc_beam_search_decode (total desc, slot*C + c asc).
//
// Mapping: ONE wavefront per utterance (the recursion is sequential in time, utterances are the
// parallel axis).  Everything that lives across frames -- the beam entries -- sits in LDS, double
// buffered; a frame is: log-softmax of the 29 logits (wave reduce), parent matching (W^2 hash compares
// spread over the lanes), one sortable 32-bit score per candidate (W*C of them, lane-strided, in registers),
// and a wave-parallel selection of the W best: the W-th largest lane maximum bounds the W-th largest
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

// CPL = candidates per lane (beam_width * C <= 64 * CPL)
template <int CPL>
__global__ __launch_bounds__(64) void ctc_beam_kernel(const float* __restrict__ logits, RowMap map, int T, int C,
                                                      const int* __restrict__ seq_lens, int W,
                                                      int2* __restrict__ node_pool, long pool_stride,
                                                      int* __restrict__ ids, int max_out,
                                                      int* __restrict__ out_lens, float* __restrict__ out_logp) {
  __shared__ BeamSet sets[2];
  __shared__ float lp_s[kMaxClasses];
  __shared__ float2 base_s[kMaxBeam];           // {total, p_blank} of the previous frame
  __shared__ float stay_pb[kMaxBeam], stay_pl[kMaxBeam], stay_total[kMaxBeam];
  __shared__ int parent_of[kMaxBeam];
  __shared__ unsigned dead[kMaxBeam];
  __shared__ int sel_k[kMaxBeam];
  __shared__ unsigned long long surv[64];         // compacted survivor keys of the fast selection
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

  float x_next = (lane < C && Tb > 0) ? logits[map.off(b, 0) + lane] : -INFINITY;
  for (int t = 0; t < Tb; ++t) {
    const BeamSet& S = sets[cur];
    BeamSet& N = sets[cur ^ 1];
    const float x = x_next;
    if (t + 1 < Tb && lane < C) x_next = logits[map.off(b, t + 1) + lane];     // hide the HBM latency of frame t+1
    // (1) log-softmax of this frame
    {
      float m = wave_max_f32(x);
      float z = wave_sum_f32(lane < C ? expf(x - m) : 0.f);
      if (lane < C) lp_s[lane] = x - m - logf(z);
      parent_of[lane] = -1;
      dead[lane] = 0u;
    }
    __syncthreads();
    // (2) which entries have their parent prefix in the beam?  (e, p) pairs spread over the lanes
    {
      int lg = 32 - __clz(max(nb - 1, 1));
      if (nb <= 1) lg = 0;
      const int wp = 1 << lg;
      for (int idx = lane; idx < (nb << lg); idx += 64) {
        int e = idx >> lg, p = idx & (wp - 1);
        if (p < nb && S.len[p] + 1 == S.len[e] && S.hash[p] == S.parent_hash[e]) {
          parent_of[e] = p;
          atomicOr(&dead[p], 1u << S.last[e]);     // child (p, last[e]) already exists: merged into e's stay
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
      stay_total[lane] = lse(npb, npl);
      base_s[lane] = make_float2(tot, S.pb[lane]);
    }
    __syncthreads();
    // (3) one sortable 32-bit score per candidate, in registers (candidate index k = lane + 64 j is implicit)
    unsigned ord[CPL];
    unsigned best_ord = 0u;                               // 0 is below every real candidate
    int best_j = 0;
#pragma unroll
    for (int j = CPL - 1; j >= 0; --j) {                  // descending j + ">=": the lowest j wins a tie
      const int slot = cslot[j], c = ccls[j];
      unsigned oj = 0u;
      if (slot < nb) {
        float v;
        if (c == blank) {
          v = stay_total[slot];
        } else {
          float2 base = base_s[slot];
          v = ((dead[slot] >> c) & 1u) ? -INFINITY : ((S.last[slot] == c) ? base.y : base.x) + lp_s[c];
        }
        oj = order_bits(v);
      }
      ord[j] = oj;
      if (oj >= best_ord) { best_ord = oj; best_j = j; }
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
      int ahead = 0;                                      // lanes whose maximum precedes mine (ties: lower lane)
#pragma unroll 16
      for (int i = 0; i < 64; ++i) {
        const unsigned o = (unsigned)__builtin_amdgcn_readlane((int)best_ord, i);
        ahead += (o > best_ord) || (o == best_ord && i < lane);
      }
      const int live_lanes = __builtin_popcountll(__ballot(best_ord > kOrdNegInf));
      unsigned thr = kOrdNegInf + 1u;                     // fewer than W live lanes: every live candidate survives
      if (live_lanes >= W) thr = (unsigned)__builtin_amdgcn_readlane((int)best_ord, __builtin_ctzll(__ballot(ahead == W - 1)));
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
        __syncthreads();
        if (lane < base) {
          const unsigned long long mine = surv[lane];
          int rank = 0;
          for (int t2 = 0; t2 < base; ++t2) rank += surv[t2] > mine;
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

size_t st_ctc_beam_ws(int batch, int frames, int beam_width) {
  if (batch <= 0 || frames < 0 || beam_width <= 0) return 0;
  return (size_t)batch * ((size_t)frames * beam_width + 1) * sizeof(int2);
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
  const int per_lane = st::ceil_div(beam_width * logits->channels, 64);
#define ST_LAUNCH_BEAM(CPL)                                                                                          \
  hipLaunchKernelGGL(ctc_beam_kernel<CPL>, dim3(logits->batch), dim3(64), 0, st::as_stream(stream), logits->base,    \
                     map, logits->frames, logits->channels, seq_lens, beam_width,                                    \
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
