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
// kernel sums (deterministic, no atomics).
#include <algorithm>

#include "st_common.h"

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
  int tiles_m, tiles_n, chunk;       // XCD-aware tile order
};

// ------------------------------------------------------------------------------------
// C[M, n_store] = epilogue(A[M, Kp] * B[Kp, Np]).  256 threads = WMW x WNW waves, each wave
// owns a (BM/WMW) x (BN/WNW) block of 32x32 MFMA tiles.  Register-staged double-buffered LDS,
// one barrier per 32-deep k-tile; the global loads of tile k+1 are in flight under the 16
// MFMA k-steps of tile k.
// ------------------------------------------------------------------------------------
template <int BM, int BN, int WMW, int WNW, int EPI>
__global__ __launch_bounds__(NTHREADS) void gemm_nn_kernel(NNParams p) {
  constexpr int WTM = BM / WMW, WTN = BN / WNW;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int AL = BM * BK / 4 / NTHREADS;
  constexpr int BL = (BK * BN / 4 + NTHREADS - 1) / NTHREADS;
  constexpr int BROW4 = BN / 4;  // float4 per B row
  static_assert(WMW * WNW == 4 && MT >= 1 && NT >= 1 && AL >= 1, "tile config");

  __shared__ __attribute__((aligned(16))) float As[2][BM * APITCH];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * BN];
  __shared__ long a_off[BM];
  __shared__ long c_off[BM];
  __shared__ long m_off[BM];

  // XCD-aware order: block b runs on XCD b%8; give each XCD one contiguous chunk of the
  // panel-major tile list so the CUs sharing an L2 stream the same filter panel together.
  const int bid = blockIdx.x;
  const int idx = (bid & 7) * p.chunk + (bid >> 3);
  if ((bid >> 3) >= p.chunk || idx >= p.tiles_m * p.tiles_n) return;
  const int tile_n = idx / p.tiles_m;
  const int tile_m = idx - tile_n * p.tiles_m;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
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

  const float* arow[AL];
#pragma unroll
  for (int i = 0; i < AL; ++i) arow[i] = p.A + a_off[(tid >> 3) + 32 * i] + (tid & 7) * 4;
  const float* bptr[BL];
  bool bact[BL];
#pragma unroll
  for (int i = 0; i < BL; ++i) {
    int f = tid + NTHREADS * i;
    bact[i] = f < BK * BROW4;
    int k = bact[i] ? f / BROW4 : 0, nq = f % BROW4;
    bptr[i] = p.Bm + (long)k * p.Np + n0 + nq * 4;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[AL], rb[BL];
  const int nk = p.Kp / BK;
  const int kq4 = (tid & 7) * 4;

  auto gload = [&](int kt) {
    const int k0 = kt * BK;
    const bool kin = k0 + kq4 < p.Kvalid;   // Kvalid is a multiple of 16: whole float4 in or out
#pragma unroll
    for (int i = 0; i < AL; ++i)
      ra[i] = kin ? *reinterpret_cast<const f32x4*>(arow[i] + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < BL; ++i)
      if (bact[i]) rb[i] = *reinterpret_cast<const f32x4*>(bptr[i] + (long)k0 * p.Np);
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AL; ++i)
      *reinterpret_cast<f32x4*>(&As[buf][((tid >> 3) + 32 * i) * APITCH + kq4]) = ra[i];
#pragma unroll
    for (int i = 0; i < BL; ++i) {
      int f = tid + NTHREADS * i;
      if (bact[i]) *reinterpret_cast<f32x4*>(&Bs[buf][f * 4]) = rb[i];
    }
  };

  gload(0);
  sstore(0);
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) gload(kt + 1);
    const float* as = &As[cur][(wm * WTM + l31) * APITCH + 4 * h];
    const float* bs = &Bs[cur][(4 * h) * BN + wn * WTN + l31];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 a[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i)
        a[i] = *reinterpret_cast<const f32x4*>(as + i * 32 * APITCH + 8 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float b[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) b[n] = bs[(8 * q + j) * BN + n * 32];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], b[n], acc[i][n], 0, 0, 0);
      }
    }
    if (more) sstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = n0 + wn * WTN + n * 32 + l31;
    const bool col_ok = col < p.n_store;
    float bv = 0.f;
    if (EPI == 0 && p.bias && col_ok) bv = p.bias[col];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const long co = c_off[row];
        if (co >= 0 && col_ok) {
          float v = acc[i][n][r];
          if (EPI == 0) {
            v += bv;
            if (p.relu) v = fmaxf(v, 0.f);
          } else if (p.mask) {
            v = p.mask[m_off[row] + col] > 0.f ? v : 0.f;
          }
          p.C[co + col] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// Filter gradient: out[split][k][n] = sum over the split's rows m of A[m][k] * Z[m][n].
// Both operands are reduction-major in memory (k resp. n contiguous within a row), which is
// exactly what the MFMA operand fetch wants: lane i reads element i of LDS row m -- bank
// conflict free with no padding.  Tile 128(k) x BN(n), 32 rows of m per stage.
// ------------------------------------------------------------------------------------
struct TNParams {
  const float* A; RowMap amap;
  const float* Z; RowMap zmap;
  float* out;            // [splits][Kp][Np]
  int M, Kvalid, Kp, Np, z_cols;   // z_cols = readable floats per Z row (its c_pitch)
  int rows_per_split;    // multiple of 32
  int tiles_k, tiles_n;
};

template <int BN, int WKW, int WNW>
__global__ __launch_bounds__(NTHREADS) void gemm_tn_kernel(TNParams p) {
  constexpr int BKO = 128;
  constexpr int BMR = 32;
  constexpr int WTK = BKO / WKW, WTN = BN / WNW;   // wave tile
  constexpr int MT = WTK / 32, NT = WTN / 32;
  constexpr int ZL = BMR * BN / 4 / NTHREADS;
  constexpr int ZROW4 = BN / 4;
  static_assert(WKW * WNW == 4 && MT >= 1 && NT >= 1 && ZL >= 1, "tile config");

  __shared__ __attribute__((aligned(16))) float As[2][BMR * BKO];
  __shared__ __attribute__((aligned(16))) float Zs[2][BMR * BN];

  const int tile = blockIdx.x;
  const int tile_n = tile % p.tiles_n, tile_k = tile / p.tiles_n;
  const int k0 = tile_k * BKO, n0 = tile_n * BN;
  const int split = blockIdx.y;
  const int m_begin = split * p.rows_per_split;
  const int m_end = min(p.M, m_begin + p.rows_per_split);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wk = wave / WNW, wn = wave % WNW;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // thread -> (row, float4 column) of the A stage [32][128] and the Z stage [32][BN]
  const int ar = tid >> 5, ac4 = (tid & 31) * 4;       // rows ar + 8*i, i < 4
  const bool a_col_ok = k0 + ac4 < p.Kvalid;
  f32x4 ra[4], rz[ZL];

  auto gload = [&](int mb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m = mb + ar + 8 * i;
      ra[i] = (m < m_end && a_col_ok)
                  ? *reinterpret_cast<const f32x4*>(p.A + p.amap.off(m) + k0 + ac4)
                  : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < ZL; ++i) {
      int f = tid + NTHREADS * i;
      int r = f / ZROW4, c4 = (f % ZROW4) * 4;
      int m = mb + r;
      rz[i] = (m < m_end && n0 + c4 < p.z_cols)
                  ? *reinterpret_cast<const f32x4*>(p.Z + p.zmap.off(m) + n0 + c4)
                  : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<f32x4*>(&As[buf][(ar + 8 * i) * BKO + ac4]) = ra[i];
#pragma unroll
    for (int i = 0; i < ZL; ++i) *reinterpret_cast<f32x4*>(&Zs[buf][(tid + NTHREADS * i) * 4]) = rz[i];
  };

  if (m_begin < m_end) {
    gload(m_begin);
    sstore(0);
  }
  __syncthreads();
  int cur = 0;
  for (int mb = m_begin; mb < m_end; mb += BMR) {
    const bool more = mb + BMR < m_end;
    if (more) gload(mb + BMR);
    const float* as = &As[cur][h * BKO + wk * WTK + l31];
    const float* zs = &Zs[cur][h * BN + wn * WTN + l31];
#pragma unroll
    for (int s = 0; s < BMR / 2; ++s) {
      float a[MT], z[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = as[(2 * s) * BKO + i * 32];
#pragma unroll
      for (int n = 0; n < NT; ++n) z[n] = zs[(2 * s) * BN + n * 32];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], z[n], acc[i][n], 0, 0, 0);
    }
    if (more) sstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  float* out = p.out + (long)split * p.Kp * p.Np;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = n0 + wn * WTN + n * 32 + l31;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = k0 + wk * WTK + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (k < p.Kp && col < p.Np) out[(long)k * p.Np + col] = (k < p.Kvalid && col < p.z_cols) ? acc[i][n][r] : 0.f;
      }
  }
}

// dst[i] = sum_s slabs[s][i]
__global__ void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dst,
                                   long n4, int splits) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    f32x4 s = reinterpret_cast<const f32x4*>(slabs)[i];
    for (int k = 1; k < splits; ++k) s += reinterpret_cast<const f32x4*>(slabs)[i + k * n4];
    reinterpret_cast<f32x4*>(dst)[i] = s;
  }
}

// column sums of dz: partial[chunk][c] over row chunks, then summed in order by a second pass.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ Z, RowMap zmap,
                                                             int M, int cols, int rows_per_chunk,
                                                             float* __restrict__ partial, int np) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int sub = threadIdx.x >> 6;     // 4 row phases
  const int m_begin = blockIdx.y * rows_per_chunk;
  const int m_end = min(M, m_begin + rows_per_chunk);
  float s = 0.f;
  if (c < cols)
    for (int m = m_begin + sub; m < m_end; m += 4) s += Z[zmap.off(m) + c];
  __shared__ float red[4][64];
  red[sub][threadIdx.x & 63] = s;
  __syncthreads();
  if (sub == 0 && c < np) {
    float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    partial[(long)blockIdx.y * np + c] = c < cols ? t : 0.f;
  }
}
__global__ void colsum_final_kernel(const float* __restrict__ partial, int chunks, int np,
                                    float* __restrict__ dbias) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= np) return;
  float s = 0.f;
  for (int k = 0; k < chunks; ++k) s += partial[(long)k * np + c];
  dbias[c] = s;
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

// out[(w' * cop + o) * NpT + c] = in[((W-1-w') * cip + c) * Np + o]   (32x32 LDS transpose)
__global__ __launch_bounds__(256) void flip_transpose_kernel(const float* __restrict__ in, int W, int cin,
                                                             int cout, int cip, int Np, int cop, int NpT,
                                                             float* __restrict__ out) {
  __shared__ float tile[32][33];
  const int w = blockIdx.z;
  const int c0 = blockIdx.y * 32, o0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    int c = c0 + r, o = o0 + tx;
    tile[r][tx] = (c < cin && o < cout) ? in[((long)(W - 1 - w) * cip + c) * Np + o] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    int o = o0 + r, c = c0 + tx;
    if (o < cout && c < cin) out[((long)w * cop + o) * NpT + c] = tile[tx][r];
  }
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

template <int BM, int BN, int WMW, int WNW>
void launch_nn(NNParams& p, int epi, hipStream_t s) {
  p.tiles_m = st::ceil_div(p.M, BM);
  p.tiles_n = p.Np / BN;
  const int total = p.tiles_m * p.tiles_n;
  p.chunk = st::ceil_div(total, 8);
  dim3 grid(p.chunk * 8), block(NTHREADS);
  if (epi == 0) hipLaunchKernelGGL((gemm_nn_kernel<BM, BN, WMW, WNW, 0>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((gemm_nn_kernel<BM, BN, WMW, WNW, 1>), grid, block, 0, s, p);
}

int run_nn(NNParams& p, int epi, hipStream_t s) {
  if (p.Np % 128 == 0) {
    long tiles128 = (long)st::ceil_div(p.M, 128) * (p.Np / 128);
    if (tiles128 >= 512) launch_nn<128, 128, 2, 2>(p, epi, s);
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

bool tensor_ok(const st_tensor3* t) {
  return t && t->base && t->batch > 0 && t->frames > 0 && t->channels > 0 && t->halo >= 0 &&
         t->c_pitch % 16 == 0 && t->c_pitch >= t->channels && t->t_pitch >= t->halo + t->frames;
}

}  // namespace

extern "C" {

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
  if (int e = st_fill_f32(packed_t, 0.f, (size_t)kpt * npt, stream)) return e;
  dim3 grid(st::ceil_div(cout, 32), st::ceil_div(cin, 32), width);
  hipLaunchKernelGGL(flip_transpose_kernel, grid, dim3(256), 0, st::as_stream(stream), packed, width,
                     cin, cout, cin_pitch, np, cout_pitch, npt, packed_t);
  return st::check_launch("flip_transpose");
}

int st_conv1d_nwc_fwd_f32(const st_tensor3* x, const float* packed, const float* bias, int width,
                          int stride, int pad_left, int relu, const st_tensor3* y, void* stream) {
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
  return run_nn(p, 0, st::as_stream(stream));
}

int st_conv1d_nwc_bwd_data_f32(const st_tensor3* dz, const float* packed_t, int width, int pad_left,
                               const st_tensor3* act, const st_tensor3* dx, void* stream) {
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
  return run_nn(p, 1, st::as_stream(stream));
}

static int bwd_filter_splits(int M, int kp, int np) {
  int tiles = st::ceil_div(kp, 128) * (np / (np % 128 == 0 ? 128 : np));
  int want = st::ceil_div(1024, tiles);                  // ~4 blocks per CU
  int max_splits = std::max(1, M / 256);                 // at least 256 rows per split
  return std::max(1, std::min(want, max_splits));
}

size_t st_conv1d_bwd_filter_ws(const st_tensor3* x, const st_tensor3* dz, int width) {
  if (!x || !dz) return 0;
  int kp = (int)st::round_up((size_t)width * x->c_pitch, BK), np = npad_of(dz->channels);
  int M = dz->batch * dz->frames;
  int splits = bwd_filter_splits(M, kp, np);
  size_t slabs = splits > 1 ? (size_t)splits * kp * np * sizeof(float) : 0;
  size_t colsum = (size_t)st::ceil_div(M, 256) * np * sizeof(float);
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
  if (p.Np % 128 == 0) {
    p.tiles_n = p.Np / 128;
    hipLaunchKernelGGL((gemm_tn_kernel<128, 2, 2>), dim3(p.tiles_k * p.tiles_n, used), dim3(NTHREADS), 0, s, p);
  } else if (p.Np == 64) {
    p.tiles_n = 1;
    hipLaunchKernelGGL((gemm_tn_kernel<64, 2, 2>), dim3(p.tiles_k, used), dim3(NTHREADS), 0, s, p);
  } else if (p.Np == 32) {
    p.tiles_n = 1;
    hipLaunchKernelGGL((gemm_tn_kernel<32, 4, 1>), dim3(p.tiles_k, used), dim3(NTHREADS), 0, s, p);
  } else {
    st::set_error("conv bwd_filter: unsupported n_pad=%d", p.Np);
    return ST_EINVAL;
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
    const int chunks = st::ceil_div(p.M, 256);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(st::ceil_div(p.Np, 64), chunks), dim3(256), 0, s, dz->base,
                       p.zmap, p.M, dz->channels, 256, partial, p.Np);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(st::ceil_div(p.Np, 256)), dim3(256), 0, s, partial, chunks,
                       p.Np, dbias);
    if (int e = st::check_launch("colsum")) return e;
  }
  return ST_OK;
}

}  // extern "C"
