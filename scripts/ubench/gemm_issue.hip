// Micro-benchmark: what one stage of the 256 x 256 bf16 GEMM tile costs a CU, instruction class by instruction class.
// One workgroup of 4 waves per CU (one wave per SIMD, 128 x 128 wave tile: 256 accumulator registers), a stage =
// 32 deep: 32 x v_mfma_f32_32x32x16_bf16 per wave.  The stage is built up in steps so that the price of each
// ingredient is visible next to the 1024 cycles the MFMAs alone need:
//   mode 0  MFMAs only
//   mode 1  + 16 ds_read_b128 fragment reads per wave (operands really come from LDS)
//   mode 2  + 8 LDS-DMA pieces per wave (global_load_lds_dwordx4, 1 KiB each), counted vmcnt + one barrier per stage
//   mode 3  like 2, but staging through registers: 8 global_load_dwordx4 + 8 ds_write_b128 per wave
//   mode 4  like 2 with HALF the DMA pieces (what a tile of twice the arithmetic intensity would need)
// `src_mb` = size of the global region the stages walk through: small (L2-resident) or large (HBM / MALL streaming).
// Build: hipcc --offload-arch=gfx950 -O3 gemm_issue.hip -o gemm_issue ; run: ./gemm_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int BT = 256, BK = 32, ST = 4, PL = BT * BK;     // one operand's stage: 256 rows x 32 bf16 = 16 KiB
constexpr int MT = 4, NT = 4;

template <int MODE>
__global__ __launch_bounds__(256) void stage_loop(const unsigned short* __restrict__ src, size_t src_elems, int stages, float* out,
                                                   int row_stride, int lds_random) {
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * ST * PL];
  unsigned short* const As = smem;
  unsigned short* const Bs = smem + ST * PL;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;
  // fragment addresses as in the product kernel: row * BK + ((2 ks + h) ^ ((row / 4) % 4)) * 8
  int a_frag[2], b_frag[2];
  {
    const int ra = wm * 128 + l31, rb = wn * 128 + l31;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      a_frag[ks] = ra * BK + (((2 * ks + h) ^ ((ra / 4) % 4)) * 8);
      b_frag[ks] = rb * BK + (((2 * ks + h) ^ ((rb / 4) % 4)) * 8);
    }
  }
  // this workgroup's window of the source: a stage = 2 operands x 16 KiB; piece = 1 KiB (16 rows x 64 B)
  constexpr int PIECES = MODE == 4 ? 4 : 8;               // per wave and stage
  const size_t wg_base = ((size_t)blockIdx.x * 4099 * 1024) & (src_elems - 1);
  // a piece = 16 rows of 64 bytes (4 lanes each); row_stride = elements between rows (32: the piece is 1 KiB contiguous;
  // 256: an activation tensor with 256 bf16 channels; 8192: transposed filters of a 8192-deep reduction)
  const unsigned short* p0 = src + (size_t)(lane >> 2) * row_stride + (lane & 3) * 8;
  const size_t piece_step = (size_t)16 * row_stride;
  bf16x8 fa[2][MT], fb[2][NT];
  // initialise LDS so that the MFMAs chew on finite numbers
  for (int i = tid; i < 2 * ST * PL; i += 256)
    smem[i] = lds_random ? (unsigned short)(0x3800 + ((i * 2654435761u) >> 21)) : (unsigned short)(0x3c00 + (i & 255));
  __syncthreads();
  auto reads = [&](int slot, int ks) {
    const unsigned short* as = As + slot * PL;
    const unsigned short* bs = Bs + slot * PL;
#pragma unroll
    for (int n = 0; n < NT; ++n) fb[ks][n] = *reinterpret_cast<const bf16x8*>(bs + b_frag[ks] + n * 32 * BK);
#pragma unroll
    for (int i = 0; i < MT; ++i) fa[ks][i] = *reinterpret_cast<const bf16x8*>(as + a_frag[ks] + i * 32 * BK);
  };
  auto mfmas = [&](int ks, int i0, int i1) {
#pragma unroll
    for (int i = i0; i < i1; ++i)
#pragma unroll
      for (int n = 0; n < NT; ++n)
        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][n], fa[ks][i], acc[i][n], 0, 0, 0);
  };
  if (MODE == 0) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int e = 0; e < 8; ++e) { fb[ks][n][e] = (__bf16)(1.f + 0.01f * (lane + n)); fa[ks][n][e] = (__bf16)(0.5f + 0.02f * (lane + e)); }
    }
  } else {
    reads(0, 0);
  }
  size_t off = wg_base;
  const size_t mask = src_elems - 1;                      // power of two
  for (int kt0 = 0; kt0 < stages; kt0 += ST)
#pragma unroll
  for (int cur = 0; cur < ST; ++cur) {                    // ring slots are compile-time constants after unrolling
    const int fill = (cur + ST - 1) % ST, nxt = (cur + 1) % ST;
    // ---- k-step 0 with the reads of k-step 1 and the staging of the stage ST - 1 ahead
    if (MODE >= 1) reads(cur, 1);
    f32x4 staged[8];
    if (MODE == 2 || MODE == 4) {
#pragma unroll
      for (int pc = 0; pc < PIECES; ++pc) {
        const unsigned short* g = p0 + ((off + (size_t)(wave * 8 + pc) * piece_step) & mask);
        unsigned short* dst = (pc < 4 ? As : Bs) + fill * PL + (wave * 4 + (pc & 3)) * 512;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)dst, 16, 0, 0);
      }
    }
    if (MODE == 3) {
#pragma unroll
      for (int pc = 0; pc < 8; ++pc)
        staged[pc] = *reinterpret_cast<const f32x4*>(p0 + ((off + (size_t)(wave * 8 + pc) * piece_step) & mask));
    }
    off += row_stride == 32 ? 32 * 512 : 32;            // strided rows: the next stage is the next 64 bytes of the same rows
    mfmas(0, 0, MT);
    if (MODE >= 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
    if (MODE >= 2) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- k-step 1, first half (mode 3: the staged registers go to LDS under these MFMAs)
    if (MODE == 3) {
#pragma unroll
      for (int pc = 0; pc < 8; ++pc)
        *reinterpret_cast<f32x4*>((pc < 4 ? As : Bs) + fill * PL + (wave * 4 + (pc & 3)) * 512 + lane * 8) = staged[pc];
    }
    mfmas(1, 0, MT / 2);
    if (MODE == 3) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 2 || MODE == 4) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((ST - 2) * PIECES) : "memory");
    if (MODE == 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- k-step 1, second half, with the first fragments of the next stage
    if (MODE >= 1) reads(nxt, 0);
    mfmas(1, MT / 2, MT);
    if (MODE >= 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int n = 0; n < NT; ++n) s += acc[i][n][0] + acc[i][n][7];
  if (s == 12345.678f) out[0] = s;
}

template <int MODE>
void run(const char* what, const unsigned short* src, size_t src_elems, int stages, float* out, int row_stride = 32, int lds_random = 0) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(stage_loop<MODE>, dim3(256), dim3(256), 0, 0, src, src_elems, stages, out, row_stride, lds_random);
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(stage_loop<MODE>, dim3(256), dim3(256), 0, 0, src, src_elems, stages, out, row_stride, lds_random);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  const double tf = 256.0 * 4 * stages * 32.0 * 32768.0 / (best * 1e-3) / 1e12;
  printf("mode %d  %-62s src %5.0f MB stride %5d %s  %7.3f ms %7.1f TF/s %5.0f cyc/stage@2.4GHz\n", MODE, what,
         src_elems * 2.0 / 1048576.0, row_stride, lds_random ? "rnd" : "   ", best, tf, best * 1e-3 * 2.4e9 / stages);
}

__global__ void fill_random(unsigned short* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = (unsigned short)(0x3800 + (((unsigned)i * 2654435761u) >> 21));      // bf16 in [0.5, 2): random mantissas
}

int main() {
  const int stages = 4000;
  const size_t big = (size_t)2048 << 20;               // 2 GiB of bf16: HBM / MALL streaming
  unsigned short* src;
  float* out;
  (void)hipMalloc(&src, big);
  (void)hipMalloc(&out, 64);
  (void)hipMemset(src, 0x3c, big);
  const size_t small_elems = ((size_t)2 << 20) / 2;    // 2 MiB: resident in every XCD's L2
  const size_t mid_elems = ((size_t)64 << 20) / 2;     // 64 MiB: beyond the L2s, inside the 256 MiB Infinity Cache
  const size_t big_elems = big / 2;
  run<0>("MFMAs only", src, small_elems, stages, out);
  run<1>("+ 16 ds_read_b128 per wave and stage", src, small_elems, stages, out);
  run<1>("+ 16 ds_read_b128 per wave and stage", src, small_elems, stages, out, 32, 1);
  run<2>("+ 8 LDS-DMA pieces per wave and stage, barrier", src, small_elems, stages, out);
  run<2>("+ 8 LDS-DMA pieces per wave and stage, barrier", src, mid_elems, stages, out);
  run<2>("+ 8 LDS-DMA pieces per wave and stage, barrier", src, big_elems, stages, out);
  run<2>("  rows 512 B apart (activation tensor)", src, small_elems, stages, out, 256);
  run<2>("  rows 16 KiB apart (transposed filters)", src, small_elems, stages, out, 8192);
  run<2>("  rows 16 KiB apart (transposed filters)", src, mid_elems, stages, out, 8192);
  hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, src, big_elems);
  (void)hipDeviceSynchronize();
  run<2>("  random data", src, small_elems, stages, out, 32, 1);
  run<2>("  random data, rows 512 B apart", src, small_elems, stages, out, 256, 1);
  run<2>("  random data", src, mid_elems, stages, out, 32, 1);
  run<3>("register staging: 8 global_load_dwordx4 + 8 ds_write_b128", src, small_elems, stages, out);
  run<4>("4 LDS-DMA pieces per wave and stage (half the bytes per MFMA)", src, small_elems, stages, out);
  run<4>("4 LDS-DMA pieces per wave and stage (half the bytes per MFMA)", src, small_elems, stages, out, 32, 1);
  run<4>("4 LDS-DMA pieces per wave and stage (half the bytes per MFMA)", src, big_elems, stages, out);
  (void)hipFree(src);
  (void)hipFree(out);
  return 0;
}
