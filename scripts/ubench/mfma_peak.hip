// Micro-benchmark: sustained v_mfma_f32_32x32x16_bf16 rate on the whole chip (what "peak" means under the
// board's power limit), 1 or 2 waves per SIMD, 8 independent accumulators per wave, nothing but MFMAs.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// same stream, but the operands are random bf16 values that change from MFMA to MFMA (data toggling costs
// power; a constant-operand stream is the optimistic bound)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void mfma_loop_random(float* out, const bf16x8* __restrict__ rnd, int iters) {
  f32x16 acc[8];
  for (int a = 0; a < 8; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  bf16x8 x[6], y[6];
  for (int k = 0; k < 6; ++k) { x[k] = rnd[(threadIdx.x * 12 + k) & 4095]; y[k] = rnd[(threadIdx.x * 12 + 6 + k) & 4095]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 6; ++rep)
#pragma unroll
      for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[(rep + a) % 6], y[(rep * 5 + a) % 6], acc[a], 0, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < 8; ++a) s += acc[a][0];
  if (s == 12345.f) out[0] = s;
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void mfma_loop(float* out, int iters) {
  f32x16 acc[8];
  for (int a = 0; a < 8; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  bf16x8 x, y;
  for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(1.0f + threadIdx.x * 1e-3f); y[e] = (__bf16)(0.5f + e); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 6; ++rep)
#pragma unroll
      for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < 8; ++a) s += acc[a][0];
  if (s == 12345.f) out[0] = s;
}

template <int THREADS>
void run(const char* name, int blocks, int iters) {
  float* out;
  hipMalloc(&out, 4);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(mfma_loop<THREADS>, dim3(blocks), dim3(THREADS), 0, 0, out, iters);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(mfma_loop<THREADS>, dim3(blocks), dim3(THREADS), 0, 0, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double mfmas = (double)blocks * (THREADS / 64) * iters * 48.0;
    printf("%s: %d blocks x %d threads, %.3f ms, %.1f TFLOP/s, %.1f cycles/MFMA/SIMD at 2.4 GHz\n", name, blocks, THREADS, ms,
           mfmas * 32768.0 / (ms * 1e-3) / 1e12, (ms * 1e-3 * 2.4e9) / (mfmas / (256.0 * 4)));
  }
  hipFree(out);
}

// fp32 matrix instruction of the default path (v_mfma_f32_32x32x2_f32), random operands
template <int THREADS, int NACC = 8>
__global__ __launch_bounds__(THREADS) void mfma_loop_f32(float* out, const float* __restrict__ rnd, int iters) {
  f32x16 acc[8];
  for (int a = 0; a < 8; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float x[6], y[6];
  for (int k = 0; k < 6; ++k) { x[k] = rnd[(threadIdx.x * 12 + k) & 4095]; y[k] = rnd[(threadIdx.x * 12 + 6 + k) & 4095]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 6; ++rep)
#pragma unroll
      for (int a = 0; a < 8; ++a) acc[a % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[(rep + a) % 6], y[(rep * 5 + a) % 6], acc[a % NACC], 0, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < 8; ++a) s += acc[a][0];
  if (s == 12345.f) out[0] = s;
}

template <int THREADS, int NACC>
void run_f32(const char* name, int blocks, int iters) {
  float *out, *rnd;
  hipMalloc(&out, 4);
  hipMalloc(&rnd, 4096 * 4);
  float* h = new float[4096];
  unsigned st = 777u;
  for (int i = 0; i < 4096; ++i) { st = st * 1664525u + 1013904223u; h[i] = ((int)(st >> 8) - (1 << 23)) / (float)(1 << 22); }
  hipMemcpy(rnd, h, 4096 * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((mfma_loop_f32<THREADS, NACC>), dim3(blocks), dim3(THREADS), 0, 0, out, rnd, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double mfmas = (double)blocks * (THREADS / 64) * iters * 48.0;
    if (rep) printf("%s: %.3f ms, %.1f TFLOP/s\n", name, ms, mfmas * 4096.0 / (ms * 1e-3) / 1e12);
  }
}

template <int THREADS>
void run_random(const char* name, int blocks, int iters) {
  float* out;
  bf16x8* rnd;
  hipMalloc(&out, 4);
  hipMalloc(&rnd, 4096 * sizeof(bf16x8));
  unsigned short* h = new unsigned short[4096 * 8];
  unsigned st = 12345u;
  for (int i = 0; i < 4096 * 8; ++i) {               // random sign/mantissa, exponents around 1.0
    st = st * 1664525u + 1013904223u;
    h[i] = (unsigned short)(((st >> 16) & 0x80FF) | ((120 + ((st >> 8) & 7)) << 7));
  }
  hipMemcpy(rnd, h, 4096 * sizeof(bf16x8), hipMemcpyHostToDevice);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(mfma_loop_random<THREADS>, dim3(blocks), dim3(THREADS), 0, 0, out, rnd, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double mfmas = (double)blocks * (THREADS / 64) * iters * 48.0;
    if (rep) printf("%s: %.3f ms, %.1f TFLOP/s, %.1f cycles/MFMA/SIMD at 2.4 GHz\n", name, ms,
                    mfmas * 32768.0 / (ms * 1e-3) / 1e12, (ms * 1e-3 * 2.4e9) / (mfmas / (256.0 * 4)));
  }
}

int main() {
  run_f32<512, 8>("fp32 32x32x2, random operands, 2 waves/SIMD, 8 accumulators", 256, 5000);
  run_f32<256, 8>("fp32 32x32x2, random operands, 1 wave/SIMD, 8 accumulators", 256, 10000);
  run_f32<256, 4>("fp32 32x32x2, random operands, 1 wave/SIMD, 4 accumulators", 256, 10000);
  run_f32<512, 4>("fp32 32x32x2, random operands, 2 waves/SIMD, 4 accumulators", 256, 5000);
  run_f32<256, 2>("fp32 32x32x2, random operands, 1 wave/SIMD, 2 accumulators", 256, 10000);
  run_random<256>("random operands, 1 wave/SIMD ", 256, 20000);
  run_random<512>("random operands, 2 waves/SIMD", 256, 10000);
  run_random<256>("random operands, 1 wave/SIMD, long", 256, 200000);
  run<256>("1 wave/SIMD ", 256, 20000);
  run<512>("2 waves/SIMD", 256, 10000);
  run<256>("1 wave/SIMD, half the CUs", 128, 20000);
  return 0;
}
