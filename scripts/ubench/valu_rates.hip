// Issue cost of the VALU instructions the CTC recursion is made of, for ONE wave alone on its SIMD (the situation of
// ctc_alpha_beta_kernel): cycles per instruction over a long run of (a) independent and (b) dependent instances.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP, bool DEP>
__global__ void k(float* out, long long* cyc, int iters) {
  float a0 = threadIdx.x * 0.5f + 1.f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int e = -3 + (int)(threadIdx.x & 1);
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) {        // v_add_f32
      if (DEP) { REP64(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(a1));) }
      else { REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(1.0f));) }
    } else if (OP == 1) { // v_ldexp_f32
      if (DEP) { REP64(asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a0) : "v"(e));) }
      else { REP8(asm volatile("v_ldexp_f32 %0, %0, %8\n v_ldexp_f32 %1, %1, %8\n v_ldexp_f32 %2, %2, %8\n v_ldexp_f32 %3, %3, %8\n v_ldexp_f32 %4, %4, %8\n v_ldexp_f32 %5, %5, %8\n v_ldexp_f32 %6, %6, %8\n v_ldexp_f32 %7, %7, %8"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(e));) }
    } else if (OP == 2) { // v_frexp_mant_f32
      if (DEP) { REP64(asm volatile("v_frexp_mant_f32 %0, %0" : "+v"(a0));) }
      else { REP8(asm volatile("v_frexp_mant_f32 %0, %0\n v_frexp_mant_f32 %1, %1\n v_frexp_mant_f32 %2, %2\n v_frexp_mant_f32 %3, %3\n v_frexp_mant_f32 %4, %4\n v_frexp_mant_f32 %5, %5\n v_frexp_mant_f32 %6, %6\n v_frexp_mant_f32 %7, %7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    } else if (OP == 3) { // v_exp_f32
      if (DEP) { REP64(asm volatile("v_exp_f32 %0, %0" : "+v"(a0));) }
      else { REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    } else if (OP == 4) { // v_max3_i32 on the float bits
      if (DEP) { REP64(asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a0) : "v"(a1), "v"(a2));) }
      else { REP8(asm volatile("v_max3_i32 %0, %0, %8, %8\n v_max3_i32 %1, %1, %8, %8\n v_max3_i32 %2, %2, %8, %8\n v_max3_i32 %3, %3, %8, %8\n v_max3_i32 %4, %4, %8, %8\n v_max3_i32 %5, %5, %8, %8\n v_max3_i32 %6, %6, %8, %8\n v_max3_i32 %7, %7, %8, %8"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(e));) }
    } else if (OP == 5) { // v_mov_b32_dpp wave_shr:1
      if (DEP) { REP64(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0));) }
      else { REP8(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 wave_shr:1 row_mask:0xf bank_mask:0xf"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    } else if (OP == 6) { // v_frexp_exp_i32_f32
      int r0;
      if (DEP) { REP64(asm volatile("v_frexp_exp_i32_f32 %0, %1\n" : "=v"(r0) : "v"(a0)); a0 += (float)0; asm volatile("" : "+v"(a0) : "v"(r0));) }
      else { REP8(asm volatile("v_frexp_exp_i32_f32 %0, %0\n v_frexp_exp_i32_f32 %1, %1\n v_frexp_exp_i32_f32 %2, %2\n v_frexp_exp_i32_f32 %3, %3\n v_frexp_exp_i32_f32 %4, %4\n v_frexp_exp_i32_f32 %5, %5\n v_frexp_exp_i32_f32 %6, %6\n v_frexp_exp_i32_f32 %7, %7"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    } else if (OP == 7) { // v_pk_add_f32 (two floats per lane)
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {1.f, 1.f};
      if (DEP) { REP64(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(q));) }
      else { REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));) }
      a0 = p0[0] + p1[0] + p2[0] + p3[0];
    }
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP, bool DEP>
void run(const char* name, float* out, long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<OP, DEP>), dim3(1), dim3(64), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL((k<OP, DEP>), dim3(1), dim3(64), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  // s_memtime / readcyclecounter ticks at 100 MHz on this part: report ns per instruction too
  printf("%-22s %-12s %8.2f ticks/1000 instr\n", name, DEP ? "dependent" : "independent", 1000.0 * c / (64.0 * iters));
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256); hipMalloc(&cyc, 8);
#define BOTH(OP, NAME) run<OP, false>(NAME, out, cyc); run<OP, true>(NAME, out, cyc);
  BOTH(0, "v_add_f32") BOTH(1, "v_ldexp_f32") BOTH(2, "v_frexp_mant_f32") BOTH(6, "v_frexp_exp_i32_f32") BOTH(3, "v_exp_f32")
  BOTH(4, "v_max3_i32") BOTH(5, "v_mov_b32_dpp shr") BOTH(7, "v_pk_add_f32")
  return 0;
}
