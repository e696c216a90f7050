// What ds_read_b64_tr_b16 (gfx950 LDS transpose read) returns: LDS holds lds[i] = i (16-bit); every lane passes the byte address
// addr(lane) of pattern P and prints its four 16-bit results.  Patterns: 0: addr = 8 * lane (lane's own four elements);
// 1: addr = 2 * (64 * (lane & 15) + 4 * (lane >> 4))  (16 rows of 64 elements, lane group g reads columns 4g..4g+3).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned short* out, int pattern) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int lane = threadIdx.x;
  unsigned addr = pattern == 0 ? 8u * lane : 2u * (64u * (lane & 15) + 4u * (lane >> 4));
  addr += (unsigned)(size_t)lds;
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main() {
  unsigned short* d;
  hipMalloc(&d, 64 * 4 * 2);
  unsigned short h[256];
  for (int pattern = 0; pattern < 2; ++pattern) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, pattern);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pattern %d\n", pattern);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
  }
  return 0;
}
