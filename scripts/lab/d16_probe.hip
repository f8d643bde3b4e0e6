// d16_probe.hip - does ds_read_u8_d16 / _d16_hi keep the other half of its destination VGPR on
// this part (gfx950 runs with SRAM ECC)?  Prints the two words a lane sees.  Lab only.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(uint32_t *out) {
  __shared__ uint8_t t[256];
  t[threadIdx.x] = uint8_t(threadIdx.x + 1);
  __syncthreads();
  uint32_t lo = 0xAAAA5555u, hi = 0xAAAA5555u;
  uint32_t a = threadIdx.x;
  asm volatile("ds_read_u8_d16 %0, %2\n\ts_waitcnt lgkmcnt(0)\n\t"
               "ds_read_u8_d16_hi %1, %2\n\ts_waitcnt lgkmcnt(0)"
               : "+v"(lo), "+v"(hi) : "v"(a) : "memory");
  out[2 * threadIdx.x] = lo;
  out[2 * threadIdx.x + 1] = hi;
}
int main() {
  uint32_t *d, h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("lane 3: d16 -> %08x (kept hi half: %s), d16_hi -> %08x (kept lo half: %s)\n", h[6],
         (h[6] >> 16) == 0xAAAA ? "yes" : "no", h[7], (h[7] & 0xffff) == 0x5555 ? "yes" : "no");
  return 0;
}
