// hbm_probe.hip - lab: what does HBM give this chip for the access patterns the streaming walk
// uses?  Streaming reads (coalesced 16 B/lane; 64-byte line per lane as k_stream requests it),
// with and without the 20 B per 64 B line of Outcome stores, plain and non-temporal, at several
// occupancies.  Prints GB/s of bytes read (+ written).  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NT> __device__ __forceinline__ u32x4 ld(const u32x4 *p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

// coalesced: a wave reads 1 KB contiguous per load instruction, INFLIGHT loads back to back
template <int INFLIGHT, int NT>
__global__ void k_read_coal(const u32x4 *p, uint64_t n16, uint32_t *sink) {
  const uint64_t step = uint64_t(gridDim.x) * blockDim.x;
  uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  for (; i + (INFLIGHT - 1) * step < n16; i += INFLIGHT * step) {
    u32x4 v[INFLIGHT];
#pragma unroll
    for (int k = 0; k < INFLIGHT; ++k) v[k] = ld<NT>(p + i + k * step);
#pragma unroll
    for (int k = 0; k < INFLIGHT; ++k) acc ^= v[k];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) atomicAdd(sink, 1);
}

// k_stream's pattern: lane = one 64-byte line (4 x 16 B back to back), LPL lines per lane in
// flight, tiles of blockDim * LPL lines handed out grid-stride; optional Outcome-shaped stores
template <int LPL, int NT_LD, int STORES, int NT_ST>
__global__ void k_lines(const uint8_t *data, uint64_t nLines, int32_t *res, uint64_t *st,
                        uint64_t *en, uint32_t *sink) {
  const uint64_t perTile = uint64_t(blockDim.x) * LPL;
  const uint64_t tiles = nLines / perTile;
  u32x4 acc = {0, 0, 0, 0};
  for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    u32x4 v[LPL][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int c = 0; c < LPL; ++c) {
        const uint64_t ln = t * perTile + uint64_t(c) * blockDim.x + threadIdx.x;
        v[c][k] = ld<NT_LD>(reinterpret_cast<const u32x4 *>(data + ln * 64) + k);
      }
#pragma unroll
    for (int c = 0; c < LPL; ++c) {
      u32x4 x = v[c][0] ^ v[c][1] ^ v[c][2] ^ v[c][3];
      acc ^= x;
      if (STORES) {
        const uint64_t ln = t * perTile + uint64_t(c) * blockDim.x + threadIdx.x;
        if (NT_ST) {
          __builtin_nontemporal_store(int32_t(x.x), res + ln);
          __builtin_nontemporal_store(uint64_t(x.y), st + ln);
          __builtin_nontemporal_store(uint64_t(x.z), en + ln);
        } else {
          res[ln] = int32_t(x.x); st[ln] = x.y; en[ln] = x.z;
        }
      }
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) atomicAdd(sink, 1);
}

// k_stream's pattern on LONG lines: a lane requests 128 bytes (a whole cache line, 8 x 16 B back to
// back) of each of its LPL lines, then the next 128 bytes of the same lines: the 64 lanes of a
// wave touch 64 cache lines that lie LINE bytes apart
template <int LPL>
__global__ void k_long(const uint8_t *data, uint64_t nLines, uint32_t lineBytes, uint32_t *sink) {
  const uint64_t perTile = uint64_t(blockDim.x) * LPL;
  const uint64_t tiles = nLines / perTile;
  const uint32_t R = lineBytes / 128;
  u32x4 acc = {0, 0, 0, 0};
  for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    for (uint32_t r = 0; r < R; ++r) {
      u32x4 v[LPL][8];
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int c = 0; c < LPL; ++c) {
          const uint64_t ln = t * perTile + uint64_t(c) * blockDim.x + threadIdx.x;
          v[c][k] = reinterpret_cast<const u32x4 *>(data + ln * lineBytes + uint64_t(r) * 128)[k];
        }
#pragma unroll
      for (int c = 0; c < LPL; ++c)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= v[c][k];
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) atomicAdd(sink, 1);
}

// the same bytes read line by line by whole waves: a wave reads 1 KB contiguous per instruction
template <int INFLIGHT>
__global__ void k_long_coal(const uint8_t *data, uint64_t bytes, uint32_t *sink) {
  // each workgroup takes 64 KB chunks (16 lines of 4 KiB); a wave reads 1 KB per load
  const uint64_t chunks = bytes / 65536;
  u32x4 acc = {0, 0, 0, 0};
  for (uint64_t ch = blockIdx.x; ch < chunks; ch += gridDim.x) {
    const u32x4 *p = reinterpret_cast<const u32x4 *>(data + ch * 65536);
    for (uint32_t i = threadIdx.x; i < 4096; i += blockDim.x * INFLIGHT) {
      u32x4 v[INFLIGHT];
#pragma unroll
      for (int k = 0; k < INFLIGHT; ++k) v[k] = p[i + k * blockDim.x];
#pragma unroll
      for (int k = 0; k < INFLIGHT; ++k) acc ^= v[k];
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) atomicAdd(sink, 1);
}

template <class F> float timeit(F f, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  const uint64_t bytes = 2ull << 30;
  const uint64_t nLines = bytes / 64;
  uint8_t *d; int32_t *res; uint64_t *st, *en; uint32_t *sink;
  CK(hipMalloc(&d, bytes)); CK(hipMalloc(&res, nLines * 4)); CK(hipMalloc(&st, nLines * 8));
  CK(hipMalloc(&en, nLines * 8)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(d, 0x5a, bytes)); CK(hipMemset(sink, 0, 4));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount;
  printf("CUs %d, buffer %.1f GiB, %llu lines of 64 B\n", cus, bytes / 1073741824.0, (unsigned long long)nLines);
#define COAL(INF, NT, THR, WG) { float ms = timeit([&] { hipLaunchKernelGGL((k_read_coal<INF, NT>), dim3(cus * WG), dim3(THR), 0, 0, reinterpret_cast<const u32x4 *>(d), bytes / 16, sink); }, 5); \
    printf("coalesced read  inflight %2d nt %d  %4d thr x %d WG/CU : %7.1f GB/s\n", INF, NT, THR, WG, bytes / ms / 1e6); }
  COAL(4, 0, 512, 1) COAL(8, 0, 512, 1) COAL(16, 0, 512, 1) COAL(8, 0, 512, 2) COAL(8, 0, 256, 4) COAL(8, 0, 256, 8)
  COAL(4, 0, 1024, 2) COAL(8, 1, 512, 1) COAL(8, 1, 512, 2) COAL(8, 1, 256, 8) COAL(16, 1, 512, 2)
#define LINES(LPL, NTL, STO, NTS, THR, WG) { float ms = timeit([&] { hipLaunchKernelGGL((k_lines<LPL, NTL, STO, NTS>), dim3(cus * WG), dim3(THR), 0, 0, d, nLines, res, st, en, sink); }, 5); \
    printf("line/lane  lines/lane %d nt_ld %d stores %d nt_st %d  %4d thr x %d WG/CU : read %7.1f GB/s  traffic %7.1f GB/s\n", LPL, NTL, STO, NTS, THR, WG, bytes / ms / 1e6, (bytes + (STO ? nLines * 20 : 0)) / ms / 1e6); }
  LINES(2, 0, 0, 0, 512, 1) LINES(2, 0, 0, 0, 512, 2) LINES(4, 0, 0, 0, 512, 1) LINES(4, 0, 0, 0, 512, 2) LINES(2, 1, 0, 0, 512, 2)
  LINES(2, 0, 1, 0, 512, 1) LINES(2, 0, 1, 1, 512, 1) LINES(2, 0, 1, 0, 512, 2) LINES(2, 0, 1, 1, 512, 2)
  LINES(4, 0, 1, 1, 512, 2) LINES(2, 0, 1, 1, 256, 4) LINES(2, 0, 1, 1, 1024, 1) LINES(2, 1, 1, 1, 512, 2)
#define LONG(LPL, LB, THR, WG) { float ms = timeit([&] { hipLaunchKernelGGL((k_long<LPL>), dim3(cus * WG), dim3(THR), 0, 0, d, bytes / LB, LB, sink); }, 5); \
    printf("long lines %5d B, 128 B per lane request, %d lines/lane  %4d thr x %d WG/CU : %7.1f GB/s\n", LB, LPL, THR, WG, bytes / ms / 1e6); }
  LONG(2, 4096, 512, 1) LONG(2, 4096, 512, 2) LONG(4, 4096, 512, 1) LONG(1, 4096, 512, 2) LONG(2, 512, 512, 1) LONG(2, 1024, 512, 1) LONG(2, 16384, 512, 1) LONG(2, 65536, 256, 1)
  { float ms = timeit([&] { hipLaunchKernelGGL((k_long_coal<4>), dim3(cus), dim3(512), 0, 0, d, bytes, sink); }, 5);
    printf("64 KB chunks per workgroup, coalesced 1 KB per wave-load, 4 in flight: %7.1f GB/s\n", bytes / ms / 1e6); }
  { float ms = timeit([&] { hipLaunchKernelGGL((k_long_coal<8>), dim3(cus), dim3(512), 0, 0, d, bytes, sink); }, 5);
    printf("64 KB chunks per workgroup, coalesced 1 KB per wave-load, 8 in flight: %7.1f GB/s\n", bytes / ms / 1e6); }
  return 0;
}
