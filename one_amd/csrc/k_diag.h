// k_diag.h - bench.py's calibration kernels: k_diag_read, k_diag_lines, k_diag_long, k_diag_l2
// (included by kernels.hip inside namespace redgpu { namespace { ... } }; see its file map).
#pragma once

// Calibration for bench.py ("the box's measured streaming-read ceiling from a
// calibration kernel run in the same session"): reads `bytes` once with 16-byte loads, 8 in
// flight per lane, and folds them into one word per wave so the loads cannot be dropped.
__global__ void __launch_bounds__(512)
k_diag_read(const uint4 *__restrict__ p, uint64_t n16, uint32_t *sink) {
  // one 512-thread workgroup per CU, 8 non-temporal 16-byte loads in flight per lane: the
  // fastest streaming read of the shapes tried on MI355X (scripts/lab/hbm_probe.hip: 6.8 TB/s;
  // 256 threads x 8 workgroups per CU, this kernel's round-1 shape, 5.0-5.3)
  typedef uint32_t v4 __attribute__((ext_vector_type(4)));
  const v4 *q = reinterpret_cast<const v4 *>(p);
  const uint64_t step = uint64_t(gridDim.x) * 512;
  uint64_t i = uint64_t(blockIdx.x) * 512 + threadIdx.x;
  v4 acc = {0, 0, 0, 0};
  for (; i + 7 * step < n16; i += 8 * step) {
    v4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(q + i + k * step);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc ^= v[k];
  }
  for (; i < n16; i += step) acc ^= q[i];
  uint32_t a = acc.x ^ acc.y ^ acc.z ^ acc.w;
  for (int o = 32; o; o >>= 1) a ^= __shfl_xor(a, o);
  if ((threadIdx.x & 63) == 0 && a == 0x9e3779b9u) atomicAdd(sink, 1u);
}

// The memory side of the streaming walk over 64-byte lines with nothing else: every lane requests
// its line as k_stream does (4 x 16 bytes back to back, 2 lines per lane) and stores an
// Outcome-shaped record per line (int32 + 2 x uint64, non-temporal) - 64 B read + 20 B written
// per line.  What HBM gives this mix is the roof of configs[1]'s shape (bench.py reports it).
__global__ void __launch_bounds__(512)
k_diag_lines(const uint8_t *__restrict__ data, uint64_t nLines, int32_t *res, uint64_t *st,
             uint64_t *en, uint32_t *sink) {
  typedef uint32_t v4 __attribute__((ext_vector_type(4)));
  const uint64_t tiles = nLines / 1024;
  v4 acc = {0, 0, 0, 0};
  for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    v4 v[2][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const uint64_t ln = t * 1024 + uint64_t(c) * 512 + threadIdx.x;
        v[c][k] = reinterpret_cast<const v4 *>(data + ln * 64)[k];
      }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const uint64_t ln = t * 1024 + uint64_t(c) * 512 + threadIdx.x;
      const v4 x = v[c][0] ^ v[c][1] ^ v[c][2] ^ v[c][3];
      acc ^= x;
      __builtin_nontemporal_store(int32_t(x.x), res + ln);
      __builtin_nontemporal_store(uint64_t(x.y), st + ln);
      __builtin_nontemporal_store(uint64_t(x.z), en + ln);
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) atomicAdd(sink, 1u);
}

// ... and over LONG lines (a multiple of 128 bytes): a lane requests one whole cache line of each
// of its two lines at a time, as k_stream's 128-byte form does - the 64 lanes of a wave touch 64
// cache lines that lie a line length apart.  No stores (one Outcome per line is noise here).
// MI355X gives this pattern 5.3 TB/s at 4 KiB lines, 3.0 at 16 KiB, 1.7 at 64 KiB, where a
// coalesced read of the same bytes gets 6.4 (scripts/lab/hbm_probe.hip).
__global__ void __launch_bounds__(512)
k_diag_long(const uint8_t *__restrict__ data, uint64_t nLines, uint32_t lineBytes, uint32_t *sink) {
  typedef uint32_t v4 __attribute__((ext_vector_type(4)));
  const uint64_t tiles = nLines / 1024;
  const uint32_t R = lineBytes / 128;
  v4 acc = {0, 0, 0, 0};
  for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    for (uint32_t r = 0; r < R; ++r) {
      v4 v[2][8];
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const uint64_t ln = t * 1024 + uint64_t(c) * 512 + threadIdx.x;
          v[c][k] = reinterpret_cast<const v4 *>(data + ln * lineBytes + uint64_t(r) * 128)[k];
        }
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= v[c][k];
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) atomicAdd(sink, 1u);
}

// The L2 request roof of a one-lookup-per-byte walk whose table lives in L2 (REDGPU_TAB_GLOBAL_*:
// SYN-4K's 2 MiB class table), measured instead of quoted: every lane runs CH chains of DEPENDENT
// 2-byte gathers over a table of 2^20 uint16 - the next index is made of the value just read, as
// the next state is - with no input side at all.  2048 lanes per CU, the occupancy k_generic
// runs the real walk at.
template <int CH>
__global__ void __launch_bounds__(256)
k_diag_l2(const uint16_t *__restrict__ tab, uint32_t rounds, uint32_t *sink) {
  uint32_t idx[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c)
    idx[c] = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + uint32_t(c) * 40503u;
  for (uint32_t r = 0; r < rounds; ++r) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const uint32_t v = tab[idx[c] & 0xfffffu];
      idx[c] = idx[c] * 33u + v + r;
    }
  }
  uint32_t a = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) a ^= idx[c];
  if (a == 0x9e3779b9u) atomicAdd(sink, 1u);
}
