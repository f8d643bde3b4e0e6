// tune.hip - kernel-variant laboratory for the fixed-stride hot path (NOT part of the product).
//
// Builds standalone:  make -C tools   ->  tools/tune
// Runs on the GPU box: tools/tune tests/golden/dfas/syn256.reda [filter]
// For every variant: checks the outputs against a trivially-correct one-lane-per-line kernel,
// then times it over rotating 64 MiB input buffers (> 256 MiB in total) with HIP events.
// Ablation switches (BOOK/LOOKUP/LOAD) produce wrong outputs on purpose and are only timed.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../one_amd/csrc/dfa_image.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

struct Dev {
  const uint8_t *table;   // fused u8 [S][256]
  const int32_t *result;  // [S]
  uint32_t nStates, init, firstAccept, tableBytes;
};

struct Io {
  const uint8_t *data;
  uint64_t n;
  uint32_t lineLen;
  int32_t *res;
  uint64_t *start;
  uint64_t *end;
};

__global__ void k_fill(uint8_t *p, uint64_t nWords, uint64_t seed) {
  uint64_t i = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x;
  uint64_t step = uint64_t(gridDim.x) * blockDim.x;
  for (; i < nWords; i += step) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    reinterpret_cast<uint64_t *>(p)[i] = z ^ (z >> 31);
  }
}

// trivially-correct reference: match<styLast,false> (Matcher.h:413-495), one lane per line
__global__ void k_ref(Dev d, Io io) {
  uint64_t line = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x;
  if (line >= io.n) return;
  const uint8_t *p = io.data + line * io.lineLen;
  uint32_t s = d.init, accS = 0, endv = 0, startv = 0;
  for (uint32_t i = 0; i < io.lineLen; ++i) {
    uint32_t was = s;
    s = d.table[(s << 8) | p[i]];
    if (was == d.init && s != was) startv = i;
    if (s >= d.firstAccept) { accS = s; endv = i + 1; }
  }
  int32_t r = endv ? d.result[accS] : 0;
  io.res[line] = r;
  io.start[line] = r ? startv : 0;
  io.end[line] = r ? endv : 0;
}

__global__ void k_cmp(const int32_t *a, const int32_t *b, const uint64_t *sa, const uint64_t *sb,
                      const uint64_t *ea, const uint64_t *eb, uint64_t n, unsigned *bad) {
  uint64_t i = blockIdx.x * uint64_t(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  if (a[i] != b[i] || sa[i] != sb[i] || ea[i] != eb[i]) atomicAdd(bad, 1u);
}

// ------------------------------------------------------------------------------------------
// Variant V1: direct per-lane 16-byte loads (the round-1 production shape)
//   THREADS per block, CHAINS lines per lane, BOOK 0/1/2, LOOKUP 0/1, LOAD 0/1
// ------------------------------------------------------------------------------------------
template <int BOOK>
struct Chain {
  uint32_t s, accS, endv, startv, wasInit;
};

template <int BOOK, int LOOKUP>
__device__ __forceinline__ void step(Chain<BOOK> &c, const uint8_t *tab, uint32_t byte,
                                     uint32_t idx, uint32_t init, uint32_t firstAccept) {
  uint32_t sNew;
  if (LOOKUP)
    sNew = tab[(c.s << 8) | byte];
  else
    sNew = (c.s * 5 + byte) & 0xff;
  if (BOOK >= 2) {
    const uint32_t isInit = (sNew == init);
    c.startv = (c.wasInit && !isInit) ? idx : c.startv;
    c.wasInit = isInit;
  }
  if (BOOK >= 1) {
    const bool acc = sNew >= firstAccept;
    c.accS = acc ? sNew : c.accS;
    c.endv = acc ? idx + 1 : c.endv;
  }
  c.s = sNew;
}

template <int THREADS>
__device__ __forceinline__ void stageTable(uint8_t *tab, int32_t *ldsRes, const Dev &d) {
  // explicit 16-byte moves; volatile-free but indexed so the compiler keeps dwordx4
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 *src = reinterpret_cast<const u32x4 *>(d.table);
  u32x4 *dst = reinterpret_cast<u32x4 *>(tab);
  const uint32_t n16 = d.tableBytes / 16;
#pragma unroll 4
  for (uint32_t i = threadIdx.x; i < n16; i += THREADS) dst[i] = src[i];
  for (uint32_t i = threadIdx.x; i < d.nStates; i += THREADS) ldsRes[i] = d.result[i];
}

template <int THREADS, int CHAINS, int BOOK, int LOOKUP, int LOAD>
__global__ void __launch_bounds__(THREADS) k_v1(Dev d, Io io) {
  extern __shared__ __align__(16) uint8_t lds[];
  uint8_t *tab = lds;
  int32_t *ldsRes = reinterpret_cast<int32_t *>(lds + d.tableBytes);
  stageTable<THREADS>(tab, ldsRes, d);
  __syncthreads();
  const uint32_t init = d.init, firstAccept = d.firstAccept, lineLen = io.lineLen;
  const uint64_t linesPerTile = uint64_t(THREADS) * CHAINS;
  const uint64_t nTiles = (io.n + linesPerTile - 1) / linesPerTile;
  for (uint64_t tile = blockIdx.x; tile < nTiles; tile += gridDim.x) {
    Chain<BOOK> cs[CHAINS];
    const uint8_t *lp[CHAINS];
    uint64_t line[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      line[c] = tile * linesPerTile + uint64_t(c) * THREADS + threadIdx.x;
      const uint64_t ln = line[c] < io.n ? line[c] : io.n - 1;
      lp[c] = io.data + ln * lineLen;
      cs[c].s = init; cs[c].accS = 0; cs[c].endv = 0; cs[c].startv = 0; cs[c].wasInit = 1;
    }
    uint4 cur[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      if (LOAD) cur[c] = *reinterpret_cast<const uint4 *>(lp[c]);
      else cur[c] = make_uint4(uint32_t(line[c]) * 2654435761u, uint32_t(line[c]) * 40503u, c, 7);
    }
    for (uint32_t off = 0; off < lineLen; off += 16) {
      uint4 nxt[CHAINS];
      const bool more = off + 16 < lineLen;
      if (more) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
          if (LOAD) nxt[c] = *reinterpret_cast<const uint4 *>(lp[c] + off + 16);
          else nxt[c] = make_uint4(cur[c].y + off, cur[c].z ^ off, cur[c].w, cur[c].x);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int c = 0; c < CHAINS; ++c) {
            const uint32_t word = k == 0 ? cur[c].x : k == 1 ? cur[c].y : k == 2 ? cur[c].z : cur[c].w;
            step<BOOK, LOOKUP>(cs[c], tab, (word >> (8 * j)) & 0xffu, off + 4 * k + j, init, firstAccept);
          }
        }
      }
      if (more) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) cur[c] = nxt[c];
      }
    }
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      if (line[c] >= io.n) continue;
      if (BOOK == 0) { io.res[line[c]] = int32_t(cs[c].s); continue; }
      const int32_t r = cs[c].endv ? ldsRes[cs[c].accS] : 0;
      io.res[line[c]] = r;
      io.end[line[c]] = r ? uint64_t(cs[c].endv) : 0;
      if (BOOK >= 2) io.start[line[c]] = r ? uint64_t(cs[c].startv) : 0;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Variant V2: each lane pulls 64 contiguous bytes of its own line with four back-to-back
// 16-byte loads (so the four requests for one 128-B cache line reach the L1 together), next
// round prefetched into a second register set.  lineLen % 64 == 0.
//   PERM=1: LDS address formed by one v_perm_b32
// ------------------------------------------------------------------------------------------
template <int PERM>
__device__ __forceinline__ uint32_t tabAddr(uint32_t s, uint32_t word, int j) {
  if (PERM) {
    // byte0 = word.byte[j], byte1 = s.byte0, bytes 2,3 = 0  (v_perm_b32: src0=s -> bytes 4..7)
    return __builtin_amdgcn_perm(s, word, 0x0c0c0400u | uint32_t(j));
  }
  return (s << 8) | ((word >> (8 * j)) & 0xffu);
}

template <int BOOK, int PERM>
__device__ __forceinline__ void step2(Chain<BOOK> &c, const uint8_t *tab, uint32_t word, int j,
                                      uint32_t idx, uint32_t init, uint32_t firstAccept) {
  const uint32_t sNew = tab[tabAddr<PERM>(c.s, word, j)];
  if (BOOK >= 2) {
    const uint32_t isInit = (sNew == init);
    c.startv = (c.wasInit && !isInit) ? idx : c.startv;
    c.wasInit = isInit;
  }
  if (BOOK >= 1) {
    const bool acc = sNew >= firstAccept;
    c.accS = acc ? sNew : c.accS;
    c.endv = acc ? idx + 1 : c.endv;
  }
  c.s = sNew;
}

struct Buf64 { uint4 q[4]; };

__device__ __forceinline__ void load64(Buf64 &b, const uint8_t *p) {
  const uint4 *s = reinterpret_cast<const uint4 *>(p);
  b.q[0] = s[0]; b.q[1] = s[1]; b.q[2] = s[2]; b.q[3] = s[3];
}

template <int THREADS, int CHAINS, int BOOK, int PERM>
__global__ void __launch_bounds__(THREADS) k_v2(Dev d, Io io) {
  extern __shared__ __align__(16) uint8_t lds[];
  uint8_t *tab = lds;
  int32_t *ldsRes = reinterpret_cast<int32_t *>(lds + d.tableBytes);
  stageTable<THREADS>(tab, ldsRes, d);
  __syncthreads();
  const uint32_t init = d.init, firstAccept = d.firstAccept, lineLen = io.lineLen;
  const uint32_t roundsPerLine = lineLen / 64;
  const uint64_t linesPerTile = uint64_t(THREADS) * CHAINS;
  const uint64_t nTiles = (io.n + linesPerTile - 1) / linesPerTile;
  if (blockIdx.x >= nTiles) return;

  auto linePtr = [&](uint64_t tile, int c) {
    uint64_t ln = tile * linesPerTile + uint64_t(c) * THREADS + threadIdx.x;
    if (ln >= io.n) ln = io.n - 1;
    return io.data + ln * lineLen;
  };

  Buf64 cur[CHAINS], nxt[CHAINS];
  uint64_t tile = blockIdx.x;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) load64(cur[c], linePtr(tile, c));

  for (; tile < nTiles; tile += gridDim.x) {
    Chain<BOOK> cs[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      cs[c].s = init; cs[c].accS = 0; cs[c].endv = 0; cs[c].startv = 0; cs[c].wasInit = 1;
    }
    for (uint32_t r = 0; r < roundsPerLine; ++r) {
      // prefetch: next 64 B of these lines, or the first 64 B of the next tile's lines
      const bool lastRound = (r + 1 == roundsPerLine);
      const uint64_t nt = tile + gridDim.x;
      if (!lastRound) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) load64(nxt[c], linePtr(tile, c) + (r + 1) * 64);
      } else if (nt < nTiles) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) load64(nxt[c], linePtr(nt, c));
      }
      const uint32_t off = r * 64;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
              const uint4 v = cur[c].q[q];
              const uint32_t word = k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w;
              step2<BOOK, PERM>(cs[c], tab, word, j, off + 16 * q + 4 * k + j, init, firstAccept);
            }
          }
        }
      }
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) cur[c] = nxt[c];
    }
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      const uint64_t ln = tile * linesPerTile + uint64_t(c) * THREADS + threadIdx.x;
      if (ln >= io.n) continue;
      if (BOOK == 0) { io.res[ln] = int32_t(cs[c].s); continue; }
      const int32_t rr = cs[c].endv ? ldsRes[cs[c].accS] : 0;
      io.res[ln] = rr;
      io.end[ln] = rr ? uint64_t(cs[c].endv) : 0;
      if (BOOK >= 2) io.start[ln] = rr ? uint64_t(cs[c].startv) : 0;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Microbenchmarks: input-read ceilings of the two access shapes, and the fixed cost of
// staging the 64 KB table.  MODE 0: fully coalesced (lane i reads 16 B at i*16 of a 1 KiB
// piece); MODE 1: each lane reads the 64 contiguous bytes of its own line (4 x 16 B back to
// back); MODE 2: as 1 but one 16-B piece per pass (4 passes over the tile).
// ------------------------------------------------------------------------------------------
template <int THREADS, int MODE, int STAGE>
__global__ void __launch_bounds__(THREADS) k_read(Dev d, Io io) {
  extern __shared__ __align__(16) uint8_t lds[];
  if (STAGE) {
    stageTable<THREADS>(lds, reinterpret_cast<int32_t *>(lds + d.tableBytes), d);
    __syncthreads();
  }
  const uint64_t nTiles = (io.n + THREADS - 1) / THREADS;
  uint32_t acc = 0;
  for (uint64_t tile = blockIdx.x; tile < nTiles; tile += gridDim.x) {
    const uint8_t *base = io.data + tile * THREADS * 64;
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        uint4 v = *reinterpret_cast<const uint4 *>(base + (uint64_t(k) * THREADS + threadIdx.x) * 16);
        acc += v.x ^ v.y ^ v.z ^ v.w;
      }
    } else if (MODE == 1) {
      Buf64 b;
      load64(b, base + uint64_t(threadIdx.x) * 64);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc += b.q[k].x ^ b.q[k].y ^ b.q[k].z ^ b.q[k].w;
    } else {
#pragma unroll 1
      for (int k = 0; k < 4; ++k) {
        uint4 v = *reinterpret_cast<const uint4 *>(base + uint64_t(threadIdx.x) * 64 + k * 16);
        acc += v.x ^ v.y ^ v.z ^ v.w;
        __builtin_amdgcn_s_sleep(8);
      }
    }
  }
  if (STAGE) acc += lds[(acc & 0xffff)];
  io.res[blockIdx.x * THREADS + threadIdx.x] = int32_t(acc);
}

template <int THREADS, int MODE, int STAGE, int BLOCKS_PER_CU>
void launchRead(const Dev &d, const Io &io, int numCUs, hipStream_t s) {
  auto kern = k_read<THREADS, MODE, STAGE>;
  size_t ldsBytes = STAGE ? size_t(d.tableBytes) + size_t(d.nStates) * 4 : 16;
  static bool attrDone = false;
  if (!attrDone) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                           hipFuncAttributeMaxDynamicSharedMemorySize, int(ldsBytes)));
    attrDone = true;
  }
  uint64_t tiles = (io.n + THREADS - 1) / THREADS;
  uint64_t blocks = std::min<uint64_t>(tiles, uint64_t(numCUs) * BLOCKS_PER_CU);
  hipLaunchKernelGGL(kern, dim3(uint32_t(blocks)), dim3(THREADS), ldsBytes, s, d, io);
}

#define RD(T, M, ST, BPC)                                                                  \
  Variant{"read T" #T " mode" #M " stage" #ST " bpc" #BPC, false, false, launchRead<T, M, ST, BPC>}

// ------------------------------------------------------------------------------------------
// LDS gather-rate microbenchmark: CH dependent chains per lane, each step one ds_read_u8 of a
// 64 KB table at (state << 8 | byte) with pseudo-random bytes made in registers (no global
// traffic).  PATTERN 0: random bytes (real conflicts); 1: byte = lane (conflict-free banks).
// Reports through the GB/s column: "bytes" = lookups.
// ------------------------------------------------------------------------------------------
template <int THREADS, int CH, int PATTERN, int WIDTH>
__global__ void __launch_bounds__(THREADS) k_lds(Dev d, Io io) {
  extern __shared__ __align__(16) uint8_t lds[];
  stageTable<THREADS>(lds, reinterpret_cast<int32_t *>(lds + d.tableBytes), d);
  __syncthreads();
  const uint32_t steps = uint32_t(io.n * io.lineLen / (uint64_t(gridDim.x) * THREADS * CH));
  uint32_t s[CH], x[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) { s[c] = (threadIdx.x + c) & 0xff; x[c] = threadIdx.x * 2654435761u + c * 40503u + blockIdx.x; }
  const uint32_t lane4 = (threadIdx.x & 31) * 4;
  for (uint32_t i = 0; i < steps; i += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        uint32_t addr;
        if (PATTERN == 0) addr = __builtin_amdgcn_perm(s[c], x[c], 0x0c0c0400u | uint32_t(j));
        else addr = (s[c] << 8) | lane4;
        if (WIDTH == 1) s[c] = lds[addr];
        else s[c] = reinterpret_cast<const uint32_t *>(lds)[addr >> 2] & 0xff;
      }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = x[c] * 1664525u + 1013904223u;
  }
  uint32_t acc = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) acc += s[c];
  io.res[blockIdx.x * THREADS + threadIdx.x] = int32_t(acc);
}

template <int THREADS, int CH, int PATTERN, int WIDTH, int BLOCKS_PER_CU>
void launchLds(const Dev &d, const Io &io, int numCUs, hipStream_t s) {
  auto kern = k_lds<THREADS, CH, PATTERN, WIDTH>;
  size_t ldsBytes = size_t(d.tableBytes) + size_t(d.nStates) * 4;
  static bool attrDone = false;
  if (!attrDone) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                           hipFuncAttributeMaxDynamicSharedMemorySize, int(ldsBytes)));
    attrDone = true;
  }
  hipLaunchKernelGGL(kern, dim3(uint32_t(numCUs * BLOCKS_PER_CU)), dim3(THREADS), ldsBytes, s, d, io);
}

#define LDSB(T, C, P, W, BPC)                                                              \
  Variant{"lds T" #T " CH" #C " pattern" #P " width" #W " bpc" #BPC, false, false,       \
          launchLds<T, C, P, W, BPC>}

// ------------------------------------------------------------------------------------------
// Variant V3: 2 chains per lane, 64 B per chain per round in registers, explicit ping-pong
// register sets (no copies), first loads issued BEFORE the table is staged, per-round relative
// positions (inline constants 1..64) folded into absolute ones once per round.
// ------------------------------------------------------------------------------------------
template <int BOOK>
struct Chain3 {
  uint32_t s, accS, endv, startv, wasInit;
};

template <int BOOK, int CH>
__device__ __forceinline__ void doRound(const Buf64 (&buf)[CH], Chain3<BOOK> (&cs)[CH],
                                        const uint8_t *tab, uint32_t off, uint32_t init,
                                        uint32_t firstAccept) {
  uint32_t endRel[CH], startRel[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) { endRel[c] = 0; startRel[c] = 0; }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const uint4 v = buf[c].q[q];
          const uint32_t word = k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w;
          const uint32_t rel1 = 16 * q + 4 * k + j + 1;  // 1..64: inline constant
          const uint32_t sNew = tab[__builtin_amdgcn_perm(cs[c].s, word, 0x0c0c0400u | uint32_t(j))];
          if (BOOK >= 2) {
            const uint32_t isInit = (sNew == init);
            startRel[c] = (cs[c].wasInit && !isInit) ? rel1 : startRel[c];
            cs[c].wasInit = isInit;
          }
          if (BOOK >= 1) {
            const bool acc = sNew >= firstAccept;
            cs[c].accS = acc ? sNew : cs[c].accS;
            endRel[c] = acc ? rel1 : endRel[c];
          }
          cs[c].s = sNew;
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    if (BOOK >= 1) cs[c].endv = endRel[c] ? off + endRel[c] : cs[c].endv;
    if (BOOK >= 2) cs[c].startv = startRel[c] ? off + startRel[c] - 1 : cs[c].startv;
  }
}

template <int THREADS, int CH, int BOOK>
__global__ void __launch_bounds__(THREADS) k_v3(Dev d, Io io) {
  extern __shared__ __align__(16) uint8_t lds[];
  uint8_t *tab = lds;
  int32_t *ldsRes = reinterpret_cast<int32_t *>(lds + d.tableBytes);
  const uint32_t init = d.init, firstAccept = d.firstAccept, lineLen = io.lineLen;
  const uint32_t R = lineLen / 64;
  const uint64_t linesPerTile = uint64_t(THREADS) * CH;
  const uint64_t nTiles = (io.n + linesPerTile - 1) / linesPerTile;
  const uint64_t G = gridDim.x;
  if (blockIdx.x >= nTiles) return;
  const uint64_t myTiles = (nTiles - blockIdx.x + G - 1) / G;
  const uint64_t Q = myTiles * R;

  const uint8_t *lp[CH];   // line pointers of the tile currently being LOADED
  uint64_t ldTile = blockIdx.x;
  uint32_t ldR = 0;
  auto setPtrs = [&](uint64_t tile) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      uint64_t ln = tile * linesPerTile + uint64_t(c) * THREADS + threadIdx.x;
      if (ln >= io.n) ln = io.n - 1;
      lp[c] = io.data + ln * lineLen;
    }
  };
  auto issue = [&](Buf64 (&b)[CH]) {
#pragma unroll
    for (int c = 0; c < CH; ++c) load64(b[c], lp[c] + ldR * 64);
    if (++ldR == R) { ldR = 0; ldTile += G; setPtrs(ldTile < nTiles ? ldTile : blockIdx.x); }
  };

  Buf64 A[CH], B[CH];
  setPtrs(ldTile);
  issue(A);                       // first HBM round trip overlaps the table staging
  stageTable<THREADS>(tab, ldsRes, d);
  __syncthreads();

  Chain3<BOOK> cs[CH];
  uint64_t tile = blockIdx.x;
  uint32_t r = 0;
  auto begin = [&]() {
    if (r == 0) {
#pragma unroll
      for (int c = 0; c < CH; ++c) { cs[c].s = init; cs[c].accS = 0; cs[c].endv = 0; cs[c].startv = 0; cs[c].wasInit = 1; }
    }
  };
  auto finish = [&]() {
    if (++r == R) {
      r = 0;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const uint64_t ln = tile * linesPerTile + uint64_t(c) * THREADS + threadIdx.x;
        if (ln < io.n) {
          if (BOOK == 0) { io.res[ln] = int32_t(cs[c].s); continue; }
          const int32_t rr = cs[c].endv ? ldsRes[cs[c].accS] : 0;
          io.res[ln] = rr;
          io.end[ln] = rr ? uint64_t(cs[c].endv) : 0;
          if (BOOK >= 2) io.start[ln] = rr ? uint64_t(cs[c].startv) : 0;
        }
      }
      tile += G;
    }
  };

  for (uint64_t q = 0; q < Q;) {
    if (q + 1 < Q) issue(B);
    begin();
    doRound<BOOK, CH>(A, cs, tab, r * 64, init, firstAccept);
    finish();
    if (++q >= Q) break;
    if (q + 1 < Q) issue(A);
    begin();
    doRound<BOOK, CH>(B, cs, tab, r * 64, init, firstAccept);
    finish();
    ++q;
  }
}

// ------------------------------------------------------------------------------------------
// Variant V4: V3 + (a) loads issued in USE order (q-major across chains) so a wave can start on
// its first 16 bytes while the rest streams in, (b) the table addressed through a constant LDS
// pointer (no base add), (c) a scheduling fence per byte step so the bookkeeping of one step
// is not deferred (the compiler otherwise parks 64 lane masks in VGPR lanes).
// ------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(3))) uint8_t lds_u8_t;

template <int BOOK, int CH, int FENCE>
__device__ __forceinline__ void doRound4(const Buf64 (&buf)[CH], Chain3<BOOK> (&cs)[CH],
                                         uint32_t off, uint32_t init, uint32_t firstAccept) {
  lds_u8_t *tab3 = reinterpret_cast<lds_u8_t *>(uintptr_t(0));
  uint32_t endRel[CH], startRel[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) { endRel[c] = 0; startRel[c] = 0; }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const uint4 v = buf[c].q[q];
          const uint32_t word = k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w;
          const uint32_t rel1 = 16 * q + 4 * k + j + 1;
          const uint32_t sNew = tab3[__builtin_amdgcn_perm(cs[c].s, word, 0x0c0c0400u | uint32_t(j))];
          if (BOOK >= 2) {
            const uint32_t isInit = (sNew == init);
            startRel[c] = (cs[c].wasInit && !isInit) ? rel1 : startRel[c];
            cs[c].wasInit = isInit;
            if (FENCE) asm volatile("" : "+v"(startRel[c]));  // pin: evaluate the select NOW
          }
          if (BOOK >= 1) {
            const bool acc = sNew >= firstAccept;
            cs[c].accS = acc ? sNew : cs[c].accS;
            endRel[c] = acc ? rel1 : endRel[c];
            if (FENCE) asm volatile("" : "+v"(endRel[c]), "+v"(cs[c].accS));
          }
          cs[c].s = sNew;
        }
        if (FENCE == 1) __builtin_amdgcn_sched_barrier(0);
      }
      if (FENCE == 2) __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    if (BOOK >= 1) cs[c].endv = endRel[c] ? off + endRel[c] : cs[c].endv;
    if (BOOK >= 2) cs[c].startv = startRel[c] ? off + startRel[c] - 1 : cs[c].startv;
  }
}

template <int THREADS, int CH, int BOOK, int FENCE>
__global__ void __launch_bounds__(THREADS) k_v4(Dev d, Io io) {
  extern __shared__ __align__(16) uint8_t lds[];
  uint8_t *tab = lds;
  int32_t *ldsRes = reinterpret_cast<int32_t *>(lds + d.tableBytes);
  const uint32_t init = d.init, firstAccept = d.firstAccept, lineLen = io.lineLen;
  const uint32_t R = lineLen / 64;
  const uint64_t linesPerTile = uint64_t(THREADS) * CH;
  const uint64_t nTiles = (io.n + linesPerTile - 1) / linesPerTile;
  const uint64_t G = gridDim.x;
  if (blockIdx.x >= nTiles) return;
  const uint64_t myTiles = (nTiles - blockIdx.x + G - 1) / G;
  const uint64_t Q = myTiles * R;

  const uint8_t *lp[CH];
  uint64_t ldTile = blockIdx.x;
  uint32_t ldR = 0;
  auto setPtrs = [&](uint64_t tile) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      uint64_t ln = tile * linesPerTile + uint64_t(c) * THREADS + threadIdx.x;
      if (ln >= io.n) ln = io.n - 1;
      lp[c] = io.data + ln * lineLen;
    }
  };
  auto issue = [&](Buf64 (&b)[CH]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int c = 0; c < CH; ++c)
        b[c].q[q] = reinterpret_cast<const uint4 *>(lp[c] + ldR * 64)[q];
    }
    if (++ldR == R) { ldR = 0; ldTile += G; setPtrs(ldTile < nTiles ? ldTile : blockIdx.x); }
  };

  Buf64 A[CH], B[CH];
  setPtrs(ldTile);
  issue(A);
  stageTable<THREADS>(tab, ldsRes, d);
  __syncthreads();

  Chain3<BOOK> cs[CH];
  uint64_t tile = blockIdx.x;
  uint32_t r = 0;
  auto begin = [&]() {
    if (r == 0) {
#pragma unroll
      for (int c = 0; c < CH; ++c) { cs[c].s = init; cs[c].accS = 0; cs[c].endv = 0; cs[c].startv = 0; cs[c].wasInit = 1; }
    }
  };
  auto finish = [&]() {
    if (++r == R) {
      r = 0;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const uint64_t ln = tile * linesPerTile + uint64_t(c) * THREADS + threadIdx.x;
        if (ln < io.n) {
          if (BOOK == 0) { io.res[ln] = int32_t(cs[c].s); continue; }
          const int32_t rr = cs[c].endv ? ldsRes[cs[c].accS] : 0;
          io.res[ln] = rr;
          io.end[ln] = rr ? uint64_t(cs[c].endv) : 0;
          if (BOOK >= 2) io.start[ln] = rr ? uint64_t(cs[c].startv) : 0;
        }
      }
      tile += G;
    }
  };

  for (uint64_t q = 0; q < Q;) {
    if (q + 1 < Q) issue(B);
    begin();
    doRound4<BOOK, CH, FENCE>(A, cs, r * 64, init, firstAccept);
    finish();
    if (++q >= Q) break;
    if (q + 1 < Q) issue(A);
    begin();
    doRound4<BOOK, CH, FENCE>(B, cs, r * 64, init, firstAccept);
    finish();
    ++q;
  }
}

// ------------------------------------------------------------------------------------------
// Variant V5: V4 with the byte step software-pipelined in the SOURCE: the lookups of step i
// for all chains are issued first (critical path: state -> perm -> ds_read), the bookkeeping of
// step i-1 runs while they are in flight.  Static LDS (table at offset 0, results behind it).
// ------------------------------------------------------------------------------------------
constexpr uint32_t kTabMax = 65536;

template <int BOOK, int CH>
struct Book5 {
  uint32_t accS[CH], endRel[CH], startRel[CH];
  bool wasInit[CH];
};

template <int BOOK, int CH>
__device__ __forceinline__ void bookStep(Book5<BOOK, CH> &b, const uint32_t (&s)[CH], uint32_t rel1,
                                         uint32_t init, uint32_t firstAccept) {
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    if (BOOK >= 2) {
      const bool isInit = (s[c] == init);
      b.startRel[c] = (b.wasInit[c] && !isInit) ? rel1 : b.startRel[c];
      b.wasInit[c] = isInit;
      asm volatile("" : "+v"(b.startRel[c]));
    }
    if (BOOK >= 1) {
      const bool acc = s[c] >= firstAccept;
      b.accS[c] = acc ? s[c] : b.accS[c];
      b.endRel[c] = acc ? rel1 : b.endRel[c];
      asm volatile("" : "+v"(b.endRel[c]), "+v"(b.accS[c]));
    }
  }
}

template <int BOOK, int CH>
__device__ __forceinline__ void doRound5(const Buf64 (&buf)[CH], Chain3<BOOK> (&cs)[CH],
                                         const uint8_t *tab, uint32_t off, uint32_t init,
                                         uint32_t firstAccept) {
  Book5<BOOK, CH> b;
  uint32_t s[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    b.accS[c] = cs[c].accS; b.endRel[c] = 0; b.startRel[c] = 0; b.wasInit[c] = cs[c].wasInit != 0;
    s[c] = cs[c].s;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = 16 * q + 4 * k + j;  // step index 0..63 within the round
        uint32_t t[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const uint4 v = buf[c].q[q];
          const uint32_t word = k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w;
          t[c] = tab[__builtin_amdgcn_perm(s[c], word, 0x0c0c0400u | uint32_t(j))];
        }
        // bookkeeping of the state reached by step i-1 (rel position i), under the LDS latency
        if (i > 0) bookStep<BOOK, CH>(b, s, uint32_t(i), init, firstAccept);
#pragma unroll
        for (int c = 0; c < CH; ++c) s[c] = t[c];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  bookStep<BOOK, CH>(b, s, 64u, init, firstAccept);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    cs[c].s = s[c];
    if (BOOK >= 1) { cs[c].accS = b.accS[c]; cs[c].endv = b.endRel[c] ? off + b.endRel[c] : cs[c].endv; }
    if (BOOK >= 2) { cs[c].startv = b.startRel[c] ? off + b.startRel[c] - 1 : cs[c].startv; cs[c].wasInit = b.wasInit[c]; }
  }
}

__device__ unsigned long long g_stamps[4096 * 16];

template <int THREADS>
struct StageRegs { uint4 v[kTabMax / 16 / THREADS]; };

template <int THREADS>
__device__ __forceinline__ void stageLoad(StageRegs<THREADS> &r, const Dev &d) {
  const uint4 *src = reinterpret_cast<const uint4 *>(d.table);
  const uint32_t n16 = d.tableBytes / 16;
#pragma unroll
  for (uint32_t k = 0; k < kTabMax / 16 / THREADS; ++k) {
    const uint32_t i = k * THREADS + threadIdx.x;
    r.v[k] = i < n16 ? src[i] : make_uint4(0, 0, 0, 0);
  }
}

template <int THREADS>
__device__ __forceinline__ void stageStore(const StageRegs<THREADS> &r, uint8_t *tab, int32_t *ldsRes,
                                           const Dev &d) {
  uint4 *dst = reinterpret_cast<uint4 *>(tab);
#pragma unroll
  for (uint32_t k = 0; k < kTabMax / 16 / THREADS; ++k) dst[k * THREADS + threadIdx.x] = r.v[k];
  for (uint32_t i = threadIdx.x; i < d.nStates; i += THREADS) ldsRes[i] = d.result[i];
}

template <int THREADS, int CH, int BOOK, int TIMING = 0, int ORDER = 0>
__global__ void __launch_bounds__(THREADS) k_v5(Dev d, Io io) {
  __shared__ __align__(16) uint8_t lds[kTabMax + 1024];
  int stampIdx = 0;
  auto stamp = [&]() {
    if (TIMING) {
      unsigned long long t = wall_clock64();
      if ((threadIdx.x & 63) == 0 && stampIdx < 16)
        g_stamps[(blockIdx.x * (THREADS / 64) + threadIdx.x / 64) * 16 + stampIdx] = t;
      ++stampIdx;
    }
  };
  stamp();
  uint8_t *tab = lds;
  int32_t *ldsRes = reinterpret_cast<int32_t *>(lds + kTabMax);
  const uint32_t init = d.init, firstAccept = d.firstAccept, lineLen = io.lineLen;
  const uint32_t R = lineLen / 64;
  const uint64_t linesPerTile = uint64_t(THREADS) * CH;
  const uint64_t nTiles = (io.n + linesPerTile - 1) / linesPerTile;
  const uint64_t G = gridDim.x;
  if (blockIdx.x >= nTiles) return;
  const uint64_t myTiles = (nTiles - blockIdx.x + G - 1) / G;
  const uint64_t Q = myTiles * R;

  const uint8_t *lp[CH];
  uint64_t ldTile = blockIdx.x;
  uint32_t ldR = 0;
  auto setPtrs = [&](uint64_t tile) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      uint64_t ln = tile * linesPerTile + uint64_t(c) * THREADS + threadIdx.x;
      if (ln >= io.n) ln = io.n - 1;
      lp[c] = io.data + ln * lineLen;
    }
  };
  auto issue = [&](Buf64 (&b)[CH]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int c = 0; c < CH; ++c)
        b[c].q[q] = reinterpret_cast<const uint4 *>(lp[c] + ldR * 64)[q];
    }
    if (++ldR == R) { ldR = 0; ldTile += G; setPtrs(ldTile < nTiles ? ldTile : blockIdx.x); }
  };

  Buf64 A[CH], B[CH];
  setPtrs(ldTile);
  if (ORDER == 0) {
    issue(A);
    stamp();
    stageTable<THREADS>(tab, ldsRes, d);
    stamp();
  } else {
    StageRegs<THREADS> sr;
    stageLoad<THREADS>(sr, d);     // table first in the memory queues (L2 hits, coalesced)
    issue(A);                      // then the first input round, in use order
    stamp();
    stageStore<THREADS>(sr, tab, ldsRes, d);  // waits for the table pieces only
    stamp();
  }
  __syncthreads();
  stamp();
  if (TIMING) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(); }

  Chain3<BOOK> cs[CH];
  uint64_t tile = blockIdx.x;
  uint32_t r = 0;
  auto begin = [&]() {
    if (r == 0) {
#pragma unroll
      for (int c = 0; c < CH; ++c) { cs[c].s = init; cs[c].accS = 0; cs[c].endv = 0; cs[c].startv = 0; cs[c].wasInit = 1; }
    }
  };
  auto finish = [&]() {
    if (++r == R) {
      r = 0;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const uint64_t ln = tile * linesPerTile + uint64_t(c) * THREADS + threadIdx.x;
        if (ln < io.n) {
          if (BOOK == 0) { io.res[ln] = int32_t(cs[c].s); continue; }
          const int32_t rr = cs[c].endv ? ldsRes[cs[c].accS] : 0;
          io.res[ln] = rr;
          io.end[ln] = rr ? uint64_t(cs[c].endv) : 0;
          if (BOOK >= 2) io.start[ln] = rr ? uint64_t(cs[c].startv) : 0;
        }
      }
      tile += G;
    }
  };

  for (uint64_t q = 0; q < Q;) {
    if (q + 1 < Q) issue(B);
    begin();
    doRound5<BOOK, CH>(A, cs, tab, r * 64, init, firstAccept);
    stamp();
    finish();
    if (++q >= Q) break;
    if (q + 1 < Q) issue(A);
    begin();
    doRound5<BOOK, CH>(B, cs, tab, r * 64, init, firstAccept);
    stamp();
    finish();
    ++q;
  }
  stamp();
}

// ------------------------------------------------------------------------------------------
// Variant V6: fine-grained streaming.  Per chain a ring of four 16-byte pieces; while piece p
// is walked, piece p+3 is requested into the slot piece p-1 has just vacated.  Nothing is
// requested in bulk at kernel start (a wave BLOCKS in its load instructions when the memory
// queues are full, which delays the table barrier): table pieces + the first input piece go
// out first, the rest streams just in time.
// ------------------------------------------------------------------------------------------
template <int BOOK, int CH, int LOOKUP = 1>
__device__ __forceinline__ void walk16(const uint4 (&piece)[CH], uint32_t (&s)[CH], Book5<BOOK, CH> &b,
                                       const uint8_t *tab, int qIdx, uint32_t init,
                                       uint32_t firstAccept) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = 16 * qIdx + 4 * k + j;
      uint32_t t[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const uint4 v = piece[c];
        const uint32_t word = k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w;
        const uint32_t a = __builtin_amdgcn_perm(s[c], word, 0x0c0c0400u | uint32_t(j));
        if (LOOKUP) t[c] = tab[a];
        else t[c] = (a * 0x9E37u >> 7) & 0xffu;
      }
      if (i > 0) bookStep<BOOK, CH>(b, s, uint32_t(i), init, firstAccept);
#pragma unroll
      for (int c = 0; c < CH; ++c) s[c] = t[c];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int THREADS, int CH, int BOOK, int LOOKUP = 1, int LOAD = 1, int TIMING = 0>
__global__ void __launch_bounds__(THREADS) k_v6(Dev d, Io io) {
  __shared__ __align__(16) uint8_t lds[kTabMax + 1024];
  int stampIdx = 0;
  auto stamp = [&]() {
    if (TIMING) {
      unsigned long long t = wall_clock64();
      if ((threadIdx.x & 63) == 0 && stampIdx < 16)
        g_stamps[(blockIdx.x * (THREADS / 64) + threadIdx.x / 64) * 16 + stampIdx] = t;
      ++stampIdx;
    }
  };
  stamp();
  uint8_t *tab = lds;
  int32_t *ldsRes = reinterpret_cast<int32_t *>(lds + kTabMax);
  const uint32_t init = d.init, firstAccept = d.firstAccept, lineLen = io.lineLen;
  const uint32_t R = lineLen / 64;
  const uint64_t linesPerTile = uint64_t(THREADS) * CH;
  const uint64_t nTiles = (io.n + linesPerTile - 1) / linesPerTile;
  const uint64_t G = gridDim.x;
  if (blockIdx.x >= nTiles) return;
  const uint64_t myTiles = (nTiles - blockIdx.x + G - 1) / G;
  const uint64_t Q = myTiles * R;  // 64-byte blocks this workgroup walks per chain

  auto blockPtr = [&](uint64_t tile, uint32_t r, int c) {
    uint64_t ln = tile * linesPerTile + uint64_t(c) * THREADS + threadIdx.x;
    if (ln >= io.n) ln = io.n - 1;
    return io.data + ln * lineLen + r * 64;
  };

  auto ld = [&](const uint8_t *p, int k) {
    if (LOAD) return reinterpret_cast<const uint4 *>(p)[k];
    const uint32_t x = uint32_t(reinterpret_cast<uintptr_t>(p)) * 2654435761u + k;
    return make_uint4(x, x * 40503u, x ^ 0x5bd1e995u, x + 0x27d4eb2fu);
  };
  uint4 slot[4][CH];
  const uint8_t *cur[CH], *nxt[CH];
  uint64_t tile = blockIdx.x;
  uint32_t r = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) cur[c] = blockPtr(tile, 0, c);

  StageRegs<THREADS> sr;
  stageLoad<THREADS>(sr, d);
#pragma unroll
  for (int c = 0; c < CH; ++c) slot[0][c] = ld(cur[c], 0);
  stamp();
  stageStore<THREADS>(sr, tab, ldsRes, d);
  stamp();
  __syncthreads();
  stamp();
#pragma unroll
  for (int c = 0; c < CH; ++c) slot[1][c] = ld(cur[c], 1);
#pragma unroll
  for (int c = 0; c < CH; ++c) slot[2][c] = ld(cur[c], 2);

  Chain3<BOOK> cs[CH];
  for (uint64_t q = 0; q < Q; ++q) {
    // where the NEXT 64-byte block of each chain lives (same lines, or the next tile's lines)
    const bool haveNext = q + 1 < Q;
    uint32_t nr = r + 1;
    uint64_t ntile = tile;
    if (nr == R) { nr = 0; ntile = tile + G; }
#pragma unroll
    for (int c = 0; c < CH; ++c) nxt[c] = blockPtr(haveNext ? ntile : tile, haveNext ? nr : r, c);

    if (r == 0) {
#pragma unroll
      for (int c = 0; c < CH; ++c) { cs[c].s = init; cs[c].accS = 0; cs[c].endv = 0; cs[c].startv = 0; cs[c].wasInit = 1; }
    }
    Book5<BOOK, CH> b;
    uint32_t s[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      b.accS[c] = cs[c].accS; b.endRel[c] = 0; b.startRel[c] = 0; b.wasInit[c] = cs[c].wasInit != 0;
      s[c] = cs[c].s;
    }
    // piece 0: request piece 3 of this block
#pragma unroll
    for (int c = 0; c < CH; ++c) slot[3][c] = ld(cur[c], 3);
    walk16<BOOK, CH, LOOKUP>(slot[0], s, b, tab, 0, init, firstAccept);
    if (q == 0) stamp();
    if (haveNext) {
#pragma unroll
      for (int c = 0; c < CH; ++c) slot[0][c] = ld(nxt[c], 0);
    }
    walk16<BOOK, CH, LOOKUP>(slot[1], s, b, tab, 1, init, firstAccept);
    if (haveNext) {
#pragma unroll
      for (int c = 0; c < CH; ++c) slot[1][c] = ld(nxt[c], 1);
    }
    walk16<BOOK, CH, LOOKUP>(slot[2], s, b, tab, 2, init, firstAccept);
    if (haveNext) {
#pragma unroll
      for (int c = 0; c < CH; ++c) slot[2][c] = ld(nxt[c], 2);
    }
    walk16<BOOK, CH, LOOKUP>(slot[3], s, b, tab, 3, init, firstAccept);
    bookStep<BOOK, CH>(b, s, 64u, init, firstAccept);
    const uint32_t off = r * 64;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      cs[c].s = s[c];
      if (BOOK >= 1) { cs[c].accS = b.accS[c]; cs[c].endv = b.endRel[c] ? off + b.endRel[c] : cs[c].endv; }
      if (BOOK >= 2) { cs[c].startv = b.startRel[c] ? off + b.startRel[c] - 1 : cs[c].startv; cs[c].wasInit = b.wasInit[c]; }
    }
    stamp();
    if (++r == R) {
      r = 0;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const uint64_t ln = tile * linesPerTile + uint64_t(c) * THREADS + threadIdx.x;
        if (ln < io.n) {
          if (BOOK == 0) { io.res[ln] = int32_t(cs[c].s); continue; }
          const int32_t rr = cs[c].endv ? ldsRes[cs[c].accS] : 0;
          io.res[ln] = rr;
          io.end[ln] = rr ? uint64_t(cs[c].endv) : 0;
          if (BOOK >= 2) io.start[ln] = rr ? uint64_t(cs[c].startv) : 0;
        }
      }
      tile += G;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) cur[c] = nxt[c];
  }
  stamp();
}

// ------------------------------------------------------------------------------------------
// Variant V7: V6's streaming ring, with the byte step written in inline asm so the
// instruction stream is exactly: CH x (v_perm + ds_read_u8) first (the dependent chain), then
// the previous state's bookkeeping (2 v_cmp + 3 v_cndmask + 1 s_andn2 per chain) under the
// LDS latency, then one s_waitcnt.  6 VALU per byte per chain, no s_nop, no zero-extension.
// The table must sit at LDS offset 0 (it is the kernel's only LDS object).
// ------------------------------------------------------------------------------------------
struct Book7 {
  uint32_t acc, end, start;
};

#define V7_PERM(c) "v_perm_b32 %[a" #c "], %[s" #c "], %[w" #c "], %[sel]\n\t"
#define V7_READ(c) "ds_read_u8 %[t" #c "], %[a" #c "]\n\t"
#define V7_CMPA(c) "v_cmp_le_u32_e64 %[m" #c "], %[T], %[s" #c "]\n\t"
#define V7_CMPI(c) "v_cmp_eq_u32_e64 %[i" #c "], %[init], %[s" #c "]\n\t"
#define V7_ACC(c)  "v_cndmask_b32_e64 %[acc" #c "], %[acc" #c "], %[s" #c "], %[m" #c "]\n\t"
#define V7_END(c)  "v_cndmask_b32_e64 %[e" #c "], %[e" #c "], %[idx], %[m" #c "]\n\t"
#define V7_LEAVE(c) "s_andn2_b64 %[l" #c "], %[was" #c "], %[i" #c "]\n\t"
#define V7_START(c) "v_cndmask_b32_e64 %[st" #c "], %[st" #c "], %[idx], %[l" #c "]\n\t"

#define V7_OUT(c)                                                                         \
  [a##c] "=&v"(a[c]), [t##c] "=&v"(t[c]), [m##c] "=&s"(m[c]), [i##c] "=&s"(isI[c]),      \
  [l##c] "=&s"(l[c]), [acc##c] "+v"(b[c].acc), [e##c] "+v"(b[c].end), [st##c] "+v"(b[c].start)
#define V7_IN(c) [s##c] "v"(s[c]), [w##c] "v"(w[c]), [was##c] "s"(wasI[c])

// IDX = position (1..64) of the state being book-kept = steps taken so far in this block
template <int IDX>
__device__ __forceinline__ void step7x2(uint32_t (&s)[2], const uint32_t (&w)[2], Book7 (&b)[2],
                                        const uint64_t (&wasI)[2], uint64_t (&isI)[2],
                                        uint32_t sel, uint32_t T, uint32_t init) {
  uint32_t a[2], t[2];
  uint64_t m[2], l[2];
  asm volatile(V7_PERM(0) V7_PERM(1) V7_READ(0) V7_READ(1)
               V7_CMPA(0) V7_CMPA(1) V7_CMPI(0) V7_CMPI(1)
               V7_ACC(0) V7_ACC(1) V7_END(0) V7_END(1)
               V7_LEAVE(0) V7_LEAVE(1) V7_START(0) V7_START(1)
               "s_waitcnt lgkmcnt(0)"
               : V7_OUT(0), V7_OUT(1)
               : V7_IN(0), V7_IN(1), [sel] "s"(sel), [T] "s"(T), [init] "s"(init), [idx] "n"(IDX)
               : "memory", "scc");  // s_andn2_b64 writes SCC
  s[0] = t[0]; s[1] = t[1];
}

template <int IDX>
__device__ __forceinline__ void step7x4(uint32_t (&s)[4], const uint32_t (&w)[4], Book7 (&b)[4],
                                        const uint64_t (&wasI)[4], uint64_t (&isI)[4],
                                        uint32_t sel, uint32_t T, uint32_t init) {
  uint32_t a[4], t[4];
  uint64_t m[4], l[4];
  asm volatile(V7_PERM(0) V7_PERM(1) V7_PERM(2) V7_PERM(3)
               V7_READ(0) V7_READ(1) V7_READ(2) V7_READ(3)
               V7_CMPA(0) V7_CMPA(1) V7_CMPA(2) V7_CMPA(3)
               V7_CMPI(0) V7_CMPI(1) V7_CMPI(2) V7_CMPI(3)
               V7_ACC(0) V7_ACC(1) V7_ACC(2) V7_ACC(3)
               V7_END(0) V7_END(1) V7_END(2) V7_END(3)
               V7_LEAVE(0) V7_LEAVE(1) V7_LEAVE(2) V7_LEAVE(3)
               V7_START(0) V7_START(1) V7_START(2) V7_START(3)
               "s_waitcnt lgkmcnt(0)"
               : V7_OUT(0), V7_OUT(1), V7_OUT(2), V7_OUT(3)
               : V7_IN(0), V7_IN(1), V7_IN(2), V7_IN(3), [sel] "s"(sel), [T] "s"(T),
                 [init] "s"(init), [idx] "n"(IDX)
               : "memory", "scc");
  s[0] = t[0]; s[1] = t[1]; s[2] = t[2]; s[3] = t[3];
}

// the step BEFORE the first byte of a 64-byte block book-keeps nothing new (IDX = 0 would
// re-record the carried-in state at relative position 0 = "no event"); handled by giving the
// first step IDX 0 and treating end/start == 0 as "no event in this block".
template <int CH, int IDX>
__device__ __forceinline__ void step7(uint32_t (&s)[CH], const uint32_t (&w)[CH], Book7 (&b)[CH],
                                      const uint64_t (&wasI)[CH], uint64_t (&isI)[CH],
                                      uint32_t sel, uint32_t T, uint32_t init) {
  if constexpr (CH == 2) step7x2<IDX>(s, w, b, wasI, isI, sel, T, init);
  else step7x4<IDX>(s, w, b, wasI, isI, sel, T, init);
}

template <int CH, int Q>
__device__ __forceinline__ void walk7(const uint4 (&piece)[CH], uint32_t (&s)[CH], Book7 (&b)[CH],
                                      uint64_t (&mA)[CH], uint64_t (&mB)[CH], uint32_t T,
                                      uint32_t init) {
  // 16 steps; lane masks ping-pong between mA (was-init on even steps) and mB
  uint32_t w[CH];
#define V7_WORD(K, FIELD)                                                                   \
  _Pragma("unroll") for (int c = 0; c < CH; ++c) w[c] = piece[c].FIELD;                    \
  step7<CH, 16 * Q + 4 * K + 0>(s, w, b, mA, mB, 0x0c0c0400u, T, init);                    \
  step7<CH, 16 * Q + 4 * K + 1>(s, w, b, mB, mA, 0x0c0c0401u, T, init);                    \
  step7<CH, 16 * Q + 4 * K + 2>(s, w, b, mA, mB, 0x0c0c0402u, T, init);                    \
  step7<CH, 16 * Q + 4 * K + 3>(s, w, b, mB, mA, 0x0c0c0403u, T, init);
  V7_WORD(0, x) V7_WORD(1, y) V7_WORD(2, z) V7_WORD(3, w)
#undef V7_WORD
}

__device__ unsigned g_oob[8];

template <int THREADS, int CH, int SAFE = 0, int TIMING = 0, int EARLY = 1>
__global__ void __launch_bounds__(THREADS) k_v7(Dev d, Io io) {
  __shared__ __align__(16) uint8_t lds[kTabMax + 1024];
  int stampIdx = 0;
  auto stamp = [&]() {
    if (TIMING) {
      unsigned long long t = wall_clock64();
      if ((threadIdx.x & 63) == 0 && stampIdx < 16)
        g_stamps[(blockIdx.x * (THREADS / 64) + threadIdx.x / 64) * 16 + stampIdx] = t;
      ++stampIdx;
    }
  };
  stamp();
  uint8_t *tab = lds;
  int32_t *ldsRes = reinterpret_cast<int32_t *>(lds + kTabMax);
  const uint32_t init = d.init, firstAccept = d.firstAccept, lineLen = io.lineLen;
  const uint32_t R = lineLen / 64;
  const uint64_t linesPerTile = uint64_t(THREADS) * CH;
  const uint64_t nTiles = (io.n + linesPerTile - 1) / linesPerTile;
  const uint64_t G = gridDim.x;
  if (blockIdx.x >= nTiles) return;
  const uint64_t myTiles = (nTiles - blockIdx.x + G - 1) / G;
  const uint64_t Q = myTiles * R;

  auto blockPtr = [&](uint64_t tile, uint32_t r, int c) {
    uint64_t ln = tile * linesPerTile + uint64_t(c) * THREADS + threadIdx.x;
    if (ln >= io.n) ln = io.n - 1;
    return io.data + ln * lineLen + r * 64;
  };
  auto ld = [&](const uint8_t *p, int k) {
    if (SAFE) {
      const uint8_t *q = p + 16 * k;
      if (q < io.data || q + 16 > io.data + io.n * io.lineLen) {
        atomicAdd(&g_oob[0], 1u);
        g_oob[2] = uint32_t(reinterpret_cast<uintptr_t>(q) - reinterpret_cast<uintptr_t>(io.data));
        g_oob[3] = uint32_t((reinterpret_cast<uintptr_t>(q) - reinterpret_cast<uintptr_t>(io.data)) >> 32);
        g_oob[4] = blockIdx.x; g_oob[5] = threadIdx.x; g_oob[6] = k;
        return make_uint4(0, 0, 0, 0);
      }
    }
    return reinterpret_cast<const uint4 *>(p)[k];
  };

  uint4 slot[4][CH];
  const uint8_t *cur[CH], *nxt[CH];
  uint64_t tile = blockIdx.x;
  uint32_t r = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) cur[c] = blockPtr(tile, 0, c);

  StageRegs<THREADS> sr;
  stageLoad<THREADS>(sr, d);
  const int32_t myRes = threadIdx.x < d.nStates ? d.result[threadIdx.x] : 0;
  if (EARLY) {
#pragma unroll
    for (int c = 0; c < CH; ++c) slot[0][c] = ld(cur[c], 0);
  }
  {
    uint4 *dst = reinterpret_cast<uint4 *>(tab);
#pragma unroll
    for (uint32_t k = 0; k < kTabMax / 16 / THREADS; ++k) dst[k * THREADS + threadIdx.x] = sr.v[k];
    if (threadIdx.x < 256) ldsRes[threadIdx.x] = myRes;
  }
  stamp();
  __syncthreads();
  stamp();
  if (!EARLY) {
#pragma unroll
    for (int c = 0; c < CH; ++c) slot[0][c] = ld(cur[c], 0);
  }
#pragma unroll
  for (int c = 0; c < CH; ++c) slot[1][c] = ld(cur[c], 1);
#pragma unroll
  for (int c = 0; c < CH; ++c) slot[2][c] = ld(cur[c], 2);

  uint32_t s[CH], accS[CH], endv[CH], startv[CH];
  uint64_t mA[CH], mB[CH];
  for (uint64_t q = 0; q < Q; ++q) {
    const bool haveNext = q + 1 < Q;
    uint32_t nr = r + 1;
    uint64_t ntile = tile;
    if (nr == R) { nr = 0; ntile = tile + G; }
#pragma unroll
    for (int c = 0; c < CH; ++c) nxt[c] = blockPtr(haveNext ? ntile : tile, haveNext ? nr : r, c);
    if (r == 0) {
#pragma unroll
      for (int c = 0; c < CH; ++c) { s[c] = init; accS[c] = 0; endv[c] = 0; startv[c] = 0; mA[c] = ~0ull; mB[c] = ~0ull; }
    }
    Book7 b[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { b[c].acc = accS[c]; b[c].end = 0; b[c].start = 0; }
    // Step with IDX = k book-keeps the state reached after k bytes of this block; the first
    // step (IDX 0) re-examines the carried-in state: its acc/end select writes 0 = "no event"
    // and its init test refreshes the was-init mask for the block's first byte.
#pragma unroll
    for (int c = 0; c < CH; ++c) slot[3][c] = ld(cur[c], 3);
    walk7<CH, 0>(slot[0], s, b, mA, mB, firstAccept, init);
    stamp();
#pragma unroll
    for (int c = 0; c < CH; ++c) slot[0][c] = ld(nxt[c], 0);  // unconditional: exact vmcnt
    walk7<CH, 1>(slot[1], s, b, mA, mB, firstAccept, init);
    stamp();
#pragma unroll
    for (int c = 0; c < CH; ++c) slot[1][c] = ld(nxt[c], 1);  // unconditional: exact vmcnt
    walk7<CH, 2>(slot[2], s, b, mA, mB, firstAccept, init);
    stamp();
#pragma unroll
    for (int c = 0; c < CH; ++c) slot[2][c] = ld(nxt[c], 2);  // unconditional: exact vmcnt
    walk7<CH, 3>(slot[3], s, b, mA, mB, firstAccept, init);
    stamp();
    // the state after the block's 64th byte is book-kept by the NEXT block's IDX-0 step when
    // the line continues; at the end of a line do it here, in plain C++ (once per line).
    const uint32_t off = r * 64;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      // fold the block-relative events into absolute positions.  A recorded relative index k
      // (1..63) means "state after k bytes": end = off + k, start = off + k - 1.  Index 0 is
      // the carried-in state, already accounted for by the previous block.
      accS[c] = b[c].acc;
      endv[c] = b[c].end ? off + b[c].end : endv[c];
      startv[c] = b[c].start ? off + b[c].start - 1 : startv[c];
      // state after byte 64 of this block
      const bool acc64 = s[c] >= firstAccept;
      const bool isInit64 = s[c] == init;
      const bool wasInit63 = (mA[c] >> (threadIdx.x & 63)) & 1;  // mask written by the last step
      if (acc64) { accS[c] = s[c]; endv[c] = off + 64; }
      if (wasInit63 && !isInit64) startv[c] = off + 63;
    }
    if (++r == R) {
      r = 0;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const uint64_t ln = tile * linesPerTile + uint64_t(c) * THREADS + threadIdx.x;
        if (SAFE && accS[c] >= 256) { atomicAdd(&g_oob[1], 1u); accS[c] = 0; }
        if (ln < io.n) {
          const int32_t rr = endv[c] ? ldsRes[accS[c]] : 0;
          io.res[ln] = rr;
          io.end[ln] = rr ? uint64_t(endv[c]) : 0;
          io.start[ln] = rr ? uint64_t(startv[c]) : 0;
        }
      }
      tile += G;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) cur[c] = nxt[c];
  }
  stamp();
}

// ------------------------------------------------------------------------------------------
// Interference microbenchmark: the LDS gather chains of k_lds and a global read stream in the
// SAME waves, with no data dependence between them (lookup bytes come from an LCG).  Per 64
// lookups per lane, LOADS 16-byte loads per lane are issued (4 = the real ratio for 64-byte
// lines) either coalesced (STRIDED 0) or one-line-per-lane (STRIDED 1).
// ------------------------------------------------------------------------------------------
template <int THREADS, int CH, int LOADS, int STRIDED>
__global__ void __launch_bounds__(THREADS) k_mix(Dev d, Io io) {
  __shared__ __align__(16) uint8_t lds[kTabMax + 1024];
  stageTable<THREADS>(lds, reinterpret_cast<int32_t *>(lds + kTabMax), d);
  __syncthreads();
  // every lane owns 64*CH-byte "lines"; rounds of 64 lookups per chain
  const uint64_t total = io.n * io.lineLen;
  const uint32_t rounds = uint32_t(total / (uint64_t(gridDim.x) * THREADS * CH * 64));
  uint32_t s[CH], x[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) { s[c] = (threadIdx.x + c) & 0xff; x[c] = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u + c * 97u); }
  uint32_t acc = 0;
  const uint8_t *base = io.data + uint64_t(blockIdx.x) * rounds * THREADS * CH * 64;
  uint4 cur[LOADS > 0 ? LOADS * CH : 1];
  auto issue = [&](uint32_t r) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
      for (int k = 0; k < LOADS; ++k) {
        const uint8_t *p = base + (uint64_t(r) * CH + c) * THREADS * 64;
        if (STRIDED) p += uint64_t(threadIdx.x) * 64 + k * 16;
        else p += (uint64_t(k) * THREADS + threadIdx.x) * 16;
        cur[c * LOADS + k] = *reinterpret_cast<const uint4 *>(p);
      }
    }
  };
  if (LOADS) issue(0);
  for (uint32_t r = 0; r < rounds; ++r) {
    if (LOADS) {
#pragma unroll
      for (int i = 0; i < LOADS * CH; ++i) acc += cur[i].x ^ cur[i].y ^ cur[i].z ^ cur[i].w;
      issue(r + 1 < rounds ? r + 1 : r);
    }
#pragma unroll 4
    for (int i = 0; i < 64; i += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          // xorshift bytes: no cross-lane arithmetic-progression artefacts
          s[c] = lds[__builtin_amdgcn_perm(s[c], x[c], 0x0c0c0400u | uint32_t(j))];
        }
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) { x[c] ^= x[c] << 13; x[c] ^= x[c] >> 17; x[c] ^= x[c] << 5; }
    }
  }
#pragma unroll
  for (int c = 0; c < CH; ++c) acc += s[c];
  io.res[blockIdx.x * THREADS + threadIdx.x] = int32_t(acc);
}

template <int THREADS, int CH, int LOADS, int STRIDED, int BLOCKS_PER_CU>
void launchMix(const Dev &d, const Io &io, int numCUs, hipStream_t s) {
  hipLaunchKernelGGL((k_mix<THREADS, CH, LOADS, STRIDED>), dim3(uint32_t(numCUs * BLOCKS_PER_CU)),
                     dim3(THREADS), 0, s, d, io);
}

#define MIX(T, C, L, S, BPC)                                                                    \
  Variant{"mix T" #T " CH" #C " loads" #L " strided" #S " bpc" #BPC, false, false, launchMix<T, C, L, S, BPC>}

// ------------------------------------------------------------------------------------------
struct Variant {
  std::string name;
  bool checkable;  // outputs are meant to be right
  bool hasStart;
  void (*launch)(const Dev &, const Io &, int numCUs, hipStream_t);
};

template <int THREADS, int CHAINS, int BOOK, int LOOKUP, int LOAD, int BLOCKS_PER_CU>
void launchV1(const Dev &d, const Io &io, int numCUs, hipStream_t s) {
  auto kern = k_v1<THREADS, CHAINS, BOOK, LOOKUP, LOAD>;
  size_t ldsBytes = size_t(d.tableBytes) + size_t(d.nStates) * 4;
  static bool attrDone = false;
  if (!attrDone) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                           hipFuncAttributeMaxDynamicSharedMemorySize, int(ldsBytes)));
    attrDone = true;
  }
  uint64_t linesPerTile = uint64_t(THREADS) * CHAINS;
  uint64_t tiles = (io.n + linesPerTile - 1) / linesPerTile;
  uint64_t blocks = std::min<uint64_t>(tiles, uint64_t(numCUs) * BLOCKS_PER_CU);
  hipLaunchKernelGGL(kern, dim3(uint32_t(blocks)), dim3(THREADS), ldsBytes, s, d, io);
}

#define V1(T, C, B, LK, LD, BPC)                                                      \
  Variant{"v1 T" #T " C" #C " book" #B " lookup" #LK " load" #LD " bpc" #BPC,        \
          (B) >= 1 && (LK) == 1 && (LD) == 1, (B) >= 2, launchV1<T, C, B, LK, LD, BPC>}


template <int THREADS, int CHAINS, int BOOK, int PERM, int BLOCKS_PER_CU>
void launchV2(const Dev &d, const Io &io, int numCUs, hipStream_t s) {
  auto kern = k_v2<THREADS, CHAINS, BOOK, PERM>;
  size_t ldsBytes = size_t(d.tableBytes) + size_t(d.nStates) * 4;
  static bool attrDone = false;
  if (!attrDone) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                           hipFuncAttributeMaxDynamicSharedMemorySize, int(ldsBytes)));
    attrDone = true;
  }
  uint64_t linesPerTile = uint64_t(THREADS) * CHAINS;
  uint64_t tiles = (io.n + linesPerTile - 1) / linesPerTile;
  uint64_t blocks = std::min<uint64_t>(tiles, uint64_t(numCUs) * BLOCKS_PER_CU);
  hipLaunchKernelGGL(kern, dim3(uint32_t(blocks)), dim3(THREADS), ldsBytes, s, d, io);
}

#define V2(T, C, B, P, BPC)                                                  \
  Variant{"v2 T" #T " C" #C " book" #B " perm" #P " bpc" #BPC, (B) >= 1, (B) >= 2, \
          launchV2<T, C, B, P, BPC>}

template <int THREADS, int CH, int BOOK, int BLOCKS_PER_CU>
void launchV3(const Dev &d, const Io &io, int numCUs, hipStream_t s) {
  auto kern = k_v3<THREADS, CH, BOOK>;
  size_t ldsBytes = size_t(d.tableBytes) + size_t(d.nStates) * 4;
  static bool attrDone = false;
  if (!attrDone) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                           hipFuncAttributeMaxDynamicSharedMemorySize, int(ldsBytes)));
    attrDone = true;
  }
  uint64_t linesPerTile = uint64_t(THREADS) * CH;
  uint64_t tiles = (io.n + linesPerTile - 1) / linesPerTile;
  uint64_t blocks = std::min<uint64_t>(tiles, uint64_t(numCUs) * BLOCKS_PER_CU);
  hipLaunchKernelGGL(kern, dim3(uint32_t(blocks)), dim3(THREADS), ldsBytes, s, d, io);
}

#define V3(T, C, B, BPC)                                                     \
  Variant{"v3 T" #T " C" #C " book" #B " bpc" #BPC, (B) >= 1, (B) >= 2, launchV3<T, C, B, BPC>}

template <int THREADS, int CH, int BOOK, int FENCE, int BLOCKS_PER_CU>
void launchV4(const Dev &d, const Io &io, int numCUs, hipStream_t s) {
  auto kern = k_v4<THREADS, CH, BOOK, FENCE>;
  size_t ldsBytes = size_t(d.tableBytes) + size_t(d.nStates) * 4;
  static bool attrDone = false;
  if (!attrDone) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                           hipFuncAttributeMaxDynamicSharedMemorySize, int(ldsBytes)));
    attrDone = true;
  }
  uint64_t linesPerTile = uint64_t(THREADS) * CH;
  uint64_t tiles = (io.n + linesPerTile - 1) / linesPerTile;
  uint64_t blocks = std::min<uint64_t>(tiles, uint64_t(numCUs) * BLOCKS_PER_CU);
  hipLaunchKernelGGL(kern, dim3(uint32_t(blocks)), dim3(THREADS), ldsBytes, s, d, io);
}

#define V4(T, C, B, F, BPC)                                                     \
  Variant{"v4 T" #T " C" #C " book" #B " fence" #F " bpc" #BPC, (B) >= 1, (B) >= 2, launchV4<T, C, B, F, BPC>}

template <int THREADS, int CH, int BOOK, int BLOCKS_PER_CU, int ORDER = 0>
void launchV5(const Dev &d, const Io &io, int numCUs, hipStream_t s) {
  auto kern = k_v5<THREADS, CH, BOOK, 0, ORDER>;
  uint64_t linesPerTile = uint64_t(THREADS) * CH;
  uint64_t tiles = (io.n + linesPerTile - 1) / linesPerTile;
  uint64_t blocks = std::min<uint64_t>(tiles, uint64_t(numCUs) * BLOCKS_PER_CU);
  hipLaunchKernelGGL(kern, dim3(uint32_t(blocks)), dim3(THREADS), 0, s, d, io);
}

#define V5(T, C, B, BPC)                                                     \
  Variant{"v5 T" #T " C" #C " book" #B " bpc" #BPC, (B) >= 1, (B) >= 2, launchV5<T, C, B, BPC>}

#define V5O(T, C, B, BPC)                                                     \
  Variant{"v5 order1 T" #T " C" #C " book" #B " bpc" #BPC, (B) >= 1, (B) >= 2, launchV5<T, C, B, BPC, 1>}

template <int THREADS, int CH, int BOOK>
void timelineV6(const Dev &d, Io io, uint8_t **in, int numCUs) {
  const int waves = numCUs * (THREADS / 64);
  std::vector<unsigned long long> h(size_t(waves) * 16, 0);
  uint64_t linesPerTile = uint64_t(THREADS) * CH;
  uint64_t tiles = (io.n + linesPerTile - 1) / linesPerTile;
  uint64_t blocks = std::min<uint64_t>(tiles, uint64_t(numCUs));
  for (int it = 0; it < 12; ++it) {  // warm: the LAST launch's stamps survive
    io.data = in[it % 6];
    hipLaunchKernelGGL((k_v6<THREADS, CH, BOOK, 1, 1, 1>), dim3(uint32_t(blocks)), dim3(THREADS), 0, 0, d, io);
  }
  CK(hipDeviceSynchronize());
  CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_stamps), h.size() * 8));
  unsigned long long t0 = ~0ull;
  for (int w = 0; w < waves; ++w) if (h[w * 16]) t0 = std::min(t0, h[w * 16]);
  printf("timeline v6 T%d C%d book%d, warm (us since first wave start): min / avg / max over waves\n", THREADS, CH, BOOK);
  const char *names[] = {"entry", "table+1st piece issued", "table in LDS", "barrier passed", "piece 0 walked",
                         "block 1 done", "block 2 done", "block 3 done", "block 4 done", "b5", "b6", "b7", "b8", "b9", "b10", "b11"};
  for (int k = 0; k < 16; ++k) {
    double mn = 1e30, mx = 0, sum = 0; int n = 0;
    for (int w = 0; w < waves; ++w) {
      unsigned long long v = h[w * 16 + k];
      if (!v) continue;
      double us = double(v - t0) / 100.0;
      mn = std::min(mn, us); mx = std::max(mx, us); sum += us; ++n;
    }
    if (n) printf("   %-24s %7.2f / %7.2f / %7.2f   (n=%d)\n", names[k], mn, sum / n, mx, n);
  }
}

template <int THREADS, int CH>
void timelineV7(const Dev &d, Io io, uint8_t **in, int numCUs) {
  const int waves = numCUs * (THREADS / 64);
  std::vector<unsigned long long> h(size_t(waves) * 16, 0);
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), h.data(), h.size() * 8));
  uint64_t linesPerTile = uint64_t(THREADS) * CH;
  uint64_t tiles = (io.n + linesPerTile - 1) / linesPerTile;
  uint64_t blocks = std::min<uint64_t>(tiles, uint64_t(numCUs));
  for (int it = 0; it < 12; ++it) {
    io.data = in[it % 6];
    hipLaunchKernelGGL((k_v7<THREADS, CH, 0, 1, 0>), dim3(uint32_t(blocks)), dim3(THREADS), 0, 0, d, io);
  }
  CK(hipDeviceSynchronize());
  CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_stamps), h.size() * 8));
  unsigned long long t0 = ~0ull;
  for (int w = 0; w < waves; ++w) if (h[w * 16]) t0 = std::min(t0, h[w * 16]);
  printf("timeline v7 LATE-loads T%d C%d, warm (us since first wave start): min / avg / max over waves\n", THREADS, CH);
  const char *names[] = {"entry", "table written", "barrier passed", "piece 0", "piece 1", "piece 2", "piece 3",
                         "piece 4", "piece 5", "piece 6", "piece 7", "p8/end", "p9", "p10", "p11", "p12"};
  for (int k = 0; k < 16; ++k) {
    double mn = 1e30, mx = 0, sum = 0; int n = 0;
    for (int w = 0; w < waves; ++w) {
      unsigned long long v = h[w * 16 + k];
      if (!v) continue;
      double us = double(v - t0) / 100.0;
      mn = std::min(mn, us); mx = std::max(mx, us); sum += us; ++n;
    }
    if (n) printf("   %-24s %7.2f / %7.2f / %7.2f   (n=%d)\n", names[k], mn, sum / n, mx, n);
  }
}

template <int THREADS, int CH, int BOOK, int ORDER = 0>
void timelineV5(const Dev &d, const Io &io, int numCUs) {
  const int waves = numCUs * (THREADS / 64);
  std::vector<unsigned long long> h(size_t(waves) * 16, 0);
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), h.data(), h.size() * 8));
  uint64_t linesPerTile = uint64_t(THREADS) * CH;
  uint64_t tiles = (io.n + linesPerTile - 1) / linesPerTile;
  uint64_t blocks = std::min<uint64_t>(tiles, uint64_t(numCUs));
  hipLaunchKernelGGL((k_v5<THREADS, CH, BOOK, 1, ORDER>), dim3(uint32_t(blocks)), dim3(THREADS), 0, 0, d, io);
  CK(hipDeviceSynchronize());
  CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_stamps), h.size() * 8));
  unsigned long long t0 = ~0ull;
  for (int w = 0; w < waves; ++w) if (h[w * 16]) t0 = std::min(t0, h[w * 16]);
  printf("timeline v5 order%d T%d C%d book%d (us since first wave start; 100 MHz clock): stamp: min / avg / max over waves\n", ORDER, THREADS, CH, BOOK);
  const char *names[] = {"entry", "loads issued", "table staged", "barrier passed", "first loads landed",
                         "round 1 done", "round 2 done", "round 3 done", "round 4 done", "round 5", "round 6", "r7", "r8", "r9", "r10", "r11"};
  for (int k = 0; k < 16; ++k) {
    double mn = 1e30, mx = 0, sum = 0; int n = 0;
    for (int w = 0; w < waves; ++w) {
      unsigned long long v = h[w * 16 + k];
      if (!v) continue;
      double us = double(v - t0) / 100.0;
      mn = std::min(mn, us); mx = std::max(mx, us); sum += us; ++n;
    }
    if (n) printf("   %-20s %7.2f / %7.2f / %7.2f   (n=%d)\n", names[k], mn, sum / n, mx, n);
  }
}

template <int THREADS, int CH, int BOOK, int BLOCKS_PER_CU, int LOOKUP = 1, int LOAD = 1>
void launchV6(const Dev &d, const Io &io, int numCUs, hipStream_t s) {
  auto kern = k_v6<THREADS, CH, BOOK, LOOKUP, LOAD>;
  uint64_t linesPerTile = uint64_t(THREADS) * CH;
  uint64_t tiles = (io.n + linesPerTile - 1) / linesPerTile;
  uint64_t blocks = std::min<uint64_t>(tiles, uint64_t(numCUs) * BLOCKS_PER_CU);
  hipLaunchKernelGGL(kern, dim3(uint32_t(blocks)), dim3(THREADS), 0, s, d, io);
}

#define V6(T, C, B, BPC)                                                     \
  Variant{"v6 T" #T " C" #C " book" #B " bpc" #BPC, (B) >= 1, (B) >= 2, launchV6<T, C, B, BPC>}
#define V6A(T, C, B, BPC, LK, LD)                                            \
  Variant{"v6 T" #T " C" #C " book" #B " bpc" #BPC " lookup" #LK " load" #LD, false, false, launchV6<T, C, B, BPC, LK, LD>}

template <int THREADS, int CH, int BLOCKS_PER_CU, int SAFE = 0, int EARLY = 1>
void launchV7(const Dev &d, const Io &io, int numCUs, hipStream_t s) {
  auto kern = k_v7<THREADS, CH, SAFE, 0, EARLY>;
  uint64_t linesPerTile = uint64_t(THREADS) * CH;
  uint64_t tiles = (io.n + linesPerTile - 1) / linesPerTile;
  uint64_t blocks = std::min<uint64_t>(tiles, uint64_t(numCUs) * BLOCKS_PER_CU);
  hipLaunchKernelGGL(kern, dim3(uint32_t(blocks)), dim3(THREADS), 0, s, d, io);
}

#define V7(T, C, BPC) Variant{"v7 T" #T " C" #C " bpc" #BPC, true, true, launchV7<T, C, BPC>}
#define V7L(T, C, BPC) Variant{"v7 late-loads T" #T " C" #C " bpc" #BPC, true, true, launchV7<T, C, BPC, 0, 0>}
#define V7S(T, C, BPC) Variant{"v7 SAFE T" #T " C" #C " bpc" #BPC, true, true, launchV7<T, C, BPC, 1>}

#include "tune_variants.inc"

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: tune <dfa.reda> [name-filter] [lines] [lineLen]\n");
    return 1;
  }
  const char *filter = argc > 2 ? argv[2] : "";
  const uint64_t nLines = argc > 3 ? strtoull(argv[3], nullptr, 0) : (1ull << 20);
  const uint32_t lineLen = argc > 4 ? uint32_t(atoi(argv[4])) : 64;
  FILE *f = fopen(argv[1], "rb");
  if (!f) { perror("open"); return 1; }
  std::vector<uint8_t> blob;
  uint8_t buf[65536];
  size_t got;
  while ((got = fread(buf, 1, sizeof buf, f)) > 0) blob.insert(blob.end(), buf, buf + got);
  fclose(f);
  redgpu::DfaImage img;
  int code = 0;
  std::string err = redgpu::buildImage(blob.data(), blob.size(), 0, false, img, code);
  if (!err.empty()) { fprintf(stderr, "image: %s\n", err.c_str()); return 1; }
  if (img.tableKind != 1) { fprintf(stderr, "needs a fused-u8 DFA\n"); return 1; }

  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int numCUs = prop.multiProcessorCount;
  printf("# device %s, %d CUs, clock %d kHz; dfa states %u; %llu lines x %u B\n", prop.name,
         numCUs, prop.clockRate, img.nStates, (unsigned long long)nLines, lineLen);

  Dev d{};
  void *dTab, *dRes;
  size_t tabBytes = (img.table.size() + 15) & ~size_t(15);
  CK(hipMalloc(&dTab, tabBytes + 16));
  CK(hipMemcpy(dTab, img.table.data(), img.table.size(), hipMemcpyHostToDevice));
  CK(hipMalloc(&dRes, img.nStates * 4 + 16));
  CK(hipMemcpy(dRes, img.result.data(), img.nStates * 4, hipMemcpyHostToDevice));
  d.table = (const uint8_t *)dTab; d.result = (const int32_t *)dRes;
  d.nStates = img.nStates; d.init = img.init; d.firstAccept = img.firstAccept;
  d.tableBytes = uint32_t(tabBytes);

  const int NBUF = 6;
  const uint64_t bytes = nLines * lineLen;
  uint8_t *in[NBUF];
  for (int i = 0; i < NBUF; ++i) {
    CK(hipMalloc((void **)&in[i], bytes + 64));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, in[i], bytes / 8, 1000 + i);
  }
  int32_t *refRes, *res;
  uint64_t *refStart, *refEnd, *start, *end;
  CK(hipMalloc((void **)&refRes, nLines * 4)); CK(hipMalloc((void **)&res, nLines * 4));
  CK(hipMalloc((void **)&refStart, nLines * 8)); CK(hipMalloc((void **)&start, nLines * 8));
  CK(hipMalloc((void **)&refEnd, nLines * 8)); CK(hipMalloc((void **)&end, nLines * 8));
  unsigned *dBad;
  CK(hipMalloc((void **)&dBad, 4));
  Io refIo{in[0], nLines, lineLen, refRes, refStart, refEnd};
  hipLaunchKernelGGL(k_ref, dim3(uint32_t((nLines + 255) / 256)), dim3(256), 0, 0, d, refIo);
  CK(hipDeviceSynchronize());

  if (strstr(filter, "timeline")) {
    Io io{in[1], nLines, lineLen, res, start, end};
    timelineV7<1024, 2>(d, io, in, numCUs);
    timelineV7<1024, 4>(d, io, in, numCUs);
    timelineV7<512, 4>(d, io, in, numCUs);
    return 0;
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int ROUNDS = 3, ITERS = 60;
  for (const Variant &v : variants()) {
    if (filter[0] && v.name.find(filter) == std::string::npos) continue;
    Io io{in[0], nLines, lineLen, res, start, end};
    CK(hipMemset(res, 0xff, nLines * 4));
    CK(hipMemset(start, 0xff, nLines * 8));
    CK(hipMemset(end, 0xff, nLines * 8));
    v.launch(d, io, numCUs, 0);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    unsigned bad = 0;
    if (v.checkable) {
      CK(hipMemset(dBad, 0, 4));
      hipLaunchKernelGGL(k_cmp, dim3(uint32_t((nLines + 255) / 256)), dim3(256), 0, 0, res, refRes,
                         v.hasStart ? start : refStart, refStart, end, refEnd, nLines, dBad);
      CK(hipMemcpy(&bad, dBad, 4, hipMemcpyDeviceToHost));
    }
    float best = 1e30f, sum = 0;
    for (int r = 0; r < ROUNDS; ++r) {
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < ITERS; ++i) {
        io.data = in[i % NBUF];
        v.launch(d, io, numCUs, 0);
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= ITERS;
      best = std::min(best, ms);
      sum += ms;
    }
    printf("%-52s %8.2f us (avg %8.2f)  %8.1f GB/s  %s\n", v.name.c_str(), best * 1e3,
           sum / ROUNDS * 1e3, bytes / (best * 1e-3) / 1e9,
           !v.checkable ? "ablation" : bad ? "MISMATCH" : "ok");
    // the same launches spread round-robin over 2..4 streams (independent batches overlap)
    if (strstr(filter, "streams") || getenv("TUNE_STREAMS")) {
      static hipStream_t st[4];
      static bool made = false;
      if (!made) { for (auto &x : st) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking)); made = true; }
      int32_t *resX[4]; uint64_t *stX[4], *enX[4];
      for (int k = 0; k < 4; ++k) {
        CK(hipMalloc((void **)&resX[k], nLines * 4)); CK(hipMalloc((void **)&stX[k], nLines * 8));
        CK(hipMalloc((void **)&enX[k], nLines * 8));
      }
      for (int ns = 2; ns <= 4; ++ns) {
        float bestS = 1e30f;
        for (int r = 0; r < ROUNDS; ++r) {
          CK(hipDeviceSynchronize());
          auto t0 = std::chrono::steady_clock::now();
          for (int i = 0; i < ITERS * 2; ++i) {
            Io io2{in[i % NBUF], nLines, lineLen, resX[i % ns], stX[i % ns], enX[i % ns]};
            v.launch(d, io2, numCUs, st[i % ns]);
          }
          CK(hipDeviceSynchronize());
          float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count() / (ITERS * 2);
          bestS = std::min(bestS, ms);
        }
        printf("      %d streams: %8.2f us per launch  %8.1f GB/s\n", ns, bestS * 1e3, bytes / (bestS * 1e-3) / 1e9);
      }
      for (int k = 0; k < 4; ++k) { CK(hipFree(resX[k])); CK(hipFree(stX[k])); CK(hipFree(enX[k])); }
    }
    if (v.checkable && bad) printf("   !! %u mismatching lines\n", bad);
    {
      unsigned oob[8];
      CK(hipMemcpyFromSymbol(oob, HIP_SYMBOL(g_oob), sizeof oob));
      if (oob[0] || oob[1])
        printf("   !! SAFE: %u out-of-range loads (last: byte offset 0x%x%08x block %u thread %u piece %u), %u bad accS\n",
               oob[0], oob[3], oob[2], oob[4], oob[5], oob[6], oob[1]);
      unsigned z[8] = {0};
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_oob), z, sizeof z));
    }
    fflush(stdout);
  }
  return 0;
}
