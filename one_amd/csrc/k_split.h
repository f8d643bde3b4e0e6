// k_split.h - line splitting on the device (lib/Util.cpp:109-130's rule): k_split_count / _scan / _scatter
// (included by kernels.hip inside namespace redgpu { namespace { ... } }; see its file map).
#pragma once

// ---- line splitting on the device ------------------------------------
// The rule is sampleLines' (lib/Util.cpp:109-130): a line is [start, position of the delimiter),
// the next one starts after the delimiter, and bytes after the last delimiter are not a line.
// Output is the offsets[n+1] array the ragged verbs take, line i = [offsets[i], offsets[i+1])
// INCLUDING its delimiter - the verbs are then called with stride = 1 (one trailing byte to
// drop).  Three passes: per-chunk delimiter counts, an exclusive scan of the counts, and the
// scatter; chunk = kSplitChunk bytes per workgroup.
constexpr uint32_t kSplitChunk = 16384;
constexpr int kSplitThreads = 256;

// bit k set <=> byte k of the 16 equals delim.  Per word: x = w ^ dddd has a zero byte where w
// holds the delimiter; the carry-free zero-byte test leaves 0x80 in exactly those bytes, and a
// multiply gathers the four flags into one nibble.
__device__ __forceinline__ uint32_t delimNibble(uint32_t w, uint32_t dddd) {
  const uint32_t x = w ^ dddd;
  const uint32_t t = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);  // 0x80 per zero byte
  return (((t >> 7) * 0x00204081u) >> 21) & 0xfu;
}
__device__ __forceinline__ uint32_t delimMask16(const uint4 v, uint32_t delim) {
  const uint32_t dddd = delim * 0x01010101u;
  return delimNibble(v.x, dddd) | (delimNibble(v.y, dddd) << 4) | (delimNibble(v.z, dddd) << 8) |
         (delimNibble(v.w, dddd) << 12);
}

// 16 bytes per lane per step; the buffer's head/tail that are not whole aligned 16-byte pieces
// are read byte by byte (lane 0 of the first / last chunk)
__device__ __forceinline__ uint32_t chunkPieceMask(const uint8_t *data, uint64_t len,
                                                   uint64_t pos, uint32_t delim) {
  if (pos >= len) return 0;
  if (pos + 16 <= len && (reinterpret_cast<uintptr_t>(data + pos) & 15u) == 0)
    return delimMask16(*reinterpret_cast<const uint4 *>(data + pos), delim);
  uint32_t m = 0;
  for (uint32_t k = 0; k < 16 && pos + k < len; ++k) m |= (data[pos + k] == delim ? 1u : 0u) << k;
  return m;
}

// pieces of a chunk per lane (kSplitChunk / (kSplitThreads * 16))
constexpr int kSplitPieces = int(kSplitChunk / (kSplitThreads * 16));

// the chunk's delimiter masks, one per piece of this lane: a chunk that lies whole inside an
// aligned buffer requests its kSplitPieces pieces back to back (round 2 asked for one at a time,
// behind a branch: 1.4-2.0 TB/s for a pass that has next to nothing to compute)
__device__ __forceinline__ void chunkMasks(const uint8_t *data, uint64_t len, uint64_t base,
                                           uint32_t delim, uint32_t (&m)[kSplitPieces]) {
  const bool fast = base + kSplitChunk <= len && (reinterpret_cast<uintptr_t>(data) & 15u) == 0;
  if (fast) {
    uint4 v[kSplitPieces];
#pragma unroll
    for (int k = 0; k < kSplitPieces; ++k)
      v[k] = *reinterpret_cast<const uint4 *>(data + base + uint64_t(k * kSplitThreads + threadIdx.x) * 16);
#pragma unroll
    for (int k = 0; k < kSplitPieces; ++k) m[k] = delimMask16(v[k], delim);
  } else {
#pragma unroll
    for (int k = 0; k < kSplitPieces; ++k)
      m[k] = chunkPieceMask(data, len, base + uint64_t(k * kSplitThreads + threadIdx.x) * 16, delim);
  }
}

__global__ void __launch_bounds__(kSplitThreads)
k_split_count(const uint8_t *data, uint64_t len, uint32_t delim, uint32_t *counts) {
  __shared__ uint32_t waveSum[kSplitThreads / 64];
  const uint64_t base = uint64_t(blockIdx.x) * kSplitChunk;
  uint32_t m[kSplitPieces];
  chunkMasks(data, len, base, delim, m);
  uint32_t c = 0;
#pragma unroll
  for (int k = 0; k < kSplitPieces; ++k) c += __popc(m[k]);
  for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) waveSum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < kSplitThreads / 64; ++w) t += waveSum[w];
    counts[blockIdx.x] = t;
  }
}

// one workgroup: exclusive scan of counts[nChunks] into bases[nChunks] (u64), total -> *nLines
__global__ void __launch_bounds__(1024)
k_split_scan(const uint32_t *counts, uint64_t nChunks, uint64_t *bases, uint64_t *nLines,
             uint64_t *offsets, uint64_t cap) {
  __shared__ uint64_t part[1024];
  const uint64_t per = (nChunks + 1023) / 1024;
  const uint64_t lo = uint64_t(threadIdx.x) * per;
  const uint64_t hi = lo + per < nChunks ? lo + per : nChunks;
  uint64_t sum = 0;
  for (uint64_t i = lo; i < hi; ++i) sum += counts[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t run = 0;
    for (int i = 0; i < 1024; ++i) { const uint64_t v = part[i]; part[i] = run; run += v; }
    *nLines = run;
    offsets[0] = 0;
  }
  __syncthreads();
  uint64_t run = part[threadIdx.x];
  for (uint64_t i = lo; i < hi; ++i) { bases[i] = run; run += counts[i]; }
}

__global__ void __launch_bounds__(kSplitThreads)
k_split_scatter(const uint8_t *data, uint64_t len, uint32_t delim, const uint64_t *bases,
                uint64_t *offsets, uint64_t cap) {
  __shared__ uint32_t waveBase[kSplitThreads / 64];
  __shared__ uint32_t roundBase;
  const uint64_t base = uint64_t(blockIdx.x) * kSplitChunk;
  const uint64_t first = bases[blockIdx.x];  // lines that end before this chunk
  if (threadIdx.x == 0) roundBase = 0;
  uint32_t masks[kSplitPieces];
  chunkMasks(data, len, base, delim, masks);
  __syncthreads();
#pragma unroll
  for (int piece = 0; piece < kSplitPieces; ++piece) {
    const uint64_t pos = base + uint64_t(piece * kSplitThreads + threadIdx.x) * 16;
    const uint32_t m = masks[piece];
    const uint32_t c = __popc(m);
    // exclusive prefix of c over the workgroup, in byte order (lane order = byte order)
    uint32_t incl = c;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = __shfl_up(incl, o);
      if ((threadIdx.x & 63) >= uint32_t(o)) incl += v;
    }
    if ((threadIdx.x & 63) == 63) waveBase[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t wb = 0, tot = 0;
    for (int w = 0; w < kSplitThreads / 64; ++w) {
      if (w < int(threadIdx.x >> 6)) wb += waveBase[w];
      tot += waveBase[w];
    }
    uint64_t k = first + roundBase + wb + (incl - c);  // index of this lane's first delimiter
    uint32_t mm = m;
    while (mm) {
      const uint32_t b = __ffs(mm) - 1;
      mm &= mm - 1;
      // line k ends at this delimiter: offsets[k + 1] = position after it
      if (k + 1 <= cap) offsets[k + 1] = pos + b + 1;
      ++k;
    }
    __syncthreads();
    if (threadIdx.x == 0) roundBase += tot;
    __syncthreads();
  }
}
