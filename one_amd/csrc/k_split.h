// k_split.h - line splitting on the device (lib/Util.cpp:109-130's rule): k_split_count / _scan / _scatter
// (included by kernels.hip inside namespace redgpu { namespace { ... } }; see its file map).
#pragma once

// ---- line splitting on the device ------------------------------------
// The rule is sampleLines' (lib/Util.cpp:109-130): a line is [start, position of the delimiter),
// the next one starts after the delimiter, and bytes after the last delimiter are not a line.
// Output is the offsets[n+1] array the ragged verbs take, line i = [offsets[i], offsets[i+1])
// INCLUDING its delimiter - the verbs are then called with stride = 1 (one trailing byte to
// drop).  Three passes: per-chunk delimiter counts, an exclusive scan of the counts, and the
// scatter; chunk = kSplitChunk bytes per workgroup.  The count pass is the only one that reads
// the text: it leaves its delimiter masks (one bit per byte, 1/8 of the input) for the scatter
// pass, which up to round 3 read the text a second time.
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

// chunks per workgroup of the count and scatter passes.  (Four - 16 requests per lane in flight, a
// quarter of the workgroups - measured in round 3 on 256 MiB: count pass 58 us either way, scatter
// pass 27 -> 37 us: its time is the sparse 8-byte stores of the offsets, and fewer workgroups hide
// them worse.)
constexpr int kSplitGroup = 1;

// masks[(chunk * kSplitPieces + piece) * kSplitThreads + lane] = the 16 delimiter bits of the
// lane's piece
__global__ void __launch_bounds__(kSplitThreads)
k_split_count(const uint8_t *data, uint64_t len, uint32_t delim, uint64_t nChunks, uint32_t *counts,
              uint16_t *masks) {
  __shared__ uint32_t waveSum[kSplitGroup][kSplitThreads / 64];
  const uint64_t chunk0 = uint64_t(blockIdx.x) * kSplitGroup;
  uint32_t m[kSplitGroup][kSplitPieces];
  const uint64_t base0 = chunk0 * kSplitChunk;
  if (base0 + uint64_t(kSplitGroup) * kSplitChunk <= len && (reinterpret_cast<uintptr_t>(data) & 15u) == 0) {
    // the whole group lies inside an aligned buffer: all its requests back to back
    uint4 v[kSplitGroup][kSplitPieces];
#pragma unroll
    for (int g = 0; g < kSplitGroup; ++g) {
#pragma unroll
      for (int k = 0; k < kSplitPieces; ++k)
        v[g][k] = *reinterpret_cast<const uint4 *>(data + base0 + uint64_t(g) * kSplitChunk +
                                                   uint64_t(k * kSplitThreads + threadIdx.x) * 16);
    }
#pragma unroll
    for (int g = 0; g < kSplitGroup; ++g) {
#pragma unroll
      for (int k = 0; k < kSplitPieces; ++k) m[g][k] = delimMask16(v[g][k], delim);
    }
  } else {
#pragma unroll
    for (int g = 0; g < kSplitGroup; ++g) chunkMasks(data, len, base0 + uint64_t(g) * kSplitChunk, delim, m[g]);
  }
#pragma unroll
  for (int g = 0; g < kSplitGroup; ++g) {
    if (chunk0 + g >= nChunks) break;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < kSplitPieces; ++k) {
      c += __popc(m[g][k]);
      masks[((chunk0 + g) * kSplitPieces + k) * kSplitThreads + threadIdx.x] = uint16_t(m[g][k]);
    }
    for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) waveSum[g][threadIdx.x >> 6] = c;
  }
  __syncthreads();
  if (threadIdx.x < kSplitGroup && chunk0 + threadIdx.x < nChunks) {
    uint32_t t = 0;
    for (int w = 0; w < kSplitThreads / 64; ++w) t += waveSum[threadIdx.x][w];
    counts[chunk0 + threadIdx.x] = t;
  }
}

// one workgroup: exclusive scan of counts[nChunks] into bases[nChunks] (u64), total -> *nLines.
// Tiles of 8 x 1024 counts, entry i of a tile at row i / 1024: eight independent requests and
// eight wave scans per lane, one pass over the 8 x 16 wave totals, a running carry between tiles.
// (The first form gave thread t the 16 contiguous counts of 256 MiB's 16384 chunks and had
// thread 0 walk the 1024 partial sums: 39 us, a third of the whole split; dependent loads and
// stores per thread still cost 28.)
__global__ void __launch_bounds__(1024)
k_split_scan(const uint32_t *counts, uint64_t nChunks, uint64_t *bases, uint64_t *nLines,
             uint64_t *offsets, uint64_t cap) {
  constexpr int kRows = 8;
  __shared__ uint32_t waveTot[2][kRows][16];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint64_t carry = 0;
  int buf = 0;
  for (uint64_t tile = 0; tile < nChunks; tile += uint64_t(kRows) * 1024, buf ^= 1) {
    uint32_t c[kRows], incl[kRows];
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
      const uint64_t i = tile + uint64_t(r) * 1024 + threadIdx.x;
      c[r] = i < nChunks ? counts[i] : 0u;
      incl[r] = c[r];
    }
    for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
      for (int r = 0; r < kRows; ++r) {
        const uint32_t v = __shfl_up(incl[r], o);
        if (lane >= uint32_t(o)) incl[r] += v;
      }
    }
    if (lane == 63) {
#pragma unroll
      for (int r = 0; r < kRows; ++r) waveTot[buf][r][wave] = incl[r];
    }
    __syncthreads();  // (the other buffer is rewritten only after the next tile's barrier)
    uint64_t before = carry;
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
      uint32_t mine = 0, tot = 0;
#pragma unroll
      for (int w = 0; w < 16; ++w) {
        const uint32_t v = waveTot[buf][r][w];
        if (w < int(wave)) mine += v;
        tot += v;
      }
      const uint64_t i = tile + uint64_t(r) * 1024 + threadIdx.x;
      if (i < nChunks) bases[i] = before + mine + (incl[r] - c[r]);
      before += tot;
    }
    carry = before;
  }
  if (threadIdx.x == 0) {
    *nLines = carry;
    offsets[0] = 0;
  }
}

// The scatter: the workgroup's delimiters in byte order are chunk-major, then piece-major (piece
// k = bytes [4096 k, 4096 (k + 1)) of its chunk), lanes in order inside a piece.  Independent wave
// scans of all pieces, one barrier, then every lane places its pieces' lines (the first form ran
// the pieces one after the other with three barriers each: 33 us per 256 MiB, now 27, for a pass
// that moves 47 MB).
__global__ void __launch_bounds__(kSplitThreads)
k_split_scatter(const uint16_t *allMasks, uint64_t nChunks, const uint64_t *bases, uint64_t *offsets,
                uint64_t cap) {
  constexpr int kWaves = kSplitThreads / 64;
  constexpr int kP = kSplitGroup * kSplitPieces;
  __shared__ uint32_t waveTot[kP][kWaves];
  const uint64_t chunk0 = uint64_t(blockIdx.x) * kSplitGroup;
  const uint64_t first = bases[chunk0];  // lines that end before this workgroup's bytes
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t m[kP], c[kP], incl[kP];
#pragma unroll
  for (int p = 0; p < kP; ++p) {
    const uint64_t chunk = chunk0 + p / kSplitPieces;
    m[p] = chunk < nChunks
               ? allMasks[(chunk * kSplitPieces + p % kSplitPieces) * kSplitThreads + threadIdx.x]
               : 0u;
    c[p] = __popc(m[p]);
    incl[p] = c[p];
  }
  for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
    for (int p = 0; p < kP; ++p) {
      const uint32_t v = __shfl_up(incl[p], o);
      if (lane >= uint32_t(o)) incl[p] += v;
    }
  }
  if (lane == 63) {
#pragma unroll
    for (int p = 0; p < kP; ++p) waveTot[p][wave] = incl[p];
  }
  __syncthreads();
  uint32_t before = 0;  // delimiters of the pieces and waves in front of (piece p, this wave)
#pragma unroll
  for (int p = 0; p < kP; ++p) {
    uint32_t mine = before, tot = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const uint32_t v = waveTot[p][w];
      if (w < int(wave)) mine += v;
      tot += v;
    }
    before += tot;
    uint64_t at = first + mine + (incl[p] - c[p]);  // index of this lane's first delimiter
    const uint64_t pos = (chunk0 + p / kSplitPieces) * kSplitChunk +
                         uint64_t((p % kSplitPieces) * kSplitThreads + threadIdx.x) * 16;
    uint32_t mm = m[p];
    while (mm) {
      const uint32_t b = __ffs(mm) - 1;
      mm &= mm - 1;
      // line `at` ends at this delimiter: offsets[at + 1] = position after it
      if (at + 1 <= cap) offsets[at + 1] = pos + b + 1;
      ++at;
    }
  }
}
