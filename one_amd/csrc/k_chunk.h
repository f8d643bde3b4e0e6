// k_chunk.h - intra-line parallelism for FEW LONG lines (BASELINE configs[4]'s shape); included
// by kernels.hip inside its namespace, after k_stream.h / k_ragged.h.
//
// A batch with fewer lines than the chip has lanes is bound by the LDS latency of each line's
// dependent chain (DESIGN 4.1 "few long lines").  For DFAs that forget their past - loose-start
// regexes spend most positions in the initial state - a line is cut into m chunks that are all
// walked AT ONCE, each from the initial state as a guess (k_stream, mode kSmChunk: raw
// Last-style records + exit state per chunk).  Then, per line and in order, a chunk whose true
// entry state (its predecessor's exit) was not the guess is walked again from the right state
// (k_chunk_rewalk, one chunk per lane); that takes one round per wrong guess in the line, all
// lines in parallel, kChunkRounds rounds at most, and whatever is still open after them is
// finished serially by k_chunk_combine, which also folds the chunk records into the Outcome.
// The guess is better than "the initial state": k_chunk_guess first walks the kChunkWarm bytes
// in front of every chunk from the initial state - what a regex DFA is in depends on the last
// few bytes only, unless the border falls inside a long match.
// Everything is queued on the caller's stream; nothing is read back on the host.
//
// Record of chunk k (scratch, SoA): st[k] = entry state in / exit state out, ent[k] = the entry
// state its record was computed from, acc[k] = state of the last accept or -1, end[k] = chunk-
// relative end of that accept (0 = none), start[k] = 1 + chunk-relative position of the last
// "left the initial state" (0 = none).
#pragma once

constexpr int kChunkRounds = 6;
constexpr uint32_t kChunkWarm = 64;

struct ChunkBufs {
  uint32_t *st, *ent, *pos, *work, *count;
  int32_t *acc;
  uint64_t *end, *start;
};

// count[r] = chunks queued for re-walk in round r (one counter per round, all zeroed here: round 2
// zeroed one shared counter with a memset in front of every round - six more launches per call)
__global__ void __launch_bounds__(256)
k_chunk_init(ChunkBufs cb, uint64_t nChunks, uint64_t n, uint32_t init) {
  const uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i < nChunks) { cb.st[i] = init; cb.ent[i] = init; }
  if (i < n) cb.pos[i] = 0;
  if (i <= uint64_t(kChunkRounds)) cb.count[i] = 0;
}

// per line: advance over the chunks whose record is known to be right; stop at the first whose
// entry state was guessed wrong and queue it for this round.  A round behind one that queued
// nothing has nothing to find (every line is through): it leaves at once.
__global__ void __launch_bounds__(256)
k_chunk_resolve(ChunkBufs cb, uint64_t n, uint32_t m, uint32_t init, int round) {
  if (round > 0 && cb.count[round - 1] == 0) return;
  cb.count += round;
  const uint64_t l = uint64_t(blockIdx.x) * 256 + threadIdx.x;
  if (l >= n) return;
  uint32_t j = cb.pos[l];
  const uint64_t base = l * m;
  while (j < m) {
    const uint32_t entry = j == 0 ? init : cb.st[base + j - 1];
    if (cb.ent[base + j] == entry) { ++j; continue; }
    cb.ent[base + j] = entry;
    cb.st[base + j] = entry;  // state in for the re-walk
    cb.work[atomicAdd(cb.count, 1u)] = uint32_t(base + j);
    break;
  }
  cb.pos[l] = j;
}

// one chunk's Last-style record from a given entry state; no early exit (an absorbing dead end
// just stays dead)
template <class T>
__device__ void chunkLane(const T &tab, const LaneCtx &c, const uint8_t *p, uint32_t len,
                          uint32_t &s, int32_t &acc, uint32_t &endv, uint32_t &start1) {
  acc = -1;
  endv = 0;
  start1 = 0;
  walkBytes(p, 0, len, [&](uint32_t byte, uint64_t idx) {
    const uint32_t was = s;
    s = tab.next(s, byte);
    if (was == c.init && s != was) start1 = uint32_t(idx) + 1;
    if (s >= c.firstAccept) { acc = int32_t(s); endv = uint32_t(idx) + 1; }
    return true;
  });
}

// entry-state guesses: chunk k > 0 of a line starts in whatever a walk from the initial state
// over the kChunkWarm bytes in front of it ends in (chunk 0 starts in the initial state, exactly)
template <int KIND, int kThreads>
__global__ void __launch_bounds__(kThreads)
k_chunk_guess(DevDfa d, const uint8_t *data, uint32_t chunkLen, uint32_t m, uint64_t nChunks,
              ChunkBufs cb) {
  extern __shared__ __align__(16) uint8_t lds[];
  const Tab<KIND> tab = stageTab<KIND, kThreads>(d, lds);
  const uint64_t step = uint64_t(gridDim.x) * kThreads;
  for (uint64_t k = uint64_t(blockIdx.x) * kThreads + threadIdx.x; k < nChunks; k += step) {
    uint32_t s = d.init;
    if (k % m) {
      const uint8_t *p = data + k * chunkLen - kChunkWarm;
      walkBytes(p, 0, kChunkWarm, [&](uint32_t byte, uint64_t) {
        s = tab.next(s, byte);
        return true;
      });
    }
    cb.st[k] = s;
    cb.ent[k] = s;
  }
}

template <int KIND, int kThreads>
__global__ void __launch_bounds__(kThreads)
k_chunk_rewalk(DevDfa d, const uint8_t *data, uint32_t chunkLen, ChunkBufs cb, int round) {
  extern __shared__ __align__(16) uint8_t lds[];
  cb.count += round;
  if (*cb.count == 0) return;  // uniform: nothing was guessed wrong this round
  uint8_t *eq = lds;
  uint8_t *leader = lds + 256;
  const Tab<KIND> tab = stageTab<KIND, kThreads>(d, lds);
  LaneCtx c{eq, leader, resOf<KIND>(d, lds), d.init, d.leaderNext, d.nPureDead, d.firstAccept,
            d.leaderLen};
  const uint32_t cnt = *cb.count;
  const uint64_t step = uint64_t(gridDim.x) * kThreads;
  for (uint64_t i = uint64_t(blockIdx.x) * kThreads + threadIdx.x; i < cnt; i += step) {
    const uint32_t k = cb.work[i];
    uint32_t s = cb.st[k];
    int32_t acc;
    uint32_t endv, start1;
    chunkLane(tab, c, data + uint64_t(k) * chunkLen, chunkLen, s, acc, endv, start1);
    cb.st[k] = s;
    cb.acc[k] = acc;
    cb.end[k] = endv;
    cb.start[k] = start1;
  }
}

// per line: finish what the rounds left open (serially), then fold the chunk records
template <int KIND, int kThreads>
__global__ void __launch_bounds__(kThreads)
k_chunk_combine(DevDfa d, Batch b, uint32_t m, uint32_t chunkLen, int style, ChunkBufs cb) {
  extern __shared__ __align__(16) uint8_t lds[];
  uint8_t *eq = lds;
  uint8_t *leader = lds + 256;
  const Tab<KIND> tab = stageTab<KIND, kThreads>(d, lds);
  LaneCtx c{eq, leader, resOf<KIND>(d, lds), d.init, d.leaderNext, d.nPureDead, d.firstAccept,
            d.leaderLen};
  const uint64_t step = uint64_t(gridDim.x) * kThreads;
  for (uint64_t l = uint64_t(blockIdx.x) * kThreads + threadIdx.x; l < b.n; l += step) {
    const uint64_t base = l * m;
    for (uint32_t j = cb.pos[l]; j < m; ++j) {
      const uint32_t entry = j == 0 ? d.init : cb.st[base + j - 1];
      if (cb.ent[base + j] == entry) continue;
      uint32_t s = entry;
      int32_t acc;
      uint32_t endv, start1;
      chunkLane(tab, c, b.data + l * b.stride + uint64_t(j) * chunkLen, chunkLen, s, acc, endv,
                start1);
      cb.ent[base + j] = entry;
      cb.st[base + j] = s;
      cb.acc[base + j] = acc;
      cb.end[base + j] = endv;
      cb.start[base + j] = start1;
    }
    int32_t rr = 0;
    uint64_t en = 0, st = 0;
    if (style == kStyFull) {
      const uint32_t fin = cb.st[base + m - 1];
      rr = c.resultOf(fin);
      en = b.stride;
    } else {
      for (uint32_t j = m; j-- > 0;)
        if (cb.end[base + j]) {
          rr = d.result[cb.acc[base + j]];
          en = uint64_t(j) * chunkLen + cb.end[base + j];
          break;
        }
    }
    if (rr && b.start)
      for (uint32_t j = m; j-- > 0;)
        if (cb.start[base + j]) { st = uint64_t(j) * chunkLen + cb.start[base + j] - 1; break; }
    b.result[l] = rr;
    if (b.end) b.end[l] = rr ? en : 0;
    if (b.start) b.start[l] = rr ? st : 0;
  }
}

template <int KIND>
hipError_t launchChunkGuess(const DevDfa &d, const Batch &b, uint32_t m, uint32_t chunkLen,
                            const ChunkBufs &cb, const LaunchCfg &cfg, hipStream_t stream) {
  constexpr bool kLds = Tab<KIND>::kInLds || KIND == REDGPU_TAB_HOT_ROWS;
  constexpr int kThreads = kLds ? 1024 : 256;
  const size_t ldsBytes = 512 + ldsTableBytes<KIND>(d);
  hipError_t e = setLds(k_chunk_guess<KIND, kThreads>, ldsBytes);
  if (e != hipSuccess) return e;
  const uint64_t nChunks = uint64_t(m) * b.n;
  const uint32_t perCu = kLds ? (ldsBytes <= 80 * 1024 ? 2 : 1) : 8;
  uint64_t blocks = (nChunks + kThreads - 1) / kThreads;
  if (blocks > uint64_t(cfg.numCUs) * perCu) blocks = uint64_t(cfg.numCUs) * perCu;
  hipLaunchKernelGGL((k_chunk_guess<KIND, kThreads>), dim3(uint32_t(blocks)), dim3(kThreads),
                     ldsBytes, stream, d, b.data, chunkLen, m, nChunks, cb);
  return hipGetLastError();
}

template <int KIND>
hipError_t launchChunkTail(const DevDfa &d, const Batch &b, uint32_t m, uint32_t chunkLen,
                           int style, const ChunkBufs &cb, const LaunchCfg &cfg,
                           hipStream_t stream) {
  constexpr bool kLds = Tab<KIND>::kInLds || KIND == REDGPU_TAB_HOT_ROWS;
  constexpr int kThreads = kLds ? 1024 : 256;
  const size_t ldsBytes = 512 + ldsTableBytes<KIND>(d);
  hipError_t e = setLds(k_chunk_rewalk<KIND, kThreads>, ldsBytes);
  if (e != hipSuccess) return e;
  e = setLds(k_chunk_combine<KIND, kThreads>, ldsBytes);
  if (e != hipSuccess) return e;
  const uint32_t perCu = kLds ? (ldsBytes <= 80 * 1024 ? 2 : 1) : 8;
  uint64_t blocks = (b.n + kThreads - 1) / kThreads;
  if (blocks > uint64_t(cfg.numCUs) * perCu) blocks = uint64_t(cfg.numCUs) * perCu;
  if (blocks == 0) blocks = 1;
  const uint32_t lineBlocks = uint32_t((b.n + 255) / 256);
  for (int round = 0; round < kChunkRounds; ++round) {
    hipLaunchKernelGGL(k_chunk_resolve, dim3(lineBlocks), dim3(256), 0, stream, cb, b.n, m, d.init,
                       round);
    hipLaunchKernelGGL((k_chunk_rewalk<KIND, kThreads>), dim3(uint32_t(blocks)), dim3(kThreads),
                       ldsBytes, stream, d, b.data, chunkLen, cb, round);
  }
  // (no resolve here: one that queued a chunk without a re-walk behind it would leave that
  // chunk marked as done; k_chunk_combine continues from pos[] on its own)
  hipLaunchKernelGGL((k_chunk_combine<KIND, kThreads>), dim3(uint32_t(blocks)), dim3(kThreads),
                     ldsBytes, stream, d, b, m, chunkLen, style, cb);
  return hipGetLastError();
}

// chunks per line: chunks of 2 KiB where the line length allows (1 - 4 KiB otherwise) - short
// enough that a round of re-walks is ~100 us of dependent lookups, long enough that a line has
// few of them; 0 = the batch is not one for this path
inline uint32_t chunksPerLine(const Batch &b, const LaunchCfg &cfg) {
  (void)cfg;
  if (b.offsets || b.n == 0 || b.stride < 4096 || b.stride % 64) return 0;
  for (uint32_t len : {2048u, 1024u, 4096u, 512u}) {
    if (b.stride % len == 0 && b.stride / len >= 2 && b.stride / len <= 65535)
      return uint32_t(b.stride / len);
  }
  return 0;
}

// verb check / match, style Last / Full, no leader work left to do (the caller filters)
inline hipError_t launchChunked(const DevDfa &d, const Batch &b, uint32_t m, int style,
                                const LaunchCfg &cfg, hipStream_t stream) {
  const uint64_t nChunks = uint64_t(m) * b.n;
  const uint32_t chunkLen = uint32_t(b.stride / m);
  // scratch: st, ent u32[nChunks]; acc i32[nChunks]; end, start u64[nChunks]; pos, work u32[n]; count
  const size_t bytes = size_t(nChunks) * (4 + 4 + 4 + 8 + 8) + size_t(b.n) * 8 + 64 + 64 +
                       size_t(kChunkRounds) * 4;
  void *scratch = nullptr;
  hipError_t e = raggedScratch(stream, bytes, &scratch);
  if (e != hipSuccess) return e;
  uint8_t *p = static_cast<uint8_t *>(scratch);
  ChunkBufs cb;
  cb.end = reinterpret_cast<uint64_t *>(p); p += nChunks * 8;
  cb.start = reinterpret_cast<uint64_t *>(p); p += nChunks * 8;
  cb.st = reinterpret_cast<uint32_t *>(p); p += nChunks * 4;
  cb.ent = reinterpret_cast<uint32_t *>(p); p += nChunks * 4;
  cb.acc = reinterpret_cast<int32_t *>(p); p += nChunks * 4;
  cb.pos = reinterpret_cast<uint32_t *>(p); p += b.n * 4;
  cb.work = reinterpret_cast<uint32_t *>(p); p += b.n * 4;
  cb.count = reinterpret_cast<uint32_t *>(p);
  const uint64_t initItems = nChunks > b.n ? nChunks : b.n;
  hipLaunchKernelGGL(k_chunk_init, dim3(uint32_t((initItems + 255) / 256)), dim3(256), 0, stream, cb,
                     nChunks, b.n, d.init);
  // the entry-state guesses
  if (d.tableKind == REDGPU_TAB_HOT_ROWS)
    e = launchChunkGuess<REDGPU_TAB_HOT_ROWS>(d, b, m, chunkLen, cb, cfg, stream);
  else if (d.tableKind == REDGPU_TAB_LDS_CLASS_U16)
    e = launchChunkGuess<REDGPU_TAB_LDS_CLASS_U16>(d, b, m, chunkLen, cb, cfg, stream);
  else if (d.tableKind == REDGPU_TAB_LDS_FUSED_U16)
    e = launchChunkGuess<REDGPU_TAB_LDS_FUSED_U16>(d, b, m, chunkLen, cb, cfg, stream);
  else
    e = launchChunkGuess<REDGPU_TAB_LDS_FUSED_U8>(d, b, m, chunkLen, cb, cfg, stream);
  if (e != hipSuccess) return e;
  // pass 1: every chunk at once, from its guess
  Batch cbatch{b.data, nullptr, chunkLen, nChunks, cb.acc, cb.start, cb.end};
  cbatch.state = cb.st;
  if (d.tableKind == REDGPU_TAB_HOT_ROWS)
    e = launchStreamHot<kSmChunk, kTabHot>(d, cbatch, cfg, stream);
  else if (d.tableKind == REDGPU_TAB_LDS_FUSED_U8)
    e = launchStreamT<kSmChunk>(d, cbatch, cfg, stream);
  else if (d.clsIndexForm)
    e = launchStreamHot<kSmChunk, kTabClsBig>(d, cbatch, cfg, stream);
  else
    e = launchStreamHot<kSmChunk, kTabCls>(d, cbatch, cfg, stream);
  if (e != hipSuccess) return e;
  if (d.tableKind == REDGPU_TAB_HOT_ROWS)
    return launchChunkTail<REDGPU_TAB_HOT_ROWS>(d, b, m, chunkLen, style, cb, cfg, stream);
  if (d.tableKind == REDGPU_TAB_LDS_CLASS_U16)
    return launchChunkTail<REDGPU_TAB_LDS_CLASS_U16>(d, b, m, chunkLen, style, cb, cfg, stream);
  if (d.tableKind == REDGPU_TAB_LDS_FUSED_U16)
    return launchChunkTail<REDGPU_TAB_LDS_FUSED_U16>(d, b, m, chunkLen, style, cb, cfg, stream);
  return launchChunkTail<REDGPU_TAB_LDS_FUSED_U8>(d, b, m, chunkLen, style, cb, cfg, stream);
}
