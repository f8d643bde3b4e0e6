// k_ragged_long.h - what k_ragged does about the LONG lines of a batch: the pre-pass that lists them
// (k_ragged_outliers: long lines go first, huge ones as pieces) and the fold of the pieces' records
// into their lines' Outcomes (k_ragged_pieces_fold).  Included by k_ragged.h in front of its
// launcher (inside kernels.hip's namespace).
#pragma once

// ---- the long lines of a batch (see k_ragged's header) ---------------------------------------
// A line is long from T = max(512, X x mean length) bytes on, delimiter bytes included; at most
// n / X lines can be (their lengths sum to no more than the buffer), which bounds the list.  Each
// workgroup counts the long lines of its contiguous share of offsets[], reserves that many list
// entries with ONE atomic, and writes them in a second pass over the same (now cached) offsets.
// Workgroup 0 also makes the tail pad (k_tail_pad's job).
//
// PIECES (`pieces` != 0: fused-u8 tables of DFAs that forget their past).  Even started first, a
// line is walked by one lane at ~105 ns per byte: 2 KB - the longest of 2^20 geometric lines of
// mean 144 - are 210 us, a megabyte of minified JSON in a log is 100 ms, and the launch waits.  A
// line of at least Y T bytes (delimiter dropped; Y = 8) is therefore listed as ceil(len / C) PIECES
// of C = T bytes instead: piece j > 0 is the "line" [j C - 64, (j + 1) C) whose first block is
// walked from the initial state only to arrive in a guess of the piece's entry state (what a
// regex DFA is in
// depends on the last few bytes, unless the border falls inside a long match); k_ragged walks
// pieces like lines - they are list entries, spread over all workgroups - and leaves a record
// per piece; k_ragged_pieces_fold then chains each line's records, re-walking a piece whose
// guess was wrong from its true entry state (k_chunk.h does this for fixed strides).
// Bounds: lines in pieces <= total / (Y T) <= n / (X Y); pieces <= sum(len / T + 1) over them
// <= n / X + n / (X Y); list entries = the other long lines + pieces <= 2 n / X + n / (X Y).
// (Cutting EVERY long line into pieces of T / 2 was measured too: it loses - 2^20 geometric lines
// 209 against 196 us, 2^23 1024 against 933, one 1 MB line 1101 against 621 us - the lead-ins, the
// list and the fold cost more than a 9-turn line's latency, which the other lanes' lines hide.)
constexpr int kOutlierThreads = 256;
constexpr uint32_t kOutlierMinBytes = 512;
constexpr uint32_t kPieceLead = 64;           // bytes walked in front of a piece: one block
constexpr uint64_t kEntryPiece = 1ull << 63;  // flags in the top bits of a list entry's end
constexpr uint64_t kEntryLead = 1ull << 62;

struct OutlierBufs {
  uint32_t *ctl;     // [0] entries, [1] T, [2] lines in pieces, [3] pieces, [4] C (8 words)
  uint32_t *ctlNext; // the control words of the NEXT call on this thread and stream: zeroed here
  uint32_t *outLn;   // [capE] line index, or the piece's index
  uint64_t *outRec;  // [capE][2] first byte walked, end (+ trailing bytes to drop) | flags
  uint32_t *hugeLn, *hugeFirst;  // [capH]
  uint32_t capE, capH, capP;
};

__global__ void __launch_bounds__(kOutlierThreads)
k_ragged_outliers(const uint8_t *data, const uint64_t *offsets, uint64_t nMax, const uint64_t *nDev,
                  uint32_t factor, uint32_t trim, uint32_t hugeX, uint8_t *pad, OutlierBufs ob) {
  __shared__ uint32_t cnt[3], base[3], fill[3];  // entries, huge lines, pieces
  const uint64_t n = raggedLineCount(nMax, nDev);
  const uint64_t total = offsets[n];
  if (blockIdx.x == 0 && threadIdx.x < 192) {
    const uint64_t i = (total >= 128 ? total - 128 : 0) + threadIdx.x;
    pad[threadIdx.x] = i < total ? data[i] : uint8_t(0);
  }
  uint64_t T = n ? (total + n - 1) / n * factor : 0xffffffffull;
  if (T < kOutlierMinBytes) T = kOutlierMinBytes;
  T = (T + 63) & ~63ull;  // pieces are whole blocks
  if (T >= 0xffffffffull) T = 0xffffffffull;  // "no line is long" to k_ragged
  const uint64_t C = T;
  if (blockIdx.x == 0 && threadIdx.x == 0) { ob.ctl[1] = uint32_t(T); ob.ctl[4] = uint32_t(C); }
  if (blockIdx.x == 0 && threadIdx.x < 8) ob.ctlNext[threadIdx.x] = 0;
  if (T == 0xffffffffull) return;
  if (threadIdx.x < 3) { cnt[threadIdx.x] = 0; fill[threadIdx.x] = 0; }
  __syncthreads();
  const uint64_t lo = n * blockIdx.x / gridDim.x, hi = n * (blockIdx.x + 1) / gridDim.x;
  // pieces of a long line: 0 = it stays whole
  auto piecesOf = [&](uint64_t raw) -> uint32_t {
    const uint64_t eff = raw >= trim ? raw - trim : 0;
    return hugeX && eff >= hugeX * T ? uint32_t((eff + C - 1) / C) : 0u;
  };
  uint32_t mine[3] = {0, 0, 0};
  // (four independent pairs of requests per trip: one pair per trip ran at 1.2 TB/s)
  for (uint64_t i = lo + threadIdx.x; i < hi; i += 4 * kOutlierThreads) {
    uint64_t a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint64_t at = i + uint64_t(k) * kOutlierThreads;
      a[k] = offsets[at < hi ? at : lo];
      b[k] = offsets[at < hi ? at + 1 : lo];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (b[k] - a[k] < T) continue;
      const uint32_t p = piecesOf(b[k] - a[k]);
      mine[0] += p ? p : 1u;
      mine[1] += p ? 1u : 0u;
      mine[2] += p;
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    for (int o = 32; o; o >>= 1) mine[q] += __shfl_xor(mine[q], o);
    if ((threadIdx.x & 63) == 0 && mine[q]) atomicAdd(&cnt[q], mine[q]);
  }
  __syncthreads();
  if (cnt[0] == 0) return;
  if (threadIdx.x == 0) base[0] = atomicAdd(&ob.ctl[0], cnt[0]);
  if (threadIdx.x == 1 && cnt[1]) base[1] = atomicAdd(&ob.ctl[2], cnt[1]);
  if (threadIdx.x == 2 && cnt[2]) base[2] = atomicAdd(&ob.ctl[3], cnt[2]);
  __syncthreads();
  for (uint64_t i = lo + threadIdx.x; i < hi; i += kOutlierThreads) {
    const uint64_t o = offsets[i], e = offsets[i + 1];
    if (e - o < T) continue;
    const uint32_t p = piecesOf(e - o);
    if (!p) {
      const uint32_t k = base[0] + atomicAdd(&fill[0], 1u);
      if (k >= ob.capE) continue;  // (cannot happen: see the bounds above)
      ob.outLn[k] = uint32_t(i);
      ob.outRec[2 * uint64_t(k)] = o;
      ob.outRec[2 * uint64_t(k) + 1] = e;
      continue;
    }
    const uint32_t k = base[0] + atomicAdd(&fill[0], p);
    const uint32_t h = base[1] + atomicAdd(&fill[1], 1u);
    const uint32_t first = base[2] + atomicAdd(&fill[2], p);
    if (uint64_t(k) + p > ob.capE || h >= ob.capH || uint64_t(first) + p > ob.capP) continue;
    ob.hugeLn[h] = uint32_t(i);
    ob.hugeFirst[h] = first;
    const uint64_t end = e - trim;  // (p != 0: the line is longer than its trailing bytes)
    for (uint32_t j = 0; j < p; ++j) {
      const uint64_t from = o + uint64_t(j) * C;
      const uint64_t to = from + C < end ? from + C : end;
      ob.outLn[k + j] = first + j;
      ob.outRec[2 * uint64_t(k + j)] = j ? from - kPieceLead : from;
      ob.outRec[2 * uint64_t(k + j) + 1] = (to + trim) | kEntryPiece | (j ? kEntryLead : 0ull);
    }
  }
}

// One WAVE per line in pieces.  A piece that was entered in the state its predecessor left stands as
// recorded; if that holds for every piece of the line (64 at a time: a megabyte is 1800 pieces,
// and a lane that chained them one dependent load after the other took 0.4 ms) the Outcome comes
// from the last piece with an accept, the last with a "left the initial state" and the last
// piece's exit state.  Otherwise lane 0 chains the records in order and walks a piece whose guess
// was wrong again from its true entry state (the table comes to LDS only if some line of the
// workgroup needs that).  The Outcome is what k_ragged reports for a whole line: the last accept
// (Last) or the final state (Full), the last "left the initial state".
constexpr int kFoldThreads = 256;
constexpr int kFoldLines = kFoldThreads / 64;  // per workgroup and trip

// (tabk: how a piece is walked again - kTabFused: the fused table, staged to LDS; kTabHot: a
// REDGPU_TAB_HOT_ROWS DFA's class table in global memory, d.table as u16 rows of d.nClasses
// entries, byte -> class in d.equivLeader; kTabCls / kTabClsBig: the class table k_ragged<cls>
// stages, at d.table + d.clsOff = 256 bytes of 2 x class per byte, then rows of d.clsRowBytes
// whose entries are the next state (kTabClsBig) or its row offset)
__global__ void __launch_bounds__(kFoldThreads)
k_ragged_pieces_fold(DevDfa d, Batch io, int acc, int wantStart, int tabk) {
  const bool hot = tabk != kTabFused;  // (no LDS table)
  extern __shared__ __align__(16) uint8_t foldTab[];
  const uint32_t nHuge = io.outCtl[2];
  if (nHuge == 0) return;
  const uint64_t C = io.outCtl[4];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  bool staged = false;
  for (uint32_t h0 = blockIdx.x * kFoldLines; h0 < nHuge; h0 += gridDim.x * kFoldLines) {
    const uint32_t h = h0 + wave;
    const bool valid = h < nHuge;
    const uint32_t ln = valid ? io.hugeLn[h] : 0u;
    const uint32_t first = valid ? io.hugeFirst[h] : 0u;
    const uint64_t o = io.offsets[ln];
    const uint64_t raw = io.offsets[ln + 1] - o;
    const uint64_t eff = raw >= io.stride ? raw - io.stride : 0;
    const uint32_t P = valid ? uint32_t((eff + C - 1) / C) : 0u;
    // a piece's record (k_ragged's report): positions count from its first walked byte
    auto walkedFrom = [&](uint32_t jj) -> uint64_t { return jj ? uint64_t(jj) * C - kPieceLead : 0; };
    auto endOf = [](uint64_t w) { return w & 0xffffffffull; };
    auto exitOf = [](uint64_t w) { return uint32_t(w >> 32) & 0xffffu; };
    auto entOf = [](uint64_t w) { return uint32_t(w >> 48) & 0xffffu; };
    bool bad = false;
    uint32_t lastAcc = 0, lastStart = 0;  // 1 + piece
    for (uint32_t j = lane; j < P; j += 64) {
      const uint64_t w = io.pieceEnd[first + j];
      const uint32_t before = j ? exitOf(io.pieceEnd[first + j - 1]) : d.init;
      bad = bad || entOf(w) != before;
      if (io.pieceRes[first + j] < 0) lastAcc = j + 1;
      if (wantStart && io.pieceStart[first + j]) lastStart = j + 1;
    }
    const bool anyBad = __builtin_amdgcn_ballot_w64(bad) != 0;
    for (int o2 = 32; o2; o2 >>= 1) {
      const uint32_t a = __shfl_xor(lastAcc, o2), b2 = __shfl_xor(lastStart, o2);
      lastAcc = a > lastAcc ? a : lastAcc;
      lastStart = b2 > lastStart ? b2 : lastStart;
    }
    uint32_t entry = d.init, accState = 0;
    bool accepted = false;
    uint64_t en = 0, st = 0;
    if (!anyBad && P) {
      if (lastAcc) {
        accepted = true;
        accState = uint32_t(io.pieceRes[first + lastAcc - 1]) & 0x7fffffffu;
        en = walkedFrom(lastAcc - 1) + endOf(io.pieceEnd[first + lastAcc - 1]);
      }
      if (lastStart) st = walkedFrom(lastStart - 1) + io.pieceStart[first + lastStart - 1];
      entry = exitOf(io.pieceEnd[first + P - 1]);
    }
    if (__syncthreads_or(anyBad ? 1 : 0)) {
      if (!staged && !hot) {
        for (uint32_t i = threadIdx.x; i < d.tableBytes / 16; i += kFoldThreads)
          reinterpret_cast<uint4 *>(foldTab)[i] = reinterpret_cast<const uint4 *>(d.table)[i];
        __syncthreads();
      }
      staged = true;
      if (anyBad && lane == 0) {
        const uint16_t *cls = reinterpret_cast<const uint16_t *>(d.table);
        for (uint32_t j = 0; j < P; ++j) {
          const uint64_t w = io.pieceEnd[first + j];
          if (entOf(w) == entry) {
            const int32_t rec = io.pieceRes[first + j];
            if (rec < 0) {
              accepted = true;
              accState = uint32_t(rec) & 0x7fffffffu;
              en = walkedFrom(j) + endOf(w);
            }
            if (wantStart) {
              const uint64_t sv = io.pieceStart[first + j];
              if (sv) st = walkedFrom(j) + sv;
            }
            entry = exitOf(w);
            continue;
          }
          const uint64_t from = uint64_t(j) * C;
          const uint64_t to = from + C < eff ? from + C : eff;
          uint32_t s = entry;
          for (uint64_t i = from; i < to; ++i) {
            const uint32_t was = s;
            const uint32_t byte = io.data[o + i];
            if (tabk == kTabFused) {
              s = foldTab[(s << 8) | byte];
            } else if (tabk == kTabHot) {
              s = cls[size_t(s) * d.nClasses + d.equivLeader[byte]];
            } else {
              const uint8_t *cb = d.table + d.clsOff;
              const uint32_t v = *reinterpret_cast<const uint16_t *>(
                  cb + 256 + size_t(s) * d.clsRowBytes + cb[byte]);
              s = tabk == kTabClsBig ? v : v / d.clsRowBytes;
            }
            if (was == d.init && s != was) st = i;
            if (s >= d.firstAccept) { accepted = true; accState = s; en = i + 1; }
          }
          entry = s;
        }
      }
    }
    if (!valid || lane != 0) continue;
    int32_t rr;
    if (acc) {
      rr = accepted ? d.result[accState] : 0;
    } else {
      rr = entry >= d.firstAccept && entry < d.nStates ? d.result[entry] : 0;
      en = eff;
    }
    io.result[ln] = rr;
    if (io.end) io.end[ln] = rr ? en : 0;
    if (wantStart && io.start) io.start[ln] = rr ? st : 0;
  }
}

// the smallest batch whose long lines are listed.  The list costs a small batch ~4 us (4096
// lines of 32-256 B: 34.6 -> 38.2 us per launch) and is what keeps ONE huge line from holding
// it for milliseconds (the same batches with a 1 MB line: 67.5 ms -> 0.17-0.36 ms;
// scripts/lab/run_small.sh).
constexpr uint64_t kLongFirstMinLines = 1024;

// X of the rule above; REDGPU_RAGGED_LONG_X overrides it (lab; 0 = no list, >= 2 otherwise)
inline uint32_t raggedLongFactor() {
  static const uint32_t x = [] {
    const char *e = getenv("REDGPU_RAGGED_LONG_X");
    const int v = e ? atoi(e) : 4;
    return uint32_t(v <= 0 ? 0 : v < 2 ? 2 : v > 64 ? 64 : v);
  }();
  return x;
}
