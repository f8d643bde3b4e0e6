// k_scan_marked.h - k_scan_marked<KIND, THREADS, VERB>: scan / search over anchored DFAs in two passes - mark
// candidate start positions by their first bytes, then visit only those
// (included by kernels.hip inside namespace redgpu { namespace { ... } }; see its file map).
#pragma once

// =========================================================================================
// k_scan_marked: scan (include/Matcher.h:498-554) in two passes per batch of lines.
//
// scanLane gives a lane a line and lets it step over the start positions; on text nearly all of
// them are rejected from the byte in hand, but the rejecting is done 64 lines wide, a few
// instructions per position, with every lane's candidate dragging the wave through the slow
// path.  Here a workgroup takes up to kThreads consecutive lines at a time - a contiguous
// piece of the input buffer - and
//   1. MARKS: all threads sweep that piece 16 bytes per lane (coalesced, every byte read once),
//      test each byte against the DFA's start bytes (the <= 4 bytes at which an attempt can
//      survive its first step; with the leader: the bytes of the leader's first class) and
//      leave one bit per position in LDS;
//   2. VISITS: lane t takes line t and calls ScanWalk::visit() - the reference's loop body,
//      the same code scanLane runs - for the marked positions of its line only, in order.
// An unmarked position changes nothing but "result = 0" (without the leader) or nothing at all
// (with it), which ScanWalk::skipped() stands for; a marked one that the partly matched
// leader of an earlier attempt consumed is recognised by visit() itself (resume).
// A line longer than the bitmap covers (64 KB) is scanned by one lane the old way.
// =========================================================================================
// 256 lines per batch, one bit per input byte in LDS (64 KB of input): small workgroups, several
// per CU - a batch is a chain of dependent steps (offsets, marks, barrier, visits, barrier) and
// only other workgroups can fill its gaps (1024-thread workgroups, one or two per CU: 116 us
// for the batch that now takes ~half)
constexpr uint32_t kMarkBytes = 8192;
constexpr int kScanThreads = 256;

// scan / search through k_scan_marked: the DFA's start bytes are few - up to 4 as a packed list
// tested a word at a time, up to 64 (a leading character class) through the flag table
inline bool scanMarkable(const DevDfa &d, int lead) {
  const uint32_t listed = lead ? d.startLeadCount : d.startFreeCount;
  if (listed >= 1 && listed <= 4) return true;
  const uint32_t total = d.startTotal[lead ? 1 : 0];
  return listed > 4 && total >= 1 && total <= 64;
}

// one bit per byte of `word` that can start a surviving attempt: walkBytesPeek's test - a start
// byte, followed (n2 != 0) by a byte that may follow one or, with the leader, by another start
// byte (StartFilter::consumes)
__device__ __forceinline__ uint32_t markNibble(uint32_t word, uint32_t nextWord, const StartFilter &f) {
  uint32_t m = wordMatchMask(word, f.set1, f.n1);
  if (m && f.n2) {
    const uint32_t follow = (word >> 8) | (nextWord << 24);
    uint32_t ok = wordMatchMask(follow, f.set2, f.n2);
    if (f.consumes) ok |= wordMatchMask(follow, f.set1, f.n1);
    m &= ok;
  }
  return ((m >> 7) & 1u) | ((m >> 14) & 2u) | ((m >> 21) & 4u) | ((m >> 28) & 8u);
}

// the same test against the full flag table (DfaImage::startFlags, 256 bytes in LDS): any number
// of start bytes - a pattern that begins with a character class.  bit 0 = start byte, bit 1 =
// may follow one; useFollow = the follower is in hand and the second filter means something.
__device__ __forceinline__ uint32_t markNibbleTbl(uint32_t word, uint32_t nextWord, const uint8_t *tbl,
                                                  bool useFollow, bool consumes) {
  const uint32_t f0 = tbl[word & 0xffu], f1 = tbl[(word >> 8) & 0xffu], f2 = tbl[(word >> 16) & 0xffu],
                 f3 = tbl[word >> 24], f4 = tbl[nextWord & 0xffu];
  const uint32_t starts = (f0 & 1u) | ((f1 & 1u) << 1) | ((f2 & 1u) << 2) | ((f3 & 1u) << 3);
  if (!useFollow) return starts;
  const uint32_t pass = consumes ? 3u : 2u;  // with the leader a start byte may follow too
  const uint32_t ok = ((f1 & pass) ? 1u : 0u) | ((f2 & pass) ? 2u : 0u) | ((f3 & pass) ? 4u : 0u) |
                      ((f4 & pass) ? 8u : 0u);
  return starts & ok;
}

template <int KIND, int kThreads, int VERB>
__global__ void __launch_bounds__(kThreads)
k_scan_marked(DevDfa d, Batch b, int style, int lead) {
  constexpr bool kSearchVerb = VERB == kSearch;
  extern __shared__ __align__(16) uint8_t lds[];
  uint8_t *eq = lds;
  uint8_t *leader = lds + 256;
  const Tab<KIND> tab = stageTab<KIND, kThreads>(d, lds);
  uint16_t *marks16 = reinterpret_cast<uint16_t *>(lds + 512 + ((ldsTableBytes<KIND>(d) + 15) & ~size_t(15)));
  const uint32_t *marks32 = reinterpret_cast<const uint32_t *>(marks16);
  // behind the bitmap: the batch's candidate list and per-line slots of the spread form below
  uint32_t *cand = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(marks16) + kMarkBytes);
  uint32_t *best = cand + kThreads;          // per line: lowest position whose attempt succeeded
  uint32_t *lineLen = best + kThreads;
  uint32_t *ran = lineLen + kThreads;        // per line: some attempt got past the leader
  uint64_t *lineOff = reinterpret_cast<uint64_t *>(ran + kThreads);
  uint32_t *candK = reinterpret_cast<uint32_t *>(lineOff + kThreads);  // scan with the leader
  uint32_t *lineFirst = candK + kThreads;   // where a line's candidates start in the list
  uint32_t *waveTot = lineFirst + kThreads;  // candidates per wave (block-wide prefix sum)
  // more than 4 start bytes: the full flag table instead of the packed list (stageTab's barrier
  // is behind us; the first use is behind the next one)
  uint8_t *flagTbl = reinterpret_cast<uint8_t *>(waveTot + 8);
  const bool useTbl = (lead ? d.startLeadCount : d.startFreeCount) > 4;
  if (useTbl)
    for (uint32_t i = threadIdx.x; i < 64; i += kThreads)
      reinterpret_cast<uint32_t *>(flagTbl)[i] =
          reinterpret_cast<const uint32_t *>(d.equivLeader + (lead ? 768 : 512))[i];
  const bool tblFollow = d.startFollow[lead ? 1 : 0] != 0;
  if (useTbl) __syncthreads();
  LaneCtx c{eq, leader, resOf<KIND>(d, lds), d.init, d.leaderNext, d.nPureDead, d.firstAccept,
            d.leaderLen};
  // The batch's candidates are SPREAD over the threads, one each, instead of every lane visiting
  // its own line's one after the other: a wave then runs visit() once, not once per candidate
  // slot of its 64 lines.  Attempts at different positions do not depend on one another except
  // in scan with the leader, where a partly matched leader consumes positions
  // (Matcher.h:511-518): there the leader prefix length at every candidate is found in
  // parallel first, each line then walks the "consumed" chain over its own candidates (no
  // memory touched), and only the candidates still standing run their attempts.
  const bool quirk = !kSearchVerb && lead != 0;
  const int32_t initRes = c.resultOf(c.init);
  const uint32_t wave = threadIdx.x >> 6, laneId = threadIdx.x & 63u;
  const uint32_t n2 = lead ? d.start2LeadCount : d.start2FreeCount;
  const StartFilter flt{lead ? d.startLeadWord : d.startFreeWord,
                        lead ? d.startLeadCount : d.startFreeCount,  // 1..4 (launchGeneric)
                        lead ? d.start2LeadWord : d.start2FreeWord, n2 <= 4 ? n2 : 0u,
                        !kSearchVerb && lead != 0};  // search only peeks at the leader (lookingAt)
  const uint64_t dataAddr = reinterpret_cast<uint64_t>(b.data);
  auto lineStart = [&](uint64_t line) -> uint64_t {
    return b.offsets ? b.offsets[line] : line * b.stride;
  };
  const uint64_t lo = b.n * blockIdx.x / gridDim.x, hi = b.n * (blockIdx.x + 1) / gridDim.x;
  for (uint64_t a = lo; a < hi;) {
    // as many lines as the bitmap covers (workgroup-uniform)
    uint64_t cnt = hi - a < uint64_t(kThreads) ? hi - a : uint64_t(kThreads);
    const uint64_t first = lineStart(a);
    const uint64_t baseAddr = (dataAddr + first) & ~15ull;
    uint64_t last = lineStart(a + cnt);
    while (cnt > 1 && dataAddr + last - baseAddr > uint64_t(kMarkBytes) * 8) {
      cnt >>= 1;
      last = lineStart(a + cnt);
    }
    const bool tooLong = dataAddr + last - baseAddr > uint64_t(kMarkBytes) * 8;  // cnt == 1
    // this lane's line (requested now, used after the marking)
    const uint64_t line = a + (threadIdx.x < cnt ? threadIdx.x : 0);
    const uint64_t o = lineStart(line);
    const uint64_t oEnd = b.offsets ? b.offsets[line + 1] : o + b.stride;
    if (!tooLong) {
      const uint64_t pieces = (dataAddr + last - baseAddr + 15) >> 4;
      // four pieces per thread and trip, requested together
      for (uint64_t k0 = threadIdx.x; k0 < pieces; k0 += 4ull * kThreads) {
        uint4 v[4];
        uint32_t after[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint64_t k = k0 + uint64_t(j) * kThreads;
          // (the first and the last piece may reach up to 15 bytes outside the buffer - inside a
          // 16-byte granule that holds valid bytes; those bits are never looked at)
          const uint64_t kk = k < pieces ? k : pieces - 1;
          v[j] = *reinterpret_cast<const uint4 *>(baseAddr + 16 * kk);
          // the follower of the piece's last byte: the next piece's first (it exists - the walk
          // of a line's LAST position is the same with or without a mark - except behind the
          // last piece, where a byte nothing may follow keeps every start byte marked)
          after[j] = *reinterpret_cast<const uint32_t *>(baseAddr + 16 * (kk + 1 < pieces ? kk + 1 : kk));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint64_t k = k0 + uint64_t(j) * kThreads;
          if (k >= pieces) break;
          if (useTbl) {
            marks16[k] = uint16_t(markNibbleTbl(v[j].x, v[j].y, flagTbl, tblFollow, flt.consumes) |
                                  (markNibbleTbl(v[j].y, v[j].z, flagTbl, tblFollow, flt.consumes) << 4) |
                                  (markNibbleTbl(v[j].z, v[j].w, flagTbl, tblFollow, flt.consumes) << 8) |
                                  (markNibbleTbl(v[j].w, after[j], flagTbl, tblFollow && k + 1 < pieces,
                                                 flt.consumes) << 12));
            continue;
          }
          StartFilter f = flt;
          if (k + 1 >= pieces) f.n2 = 0;
          marks16[k] = uint16_t(markNibble(v[j].x, v[j].y, flt) | (markNibble(v[j].y, v[j].z, flt) << 4) |
                                (markNibble(v[j].z, v[j].w, flt) << 8) |
                                (markNibble(v[j].w, after[j], f) << 12));
        }
      }
    }
    __syncthreads();
    const uint64_t n = b.offsets ? (oEnd - o >= b.stride ? oEnd - o - b.stride : 0)  // stride =
                                 : b.stride;                      // trailing bytes to drop (ragged)
    bool spreadDone = false;
    if (!tooLong) {
      // A: every line counts its marked positions; a block-wide prefix sum gives each line its
      // place in the batch's candidate list (position order within a line)
      const uint64_t bit0 = dataAddr + o - baseAddr;
      auto lineWord = [&](uint64_t wd) -> uint32_t {
        uint32_t m = marks32[wd];
        const uint64_t wordBit = wd << 5;
        if (wordBit < bit0) m &= ~0u << uint32_t(bit0 - wordBit);
        if (wordBit + 32 > bit0 + n) m &= ~0u >> uint32_t(wordBit + 32 - (bit0 + n));
        return m;
      };
      uint32_t mine = 0;
      if (threadIdx.x < cnt)
        for (uint64_t wd = bit0 >> 5; (wd << 5) < bit0 + n; ++wd) mine += uint32_t(__builtin_popcount(lineWord(wd)));
      uint32_t incl = mine;
#pragma unroll
      for (int sh = 1; sh < 64; sh <<= 1) {
        const uint32_t up = uint32_t(__shfl_up(int(incl), sh, 64));
        if (laneId >= uint32_t(sh)) incl += up;
      }
      if (laneId == 63) waveTot[wave] = incl;
      __syncthreads();
      uint32_t before = 0, total = 0;
      for (uint32_t wv = 0; wv < uint32_t(kThreads / 64); ++wv) {
        if (wv < wave) before += waveTot[wv];
        total += waveTot[wv];
      }
      if (total <= uint32_t(kThreads)) {  // else: the lines visit their own (below)
        spreadDone = true;
        const uint32_t firstAt = before + incl - mine;
        if (threadIdx.x < cnt) {
          lineOff[threadIdx.x] = o;
          lineLen[threadIdx.x] = uint32_t(n);
          lineFirst[threadIdx.x] = firstAt;
          best[threadIdx.x] = 0xffffffffu;
          ran[threadIdx.x] = 0;
          uint32_t at = firstAt;
          for (uint64_t wd = bit0 >> 5; (wd << 5) < bit0 + n; ++wd) {
            uint32_t m = lineWord(wd);
            while (m) {
              const uint32_t k = uint32_t(__builtin_ctz(m));
              m &= m - 1;
              cand[at++] = (threadIdx.x << 16) | uint32_t((wd << 5) + k - bit0);
            }
          }
        }
        __syncthreads();
        // B: one candidate per thread
        uint32_t li = 0, at = 0;
        const uint8_t *q = b.data;
        uint64_t qn = 0;
        bool go = threadIdx.x < total;
        if (go) {
          const uint32_t entry = cand[threadIdx.x];
          li = entry >> 16;
          at = entry & 0xffffu;
          q = b.data + lineOff[li];
          qn = lineLen[li];
        }
        if (quirk) {
          // B1: how much of the leader matches here; B2: which candidates an earlier one's
          // partly (or wholly) matched leader has consumed
          if (go) {
            uint32_t kk = 0;
            while (kk < c.leaderLen && at + kk < qn && c.leader[kk] == c.eq[q[at + kk]]) ++kk;
            candK[threadIdx.x] = kk;
          }
          __syncthreads();
          if (threadIdx.x < cnt) {
            uint64_t resume = 0;
            for (uint32_t r = 0; r < mine; ++r) {
              const uint32_t ci = lineFirst[threadIdx.x] + r;
              const uint64_t i = cand[ci] & 0xffffu;
              const uint32_t kk = candK[ci];
              if (i < resume) { candK[ci] = 0xffffffffu; continue; }   // consumed: never visited
              resume = i + kk + 1;  // on the mismatching byte (or past the leader), then ++in
              if (kk != c.leaderLen) candK[ci] = 0xffffffffu;         // visited, no attempt
            }
          }
          __syncthreads();
          if (go && candK[threadIdx.x] == 0xffffffffu) go = false;
        }
        bool found = false;
        int32_t fr = 0;
        uint64_t fs = 0, fe = 0;
        if (go) {
          typename std::conditional<kSearchVerb, SearchWalk<Tab<KIND>>, ScanWalk<Tab<KIND>>>::type
              w(tab, c, q, qn, style, lead != 0);
          w.skipped();
          found = !w.visit(q[at], at, at + 1 < qn ? uint32_t(q[at + 1]) : kNoPeek);
          if constexpr (kSearchVerb) {
            fr = w.result; fs = w.matchStart; fe = w.matchEnd;
          } else {
            fr = w.ret;
          }
          if (found) atomicMin(&best[li], at);
          else if (w.result != initRes) ran[li] = 1;
        }
        __syncthreads();
        // C: the winner of each line reports; lines without one report "no match"
        if (found && best[li] == at) {
          b.result[a + li] = fr;
          if (kSearchVerb) {
            if (b.start) b.start[a + li] = fr != 0 ? fs : 0;
            if (b.end) b.end[a + li] = fr != 0 ? fe : 0;
          }
        }
        if (threadIdx.x < cnt && best[threadIdx.x] == 0xffffffffu) {
          // what the sequential walk is left with: the initial state's result when no attempt
          // ran (no byte, or - with the leader - no position past it), else 0
          const int32_t r = n == 0 ? initRes : !lead ? 0 : ran[threadIdx.x] ? 0 : initRes;
          b.result[line] = r;
          if (kSearchVerb) {
            if (b.start) b.start[line] = 0;
            if (b.end) b.end[line] = 0;
          }
        }
      }
    }
    if (!spreadDone && threadIdx.x < cnt) {
      const uint8_t *p = b.data + o;
      int32_t r;
      uint64_t st = 0, en = 0;
      if (tooLong) {
        if (kSearchVerb) r = searchLane(tab, c, p, n, style, lead != 0, st, en);
        else r = scanLane(tab, c, p, n, style, lead != 0);
      } else {
        typename std::conditional<kSearchVerb, SearchWalk<Tab<KIND>>, ScanWalk<Tab<KIND>>>::type
            w(tab, c, p, n, style, lead != 0);
        const uint64_t bit0 = dataAddr + o - baseAddr;  // this line's first bit
        bool going = true;
        for (uint64_t wd = bit0 >> 5; going && (wd << 5) < bit0 + n; ++wd) {
          uint32_t m = marks32[wd];
          const uint64_t wordBit = wd << 5;
          if (wordBit < bit0) m &= ~0u << uint32_t(bit0 - wordBit);
          if (wordBit + 32 > bit0 + n) m &= ~0u >> uint32_t(wordBit + 32 - (bit0 + n));
          while (m) {
            const uint32_t k = uint32_t(__builtin_ctz(m));
            m &= m - 1;
            const uint64_t i = wordBit + k - bit0;
            w.skipped();  // harmless when nothing was: a visit that does not return leaves 0
            if (!w.visit(p[i], i, i + 1 < n ? uint32_t(p[i + 1]) : kNoPeek)) { going = false; break; }
          }
        }
        if (going && n) w.skipped();
        if constexpr (kSearchVerb) {
          r = w.result;
          if (r != 0) { st = w.matchStart; en = w.matchEnd; }
        } else {
          r = w.value();
        }
      }
      b.result[line] = r;
      if (kSearchVerb) {
        if (b.start) b.start[line] = st;
        if (b.end) b.end[line] = en;
      }
    }
    __syncthreads();
    a += cnt;
  }
}
