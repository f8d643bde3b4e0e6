// k_lists.h - the verbs that emit a record list per line: k_collect (Red::collect), k_matchall and
// k_matchall_blocks (matchAllCore, include/Matcher.h:711-766)
// (included by kernels.hip inside namespace redgpu { namespace { ... } }; see its file map).
#pragma once

// Red::collect (lib/Red.cpp:103-116): all non-overlapping matches of a line, in order, by
// repeated search<styLast,false> from the end of the previous match.  One line per lane.
template <int KIND, int kThreads>
__global__ void __launch_bounds__(kThreads)
k_collect(DevDfa d, Batch b, uint64_t cap, uint64_t *counts) {
  extern __shared__ __align__(16) uint8_t lds[];
  uint8_t *eq = lds;
  uint8_t *leader = lds + 256;
  const Tab<KIND> tab = stageTab<KIND, kThreads>(d, lds);
  LaneCtx c{eq, leader, resOf<KIND>(d, lds), d.init, d.leaderNext, d.nPureDead, d.firstAccept,
            d.leaderLen};
  c.startWord[0] = d.startFreeWord; c.startCount[0] = d.startFreeCount;
  c.startWord[1] = d.startLeadWord; c.startCount[1] = d.startLeadCount;
  c.start2Word[0] = d.start2FreeWord; c.start2Count[0] = d.start2FreeCount;
  c.start2Word[1] = d.start2LeadWord; c.start2Count[1] = d.start2LeadCount;
  c.suffixClosed = d.suffixClosed;
  const uint64_t step = uint64_t(gridDim.x) * kThreads;
  for (uint64_t line = uint64_t(blockIdx.x) * kThreads + threadIdx.x; line < b.n; line += step) {
    const uint8_t *p;
    uint64_t n;
    if (b.offsets) {
      const uint64_t o = b.offsets[line];
      p = b.data + o;
      n = b.offsets[line + 1] - o;
      n = n >= b.stride ? n - b.stride : 0;  // stride = trailing bytes to drop (ragged)
    } else {
      p = b.data + line * b.stride;
      n = b.stride;
    }
    const StartFilter flt{c.startWord[0], c.startCount[0] <= 4 ? c.startCount[0] : 0u,
                          c.start2Word[0], c.start2Count[0] <= 4 ? c.start2Count[0] : 0u, false};
    uint64_t found = 0, pos = 0;
    while (pos < n) {
      // search<styLast,false> from pos (Matcher.h:557-640), lean: an attempt carries the state,
      // the last accepting state, its end and the last "left the initial state" position; the
      // result table is read once per match.  Attempts that outlive a few bytes go on in
      // 16-byte requests (a dense DFA's attempt runs to the end of the line).
      bool got = false;
      uint32_t accS = 0;
      uint64_t mS = 0, mE = 0;
      walkBytesPeek(p, pos, n, flt, [] {}, [&](uint32_t byte, uint64_t i, uint32_t nextByte) -> bool {
        uint32_t st = tab.next(c.init, byte);
        bool any = false;
        uint64_t ms = i, me = i;
        uint32_t aS = 0;
        if (st >= c.firstAccept) { aS = st; me = i + 1; any = true; }
        else if (st < c.nPureDead) return true;
        else if (nextByte != kNoPeek && tab.next(st, nextByte) < c.nPureDead) return true;
        auto stepOne = [&](uint32_t b2, uint64_t q) -> bool {
          const uint32_t was = st;
          st = tab.next(st, b2);
          if (was == c.init && st != was) ms = q;
          const bool acc = st >= c.firstAccept;
          if (acc) { aS = st; me = q + 1; any = true; }
          return acc || st >= c.nPureDead;
        };
        uint64_t q = i + 1;
        bool alive = true;
        for (uint32_t k = 0; k < 6 && q < n && alive; ++k, ++q) alive = stepOne(uint32_t(p[q]), q);
        if (alive) {
          // (walkBytes stops when stepOne says so: alive = the walk reached the end of the line)
          walkBytes(p, q, n, [&](uint32_t b2, uint64_t q2) -> bool { return alive = stepOne(b2, q2); });
        }
        if (!any) return !(c.suffixClosed && alive);  // L = SIGMA* L: no later start can match either
        got = true; accS = aS; mS = ms; mE = me;
        return false;
      });
      if (!got) break;
      if (found < cap) {
        b.result[line * cap + found] = c.res[accS];
        if (b.start) b.start[line * cap + found] = mS;
        if (b.end) b.end[line * cap + found] = mE;
      }
      ++found;
      pos = mE;
    }
    counts[line] = found;
  }
}

// matchAllCore (include/Matcher.h:711-766; public entry matchAll, lib/Matcher.cpp:97-102, which
// instantiates <styTangent, doLeader = true>): ONE anchored walk that reports every maximal run
// of bytes over which the accepted result stays the same - a la RE2::Set::Match.  A run's end_
// grows while the same result repeats (:747-748); a different positive result opens a new
// record (:749-752); a non-accepting byte resets the run (:757) and a pure dead end stops the
// walk (:755-756).  The record being extended keeps its end in a register and is flushed when
// the run closes, instead of re-storing it per byte.
template <class T, bool NOEXIT = false>
__device__ uint64_t matchAllLane(const T &tab, const LaneCtx &c, const uint8_t *p, uint64_t n,
                                 bool lead, uint64_t cap, int32_t *res, uint64_t *st,
                                 uint64_t *en) {
  if (lead && !lookingAt(c, p, 0, n)) return 0;
  uint32_t s = c.init;
  int32_t prev = 0;
  uint64_t matchStart = 0, found = 0, curEnd = 0;
  auto step = [&](uint32_t byte, uint64_t idx) -> bool {
    const uint32_t was = s;
    s = tab.next(s, byte);
    if (was == c.init && s != was) matchStart = idx;
    if (s >= c.firstAccept) {
      const int32_t r = c.res[s];
      if (r != prev) {
        if (found && found - 1 < cap && en) en[found - 1] = curEnd;
        prev = r;
        if (found < cap) {
          res[found] = r;
          if (st) st[found] = matchStart;
        }
        ++found;
      }
      curEnd = idx + 1;
    } else {
      if (!NOEXIT && s < c.nPureDead) return false;
      prev = 0;
    }
    return true;
  };
  if constexpr (NOEXIT)
    walkAllBytes(p, n, [&](uint32_t byte, uint64_t idx) { (void)step(byte, idx); });
  else
    walkBytes(p, 0, n, step);
  if (found && found - 1 < cap && en) en[found - 1] = curEnd;
  return found;
}

// (Keeping the first four records in registers and storing them once at the end of the line was
// tried for cap <= 4: the four-way selects per accepting byte cost more than the scattered stores
// they replace - SYN-256 2^20 x 64 B 541 -> 355 GB/s.)
template <int KIND, int kThreads>
__global__ void __launch_bounds__(kThreads)
k_matchall(DevDfa d, Batch b, uint64_t cap, uint64_t *counts, int lead) {
  extern __shared__ __align__(16) uint8_t lds[];
  uint8_t *eq = lds;
  uint8_t *leader = lds + 256;
  const Tab<KIND> tab = stageTab<KIND, kThreads>(d, lds);
  LaneCtx c{eq, leader, resOf<KIND>(d, lds), d.init, d.leaderNext, d.nPureDead, d.firstAccept,
            d.leaderLen};
  c.startWord[0] = d.startFreeWord; c.startCount[0] = d.startFreeCount;
  c.startWord[1] = d.startLeadWord; c.startCount[1] = d.startLeadCount;
  c.start2Word[0] = d.start2FreeWord; c.start2Count[0] = d.start2FreeCount;
  c.start2Word[1] = d.start2LeadWord; c.start2Count[1] = d.start2LeadCount;
  c.suffixClosed = d.suffixClosed;
  const uint64_t step = uint64_t(gridDim.x) * kThreads;
  for (uint64_t line = uint64_t(blockIdx.x) * kThreads + threadIdx.x; line < b.n; line += step) {
    const uint8_t *p;
    uint64_t n;
    if (b.offsets) {
      const uint64_t o = b.offsets[line];
      p = b.data + o;
      n = b.offsets[line + 1] - o;
      n = n >= b.stride ? n - b.stride : 0;  // stride = trailing bytes to drop (ragged)
    } else {
      p = b.data + line * b.stride;
      n = b.stride;
    }
    // pure dead ends that are absorbing: the straight-line walk (nothing can happen past one)
    counts[line] = d.deadAbsorbing
                       ? matchAllLane<Tab<KIND>, true>(tab, c, p, n, lead != 0, cap, b.result + line * cap,
                                                       b.start ? b.start + line * cap : nullptr,
                                                       b.end ? b.end + line * cap : nullptr)
                       : matchAllLane(tab, c, p, n, lead != 0, cap, b.result + line * cap,
                                      b.start ? b.start + line * cap : nullptr,
                                      b.end ? b.end + line * cap : nullptr);
  }
}

// One byte of k_matchall_blocks' walk over a fused u8 table at LDS offset 512 (the address is
// (state << 8) | byte, formed by v_perm_b32), as ONE asm statement so the lookup's round trip
// is covered by the bookkeeping of the state in hand - the state BEFORE this byte, i.e. the
// masks and the packed word run one position behind: "accepting" and "is the initial state"
// are shifted into accR / iniR by add-with-carry (first position = highest bit), the state
// into `packed` from the top (first state = lowest byte).  The two compares write SGPR pairs
// that the add-with-carrys read three instructions later (gfx950 wants two wait states
// between a VALU writing an SGPR and a VALU reading it).
template <bool BOOK>
__device__ __forceinline__ void mabStep(uint32_t &s, uint32_t w, uint32_t sel, uint32_t &accR,
                                        uint32_t &iniR, uint32_t &packed, uint32_t T,
                                        uint32_t init) {
  uint32_t a, t;
  uint64_t m, i2, junk;
  if constexpr (BOOK) {
    asm volatile("v_perm_b32 %[a], %[s], %[w], %[sel]\n\t"
                 "ds_read_u8 %[t], %[a] offset:512\n\t"
                 "v_cmp_le_u32_e64 %[m], %[T], %[s]\n\t"
                 "v_cmp_eq_u32_e64 %[i], %[init], %[s]\n\t"
                 "v_alignbit_b32 %[p], %[s], %[p], 8\n\t"
                 "v_addc_co_u32_e64 %[acc], %[j], %[acc], %[acc], %[m]\n\t"
                 "v_addc_co_u32_e64 %[ini], %[j], %[ini], %[ini], %[i]\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : [a] "=&v"(a), [t] "=&v"(t), [m] "=&s"(m), [i] "=&s"(i2), [j] "=&s"(junk),
                   [p] "+v"(packed), [acc] "+v"(accR), [ini] "+v"(iniR)
                 : [s] "v"(s), [w] "v"(w), [sel] "s"(sel), [T] "s"(T), [init] "s"(init)
                 : "memory");
  } else {
    asm volatile("v_perm_b32 %[a], %[s], %[w], %[sel]\n\t"
                 "ds_read_u8 %[t], %[a] offset:512\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : [a] "=&v"(a), [t] "=&v"(t)
                 : [s] "v"(s), [w] "v"(w), [sel] "s"(sel)
                 : "memory");
  }
  s = t;
}

// the bookkeeping alone, for the state behind the block's last byte
__device__ __forceinline__ void mabBook(uint32_t s, uint32_t &accR, uint32_t &iniR,
                                        uint32_t &packed, uint32_t T, uint32_t init) {
  uint64_t m, i2, junk;
  asm volatile("v_cmp_le_u32_e64 %[m], %[T], %[s]\n\t"
               "v_cmp_eq_u32_e64 %[i], %[init], %[s]\n\t"
               "v_alignbit_b32 %[p], %[s], %[p], 8\n\t"
               "v_addc_co_u32_e64 %[acc], %[j], %[acc], %[acc], %[m]\n\t"
               "v_addc_co_u32_e64 %[ini], %[j], %[ini], %[ini], %[i]"
               : [m] "=&s"(m), [i] "=&s"(i2), [j] "=&s"(junk), [p] "+v"(packed), [acc] "+v"(accR),
                 [ini] "+v"(iniR)
               : [s] "v"(s), [T] "s"(T), [init] "s"(init));
}

// Phase A of the block kernels (k_matchall_blocks, k_style_blocks): walks up to kPos = 64 / W
// positions of a line from p (rem = bytes left in the line), straight-line, and leaves behind
//   acc / ini : one bit per position - the state after it is accepting / is the initial state;
//   stage     : the states themselves, W bytes each, lane-interleaved in LDS
//               (word w of lane t at stage[w * THREADS + t]: no bank conflicts);
//   s         : the state after the last position walked.
// `safe` = bytes that may be read from p on (to the end of the batch's buffer).
// Returns the number of positions walked (kPos, or all that was left of the line).
template <int KIND, int THREADS, int W>
__device__ __forceinline__ uint32_t mabWalkBlock(const Tab<KIND> &tab, const LaneCtx &c,
                                                 const uint8_t *p, uint64_t rem, uint64_t safe,
                                                 uint32_t &s, uint32_t *stage, bool tableAt512,
                                                 uint64_t &acc, uint64_t &ini) {
  constexpr uint32_t kPos = 64 / W;
  constexpr uint32_t kPerWord = 4 / W;
  const uint32_t nq = rem >= kPos ? kPos / 16 : uint32_t(rem >> 4);  // whole 16-byte pieces
  uint4 piece[kPos / 16];
#pragma unroll
  for (uint32_t q = 0; q < kPos / 16; ++q)
    piece[q] = q < nq ? *reinterpret_cast<const uint4 *>(p + 16 * q) : make_uint4(0, 0, 0, 0);
  bool walked = false;
  if constexpr (KIND == REDGPU_TAB_LDS_FUSED_U8 && W == 1) {
    // a line's LAST block, shorter than 64 bytes, takes the same 64 straight-line steps when the
    // buffer has the bytes (they belong to the next line): the masks are cut to the line's
    // positions afterwards and the state is read back from the staged ones
    // (from 44 bytes up: 64 steps of 7 instructions against `rem` steps of 13)
    const bool whole = nq < kPos / 16 && rem >= 44 && safe >= 64;
    if (whole && tableAt512) {
#pragma unroll
      for (uint32_t q = 0; q < kPos / 16; ++q) piece[q] = *reinterpret_cast<const uint4 *>(p + 16 * q);
    }
    if ((nq == kPos / 16 || whole) && tableAt512) {
      // a whole block over the fused table: mabStep, masks first-position-high, two halves
      uint32_t aR[2] = {0, 0}, iR[2] = {0, 0}, packed = 0;
#pragma unroll
      for (uint32_t pos = 0; pos < 64; ++pos) {
        const uint4 &pc = piece[pos >> 4];
        const uint32_t word = (pos >> 2) % 4 == 0 ? pc.x : (pos >> 2) % 4 == 1 ? pc.y
                              : (pos >> 2) % 4 == 2 ? pc.z : pc.w;
        const uint32_t sel = 0x0c0c0400u + (pos & 3u);
        // the bookkeeping inside step `pos` is for position pos - 1
        if (pos == 0) mabStep<false>(s, word, sel, aR[0], iR[0], packed, c.firstAccept, c.init);
        else mabStep<true>(s, word, sel, aR[(pos - 1) >> 5], iR[(pos - 1) >> 5], packed,
                           c.firstAccept, c.init);
        if (pos && pos % 4 == 0) stage[(pos / 4 - 1) * THREADS + threadIdx.x] = packed;
      }
      mabBook(s, aR[1], iR[1], packed, c.firstAccept, c.init);
      stage[15 * THREADS + threadIdx.x] = packed;
      acc = (uint64_t(__builtin_bitreverse32(aR[1])) << 32) | __builtin_bitreverse32(aR[0]);
      ini = (uint64_t(__builtin_bitreverse32(iR[1])) << 32) | __builtin_bitreverse32(iR[0]);
      walked = true;
      if (rem < kPos) {  // cut back to the line
        const uint32_t cntv = uint32_t(rem);
        const uint64_t valid = (1ull << cntv) - 1;
        acc &= valid;
        ini &= valid;
        s = reinterpret_cast<const uint8_t *>(stage)[(((cntv - 1) / 4) * THREADS + threadIdx.x) * 4 +
                                                     (cntv - 1) % 4];
        return cntv;
      }
    }
  }
#pragma unroll
  for (uint32_t q = 0; q < kPos / 16; ++q) {
    if (!walked && q < nq) {
      const uint32_t words[4] = {piece[q].x, piece[q].y, piece[q].z, piece[q].w};
      uint32_t packed = 0;
#pragma unroll
      for (uint32_t k = 0; k < 16; ++k) {
        const uint32_t pos = 16 * q + k;
        s = tab.next(s, (words[k >> 2] >> (8 * (k & 3))) & 0xffu);
        acc |= s >= c.firstAccept ? 1ull << pos : 0ull;
        ini |= s == c.init ? 1ull << pos : 0ull;
        packed |= s << (8 * W * (pos % kPerWord));
        if (pos % kPerWord == kPerWord - 1) {
          stage[(pos / kPerWord) * THREADS + threadIdx.x] = packed;
          packed = 0;
        }
      }
    }
  }
  uint32_t cnt = 16 * nq;
  if (cnt < kPos && cnt < rem && safe >= uint64_t(cnt) + 16) {
    // the last < 16 bytes of the line, from one more 16-byte request (it reaches into the next
    // line, never past the buffer: `safe`) - a byte load per step is a memory round trip per step
    const uint32_t left = uint32_t(rem) - cnt;
    const uint4 v = *reinterpret_cast<const uint4 *>(p + cnt);
    const uint32_t words[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (uint32_t k = 0; k < 15; ++k) {
      if (k < left) {
        const uint32_t pos = cnt + k;
        s = tab.next(s, (words[k >> 2] >> (8 * (k & 3))) & 0xffu);
        acc |= uint64_t(s >= c.firstAccept) << pos;
        ini |= uint64_t(s == c.init) << pos;
        uint8_t *slot = reinterpret_cast<uint8_t *>(stage) +
                        (((pos / kPerWord) * THREADS + threadIdx.x) << 2) + W * (pos % kPerWord);
        if (W == 1) *slot = uint8_t(s);
        else *reinterpret_cast<uint16_t *>(slot) = uint16_t(s);
      }
    }
    cnt += left;
  }
  if (cnt < kPos && cnt < rem) {  // ... or byte by byte at the very end of the buffer
    const uint32_t last = uint32_t(rem);  // < kPos here
    for (; cnt < last; ++cnt) {
      s = tab.next(s, uint32_t(p[cnt]));
      acc |= uint64_t(s >= c.firstAccept) << cnt;
      ini |= uint64_t(s == c.init) << cnt;
      uint8_t *slot = reinterpret_cast<uint8_t *>(stage) +
                      (((cnt / kPerWord) * THREADS + threadIdx.x) << 2) + W * (cnt % kPerWord);
      if (W == 1) *slot = uint8_t(s);
      else *reinterpret_cast<uint16_t *>(slot) = uint16_t(s);
    }
  }
  return cnt;
}

// =========================================================================================
// k_matchall_blocks: matchAllCore (include/Matcher.h:711-766) in two phases per block of a line.
//
// matchAllLane tests "did this byte accept, and is it a new record" at every byte: with 64 lanes
// some lane nearly always says yes (SYN-256: one state in seven accepts), so the wave runs the
// record path - result lookup, compare, three scattered stores - at every byte of every line.
// Here a lane takes its line in blocks of kPos positions and
//   A. WALKS the block straight-line with nothing data-dependent in it: per byte the lookup, one
//      bit "accepting" and one bit "is the initial state" shifted into two masks, and the state
//      itself packed into a word that goes to LDS every fourth (second) byte - the lane's kPos
//      states, at a lane-interleaved address (no bank conflicts);
//   B. VISITS the accepting positions of the block only (a per-lane loop over the set bits of
//      the mask): the state comes back from LDS, its result from the LDS result table, "same run
//      as the byte before" from the mask, the record's start from the highest "left the initial
//      state" bit at or below the position (Matcher.h:726-731).  The wave's trip count is the
//      largest accept count among its 64 lines' blocks, not the block length.
// Requires absorbing pure dead ends (nothing accepts past one, so not leaving at :755-756 changes
// nothing), an LDS-resident table kind and the result table in LDS.  W = bytes per staged state.
// =========================================================================================
template <int KIND, int THREADS, int W>
__global__ void __launch_bounds__(THREADS)
k_matchall_blocks(DevDfa d, Batch b, uint64_t cap, uint64_t *counts, int lead) {
  constexpr uint32_t kPos = 64 / W;       // positions per block: 64 bytes of staged states per lane
  constexpr uint32_t kPerWord = 4 / W;    // states per staged 32-bit word
  extern __shared__ __align__(16) uint8_t lds[];
  const Tab<KIND> tab = stageTab<KIND, THREADS>(d, lds);
  LaneCtx c{lds, lds + 256, resOf<KIND>(d, lds), d.init, d.leaderNext, d.nPureDead, d.firstAccept,
            d.leaderLen};
  uint32_t *stage = reinterpret_cast<uint32_t *>(lds + 512 + ((ldsTableBytes<KIND>(d) + 15) & ~size_t(15)));
  const uint8_t *stageBytes = reinterpret_cast<const uint8_t *>(stage);
  // the result table's LDS copy, addressed as LDS (through LaneCtx it is a generic pointer: flat loads)
  const int32_t *ldsRes = reinterpret_cast<const int32_t *>(lds + 512 + ((tableOnlyBytes<KIND>(d) + 15) & ~size_t(15)));
  // the asm walk addresses the table at LDS offset 512: true while the kernel has no static LDS
  const bool tableAt512 = uint32_t(reinterpret_cast<uintptr_t>(lds)) == 0u;
  const uint8_t *bufEnd = b.data + (b.offsets ? b.offsets[b.n] : b.n * b.stride);
  const uint64_t step = uint64_t(gridDim.x) * THREADS;
  for (uint64_t line = uint64_t(blockIdx.x) * THREADS + threadIdx.x; line < b.n; line += step) {
    const uint8_t *p;
    uint64_t n;
    if (b.offsets) {
      const uint64_t o = b.offsets[line];
      p = b.data + o;
      n = b.offsets[line + 1] - o;
      n = n >= b.stride ? n - b.stride : 0;  // stride = trailing bytes to drop (ragged)
    } else {
      p = b.data + line * b.stride;
      n = b.stride;
    }
    int32_t *res = b.result + line * cap;
    uint64_t *st = b.start ? b.start + line * cap : nullptr;
    uint64_t *en = b.end ? b.end + line * cap : nullptr;
    if (lead && !lookingAt(c, p, 0, n)) n = 0;  // (found stays 0)
    uint32_t s = c.init;
    int32_t prevR = 0;       // result at the last position of the block before, 0 if it did not accept
    uint64_t matchStart = 0, found = 0, curEnd = 0;
    for (uint64_t base = 0; base < n; base += kPos) {
      const uint64_t wasInit = s == c.init ? 1u : 0u;
      uint64_t acc = 0, ini = 0;
      // ---- A: the walk (mabWalkBlock) ------------------------------------------------------------
      const uint32_t cnt = mabWalkBlock<KIND, THREADS, W>(tab, c, p + base, n - base,
                                                          bufEnd - (p + base), s, stage,
                                                          tableAt512, acc, ini);
      // ---- B: the accepting positions ----------------------------------------------------------
      const uint64_t valid = cnt >= 64 ? ~0ull : (1ull << cnt) - 1;
      const uint64_t esc = (((ini << 1) | wasInit) & ~ini) & valid;  // "left the initial state" here
      const bool lastAcc = prevR != 0;
      auto resultAt = [&](uint32_t i) -> int32_t {
        const uint8_t *slot = stageBytes + (((i / kPerWord) * THREADS + threadIdx.x) << 2) +
                              W * (i % kPerWord);
        const uint32_t si = W == 1 ? uint32_t(*slot) : uint32_t(*reinterpret_cast<const uint16_t *>(slot));
        return ldsRes[si];
      };
      // B1. where records OPEN.  An accepting position behind a non-accepting one always does
      //     (prev is 0 there, :757); one behind an accepting position does when the two results
      //     differ (:747-752) - only those pairs need their results looked up.
      const uint64_t behindAcc = (acc << 1) | (lastAcc ? 1u : 0u);
      uint64_t opens = acc & ~behindAcc;
      for (uint64_t pairs = acc & behindAcc; pairs; pairs &= pairs - 1) {
        const uint32_t i = uint32_t(__builtin_ctzll(pairs));
        const int32_t before = i ? resultAt(i - 1) : prevR;
        if (resultAt(i) != before) opens |= 1ull << i;
      }
      // B2. the records themselves, while there is something to store (record cap - 1 waits for
      //     its end until the next one opens): the k-th trip stores every lane's k-th record of
      //     the block - a record ends behind the last accepting position before the next open.
      while (opens && cap && found <= cap) {
        const uint32_t i = uint32_t(__builtin_ctzll(opens));
        opens &= opens - 1;
        if (found && found - 1 < cap && en) {
          const uint64_t below = acc & ((1ull << i) - 1);
          en[found - 1] = below ? base + 64 - uint32_t(__builtin_clzll(below)) : curEnd;
        }
        if (found < cap) {
          res[found] = resultAt(i);
          if (st) {
            const uint64_t m = esc & ((2ull << i) - 1);
            st[found] = m ? base + 63 - uint32_t(__builtin_clzll(m)) : matchStart;
          }
        }
        ++found;
      }
      found += uint64_t(__builtin_popcountll(opens));  // the rest is only counted
      // carried into the next block: the end of the run in progress, whether its first position
      // continues a run (and with which result), and the last escape from the initial state
      if (acc) curEnd = base + 64 - uint32_t(__builtin_clzll(acc));
      prevR = cnt && ((acc >> (cnt - 1)) & 1u) ? resultAt(cnt - 1) : 0;
      if (esc) matchStart = base + 63 - uint32_t(__builtin_clzll(esc));
    }
    if (found && found - 1 < cap && en) en[found - 1] = curEnd;
    counts[line] = found;
  }
}
