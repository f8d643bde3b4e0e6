// k_common.h - table accessors (Tab<KIND>), table staging, the per-lane context and the byte readers
// (walkBytes, walkBytesPeek, the start-byte filters) every per-lane kernel is built on
// (included by kernels.hip inside namespace redgpu { namespace { ... } }; see its file map).
#pragma once

constexpr int kStyInstant = REDGPU_STY_INSTANT;
constexpr int kStyFirst = REDGPU_STY_FIRST;
constexpr int kStyTangent = REDGPU_STY_TANGENT;
constexpr int kStyLast = REDGPU_STY_LAST;
constexpr int kStyFull = REDGPU_STY_FULL;

// ---- table accessors -------------------------------------------------------------------
template <int KIND> struct Tab;

template <> struct Tab<REDGPU_TAB_LDS_FUSED_U8> {
  static constexpr bool kInLds = true;
  const uint8_t *t;
  __device__ Tab(const uint8_t *tab, const uint8_t *, uint32_t) : t(tab) {}
  __device__ __forceinline__ uint32_t next(uint32_t s, uint32_t byte) const {
    return t[(s << 8) | byte];
  }
};

template <> struct Tab<REDGPU_TAB_LDS_FUSED_U16> {
  static constexpr bool kInLds = true;
  const uint16_t *t;
  __device__ Tab(const uint8_t *tab, const uint8_t *, uint32_t)
      : t(reinterpret_cast<const uint16_t *>(tab)) {}
  __device__ __forceinline__ uint32_t next(uint32_t s, uint32_t byte) const {
    return t[(s << 8) | byte];
  }
};

template <> struct Tab<REDGPU_TAB_LDS_CLASS_U16> {
  static constexpr bool kInLds = true;
  const uint16_t *t;
  const uint8_t *eq;
  uint32_t nc;
  __device__ Tab(const uint8_t *tab, const uint8_t *equiv, uint32_t nClasses)
      : t(reinterpret_cast<const uint16_t *>(tab)), eq(equiv), nc(nClasses) {}
  __device__ __forceinline__ uint32_t next(uint32_t s, uint32_t byte) const {
    return t[s * nc + eq[byte]];
  }
};

template <> struct Tab<REDGPU_TAB_GLOBAL_U16> {
  static constexpr bool kInLds = false;
  const uint16_t *t;
  const uint8_t *eq;
  uint32_t nc;
  uint32_t nt = 0;  // DevDfa::gatherNt: gather with non-temporal loads (tuning experiment)
  __device__ Tab(const uint8_t *tab, const uint8_t *equiv, uint32_t nClasses)
      : t(reinterpret_cast<const uint16_t *>(tab)), eq(equiv), nc(nClasses) {}
  __device__ __forceinline__ uint32_t next(uint32_t s, uint32_t byte) const {
    const uint16_t *p = t + size_t(s) * nc + eq[byte];
    return nt ? __builtin_nontemporal_load(p) : *p;
  }
};

template <> struct Tab<REDGPU_TAB_GLOBAL_U32> {
  static constexpr bool kInLds = false;
  const uint32_t *t;
  const uint8_t *eq;
  uint32_t nc;
  __device__ Tab(const uint8_t *tab, const uint8_t *equiv, uint32_t nClasses)
      : t(reinterpret_cast<const uint32_t *>(tab)), eq(equiv), nc(nClasses) {}
  __device__ __forceinline__ uint32_t next(uint32_t s, uint32_t byte) const {
    return t[size_t(s) * nc + eq[byte]];
  }
};

// Hot rows (north star: "hot transition rows staged in LDS"): the n_hot most-visited states
// share a 64 KB [hot index][byte] u8 table in LDS - one ds_read_u8 per byte, no class lookup,
// for every transition that stays inside the hot set; 255 there (the target is not hot) and
// every cold state go through the class table in HBM/L2.  Hot states are one index range.
template <> struct Tab<REDGPU_TAB_HOT_ROWS> {
  static constexpr bool kInLds = false;
  const uint16_t *t;
  const uint8_t *hot;
  const uint8_t *eq;
  uint32_t nc, hotLo, nHot, shift;
  __device__ Tab(const uint8_t *tab, const uint8_t *equiv, uint32_t nClasses)
      : t(reinterpret_cast<const uint16_t *>(tab)), hot(nullptr), eq(equiv), nc(nClasses),
        hotLo(0), nHot(0), shift(0) {}
  __device__ __forceinline__ uint32_t next(uint32_t s, uint32_t byte) const {
    const uint32_t hr = s - hotLo;
    if (hr < nHot) {
      const uint32_t v = hot[((hr + shift) << 8) | byte];
      if (v != 255u) return (shift && v == 0) ? 0u : hotLo + v - shift;  // 0: a pure dead end
    }
    return t[size_t(s) * nc + eq[byte]];
  }
};

// Sparse rows (dfa_image.cpp): the whole DFA in LDS in row-displacement form.  Two dependent
// LDS reads per byte (base[state], then the slot) instead of an L2 round trip.
template <> struct Tab<REDGPU_TAB_LDS_SPARSE> {
  static constexpr bool kInLds = true;
  const uint16_t *base;
  const uint32_t *slot;
  const uint8_t *eq;
  uint32_t dflt;
  __device__ Tab(const uint8_t *tab, const uint8_t *equiv, uint32_t)
      : base(reinterpret_cast<const uint16_t *>(tab)), slot(nullptr), eq(equiv), dflt(0) {}
  __device__ __forceinline__ uint32_t next(uint32_t s, uint32_t byte) const {
    const uint32_t e = slot[uint32_t(base[s]) + eq[byte]];
    return (e >> 16) == s ? (e & 0xffffu) : dflt;
  }
};

// What a workgroup stages behind its 512 bytes of equivalence map + leader, and the accessor
// over it.  Whole table for the LDS kinds, the hot rows for REDGPU_TAB_HOT_ROWS, nothing else.
template <int KIND>
__host__ __device__ inline size_t tableOnlyBytes(const DevDfa &d) {
  if (Tab<KIND>::kInLds) return d.tableBytes;
  if (KIND == REDGPU_TAB_HOT_ROWS) return 65536u;
  return 0;
}

// The result table (int32 per state) rides along behind the table when it is small enough: the
// reference reads result() at every accepting state (include/Proxy.h:131-133), and read from
// global memory that is a dependent L2 round trip inside the loop of every lane function that
// does (check, scan, search, matchAll, collect, match with the early-exit styles).
template <int KIND>
__host__ __device__ inline bool resStaged(const DevDfa &d) {
  return d.nStates <= 4096 && ((tableOnlyBytes<KIND>(d) + 15) & ~size_t(15)) + size_t(d.nStates) * 4 <=
                                  size_t(146) * 1024;
}

template <int KIND>
__host__ __device__ inline size_t ldsTableBytes(const DevDfa &d) {
  const size_t t = (tableOnlyBytes<KIND>(d) + 15) & ~size_t(15);
  return resStaged<KIND>(d) ? t + ((size_t(d.nStates) * 4 + 15) & ~size_t(15)) : t;
}

// where the lane functions read results: the LDS copy when staged (stageTab), else global memory
template <int KIND>
__device__ __forceinline__ const int32_t *resOf(const DevDfa &d, const uint8_t *lds) {
  return resStaged<KIND>(d) ? reinterpret_cast<const int32_t *>(
                                  lds + 512 + ((tableOnlyBytes<KIND>(d) + 15) & ~size_t(15)))
                            : d.result;
}

template <int KIND, int THREADS, bool WITH_RES = true>
__device__ __forceinline__ Tab<KIND> stageTab(const DevDfa &d, uint8_t *lds) {
  uint8_t *eq = lds;
  uint8_t *ldsTab = lds + 512;
  for (uint32_t i = threadIdx.x; i < 512 / 4; i += THREADS)
    reinterpret_cast<uint32_t *>(lds)[i] = reinterpret_cast<const uint32_t *>(d.equivLeader)[i];
  const uint32_t n16 = uint32_t(tableOnlyBytes<KIND>(d) / 16);
  if (n16) {
    const uint4 *src = reinterpret_cast<const uint4 *>(
        d.table + (KIND == REDGPU_TAB_HOT_ROWS ? d.hot8Off : 0u));
    uint4 *dst = reinterpret_cast<uint4 *>(ldsTab);
    for (uint32_t i = threadIdx.x; i < n16; i += THREADS) dst[i] = src[i];
  }
  if (WITH_RES && resStaged<KIND>(d)) {
    int32_t *dst = reinterpret_cast<int32_t *>(ldsTab + ((tableOnlyBytes<KIND>(d) + 15) & ~size_t(15)));
    for (uint32_t i = threadIdx.x; i < d.nStates; i += THREADS) dst[i] = d.result[i];
  }
  __syncthreads();
  Tab<KIND> tab(Tab<KIND>::kInLds ? ldsTab : d.table, eq, d.nClasses);
  if constexpr (KIND == REDGPU_TAB_HOT_ROWS) {
    tab.hot = ldsTab;
    tab.hotLo = d.hotLo;
    tab.nHot = d.nHot;
    tab.shift = d.hotShift;
  }
  if constexpr (KIND == REDGPU_TAB_GLOBAL_U16) tab.nt = d.gatherNt;
  if constexpr (KIND == REDGPU_TAB_LDS_SPARSE) {
    tab.slot = reinterpret_cast<const uint32_t *>(ldsTab + d.sparseCombOff);
    tab.dflt = d.sparseDefault;
  }
  return tab;
}

struct LaneCtx {
  const uint8_t *eq;      // LDS: byte -> class
  const uint8_t *leader;  // LDS: class-space leader
  const int32_t *res;     // global: result per device state
  uint32_t init, leaderNext, nPureDead, firstAccept, leaderLen;
  // start bytes of scan / search attempts (DevDfa): [0] without the leader, [1] with it
  uint32_t startWord[2] = {0, 0}, startCount[2] = {0xff, 0xff};
  uint32_t start2Word[2] = {0, 0}, start2Count[2] = {0xff, 0xff};  // ... and of their second bytes
  uint32_t suffixClosed = 0;  // DevDfa::suffixClosed: a failed attempt at the end of the line ends the scan
  __device__ __forceinline__ int32_t resultOf(uint32_t s) const {
    return s >= firstAccept ? res[s] : 0;
  }
};

// include/Matcher.h:333-345 lookingAt: cursor by value, nothing consumed
__device__ __forceinline__ bool lookingAt(const LaneCtx &c, const uint8_t *p, uint64_t i,
                                          uint64_t n) {
  for (uint32_t k = 0; k < c.leaderLen; ++k, ++i) {
    if (i >= n) return false;
    if (c.leader[k] != c.eq[p[i]]) return false;
  }
  return true;
}

// include/Matcher.h:348-360 compareThrough: cursor by reference; on a mismatch the cursor
// stays AT the mismatching byte (the return precedes the increment)
__device__ __forceinline__ bool compareThrough(const LaneCtx &c, const uint8_t *p, uint64_t &i,
                                               uint64_t n) {
  for (uint32_t k = 0; k < c.leaderLen; ++k, ++i) {
    if (i >= n) return false;
    if (c.leader[k] != c.eq[p[i]]) return false;
  }
  return true;
}

// Feeds f(byte, index) the bytes p[from..n) in order until it returns false.  The body reads
// 16-byte aligned chunks (one global_load_dwordx4 per 16 input bytes instead of 16 byte loads);
// the unaligned head and the tail go byte by byte.
// Trip sizes: the first trip takes ONE 16-byte chunk (a line that dies in its first bytes -
// the anchored DFAs that live on these kernels - touches nothing else), every later trip takes
// up to four, requested back to back: a lane that takes its line 16 bytes at a time comes back
// to every 128-byte cache line 8 times, and with 64 lanes x 16+ waves per CU the lines are
// long gone from L1 and L2 by then (measured: ~1 TB/s of HBM-amplified traffic on 256-byte
// lines whatever the per-byte work; 1.9 TB/s with 64-byte trips).  Chunks are loaded at the
// line's own alignment (the memory pipeline splits unaligned requests); only the last < 16
// bytes go byte by byte, so no request reaches past the line.
template <class F>
__device__ __forceinline__ void walkBytes(const uint8_t *p, uint64_t from, uint64_t n, F &&f) {
  uint64_t i = from;
  uint32_t want = 1;
  while (i + 16 <= n) {
    const uint64_t avail = (n - i) >> 4;
    const uint32_t nc = avail < want ? uint32_t(avail) : want;
    const uint4 z = make_uint4(0, 0, 0, 0);
    const uint4 b0 = *reinterpret_cast<const uint4 *>(p + i);
    const uint4 b1 = nc > 1 ? *reinterpret_cast<const uint4 *>(p + i + 16) : z;
    const uint4 b2 = nc > 2 ? *reinterpret_cast<const uint4 *>(p + i + 32) : z;
    const uint4 b3 = nc > 3 ? *reinterpret_cast<const uint4 *>(p + i + 48) : z;
#pragma unroll 1
    for (uint32_t c = 0; c < nc; ++c) {
      const uint4 v = c == 0 ? b0 : c == 1 ? b1 : c == 2 ? b2 : b3;
      // the 16 byte steps of a chunk straight-line (round 1 rolled the words to keep the body
      // small: the dynamic word selects and the loop cost more than the code they saved)
      const uint32_t words[4] = {v.x, v.y, v.z, v.w};
      const uint64_t at = i + 16 * c;
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (!f((words[k >> 2] >> (8 * (k & 3))) & 0xffu, at + k)) return;
    }
    i += 16ull * nc;
    want = 4;
  }
  for (; i < n; ++i)
    if (!f(uint32_t(p[i]), i)) return;
}

// Every byte of p[0..n) to f(byte, index), no early exit, straight-line: 64-byte trips of four
// back-to-back requests, 16 byte steps per chunk unrolled (no rolled word loop, no per-byte
// branch).  For walks that never leave before the end of the line - matchAll over a DFA whose
// pure dead ends are absorbing: past one nothing accepts and nothing is recorded.
template <class F>
__device__ __forceinline__ void walkAllBytes(const uint8_t *p, uint64_t n, F &&f) {
  uint64_t i = 0;
  auto chunk = [&](const uint4 &v, uint64_t at) {
    const uint32_t words[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 16; ++k) f((words[k >> 2] >> (8 * (k & 3))) & 0xffu, at + k);
  };
#pragma unroll 1
  while (i + 64 <= n) {
    const uint4 b0 = *reinterpret_cast<const uint4 *>(p + i);
    const uint4 b1 = *reinterpret_cast<const uint4 *>(p + i + 16);
    const uint4 b2 = *reinterpret_cast<const uint4 *>(p + i + 32);
    const uint4 b3 = *reinterpret_cast<const uint4 *>(p + i + 48);
    chunk(b0, i);
    chunk(b1, i + 16);
    chunk(b2, i + 32);
    chunk(b3, i + 48);
    i += 64;
  }
#pragma unroll 1
  while (i + 16 <= n) {
    chunk(*reinterpret_cast<const uint4 *>(p + i), i);
    i += 16;
  }
  for (; i < n; ++i) f(uint32_t(p[i]), i);
}

// walkBytes that also hands f the NEXT byte (kNoPeek when it is not in the chunk in hand or
// past the end): scan and search reject almost every start position from two bytes in
// registers.  A lane that has to go back to memory for a survivor stalls its whole wave, and
// with one byte of filtering some lane of the 64 survives nearly every step (1 in 47 per lane
// on text); with two it is 1 in ~2000.
constexpr uint32_t kNoPeek = 0x100u;

// 0x80 in every byte of `word` that equals one of the `count` (1..4) bytes packed in `set`.
// Exact SWAR zero-byte test per member: ((x & 0x7f7f7f7f) + 0x7f7f7f7f) | x has the top bit of
// a byte clear iff that byte of x is zero - no borrow crosses bytes.
__device__ __forceinline__ uint32_t wordMatchMask(uint32_t word, uint32_t set, uint32_t count) {
  uint32_t hit = 0;
  for (uint32_t k = 0; k < count; ++k) {
    const uint32_t x = word ^ (((set >> (8 * k)) & 0xffu) * 0x01010101u);
    hit |= ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x);
  }
  return hit & 0x80808080u;
}

// Input words none of whose positions can start a surviving attempt (StartFilter) are stepped
// over whole - onSkip() stands for the four rejected attempts.  The test is on byte PAIRS where
// the DFA allows it: what matters is not how rare a candidate is per lane but per WAVE - one
// lane with a candidate drags all 64 through the per-byte path (on text, a lone 'e' turns up in
// some lane's word at 99.5 % of the steps; "er" at 11 %).
struct StartFilter {
  uint32_t set1, n1;  // start bytes (n1 in 1..4, or 0 = no filter)
  uint32_t set2, n2;  // bytes that may follow one (0 = no second filter)
  // scan with the leader: a start byte followed by a wrong second byte makes compareThrough
  // stop ON that second byte and the outer ++in skip it (Matcher.h:511-518) - if that byte is a
  // start byte itself, skipping it changes the outcome ("aab" on "aaab"), so such a position
  // must still be walked: followers that are start bytes count as possible too
  bool consumes;
};

template <class S, class F>
__device__ __forceinline__ void walkBytesPeek(const uint8_t *p, uint64_t from, uint64_t n,
                                              const StartFilter flt, S &&onSkip, F &&f) {
  uint64_t i = from;
  uint32_t want = 1;  // trip sizes as in walkBytes
  while (i + 16 <= n) {
    const uint64_t avail = (n - i) >> 4;
    const uint32_t nc = avail < want ? uint32_t(avail) : want;
    const uint4 z = make_uint4(0, 0, 0, 0);
    const uint4 b0 = *reinterpret_cast<const uint4 *>(p + i);
    const uint4 b1 = nc > 1 ? *reinterpret_cast<const uint4 *>(p + i + 16) : z;
    const uint4 b2 = nc > 2 ? *reinterpret_cast<const uint4 *>(p + i + 32) : z;
    const uint4 b3 = nc > 3 ? *reinterpret_cast<const uint4 *>(p + i + 48) : z;
#pragma unroll 1
    for (uint32_t c = 0; c < nc; ++c) {
      const uint4 v = c == 0 ? b0 : c == 1 ? b1 : c == 2 ? b2 : b3;
      const uint32_t after = c == 0 ? b1.x : c == 1 ? b2.x : b3.x;  // first word of the next chunk
      const bool haveAfter = c + 1 < nc;
#pragma unroll 1
      for (int wi = 0; wi < 4; ++wi) {
        const uint32_t word = wi == 0 ? v.x : wi == 1 ? v.y : wi == 2 ? v.z : v.w;
        const uint32_t nextWord = wi == 0 ? v.y : wi == 1 ? v.z : wi == 2 ? v.w : after;
        if (flt.n1) {
          // positions of this word that can start a surviving attempt: a start byte, followed
          // (when the follower is in hand) by a byte that may follow one
          uint32_t cand = wordMatchMask(word, flt.set1, flt.n1);
          if (cand && flt.n2) {
            const bool haveNext = wi < 3 || haveAfter;
            const uint32_t follow = (word >> 8) | (nextWord << 24);
            uint32_t ok = wordMatchMask(follow, flt.set2, flt.n2);
            if (flt.consumes) ok |= wordMatchMask(follow, flt.set1, flt.n1);
            if (!haveNext) ok |= 0x80000000u;  // the last byte's follower is not in hand
            cand &= ok;
          }
          if (!cand) {
            onSkip();
            continue;
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t nb = k < 3 ? (word >> (8 * (k + 1))) & 0xffu
                                    : ((wi < 3 || haveAfter) ? nextWord & 0xffu : kNoPeek);
          if (!f((word >> (8 * k)) & 0xffu, i + 16 * c + 4 * wi + k, nb)) return;
        }
      }
    }
    i += 16ull * nc;
    want = 4;
  }
  for (; i < n; ++i)
    if (!f(uint32_t(p[i]), i, kNoPeek)) return;
}

// include/Matcher.h:363-410
template <class T>
__device__ int32_t checkLane(const T &tab, const LaneCtx &c, const uint8_t *p, uint64_t n,
                             int style, bool lead) {
  uint64_t i = 0;
  uint32_t s;
  if (lead) {
    if (!compareThrough(c, p, i, n)) return 0;
    s = c.leaderNext;
  } else
    s = c.init;
  int32_t result = c.resultOf(s);
  int32_t prev = 0;
  bool returned = false;
  int32_t retval = 0;
  walkBytes(p, i, n, [&](uint32_t byte, uint64_t) {
    s = tab.next(s, byte);
    if (s >= c.firstAccept) {
      result = c.res[s];
      if (style == kStyInstant) { returned = true; retval = result; return false; }
      if (style == kStyFirst) {
        if (prev && result != prev) { returned = true; retval = prev; return false; }
        prev = result;
      }
      if (style == kStyTangent || style == kStyLast) prev = result;
    } else {
      result = 0;
      if ((style == kStyFirst || style == kStyTangent) && prev > 0) {
        returned = true; retval = prev; return false;
      }
      if (s < c.nPureDead) return false;
    }
    return true;
  });
  if (returned) return retval;
  if (style == kStyLast && result == 0 && prev > 0) return prev;
  return result;
}
