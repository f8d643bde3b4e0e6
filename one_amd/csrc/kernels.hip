// kernels.hip - hand-written gfx950 (CDNA4 / MI355X) kernels for RED's DFA match execution.
//
// What is restated here, one input line per lane (citations relative to
// /root/reference/quol/red/):
//   checkCore  include/Matcher.h:363-410      matchCore  include/Matcher.h:413-495
//   scanCore   include/Matcher.h:498-554      lookingAt / compareThrough  :333-360
//   the per-byte step DfaProxy::next/result/pureDeadEnd   include/Proxy.h:131-147
// over the renumbered device image of dfa_image.h (s < nPureDead <=> pureDeadEnd,
// s >= firstAccept <=> result > 0, so the per-byte predicates are integer compares).
//
// Map of this file and its includes (DESIGN.md section 4 has the measurements):
//   k_stream.h   k_stream<MODE, HALVES, THREADS, TABK>: the hot path - fixed-stride lines that are
//                whole 64-byte blocks, styles Last / Full of check / match, StatefulMatcher
//                chunks; inline-asm byte step over a fused u8 table (<= 256 states), the hot-row
//                table of a big DFA (sink + re-walk) or a class table of <= 64 KB in LDS.
//   k_ragged.h   k_ragged<MODE, TABK>: the same walk over ragged lines (offsets[n+1]), lanes
//                refilled from a workgroup cursor; the tail pad; k_generic's bucketing pre-pass.
//   k_chunk.h    few long lines: chunks walked at once from guessed entry states, wrong guesses
//                re-walked (speculative chunking).
//   here         the lane functions (checkLane ... replaceLane: direct restatements of the
//                reference's cores), k_generic<KIND, THREADS, VERB> and the list / rewrite
//                kernels built on them (k_collect, k_matchall, k_replace, k_advance, k_visits),
//                k_scan_marked (scan / search in two passes: mark candidate positions, visit them),
//                k_fixed (strides that are not whole blocks, early-exit styles), line splitting,
//                and launchBatch: which kernel runs what.
// No MFMA anywhere: this is a gather workload bounded by the LDS gather rate and HBM streaming.
#include "kernels.h"

#include <cstdlib>
#include <type_traits>

#include "../../include/redgpu.h"

// This file is compiled three times in parallel (Makefile: -DREDGPU_TU=1/2/3): the templates are
// instantiated where their launchers are CALLED, so each translation unit only pays for one
// family of kernels - 1 = the fixed-stride family (k_stream, k_chunk, k_fixed), 2 = k_ragged,
// 3 = everything else and the dispatch.  REDGPU_TU undefined or 0 = all of it in one unit.
#ifndef REDGPU_TU
#define REDGPU_TU 0
#endif
#define REDGPU_TU_STREAM (REDGPU_TU == 0 || REDGPU_TU == 1)
#define REDGPU_TU_RAGGED (REDGPU_TU == 0 || REDGPU_TU == 2)
#define REDGPU_TU_GENERIC (REDGPU_TU == 0 || REDGPU_TU == 3)

namespace redgpu {

namespace {

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

// include/Matcher.h:413-495: the state matchCore's loop carries from byte to byte, resumable -
// matchLane runs it over a whole line; k_early stops after a few bytes, parks the survivors in
// LDS and lets other lanes pick them up.
struct MatchWalk {
  uint32_t s;
  int32_t result, prev;
  uint64_t matchStart, matchEnd;
  __device__ __forceinline__ void begin(const LaneCtx &c) {
    s = c.init;
    result = c.resultOf(s);
    prev = 0;
    matchStart = 0;
    matchEnd = 0;
  }
  // one iteration of the loop at :443-479; false = the loop breaks
  template <class T>
  __device__ __forceinline__ bool step(const T &tab, const LaneCtx &c, int style, uint32_t byte,
                                       uint64_t idx) {
    const uint32_t was = s;
    s = tab.next(s, byte);
    if (was == c.init && s != was) matchStart = idx;  // "escaped the initial state" :446-451
    if (s >= c.firstAccept) {
      result = c.res[s];
      if (style == kStyFirst) {
        if (prev && result != prev) { result = prev; return false; }
        prev = result;
      }
      matchEnd = idx + 1;
      if (style == kStyInstant) return false;
      if (style == kStyTangent || style == kStyLast) prev = result;
    } else {
      result = 0;
      if (style == kStyFirst && prev > 0) { result = prev; return false; }
      if (style == kStyTangent && prev > 0) return false;
      if (s < c.nPureDead) return false;
    }
    return true;
  }
  // the fix-up behind the loop, :481-494
  __device__ __forceinline__ int32_t finish(int style, uint64_t &startOut, uint64_t &endOut) {
    startOut = 0;
    endOut = 0;
    if ((style == kStyTangent || style == kStyLast) && result == 0 && prev > 0) result = prev;
    if (result != 0) {
      startOut = matchStart;
      endOut = matchEnd;
    }
    return result;
  }
};

// match<styLast> ("matchLong") alone, lean: what the loop leaves behind is the LAST accepting
// state, its end and the last "left the initial state" position - result = res[that state] is
// looked up once at the end instead of at every accept (a dependent global load in the loop), and
// no style is tested per byte.  Same Outcome as MatchWalk with style == kStyLast: there
// `prev` is the last accept's result, `result` is 0 or that same value, and finish() returns it.
struct LastWalk {
  uint32_t s, accS;
  uint64_t matchStart, matchEnd;  // matchEnd > 0 <=> some state accepted
  bool fresh;                     // no byte consumed yet: the Outcome is the initial state's (:435)
  __device__ __forceinline__ void begin(const LaneCtx &c) {
    s = c.init;
    accS = 0;
    matchStart = 0;
    matchEnd = 0;
    fresh = true;
  }
  template <class T>
  __device__ __forceinline__ bool step(const T &tab, const LaneCtx &c, int, uint32_t byte,
                                       uint64_t idx) {
    const uint32_t was = s;
    s = tab.next(s, byte);
    fresh = false;
    if (was == c.init && s != was) matchStart = idx;
    const bool acc = s >= c.firstAccept;
    if (acc) { accS = s; matchEnd = idx + 1; }
    return acc || s >= c.nPureDead;
  }
  __device__ __forceinline__ int32_t finish(const LaneCtx &c, int, uint64_t &startOut,
                                            uint64_t &endOut) {
    // an accepting initial state is only ever reported for an empty input
    const int32_t r = fresh ? c.resultOf(s) : matchEnd ? c.res[accS] : 0;
    startOut = r ? matchStart : 0;
    endOut = r ? matchEnd : 0;
    return r;
  }
  // parked in LDS after at most 255 bytes
  __device__ __forceinline__ uint4 pack(uint32_t line) const {
    return make_uint4(line, s | (accS << 16), uint32_t(matchStart) | (uint32_t(matchEnd) << 8), 0u);
  }
  __device__ __forceinline__ void unpack(const uint4 &e) {
    s = e.y & 0xffffu;
    accS = e.y >> 16;
    matchStart = e.z & 0xffu;
    matchEnd = (e.z >> 8) & 0xffu;
    fresh = false;
  }
};

// the general form behind the same interface (any style, tested per byte)
struct AnyWalk : MatchWalk {
  __device__ __forceinline__ int32_t finish(const LaneCtx &, int style, uint64_t &startOut,
                                            uint64_t &endOut) {
    return MatchWalk::finish(style, startOut, endOut);
  }
  __device__ __forceinline__ uint4 pack(uint32_t line) const {
    return make_uint4(line, s | (uint32_t(matchStart) << 16) | (uint32_t(matchEnd) << 24),
                      uint32_t(prev), uint32_t(result));
  }
  __device__ __forceinline__ void unpack(const uint4 &e) {
    s = e.y & 0xffffu;
    matchStart = (e.y >> 16) & 0xffu;
    matchEnd = e.y >> 24;
    prev = int32_t(e.z);
    result = int32_t(e.w);
  }
};

// include/Matcher.h:363-410 without the leader (doLeader false, or a DFA that has none): the
// state checkCore's loop carries from byte to byte, resumable like MatchWalk, for k_early.
// checkLane below stays the general form (it also consumes a leader).
struct CheckWalk {
  uint32_t s;
  int32_t result, prev;
  bool returned;
  int32_t retval;
  __device__ __forceinline__ void begin(const LaneCtx &c) {
    s = c.init;
    result = c.resultOf(s);
    prev = 0;
    returned = false;
    retval = 0;
  }
  template <class T>
  __device__ __forceinline__ bool step(const T &tab, const LaneCtx &c, int style, uint32_t byte,
                                       uint64_t) {
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
  }
  __device__ __forceinline__ int32_t finish(const LaneCtx &, int style, uint64_t &startOut,
                                            uint64_t &endOut) {
    startOut = 0;
    endOut = 0;
    if (returned) return retval;
    if (style == kStyLast && result == 0 && prev > 0) return prev;
    return result;
  }
  __device__ __forceinline__ uint4 pack(uint32_t line) const {
    return make_uint4(line, s, uint32_t(prev), uint32_t(result));
  }
  __device__ __forceinline__ void unpack(const uint4 &e) {
    s = e.y;
    prev = int32_t(e.z);
    result = int32_t(e.w);
    returned = false;
    retval = 0;
  }
};

template <class T>
__device__ int32_t matchLane(const T &tab, const LaneCtx &c, const uint8_t *p, uint64_t n,
                             int style, bool lead, uint64_t &startOut, uint64_t &endOut) {
  startOut = 0;
  endOut = 0;
  if (lead && !lookingAt(c, p, 0, n)) return 0;
  MatchWalk w;
  w.begin(c);
  walkBytes(p, 0, n, [&](uint32_t byte, uint64_t idx) { return w.step(tab, c, style, byte, idx); });
  return w.finish(style, startOut, endOut);
}

// check<styLast / styFull> without a leader over a DFA whose dead ends are absorbing (and that is
// not an early-death DFA): every byte, no exit test, no per-byte result lookup - styFull is the
// final state's result, styLast the last accepting state's (include/Matcher.h:382-409; an empty
// input answers with the initial state's result either way).
template <class T, bool FULL>
__device__ int32_t checkLeanLane(const T &tab, const LaneCtx &c, const uint8_t *p, uint64_t n) {
  uint32_t s = c.init, accS = 0;
  bool any = false;
  walkAllBytes(p, n, [&](uint32_t byte, uint64_t) {
    s = tab.next(s, byte);
    if (!FULL && s >= c.firstAccept) { accS = s; any = true; }
  });
  if (n == 0) return c.resultOf(c.init);
  if (FULL) return c.resultOf(s);
  return any ? c.res[accS] : 0;
}

// match<styLast> through the lean walk: the result table is read once, after the loop - with
// c.res[s] inside it every accepting step is a second dependent global load on the wave's
// critical path (a table in L2: two round trips per byte instead of one)
template <class T, bool NOEXIT = false>
__device__ int32_t matchLastLane(const T &tab, const LaneCtx &c, const uint8_t *p, uint64_t n,
                                 bool lead, uint64_t &startOut, uint64_t &endOut) {
  startOut = 0;
  endOut = 0;
  if (lead && !lookingAt(c, p, 0, n)) return 0;
  LastWalk w;
  w.begin(c);
  if constexpr (NOEXIT)  // absorbing dead ends, not an early-death DFA: every byte, no exit test
    walkAllBytes(p, n, [&](uint32_t byte, uint64_t idx) { (void)w.step(tab, c, kStyLast, byte, idx); });
  else
    walkBytes(p, 0, n, [&](uint32_t byte, uint64_t idx) { return w.step(tab, c, kStyLast, byte, idx); });
  return w.finish(c, kStyLast, startOut, endOut);
}

// include/Matcher.h:498-554.  The start positions are visited through walkBytes (16-byte
// chunks in registers, one per 16 positions) and almost every one is rejected from the byte in
// hand: with a leader, when its class is not the leader's first (compareThrough fails at k = 0:
// the cursor stays put and the loop's ++in moves on - nothing else changes); without one, when
// the first transition lands on a pure dead end.  Only the survivors touch memory again.
// One scan in progress: the state scanCore's outer loop carries from start position to start
// position, and visit() = one iteration of that loop for the position in hand.  Shared by the
// per-lane walk (scanLane) and the candidate-list walk of k_scan_marked.
template <class T>
struct ScanWalk {
  const T &tab;
  const LaneCtx &c;
  const uint8_t *p;
  uint64_t n;
  int style;
  bool lead;
  int32_t result, ret;
  bool returned;
  uint64_t resume;  // the next start position the reference's outer loop would visit
  uint32_t lead0, lead1;
  __device__ ScanWalk(const T &tab_, const LaneCtx &c_, const uint8_t *p_, uint64_t n_, int style_,
                      bool lead_)
      : tab(tab_), c(c_), p(p_), n(n_), style(style_), lead(lead_), result(c_.resultOf(c_.init)),
        ret(0), returned(false), resume(0), lead0(lead_ ? c_.leader[0] : 0u),
        lead1(lead_ && c_.leaderLen > 1 ? uint32_t(c_.leader[1]) : kNoPeek) {}
  // positions stepped over because no attempt can survive there: with the leader nothing changes
  // (compareThrough fails at k = 0), without it each attempt ends on a dead first step, result 0
  __device__ __forceinline__ void skipped() { if (!lead) result = 0; }
  __device__ __forceinline__ int32_t value() const { return returned ? ret : result; }
  // false = the scan has returned
  __device__ bool visit(uint32_t byte, uint64_t i, uint32_t nextByte) {
    if (i < resume) return true;
    uint32_t s;
    uint64_t q;  // the inner walk reads p[q..n)
    int32_t prev = 0;
    bool alive = true;
    if (lead) {
      if (c.eq[byte] != lead0) return true;
      if (lead1 != kNoPeek && nextByte != kNoPeek && c.eq[nextByte] != lead1) {
        resume = i + 2;  // compareThrough stops ON the second byte; ++in steps past it
        return true;
      }
      uint64_t j = i;
      if (!compareThrough(c, p, j, n)) {  // j sits on the mismatching byte; ++in skips it
        resume = j + 1;
        return true;
      }
      s = c.leaderNext;
      result = c.resultOf(s);
      q = j;
      resume = j + 1;
    } else {
      // first transition from the byte in hand
      s = tab.next(c.init, byte);
      q = i + 1;
      resume = i + 1;
      if (s >= c.firstAccept) {
        result = c.res[s];
        if (style == kStyInstant) { ret = result; returned = true; return false; }
        prev = result;  // First: prev was 0, so no early return; Tangent / Last: prev = result
        if (style == kStyFull) prev = 0;
      } else {
        result = 0;
        if (s < c.nPureDead) alive = false;
        // second transition from the byte in hand: most survivors of the first die here
        else if (nextByte != kNoPeek && tab.next(s, nextByte) < c.nPureDead) alive = false;
      }
    }
    if (alive) {
      for (; q < n; ++q) {
        s = tab.next(s, p[q]);
        if (s >= c.firstAccept) {
          result = c.res[s];
          if (style == kStyInstant) { ret = result; returned = true; return false; }
          if (style == kStyFirst) {
            if (prev && result != prev) { ret = prev; returned = true; return false; }
            prev = result;
          }
          if (style == kStyTangent || style == kStyLast) prev = result;
        } else {
          result = 0;
          if ((style == kStyFirst || style == kStyTangent) && prev > 0) {
            ret = prev; returned = true; return false;
          }
          if (s < c.nPureDead) break;
        }
      }
    }
    if (style == kStyLast && result == 0 && prev > 0) { ret = prev; returned = true; return false; }
    if (result > 0) { ret = result; returned = true; return false; }
    return true;
  }
};

template <class T>
__device__ int32_t scanLane(const T &tab, const LaneCtx &c, const uint8_t *p, uint64_t n,
                            int style, bool lead) {
  ScanWalk<T> w(tab, c, p, n, style, lead);
  const int li = lead ? 1 : 0;
  const StartFilter flt{c.startWord[li], c.startCount[li] <= 4 ? c.startCount[li] : 0u,
                        c.start2Word[li], c.start2Count[li] <= 4 ? c.start2Count[li] : 0u, lead};
  walkBytesPeek(p, 0, n, flt, [&]() { w.skipped(); },
                [&](uint32_t byte, uint64_t i, uint32_t nextByte) { return w.visit(byte, i, nextByte); });
  return w.value();
}

// include/Matcher.h:557-640 searchCore: sliding-window match; the leader is only PEEKED
// (lookingAt), so no start position is skipped - unlike scanCore.  Start positions come
// through walkBytes and are rejected from the byte in hand like scanLane's.
// searchLane's loop body as an object, like ScanWalk (shared with k_scan_marked)
template <class T>
struct SearchWalk {
  const T &tab;
  const LaneCtx &c;
  const uint8_t *p;
  uint64_t n;
  int style;
  bool lead;
  int32_t result;
  uint64_t matchStart, matchEnd;
  uint32_t lead0, lead1;
  __device__ SearchWalk(const T &tab_, const LaneCtx &c_, const uint8_t *p_, uint64_t n_, int style_,
                        bool lead_)
      : tab(tab_), c(c_), p(p_), n(n_), style(style_), lead(lead_), result(c_.resultOf(c_.init)),
        matchStart(0), matchEnd(0), lead0(lead_ ? c_.leader[0] : 0u),
        lead1(lead_ && c_.leaderLen > 1 ? uint32_t(c_.leader[1]) : kNoPeek) {}
  __device__ __forceinline__ void skipped() { if (!lead) result = 0; }
  // false = the search has found its match
  __device__ bool visit(uint32_t byte, uint64_t idx, uint32_t nextByte) {
    if (lead) {
      if (c.eq[byte] != lead0) return true;
      if (lead1 != kNoPeek && nextByte != kNoPeek && c.eq[nextByte] != lead1) return true;
      if (!lookingAt(c, p, idx, n)) return true;
    }
    // first transition from the byte in hand (:589-600 with q == idx)
    uint32_t s = tab.next(c.init, byte);
    int32_t prev = 0;
    matchStart = idx;  // set at the top of the attempt, and again if the step leaves init
    matchEnd = idx;
    bool walk = true;
    if (s >= c.firstAccept) {
      result = c.res[s];
      if (style == kStyFirst) prev = result;
      matchEnd = idx + 1;
      if (style == kStyInstant) walk = false;
      if (style == kStyTangent || style == kStyLast) prev = result;
    } else {
      result = 0;
      if (s < c.nPureDead) walk = false;
      // second transition from the byte in hand (a non-accepting dead end leaves result 0 and
      // the positions are only reported for a positive result)
      else if (nextByte != kNoPeek && tab.next(s, nextByte) < c.nPureDead) walk = false;
    }
    if (walk) {
      for (uint64_t q = idx + 1; q < n; ++q) {
        const uint32_t was = s;
        s = tab.next(s, p[q]);
        if (was == c.init && s != was) matchStart = q;
        if (s >= c.firstAccept) {
          result = c.res[s];
          if (style == kStyFirst) {
            if (prev && result != prev) { result = prev; break; }
            prev = result;
          }
          matchEnd = q + 1;
          if (style == kStyInstant) break;
          if (style == kStyTangent || style == kStyLast) prev = result;
        } else {
          result = 0;
          if (style == kStyFirst && prev > 0) { result = prev; break; }
          if (style == kStyTangent && prev > 0) break;
          if (s < c.nPureDead) break;
        }
      }
    }
    if ((style == kStyTangent || style == kStyLast) && result == 0 && prev > 0) result = prev;
    return !(result > 0);
  }
};

template <class T>
__device__ int32_t searchLane(const T &tab, const LaneCtx &c, const uint8_t *p, uint64_t n,
                              int style, bool lead, uint64_t &startOut, uint64_t &endOut) {
  startOut = 0;
  endOut = 0;
  SearchWalk<T> w(tab, c, p, n, style, lead);
  const int li = lead ? 1 : 0;
  const StartFilter flt{c.startWord[li], c.startCount[li] <= 4 ? c.startCount[li] : 0u,
                        c.start2Word[li], c.start2Count[li] <= 4 ? c.start2Count[li] : 0u, false};
  walkBytesPeek(p, 0, n, flt, [&]() { w.skipped(); },
                [&](uint32_t byte, uint64_t idx, uint32_t nextByte) { return w.visit(byte, idx, nextByte); });
  if (w.result != 0) {
    startOut = w.matchStart;
    endOut = w.matchEnd;
  }
  return w.result;
}

// dynamic LDS: [equiv 256][leader 256][table (LDS kinds only)].  An LDS-resident table is
// shared by one 1024-thread workgroup per CU; a table in HBM/L2 runs 256-thread workgroups.
// One instantiation per verb: the four lane functions together need twice the registers any
// one of them does.
template <int KIND, int kGenericThreads, int VERB>
__global__ void __launch_bounds__(kGenericThreads)
k_generic(DevDfa d, Batch b, int style, int lead) {
  constexpr int verb = VERB;
  extern __shared__ __align__(16) uint8_t lds[];
  uint8_t *eq = lds;
  uint8_t *leader = lds + 256;
  const Tab<KIND> tab = stageTab<KIND, kGenericThreads>(d, lds);
  LaneCtx c{eq, leader, resOf<KIND>(d, lds), d.init, d.leaderNext, d.nPureDead, d.firstAccept,
            d.leaderLen};
  c.startWord[0] = d.startFreeWord; c.startCount[0] = d.startFreeCount;
  c.startWord[1] = d.startLeadWord; c.startCount[1] = d.startLeadCount;
  c.start2Word[0] = d.start2FreeWord; c.start2Count[0] = d.start2FreeCount;
  c.start2Word[1] = d.start2LeadWord; c.start2Count[1] = d.start2LeadCount;
  c.suffixClosed = d.suffixClosed;

  const uint64_t step = uint64_t(gridDim.x) * kGenericThreads;
  // ragged lines bucketed by length (k_ragged.h): a wave's 64 lines then end together
  const bool usePerm = b.perm && b.perm[b.n] != 0;
  // Batch::spread > 1 (fewer lines than lanes, table in L2): one line per `spread` lanes - a wave
  // then gathers 64 / spread table rows per step instead of 64, and more waves share the CU
  if (b.spread > 1 && (threadIdx.x % b.spread)) return;
  for (uint64_t idx = (uint64_t(blockIdx.x) * kGenericThreads + threadIdx.x) / b.spread; idx < b.n;
       idx += step / b.spread) {
    const uint64_t line = usePerm ? b.perm[idx] : idx;
    const uint8_t *p;
    uint64_t n;
    if (b.offsets) {
      const uint64_t o = b.offsets[line];
      const uint64_t e = b.offsets[line + 1];
      p = b.data + o;
      n = e - o >= b.stride ? e - o - b.stride : 0;  // stride = trailing bytes to drop (ragged)
    } else {
      p = b.data + line * b.stride;
      n = b.stride;
    }
    if (verb == kCheck) {
      const bool lean = !lead && d.deadAbsorbing && !d.earlyDeath;
      b.result[line] = lean && style == kStyFull   ? checkLeanLane<Tab<KIND>, true>(tab, c, p, n)
                       : lean && style == kStyLast ? checkLeanLane<Tab<KIND>, false>(tab, c, p, n)
                                                   : checkLane(tab, c, p, n, style, lead != 0);
    } else if (verb == kScan) {
      b.result[line] = scanLane(tab, c, p, n, style, lead != 0);
    } else {
      uint64_t st, en;
      b.result[line] = verb == kSearch ? searchLane(tab, c, p, n, style, lead != 0, st, en)
                       : style == kStyLast
                           ? (d.deadAbsorbing && !d.earlyDeath
                                  ? matchLastLane<Tab<KIND>, true>(tab, c, p, n, lead != 0, st, en)
                                  : matchLastLane(tab, c, p, n, lead != 0, st, en))
                                           : matchLane(tab, c, p, n, style, lead != 0, st, en);
      if (b.start) b.start[line] = st;
      if (b.end) b.end[line] = en;
    }
  }
}

// =========================================================================================
// k_early<KIND>: match<style,doLeader> for EARLY-DEATH DFAs - anchored patterns and signature
// sets on arbitrary lines (BASELINE configs[3]: LOG-100, matchLong over 8 M ragged lines), where
// most lines are in a pure dead end after a byte or two and the others walk a whole signature.
// k_generic gives a lane a line: a wave then holds 64 lines until its slowest one is done (half
// of its lanes idle on configs[3]) and pays the line's memory round trips - offsets, first
// bytes, next trip - one behind the other.  Here a workgroup takes LPL lines per lane at a time
// and
//   1. PROBES them: offsets and the first PC x 16 bytes of all of them are requested together,
//      then each is walked through those bytes from registers (eight at a time; a wave moves on
//      once none of its lanes is alive).  A line that is done by then (pure dead end, an
//      early-exit style, end of line) stores its Outcome; a survivor's loop state (MatchWalk) is
//      parked in an LDS queue (one wave-aggregated atomic per wave and line slot);
//   2. DRAINS the queue: the survivors, now dense, are dealt out again - every lane resumes one
//      behind the bytes the probe held and walks it to its end.
// Same lane code as matchLane (MatchWalk::step / finish), so the results are the reference's
// for every style; the table kinds are the LDS-resident ones.
// =========================================================================================
// bytes of a line (>= 16 long) the probe holds in registers: whole 16-byte pieces, at most PC
template <int PC>
__device__ __forceinline__ uint32_t earlyHave(uint64_t n) {
  const uint64_t pieces = n >> 4;
  return 16u * uint32_t(pieces < uint64_t(PC) ? pieces : uint64_t(PC));
}

// LPL = lines per lane and round; WPS = waves per SIMD the register allocation must allow: LDS
// decides how many workgroups share a CU, and occupancy is what this kernel lives on.  Measured on
// configs[3] (2^23 lines, LOG-100; scripts/gpu_run10.sh): 4 lines per lane, 2 workgroups per CU
// 443 us; 2 lines per lane, 3 workgroups per CU 415 us; 1 line, 3 workgroups 437 us.  Requesting
// the next round's offsets and first bytes a phase ahead, and draining two survivors per lane with
// their next 64 bytes requested together, both made it slower (446-569 us: more registers, and
// the launch moves ~2.2 GB through L2 - nearly every cache line of the input is touched by a line
// start, and again when a survivor is drained - so it sits near the memory system's rate for
// scattered 128-byte requests, not on the latency of any one of them).
// PC = 16-byte pieces of a line the probe holds: all 16 bytes of the first piece walked in the
// probe (it was 8: the lines that die between byte 8 and 16 no longer pay a queue slot and a
// reload) 415 -> 345 us; two or four pieces (fewer reloads: 1.6 GB instead of 2.1 GB missing L2)
// 346 / 377 us - no faster; 1024-thread workgroups (32 waves per CU) 363 us; the drain's next
// two pieces requested together 360 us (scripts/gpu_run17.sh).
template <int KIND, class WALK, int LPL, int WPS, int PC, int THREADS, bool LEAN_DRAIN>
__global__ void __launch_bounds__(THREADS, WPS)
k_early(DevDfa d, Batch b, int style, int lead) {
  constexpr uint32_t kEarlyChunk = THREADS * LPL;
  constexpr uint32_t L = LPL;
  extern __shared__ __align__(16) uint8_t lds[];
  // (no LDS copy of the result table: this kernel reads it once per line, and the space buys a
  // third workgroup per CU)
  const Tab<KIND> tab = stageTab<KIND, THREADS, false>(d, lds);
  uint4 *queue = reinterpret_cast<uint4 *>(lds + 512 + ((tableOnlyBytes<KIND>(d) + 15) & ~size_t(15)));
  __shared__ uint32_t qCount;
  LaneCtx c;
  c.eq = lds;
  c.leader = lds + 256;
  c.res = d.result;
  c.init = d.init; c.leaderNext = d.leaderNext; c.nPureDead = d.nPureDead;
  c.firstAccept = d.firstAccept; c.leaderLen = d.leaderLen;
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t nChunks = (b.n + kEarlyChunk - 1) / kEarlyChunk;

  // line -> (byte offset, length); lines past the end of the batch read the last line (never
  // stored: `valid` below)
  auto spanOf = [&](uint64_t line, uint64_t &o, uint64_t &n) {
    const uint64_t ln = line < b.n ? line : b.n - 1;
    if (b.offsets) {
      o = b.offsets[ln];
      const uint64_t e = b.offsets[ln + 1];
      n = e - o >= b.stride ? e - o - b.stride : 0;  // stride = trailing bytes to drop (ragged)
    } else {
      o = ln * b.stride;
      n = b.stride;
    }
  };
  auto store = [&](uint64_t line, WALK &w) {
    uint64_t st, en;
    b.result[line] = w.finish(c, style, st, en);
    if (b.start) b.start[line] = st;
    if (b.end) b.end[line] = en;
  };

  for (uint64_t chunk = blockIdx.x; chunk < nChunks; chunk += gridDim.x) {
    if (threadIdx.x == 0) qCount = 0;
    __syncthreads();
    // ---- 1. probe: offsets and first bytes of all the lane's lines requested together --------
    uint64_t o[L], n[L];
    uint4 head[L][PC];
#pragma unroll
    for (uint32_t k = 0; k < L; ++k)
      spanOf(chunk * kEarlyChunk + uint64_t(k) * THREADS + threadIdx.x, o[k], n[k]);
#pragma unroll
    for (uint32_t k = 0; k < L; ++k)
#pragma unroll
      for (uint32_t j = 0; j < uint32_t(PC); ++j)
        head[k][j] = n[k] >= 16 * (j + 1) ? *reinterpret_cast<const uint4 *>(b.data + o[k] + 16 * j)
                                          : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (uint32_t k = 0; k < L; ++k) {
      const uint64_t line = chunk * kEarlyChunk + uint64_t(k) * THREADS + threadIdx.x;
      const bool valid = line < b.n;
      const uint8_t *p = b.data + o[k];
      WALK w;
      w.begin(c);
      bool alive = valid;
      if (valid && lead && !lookingAt(c, p, 0, n[k])) {
        b.result[line] = 0;
        if (b.start) b.start[line] = 0;
        if (b.end) b.end[line] = 0;
        alive = false;
      }
      if (alive) {
        if (n[k] >= 16) {
          const uint32_t have = earlyHave<PC>(n[k]);
#pragma unroll
          for (uint32_t g = 0; g < 2 * uint32_t(PC); ++g) {
            // past the probe proper only the survivors walk on, from the bytes already in registers
            if (g > 0 && !__builtin_amdgcn_ballot_w64(alive && 8 * g < have)) break;
            const uint4 &h = head[k][g >> 1];
            const uint32_t words[2] = {g & 1 ? h.z : h.x, g & 1 ? h.w : h.y};
#pragma unroll
            for (uint32_t i = 0; i < 8; ++i)
              if (alive && 8 * g < have)
                alive = w.step(tab, c, style, (words[i >> 2] >> (8 * (i & 3))) & 0xffu, 8 * g + i);
          }
          if (!alive || have == n[k]) {  // done within the probe; the others have bytes left
            store(line, w);
            alive = false;
          }
        } else {  // a short line: all of it, byte by byte
          for (uint64_t i = 0; i < n[k] && alive; ++i) alive = w.step(tab, c, style, p[i], i);
          store(line, w);
          alive = false;
        }
      }
      // survivors: one queue slot each, claimed per wave
      const uint64_t mask = __builtin_amdgcn_ballot_w64(alive);
      if (mask) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&qCount, uint32_t(__builtin_popcountll(mask)));
        base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
        if (alive) {
          const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32),
                                    __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
          queue[base + rank] = w.pack(uint32_t(k * THREADS + threadIdx.x));
        }
      }
    }
    __syncthreads();
    // ---- 2. drain: the survivors, dense again, walked to their end ---------------------------
    const uint32_t qn = qCount;
    for (uint32_t q = threadIdx.x; q < qn; q += THREADS) {
      const uint4 en = queue[q];
      WALK w;
      w.unpack(en);
      const uint64_t ln = chunk * kEarlyChunk + (en.x & 0xfffu);
      uint64_t oo, nl;
      spanOf(ln, oo, nl);
      if constexpr (LEAN_DRAIN) {
        // the lean walk without a branch per byte: a piece's 16 steps are selects under the
        // lane's `alive` flag (a lane that has met its pure dead end changes nothing any more),
        // the loop asks once per piece; positions in 32 bits (longer lines: the walk below)
        if (nl < (1ull << 32)) {
          uint32_t st = w.s, accS = w.accS, ms = uint32_t(w.matchStart), me = uint32_t(w.matchEnd);
          uint32_t pos = earlyHave<PC>(nl);
          const uint32_t n32 = uint32_t(nl);
          const uint8_t *p = b.data + oo;
          bool alive = true;
          auto lean = [&](uint32_t byte, uint32_t idx) {
            const uint32_t s2 = tab.next(st, byte);
            const bool leaves = st == c.init && s2 != st;
            const bool acc = s2 >= c.firstAccept;
            ms = alive && leaves ? idx : ms;
            accS = alive && acc ? s2 : accS;
            me = alive && acc ? idx + 1 : me;
            st = alive ? s2 : st;
            alive = alive && (acc || s2 >= c.nPureDead);
          };
          while (alive && pos + 16 <= n32) {
            const uint4 v = *reinterpret_cast<const uint4 *>(p + pos);
            const uint32_t words[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t k = 0; k < 16; ++k) lean((words[k >> 2] >> (8 * (k & 3))) & 0xffu, pos + k);
            pos += 16;
          }
          for (; alive && pos < n32; ++pos) lean(uint32_t(p[pos]), pos);
          w.s = st; w.accS = accS; w.matchStart = ms; w.matchEnd = me;
          store(ln, w);
          continue;
        }
      }
      walkBytes(b.data + oo, earlyHave<PC>(nl), nl,
                [&](uint32_t byte, uint64_t idx) { return w.step(tab, c, style, byte, idx); });
      store(ln, w);
    }
    __syncthreads();
  }
}

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

// =========================================================================================
// The hot path: fixed-stride lines, fused u8 table in LDS.
//
// Layout in LDS: [table nStates*256 B][result nStates*4 B].  One workgroup of 1024 threads
// (16 waves) per CU shares one copy of the table; each lane walks CHAINS independent lines
// (line = tile*1024*CHAINS + chain*1024 + thread) so that CHAINS ds_read_u8 are in flight per
// lane while each chain's own lookup->lookup dependency (~64+ cycles of LDS latency) resolves.
// Per input byte and chain: 1 VALU to form the LDS address ((state << 8) | byte),
// 1 ds_read_u8, and 2-5 VALU of style bookkeeping.  Styles Last and Full never leave the loop
// early (a pure dead end is absorbing - verified on the host - so walking on is a no-op),
// which keeps the wave uniform.  Early-exit styles freeze the lane's bookkeeping instead.
// =========================================================================================
constexpr int kFixedThreads = 1024;

template <int STYLE, bool POS, bool WANT_START>
struct ChainState {
  uint32_t s;        // current device state
  uint32_t accS;     // last accepting state seen (valid when endv != 0)
  uint32_t endv;     // idx+1 of the last accept (0 = none yet)
  uint32_t startv;   // idx at which the walk last escaped the initial state
  uint32_t wasInit;  // s == init before this step
  uint32_t live;     // early-exit styles: 0 once the reference loop would have left
};

template <int STYLE, bool POS, bool WANT_START>
__device__ __forceinline__ void stepChain(ChainState<STYLE, POS, WANT_START> &c,
                                          const uint8_t *__restrict__ tab, uint32_t byte,
                                          uint32_t idx, uint32_t init, uint32_t firstAccept,
                                          const int32_t *__restrict__ ldsRes) {
  const uint32_t sNew = tab[(c.s << 8) | byte];
  if (STYLE == kStyLast || STYLE == kStyFull) {
    if (POS && WANT_START) {
      const uint32_t isInit = (sNew == init);
      c.startv = (c.wasInit && !isInit) ? idx : c.startv;
      c.wasInit = isInit;
    }
    if (STYLE == kStyLast) {
      const bool acc = sNew >= firstAccept;
      c.accS = acc ? sNew : c.accS;
      c.endv = acc ? idx + 1 : c.endv;
    }
    c.s = sNew;
  } else {
    // Instant / First / Tangent: once the reference would `break`/`return`, stop updating.
    if (c.live) {
      if (POS && WANT_START) {
        const uint32_t isInit = (sNew == init);
        if (c.wasInit && !isInit) c.startv = idx;
        c.wasInit = isInit;
      }
      c.s = sNew;
      if (sNew >= firstAccept) {
        if (STYLE == kStyFirst && c.endv && ldsRes[sNew] != ldsRes[c.accS]) {
          c.live = 0;  // result changed: keep the previous accept (Matcher.h:457-460)
        } else {
          c.accS = sNew;
          c.endv = idx + 1;
          if (STYLE == kStyInstant) c.live = 0;
        }
      } else if (c.endv) {
        c.live = 0;  // First/Tangent: left the accepting run (Matcher.h:470-475)
      }
    }
  }
}

template <int STYLE, bool POS, bool WANT_START, int CHAINS>
__global__ void __launch_bounds__(kFixedThreads)
k_fixed(DevDfa d, Batch b, uint32_t lineLen, uint32_t startByte, uint32_t startState) {
  extern __shared__ __align__(16) uint8_t lds[];
  uint8_t *tab = lds;
  int32_t *ldsRes = reinterpret_cast<int32_t *>(lds + d.tableBytes);
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(d.table);
    uint4 *dst = reinterpret_cast<uint4 *>(tab);
    for (uint32_t i = threadIdx.x; i < d.tableBytes / 16; i += kFixedThreads) dst[i] = src[i];
    for (uint32_t i = threadIdx.x; i < d.nStates; i += kFixedThreads) ldsRes[i] = d.result[i];
  }
  __syncthreads();

  const uint32_t init = d.init;
  const uint32_t firstAccept = d.firstAccept;
  const uint64_t linesPerTile = uint64_t(kFixedThreads) * CHAINS;
  const uint64_t nTiles = (b.n + linesPerTile - 1) / linesPerTile;

  for (uint64_t tile = blockIdx.x; tile < nTiles; tile += gridDim.x) {
    ChainState<STYLE, POS, WANT_START> cs[CHAINS];
    const uint8_t *lp[CHAINS];
    uint64_t line[CHAINS];
    bool valid[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      line[c] = tile * linesPerTile + uint64_t(c) * kFixedThreads + threadIdx.x;
      valid[c] = line[c] < b.n;
      // out-of-range chains re-walk the last line and are not stored: keeps the wave uniform
      const uint64_t ln = valid[c] ? line[c] : b.n - 1;
      lp[c] = b.data + ln * b.stride;
      cs[c].s = startState;
      cs[c].accS = 0;
      cs[c].endv = 0;
      cs[c].startv = 0;
      cs[c].wasInit = (startState == init);
      cs[c].live = 1;
    }

    // 16 bytes per chain per round, next round's loads issued before this round's walk
    uint4 cur[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
      cur[c] = *reinterpret_cast<const uint4 *>(lp[c] + startByte);

    for (uint32_t off = startByte; off < lineLen; off += 16) {
      if constexpr (STYLE != kStyLast && STYLE != kStyFull) {
        // the early-exit styles: once the reference's loop has left every line this wave holds,
        // the rest of those lines is not read (a dense DFA under styInstant is done within its
        // first piece: SYN-256 on 4 KiB lines 1.3 -> 39 TB/s of line bytes, as k_generic already did)
        bool any = false;
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) any = any || cs[c].live != 0;
        if (!__builtin_amdgcn_ballot_w64(any)) break;
      }
      uint4 nxt[CHAINS];
      const bool more = off + 16 < lineLen;
      if (more) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c)
          nxt[c] = *reinterpret_cast<const uint4 *>(lp[c] + off + 16);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int c = 0; c < CHAINS; ++c) {
            const uint32_t word = k == 0 ? cur[c].x : k == 1 ? cur[c].y : k == 2 ? cur[c].z
                                                                                   : cur[c].w;
            const uint32_t byte = (word >> (8 * j)) & 0xffu;
            stepChain<STYLE, POS, WANT_START>(cs[c], tab, byte, off + 4 * k + j, init,
                                              firstAccept, ldsRes);
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
      if (!valid[c]) continue;
      int32_t r;
      uint32_t en;
      if (STYLE == kStyFull) {
        // result of the final state; end is the line length when it accepts (Matcher.h:463)
        r = cs[c].s >= firstAccept ? ldsRes[cs[c].s] : 0;
        en = lineLen;
      } else {
        r = cs[c].endv ? ldsRes[cs[c].accS] : 0;
        en = cs[c].endv;
      }
      b.result[line[c]] = r;
      if (POS) {
        if (b.end) b.end[line[c]] = r ? uint64_t(en) : 0;
        if (WANT_START && b.start) b.start[line[c]] = r ? uint64_t(cs[c].startv) : 0;
      }
    }
  }
}

// leader pre-pass for the fixed kernels: marks lines whose first leaderLen bytes do not
// match the fixed prefix (lookingAt / compareThrough, Matcher.h:333-360) by zeroing outputs.
__global__ void __launch_bounds__(256)
k_leader_filter(DevDfa d, Batch b) {
  __shared__ uint8_t eq[512];
  for (uint32_t i = threadIdx.x; i < 128; i += 256)
    reinterpret_cast<uint32_t *>(eq)[i] = reinterpret_cast<const uint32_t *>(d.equivLeader)[i];
  __syncthreads();
  const uint8_t *leader = eq + 256;
  const uint64_t step = uint64_t(gridDim.x) * 256;
  for (uint64_t line = uint64_t(blockIdx.x) * 256 + threadIdx.x; line < b.n; line += step) {
    const uint8_t *p = b.data + line * b.stride;
    bool ok = d.leaderLen <= b.stride;
    for (uint32_t k = 0; ok && k < d.leaderLen; ++k) ok = leader[k] == eq[p[k]];
    if (!ok) {
      b.result[line] = 0;
      if (b.start) b.start[line] = 0;
      if (b.end) b.end[line] = 0;
    }
  }
}

#include "k_stream.h"
#include "k_stream_lean.h"
#include "k_stream_multi.h"
#include "k_ragged.h"

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

// =========================================================================================
// k_style_blocks: check / match with the EARLY-EXIT styles (styInstant, styFirst, styTangent;
// include/Matcher.h:382-403, :443-479), no leader, over the same two phases as k_matchall_blocks.
// What those styles report is decided by the FIRST run of accepting positions:
//   styInstant : the first accepting position a0 - result of its state, end = a0 + 1;
//   styTangent : the run of consecutive accepting positions from a0 - result of its last state,
//                end behind it (the loop leaves at the first non-accepting position behind one);
//   styFirst   : the same run cut where the result changes (:457-460) - result of a0's state;
//   start      : the last "left the initial state" position up to and including the position
//                at which the loop left (the update at :446-451 precedes the tests).
// Phase A (mabWalkBlock) walks a block without looking; phase B reads the masks: first set bit,
// first clear bit behind it, and only for styFirst the results along the run.  A lane whose loop
// has left stops taking blocks, a wave whose lanes all have stops reading: a dense DFA is done
// within its first block, where k_fixed walked every line to its end.  Requires what
// k_matchall_blocks requires (absorbing pure dead ends: nothing accepts past one).
// =========================================================================================
template <int KIND, int THREADS, int W, bool POS>
__global__ void __launch_bounds__(THREADS)
k_style_blocks(DevDfa d, Batch b, int style) {
  constexpr uint32_t kPos = 64 / W;
  constexpr uint32_t kPerWord = 4 / W;
  extern __shared__ __align__(16) uint8_t lds[];
  const Tab<KIND> tab = stageTab<KIND, THREADS>(d, lds);
  LaneCtx c{lds, lds + 256, resOf<KIND>(d, lds), d.init, d.leaderNext, d.nPureDead, d.firstAccept,
            d.leaderLen};
  uint32_t *stage = reinterpret_cast<uint32_t *>(lds + 512 + ((ldsTableBytes<KIND>(d) + 15) & ~size_t(15)));
  const uint8_t *stageBytes = reinterpret_cast<const uint8_t *>(stage);
  const int32_t *ldsRes = reinterpret_cast<const int32_t *>(lds + 512 + ((tableOnlyBytes<KIND>(d) + 15) & ~size_t(15)));
  const bool tableAt512 = uint32_t(reinterpret_cast<uintptr_t>(lds)) == 0u;
  const uint8_t *bufEnd = b.data + (b.offsets ? b.offsets[b.n] : b.n * b.stride);
  const int32_t initRes = d.init >= d.firstAccept ? ldsRes[d.init] : 0;
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
    uint32_t s = c.init;
    // mode 0: no accepting position yet; 1: inside the first run; 2: the loop has left
    uint32_t mode = 0;
    int32_t result = n ? 0 : initRes;  // an accepting initial state only counts for empty input
    int32_t r0 = 0, prevR = 0;
    uint64_t matchStart = 0, startOut = 0, curEnd = 0;
    for (uint64_t base = 0; base < n && mode != 2; base += kPos) {
      const uint64_t wasInit = s == c.init ? 1u : 0u;
      uint64_t acc = 0, ini = 0;
      const uint32_t cnt = mabWalkBlock<KIND, THREADS, W>(tab, c, p + base, n - base,
                                                          bufEnd - (p + base), s, stage,
                                                          tableAt512, acc, ini);
      const uint64_t valid = cnt >= 64 ? ~0ull : (1ull << cnt) - 1;
      const uint64_t esc = (((ini << 1) | wasInit) & ~ini) & valid;
      auto resultAt = [&](uint32_t i) -> int32_t {
        const uint8_t *slot = stageBytes + (((i / kPerWord) * THREADS + threadIdx.x) << 2) +
                              W * (i % kPerWord);
        const uint32_t si = W == 1 ? uint32_t(*slot) : uint32_t(*reinterpret_cast<const uint16_t *>(slot));
        return ldsRes[si];
      };
      // the last escape at or below position i of this block, else the one carried in
      auto startAt = [&](uint32_t i) -> uint64_t {
        const uint64_t m = esc & ((2ull << i) - 1);
        return m ? base + 63 - uint32_t(__builtin_clzll(m)) : matchStart;
      };
      if (style == kStyLast || style == kStyFull) {
        // the whole-line styles (fixed strides the streaming kernels do not take): styLast wants the
        // LAST accepting position - the top bit of the mask, its state read back while the block is
        // still staged; styFull only the final state
        if (style == kStyLast && acc) {
          const uint32_t i = 63 - uint32_t(__builtin_clzll(acc));
          prevR = resultAt(i);
          curEnd = base + i + 1;
        }
        if (esc) matchStart = base + 63 - uint32_t(__builtin_clzll(esc));
        continue;
      }
      uint32_t q = 0;  // first position of this block the run still has to look at
      if (mode == 0 && acc) {
        const uint32_t a0 = uint32_t(__builtin_ctzll(acc));
        r0 = prevR = resultAt(a0);
        curEnd = base + a0 + 1;
        if (style == kStyInstant) {
          result = r0;
          if (POS) startOut = startAt(a0);
          mode = 2;
        } else {
          mode = 1;
          q = a0 + 1;
        }
      }
      if (mode == 1) {
        // the run goes on over accepting positions from q; zf = the first one that is not
        const uint64_t clear = ~acc & valid & (q >= 64 ? 0ull : ~0ull << q);
        const uint32_t zf = clear ? uint32_t(__builtin_ctzll(clear)) : cnt;
        uint32_t stop = 0xffffffffu;
        if (style == kStyFirst) {
          uint32_t i = q;
          for (; i < zf; ++i) {
            if (resultAt(i) != r0) break;  // another result: the loop leaves, keeping the first (:457-460)
            curEnd = base + i + 1;
          }
          if (i < zf) stop = i;
          else if (zf < cnt) stop = zf;
          prevR = r0;
        } else {  // styTangent: the result of the run's last accepting position
          if (zf > q) {
            prevR = resultAt(zf - 1);
            curEnd = base + zf;
          }
          if (zf < cnt) stop = zf;
        }
        if (stop != 0xffffffffu) {
          result = prevR;
          if (POS) startOut = startAt(stop);
          mode = 2;
        }
      }
      if (esc) matchStart = base + 63 - uint32_t(__builtin_clzll(esc));
    }
    if (style == kStyLast) {
      if (n) result = prevR;
      startOut = matchStart;
    } else if (style == kStyFull) {
      if (n) result = s >= d.firstAccept ? ldsRes[s] : 0;
      startOut = matchStart;
      curEnd = n;  // end is the line length when the final state accepts (Matcher.h:463)
    } else if (mode == 1) {  // the line ended inside the run
      result = prevR;
      startOut = matchStart;
    }
    b.result[line] = result;
    if (POS) {
      if (b.start) b.start[line] = result ? startOut : 0;
      if (b.end) b.end[line] = result ? curEnd : 0;
    }
  }
}

// StatefulMatcher::advance (include/Matcher.h:770-792, lib/Matcher.cpp:106-158) over a whole
// chunk per line: state[line] is the matcher's state_ (a device state index; REDGPU_STATE_INITIAL
// = a freshly constructed matcher, lib/Matcher.cpp:113-136), advanced by every byte of the
// chunk with no early exit and no style rules, then stored back; result[line] = result() after
// the last byte (= the state's result; for an empty chunk the current state's).
template <int KIND, int kThreads>
__global__ void __launch_bounds__(kThreads)
k_advance(DevDfa d, Batch b, uint32_t *state) {
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
    uint32_t s = state[line];
    if (s >= d.nStates) s = d.init;  // REDGPU_STATE_INITIAL (and any token that is not ours)
    if (d.deadAbsorbing && d.earlyDeath) {
      // an absorbing dead end stays put: stop reading (anchored patterns die in their first bytes)
      walkBytes(p, 0, n, [&](uint32_t byte, uint64_t) {
        s = tab.next(s, byte);
        return s >= d.nPureDead;
      });
    } else {
      walkAllBytes(p, n, [&](uint32_t byte, uint64_t) { s = tab.next(s, byte); });
    }
    state[line] = s;
    b.result[line] = c.resultOf(s);
  }
}

// include/Matcher.h:643-706 replaceCore.  out == nullptr: only count and measure.
// Returns the number of replacements; outLen = length of the rewritten line.
template <class T>
__device__ uint64_t replaceLane(const T &tab, const LaneCtx &c, const uint8_t *p, uint64_t n,
                                int style, bool lead, const uint8_t *repl, uint64_t replLen,
                                uint64_t max, uint8_t *out, uint64_t &outLen) {
  uint64_t cnt = 0, w = 0, in = 0;
  while (in < n) {
    if (cnt >= max) {
      if (out)
        for (uint64_t k = in; k < n; ++k) out[w + (k - in)] = p[k];
      w += n - in;
      break;
    }
    uint64_t found = ~0ull;
    bool toEnd = false;  // the attempt read p[in..n) to its end
    if (!lead || lookingAt(c, p, in, n)) {
      uint32_t s = c.init;
      int32_t prev = 0;
      uint64_t q = in;
      for (; q < n; ++q) {
        s = tab.next(s, p[q]);
        if (s >= c.firstAccept) {
          const int32_t r = c.res[s];
          if (style == kStyFirst) {
            if (prev && r != prev) break;
            prev = r;
          }
          found = q;
          if (style == kStyInstant) break;
        } else {
          if (style == kStyFull) found = ~0ull;
          if (((style == kStyFirst || style == kStyTangent) && found != ~0ull) ||
              s < c.nPureDead)
            break;
        }
      }
      toEnd = q == n;
    }
    if (found == ~0ull && toEnd && c.suffixClosed && !lead) {
      // L = SIGMA* L: nothing matched on p[in..n), so nothing can at any later position (they read
      // suffixes of it) - the rest of the line is copied as the reference's loop would, byte by byte
      if (out)
        for (uint64_t k = in; k < n; ++k) out[w + (k - in)] = p[k];
      w += n - in;
      break;
    }
    if (found != ~0ull) {
      if (out)
        for (uint64_t k = 0; k < replLen; ++k) out[w + k] = repl[k];
      w += replLen;
      in = found + 1;
      ++cnt;
    } else {
      if (out) out[w] = p[in];
      ++w;
      ++in;
    }
  }
  outLen = w;
  return cnt;
}

// pass 1 (out == nullptr): counts[line], outLens[line].  pass 2: writes line i's rewritten
// bytes at out + outOffsets[i] when outOffsets[i + 1] <= outCap.
template <int KIND, int kThreads>
__global__ void __launch_bounds__(kThreads)
k_replace(DevDfa d, Batch b, int style, int lead, const uint8_t *repl, uint64_t replLen,
          uint64_t max, uint64_t *counts, uint64_t *outLens, const uint64_t *outOffsets,
          uint8_t *out, uint64_t outCap) {
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
    uint64_t len = 0;
    if (!out) {
      counts[line] = replaceLane(tab, c, p, n, style, lead != 0, repl, replLen, max, nullptr, len);
      outLens[line] = len;
    } else if (outOffsets[line + 1] <= outCap) {
      replaceLane(tab, c, p, n, style, lead != 0, repl, replLen, max, out + outOffsets[line], len);
    }
  }
}

// exclusive scan of lens[n] into offs[n + 1] (offs[n] = total): per-1024 partial sums, one
// workgroup over the partials, then the fill
__global__ void __launch_bounds__(256)
k_scan_partials(const uint64_t *lens, uint64_t n, uint64_t *partials) {
  __shared__ uint64_t ws[4];
  const uint64_t base = uint64_t(blockIdx.x) * 1024;
  uint64_t v = 0;
  for (uint32_t k = 0; k < 4; ++k) {
    const uint64_t i = base + k * 256 + threadIdx.x;
    v += i < n ? lens[i] : 0;
  }
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ void __launch_bounds__(1024)
k_scan_tops(uint64_t *partials, uint64_t nPart) {
  __shared__ uint64_t part[1024];
  const uint64_t per = (nPart + 1023) / 1024;
  const uint64_t lo = uint64_t(threadIdx.x) * per;
  const uint64_t hi = lo + per < nPart ? lo + per : nPart;
  uint64_t sum = 0;
  for (uint64_t i = lo; i < hi; ++i) sum += partials[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t run = 0;
    for (int i = 0; i < 1024; ++i) { const uint64_t v = part[i]; part[i] = run; run += v; }
  }
  __syncthreads();
  uint64_t run = part[threadIdx.x];
  for (uint64_t i = lo; i < hi; ++i) { const uint64_t v = partials[i]; partials[i] = run; run += v; }
}

__global__ void __launch_bounds__(256)
k_scan_fill(const uint64_t *lens, uint64_t n, const uint64_t *partials, uint64_t *offs) {
  // one wave per 256 elements would do; keep it simple: thread 0 of each 64-lane group walks
  // its 64 elements after a wave-level prefix
  const uint64_t base = uint64_t(blockIdx.x) * 1024;
  __shared__ uint64_t ws[4];
  uint64_t carry = partials[blockIdx.x];
  for (uint32_t k = 0; k < 4; ++k) {
    const uint64_t i = base + k * 256 + threadIdx.x;
    const uint64_t v = i < n ? lens[i] : 0;
    uint64_t incl = v;
    for (int o = 1; o < 64; o <<= 1) {
      const uint64_t u = __shfl_up(incl, o);
      if ((threadIdx.x & 63) >= uint32_t(o)) incl += u;
    }
    if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t wb = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) wb += ws[w];
    if (i < n) offs[i] = carry + wb + incl - v;
    if (i + 1 == n) offs[n] = carry + wb + incl;
    carry += ws[0] + ws[1] + ws[2] + ws[3];
    __syncthreads();
  }
}

// Visit histogram for redgpu_dfa_tune: the anchored walk of match<styLast,false> over every
// line of a SAMPLE, hist[state] += 1 per byte consumed.  A profiling pass, not a hot path:
// plain global atomics.
template <int KIND, int kThreads>
__global__ void __launch_bounds__(kThreads)
k_visits(DevDfa d, Batch b, uint32_t *hist) {
  extern __shared__ __align__(16) uint8_t lds[];
  const Tab<KIND> tab = stageTab<KIND, kThreads>(d, lds);
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
    uint32_t s = d.init;
    walkBytes(p, 0, n, [&](uint32_t byte, uint64_t) {
      s = tab.next(s, byte);
      atomicAdd(&hist[s], 1u);
      return s >= d.nPureDead;
    });
  }
}

// bench.py's "bytes actually walked": what the loop of match<styLast,lead> (include/Matcher.h:
// 424-479) consumes per line - nothing when the leader peek fails, else every byte up to and
// including the one that reaches a pure dead end.  One atomic per wave.
template <int KIND, int kThreads>
__global__ void __launch_bounds__(kThreads)
k_walked(DevDfa d, Batch b, int lead, unsigned long long *walked) {
  extern __shared__ __align__(16) uint8_t lds[];
  const Tab<KIND> tab = stageTab<KIND, kThreads>(d, lds);
  LaneCtx c;
  c.eq = lds;
  c.leader = lds + 256;
  c.res = resOf<KIND>(d, lds);
  c.init = d.init; c.leaderNext = d.leaderNext; c.nPureDead = d.nPureDead;
  c.firstAccept = d.firstAccept; c.leaderLen = d.leaderLen;
  const uint64_t step = uint64_t(gridDim.x) * kThreads;
  unsigned long long mine = 0;
  for (uint64_t line = uint64_t(blockIdx.x) * kThreads + threadIdx.x; line < b.n; line += step) {
    const uint8_t *p;
    uint64_t n;
    if (b.offsets) {
      const uint64_t o = b.offsets[line];
      p = b.data + o;
      n = b.offsets[line + 1] - o;
      n = n >= b.stride ? n - b.stride : 0;
    } else {
      p = b.data + line * b.stride;
      n = b.stride;
    }
    if (lead && !lookingAt(c, p, 0, n)) continue;
    uint32_t s = d.init;
    walkBytes(p, 0, n, [&](uint32_t byte, uint64_t) {
      s = tab.next(s, byte);
      ++mine;
      return s >= d.nPureDead;
    });
  }
  for (int o = 32; o; o >>= 1) mine += __shfl_xor(mine, o);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(walked, mine);
}

// ---- line splitting on the device ------------------------------------
// The rule is sampleLines' (lib/Util.cpp:109-130): a line is [start, position of the delimiter),
// the next one starts after the delimiter, and bytes after the last delimiter are not a line.
// Output is the offsets[n+1] array the ragged verbs take, line i = [offsets[i], offsets[i+1])
// INCLUDING its delimiter - the verbs are then called with stride = 1 (one trailing byte to
// drop).  Three passes: per-chunk delimiter counts, an exclusive scan of the counts, and the
// scatter; chunk = kSplitChunk bytes per workgroup.
constexpr uint32_t kSplitChunk = 16384;
constexpr int kSplitThreads = 256;

__device__ __forceinline__ uint32_t delimMask16(const uint4 v, uint32_t delim) {
  // bit k set <=> byte k of the 16 equals delim
  uint32_t m = 0;
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k)
      m |= (((w[i] >> (8 * k)) & 0xffu) == delim ? 1u : 0u) << (4 * i + k);
  return m;
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

__global__ void __launch_bounds__(kSplitThreads)
k_split_count(const uint8_t *data, uint64_t len, uint32_t delim, uint32_t *counts) {
  __shared__ uint32_t waveSum[kSplitThreads / 64];
  const uint64_t base = uint64_t(blockIdx.x) * kSplitChunk;
  uint32_t c = 0;
  for (uint32_t off = threadIdx.x * 16; off < kSplitChunk; off += kSplitThreads * 16)
    c += __popc(chunkPieceMask(data, len, base + off, delim));
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
  __syncthreads();
  for (uint32_t off0 = 0; off0 < kSplitChunk; off0 += kSplitThreads * 16) {
    const uint64_t pos = base + off0 + threadIdx.x * 16;
    const uint32_t m = chunkPieceMask(data, len, pos, delim);
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

template <class K>
hipError_t setLds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes));
}

#include "k_chunk.h"

template <int KIND>
hipError_t launchGeneric(const DevDfa &d, const Batch &b, int verb, int style, int lead,
                         const LaunchCfg &cfg, hipStream_t stream) {
  constexpr bool kLds = Tab<KIND>::kInLds || KIND == REDGPU_TAB_HOT_ROWS;
  constexpr int kThreads = kLds ? 1024 : 256;
  const size_t ldsBytes = 512 + ldsTableBytes<KIND>(d);
  uint64_t blocks = (b.n + kThreads - 1) / kThreads;
  // an LDS-resident table is re-staged per block: keep the grid near one wave of blocks
  const uint64_t perCu = kLds ? (ldsBytes <= 80 * 1024 ? 2 : 1) : 8;
  const uint64_t cap = uint64_t(cfg.numCUs) * perCu;
  Batch sb = b;
  if (!kLds && (verb == kCheck || verb == kMatch)) {
    // a table in L2 and fewer lines than lanes (BASELINE configs[4]: 65,536 x 64 KiB): spread
    // the lines over more waves (measured: profiles/r02_spread_syn4k.log)
    static const int forced = [] { const char *e = getenv("REDGPU_GENERIC_SPREAD"); return e ? atoi(e) : 0; }();
    uint32_t spread = 1;
    while (spread < 8 && b.n * (spread * 2) <= cap * kThreads) spread *= 2;
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8) spread = uint32_t(forced);
    else spread = 1;  // default until measured
    sb.spread = spread;
    blocks = (b.n * spread + kThreads - 1) / kThreads;
  }
  if (blocks > cap) blocks = cap;
  if (blocks == 0) blocks = 1;
#define GEN_LAUNCH(V)                                                                        \
  do {                                                                                       \
    hipError_t e_ = setLds(k_generic<KIND, kThreads, V>, ldsBytes);                          \
    if (e_ != hipSuccess) return e_;                                                         \
    hipLaunchKernelGGL((k_generic<KIND, kThreads, V>), dim3(uint32_t(blocks)), dim3(kThreads), \
                       ldsBytes, stream, d, pb, style, lead);                                \
  } while (0)
  // scan / search over a DFA with at most 4 start bytes: mark the candidates, visit only those
  if ((verb == kScan || verb == kSearch) && scanMarkable(d, lead) && !cfg.forceGeneric) {
    // table, bitmap, candidate list + per-line slots (k_scan_marked's spread form), flag table
    const size_t markLds = 512 + ((ldsTableBytes<KIND>(d) + 15) & ~size_t(15)) + kMarkBytes +
                           size_t(kScanThreads) * (6 * 4 + 8) + 32 + 256;
    if (markLds <= 158 * 1024) {
      hipError_t e_ = verb == kScan ? setLds(k_scan_marked<KIND, kScanThreads, kScan>, markLds)
                                    : setLds(k_scan_marked<KIND, kScanThreads, kSearch>, markLds);
      if (e_ != hipSuccess) return e_;
      uint64_t mb = (b.n + kScanThreads - 1) / kScanThreads;
      const uint64_t fit = (158 * 1024) / markLds;  // workgroups per CU by LDS
      const uint64_t mcap = uint64_t(cfg.numCUs) * (fit < 8 ? fit : 8);
      if (mb > mcap) mb = mcap;
      if (verb == kScan)
        hipLaunchKernelGGL((k_scan_marked<KIND, kScanThreads, kScan>), dim3(uint32_t(mb)),
                           dim3(kScanThreads), markLds, stream, d, b, style, lead);
      else
        hipLaunchKernelGGL((k_scan_marked<KIND, kScanThreads, kSearch>), dim3(uint32_t(mb)),
                           dim3(kScanThreads), markLds, stream, d, b, style, lead);
      return hipGetLastError();
    }
  }
  // whole-line walks over ragged lines profit from the length bucketing; walks that die in
  // their first bytes (early-death DFAs under check / match) do not care how long the line is
  Batch pb = sb;
  if (b.offsets && !(d.earlyDeath && (verb == kCheck || verb == kMatch))) {
    hipError_t pe = prepareRagged(sb, cfg, stream, false, pb);
    if (pe != hipSuccess) return pe;
  }
  switch (verb) {
  case kCheck: GEN_LAUNCH(kCheck); break;
  case kScan: GEN_LAUNCH(kScan); break;
  case kSearch: GEN_LAUNCH(kSearch); break;
  default: GEN_LAUNCH(kMatch); break;
  }
#undef GEN_LAUNCH
  return hipGetLastError();
}

template <int KIND, class WALK, int LPL, int WPS, int PC = 1, int THREADS = 512, bool LEAN_DRAIN = false>
hipError_t launchEarlyV(const DevDfa &d, const Batch &b, int style, int lead, const LaunchCfg &cfg,
                        hipStream_t stream) {
  const size_t tabBytes = (tableOnlyBytes<KIND>(d) + 15) & ~size_t(15);
  const size_t ldsBytes = 512 + tabBytes + size_t(THREADS) * LPL * 16;
  hipError_t e = setLds(k_early<KIND, WALK, LPL, WPS, PC, THREADS, LEAN_DRAIN>, ldsBytes);
  if (e != hipSuccess) return e;
  // as many workgroups per CU as LDS and the register budget allow (their probe / drain phases
  // overlap each other's memory round trips)
  uint64_t perCu = (160 * 1024) / (ldsBytes + 256);
  const uint64_t byRegs = uint64_t(WPS) * 4 / (THREADS / 64);
  perCu = perCu > byRegs ? byRegs : perCu;
  if (perCu < 1) perCu = 1;
  const uint64_t chunk = uint64_t(THREADS) * LPL;
  const uint64_t chunks = (b.n + chunk - 1) / chunk;
  uint64_t blocks = uint64_t(cfg.numCUs) * perCu;
  if (blocks > chunks) blocks = chunks;
  hipLaunchKernelGGL((k_early<KIND, WALK, LPL, WPS, PC, THREADS, LEAN_DRAIN>), dim3(uint32_t(blocks)), dim3(THREADS),
                     ldsBytes, stream, d, b, style, lead);
  return hipGetLastError();
}

template <int KIND>
hipError_t launchEarlyK(const DevDfa &d, const Batch &b, int verb, int style, int lead,
                        const LaunchCfg &cfg, hipStream_t stream) {
  if (verb == kCheck) return launchEarlyV<KIND, CheckWalk, 2, 4>(d, b, style, 0, cfg, stream);
  if (style != kStyLast) return launchEarlyV<KIND, AnyWalk, 2, 4>(d, b, style, lead, cfg, stream);
  // (the drain without a branch per byte: configs[3] 350 -> 337 us, scripts/gpu_run37.sh)
  return launchEarlyV<KIND, LastWalk, 2, 6, 1, 512, true>(d, b, style, lead, cfg, stream);
}

template <int KIND>
hipError_t launchCollectK(const DevDfa &d, const Batch &b, uint64_t cap, uint64_t *counts,
                          const LaunchCfg &cfg, hipStream_t stream) {
  constexpr bool kLds = Tab<KIND>::kInLds || KIND == REDGPU_TAB_HOT_ROWS;
  constexpr int kThreads = kLds ? 1024 : 256;
  const size_t ldsBytes = 512 + ldsTableBytes<KIND>(d);
  hipError_t e = setLds(k_collect<KIND, kThreads>, ldsBytes);
  if (e != hipSuccess) return e;
  uint64_t blocks = (b.n + kThreads - 1) / kThreads;
  const uint64_t perCu = kLds ? (ldsBytes <= 80 * 1024 ? 2 : 1) : 8;
  const uint64_t capBlocks = uint64_t(cfg.numCUs) * perCu;
  if (blocks > capBlocks) blocks = capBlocks;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL((k_collect<KIND, kThreads>), dim3(uint32_t(blocks)), dim3(kThreads), ldsBytes,
                     stream, d, b, cap, counts);
  return hipGetLastError();
}

template <int KIND>
hipError_t launchMatchAllK(const DevDfa &d, const Batch &b, uint64_t cap, uint64_t *counts,
                           int lead, const LaunchCfg &cfg, hipStream_t stream) {
  constexpr bool kLds = Tab<KIND>::kInLds || KIND == REDGPU_TAB_HOT_ROWS;
  constexpr int kThreads = kLds ? 1024 : 256;
  if constexpr (Tab<KIND>::kInLds) {
    // the block-wise form: table + results + 64 bytes of staged states per lane in LDS
    const size_t tab = 512 + ((ldsTableBytes<KIND>(d) + 15) & ~size_t(15));
    if (!cfg.forceGeneric && d.deadAbsorbing && resStaged<KIND>(d) && d.nStates <= 65535 &&
        tab + 512 * 64 + 256 <= size_t(160) * 1024) {
      // 1024- or 512-thread workgroups, whichever keeps more lanes resident on a CU
      auto resident = [&](uint64_t threads) -> uint64_t {
        uint64_t wgs = (size_t(160) * 1024) / (tab + threads * 64 + 256);
        if (wgs > 2048 / threads) wgs = 2048 / threads;
        return wgs;
      };
      const bool big = resident(1024) * 1024 >= resident(512) * 512;
      const int threads = big ? 1024 : 512;
      const size_t ldsBytes = tab + size_t(threads) * 64;
      uint64_t blocks = (b.n + threads - 1) / threads;
      const uint64_t perCu = resident(uint64_t(threads));
      if (blocks > uint64_t(cfg.numCUs) * perCu) blocks = uint64_t(cfg.numCUs) * perCu;
#define MAB_LAUNCH(T, W)                                                                     \
  do {                                                                                         \
    hipError_t e2 = setLds(k_matchall_blocks<KIND, T, W>, ldsBytes);                          \
    if (e2 != hipSuccess) return e2;                                                           \
    hipLaunchKernelGGL((k_matchall_blocks<KIND, T, W>), dim3(uint32_t(blocks)), dim3(T),      \
                       ldsBytes, stream, d, b, cap, counts, lead);                             \
  } while (0)
      if (d.nStates <= 256) { if (big) MAB_LAUNCH(1024, 1); else MAB_LAUNCH(512, 1); }
      else { if (big) MAB_LAUNCH(1024, 2); else MAB_LAUNCH(512, 2); }
#undef MAB_LAUNCH
      return hipGetLastError();
    }
  }
  const size_t ldsBytes = 512 + ldsTableBytes<KIND>(d);
  hipError_t e = setLds(k_matchall<KIND, kThreads>, ldsBytes);
  if (e != hipSuccess) return e;
  uint64_t blocks = (b.n + kThreads - 1) / kThreads;
  const uint64_t perCu = kLds ? (ldsBytes <= 80 * 1024 ? 2 : 1) : 8;
  const uint64_t capBlocks = uint64_t(cfg.numCUs) * perCu;
  if (blocks > capBlocks) blocks = capBlocks;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL((k_matchall<KIND, kThreads>), dim3(uint32_t(blocks)), dim3(kThreads),
                     ldsBytes, stream, d, b, cap, counts, lead);
  return hipGetLastError();
}

// k_style_blocks: same residency rule as k_matchall_blocks; *taken = false when the DFA / batch
// does not qualify (the caller goes on to the other kernels)
template <int KIND>
hipError_t launchStyleBlocksK(const DevDfa &d, const Batch &b, int style, bool pos,
                              const LaunchCfg &cfg, hipStream_t stream, bool *taken) {
  *taken = false;
  if constexpr (Tab<KIND>::kInLds) {
    const size_t tab = 512 + ((ldsTableBytes<KIND>(d) + 15) & ~size_t(15));
    if (!d.deadAbsorbing || !resStaged<KIND>(d) || d.nStates > 65535 ||
        tab + 512 * 64 + 256 > size_t(160) * 1024)
      return hipSuccess;
    auto resident = [&](uint64_t threads) -> uint64_t {
      uint64_t wgs = (size_t(160) * 1024) / (tab + threads * 64 + 256);
      if (wgs > 2048 / threads) wgs = 2048 / threads;
      return wgs;
    };
    const bool big = resident(1024) * 1024 >= resident(512) * 512;
    const int threads = big ? 1024 : 512;
    const size_t ldsBytes = tab + size_t(threads) * 64;
    uint64_t blocks = (b.n + threads - 1) / threads;
    const uint64_t perCu = resident(uint64_t(threads));
    if (blocks > uint64_t(cfg.numCUs) * perCu) blocks = uint64_t(cfg.numCUs) * perCu;
#define SB_LAUNCH(T, W, P)                                                                   \
  do {                                                                                         \
    hipError_t e2 = setLds(k_style_blocks<KIND, T, W, P>, ldsBytes);                          \
    if (e2 != hipSuccess) return e2;                                                           \
    hipLaunchKernelGGL((k_style_blocks<KIND, T, W, P>), dim3(uint32_t(blocks)), dim3(T),      \
                       ldsBytes, stream, d, b, style);                                         \
  } while (0)
#define SB_POS(T, W) do { if (pos) SB_LAUNCH(T, W, true); else SB_LAUNCH(T, W, false); } while (0)
    if (d.nStates <= 256) { if (big) SB_POS(1024, 1); else SB_POS(512, 1); }
    else { if (big) SB_POS(1024, 2); else SB_POS(512, 2); }
#undef SB_POS
#undef SB_LAUNCH
    *taken = true;
    return hipGetLastError();
  }
  return hipSuccess;
}

template <int KIND>
hipError_t launchVisitsK(const DevDfa &d, const Batch &b, uint32_t *hist, const LaunchCfg &cfg,
                         hipStream_t stream) {
  constexpr bool kLds = Tab<KIND>::kInLds || KIND == REDGPU_TAB_HOT_ROWS;
  constexpr int kThreads = kLds ? 1024 : 256;
  const size_t ldsBytes = 512 + ldsTableBytes<KIND>(d);
  hipError_t e = setLds(k_visits<KIND, kThreads>, ldsBytes);
  if (e != hipSuccess) return e;
  uint64_t blocks = (b.n + kThreads - 1) / kThreads;
  const uint64_t perCu = kLds ? (ldsBytes <= 80 * 1024 ? 2 : 1) : 8;
  const uint64_t capBlocks = uint64_t(cfg.numCUs) * perCu;
  if (blocks > capBlocks) blocks = capBlocks;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL((k_visits<KIND, kThreads>), dim3(uint32_t(blocks)), dim3(kThreads), ldsBytes,
                     stream, d, b, hist);
  return hipGetLastError();
}

template <int KIND>
hipError_t launchAdvanceK(const DevDfa &d, const Batch &b, uint32_t *state, const LaunchCfg &cfg,
                          hipStream_t stream) {
  constexpr bool kLds = Tab<KIND>::kInLds || KIND == REDGPU_TAB_HOT_ROWS;
  constexpr int kThreads = kLds ? 1024 : 256;
  const size_t ldsBytes = 512 + ldsTableBytes<KIND>(d);
  hipError_t e = setLds(k_advance<KIND, kThreads>, ldsBytes);
  if (e != hipSuccess) return e;
  uint64_t blocks = (b.n + kThreads - 1) / kThreads;
  const uint64_t perCu = kLds ? (ldsBytes <= 80 * 1024 ? 2 : 1) : 8;
  const uint64_t capBlocks = uint64_t(cfg.numCUs) * perCu;
  if (blocks > capBlocks) blocks = capBlocks;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL((k_advance<KIND, kThreads>), dim3(uint32_t(blocks)), dim3(kThreads),
                     ldsBytes, stream, d, b, state);
  return hipGetLastError();
}

template <int STYLE, bool POS, bool WANT_START, int CHAINS>
hipError_t launchFixedT(const DevDfa &d, const Batch &b, uint32_t startByte,
                        uint32_t startState, const LaunchCfg &cfg, hipStream_t stream) {
  auto kern = k_fixed<STYLE, POS, WANT_START, CHAINS>;
  const size_t ldsBytes = size_t(d.tableBytes) + size_t(d.nStates) * 4;
  hipError_t e = setLds(kern, ldsBytes);
  if (e != hipSuccess) return e;
  const uint64_t linesPerTile = uint64_t(kFixedThreads) * CHAINS;
  uint64_t tiles = (b.n + linesPerTile - 1) / linesPerTile;
  uint64_t blocks = tiles < uint64_t(cfg.numCUs) ? tiles : uint64_t(cfg.numCUs);
  hipLaunchKernelGGL(kern, dim3(uint32_t(blocks)), dim3(kFixedThreads), ldsBytes, stream, d, b,
                     uint32_t(b.stride), startByte, startState);
  return hipGetLastError();
}

template <int STYLE, bool POS, bool WANT_START>
hipError_t launchFixedC(const DevDfa &d, const Batch &b, uint32_t startByte,
                        uint32_t startState, const LaunchCfg &cfg, hipStream_t stream) {
  // enough chains per lane to give every CU one full tile; small batches use fewer chains
  const uint64_t perCu = b.n / (uint64_t(cfg.numCUs) * kFixedThreads);
  if (perCu >= 4)
    return launchFixedT<STYLE, POS, WANT_START, 4>(d, b, startByte, startState, cfg, stream);
  if (perCu >= 2)
    return launchFixedT<STYLE, POS, WANT_START, 2>(d, b, startByte, startState, cfg, stream);
  return launchFixedT<STYLE, POS, WANT_START, 1>(d, b, startByte, startState, cfg, stream);
}

template <bool POS, bool WANT_START>
hipError_t launchFixedS(int style, const DevDfa &d, const Batch &b, uint32_t startByte,
                        uint32_t startState, const LaunchCfg &cfg, hipStream_t stream) {
  switch (style) {
  case kStyInstant:
    return launchFixedC<kStyInstant, POS, WANT_START>(d, b, startByte, startState, cfg, stream);
  case kStyFirst:
    return launchFixedC<kStyFirst, POS, WANT_START>(d, b, startByte, startState, cfg, stream);
  case kStyTangent:
    return launchFixedC<kStyTangent, POS, WANT_START>(d, b, startByte, startState, cfg, stream);
  case kStyLast:
    return launchFixedC<kStyLast, POS, WANT_START>(d, b, startByte, startState, cfg, stream);
  default:
    return launchFixedC<kStyFull, POS, WANT_START>(d, b, startByte, startState, cfg, stream);
  }
}

} // namespace

// REDGPU_TAB_HOT_ROWS DFAs whose hot set the streaming kernel can index with one byte
[[maybe_unused]] static bool hotStreamEligible(const DevDfa &d) {
  return d.tableKind == REDGPU_TAB_HOT_ROWS && d.nHot > 0 && d.hot8Off != 0 &&
         d.deadAbsorbing;
}

// DFAs of more than 256 states with a class table of at most 64 KB: k_stream's class-table form
[[maybe_unused]] static bool clsStreamEligible(const DevDfa &d) {
  return d.clsOff != 0 && d.deadAbsorbing &&
         d.clsBytes <= (d.clsIndexForm ? kStreamBigLds : kStreamTabBytes + 1024);
}

#if REDGPU_TU_GENERIC
bool fastPathEligible(const DevDfa &d) {
  return d.tableKind == REDGPU_TAB_LDS_FUSED_U8 && d.deadAbsorbing &&
         size_t(d.tableBytes) + size_t(d.nStates) * 4 <= 150 * 1024;
}
#endif

#if REDGPU_TU_STREAM
// StatefulMatcher chunks the streaming kernels can take (k_stream<advance...>); *handled says so
hipError_t launchAdvanceStream(const DevDfa &d, const Batch &b, uint32_t *state,
                               const LaunchCfg &cfg, hipStream_t stream, const char **kernelName,
                               bool *handled) {
  *handled = false;
  // chunks that are whole 64-byte blocks at a fixed stride: the streaming kernel
  if (!cfg.forceGeneric && fastPathEligible(d) && !b.offsets && b.stride >= 64 &&
      b.stride % 64 == 0 && b.stride < (1ull << 31) &&
      (reinterpret_cast<uintptr_t>(b.data) % 16) == 0 && d.tableBytes <= kStreamTabBytes &&
      d.nStates <= 256) {
    *kernelName = "k_stream<advance>";
    Batch sb = b;
    sb.state = state;
    sb.start = nullptr;
    sb.end = nullptr;
    *handled = true;
    return launchStreamT<kSmAdvance>(d, sb, cfg, stream);
  }
  if (!cfg.forceGeneric && clsStreamEligible(d) && !b.offsets && b.stride >= 64 &&
      b.stride % 64 == 0 && b.stride < (1ull << 31) &&
      (reinterpret_cast<uintptr_t>(b.data) % 16) == 0) {
    *kernelName = "k_stream<advance,cls>";
    Batch sb = b;
    sb.state = state;
    sb.start = nullptr;
    sb.end = nullptr;
    *handled = true;
    return d.clsIndexForm ? launchStreamHot<kSmAdvance, kTabClsBig>(d, sb, cfg, stream)
                          : launchStreamHot<kSmAdvance, kTabCls>(d, sb, cfg, stream);
  }
  if (!cfg.forceGeneric && hotStreamEligible(d) && !b.offsets && b.stride >= 64 &&
      b.stride % 64 == 0 && b.stride < (1ull << 31) &&
      (reinterpret_cast<uintptr_t>(b.data) % 16) == 0) {
    *kernelName = "k_stream<advance,hot>";
    Batch sb = b;
    sb.state = state;
    sb.start = nullptr;
    sb.end = nullptr;
    *handled = true;
    return launchStreamHot<kSmAdvance>(d, sb, cfg, stream);
  }
  return hipSuccess;
}
#endif  // REDGPU_TU_STREAM

#if REDGPU_TU_STREAM
// The fixed-stride family of launchBatch: speculative chunks, k_stream (fused / hot / class
// table), k_fixed.  *handled = false: not this family's batch.
hipError_t launchStreamBatches(const DevDfa &d, const Batch *bs, uint32_t nb, int verb, int style,
                               int doLeader, const LaunchCfg &cfg, hipStream_t stream,
                               const char **kernelName, uint32_t *taken);
hipError_t launchFixedFamily(const DevDfa &d, const Batch &b, int verb, int style, int doLeader,
                             const LaunchCfg &cfg, hipStream_t stream, const char **kernelName,
                             bool *handled) {
  *handled = true;
  const bool lead = doLeader && d.leaderLen > 0;
  const bool dying = d.earlyDeath && !cfg.forceStream;
  // Fixed-stride hot path: check / match, whole 16-byte multiples, 16-byte aligned base.
  // check<.., true> consumes the leader and starts in the post-leader state at byte
  // leaderLen (Matcher.h:370-375); match only peeks it (Matcher.h:424-435).
  const bool fixedOk = !cfg.forceGeneric && !dying && fastPathEligible(d) && !b.offsets &&
                       (verb == kCheck || verb == kMatch) && b.stride >= 16 &&
                       b.stride % 16 == 0 && b.stride < (1ull << 31) &&
                       (reinterpret_cast<uintptr_t>(b.data) % 16) == 0 &&
                       !(lead && verb == kCheck);
  // Streaming kernel: styles Last / Full on lines that are whole 64-byte blocks.
  const bool streamOk = fixedOk && (style == kStyLast || style == kStyFull) &&
                        b.stride % 64 == 0 && d.tableBytes <= kStreamTabBytes && d.nStates <= 256;
  // The same streaming walk for DFAs too big for LDS: hot rows as a one-byte-indexed table,
  // cold excursions re-walked per 64-byte half-block (k_stream.h, HOT).
  const bool hotStreamOk = !cfg.forceGeneric && !dying && hotStreamEligible(d) && !b.offsets &&
                           (verb == kCheck || verb == kMatch) &&
                           (style == kStyLast || style == kStyFull) && b.stride >= 64 &&
                           b.stride % 64 == 0 && b.stride < (1ull << 31) &&
                           (reinterpret_cast<uintptr_t>(b.data) % 16) == 0 &&
                           !(lead && verb == kCheck);
  // ... and for mid-size DFAs whose class table fits 64 KB of LDS (two lookups per byte)
  const bool clsStreamOk = !cfg.forceGeneric && !dying && clsStreamEligible(d) && !b.offsets &&
                           (verb == kCheck || verb == kMatch) &&
                           (style == kStyLast || style == kStyFull) && b.stride >= 64 &&
                           b.stride % 64 == 0 && b.stride < (1ull << 31) &&
                           (reinterpret_cast<uintptr_t>(b.data) % 16) == 0 &&
                           !(lead && verb == kCheck);
  // Few long lines over a DFA that forgets its past: chunks of every line walked at once from
  // the initial state as a guess, wrong guesses re-walked (k_chunk.h)
  if ((streamOk || hotStreamOk || clsStreamOk) && !cfg.noChunking &&
      (d.forgetful || cfg.forceChunking) &&
      (fewLines(b, cfg) || cfg.forceChunking)) {
    const uint32_t m = chunksPerLine(b, cfg);
    if (m) {
      Batch sb = b;
      if (verb == kCheck) { sb.start = nullptr; sb.end = nullptr; }
      *kernelName = d.tableKind == REDGPU_TAB_HOT_ROWS ? "k_stream<chunk,hot>+k_chunk"
                    : d.tableKind == REDGPU_TAB_LDS_FUSED_U8 ? "k_stream<chunk>+k_chunk"
                                                             : "k_stream<chunk,cls>+k_chunk";
      hipError_t e = launchChunked(d, sb, m, style, cfg, stream);
      if (e != hipSuccess) return e;
      if (lead) {
        hipLaunchKernelGGL(k_leader_filter, dim3(uint32_t(cfg.numCUs) * 8), dim3(256), 0, stream,
                           d, b);
        return hipGetLastError();
      }
      return hipSuccess;
    }
  }
  if (streamOk) {
    hipError_t e;
    Batch sb = b;
    if (verb == kCheck) { sb.start = nullptr; sb.end = nullptr; }
    static const int labLean = multiLabInt("REDGPU_LEAN", 0, 1, 0);
    static const int labSingle = multiLabInt("REDGPU_MULTI_SINGLE", 0, 1, 0);
    if (cfg.forceLean || labLean || labSingle) {
      // opt-in: k_stream_multi (its deferred-bookkeeping form for long lines) for a batch on its own
      uint32_t taken = 0;
      e = launchStreamBatches(d, &b, 1, verb, style, doLeader, cfg, stream, kernelName, &taken);
      if (e != hipSuccess || taken) return e;
    }
    if (stream4Eligible(d, sb, cfg)) {
      const int mode = style == kStyLast ? (sb.start ? kSmLastStartEnd : kSmLastEnd)
                                         : (sb.start ? kSmFullStart : kSmFull);
      *kernelName = style == kStyLast ? (sb.start ? "k_stream4<last,start,end>" : "k_stream4<last,end>")
                                      : (sb.start ? "k_stream4<full,start>" : "k_stream4<full>");
      e = launchStream4(mode, d, sb, cfg, stream);
    } else
    if (style == kStyLast) {
      if (sb.start) { *kernelName = "k_stream<last,start,end>"; e = launchStreamT<kSmLastStartEnd>(d, sb, cfg, stream); }
      else { *kernelName = "k_stream<last,end>"; e = launchStreamT<kSmLastEnd>(d, sb, cfg, stream); }
    } else {
      if (sb.start) { *kernelName = "k_stream<full,start>"; e = launchStreamT<kSmFullStart>(d, sb, cfg, stream); }
      else { *kernelName = "k_stream<full>"; e = launchStreamT<kSmFull>(d, sb, cfg, stream); }
    }
    if (e != hipSuccess) return e;
    if (lead) {
      hipLaunchKernelGGL(k_leader_filter, dim3(uint32_t(cfg.numCUs) * 8), dim3(256), 0, stream,
                         d, b);
      return hipGetLastError();
    }
    return hipSuccess;
  }
  if (hotStreamOk) {
    hipError_t e;
    Batch sb = b;
    if (verb == kCheck) { sb.start = nullptr; sb.end = nullptr; }
    if (style == kStyLast) {
      if (sb.start) { *kernelName = "k_stream<last,start,end,hot>"; e = launchStreamHot<kSmLastStartEnd>(d, sb, cfg, stream); }
      else { *kernelName = "k_stream<last,end,hot>"; e = launchStreamHot<kSmLastEnd>(d, sb, cfg, stream); }
    } else {
      if (sb.start) { *kernelName = "k_stream<full,start,hot>"; e = launchStreamHot<kSmFullStart>(d, sb, cfg, stream); }
      else { *kernelName = "k_stream<full,hot>"; e = launchStreamHot<kSmFull>(d, sb, cfg, stream); }
    }
    if (e != hipSuccess) return e;
    if (lead) {
      hipLaunchKernelGGL(k_leader_filter, dim3(uint32_t(cfg.numCUs) * 8), dim3(256), 0, stream,
                         d, b);
      return hipGetLastError();
    }
    return hipSuccess;
  }
  if (clsStreamOk) {
    hipError_t e;
    Batch sb = b;
    if (verb == kCheck) { sb.start = nullptr; sb.end = nullptr; }
    if (style == kStyLast) {
      if (sb.start) { *kernelName = "k_stream<last,start,end,cls>"; e = d.clsIndexForm ? launchStreamHot<kSmLastStartEnd, kTabClsBig>(d, sb, cfg, stream) : launchStreamHot<kSmLastStartEnd, kTabCls>(d, sb, cfg, stream); }
      else { *kernelName = "k_stream<last,end,cls>"; e = d.clsIndexForm ? launchStreamHot<kSmLastEnd, kTabClsBig>(d, sb, cfg, stream) : launchStreamHot<kSmLastEnd, kTabCls>(d, sb, cfg, stream); }
    } else {
      if (sb.start) { *kernelName = "k_stream<full,start,cls>"; e = d.clsIndexForm ? launchStreamHot<kSmFullStart, kTabClsBig>(d, sb, cfg, stream) : launchStreamHot<kSmFullStart, kTabCls>(d, sb, cfg, stream); }
      else { *kernelName = "k_stream<full,cls>"; e = d.clsIndexForm ? launchStreamHot<kSmFull, kTabClsBig>(d, sb, cfg, stream) : launchStreamHot<kSmFull, kTabCls>(d, sb, cfg, stream); }
    }
    if (e != hipSuccess) return e;
    if (lead) {
      hipLaunchKernelGGL(k_leader_filter, dim3(uint32_t(cfg.numCUs) * 8), dim3(256), 0, stream,
                         d, b);
      return hipGetLastError();
    }
    return hipSuccess;
  }
  if (fixedOk) {
    hipError_t e;
    if (verb == kCheck) {
      *kernelName = "k_fixed<check>";
      e = launchFixedS<false, false>(style, d, b, 0, d.init, cfg, stream);
    } else if (b.start) {
      *kernelName = "k_fixed<match,start>";
      e = launchFixedS<true, true>(style, d, b, 0, d.init, cfg, stream);
    } else {
      *kernelName = "k_fixed<match>";
      e = launchFixedS<true, false>(style, d, b, 0, d.init, cfg, stream);
    }
    if (e != hipSuccess) return e;
    if (lead) {
      hipLaunchKernelGGL(k_leader_filter, dim3(uint32_t(cfg.numCUs) * 8), dim3(256), 0, stream,
                         d, b);
      return hipGetLastError();
    }
    return hipSuccess;
  }

  *handled = false;
  return hipSuccess;
}

// Several batches, one launch (k_stream_multi.h).  *taken = how many of bs[0..nb) the launch
// covers: the longest run of leading batches that launchFixedFamily would each give to
// k_stream's plain form, with one stride and the same outputs asked for; 0 = bs[0] is not
// such a batch, or the run is a single batch (the caller then runs it on its own).
hipError_t launchStreamBatches(const DevDfa &d, const Batch *bs, uint32_t nb, int verb, int style,
                               int doLeader, const LaunchCfg &cfg, hipStream_t stream,
                               const char **kernelName, uint32_t *taken) {
  *taken = 0;
  const bool lead = doLeader && d.leaderLen > 0;
  const bool dying = d.earlyDeath && !cfg.forceStream;
  if (cfg.forceGeneric || cfg.forceEarly || cfg.forceChunking || dying || !fastPathEligible(d) ||
      !(verb == kCheck || verb == kMatch) || !(style == kStyLast || style == kStyFull) ||
      (lead && verb == kCheck) || d.tableBytes > kStreamTabBytes || d.nStates > 256)
    return hipSuccess;
  auto plain = [&](const Batch &b) {
    // (few long lines over a forgetful DFA are launchFixedFamily's speculative chunks)
    const bool chunky = fewLines(b, cfg) && d.forgetful && !cfg.noChunking && b.stride >= 4096 &&
                        !cfg.forceLean;
    return !b.offsets && b.stride >= 64 && b.stride % 64 == 0 && b.stride < (1ull << 31) &&
           (reinterpret_cast<uintptr_t>(b.data) % 16) == 0 && b.n > 0 && !chunky &&
           b.n < (1ull << 40) && !stream4Eligible(d, b, cfg);
  };
  const bool wantStart = verb == kMatch && bs[0].start;
  MultiIo m;
  m.nb = 0;
  m.lineLen = uint32_t(bs[0].stride);
  m.tileStart[0] = 0;
  const uint64_t lpt = uint64_t(kStreamThreads) * kStreamChains;
  for (uint32_t k = 0; k < nb && m.nb < uint32_t(kMultiMax); ++k) {
    const Batch &b = bs[k];
    if (!plain(b) || b.stride != bs[0].stride || (verb == kMatch && bool(b.start) != wantStart))
      break;
    const uint64_t tiles = (b.n + lpt - 1) / lpt;
    if (uint64_t(m.tileStart[m.nb]) + tiles >= (1ull << 31)) break;
    m.b[m.nb] = MultiPtrs{b.data, b.result, verb == kMatch ? b.start : nullptr,
                          verb == kMatch ? b.end : nullptr, b.n};
    m.tileStart[m.nb + 1] = m.tileStart[m.nb] + uint32_t(tiles);
    ++m.nb;
  }
  // (fewer tiles than CUs in all: the single-batch launches spread such lines better)
  // (one batch on its own keeps k_stream - unless its lines are long enough for the lean step,
  // or the lab says otherwise: REDGPU_MULTI_SINGLE=1)
  static const int labSingle = multiLabInt("REDGPU_MULTI_SINGLE", 0, 1, 0);
  const bool lean = leanWanted(m, cfg) && !(style == kStyFull && !wantStart);
  if (m.nb < ((labSingle || lean) ? 1u : 2u) ||
      (m.tileStart[m.nb] < uint32_t(cfg.numCUs) && !(lean && cfg.forceLean)))
    return hipSuccess;
  for (uint32_t k = m.nb; k < uint32_t(kMultiMax) + 1; ++k) m.tileStart[k + 1] = m.tileStart[m.nb];
  hipError_t e;
  if (style == kStyLast) {
    if (wantStart) { *kernelName = lean ? "k_stream_multi<last,start,end,lean>" : "k_stream_multi<last,start,end>"; e = launchStreamMultiT<kSmLastStartEnd>(d, m, cfg, stream); }
    else { *kernelName = lean ? "k_stream_multi<last,end,lean>" : "k_stream_multi<last,end>"; e = launchStreamMultiT<kSmLastEnd>(d, m, cfg, stream); }
  } else {
    if (wantStart) { *kernelName = lean ? "k_stream_multi<full,start,lean>" : "k_stream_multi<full,start>"; e = launchStreamMultiT<kSmFullStart>(d, m, cfg, stream); }
    else { *kernelName = "k_stream_multi<full>"; e = launchStreamMultiT<kSmFull>(d, m, cfg, stream); }
  }
  if (e != hipSuccess) return e;
  if (lead) {
    // match<..., true> only peeks the leader (Matcher.h:424-435): lines that fail it report {0,0,0}
    for (uint32_t k = 0; k < m.nb; ++k) {
      hipLaunchKernelGGL(k_leader_filter, dim3(uint32_t(cfg.numCUs) * 8), dim3(256), 0, stream, d,
                         bs[k]);
      if ((e = hipGetLastError()) != hipSuccess) return e;
    }
  }
  *taken = m.nb;
  return hipSuccess;
}
#endif  // REDGPU_TU_STREAM

#if REDGPU_TU_RAGGED
// The ragged family of launchBatch: k_ragged over the fused, hot-row and class tables.
hipError_t launchRaggedFamily(const DevDfa &d, const Batch &b, int verb, int style, int doLeader,
                              const LaunchCfg &cfg, hipStream_t stream, const char **kernelName,
                              bool *handled) {
  *handled = true;
  const bool lead = doLeader && d.leaderLen > 0;
  const bool dying = d.earlyDeath && !cfg.forceStream;
  // Ragged lines, fused-u8 table, styles Last / Full of check / match, no leader to honour:
  // k_ragged walks every line; blocks that would reach past the end of the buffer come from a
  // padded copy of its last bytes.
  const bool raggedOk = !cfg.forceGeneric && !dying && fastPathEligible(d) && b.offsets &&
                        b.n < (1ull << 32) &&
                        (verb == kCheck || verb == kMatch) &&
                        (style == kStyLast || style == kStyFull) && !lead &&
                        d.tableBytes <= kStreamTabBytes && d.nStates <= 256;
  if (raggedOk) {
    hipError_t e;
    Batch sb = b;
    if (verb == kCheck) { sb.start = nullptr; sb.end = nullptr; }
    if (style == kStyLast) {
      if (sb.start) { *kernelName = "k_ragged<last,start,end>"; e = launchRaggedT<kSmLastStartEnd>(d, sb, cfg, stream); }
      else { *kernelName = "k_ragged<last,end>"; e = launchRaggedT<kSmLastEnd>(d, sb, cfg, stream); }
    } else {
      if (sb.start) { *kernelName = "k_ragged<full,start>"; e = launchRaggedT<kSmFullStart>(d, sb, cfg, stream); }
      else { *kernelName = "k_ragged<full>"; e = launchRaggedT<kSmFull>(d, sb, cfg, stream); }
    }
    return e;
  }

  // ... and the same for DFAs too big for LDS (hot rows, cold excursions re-walked per block).
  // On ragged text matches sit at any offset, so with the create-time ranking some lane of a
  // wave is in a cold excursion in nearly every block and the re-walks dominate (URI-V6 on
  // geometric-length text with a URL every ~8 lines: 128 GB/s untuned, 556 GB/s after
  // redgpu_dfa_tune) - still ahead of k_generic on the same lines (98 GB/s: its one-line-per-
  // lane walk also pays the wave-max of the line lengths); without URLs 629 vs 107 GB/s.
  const bool hotRaggedOk = !cfg.forceGeneric && !dying && hotStreamEligible(d) && b.offsets &&
                           b.n < (1ull << 32) &&
                           (verb == kCheck || verb == kMatch) &&
                           (style == kStyLast || style == kStyFull) && !lead;
  if (hotRaggedOk) {
    Batch sb = b;
    if (verb == kCheck) { sb.start = nullptr; sb.end = nullptr; }
    if (style == kStyLast) {
      if (sb.start) { *kernelName = "k_ragged<last,start,end,hot>"; return launchRaggedT<kSmLastStartEnd, kTabHot>(d, sb, cfg, stream); }
      *kernelName = "k_ragged<last,end,hot>";
      return launchRaggedT<kSmLastEnd, kTabHot>(d, sb, cfg, stream);
    }
    if (sb.start) { *kernelName = "k_ragged<full,start,hot>"; return launchRaggedT<kSmFullStart, kTabHot>(d, sb, cfg, stream); }
    *kernelName = "k_ragged<full,hot>";
    return launchRaggedT<kSmFull, kTabHot>(d, sb, cfg, stream);
  }

  // ... and for mid-size DFAs with a class table of at most 64 KB
  const bool clsRaggedOk = !cfg.forceGeneric && !dying && clsStreamEligible(d) && b.offsets &&
                           b.n < (1ull << 32) &&
                           (verb == kCheck || verb == kMatch) &&
                           (style == kStyLast || style == kStyFull) && !lead;
  if (clsRaggedOk) {
    Batch sb = b;
    if (verb == kCheck) { sb.start = nullptr; sb.end = nullptr; }
    if (style == kStyLast) {
      if (sb.start) { *kernelName = "k_ragged<last,start,end,cls>"; return d.clsIndexForm ? launchRaggedT<kSmLastStartEnd, kTabClsBig>(d, sb, cfg, stream) : launchRaggedT<kSmLastStartEnd, kTabCls>(d, sb, cfg, stream); }
      *kernelName = "k_ragged<last,end,cls>";
      return d.clsIndexForm ? launchRaggedT<kSmLastEnd, kTabClsBig>(d, sb, cfg, stream) : launchRaggedT<kSmLastEnd, kTabCls>(d, sb, cfg, stream);
    }
    if (sb.start) { *kernelName = "k_ragged<full,start,cls>"; return d.clsIndexForm ? launchRaggedT<kSmFullStart, kTabClsBig>(d, sb, cfg, stream) : launchRaggedT<kSmFullStart, kTabCls>(d, sb, cfg, stream); }
    *kernelName = "k_ragged<full,cls>";
    return d.clsIndexForm ? launchRaggedT<kSmFull, kTabClsBig>(d, sb, cfg, stream) : launchRaggedT<kSmFull, kTabCls>(d, sb, cfg, stream);
  }

  *handled = false;
  return hipSuccess;
}
#endif  // REDGPU_TU_RAGGED

#if REDGPU_TU_GENERIC
hipError_t launchCollect(const DevDfa &d, const Batch &b, uint64_t cap, uint64_t *counts,
                         const LaunchCfg &cfg, hipStream_t stream) {
  if (b.n == 0) return hipSuccess;
  switch (d.tableKind) {
  case REDGPU_TAB_LDS_FUSED_U8:
    return launchCollectK<REDGPU_TAB_LDS_FUSED_U8>(d, b, cap, counts, cfg, stream);
  case REDGPU_TAB_LDS_FUSED_U16:
    return launchCollectK<REDGPU_TAB_LDS_FUSED_U16>(d, b, cap, counts, cfg, stream);
  case REDGPU_TAB_LDS_CLASS_U16:
    return launchCollectK<REDGPU_TAB_LDS_CLASS_U16>(d, b, cap, counts, cfg, stream);
  case REDGPU_TAB_GLOBAL_U16:
    return launchCollectK<REDGPU_TAB_GLOBAL_U16>(d, b, cap, counts, cfg, stream);
  case REDGPU_TAB_HOT_ROWS:
    return launchCollectK<REDGPU_TAB_HOT_ROWS>(d, b, cap, counts, cfg, stream);
  case REDGPU_TAB_LDS_SPARSE:
    return launchCollectK<REDGPU_TAB_LDS_SPARSE>(d, b, cap, counts, cfg, stream);
  default:
    return launchCollectK<REDGPU_TAB_GLOBAL_U32>(d, b, cap, counts, cfg, stream);
  }
}

#define REDGPU_KIND_SWITCH(CALL)                                                          \
  switch (d.tableKind) {                                                                  \
  case REDGPU_TAB_LDS_FUSED_U8: return CALL(REDGPU_TAB_LDS_FUSED_U8);                     \
  case REDGPU_TAB_LDS_FUSED_U16: return CALL(REDGPU_TAB_LDS_FUSED_U16);                   \
  case REDGPU_TAB_LDS_CLASS_U16: return CALL(REDGPU_TAB_LDS_CLASS_U16);                   \
  case REDGPU_TAB_GLOBAL_U16: return CALL(REDGPU_TAB_GLOBAL_U16);                         \
  case REDGPU_TAB_HOT_ROWS: return CALL(REDGPU_TAB_HOT_ROWS);                               \
  case REDGPU_TAB_LDS_SPARSE: return CALL(REDGPU_TAB_LDS_SPARSE);                         \
  default: return CALL(REDGPU_TAB_GLOBAL_U32);                                            \
  }

hipError_t launchMatchAll(const DevDfa &d, const Batch &b, uint64_t cap, uint64_t *counts,
                          int doLeader, const LaunchCfg &cfg, hipStream_t stream) {
  if (b.n == 0) return hipSuccess;
  const int lead = doLeader && d.leaderLen > 0;
#define MA_CALL(K) launchMatchAllK<K>(d, b, cap, counts, lead, cfg, stream)
  REDGPU_KIND_SWITCH(MA_CALL)
#undef MA_CALL
}

template <int KIND>
hipError_t launchReplaceK(const DevDfa &d, const Batch &b, int style, int lead, const uint8_t *repl,
                          uint64_t replLen, uint64_t max, uint64_t *counts, uint64_t *outLens,
                          const uint64_t *outOffsets, uint8_t *out, uint64_t outCap,
                          const LaunchCfg &cfg, hipStream_t stream) {
  constexpr bool kLds = Tab<KIND>::kInLds || KIND == REDGPU_TAB_HOT_ROWS;
  constexpr int kThreads = kLds ? 1024 : 256;
  const size_t ldsBytes = 512 + ldsTableBytes<KIND>(d);
  hipError_t e = setLds(k_replace<KIND, kThreads>, ldsBytes);
  if (e != hipSuccess) return e;
  uint64_t blocks = (b.n + kThreads - 1) / kThreads;
  const uint64_t perCu = kLds ? (ldsBytes <= 80 * 1024 ? 2 : 1) : 8;
  const uint64_t capBlocks = uint64_t(cfg.numCUs) * perCu;
  if (blocks > capBlocks) blocks = capBlocks;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL((k_replace<KIND, kThreads>), dim3(uint32_t(blocks)), dim3(kThreads), ldsBytes,
                     stream, d, b, style, lead, repl, replLen, max, counts, outLens, outOffsets,
                     out, outCap);
  return hipGetLastError();
}

static hipError_t launchReplacePass(const DevDfa &d, const Batch &b, int style, int lead,
                                    const uint8_t *repl, uint64_t replLen, uint64_t max,
                                    uint64_t *counts, uint64_t *outLens,
                                    const uint64_t *outOffsets, uint8_t *out, uint64_t outCap,
                                    const LaunchCfg &cfg, hipStream_t stream) {
#define RP_CALL(K) launchReplaceK<K>(d, b, style, lead, repl, replLen, max, counts, outLens, \
                                     outOffsets, out, outCap, cfg, stream)
  REDGPU_KIND_SWITCH(RP_CALL)
#undef RP_CALL
}

hipError_t launchReplace(const DevDfa &d, const Batch &b, int style, int doLeader,
                         const uint8_t *repl, uint64_t replLen, uint64_t max, uint64_t *counts,
                         uint64_t *outOffsets, uint8_t *out, uint64_t outCap,
                         const LaunchCfg &cfg, hipStream_t stream) {
  if (b.n == 0) return hipSuccess;
  const int lead = doLeader && d.leaderLen > 0;
  // scratch: lens u64[n] + partials u64[ceil(n / 1024)]
  const uint64_t nPart = (b.n + 1023) / 1024;
  void *scratch = nullptr;
  hipError_t e = raggedScratch(stream, size_t(b.n + nPart) * 8 + 64, &scratch);
  if (e != hipSuccess) return e;
  uint64_t *lens = static_cast<uint64_t *>(scratch);
  uint64_t *partials = lens + b.n;
  e = launchReplacePass(d, b, style, lead, repl, replLen, max, counts, lens, nullptr, nullptr, 0,
                        cfg, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_scan_partials, dim3(uint32_t(nPart)), dim3(256), 0, stream, lens, b.n,
                     partials);
  hipLaunchKernelGGL(k_scan_tops, dim3(1), dim3(1024), 0, stream, partials, nPart);
  hipLaunchKernelGGL(k_scan_fill, dim3(uint32_t(nPart)), dim3(256), 0, stream, lens, b.n, partials,
                     outOffsets);
  e = hipGetLastError();
  if (e != hipSuccess || !out) return e;
  return launchReplacePass(d, b, style, lead, repl, replLen, max, counts, lens, outOffsets, out,
                           outCap, cfg, stream);
}

uint64_t splitChunks(uint64_t len) { return (len + kSplitChunk - 1) / kSplitChunk; }

hipError_t launchSplitLines(const uint8_t *data, uint64_t len, uint8_t delim, uint64_t *offsets,
                            uint64_t cap, uint64_t *nLines, uint32_t *counts, uint64_t *bases,
                            hipStream_t stream) {
  const uint64_t nChunks = splitChunks(len);
  if (nChunks)
    hipLaunchKernelGGL(k_split_count, dim3(uint32_t(nChunks)), dim3(kSplitThreads), 0, stream, data,
                       len, uint32_t(delim), counts);
  hipLaunchKernelGGL(k_split_scan, dim3(1), dim3(1024), 0, stream, counts, nChunks, bases, nLines,
                     offsets, cap);
  if (nChunks)
    hipLaunchKernelGGL(k_split_scatter, dim3(uint32_t(nChunks)), dim3(kSplitThreads), 0, stream,
                       data, len, uint32_t(delim), bases, offsets, cap);
  return hipGetLastError();
}

hipError_t launchDiagRead(const void *data, uint64_t bytes, uint32_t *sink, int numCUs,
                          hipStream_t stream) {
  if (bytes < 16) return hipSuccess;
  hipLaunchKernelGGL(k_diag_read, dim3(uint32_t(numCUs)), dim3(512), 0, stream,
                     static_cast<const uint4 *>(data), bytes / 16, sink);
  return hipGetLastError();
}

hipError_t launchDiagLines(const uint8_t *data, uint64_t nLines, uint32_t lineBytes, int32_t *res,
                           uint64_t *st, uint64_t *en, uint32_t *sink, int numCUs,
                           hipStream_t stream) {
  if (nLines < 1024) return hipSuccess;
  if (lineBytes != 64) {
    hipLaunchKernelGGL(k_diag_long, dim3(uint32_t(numCUs)), dim3(512), 0, stream, data, nLines,
                       lineBytes, sink);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_diag_lines, dim3(uint32_t(numCUs)), dim3(512), 0, stream, data, nLines, res,
                     st, en, sink);
  return hipGetLastError();
}

template <int KIND>
hipError_t launchWalkedK(const DevDfa &d, const Batch &b, int lead, unsigned long long *walked,
                         const LaunchCfg &cfg, hipStream_t stream) {
  constexpr bool kLds = Tab<KIND>::kInLds || KIND == REDGPU_TAB_HOT_ROWS;
  constexpr int kThreads = kLds ? 1024 : 256;
  const size_t ldsBytes = 512 + ldsTableBytes<KIND>(d);
  hipError_t e = setLds(k_walked<KIND, kThreads>, ldsBytes);
  if (e != hipSuccess) return e;
  uint64_t blocks = (b.n + kThreads - 1) / kThreads;
  const uint64_t perCu = kLds ? (ldsBytes <= 80 * 1024 ? 2 : 1) : 8;
  const uint64_t capBlocks = uint64_t(cfg.numCUs) * perCu;
  if (blocks > capBlocks) blocks = capBlocks;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL((k_walked<KIND, kThreads>), dim3(uint32_t(blocks)), dim3(kThreads), ldsBytes,
                     stream, d, b, lead, walked);
  return hipGetLastError();
}

hipError_t launchWalked(const DevDfa &d, const Batch &b, int doLeader, unsigned long long *walked,
                        const LaunchCfg &cfg, hipStream_t stream) {
  if (b.n == 0) return hipSuccess;
  const int lead = doLeader && d.leaderLen > 0;
#define WK_CALL(K) launchWalkedK<K>(d, b, lead, walked, cfg, stream)
  REDGPU_KIND_SWITCH(WK_CALL)
#undef WK_CALL
}

hipError_t launchVisits(const DevDfa &d, const Batch &b, uint32_t *hist, const LaunchCfg &cfg,
                        hipStream_t stream) {
  if (b.n == 0) return hipSuccess;
#define VI_CALL(K) launchVisitsK<K>(d, b, hist, cfg, stream)
  REDGPU_KIND_SWITCH(VI_CALL)
#undef VI_CALL
}

hipError_t launchAdvance(const DevDfa &d, const Batch &b, uint32_t *state, const LaunchCfg &cfg,
                         hipStream_t stream, const char **kernelName) {
  *kernelName = "k_advance";
  if (b.n == 0) return hipSuccess;
  bool handled = false;
  hipError_t se = launchAdvanceStream(d, b, state, cfg, stream, kernelName, &handled);
  if (handled || se != hipSuccess) return se;
#define AD_CALL(K) launchAdvanceK<K>(d, b, state, cfg, stream)
  REDGPU_KIND_SWITCH(AD_CALL)
#undef AD_CALL
}

// The identities launchBatch applies before it picks a kernel.
static void normalizeVerbStyle(const DevDfa &d, const LaunchCfg &cfg, bool lead, int &verb,
                               int &style) {
  // L = SIGMA* L (DfaImage::suffixClosed - patterns added with a loose start) and no leader: the
  // sliding loops of scan and search (Matcher.h:511-553, :575-621) ARE their first attempt.  It
  // cannot meet a pure dead end (from any reachable state something is still accepted), so it
  // either returns per the style's rules - as check / match from position 0 would, same loop
  // body - or reads the line to its end without an accepting state, and then no later start
  // position can accept either: the text it would read is a suffix of what this attempt read.
  // The reference walks every one of them (O(n^2) on a line without a match) to that same 0.
  if (d.suffixClosed && !lead) {
    if (verb == kScan) verb = kCheck;
    else if (verb == kSearch) verb = kMatch;
  }
  // check over a DFA whose accepting states all report ONE result (a single pattern): styInstant,
  // styFirst and styTangent each return the result of SOME accepting state of the walk, styLast
  // that of the last one - the same value, and 0 alike when none accepts (Matcher.h:382-409).
  // styLast is the style the streaming kernels run; an early-death DFA keeps its early exits.
  if (verb == kCheck && !lead && d.uniformResult && !d.earlyDeath && !cfg.forceGeneric &&
      (style == kStyInstant || style == kStyFirst || style == kStyTangent))
    style = kStyLast;
}

hipError_t launchBatches(const DevDfa &d, const Batch *bs, uint32_t nb, int verb, int style,
                         int doLeader, const LaunchCfg &cfg, hipStream_t stream,
                         const char **kernelName) {
  *kernelName = "none";
  int nverb = verb, nstyle = style;
  normalizeVerbStyle(d, cfg, doLeader && d.leaderLen > 0, nverb, nstyle);
  for (uint32_t k = 0; k < nb;) {
    if (bs[k].n == 0) { ++k; continue; }
    uint32_t taken = 0;
    hipError_t e = launchStreamBatches(d, bs + k, nb - k, nverb, nstyle, doLeader, cfg, stream,
                                       kernelName, &taken);
    if (e != hipSuccess) return e;
    if (!taken) {
      e = launchBatch(d, bs[k], verb, style, doLeader, cfg, stream, kernelName);
      if (e != hipSuccess) return e;
      taken = 1;
    }
    k += taken;
  }
  return hipSuccess;
}

hipError_t launchBatch(const DevDfa &d, const Batch &b, int verb, int style, int doLeader,
                       const LaunchCfg &cfg, hipStream_t stream, const char **kernelName) {
  if (b.n == 0) {
    *kernelName = "none";
    return hipSuccess;
  }
  const bool lead = doLeader && d.leaderLen > 0;
  normalizeVerbStyle(d, cfg, lead, verb, style);
  // DFAs the visit model sees dying within 16 bytes (anchored patterns on arbitrary text) stay
  // with k_generic: the whole-line kernels below read and walk every byte, k_generic stops
  // where the reference's loop stops - measured on ERR 1.3x (64-byte lines) to 38x (4 KiB
  // lines) faster (scripts/bench_anchored.py).
  // match over an early-death DFA whose table lives in LDS: probe every line for a few bytes,
  // park the survivors, walk them densely (k_early)
  const bool earlyKind = d.tableKind == REDGPU_TAB_LDS_FUSED_U8 || d.tableKind == REDGPU_TAB_LDS_FUSED_U16 ||
                         d.tableKind == REDGPU_TAB_LDS_CLASS_U16 || d.tableKind == REDGPU_TAB_LDS_SPARSE;
  // (check with a leader consumes it and starts in the post-leader state: k_generic's checkLane)
  if ((verb == kMatch || (verb == kCheck && !lead)) && earlyKind && !cfg.forceGeneric &&
      b.n < (1ull << 32) && d.nStates <= 65535 &&
      size_t(d.tableBytes) + 512 + 16384 + 1024 <= size_t(160) * 1024 &&
      (cfg.forceEarly || (d.earlyDeath && !cfg.forceStream && b.n >= 16384))) {
    *kernelName = verb == kMatch ? "k_early<match>" : "k_early<check>";
    switch (d.tableKind) {
    case REDGPU_TAB_LDS_FUSED_U8: return launchEarlyK<REDGPU_TAB_LDS_FUSED_U8>(d, b, verb, style, lead, cfg, stream);
    case REDGPU_TAB_LDS_FUSED_U16: return launchEarlyK<REDGPU_TAB_LDS_FUSED_U16>(d, b, verb, style, lead, cfg, stream);
    case REDGPU_TAB_LDS_CLASS_U16: return launchEarlyK<REDGPU_TAB_LDS_CLASS_U16>(d, b, verb, style, lead, cfg, stream);
    default: return launchEarlyK<REDGPU_TAB_LDS_SPARSE>(d, b, verb, style, lead, cfg, stream);
    }
  }

  // check / match with an early-exit style, no leader, table in LDS: the block kernel
  // ... and the whole-line styles on a fixed stride that is not a multiple of 64 (100-byte
  // records, 250-byte lines: k_fixed / k_generic gave those a lane each at 1-1.7 TB/s)
  const bool oddStride = !b.offsets && b.stride >= 32 && b.stride % 64 != 0;
  if ((verb == kCheck || verb == kMatch) && !lead && !cfg.forceGeneric && !d.earlyDeath &&
      (style == kStyInstant || style == kStyFirst || style == kStyTangent ||
       ((style == kStyLast || style == kStyFull) && oddStride)) && b.n >= 4096) {
    bool taken = false;
    const bool pos = verb == kMatch && (b.start || b.end);
    hipError_t se = hipSuccess;
    switch (d.tableKind) {
    case REDGPU_TAB_LDS_FUSED_U8: se = launchStyleBlocksK<REDGPU_TAB_LDS_FUSED_U8>(d, b, style, pos, cfg, stream, &taken); break;
    case REDGPU_TAB_LDS_FUSED_U16: se = launchStyleBlocksK<REDGPU_TAB_LDS_FUSED_U16>(d, b, style, pos, cfg, stream, &taken); break;
    case REDGPU_TAB_LDS_CLASS_U16: se = launchStyleBlocksK<REDGPU_TAB_LDS_CLASS_U16>(d, b, style, pos, cfg, stream, &taken); break;
    case REDGPU_TAB_LDS_SPARSE: se = launchStyleBlocksK<REDGPU_TAB_LDS_SPARSE>(d, b, style, pos, cfg, stream, &taken); break;
    default: break;
    }
    if (se != hipSuccess) return se;
    if (taken) {
      *kernelName = verb == kMatch ? "k_style_blocks<match>" : "k_style_blocks<check>";
      return hipSuccess;
    }
  }

  {
    bool handled = false;
    hipError_t fe = launchFixedFamily(d, b, verb, style, doLeader, cfg, stream, kernelName, &handled);
    if (handled || fe != hipSuccess) return fe;
    fe = launchRaggedFamily(d, b, verb, style, doLeader, cfg, stream, kernelName, &handled);
    if (handled || fe != hipSuccess) return fe;
  }

  {
    *kernelName = (verb == kScan || verb == kSearch) && scanMarkable(d, lead) &&
                          !cfg.forceGeneric
                      ? "k_scan_marked" : "k_generic";
  }
  switch (d.tableKind) {
  case REDGPU_TAB_LDS_FUSED_U8:
    return launchGeneric<REDGPU_TAB_LDS_FUSED_U8>(d, b, verb, style, lead, cfg, stream);
  case REDGPU_TAB_LDS_FUSED_U16:
    return launchGeneric<REDGPU_TAB_LDS_FUSED_U16>(d, b, verb, style, lead, cfg, stream);
  case REDGPU_TAB_LDS_CLASS_U16:
    return launchGeneric<REDGPU_TAB_LDS_CLASS_U16>(d, b, verb, style, lead, cfg, stream);
  case REDGPU_TAB_GLOBAL_U16:
    return launchGeneric<REDGPU_TAB_GLOBAL_U16>(d, b, verb, style, lead, cfg, stream);
  case REDGPU_TAB_HOT_ROWS:
    return launchGeneric<REDGPU_TAB_HOT_ROWS>(d, b, verb, style, lead, cfg, stream);
  case REDGPU_TAB_LDS_SPARSE:
    return launchGeneric<REDGPU_TAB_LDS_SPARSE>(d, b, verb, style, lead, cfg, stream);
  default:
    return launchGeneric<REDGPU_TAB_GLOBAL_U32>(d, b, verb, style, lead, cfg, stream);
  }
}

#endif  // REDGPU_TU_GENERIC

} // namespace redgpu
