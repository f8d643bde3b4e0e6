// k_lanes.h - the lane functions - direct restatements of the reference's cores over the renumbered
// device image: MatchWalk / LastWalk / CheckWalk / ScanWalk / SearchWalk and checkLane ... searchLane
// (include/Matcher.h:363-640)
// (included by kernels.hip inside namespace redgpu { namespace { ... } }; see its file map).
#pragma once

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
