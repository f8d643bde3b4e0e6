/* red_oracle.c - CPU restatement of RED's DFA match-execution path (plain C11).
 *
 * TEST INFRASTRUCTURE ONLY - see red_oracle.h.  Written from a reading of the reference's
 * algorithm; every function names the reference file:line (relative to
 * /root/reference/quol/red/) it restates.  It works on the serialized "REDA" blob in its
 * native fmtDirect1/2/4 layout, exactly as the reference's matcher does - no repacking -
 * so it is an independent check of the GPU path's repacked tables.
 */
#include "red_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define ORA_INLINE static inline __attribute__((always_inline))
#define ORA_UNLIKELY(x) __builtin_expect(!!(x), 0)

/* ---- include/Serializer.h:42-59 FileHeader field offsets (packed as declared) ---------- */
enum {
  H_MAGIC = 0, H_MAJ = 4, H_MIN = 6, H_CSUM = 8, H_FMT = 12, H_MAXCHAR = 13, H_LEADLEN = 14,
  H_STATECNT = 16, H_INITOFF = 20, H_LEADOFF = 24, H_EQUIV = 32, H_BYTES = 288
};

ORA_INLINE uint16_t ld16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
ORA_INLINE uint32_t ld32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }

/* ---- include/Fnv.h:36-66 -------------------------------------------------------------- */
uint32_t oracle_fnv1a32(const void *p, size_t n) {
  const uint8_t *b = (const uint8_t *)p;
  uint32_t h = 0x811c9dc5u;
  for (size_t i = 0; i < n; ++i) {
    h ^= b[i];
    h *= 0x01000193u;
  }
  return h;
}

uint64_t oracle_fnv1a64(const void *p, size_t n) {
  const uint8_t *b = (const uint8_t *)p;
  uint64_t h = 0xcbf29ce484222325ull;
  for (size_t i = 0; i < n; ++i) {
    h ^= b[i];
    h *= 0x00000100000001B3ull;
  }
  return h;
}

/* ---- lib/Serializer.cpp:301-306: hash of everything from format_ (offset 12) on -------- */
uint32_t oracle_calc_checksum(const void *blob, size_t len) {
  return oracle_fnv1a32((const uint8_t *)blob + H_FMT, len - H_FMT);
}

/* ---- lib/Serializer.cpp:270-298 ------------------------------------------------------- */
const char *oracle_check_header(const void *blob, size_t len) {
  const uint8_t *h = (const uint8_t *)blob;
  if (len < H_BYTES)
    return "Serialized DFA: header too short";
  if (h[0] != 'R' || h[1] != 'E' || h[2] != 'D' || h[3] != 'A')
    return "Serialized DFA: bad magic number";
  if (ld16(h + H_MAJ) != 1 || ld16(h + H_MIN) != 0)
    return "Serialized DFA: unrecognized version";
  uint32_t csum = oracle_calc_checksum(blob, len);
  uint32_t want = ld32(h + H_CSUM);
  if (want != csum) {
    if (want == __builtin_bswap32(csum))
      return "serialized DFA: foreign endian-ness";
    return "serialized DFA: checksum mismatch";
  }
  switch (h[H_FMT]) {
  case 1: case 2: case 4: break;
  default: return "Serialized DFA: unsupported format";
  }
  return NULL;
}

/* ---- lib/Executable.cpp:159-170 ------------------------------------------------------- */
const char *oracle_dfa_init(oracle_dfa *d, const void *blob, size_t len) {
  const char *msg = oracle_check_header(blob, len);
  if (msg)
    return msg;
  const uint8_t *h = (const uint8_t *)blob;
  d->blob = h;
  d->len = len;
  d->equiv = h + H_EQUIV;
  d->leaderLen = h[H_LEADLEN];
  unsigned pad = (d->leaderLen + 7u) & ~7u;
  d->leader = d->leaderLen ? h + H_BYTES : NULL;
  d->base = h + H_BYTES + pad;
  d->fmt = h[H_FMT];
  d->maxChar = h[H_MAXCHAR];
  d->stateCnt = ld32(h + H_STATECNT);
  d->initialOff = ld32(h + H_INITOFF);
  d->leaderOff = ld32(h + H_LEADOFF);
  return NULL;
}

/* ---- include/Proxy.h:116-149 DfaProxy<fmt>, V = sizeof(Value) -------------------------- */
ORA_INLINE uint32_t row_head(const uint8_t *st, const int V) {
  return V == 1 ? st[0] : V == 2 ? ld16(st) : ld32(st);
}
/* result(): Proxy.h:131-133 */
ORA_INLINE int32_t st_result(const uint8_t *st, const int V) {
  uint32_t mask = V == 1 ? 0x7fu : V == 2 ? 0x7fffu : 0x7fffffffu;
  return (int32_t)(row_head(st, V) & mask);
}
/* pureDeadEnd(): Proxy.h:139-141 */
ORA_INLINE int st_pure_dead(const uint8_t *st, const int V) {
  return row_head(st, V) == (1u << (V * 8 - 1));
}
/* next(): Proxy.h:143-147 - offsets_[cls] * sizeof(Value) from base */
ORA_INLINE const uint8_t *st_next(const uint8_t *base, const uint8_t *st, unsigned cls,
                                  const int V) {
  const uint8_t *e = st + (size_t)V * (1u + cls);
  size_t off = V == 1 ? e[0] : V == 2 ? ld16(e) : ld32(e);
  return base + off * (size_t)V;
}

/* ---- include/Matcher.h:333-345 lookingAt (cursor by value) ----------------------------- */
ORA_INLINE int looking_at(const uint8_t *p, const uint8_t *end, const oracle_dfa *d) {
  for (size_t i = 0; i < d->leaderLen; ++i, ++p) {
    if (!(p < end))
      return 0;
    if (d->leader[i] != d->equiv[*p])
      return 0;
  }
  return 1;
}

/* ---- include/Matcher.h:348-360 compareThrough (cursor by reference: on a mismatch the
 * cursor is left AT the mismatching byte, because the return precedes ++inOut) ----------- */
ORA_INLINE int compare_through(const uint8_t **pp, const uint8_t *end, const oracle_dfa *d) {
  const uint8_t *p = *pp;
  for (size_t i = 0; i < d->leaderLen; ++i, ++p) {
    if (!(p < end)) { *pp = p; return 0; }
    if (d->leader[i] != d->equiv[*p]) { *pp = p; return 0; }
  }
  *pp = p;
  return 1;
}

/* ---- include/Matcher.h:363-410 checkCore ----------------------------------------------- */
ORA_INLINE int32_t check_core(const oracle_dfa *d, const uint8_t *p, size_t n, const int style,
                              const int lead, const int V) {
  const uint8_t *end = p + n;
  const uint8_t *base = d->base;
  const uint8_t *equiv = d->equiv;
  const uint8_t *st;

  if (lead) {
    if (!compare_through(&p, end, d))
      return 0;
    st = base + d->leaderOff;
  } else
    st = base + d->initialOff;

  int32_t result = st_result(st, V);
  int32_t prev = 0;

  for (; p < end; ++p) {
    st = st_next(base, st, equiv[*p], V);
    result = st_result(st, V);
    if (ORA_UNLIKELY(result > 0)) {
      if (style == ORA_STY_INSTANT)
        return result;
      if (style == ORA_STY_FIRST) {
        if (prev && result != prev)
          return prev;
        prev = result;
      }
      if (style == ORA_STY_TANGENT || style == ORA_STY_LAST)
        prev = result;
    } else {
      if ((style == ORA_STY_FIRST || style == ORA_STY_TANGENT) && prev > 0)
        return prev;
      if (st_pure_dead(st, V))
        break;
    }
  }

  if (style == ORA_STY_LAST)
    if (result == 0 && prev > 0)
      return prev;
  return result;
}

/* ---- include/Matcher.h:413-495 matchCore ----------------------------------------------- */
ORA_INLINE int32_t match_core(const oracle_dfa *d, const uint8_t *p, size_t n, const int style,
                              const int lead, const int V, uint64_t *startOut,
                              uint64_t *endOut) {
  const uint8_t *end = p + n;
  const uint8_t *base = d->base;
  const uint8_t *equiv = d->equiv;

  if (lead && !looking_at(p, end, d)) {
    *startOut = 0;
    *endOut = 0;
    return 0;
  }

  const uint8_t *init = base + d->initialOff;
  const uint8_t *st = init;
  int32_t result = st_result(st, V);
  int32_t prev = 0;
  size_t idx = 0, matchStart = 0, matchEnd = 0;

  for (; p < end; ++p, ++idx) {
    unsigned cls = equiv[*p];
    if (ORA_UNLIKELY(st == init)) {
      const uint8_t *was = st;
      st = st_next(base, st, cls, V);
      if (st != was)
        matchStart = idx;
    } else
      st = st_next(base, st, cls, V);
    result = st_result(st, V);
    if (ORA_UNLIKELY(result > 0)) {
      if (style == ORA_STY_FIRST) {
        if (prev && result != prev) {
          result = prev;
          break;
        }
        prev = result;
      }
      matchEnd = idx + 1;
      if (style == ORA_STY_INSTANT)
        break;
      if (style == ORA_STY_TANGENT || style == ORA_STY_LAST)
        prev = result;
    } else {
      if (style == ORA_STY_FIRST && prev > 0) {
        result = prev;
        break;
      }
      if (style == ORA_STY_TANGENT && prev > 0)
        break;
      if (st_pure_dead(st, V))
        break;
    }
  }

  if (style == ORA_STY_TANGENT || style == ORA_STY_LAST)
    if (result == 0 && prev > 0)
      result = prev;

  if (result == 0) {
    *startOut = 0;
    *endOut = 0;
  } else {
    *startOut = matchStart;
    *endOut = matchEnd;
  }
  return result;
}

/* ---- include/Matcher.h:498-554 scanCore ------------------------------------------------ */
ORA_INLINE int32_t scan_core(const oracle_dfa *d, const uint8_t *p, size_t n, const int style,
                             const int lead, const int V) {
  const uint8_t *end = p + n;
  const uint8_t *base = d->base;
  const uint8_t *equiv = d->equiv;
  const uint8_t *initSt = base + d->initialOff;
  const uint8_t *leadSt = base + d->leaderOff;

  int32_t result = st_result(initSt, V);

  for (; p < end; ++p) {
    const uint8_t *st;
    if (lead) {
      if (!compare_through(&p, end, d))
        continue; /* p sits at the mismatching byte (or at end); the loop's ++p skips it */
      st = leadSt;
      result = st_result(st, V);
    } else
      st = initSt;

    int32_t prev = 0;
    for (const uint8_t *q = p; q < end; ++q) {
      st = st_next(base, st, equiv[*q], V);
      result = st_result(st, V);
      if (ORA_UNLIKELY(result > 0)) {
        if (style == ORA_STY_INSTANT)
          return result;
        if (style == ORA_STY_FIRST) {
          if (prev && result != prev)
            return prev;
          prev = result;
        }
        if (style == ORA_STY_TANGENT || style == ORA_STY_LAST)
          prev = result;
      } else {
        if ((style == ORA_STY_FIRST || style == ORA_STY_TANGENT) && prev > 0)
          return prev;
        if (st_pure_dead(st, V))
          break;
      }
    }

    if (style == ORA_STY_LAST)
      if (result == 0 && prev > 0)
        return prev;
    if (result > 0)
      return result;
  }
  return result;
}

/* ---- include/Matcher.h:557-640 searchCore ---------------------------------------------- */
ORA_INLINE int32_t search_core(const oracle_dfa *d, const uint8_t *p, size_t n, const int style,
                               const int lead, const int V, uint64_t *startOut,
                               uint64_t *endOut) {
  const uint8_t *end = p + n;
  const uint8_t *base = d->base;
  const uint8_t *equiv = d->equiv;
  const uint8_t *init = base + d->initialOff;

  int32_t result = st_result(init, V);
  size_t idx = 0, matchStart = 0, matchEnd = 0;

  for (; p < end; ++p, ++idx) {
    if (lead && !looking_at(p, end, d))
      continue;

    const uint8_t *st = init;
    int32_t prev = 0;
    size_t innerIdx = idx;
    matchStart = idx;
    matchEnd = idx;
    for (const uint8_t *q = p; q < end; ++q, ++innerIdx) {
      unsigned cls = equiv[*q];
      if (ORA_UNLIKELY(st == init)) {
        const uint8_t *was = st;
        st = st_next(base, st, cls, V);
        if (st != was)
          matchStart = innerIdx;
      } else
        st = st_next(base, st, cls, V);
      result = st_result(st, V);
      if (ORA_UNLIKELY(result > 0)) {
        if (style == ORA_STY_FIRST) {
          if (prev && result != prev) {
            result = prev;
            break;
          }
          prev = result;
        }
        matchEnd = innerIdx + 1;
        if (style == ORA_STY_INSTANT)
          break;
        if (style == ORA_STY_TANGENT || style == ORA_STY_LAST)
          prev = result;
      } else {
        if (style == ORA_STY_FIRST && prev > 0) {
          result = prev;
          break;
        }
        if (style == ORA_STY_TANGENT && prev > 0)
          break;
        if (st_pure_dead(st, V))
          break;
      }
    }

    if (style == ORA_STY_TANGENT || style == ORA_STY_LAST)
      if (result == 0 && prev > 0)
        result = prev;
    if (result > 0)
      break;
  }

  if (result == 0) {
    *startOut = 0;
    *endOut = 0;
  } else {
    *startOut = matchStart;
    *endOut = matchEnd;
  }
  return result;
}

/* ---- dispatch: ZEZAX_RED_FMT_SWITCH (Matcher.h:249-262) x STYLE_SWITCH (Matcher.cpp:37-46).
 * Constant arguments + always_inline give one specialised body per (fmt, style, doLeader),
 * as the reference's templates do. ------------------------------------------------------ */
#define ORA_BAD (-1000)

#define ORA_STYLES(CALL, L, V)                 \
  switch (style) {                             \
  case ORA_STY_INSTANT: CALL(ORA_STY_INSTANT, L, V); \
  case ORA_STY_FIRST:   CALL(ORA_STY_FIRST, L, V);   \
  case ORA_STY_TANGENT: CALL(ORA_STY_TANGENT, L, V); \
  case ORA_STY_LAST:    CALL(ORA_STY_LAST, L, V);    \
  case ORA_STY_FULL:    CALL(ORA_STY_FULL, L, V);    \
  default: return ORA_BAD;                     \
  }

#define ORA_DISPATCH(CALL)                                        \
  switch (d->fmt) {                                               \
  case 1: if (doLeader) { ORA_STYLES(CALL, 1, 1) } else { ORA_STYLES(CALL, 0, 1) } \
  case 2: if (doLeader) { ORA_STYLES(CALL, 1, 2) } else { ORA_STYLES(CALL, 0, 2) } \
  case 4: if (doLeader) { ORA_STYLES(CALL, 1, 4) } else { ORA_STYLES(CALL, 0, 4) } \
  default: return ORA_BAD;                                        \
  }

int32_t oracle_check(const oracle_dfa *d, const uint8_t *p, size_t n, int style, int doLeader) {
#define CALL(S, L, V) return check_core(d, p, n, S, L, V)
  ORA_DISPATCH(CALL)
#undef CALL
}

int32_t oracle_scan(const oracle_dfa *d, const uint8_t *p, size_t n, int style, int doLeader) {
#define CALL(S, L, V) return scan_core(d, p, n, S, L, V)
  ORA_DISPATCH(CALL)
#undef CALL
}

int32_t oracle_match(const oracle_dfa *d, const uint8_t *p, size_t n, int style, int doLeader,
                     uint64_t *start, uint64_t *end) {
#define CALL(S, L, V) return match_core(d, p, n, S, L, V, start, end)
  ORA_DISPATCH(CALL)
#undef CALL
}

int32_t oracle_search(const oracle_dfa *d, const uint8_t *p, size_t n, int style, int doLeader,
                      uint64_t *start, uint64_t *end) {
#define CALL(S, L, V) return search_core(d, p, n, S, L, V, start, end)
  ORA_DISPATCH(CALL)
#undef CALL
}

/* ---- lib/Red.cpp:103-116 Red::collect ------------------------------------------------- */
uint64_t oracle_collect(const oracle_dfa *d, const uint8_t *p, size_t n, uint64_t cap,
                        int32_t *res, uint64_t *start, uint64_t *end) {
  uint64_t found = 0;
  size_t pos = 0;
  while (pos < n) {
    uint64_t s = 0, e = 0;
    int32_t r = oracle_search(d, p + pos, n - pos, ORA_STY_LAST, 0, &s, &e);
    if (!(r > 0))
      break;
    if (found < cap) {
      res[found] = r;
      start[found] = pos + s;
      end[found] = pos + e;
    }
    ++found;
    pos += e;
  }
  return found;
}

void oracle_collect_batch(const oracle_dfa *d, const uint8_t *data, const uint64_t *offsets,
                          uint64_t stride, uint64_t lineLen, uint64_t n, uint64_t cap,
                          uint64_t *counts, int32_t *res, uint64_t *start, uint64_t *end) {
  for (uint64_t i = 0; i < n; ++i) {
    const uint8_t *p = offsets ? data + offsets[i] : data + i * stride;
    size_t len = offsets ? (size_t)(offsets[i + 1] - offsets[i]) : (size_t)lineLen;
    counts[i] = oracle_collect(d, p, len, cap, res + i * cap, start + i * cap, end + i * cap);
  }
}

/* ---- include/Matcher.h:643-706 replaceCore ---------------------------------------------
 * Writes at most outCap bytes to out (may be NULL with outCap 0); *outLen = length of the full
 * result.  Returns the number of replacements. */
ORA_INLINE uint64_t replace_core(const oracle_dfa *d, const uint8_t *p, size_t n, const int style,
                                 const int lead, const int V, const uint8_t *repl, size_t replLen,
                                 uint64_t max, uint8_t *out, uint64_t outCap, uint64_t *outLen) {
  const uint8_t *stop = p + n;
  const uint8_t *base = d->base;
  const uint8_t *equiv = d->equiv;
  const uint8_t *init = base + d->initialOff;
  uint64_t cnt = 0, w = 0;
#define ORA_PUT(ch) do { if (w < outCap) out[w] = (ch); ++w; } while (0)
  const uint8_t *in = p;
  while (in < stop) {
    if (cnt >= max) {
      for (; in < stop; ++in)
        ORA_PUT(*in);
      break;
    }
    const uint8_t *found = NULL;
    if (!lead || looking_at(in, stop, d)) {
      const uint8_t *st = init;
      int32_t prevResult = 0;
      for (const uint8_t *inner = in; inner < stop; ++inner) {
        st = st_next(base, st, equiv[*inner], V);
        int32_t result = st_result(st, V);
        if (ORA_UNLIKELY(result > 0)) {
          if (style == ORA_STY_FIRST) {
            if (prevResult && (result != prevResult))
              break;
            prevResult = result;
          }
          found = inner;
          if (style == ORA_STY_INSTANT)
            break;
        } else {
          if (style == ORA_STY_FULL)
            found = NULL;
          if ((((style == ORA_STY_FIRST) || (style == ORA_STY_TANGENT)) && found) ||
              st_pure_dead(st, V))
            break;
        }
      }
    }
    if (found) {
      for (size_t k = 0; k < replLen; ++k)
        ORA_PUT(repl[k]);
      in = found + 1;
      ++cnt;
    } else {
      ORA_PUT(*in);
      ++in;
    }
  }
#undef ORA_PUT
  *outLen = w;
  return cnt;
}

uint64_t oracle_replace(const oracle_dfa *d, const uint8_t *p, size_t n, int style, int doLeader,
                        const uint8_t *repl, size_t replLen, uint64_t max, uint8_t *out,
                        uint64_t outCap, uint64_t *outLen) {
#define CALL(S, L, V) return replace_core(d, p, n, S, L, V, repl, replLen, max, out, outCap, outLen)
  ORA_DISPATCH(CALL)
#undef CALL
}

/* ---- include/Matcher.h:711-766 matchAllCore (public entry matchAll, lib/Matcher.cpp:97-102,
 * always <styTangent, doLeader=true>; the style parameter is unused by the core) ---------- */
ORA_INLINE uint64_t match_all_core(const oracle_dfa *d, const uint8_t *p, size_t n,
                                   const int lead, const int V, uint64_t cap, int32_t *res,
                                   uint64_t *start, uint64_t *end) {
  const uint8_t *stop = p + n;
  const uint8_t *base = d->base;
  const uint8_t *equiv = d->equiv;
  uint64_t found = 0; /* out.size() */

  if (lead && !looking_at(p, stop, d))
    return 0;

  const uint8_t *init = base + d->initialOff;
  const uint8_t *st = init;
  int32_t prevResult = 0;
  size_t idx = 0;
  size_t matchStart = 0;

  for (; p < stop; ++p, ++idx) {
    unsigned cls = equiv[*p];
    if (ORA_UNLIKELY(st == init)) {
      const uint8_t *prevState = st;
      st = st_next(base, st, cls, V);
      if (st != prevState)
        matchStart = idx;
    } else
      st = st_next(base, st, cls, V);
    int32_t result = st_result(st, V);
    if (ORA_UNLIKELY(result > 0)) {
      if (result == prevResult) {
        if (found - 1 < cap)
          end[found - 1] = idx + 1; /* out.back().end_ */
      } else {
        prevResult = result;
        if (found < cap) {
          res[found] = result;
          start[found] = matchStart;
          end[found] = idx + 1;
        }
        ++found;
      }
    } else {
      if (st_pure_dead(st, V))
        break;
      prevResult = 0;
    }
  }
  return found;
}

uint64_t oracle_match_all(const oracle_dfa *d, const uint8_t *p, size_t n, int doLeader,
                          uint64_t cap, int32_t *res, uint64_t *start, uint64_t *end) {
  const int lead = doLeader ? 1 : 0;
  switch (d->fmt) {
  case 1: return match_all_core(d, p, n, lead, 1, cap, res, start, end);
  case 2: return match_all_core(d, p, n, lead, 2, cap, res, start, end);
  case 4: return match_all_core(d, p, n, lead, 4, cap, res, start, end);
  default: return 0;
  }
}

void oracle_match_all_batch(const oracle_dfa *d, int doLeader, const uint8_t *data,
                            const uint64_t *offsets, uint64_t stride, uint64_t lineLen,
                            uint64_t n, uint64_t cap, uint64_t *counts, int32_t *res,
                            uint64_t *start, uint64_t *end) {
  for (uint64_t i = 0; i < n; ++i) {
    const uint8_t *p = offsets ? data + offsets[i] : data + i * stride;
    size_t len = offsets ? (size_t)(offsets[i + 1] - offsets[i]) : (size_t)lineLen;
    counts[i] = oracle_match_all(d, p, len, doLeader, cap, res + i * cap, start + i * cap,
                                 end + i * cap);
  }
}

/* ---- include/Matcher.h:770-792, lib/Matcher.cpp:106-158 StatefulMatcher ----------------
 * *state is the matcher's state_ as a byte offset from base; ORA_STATE_INITIAL stands for a
 * freshly constructed matcher (state_ = the initial row, lib/Matcher.cpp:113-136).  Advances
 * over p[0..n) one advance() per byte; returns result() after the last one.  perByte (may be
 * NULL) receives the return value of every advance(). */
ORA_INLINE int32_t advance_core(const oracle_dfa *d, uint32_t *state, const uint8_t *p, size_t n,
                                const int V, int32_t *perByte) {
  const uint8_t *base = d->base;
  const uint8_t *st = base + (*state == ORA_STATE_INITIAL ? d->initialOff : *state);
  for (size_t i = 0; i < n; ++i) {
    st = st_next(base, st, d->equiv[p[i]], V);
    if (perByte)
      perByte[i] = st_result(st, V);
  }
  *state = (uint32_t)(st - base);
  return st_result(st, V);
}

int32_t oracle_advance(const oracle_dfa *d, uint32_t *state, const uint8_t *p, size_t n,
                       int32_t *perByte) {
  switch (d->fmt) {
  case 1: return advance_core(d, state, p, n, 1, perByte);
  case 2: return advance_core(d, state, p, n, 2, perByte);
  case 4: return advance_core(d, state, p, n, 4, perByte);
  default: return ORA_BAD;
  }
}

void oracle_advance_batch(const oracle_dfa *d, const uint8_t *data, const uint64_t *offsets,
                          uint64_t stride, uint64_t lineLen, uint64_t n, uint32_t *state,
                          int32_t *res) {
  for (uint64_t i = 0; i < n; ++i) {
    const uint8_t *p = offsets ? data + offsets[i] : data + i * stride;
    size_t len = offsets ? (size_t)(offsets[i + 1] - offsets[i]) : (size_t)lineLen;
    res[i] = oracle_advance(d, &state[i], p, len, NULL);
  }
}

/* ---- batch: the callers' outer loop, N threads over contiguous shards
 * (tools/thr_red.cpp:36-47,86-91; tools/bench.cpp:60-71) -------------------------------- */
typedef struct {
  const oracle_dfa *d;
  int verb, style, lead;
  const uint8_t *data;
  const uint64_t *offsets;
  uint64_t stride, lineLen, lo, hi;
  int32_t *res;
  uint64_t *start, *end;
} batch_job;

static void *batch_run(void *arg) {
  batch_job *j = (batch_job *)arg;
  for (uint64_t i = j->lo; i < j->hi; ++i) {
    const uint8_t *p;
    size_t n;
    if (j->offsets) {
      p = j->data + j->offsets[i];
      n = (size_t)(j->offsets[i + 1] - j->offsets[i]);
    } else {
      p = j->data + i * j->stride;
      n = (size_t)j->lineLen;
    }
    uint64_t s = 0, e = 0;
    int32_t r;
    switch (j->verb) {
    case ORA_CHECK: r = oracle_check(j->d, p, n, j->style, j->lead); break;
    case ORA_SCAN:  r = oracle_scan(j->d, p, n, j->style, j->lead); break;
    case ORA_MATCH: r = oracle_match(j->d, p, n, j->style, j->lead, &s, &e); break;
    default:        r = oracle_search(j->d, p, n, j->style, j->lead, &s, &e); break;
    }
    j->res[i] = r;
    if (j->start) j->start[i] = s;
    if (j->end) j->end[i] = e;
  }
  return NULL;
}

void oracle_batch(const oracle_dfa *d, int verb, int style, int doLeader, const uint8_t *data,
                  const uint64_t *offsets, uint64_t stride, uint64_t lineLen, uint64_t n,
                  int32_t *res, uint64_t *start, uint64_t *end, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if ((uint64_t)nthreads > n) nthreads = n ? (int)n : 1;
  batch_job *jobs = (batch_job *)calloc((size_t)nthreads, sizeof(batch_job));
  pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
  uint64_t per = (n + (uint64_t)nthreads - 1) / (uint64_t)nthreads;
  int started = 0;
  for (int t = 0; t < nthreads; ++t) {
    uint64_t lo = per * (uint64_t)t;
    uint64_t hi = lo + per < n ? lo + per : n;
    if (lo >= hi) break;
    batch_job j = { d, verb, style, doLeader, data, offsets, stride, lineLen, lo, hi,
                    res, start, end };
    jobs[t] = j;
    if (nthreads == 1)
      batch_run(&jobs[t]);
    else {
      pthread_create(&th[t], NULL, batch_run, &jobs[t]);
      ++started;
    }
  }
  for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
  free(jobs);
  free(th);
}
