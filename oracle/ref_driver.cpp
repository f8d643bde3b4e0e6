/* ref_driver.cpp - thin extern "C" face over the REAL reference library.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is compiled together with the reference's own
 * sources where they lie under /root/reference/quol/red/{include,lib} (see oracle/Makefile,
 * target `ref`); the result goes to oracle/_ref/libredref.so, which is git-ignored.  No
 * reference source is copied here: this file only *calls* the reference's public API
 *   - Parser::add/addAuto/addGlob/addExact      (include/Parser.h:52-91)
 *   - compileToSerialized                        (include/Compile.h:30, lib/Compile.cpp:22-45)
 *   - DfaObj / DfaMinimizer / Serializer         (include/Dfa.h:170-216, Minimizer.h:134-139,
 *                                                 Serializer.h:80-85)
 *   - Executable(gCopyTag, sv)                   (include/Executable.h:37)
 *   - check/match/scan/search<style,doLeader>    (include/Matcher.h:133-182)
 *   - checkHeader                                (include/Serializer.h:109)
 *   - Red(gCopyTag, sv), Red::collect            (include/Red.h:103,115; lib/Red.cpp:103-116)
 * so that (1) the C restatement in oracle/red_oracle.c can be validated against the real
 * thing, (2) golden vectors can be generated (oracle/gen_golden.py) and (3) bench.py's
 * cpu_baseline leg can time the reference itself ("kind": "reference").
 *
 * Nothing in the product path (one_amd/, include/) may link or load this.
 */

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <thread>
#include <vector>

#include "Compile.h"
#include "Executable.h"
#include "Matcher.h"
#include "Minimizer.h"
#include "Parser.h"
#include "Red.h"
#include "Serializer.h"

using namespace zezax::red;

namespace {

void setErr(char *err, size_t errLen, const char *msg) {
  if (err && errLen) {
    std::strncpy(err, msg, errLen - 1);
    err[errLen - 1] = '\0';
  }
}

int exceptCode(const std::exception &e) {
  if (dynamic_cast<const RedExceptParse *>(&e)) return -4;
  if (dynamic_cast<const RedExceptApi *>(&e)) return -1;
  if (dynamic_cast<const RedExceptExec *>(&e)) return -2;
  if (dynamic_cast<const RedExceptLimit *>(&e)) return -3;
  return -9;
}

void *dupBlob(const std::string &s) {
  void *p = std::malloc(s.size());
  std::memcpy(p, s.data(), s.size());
  return p;
}

#define STYLE_DISPATCH(A_call)                            \
  switch (style) {                                        \
  case 1: A_call(styInstant) break;                       \
  case 2: A_call(styFirst) break;                         \
  case 3: A_call(styTangent) break;                       \
  case 4: A_call(styLast) break;                          \
  case 5: A_call(styFull) break;                          \
  default: throw RedExceptExec("unsupported style");      \
  }

Result doCheck(const Executable &ex, const void *p, size_t n, int style, int lead) {
#define C(S) { return lead ? check<S, true>(ex, p, n) : check<S, false>(ex, p, n); }
  STYLE_DISPATCH(C)
#undef C
  return 0;
}

Outcome doMatch(const Executable &ex, const void *p, size_t n, int style, int lead) {
#define C(S) { return lead ? match<S, true>(ex, p, n) : match<S, false>(ex, p, n); }
  STYLE_DISPATCH(C)
#undef C
  return Outcome::fail();
}

Result doScan(const Executable &ex, const void *p, size_t n, int style, int lead) {
#define C(S) { return lead ? scan<S, true>(ex, p, n) : scan<S, false>(ex, p, n); }
  STYLE_DISPATCH(C)
#undef C
  return 0;
}

Outcome doSearch(const Executable &ex, const void *p, size_t n, int style, int lead) {
#define C(S) { return lead ? search<S, true>(ex, p, n) : search<S, false>(ex, p, n); }
  STYLE_DISPATCH(C)
#undef C
  return Outcome::fail();
}

inline void lineOf(const uint8_t *data, const uint64_t *offsets, uint64_t stride,
                   uint64_t lineLen, uint64_t i, const uint8_t *&p, size_t &n) {
  if (offsets) {
    p = data + offsets[i];
    n = offsets[i + 1] - offsets[i];
  } else {
    p = data + i * stride;
    n = lineLen;
  }
}

template <class F>
void parallelFor(uint64_t n, int nthreads, F f) {
  if (nthreads <= 1) {
    f(0, n);
    return;
  }
  std::vector<std::thread> th;
  uint64_t per = (n + nthreads - 1) / nthreads;
  for (int t = 0; t < nthreads; ++t) {
    uint64_t lo = per * t, hi = std::min<uint64_t>(n, lo + per);
    if (lo >= hi) break;
    th.emplace_back([=] { f(lo, hi); });
  }
  for (auto &t : th) t.join();
}

} // anonymous

extern "C" {

/* lang: 1 add (raw), 2 addAuto, 3 addGlob, 4 addExact  (Parser.h:43-49) */
int ref_compile(int npat, const char **pats, const size_t *patLens, const int32_t *results,
                const uint32_t *flags, const int *langs, int fmt, void **blobOut,
                size_t *lenOut, char *err, size_t errLen) {
  try {
    Parser p;
    for (int i = 0; i < npat; ++i)
      p.addAs(static_cast<Language>(langs[i]), std::string_view(pats[i], patLens[i]),
              results[i], flags[i]);
    std::string buf = compileToSerialized(p, static_cast<Format>(fmt));
    *blobOut = dupBlob(buf);
    *lenOut = buf.size();
    return 0;
  } catch (const std::exception &e) {
    setErr(err, errLen, e.what());
    return exceptCode(e);
  }
}

/* Random dense DFA pushed through the reference's DfaMinimizer + Serializer
 * (SURVEY.md section 8a "SYN-256"/"SYN-4K").  States 1..nstates, every transition targets
 * 1..nstates, so error state 0 is unreachable.  acceptEvery: state s accepts with result
 * 1 + (s % maxResult) when (s % acceptEvery) == 0.  Deterministic SplitMix64. */
int ref_syn_dfa(uint32_t nstates, uint64_t seed, uint32_t acceptEvery, uint32_t maxResult,
                int fmt, void **blobOut, size_t *lenOut, char *err, size_t errLen) {
  try {
    DfaObj dfa;
    for (uint32_t i = 0; i <= nstates; ++i) dfa.newState();
    uint64_t x = seed;
    auto next = [&x]() {
      uint64_t z = (x += 0x9E3779B97F4A7C15ULL);
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
      return z ^ (z >> 31);
    };
    for (uint32_t s = 1; s <= nstates; ++s) {
      DfaState &ds = dfa[s];
      ds.deadEnd_ = false;
      ds.result_ = (acceptEvery && (s % acceptEvery) == 0)
                       ? static_cast<Result>(1 + (s % maxResult)) : 0;
      for (CharIdx ch = 0; ch < gAlphabetSize; ++ch)
        ds.transitions_.set(ch, static_cast<DfaId>(1 + next() % nstates));
    }
    {
      DfaMinimizer dm(dfa);
      dm.minimize();
    }
    Serializer ser(dfa);
    std::string buf = ser.serializeToString(static_cast<Format>(fmt));
    *blobOut = dupBlob(buf);
    *lenOut = buf.size();
    return 0;
  } catch (const std::exception &e) {
    setErr(err, errLen, e.what());
    return exceptCode(e);
  }
}

void ref_free(void *p) { std::free(p); }

/* returns NULL if the blob is good, else the reference's message (Serializer.cpp:270-298) */
const char *ref_check_header(const void *blob, size_t len) { return checkHeader(blob, len); }

void *ref_exec_create(const void *blob, size_t len, char *err, size_t errLen) {
  try {
    return new Executable(gCopyTag,
                          std::string_view(static_cast<const char *>(blob), len));
  } catch (const std::exception &e) {
    setErr(err, errLen, e.what());
    return nullptr;
  }
}

void ref_exec_destroy(void *ex) { delete static_cast<Executable *>(ex); }

int32_t ref_check(void *ex, const void *p, size_t n, int style, int lead) {
  return doCheck(*static_cast<Executable *>(ex), p, n, style, lead);
}

int32_t ref_scan(void *ex, const void *p, size_t n, int style, int lead) {
  return doScan(*static_cast<Executable *>(ex), p, n, style, lead);
}

void ref_match(void *ex, const void *p, size_t n, int style, int lead, int32_t *res,
               uint64_t *start, uint64_t *end) {
  Outcome oc = doMatch(*static_cast<Executable *>(ex), p, n, style, lead);
  *res = oc.result_;
  *start = oc.start_;
  *end = oc.end_;
}

void ref_search(void *ex, const void *p, size_t n, int style, int lead, int32_t *res,
                uint64_t *start, uint64_t *end) {
  Outcome oc = doSearch(*static_cast<Executable *>(ex), p, n, style, lead);
  *res = oc.result_;
  *start = oc.start_;
  *end = oc.end_;
}

/* Red::collect over one text; writes at most cap records, returns the number found */
uint64_t ref_collect(const void *blob, size_t len, const void *text, size_t n, uint64_t cap,
                     int32_t *res, uint64_t *start, uint64_t *end) {
  Red re(gCopyTag, std::string_view(static_cast<const char *>(blob), len));
  std::vector<Outcome> out;
  re.collect(std::string_view(static_cast<const char *>(text), n), out);
  for (size_t i = 0; i < out.size() && i < cap; ++i) {
    res[i] = out[i].result_;
    start[i] = out[i].start_;
    end[i] = out[i].end_;
  }
  return out.size();
}

/* replace<style,doLeader>(exec, ptr, len, repl, out, max) (include/Matcher.h:186-191,643-706) */
uint64_t ref_replace(void *exv, const void *text, size_t n, int style, int lead, const void *repl,
                     size_t replLen, uint64_t max, void *out, uint64_t outCap, uint64_t *outLen) {
  const Executable &ex = *static_cast<Executable *>(exv);
  std::string_view rv(static_cast<const char *>(repl), replLen);
  std::string o;
  size_t cnt = 0;
#define C(S) { cnt = lead ? replace<S, true>(ex, text, n, rv, o, size_t(max))               \
                          : replace<S, false>(ex, text, n, rv, o, size_t(max)); }
  STYLE_DISPATCH(C)
#undef C
  *outLen = o.size();
  if (out && outCap) std::memcpy(out, o.data(), o.size() < outCap ? o.size() : outCap);
  return cnt;
}

/* matchAll(exec, string_view, vector<Outcome>&) (include/Matcher.h:127, lib/Matcher.cpp:97-102) */
uint64_t ref_match_all(void *ex, const void *text, size_t n, uint64_t cap, int32_t *res,
                       uint64_t *start, uint64_t *end) {
  std::vector<Outcome> out;
  matchAll(*static_cast<Executable *>(ex),
           std::string_view(static_cast<const char *>(text), n), out);
  for (size_t i = 0; i < out.size() && i < cap; ++i) {
    res[i] = out[i].result_;
    start[i] = out[i].start_;
    end[i] = out[i].end_;
  }
  return out.size();
}

/* StatefulMatcher (include/Matcher.h:770-792): a fresh matcher advanced over text byte by
 * byte; perByte[i] = advance(text[i]); *initial = result() before the first byte; returns
 * result() after the last. */
int32_t ref_stateful(void *ex, const void *text, size_t n, int32_t *initial, int32_t *perByte) {
  StatefulMatcher sm(*static_cast<Executable *>(ex));
  if (initial) *initial = sm.result();
  const Byte *p = static_cast<const Byte *>(text);
  for (size_t i = 0; i < n; ++i) {
    Result r = sm.advance(p[i]);
    if (perByte) perByte[i] = r;
  }
  return sm.result();
}

/* Batch forms: the outer per-input loop of tools/bench.cpp:60-71 / thr_red.cpp:36-47,
 * with N std::threads over contiguous shards exactly as thr_red.cpp:86-91 does.
 * verb: 0 check, 1 match, 2 scan, 3 search.  start/end may be NULL. */
void ref_batch(void *exv, int verb, int style, int lead, const uint8_t *data,
               const uint64_t *offsets, uint64_t stride, uint64_t lineLen, uint64_t n,
               int32_t *res, uint64_t *start, uint64_t *end, int nthreads) {
  const Executable &ex = *static_cast<Executable *>(exv);
  parallelFor(n, nthreads, [&](uint64_t lo, uint64_t hi) {
    for (uint64_t i = lo; i < hi; ++i) {
      const uint8_t *p;
      size_t len;
      lineOf(data, offsets, stride, lineLen, i, p, len);
      switch (verb) {
      case 0: res[i] = doCheck(ex, p, len, style, lead); break;
      case 2: res[i] = doScan(ex, p, len, style, lead); break;
      default: {
        Outcome oc = (verb == 1) ? doMatch(ex, p, len, style, lead)
                                 : doSearch(ex, p, len, style, lead);
        res[i] = oc.result_;
        if (start) start[i] = oc.start_;
        if (end) end[i] = oc.end_;
      }
      }
    }
  });
}

} // extern "C"
