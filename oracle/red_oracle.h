/* red_oracle.h - CPU restatement of RED's DFA match-execution path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, load or call this.  The product (one_amd/, include/) never does.
 *
 * Parity status: PINNED.  Checked in this container against the real reference compiled
 * from /root/reference (oracle/_ref, see oracle/ref_driver.cpp + tests/test_oracle_vs_ref.py)
 * and against the golden vectors transcribed from the reference's own tests
 * (tests/golden/, made by oracle/gen_golden.py).
 *
 * All file:line citations are relative to /root/reference/quol/red/.
 */
#ifndef RED_ORACLE_H
#define RED_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* include/Matcher.h:67-74 */
enum { ORA_STY_INSTANT = 1, ORA_STY_FIRST = 2, ORA_STY_TANGENT = 3, ORA_STY_LAST = 4,
       ORA_STY_FULL = 5 };

/* verbs for oracle_batch */
enum { ORA_CHECK = 0, ORA_MATCH = 1, ORA_SCAN = 2, ORA_SEARCH = 3 };

/* what Executable caches after validate(): include/Executable.h:52-60,
 * lib/Executable.cpp:159-170 */
typedef struct oracle_dfa {
  const uint8_t *blob;
  size_t         len;
  const uint8_t *equiv;   /* header + 32, 256 bytes */
  const uint8_t *leader;  /* header + 288, leaderLen bytes in class space (NULL if none) */
  const uint8_t *base;    /* header + 288 + pad8(leaderLen) */
  uint32_t       stateCnt;
  uint32_t       initialOff;
  uint32_t       leaderOff;
  uint8_t        fmt;     /* 1, 2 or 4 */
  uint8_t        maxChar;
  uint8_t        leaderLen;
} oracle_dfa;

/* include/Fnv.h:36-66 (32-bit parameters) */
uint32_t oracle_fnv1a32(const void *p, size_t n);
/* include/Fnv.h (64-bit parameters); known answers in test/fnv.cpp:11-21 */
uint64_t oracle_fnv1a64(const void *p, size_t n);

/* lib/Serializer.cpp:301-306 */
uint32_t oracle_calc_checksum(const void *blob, size_t len);

/* lib/Serializer.cpp:270-298.  NULL when good, else the reference's message, verbatim. */
const char *oracle_check_header(const void *blob, size_t len);

/* lib/Executable.cpp:159-170.  Returns NULL when good (d filled; d borrows blob). */
const char *oracle_dfa_init(oracle_dfa *d, const void *blob, size_t len);

/* include/Matcher.h:363-410 checkCore, :413-495 matchCore, :498-554 scanCore,
 * :557-640 searchCore; RangeIter input (include/Proxy.h:53-73).
 * Returns -1000 on unsupported style/format (the reference throws RedExceptExec). */
int32_t oracle_check(const oracle_dfa *d, const uint8_t *p, size_t n, int style, int doLeader);
int32_t oracle_scan(const oracle_dfa *d, const uint8_t *p, size_t n, int style, int doLeader);
int32_t oracle_match(const oracle_dfa *d, const uint8_t *p, size_t n, int style, int doLeader,
                     uint64_t *start, uint64_t *end);
int32_t oracle_search(const oracle_dfa *d, const uint8_t *p, size_t n, int style, int doLeader,
                      uint64_t *start, uint64_t *end);

/* lib/Red.cpp:103-116 Red::collect(string_view): all non-overlapping matches in order, by
 * repeated search<styLast,false> from the end of the previous match.  Writes at most `cap`
 * records; returns the number FOUND (may exceed cap). */
uint64_t oracle_collect(const oracle_dfa *d, const uint8_t *p, size_t n, uint64_t cap,
                        int32_t *res, uint64_t *start, uint64_t *end);

/* batch form of oracle_collect: per line i, records go to [i*cap, i*cap+cap), counts[i] = found */
void oracle_collect_batch(const oracle_dfa *d, const uint8_t *data, const uint64_t *offsets,
                          uint64_t stride, uint64_t lineLen, uint64_t n, uint64_t cap,
                          uint64_t *counts, int32_t *res, uint64_t *start, uint64_t *end);

/* include/Matcher.h:643-706 replaceCore (public: replace<style,doLeader>, and the run-time-style
 * overloads with doLeader = 1, lib/Matcher.cpp:72-92).  At most outCap bytes are written;
 * *outLen = full length of the result; returns the number of replacements, or (uint64_t)-1000
 * on an unsupported style / format. */
uint64_t oracle_replace(const oracle_dfa *d, const uint8_t *p, size_t n, int style, int doLeader,
                        const uint8_t *repl, size_t replLen, uint64_t max, uint8_t *out,
                        uint64_t outCap, uint64_t *outLen);

/* include/Matcher.h:711-766 matchAllCore; the public matchAll (lib/Matcher.cpp:97-102) runs it
 * with doLeader = 1.  Same cap / count convention as oracle_collect. */
uint64_t oracle_match_all(const oracle_dfa *d, const uint8_t *p, size_t n, int doLeader,
                          uint64_t cap, int32_t *res, uint64_t *start, uint64_t *end);
void oracle_match_all_batch(const oracle_dfa *d, int doLeader, const uint8_t *data,
                            const uint64_t *offsets, uint64_t stride, uint64_t lineLen,
                            uint64_t n, uint64_t cap, uint64_t *counts, int32_t *res,
                            uint64_t *start, uint64_t *end);

/* include/Matcher.h:770-792, lib/Matcher.cpp:106-158 StatefulMatcher: *state = state_ as a byte
 * offset from base (ORA_STATE_INITIAL = freshly constructed); one advance() per byte of p;
 * returns result(); perByte (may be NULL) gets every advance()'s return value. */
#define ORA_STATE_INITIAL 0xffffffffu
int32_t oracle_advance(const oracle_dfa *d, uint32_t *state, const uint8_t *p, size_t n,
                       int32_t *perByte);
void oracle_advance_batch(const oracle_dfa *d, const uint8_t *data, const uint64_t *offsets,
                          uint64_t stride, uint64_t lineLen, uint64_t n, uint32_t *state,
                          int32_t *res);

/* The callers' per-input loop (tools/bench.cpp:60-71, tools/thr_red.cpp:36-47,86-91):
 * line i = data[offsets[i], offsets[i+1]) or, with offsets == NULL,
 * data[i*stride, i*stride + lineLen).  start/end may be NULL.  nthreads contiguous shards. */
void oracle_batch(const oracle_dfa *d, int verb, int style, int doLeader, const uint8_t *data,
                  const uint64_t *offsets, uint64_t stride, uint64_t lineLen, uint64_t n,
                  int32_t *res, uint64_t *start, uint64_t *end, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
