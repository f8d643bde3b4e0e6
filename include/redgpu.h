/* redgpu.h - C-ABI of the MI355X-native DFA match-execution path for RED (zezax/one, quol/red).
 *
 * This is the drop-in boundary.  RED has no FFI layer of its own: the seam is the C++ value
 * type `Executable` (a validated serialized-DFA blob, include/Executable.h:28-76) and the free
 * functions `check / match / scan (const Executable&, const void* ptr, size_t len[, Style])`
 * (include/Matcher.h:79-98,133-169).  Every entry point below names the reference interface it
 * replaces; citations are relative to /root/reference/quol/red/.  The reference-side binding a
 * maintainer would add is shown in INTEGRATION.md; the C++ mirror of the reference's names is
 * include/redgpu.hpp.
 *
 * Conventions
 *   - plain pointers and sizes only; every function returns 0 (REDGPU_OK) or a negative code
 *     and leaves a message for redgpu_last_error() (thread-local).  The reference signals the
 *     same conditions with exceptions (include/Except.h:9-19): EAPI <-> RedExceptApi,
 *     EEXEC <-> RedExceptExec, ELIMIT <-> RedExceptLimit.  "No match" is result 0, not an error.
 *   - a redgpu_dfa is immutable after creation; all *_batch calls are re-entrant on a shared
 *     handle from any number of host threads (the reference's threading contract,
 *     doc/Performance.md:81-84, tools/thr_red.cpp:86-91).
 *   - inputs: line i is data[offsets[i], offsets[i+1]) when offsets != NULL (n+1 entries),
 *     else data[i*stride, (i+1)*stride).  Inputs/outputs are caller-owned and never retained.
 *     A single ragged line of 4 GiB or more: the host-buffer entry points route the batch to
 *     the general kernel (64-bit positions); through the _dev entry points such a batch must be
 *     created with REDGPU_F_FORCE_GENERIC (the block-wise kernels keep positions in 32 bits and
 *     cannot see the offsets before the launch).
 *     Device-pointer (_dev) entry points may read any byte of data[0, offsets[n]) (idle lanes
 *     re-read the first block), so the whole range must be device memory even when
 *     offsets[0] > 0; the host-buffer entry points copy only [offsets[0], offsets[n]).
 *   - outputs are the fields of the reference's Outcome (include/Outcome.h:32-35):
 *     result int32, start/end uint64 (size_t).  `start` and `end` may be NULL.
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point fails
 *     with REDGPU_EHIP.
 */
#ifndef REDGPU_H
#define REDGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REDGPU_OK      0
#define REDGPU_EAPI   (-1) /* bad blob / bad arguments      (RedExceptApi)   */
#define REDGPU_EEXEC  (-2) /* unsupported format / style    (RedExceptExec)  */
#define REDGPU_ELIMIT (-3) /* capacity exceeded             (RedExceptLimit) */
#define REDGPU_EHIP   (-5) /* HIP runtime / device failure  (new)            */
#define REDGPU_ERCCL  (-6) /* RCCL unavailable or failed    (new, groups)    */

/* include/Matcher.h:67-74 */
#define REDGPU_STY_INSTANT 1
#define REDGPU_STY_FIRST   2
#define REDGPU_STY_TANGENT 3
#define REDGPU_STY_LAST    4
#define REDGPU_STY_FULL    5

#define REDGPU_DEVICE_CURRENT (-1) /* upload to the calling thread's current HIP device */
#define REDGPU_DEVICE_NONE    (-2) /* validate + repack on the host only (no HIP call);
                                      batch calls on such a handle fail with REDGPU_EAPI */

typedef struct redgpu_dfa redgpu_dfa;

typedef struct redgpu_opts {
  int32_t  device;        /* HIP device ordinal, or REDGPU_DEVICE_CURRENT / _NONE */
  uint32_t lds_table_max; /* 0 = default; cap (bytes) on the LDS-resident table */
  uint32_t flags;         /* REDGPU_F_* */
  uint32_t reserved[5];
} redgpu_opts;

#define REDGPU_F_FORCE_GENERIC 1u /* never pick the specialised fixed-stride kernels */
#define REDGPU_F_FORCE_GLOBAL  2u /* keep the transition table in HBM/L2 even if it fits LDS */
#define REDGPU_F_NO_BUCKETING  8u /* ragged fast path: walk lines in input order instead of
                                     bucketing them by length first (tests, tuning) */
#define REDGPU_F_FORCE_STREAM 16u /* whole-line streaming kernels even for a DFA flagged
                                     early_death (tests, tuning) */
#define REDGPU_F_NO_CHUNKING  32u /* never cut long lines into speculatively walked chunks */
#define REDGPU_F_FORCE_CHUNKING 64u /* ... or always, whatever the DFA looks like (tests) */
#define REDGPU_F_STREAM_CHAINS_2 128u /* fixed-stride hot path: always the two-chain kernel ... */
#define REDGPU_F_STREAM_CHAINS_4 256u /* ... or always the four-chain one, whatever the batch size
                                         (default: by batch size; tests, tuning) */
#define REDGPU_F_FORCE_EARLY 512u /* match: the probe-and-drain kernel of the early-death DFAs for any
                                    LDS-resident table and any batch size (tests, tuning) */
#define REDGPU_F_FORCE_LEAN 1024u /* fixed-stride lines of 512 bytes and more: the deferred-bookkeeping
                                     step (k_stream_lean.h: 2.75 VALU per byte instead of 6, the two
                                     pieces that hold a line's last accept and last start re-walked at
                                     its end) instead of the exact one.  Built because the exact step
                                     looked issue-bound; measured it is not - 4 KiB lines +2 % (URI-D)
                                     / -3 % (SYN-256), 512-byte lines -8 % - so it is opt-in */
#define REDGPU_F_LEAN_CHAINS_4 2048u /* ... with four lines per lane (two chain groups, counted
                                     waits) instead of two */
#define REDGPU_F_FORCE_PIECES 4096u /* ragged lines: huge lines are walked in pieces (k_ragged.h) even
                                     when the DFA is not flagged `forgetful` - the entry-state
                                     guesses are then mostly wrong and the pieces walked again
                                     one by one: correct, slow (tests) */
#define REDGPU_F_FORCE_HOT     4u /* a table too big for LDS always gets hot rows in LDS, even
                                     when the visit model finds no locality (tests, tuning) */

/* table placements reported by redgpu_dfa_info */
#define REDGPU_TAB_LDS_FUSED_U8   1 /* [state][byte] -> next state, u8,  in LDS */
#define REDGPU_TAB_LDS_FUSED_U16  2 /* [state][byte] -> next state, u16, in LDS */
#define REDGPU_TAB_LDS_CLASS_U16  3 /* [state][class] u16 + equivalence map, both in LDS */
#define REDGPU_TAB_GLOBAL_U16     4 /* [state][class] u16 in HBM/L2, equivalence map in LDS */
#define REDGPU_TAB_GLOBAL_U32     5 /* [state][class] u32 in HBM/L2, equivalence map in LDS */
#define REDGPU_TAB_HOT_ROWS       6 /* [state][class] u16 in HBM/L2 for every state, plus a 64 KB
                                       [hot index][byte] u8 table in LDS over the n_hot (<= 254)
                                       most-visited states (device states [hot_lo, hot_lo +
                                       n_hot)): entry = hot index of the target, 255 = the
                                       target is not hot (look it up in the class table) */
#define REDGPU_TAB_LDS_SPARSE     7 /* class table too big for LDS but sparse (signature sets:
                                       94 % of LOG-100's transitions lead to the dead state):
                                       row-displacement ("comb") form in LDS - base[state] u16,
                                       then u32 slots (owner state << 16 | target) at
                                       base + class; a slot owned by another state means the
                                       DFA's most common target */

typedef struct redgpu_info {
  uint32_t format;        /* 1, 2, 4: FileHeader.format_ (include/Serializer.h:34-40,47) */
  uint32_t n_classes;     /* maxChar_ + 1 */
  uint32_t leader_len;
  uint32_t states_total;  /* FileHeader.stateCnt_ */
  uint32_t states_used;   /* reachable from the initial state (what the device image holds) */
  uint32_t n_pure_dead;   /* device states [0, n_pure_dead) are pure dead ends (Proxy.h:139) */
  uint32_t first_accept;  /* device states [first_accept, states_used) have result > 0 */
  uint32_t table_kind;    /* REDGPU_TAB_* */
  uint64_t table_bytes;
  int32_t  max_result;
  int32_t  device;
  uint32_t checksum;      /* FileHeader.checksum_ */
  uint32_t fast_path;     /* 1 if the fixed-stride specialised kernels apply to this DFA */
  uint32_t n_hot;         /* REDGPU_TAB_HOT_ROWS: rows resident in LDS (else 0) */
  uint32_t hot_lo;        /* REDGPU_TAB_HOT_ROWS: first device state with an LDS row */
  uint32_t hot_coverage_ppm; /* REDGPU_TAB_HOT_ROWS: share of the modelled visits (random-byte
                             walk from the initial state) that land on LDS rows, per million */
  uint32_t early_death;   /* 1 if the same model sees most walks reach a pure dead end within 16
                             bytes (an anchored DFA on arbitrary text): such DFAs keep the
                             early-exit kernels */
  uint32_t forgetful;     /* 1 if the model's walk is back in the initial state >= 70 % of the
                             time: long lines may be cut into speculatively walked chunks */
  uint32_t image_refs;    /* handles currently sharing this handle's device image (the loader
                             cache: same blob + device + build options -> one repack, one upload) */
  uint32_t suffix_closed; /* 1 if the DFA accepts behind any prefix whatever it accepts (L = SIGMA* L:
                             patterns added with a loose start).  scan / search / collect then stop at
                             the first attempt that reaches the end of the line without accepting -
                             no later start position can accept; the reference walks them all
                             (include/Matcher.h:511-553, :575-621) to the same result */
} redgpu_info;

/* Replaces checkHeader (include/Serializer.h:109, lib/Serializer.cpp:270-298): returns
 * REDGPU_OK, or REDGPU_EAPI with *msg (if msg != NULL) pointing at a static string that is
 * byte-for-byte the reference's message. */
int redgpu_reda_check(const void *reda, size_t len, const char **msg);

/* Loader cache (cf. the reference's load path lib/Serializer.cpp:257-267, which
 * builds a fresh Executable per call): handles created from byte-identical blobs on the same
 * device with the same lds_table_max and FORCE_GLOBAL / FORCE_HOT flags share one validated
 * copy, one repacked image and one set of device tables (redgpu_info.image_refs counts them);
 * the image is released with its last handle.  redgpu_dfa_tune detaches the tuned handle. */

/* Replaces Executable(gCopyTag, string_view) + Executable::validate
 * (include/Executable.h:37, lib/Executable.cpp:53-70,159-170): validates exactly as
 * checkHeader does, additionally bounds-checks every row offset (the GPU must never chase a
 * corrupt pointer), COPIES the blob, repacks it and uploads the image to opts->device.
 * opts == NULL means {REDGPU_DEVICE_CURRENT, 0, 0}.  The caller's buffer may be freed or
 * scrambled immediately (test/red.cpp:55-78). */
int redgpu_dfa_create(const void *reda, size_t len, const redgpu_opts *opts, redgpu_dfa **out);

/* Replaces ~Executable (lib/Executable.cpp:127-139). */
void redgpu_dfa_destroy(redgpu_dfa *dfa);

/* Replaces the Executable accessors (include/Executable.h:52-60). */
int redgpu_dfa_info(const redgpu_dfa *dfa, redgpu_info *out);

/* No counterpart in the reference - there the table lives in CPU caches, which adapt to the
 * input on their own.  For a DFA placed as REDGPU_TAB_HOT_ROWS this is the explicit version:
 * walks a SAMPLE of real input (anchored, as match<styLast,false> does: include/Matcher.h:413-495)
 * counting visits per state, then re-ranks the hot rows by those counts and rebuilds and
 * re-uploads the image.  Results never change, only which transitions are served from LDS.
 * Not re-entrant: call it before the handle is shared between threads; it waits for all work
 * queued on the device.  State tokens of redgpu_advance_batch taken before the call are
 * invalid after it.  Other placements: validates its arguments and does nothing. */
int redgpu_dfa_tune(redgpu_dfa *dfa, const uint8_t *data, const uint64_t *offsets,
                    uint64_t stride, uint64_t n);
int redgpu_dfa_tune_dev(redgpu_dfa *dfa, const uint8_t *data, const uint64_t *offsets,
                        uint64_t stride, uint64_t n, void *stream);

/* Replaces Executable::serialized() (include/Executable.h:50): the handle's own copy. */
int redgpu_dfa_serialized(const redgpu_dfa *dfa, const void **reda, size_t *len);

/* ---- batch verbs over HOST buffers (allocates device staging, copies in and out) ---------
 * Each replaces the caller's per-input loop (tools/bench.cpp:60-71, tools/thr_red.cpp:36-47)
 * around one reference function:
 *   redgpu_check_batch <-> check<style,doLeader>(exec, ptr, len)  include/Matcher.h:133-134,363-410
 *   redgpu_match_batch <-> match<style,doLeader>(exec, ptr, len)  include/Matcher.h:146-147,413-495
 *                          (style 4 = styLast is the "matchLong" of BASELINE.json)
 *   redgpu_scan_batch  <-> scan<style,doLeader>(exec, ptr, len)   include/Matcher.h:159-160,498-554
 * The run-time-style overloads of the reference (Matcher.h:79-92, Matcher.cpp:53-67) are these
 * with do_leader = 1.
 * Lines: offsets == NULL -> line i = data[i*stride, (i+1)*stride).  offsets != NULL (n + 1
 * entries) -> line i = data[offsets[i], offsets[i+1] - stride): here `stride` is the number of
 * trailing bytes each line carries that the matcher must not see - 0 normally, 1 for the
 * delimiter-terminated lines redgpu_split_lines produces. */
int redgpu_check_batch(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                       const uint64_t *offsets, uint64_t stride, uint64_t n, int32_t *result);
int redgpu_match_batch(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                       const uint64_t *offsets, uint64_t stride, uint64_t n, int32_t *result,
                       uint64_t *start, uint64_t *end);
int redgpu_scan_batch(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                      const uint64_t *offsets, uint64_t stride, uint64_t n, int32_t *result);
/*   redgpu_search_batch <-> search<style,doLeader>(exec, ptr, len) include/Matcher.h:172-173,557-640
 *                           (scan with positions; the first of SURVEY 8(f)'s "next" rows) */
int redgpu_search_batch(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                        const uint64_t *offsets, uint64_t stride, uint64_t n, int32_t *result,
                        uint64_t *start, uint64_t *end);

/*   redgpu_collect_batch <-> Red::collect(string_view, vector<Outcome>&)  include/Red.h:115,
 *                            lib/Red.cpp:103-116: every non-overlapping match of each line, in
 *                            order (repeated search<styLast,false>).  Line i's records go to
 *                            result/start/end[i*cap .. i*cap+cap); counts[i] = matches FOUND,
 *                            which may exceed cap (then only the first cap are stored). */
int redgpu_collect_batch(const redgpu_dfa *dfa, const uint8_t *data, const uint64_t *offsets,
                         uint64_t stride, uint64_t n, uint64_t cap, uint64_t *counts,
                         int32_t *result, uint64_t *start, uint64_t *end);

/*   redgpu_match_all_batch <-> matchAll(exec, string_view, vector<Outcome>&)  include/Matcher.h:127,
 *                              lib/Matcher.cpp:97-102, core include/Matcher.h:711-766 (also
 *                              Red::allMatches, lib/Red.cpp:713-715): ONE anchored walk per line
 *                              reporting every maximal run of bytes with the same accepted result
 *                              (RE2::Set::Match style).  The reference's public entry always
 *                              runs with doLeader = true: pass do_leader = 1 for parity with it.
 *                              Record layout and counts[] exactly as redgpu_collect_batch. */
int redgpu_match_all_batch(const redgpu_dfa *dfa, int do_leader, const uint8_t *data,
                           const uint64_t *offsets, uint64_t stride, uint64_t n, uint64_t cap,
                           uint64_t *counts, int32_t *result, uint64_t *start, uint64_t *end);

/*   redgpu_advance_batch <-> StatefulMatcher (include/Matcher.h:770-792, lib/Matcher.cpp:106-158),
 *                            n independent matchers advanced by one CHUNK each: state[i] is
 *                            matcher i's state_ on entry and on return (an opaque token, valid
 *                            only with the handle that produced it; REDGPU_STATE_INITIAL = a
 *                            freshly constructed StatefulMatcher - a buffer memset to 0xff is n
 *                            fresh matchers), result[i] = result() after the chunk's last byte.
 *                            Feeding a stream in chunks of any sizes gives the same states and
 *                            results as advance() byte by byte: inputs larger than device memory
 *                            are walked chunk by chunk. */
#define REDGPU_STATE_INITIAL 0xffffffffu
int redgpu_advance_batch(const redgpu_dfa *dfa, const uint8_t *data, const uint64_t *offsets,
                         uint64_t stride, uint64_t n, uint32_t *state, int32_t *result);

/*   redgpu_replace_batch <-> replace<style,doLeader>(exec, ptr, len, repl, out, max)
 *                            include/Matcher.h:186-191, core :643-706 (run-time-style overloads:
 *                            do_leader = 1, lib/Matcher.cpp:72-92): every line rewritten with each
 *                            match replaced by `repl`, at most max_count replacements per line.
 *                            counts[i] = replacements made in line i (the function's return
 *                            value); out_offsets[n + 1] = exclusive scan of the rewritten lengths
 *                            (out_offsets[n] = total bytes); line i's result is
 *                            out[out_offsets[i] .. out_offsets[i + 1]).  out may be NULL (sizes
 *                            only); with out != NULL every line that fits entirely below out_cap
 *                            is written - check out_offsets[n] <= out_cap, else call again with
 *                            a buffer of out_offsets[n] bytes. */
int redgpu_replace_batch(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                         const uint64_t *offsets, uint64_t stride, uint64_t n, const uint8_t *repl,
                         uint64_t repl_len, uint64_t max_count, uint64_t *counts,
                         uint64_t *out_offsets, uint8_t *out, uint64_t out_cap);

/* ---- the same verbs over DEVICE-resident buffers, asynchronous on `stream` ---------------
 * data/offsets/result/start/end are device pointers on the handle's device; `stream` is a
 * hipStream_t (NULL = the default stream).  Nothing is copied or synchronised; the only
 * allocation is the scratch some launches need (the ragged kernels' tail pad, their list of long
 * lines and the records of huge lines walked in pieces - up to ~16 bytes per line of the batch -,
 * the chunked walk's records, replace's partial sums, line splitting's delimiter masks - 1/8 of
 * the text): kept per HOST THREAD and (device, stream)
 * and re-used by that thread's later calls on the stream - so any number of threads may issue
 * _dev calls on one stream, the default one included - (re)allocated, with a device
 * synchronisation, only when a call needs more than the cached buffer holds; freed when the
 * thread exits or calls redgpu_thread_release(). */
int redgpu_check_batch_dev(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                           const uint64_t *offsets, uint64_t stride, uint64_t n,
                           int32_t *result, void *stream);
int redgpu_match_batch_dev(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                           const uint64_t *offsets, uint64_t stride, uint64_t n,
                           int32_t *result, uint64_t *start, uint64_t *end, void *stream);
int redgpu_scan_batch_dev(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                          const uint64_t *offsets, uint64_t stride, uint64_t n, int32_t *result,
                          void *stream);
int redgpu_search_batch_dev(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                            const uint64_t *offsets, uint64_t stride, uint64_t n,
                            int32_t *result, uint64_t *start, uint64_t *end, void *stream);

/* ---- several device-resident batches per call ----------------------------------------------
 * The reference's callers hold many inputs and loop over them (tools/bench.cpp:60-71: every
 * line of the file, `iters` times; tools/thr_red.cpp:36-47).  A caller that holds K buffers of
 * lines hands all K over at once:
 *   redgpu_check_batches_dev <-> for each batch: check<style,doLeader> per line, Matcher.h:363-410
 *   redgpu_match_batches_dev <-> for each batch: match<style,doLeader> per line, Matcher.h:413-495
 * The results are exactly those of n_batches calls of the single-batch entry point in the order
 * given, on `stream` (a batch may read what an earlier one wrote only through the single-batch
 * calls: the batches of ONE call must not alias each other's outputs).  What the call buys: runs
 * of consecutive batches that the streaming kernel takes - fixed stride, the same for all, a
 * multiple of 64 bytes; styLast / styFull; the same outputs asked for (start NULL in all or in
 * none) - go out as ONE launch per 32 batches in which the DFA table is staged once per CU and
 * tiles of lines are handed out across batch boundaries, so the head and tail a 64 MiB launch
 * pays (a third of it) are paid once per call.  Anything else in the list runs as its own launch,
 * as redgpu_*_batch_dev would run it.  start / end of redgpu_batch are ignored by check. */
typedef struct redgpu_batch {
  const uint8_t  *data;
  const uint64_t *offsets; /* n + 1 entries, or NULL: fixed stride */
  uint64_t stride;
  uint64_t n;
  int32_t  *result;
  uint64_t *start;         /* may be NULL */
  uint64_t *end;           /* may be NULL */
} redgpu_batch;
int redgpu_check_batches_dev(const redgpu_dfa *dfa, int style, int do_leader,
                             const redgpu_batch *batches, uint32_t n_batches, void *stream);
int redgpu_match_batches_dev(const redgpu_dfa *dfa, int style, int do_leader,
                             const redgpu_batch *batches, uint32_t n_batches, void *stream);

int redgpu_collect_batch_dev(const redgpu_dfa *dfa, const uint8_t *data, const uint64_t *offsets,
                             uint64_t stride, uint64_t n, uint64_t cap, uint64_t *counts,
                             int32_t *result, uint64_t *start, uint64_t *end, void *stream);

int redgpu_match_all_batch_dev(const redgpu_dfa *dfa, int do_leader, const uint8_t *data,
                               const uint64_t *offsets, uint64_t stride, uint64_t n,
                               uint64_t cap, uint64_t *counts, int32_t *result, uint64_t *start,
                               uint64_t *end, void *stream);

int redgpu_advance_batch_dev(const redgpu_dfa *dfa, const uint8_t *data, const uint64_t *offsets,
                             uint64_t stride, uint64_t n, uint32_t *state, int32_t *result,
                             void *stream);

/* repl is device memory too */
int redgpu_replace_batch_dev(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                             const uint64_t *offsets, uint64_t stride, uint64_t n,
                             const uint8_t *repl, uint64_t repl_len, uint64_t max_count,
                             uint64_t *counts, uint64_t *out_offsets, uint8_t *out,
                             uint64_t out_cap, void *stream);

/* The step BEFORE the path, on the device (the reference does it on the host:
 * lib/Util.cpp:109-130 sampleLines, and every tool that walks a text blob line by line): finds
 * the delimiters of a raw text buffer and writes the offsets[] array the ragged verbs take.
 * A line is [start, delimiter); the next starts after the delimiter; bytes after the last
 * delimiter are not a line (sampleLines' rule).  offsets[0] = 0 and offsets[k+1] = position
 * just past the k-th delimiter, so line k = [offsets[k], offsets[k+1]) INCLUDES its delimiter:
 * pass these offsets to the *_batch verbs with stride = 1 ("one trailing byte per line that the
 * matcher must not see").  cap = lines offsets[] has room for (cap + 1 entries); *n_lines =
 * delimiters FOUND, which may exceed cap (then only the first cap lines are stored).
 * _dev: everything device-resident (n_lines too), asynchronous on `stream`. */
int redgpu_split_lines(const redgpu_dfa *dfa, const uint8_t *data, uint64_t len, uint8_t delim,
                       uint64_t *offsets, uint64_t cap, uint64_t *n_lines);
int redgpu_split_lines_dev(const redgpu_dfa *dfa, const uint8_t *data, uint64_t len, uint8_t delim,
                           uint64_t *offsets, uint64_t cap, uint64_t *n_lines, void *stream);

/* Raw text in, one Outcome per line out, in ONE call: the loop of tools/skim_red.cpp:36-46 over
 * the lines lib/Util.cpp:109-130 cuts (find the delimiter, hand [start, delimiter) to the
 * matcher, go on behind it).  Equal to redgpu_split_lines[_dev] followed by
 * redgpu_check/match_batch[_dev] over (offsets, stride = 1) - same offsets[] (cap + 1 entries, the
 * caller's: positions are relative to a line's start, so the caller needs them to place a match
 * in the buffer), same *n_lines (delimiters found, may exceed cap), result/start/end filled for
 * the first min(*n_lines, cap) lines - but the line count never visits the host in between:
 * for the DFAs the ragged streaming kernels take (styles Last / Full, no leader to honour, table
 * in LDS or hot rows - redgpu_last_kernel says "k_ragged...") the walk reads it on the device and
 * the _dev call is asynchronous on `stream` from end to end; for the others the call waits for
 * the split once (8 bytes back) and launches what redgpu_*_batch_dev would.
 * _dev: all pointers device memory.  redgpu_match_text: host buffers; start / end may be NULL
 * (both NULL = check). */
int redgpu_check_text_dev(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                          uint64_t len, uint8_t delim, uint64_t *offsets, uint64_t cap,
                          uint64_t *n_lines, int32_t *result, void *stream);
int redgpu_match_text_dev(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                          uint64_t len, uint8_t delim, uint64_t *offsets, uint64_t cap,
                          uint64_t *n_lines, int32_t *result, uint64_t *start, uint64_t *end,
                          void *stream);
int redgpu_match_text(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                      uint64_t len, uint8_t delim, uint64_t *offsets, uint64_t cap,
                      uint64_t *n_lines, int32_t *result, uint64_t *start, uint64_t *end);

/* Measurement aid, no counterpart in the reference: one streaming read of `bytes` of device
 * memory (16-byte aligned) on the handle's device, asynchronous on `stream` - the read-bandwidth
 * calibration bench.py reports beside the roofline.  `sink` is a device uint32 the kernel may
 * increment (it keeps the loads alive); its value carries no meaning. */
int redgpu_diag_read_dev(const redgpu_dfa *dfa, const void *data, uint64_t bytes, uint32_t *sink,
                         void *stream);

/* More calibration kernels behind bench.py's roofline block (device-resident, asynchronous):
 *  - redgpu_diag_lds_dev: the walk without its memory side - `rounds` x 64 dependent table
 *    lookups per chain, 4 chains per lane, 512 lanes per CU, input bytes from registers; times
 *    the LDS gather rate that bounds a one-lookup-per-byte walk.  *lookups (host) receives the
 *    number of lookups the launch performs.
 *  - redgpu_diag_walked_dev: adds to *walked (device uint64, zeroed by the caller) the bytes the
 *    loop of match<styLast,doLeader> (include/Matcher.h:424-479) consumes over the batch: the
 *    bytes an early-exit walk actually reads, as opposed to the bytes of the lines. */
int redgpu_diag_lds_dev(const redgpu_dfa *dfa, uint32_t rounds, uint32_t *sink, uint64_t *lookups,
                        void *stream);
/*  - redgpu_diag_lines_dev: the memory side of the fixed-stride walk and nothing else.
 *    line_bytes = 64: every lane requests its line the way the streaming kernel does and stores
 *    one Outcome-shaped record per line (result int32, start / end uint64; the values are
 *    meaningless) - 64 B read + 20 B written per line; what HBM sustains for that mix is the roof
 *    of BASELINE configs[1]'s shape.  line_bytes a multiple of 128: the long-line request pattern
 *    (a lane asks for one whole cache line of each of its two lines at a time, so a wave touches
 *    64 cache lines a line length apart), reads only (result / start / end may be NULL) - the
 *    roof of configs[2]'s and configs[4]'s shapes.  n_lines is rounded down to a multiple of
 *    1024. */
int redgpu_diag_lines_dev(const redgpu_dfa *dfa, const uint8_t *data, uint64_t n_lines,
                          uint64_t line_bytes, int32_t *result, uint64_t *start, uint64_t *end,
                          uint32_t *sink, void *stream);
/*  - redgpu_diag_l2_dev: the gather rate of a table that lives in L2 - `rounds` DEPENDENT 2-byte
 *    gathers per chain (the next index is computed from the value just read, as a walk's next
 *    state is) over `table`, 2^20 uint16 of device memory (2 MiB, any contents), 2 chains per lane,
 *    2048 lanes per CU, no input side: the roof of a one-lookup-per-byte walk over a
 *    REDGPU_TAB_GLOBAL_* DFA (SYN-4K), measured in the run instead of derived from L2's nominal
 *    bandwidth.  *lookups (host) = gathers performed. */
int redgpu_diag_l2_dev(const redgpu_dfa *dfa, const uint16_t *table, uint32_t rounds, uint32_t *sink,
                       uint64_t *lookups, void *stream);
int redgpu_diag_walked_dev(const redgpu_dfa *dfa, int do_leader, const uint8_t *data,
                           const uint64_t *offsets, uint64_t stride, uint64_t n, uint64_t *walked,
                           void *stream);

/* The host-buffer entry points stage through device buffers and two private streams that each
 * HOST THREAD keeps per device (no allocation, stream creation or device-wide synchronisation per
 * call).  They - and the thread's launch scratch - are released when the thread exits; a
 * long-lived thread that is done with the library may release them early.
 * redgpu_scratch_entries: launch scratch buffers currently cached by all threads together
 * (bounded per thread; diagnostics / tests). */
void redgpu_thread_release(void);
uint64_t redgpu_scratch_entries(void);
/* Diagnostics: how the host-buffer entry points have moved caller memory so far, transfers by
 * route - [0] memory the caller pinned, [1] through the calling thread's pinned arena (calls that
 * upload < 8 MiB), [2] registered for the duration of the call, [3] handed to the runtime
 * pageable (nothing else was possible). */
void redgpu_host_route_counts(uint64_t counts[4]);

/* A caller that keeps its buffers from call to call (tools/thr_red.cpp's workers do) can pin them
 * ONCE: the host-buffer entry points recognise pinned memory (registered here, or allocated by
 * hipHostMalloc / torch's pin_memory) and then cut even a batch of a few MiB into 8 MiB chunks
 * whose uploads, walks and downloads overlap on the thread's two streams - pageable memory is
 * only pipelined above 64 MiB, where pinning it for the duration of one call pays.  The range
 * must stay mapped until redgpu_host_unregister; (un)registering costs about a millisecond per
 * 16 MiB, which is why the library does not do it per call on its own. */
int redgpu_host_register(void *ptr, size_t bytes);
int redgpu_host_unregister(void *ptr);

/* ---- several GPUs of one node --------------------------------------------------------------
 * The reference scales by calling its read-only matcher from N threads over one shared Red
 * (tools/thr_red.cpp:84-91).  The device form of that picture: a GROUP holds one image of the
 * same blob per device ("uploaded once" per GPU); a batch is cut into contiguous shards - equal
 * line counts for a fixed stride, equal BYTES for ragged lines - every device scans its shard
 * with no data-path exchange, and only the per-line results travel.
 *   redgpu_group_create      devices[] may name a device more than once (the shards then share
 *                            that GPU: how a one-GPU box rehearses the sharding); opts->device
 *                            is ignored.  Fails as redgpu_dfa_create does.
 *   redgpu_group_plan        cuts[0..n_devices]: shard g = lines [cuts[g], cuts[g+1]).
 *   redgpu_group_batch       HOST buffers, verb = REDGPU_VERB_*: one host thread per device runs
 *                            its shard through the host-buffer entry point of that verb; results
 *                            land in the caller's arrays (start / end NULL for check and scan).
 *   redgpu_group_batch_dev   shard g already resident on device g (data[g], offsets[g] or NULL,
 *                            n[g] lines; ragged offsets are relative to data[g]).  Every device
 *                            scans on a stream of its own; results are packed into compact
 *                            records (result in 1/2/4 bytes by the DFA's largest result, start /
 *                            end in 1/2/4 bytes by the stride, 4 bytes for ragged lines - 8 on a
 *                            handle made with REDGPU_F_FORCE_GENERIC), moved to the ROOT device
 *                            (devices[0]) over xGMI - REDGPU_GATHER_PEER: peer copies,
 *                            REDGPU_GATHER_RCCL: ncclSend / ncclRecv (librccl.so loaded on first
 *                            use) - and widened there into result / start / end (memory of the
 *                            root device, sum(n[]) entries, shard order).  Asynchronous, no host
 *                            synchronisation: the results are complete when `root_stream` (a
 *                            stream of the root device, NULL = its null stream) has reached this
 *                            point.  shard_streams[g] (a stream of device g, or shard_streams ==
 *                            NULL): the stream on which shard g's inputs were produced and on
 *                            which the caller will touch them next - the shard's scan starts where
 *                            that stream stands at the call, and the stream in turn waits until
 *                            the scan has read the shard, so inputs may be reused or freed
 *                            stream-ordered.  With shard_streams == NULL every input must be
 *                            complete at the call and stay untouched until root_stream has passed
 *                            it.  One such call at a time per group (serialised internally).
 *                            The peer / RCCL paths between DISTINCT devices have not run on
 *                            hardware yet (single-GPU boxes only): see INTEGRATION.md. */
typedef struct redgpu_group redgpu_group;
#define REDGPU_VERB_CHECK  0
#define REDGPU_VERB_MATCH  1
#define REDGPU_VERB_SCAN   2
#define REDGPU_VERB_SEARCH 3
#define REDGPU_GATHER_PEER 0
#define REDGPU_GATHER_RCCL 1
int redgpu_group_create(const void *reda, size_t len, const redgpu_opts *opts,
                        const int32_t *devices, uint32_t n_devices, redgpu_group **out);
void redgpu_group_destroy(redgpu_group *group);
uint32_t redgpu_group_size(const redgpu_group *group);
const redgpu_dfa *redgpu_group_member(const redgpu_group *group, uint32_t i);
int redgpu_group_plan(const redgpu_group *group, const uint64_t *offsets, uint64_t stride,
                      uint64_t n, uint64_t *cuts);
int redgpu_group_batch(const redgpu_group *group, int verb, int style, int do_leader,
                       const uint8_t *data, const uint64_t *offsets, uint64_t stride, uint64_t n,
                       int32_t *result, uint64_t *start, uint64_t *end);
int redgpu_group_batch_dev(redgpu_group *group, int verb, int style, int do_leader,
                           const uint8_t *const *data, const uint64_t *const *offsets,
                           uint64_t stride, const uint64_t *n, int32_t *result, uint64_t *start,
                           uint64_t *end, int gather, void *const *shard_streams,
                           void *root_stream);

/* The compact record format of redgpu_group_batch_dev, for callers that move the results
 * themselves (one process per GPU over torch.distributed / RCCL: one_amd/sharding.py).  A
 * shard's n Outcomes become three planes, each starting on a 16-byte boundary:
 *   [result: n x result_width][start: n x pos_width (absent when start == NULL)][end: n x pos_width]
 * little-endian, result_width in {1,2,4} (results are >= 0, include/Types.h:22), pos_width in
 * {1,2,4,8}; the caller picks widths that hold the DFA's largest result and the longest line.
 *   redgpu_records_bytes       size of a packed shard
 *   redgpu_records_pack_dev    device arrays -> packed (device) buffer, on `stream`
 *   redgpu_records_unpack_dev  packed buffer -> int32 result / 64-bit start, end (device), on `stream`
 * All pointers are memory of `device` (REDGPU_DEVICE_CURRENT = the caller's current device). */
uint64_t redgpu_records_bytes(uint64_t n, int result_width, int pos_width, int with_start);
int redgpu_records_pack_dev(int32_t device, const int32_t *result, const uint64_t *start,
                            const uint64_t *end, uint64_t n, int result_width, int pos_width,
                            void *records, void *stream);
int redgpu_records_unpack_dev(int32_t device, const void *records, uint64_t n, int result_width,
                              int pos_width, int32_t *result, uint64_t *start, uint64_t *end,
                              void *stream);

/* Name of the kernel the last *_batch* call on this thread launched (for profiles), or "". */
const char *redgpu_last_kernel(void);

/* Thread-local message of the last failing call on this thread ("" if none). */
const char *redgpu_last_error(void);

/* Library version: major*10000 + minor*100 + patch. */
int redgpu_version(void);

#ifdef __cplusplus
}
#endif
#endif /* REDGPU_H */
