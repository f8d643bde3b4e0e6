// kernels.h - launch interface between the C-ABI (redgpu.cpp) and the gfx950 kernels.
#pragma once

#include <cstdint>

#include <hip/hip_runtime.h>

namespace redgpu {

enum Verb : int { kCheck = 0, kMatch = 1, kScan = 2, kSearch = 3 };

// Device-resident DFA image (pointers are device addresses). Built once per redgpu_dfa.
struct DevDfa {
  const uint8_t *table;    // packed table, layout per tableKind (REDGPU_TAB_*)
  const int32_t *result;   // [nStates] result code per device state
  const uint8_t *equivLeader; // 256 bytes equivalence map, then 256 bytes class-space leader
  uint8_t *sink;           // 64 writable bytes nobody reads: where lanes with nothing to report
                           // store, so that result stores need no branch (k_stream.h)
  uint32_t tableKind;
  uint32_t tableBytes;
  uint32_t nStates;
  uint32_t nClasses;
  uint32_t init;
  uint32_t leaderNext;
  uint32_t nPureDead;
  uint32_t firstAccept;
  uint32_t leaderLen;
  uint32_t deadAbsorbing;
  // REDGPU_TAB_HOT_ROWS (dfa_image.h), else 0: hot states [hotLo, hotLo + nHot), the 64 KB
  // [hot index][byte] u8 table at table + hot8Off, hot index = s - hotLo + hotShift
  uint32_t hotLo, nHot, hot8Off, hotShift;
  uint32_t clsOff, clsRowBytes, clsBytes;  // streaming form of the class table, or 0
  uint32_t clsIndexForm;         // its entries are state indices (tables above 64 KB), not row offsets
  uint32_t sparseCombOff, sparseDefault;  // REDGPU_TAB_LDS_SPARSE: byte offset of the slots, default target
  uint32_t earlyDeath;           // the visit model sees walks die within 16 bytes
  uint32_t tuned;                // hot rows ranked by observed visits
  uint32_t forgetful;            // the walk is mostly in the initial state (k_chunk.h)
  uint32_t uniformResult;        // every accepting state has the same result (dfa_image.h)
  uint32_t suffixClosed;         // L = SIGMA* L (dfa_image.h): a failed attempt that reached the end
                                 // of the line ends scan / search / collect
  uint32_t gatherNt;             // REDGPU_GATHER_NT=1: non-temporal table gathers (tuning experiment)
  uint32_t leaderForced;         // the leader is the table's own forced prefix chain (dfa_image.h)
  // start bytes for scan / search (dfa_image.h): packed members, count (0xff = no filter)
  uint32_t startLeadWord, startLeadCount, startFreeWord, startFreeCount;
  uint32_t start2LeadWord, start2LeadCount, start2FreeWord, start2FreeCount;
  // full start / follow flag tables behind equivLeader: [512..768) without the leader,
  // [768..1024) with it (DfaImage::startFlags); how many start bytes; whether bit 1 means anything
  uint32_t startTotal[2], startFollow[2];
};

// One batch of lines (device pointers).
struct Batch {
  const uint8_t  *data;
  const uint64_t *offsets; // n+1 entries, or nullptr => fixed stride
  uint64_t stride;         // fixed stride; with offsets: trailing delimiter bytes each line
                           // carries and the verbs must not see (0, or 1 after redgpu_split_lines)
  uint64_t n;
  int32_t  *result;
  uint64_t *start;         // may be nullptr
  uint64_t *end;           // may be nullptr
  uint32_t *state = nullptr; // advance only: per-line StatefulMatcher state, in/out
  const uint32_t *perm = nullptr; // k_ragged only: slot -> line (lines bucketed by length)
  const uint8_t *pad = nullptr;   // k_ragged only: copy of the buffer's last 128 bytes + zeros
  // k_ragged only: the lines much longer than the batch's mean (k_ragged_outliers), walked first
  const uint64_t *outRec = nullptr; // [nOut][2] = the line's offsets[] pair
  const uint32_t *outLn = nullptr;  // [nOut] = its index
  const uint32_t *outCtl = nullptr; // [0] = nOut, [1] = the length from which a line is one
  // ... and the lines of at least 8 x that length, cut into pieces that are walked at once
  // (k_ragged_long.h "pieces"; DFAs that forget their past): records
  // per piece, folded into the lines' Outcomes by k_ragged_pieces_fold; states as global ids
  int32_t *pieceRes = nullptr;      // [nPieces] last accepting state | accepted << 31
  uint64_t *pieceEnd = nullptr;     // [nPieces] end of that accept, from the piece's first walked
                                    // byte | exit state << 32 | entry guess << 48
  uint64_t *pieceStart = nullptr;   // [nPieces] last "left the initial state", likewise
  uint32_t *hugeLn = nullptr;       // [nHuge] the line, [nHuge] its first piece (outCtl[2] = nHuge)
  uint32_t *hugeFirst = nullptr;
  // k_ragged family only (launchBatch answers hipErrorNotSupported for the others): the line
  // count is still on the device - min(*nDev, n) lines, n = what the arrays have room for
  // (redgpu_*_text_dev: the count comes from the line split queued just before)
  const uint64_t *nDev = nullptr;
  uint32_t spread = 1;            // k_generic only: one line per `spread` lanes (table in L2)
  uint32_t ignoreAcceptUpTo = 0;  // k_stream / k_stream_multi, check<..., true> over a forced leader:
                                  // accepts at positions <= this (the post-leader state's own
                                  // result, include/Matcher.h:370-381) are not seen
};

struct LaunchCfg {
  int numCUs;
  int forceGeneric;
  int noBucketing = 0;  // k_ragged: keep lines in input order (REDGPU_F_NO_BUCKETING)
  int forceStream = 0;  // whole-line kernels even for early-death DFAs (REDGPU_F_FORCE_STREAM)
  int noChunking = 0;   // never cut long lines into speculative chunks (REDGPU_F_NO_CHUNKING)
  int forceChunking = 0; // ... or whenever the shape allows, whatever the DFA (REDGPU_F_FORCE_CHUNKING)
  int forceEarly = 0;    // k_early for match over any LDS-resident table, whatever the DFA and
                         // the batch size (REDGPU_F_FORCE_EARLY; tests)
  int forceLean = 0;     // long fixed-stride lines: k_stream_multi's lean step (REDGPU_F_FORCE_LEAN)
  int leanChains4 = 0;   // ... with four lines per lane (REDGPU_F_LEAN_CHAINS_4)
  int streamChains = 0;  // fixed-stride hot path: 0 = by batch size, 2 = k_stream.h always,
                         // 3 / 4 = k_stream4.hip always (REDGPU_F_STREAM_CHAINS_*; tests, tuning)
  int forcePieces = 0;   // k_ragged: huge lines in pieces whatever the DFA (REDGPU_F_FORCE_PIECES)
};

// Launches the kernel for (verb, style, doLeader) on `stream`; returns hipSuccess or the
// launch error. *kernelName receives a static string naming the kernel chosen.
hipError_t launchBatch(const DevDfa &dfa, const Batch &b, int verb, int style, int doLeader,
                       const LaunchCfg &cfg, hipStream_t stream, const char **kernelName);

// The same for nb batches in the order given, as nb calls of launchBatch on `stream` would run
// them - except that runs of batches the streaming kernel takes (fixed stride, whole 64-byte
// blocks, styles Last / Full of check / match) with one stride go out as ONE launch each
// (k_stream_multi.h: table staged once, tiles handed out across batch boundaries).
hipError_t launchBatches(const DevDfa &dfa, const Batch *bs, uint32_t nb, int verb, int style,
                         int doLeader, const LaunchCfg &cfg, hipStream_t stream,
                         const char **kernelName);
hipError_t launchStreamBatches(const DevDfa &dfa, const Batch *bs, uint32_t nb, int verb, int style,
                               int doLeader, const LaunchCfg &cfg, hipStream_t stream,
                               const char **kernelName, uint32_t *taken);

// Red::collect per line (lib/Red.cpp:103-116): up to `cap` records per line at
// [line*cap, line*cap+cap) of result/start/end, counts[line] = number found.
hipError_t launchCollect(const DevDfa &dfa, const Batch &b, uint64_t cap, uint64_t *counts,
                         const LaunchCfg &cfg, hipStream_t stream);

// matchAll per line (include/Matcher.h:711-766): same record layout as launchCollect.
hipError_t launchMatchAll(const DevDfa &dfa, const Batch &b, uint64_t cap, uint64_t *counts,
                          int doLeader, const LaunchCfg &cfg, hipStream_t stream);

// StatefulMatcher::advance over one chunk per line (include/Matcher.h:770-792):
// state[line] in/out (device state index, >= nStates means "fresh matcher"), b.result out.
hipError_t launchAdvance(const DevDfa &dfa, const Batch &b, uint32_t *state, const LaunchCfg &cfg,
                         hipStream_t stream, const char **kernelName);

// redgpu_dfa_tune's profiling pass: hist[device state] += 1 per byte the anchored walk of
// match<styLast,false> consumes over the sample (hist zeroed by the caller).
hipError_t launchVisits(const DevDfa &dfa, const Batch &b, uint32_t *hist, const LaunchCfg &cfg,
                        hipStream_t stream);

// replaceCore per line (include/Matcher.h:643-706): counts[n], outOffsets[n + 1] (exclusive scan
// of the rewritten lengths, [n] = total) always; with out != nullptr also the rewritten bytes of
// every line that fits below outCap.  repl is device memory.
hipError_t launchReplace(const DevDfa &dfa, const Batch &b, int style, int doLeader,
                         const uint8_t *repl, uint64_t replLen, uint64_t max, uint64_t *counts,
                         uint64_t *outOffsets, uint8_t *out, uint64_t outCap,
                         const LaunchCfg &cfg, hipStream_t stream);

// Line splitting (lib/Util.cpp:109-130's rule): offsets[0] = 0, offsets[k+1] = position after the
// k-th delimiter, for k < cap; *nLines = delimiters found.  counts: uint32[splitChunks(len)],
// bases: uint64[splitChunks(len)], masks: splitMaskBytes(len) bytes (16-byte aligned) - scratch,
// all device memory.
uint64_t splitChunks(uint64_t len);
uint64_t splitMaskBytes(uint64_t len);
hipError_t launchSplitLines(const uint8_t *data, uint64_t len, uint8_t delim, uint64_t *offsets,
                            uint64_t cap, uint64_t *nLines, uint32_t *counts, uint64_t *bases,
                            uint16_t *masks, hipStream_t stream);

// bench.py's read-bandwidth calibration: one streaming pass over `bytes` (16-byte aligned).
hipError_t launchDiagRead(const void *data, uint64_t bytes, uint32_t *sink, int numCUs,
                          hipStream_t stream);

// bench.py's read + write calibration: the streaming walk's memory side alone over nLines lines of
// 64 bytes - each lane requests its line as k_stream does and stores one Outcome-shaped record
// (int32 + 2 x uint64) per line; nLines is rounded down to a multiple of 1024.
// lineBytes = 64, or a multiple of 128: the long-line request pattern, reads only.
hipError_t launchDiagLines(const uint8_t *data, uint64_t nLines, uint32_t lineBytes, int32_t *res,
                           uint64_t *st, uint64_t *en, uint32_t *sink, int numCUs,
                           hipStream_t stream);

// bench.py's L2 gather calibration: `rounds` dependent 2-byte gathers per chain from a table of
// 2^20 uint16 (2 MiB, device memory, any contents), 2 chains per lane, 2048 lanes per CU, no input
// side; *lookups = gathers the launch performs.
hipError_t launchDiagL2(const uint16_t *table, uint32_t rounds, uint32_t *sink, int numCUs,
                        hipStream_t stream, uint64_t *lookups);

// k_stream4.hip: the fixed-stride hot path with 3-4 chains per lane (mode = k_stream.h's
// StreamMode value: Last+start+end 0, Last+end 1, Full+start 3, Full 4).
bool stream4Eligible(const DevDfa &dfa, const Batch &b, const LaunchCfg &cfg);
hipError_t launchStream4(int mode, const DevDfa &dfa, const Batch &b, const LaunchCfg &cfg,
                         hipStream_t stream);

// Device scratch for the launches that need some (host_stage.cpp): a pool per HOST THREAD keyed
// by (current device, stream) - the buffer carries state between the kernels of one call, so no
// other thread may be handed it - bounded, freed at thread exit / redgpu_thread_release(),
// entries of a library-owned stream dropped with the stream.  scratchEntries: all threads'.
hipError_t scratchFor(hipStream_t stream, size_t bytes, void **out);
// The same entry's CONTROL words: two slots of 8 x uint32 that only k_ragged's launch sequence
// touches (the scratch buffer itself is shared with every other launch family of the thread and
// stream).  *out = the slot of this call, zero on arrival; *next = the other one, which the
// call's first kernel zeroes for the call after it - so no launch sequence starts with a memset.
hipError_t scratchCtlFor(hipStream_t stream, uint32_t **out, uint32_t **next);
void scratchDrop(int device, hipStream_t stream);
void scratchReleaseThread();
size_t scratchEntries();

// bench.py's calibrations.  launchDiagLds: `rounds` x 64 dependent table lookups per chain, 4
// chains per lane, 512 lanes per CU, bytes from registers (no input traffic): the LDS gather
// roof of the one-lookup-per-byte walk; *lookups receives the number issued.
// launchWalked: *walked (device u64, zeroed by the caller) += the bytes match<styLast,lead>'s loop
// (include/Matcher.h:443-479) consumes per line - what an early-exit walk actually reads.
hipError_t launchDiagLds(const DevDfa &dfa, uint32_t rounds, uint32_t *sink, int numCUs,
                         hipStream_t stream, uint64_t *lookups);
hipError_t launchWalked(const DevDfa &dfa, const Batch &b, int doLeader, unsigned long long *walked,
                        const LaunchCfg &cfg, hipStream_t stream);

// launchBatch's / launchAdvance's kernel families, one translation unit each (kernels.hip is
// compiled per family, -DREDGPU_TU): *handled = false when the batch is not the family's.
hipError_t launchFixedFamily(const DevDfa &dfa, const Batch &b, int verb, int style, int doLeader,
                             const LaunchCfg &cfg, hipStream_t stream, const char **kernelName,
                             bool *handled);
hipError_t launchRaggedFamily(const DevDfa &dfa, const Batch &b, int verb, int style, int doLeader,
                              const LaunchCfg &cfg, hipStream_t stream, const char **kernelName,
                              bool *handled);
hipError_t launchAdvanceStream(const DevDfa &dfa, const Batch &b, uint32_t *state,
                               const LaunchCfg &cfg, hipStream_t stream, const char **kernelName,
                               bool *handled);

// True when the specialised fixed-stride kernels can run this DFA at all.
bool fastPathEligible(const DevDfa &dfa);

} // namespace redgpu
