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
// Map of the kernel families, one header each, all included below inside the namespace
// (DESIGN.md section 4 has the measurements):
//   k_common.h        table accessors, table staging, lane context, the byte readers
//   k_lanes.h         the lane functions: direct restatements of the reference's cores
//   k_generic.h       k_generic<KIND, THREADS, VERB>: any verb / style / table, one line per lane
//   k_early.h         k_early: match / check over early-death DFAs (probe, park, drain)
//   k_scan_marked.h   k_scan_marked: scan / search in two passes (mark candidates, visit them)
//   k_fixed.h         k_fixed: strides that are not whole blocks, early exits on small batches
//   k_stream.h        k_stream<MODE, HALVES, THREADS, TABK>: the hot path - fixed-stride lines of
//                     whole 64-byte blocks, styles Last / Full, inline-asm byte step over a fused u8
//                     table, a big DFA's hot rows (sink + re-walk) or a class table in LDS
//   k_stream_lean.h   the same step with its bookkeeping deferred (opt-in, long lines)
//   k_stream_multi.h  k_stream_multi: several batches in one launch (redgpu_*_batches_dev)
//   k_ragged.h        k_ragged<MODE, TABK>: the walk over ragged lines, lanes refilled from a
//                     workgroup cursor; the tail pad; k_generic's bucketing pre-pass
//   k_ragged_long.h   k_ragged_outliers (the long lines of a batch: listed, walked first, huge ones as
//                     pieces), k_ragged_pieces_fold (the pieces' records -> their lines' Outcomes)
//   k_lists.h         k_collect, k_matchall, k_matchall_blocks (record lists per line)
//   k_style_blocks.h  k_style_blocks: early-exit styles and odd strides over the block walk
//   k_misc.h          k_advance, k_replace (+ scan), k_visits, k_walked
//   k_split.h         line splitting on the device
//   k_diag.h          bench.py's calibration kernels
//   launchers.h       grid / LDS / instantiation per family; includes k_chunk.h (speculative
//                     chunking of few long lines)
//   here              the dispatch: launchBatch / launchBatches and friends - which kernel runs what.
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

#include "k_common.h"
#include "k_lanes.h"
#include "k_generic.h"
#include "k_early.h"
#include "k_scan_marked.h"
#include "k_fixed.h"
#include "k_stream.h"
#include "k_stream_lean.h"
#include "k_stream_multi.h"
#include "k_ragged.h"
#include "k_lists.h"
#include "k_style_blocks.h"
#include "k_misc.h"
#include "k_split.h"
#include "k_diag.h"
#include "launchers.h"

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
  // ... and check<styLast / styFull, true> when the leader is the table's own forced chain
  // (dfa_image.h: leaderForced): the plain walk from the initial state, deaf to whatever accepts
  // up to the leader's end (a line that IS the leader would report the post-leader state's
  // result: left to k_generic)
  const bool leadCheckOk = lead && verb == kCheck && d.leaderForced && b.stride > d.leaderLen &&
                           (style == kStyLast || style == kStyFull) && !cfg.forceGeneric && !dying &&
                           fastPathEligible(d) && !b.offsets && b.stride % 64 == 0 &&
                           b.stride < (1ull << 31) && (reinterpret_cast<uintptr_t>(b.data) % 16) == 0;
  // Streaming kernel: styles Last / Full on lines that are whole 64-byte blocks.
  const bool streamOk = (fixedOk || leadCheckOk) && (style == kStyLast || style == kStyFull) &&
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
  const bool leadCheck = lead && verb == kCheck;  // (only ever true here with leadCheckOk)
  if ((streamOk || hotStreamOk || clsStreamOk) && !cfg.noChunking && !leadCheck &&
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
    if (leadCheck) sb.ignoreAcceptUpTo = d.leaderLen;
    static const int labLean = multiLabInt("REDGPU_LEAN", 0, 1, 0);
    static const int labSingle = multiLabInt("REDGPU_MULTI_SINGLE", 0, 1, 0);
    if (cfg.forceLean || labLean || labSingle) {
      // opt-in: k_stream_multi (its deferred-bookkeeping form for long lines) for a batch on its own
      uint32_t taken = 0;
      e = launchStreamBatches(d, &b, 1, verb, style, doLeader, cfg, stream, kernelName, &taken);
      if (e != hipSuccess || taken) return e;
    }
    if (!leadCheck && stream4Eligible(d, sb, cfg)) {
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
    if (lead && !leadCheck) {
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
  m.ignoreAcceptUpTo = 0;
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
  for (uint32_t k = m.nb; k < uint32_t(kMultiMax); ++k) m.tileStart[k + 1] = m.tileStart[m.nb];
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

uint64_t splitMaskBytes(uint64_t len) { return splitChunks(len) * (kSplitChunk / 8); }

hipError_t launchSplitLines(const uint8_t *data, uint64_t len, uint8_t delim, uint64_t *offsets,
                            uint64_t cap, uint64_t *nLines, uint32_t *counts, uint64_t *bases,
                            uint16_t *masks, hipStream_t stream) {
  const uint64_t nChunks = splitChunks(len);
  const uint32_t groups = uint32_t((nChunks + kSplitGroup - 1) / kSplitGroup);
  if (nChunks)
    hipLaunchKernelGGL(k_split_count, dim3(groups), dim3(kSplitThreads), 0, stream, data, len,
                       uint32_t(delim), nChunks, counts, masks);
  hipLaunchKernelGGL(k_split_scan, dim3(1), dim3(1024), 0, stream, counts, nChunks, bases, nLines,
                     offsets, cap);
  if (nChunks)
    hipLaunchKernelGGL(k_split_scatter, dim3(groups), dim3(kSplitThreads), 0, stream, masks,
                       nChunks, bases, offsets, cap);
  return hipGetLastError();
}

hipError_t launchDiagRead(const void *data, uint64_t bytes, uint32_t *sink, int numCUs,
                          hipStream_t stream) {
  if (bytes < 16) return hipSuccess;
  hipLaunchKernelGGL(k_diag_read, dim3(uint32_t(numCUs)), dim3(512), 0, stream,
                     static_cast<const uint4 *>(data), bytes / 16, sink);
  return hipGetLastError();
}

hipError_t launchDiagL2(const uint16_t *table, uint32_t rounds, uint32_t *sink, int numCUs,
                        hipStream_t stream, uint64_t *lookups) {
  constexpr int kCh = 2;
  hipLaunchKernelGGL((k_diag_l2<kCh>), dim3(uint32_t(numCUs) * 8), dim3(256), 0, stream, table, rounds,
                     sink);
  if (lookups) *lookups = uint64_t(numCUs) * 8 * 256 * kCh * rounds;
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
  if (b.nDev) {
    // the line count is on the device: only the k_ragged family reads it there
    bool handled = false;
    hipError_t fe = launchRaggedFamily(d, b, verb, style, doLeader, cfg, stream, kernelName, &handled);
    return handled || fe != hipSuccess ? fe : hipErrorNotSupported;
  }
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
