// launchers.h - launch helpers: which grid, how much LDS, which instantiation - one template per kernel
// family, instantiated by the translation unit that calls it (kernels.hip, -DREDGPU_TU)
// (included by kernels.hip inside namespace redgpu { namespace { ... } }; see its file map).
#pragma once

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
