// k_generic.h - k_generic<KIND, THREADS, VERB>: any verb, style, doLeader, ragged or fixed lines, any table
// placement - one line per lane through the lane functions of k_lanes.h
// (included by kernels.hip inside namespace redgpu { namespace { ... } }; see its file map).
#pragma once

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
