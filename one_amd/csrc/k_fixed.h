// k_fixed.h - k_fixed<STYLE, POS, START, CHAINS>: fixed strides that are multiples of 16 but not of 64 and the
// early-exit styles on small batches; k_leader_filter
// (included by kernels.hip inside namespace redgpu { namespace { ... } }; see its file map).
#pragma once

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
