// k_stream_multi.h - several batches in ONE launch of the streaming walk (included by
// kernels.hip behind k_stream.h, inside its namespace).
//
// The reference's callers run a plain loop over their inputs (tools/bench.cpp:60-71,
// tools/thr_red.cpp:36-47); a caller that holds K buffers of lines hands all K to
// redgpu_match_batches_dev.  One launch of k_stream pays ~6.4 us that do not depend on the
// batch - the 64 KB table staged by every workgroup, the first blocks of all 256 CUs arriving
// together, the last lines' 64 dependent steps, the gap to the next launch - against 18 us of
// walk for 2^20 lines of 64 B.  Here ONE grid stages the table once and its
// workgroups take 1024-line tiles from the CONCATENATION of the batches' tile ranges: batch
// k + 1's first blocks stream in under batch k's last ones, and head and tail are paid once per
// K batches.  The walk itself is k_stream's (streamWalk16 over the fused u8 table, 2 lines per
// lane, ping-pong register blocks, results stored after every block without a branch); only the
// two cursors - which tile is being requested, which is being walked - know about batches:
// tile t belongs to batch k iff tileStart[k] <= t < tileStart[k + 1], and a cursor moves to the
// next batch by comparing against the bound it keeps in an SGPR (the descriptors are read from
// the kernel argument segment, scalar loads, once per batch and workgroup).
#pragma once

constexpr int kMultiMax = 32;  // batches per launch (descriptors ride in the kernel arguments)

struct MultiPtrs {
  const uint8_t *data;
  int32_t *result;
  uint64_t *start;  // may be nullptr (then every batch's is)
  uint64_t *end;    // may be nullptr
  uint64_t n;       // lines, > 0
};

struct MultiIo {
  uint32_t nb;       // batches, 1..kMultiMax
  uint32_t lineLen;  // bytes per line, a multiple of 64, the same for every batch
  uint32_t tileStart[kMultiMax + 1];  // [k] = first tile of batch k; [nb] = tiles in all
  uint32_t ignoreAcceptUpTo;          // Batch::ignoreAcceptUpTo, the same for every batch
  MultiPtrs b[kMultiMax];
};

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ntLoad16(const uint8_t *p) {
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}

// NT bit 1: result stores are non-temporal (the product path - the Outcomes are written once and
// not read by this launch; with two workgroups per CU the 20-batch configs[1] launch took 319 us
// against 345-349 us for either measure alone and for neither, r3 lab); bit 0: non-temporal
// input loads too (lab only: slower, 465 against 421 us).  REDGPU_MULTI_NT picks another value.
// LEAN: the deferred-bookkeeping step of k_stream_lean.h (long lines; the two remembered pieces of
// a line are re-walked exactly when the line ends).
// GR: groups of two chains per lane (2 = four lines per lane; LEAN only, k_stream_lean.h).
template <int MODE, int HALVES, int THREADS, int NT = 2, bool LEAN = false, int GR = 1>
__global__ void __launch_bounds__(THREADS)
k_stream_multi(DevDfa d, MultiIo m) {
  static_assert(MODE == kSmLastStartEnd || MODE == kSmLastEnd || MODE == kSmFullStart ||
                MODE == kSmFull, "the plain output modes only");
  static_assert(!LEAN || MODE != kSmFull, "styFull without start has no bookkeeping to defer");
  static_assert(GR == 1 || LEAN, "more than two chains per lane: the lean step only");
  constexpr uint32_t BLK = 64 * HALVES;
  constexpr int CH = kStreamChains * GR;
  constexpr bool kAcc = MODE == kSmLastStartEnd || MODE == kSmLastEnd;
  constexpr bool kStart = MODE == kSmLastStartEnd || MODE == kSmFullStart;
  constexpr uint32_t LPT = uint32_t(THREADS) * CH;  // lines per tile
  __shared__ __align__(16) uint8_t lds[kStreamTabBytes + 1024];  // table at LDS offset 0
  uint8_t *tab = lds;
  int32_t *ldsRes = reinterpret_cast<int32_t *>(lds + kStreamTabBytes);

  const uint32_t init = d.init, firstAccept = d.firstAccept;
  const uint32_t lineLen = m.lineLen;
  const uint32_t R = lineLen / BLK;  // blocks per line
  const uint32_t nTiles = m.tileStart[m.nb];
  const uint32_t G = gridDim.x;
  if (blockIdx.x >= nTiles) return;
  const uint32_t myTiles = (nTiles - blockIdx.x + G - 1) / G;
  const uint64_t Q = uint64_t(myTiles) * R;  // blocks this workgroup walks

  // ---- load cursor: one block ahead of the walk; once on the last block it stays there, so
  // requests past the end of the work re-read that block (unconditional requests keep the
  // compiler's in-order vmcnt counts exact, k_stream.h)
  // (the batch's descriptor is held in registers and re-read - scalar loads from the kernel
  // argument segment - only when the cursor crosses into the next batch: read on every tile,
  // each tile's first requests waited ~0.2 us for them)
  uint32_t ldTile = blockIdx.x, ldR = 0, ldK = 0;
  uint32_t ldLo = 0, ldHi = m.tileStart[1];
  const uint8_t *ldData = m.b[0].data;
  uint64_t ldN = m.b[0].n;
  uint64_t ldQ = 0;
  const uint8_t *ldP[CH];
  auto setLoadTile = [&]() {
    while (ldTile >= ldHi) {
      ++ldK;
      ldLo = ldHi;
      ldHi = m.tileStart[ldK + 1];
      ldData = m.b[ldK].data;
      ldN = m.b[ldK].n;
    }
    const uint64_t first = uint64_t(ldTile - ldLo) * LPT;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      uint64_t ln = first + uint64_t(c) * THREADS + threadIdx.x;
      if (ln >= ldN) ln = ldN - 1;  // surplus lanes re-walk the batch's last line
      ldP[c] = ldData + ln * lineLen;
    }
  };
  auto issue = [&](BlockRegs<HALVES> (&blk)[CH]) {
    const uint32_t byteOff = ldR * BLK;
#pragma unroll
    for (int k = 0; k < 4 * HALVES; ++k) {
#pragma unroll
      for (int c = 0; c < CH; ++c)
        blk[c].p[k] = (NT & 1)
                          ? ntLoad16(ldP[c] + byteOff + 16 * k)
                          : reinterpret_cast<const uint4 *>(ldP[c] + byteOff)[k];
    }
    if (ldQ + 1 < Q) {
      ++ldQ;
      if (++ldR == R) {
        ldR = 0;
        ldTile += G;
        setLoadTile();
      }
    }
  };

  // table requests first, the first input block's right behind them, the LDS stores and the
  // barrier after (k_stream.h)
  const uint4 *tsrc = reinterpret_cast<const uint4 *>(d.table);
  const uint32_t n16 = d.tableBytes / 16;
  constexpr uint32_t kStagePieces = (kStreamTabBytes / 16 + THREADS - 1) / THREADS;
  uint4 tv[kStagePieces];
#pragma unroll
  for (uint32_t k = 0; k < kStagePieces; ++k) {
    const uint32_t i = k * THREADS + threadIdx.x;
    tv[k] = i < n16 ? tsrc[i] : make_uint4(0, 0, 0, 0);
  }
  const int32_t myRes = threadIdx.x < d.nStates ? d.result[threadIdx.x] : 0;
  BlockRegs<HALVES> A[CH], B[CH];
  setLoadTile();
  issue(A);
  {
    uint4 *dst = reinterpret_cast<uint4 *>(tab);
#pragma unroll
    for (uint32_t k = 0; k < kStagePieces; ++k) {
      const uint32_t i = k * THREADS + threadIdx.x;
      if (i < kStreamTabBytes / 16) dst[i] = tv[k];
    }
    if (threadIdx.x < 256) ldsRes[threadIdx.x] = myRes;
  }
  asm volatile("" : : "v"(tab) : "memory");  // the byte steps read the table from asm only
  __syncthreads();

  // ---- walk cursor
  uint32_t tile = blockIdx.x, r = 0, wkK = 0;
  uint32_t wkLo = 0, wkHi = m.tileStart[1];
  int32_t *oRes = m.b[0].result;
  uint64_t *oStart = m.b[0].start, *oEnd = m.b[0].end;
  uint64_t oN = m.b[0].n;
  const uint8_t *oData = m.b[0].data;  // LEAN: the pieces re-read at a line's end
  auto setWalkTile = [&]() {
    while (tile >= wkHi) {
      ++wkK;
      wkLo = wkHi;
      wkHi = m.tileStart[wkK + 1];
      oRes = m.b[wkK].result;
      oStart = m.b[wkK].start;
      oEnd = m.b[wkK].end;
      oN = m.b[wkK].n;
      if (LEAN) oData = m.b[wkK].data;
    }
  };
  setWalkTile();

  uint32_t s[CH], accS[CH], endv[CH], startv[CH];
  uint64_t mA[CH], mB[CH];
  LeanRegs L[GR];
  const uint32_t T8 = firstAccept << 8;

  auto walkLean = [&](const BlockRegs<HALVES> (&blk)[CH]) {
    if (r == 0) {
#pragma unroll
      for (int g = 0; g < GR; ++g) leanBegin(L[g], init);
    }
#pragma unroll
    for (int g = 0; g < GR; ++g) leanPin(L[g]);
    if (GR > 1) leanDrain();  // counted waits below: nothing else may be in flight on lgkmcnt
#pragma unroll
    for (int h = 0; h < HALVES; ++h) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint4 piece[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) piece[c] = blk[c].p[4 * h + q];
        leanWalk16<kAcc, kStart, GR>(piece, L, T8, init);
      }
    }
    if (GR > 1) leanDrain();  // every state register has landed before C++ code sees it
    if (r + 1 == R) {
      // the line ends: close the last piece's windows, then read the two remembered pieces again
      // and walk them from their entry states with the exact step (k_stream.h) - 32 steps per
      // line.  Loads and stores under this branch: the next block's wait sits them out once per
      // LINE, which for the lines this form takes is once per >= 8 blocks.
      const uint32_t lane = threadIdx.x & 63;
      const uint64_t first = uint64_t(tile - wkLo) * LPT;
#pragma unroll
      for (int g = 0; g < GR; ++g) {
        LeanRegs &G = L[g];
        leanEnd(G, T8, init, kAcc, kStart);
        uint64_t ln[2];
        uint4 pa[2], ps[2];
        uint32_t PA[2], PS[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          ln[c] = first + uint64_t(2 * g + c) * THREADS + threadIdx.x;
          const uint64_t lc = ln[c] < oN ? ln[c] : oN - 1;
          const uint8_t *base = oData + lc * lineLen;
          PA[c] = G.recA[c] ? (G.recA[c] >> 8) - 1 : 0;
          PS[c] = G.recS[c] ? (G.recS[c] >> 8) - 1 : 0;
          if (kAcc) pa[c] = *reinterpret_cast<const uint4 *>(base + 16ull * PA[c]);
          if (kStart) ps[c] = *reinterpret_cast<const uint4 *>(base + 16ull * PS[c]);
        }
        int32_t rr[2];
        uint32_t en[2], sv[2];
        if (kAcc) {
          uint32_t s2[2];
          StreamBook b2[2];
          uint64_t u0[2] = {0, 0}, u1[2] = {0, 0};
#pragma unroll
          for (int c = 0; c < 2; ++c) { s2[c] = G.recA[c] & 0xffu; b2[c].acc = 0; b2[c].end = 0; b2[c].start = 0; }
          streamWalk16<kSmLastEnd, 0>(pa, s2, b2, u0, u1, firstAccept, init);
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            uint32_t acc = b2[c].acc, er = b2[c].end;
            if (s2[c] >= firstAccept) { acc = s2[c]; er = 16; }
            const bool hit = G.recA[c] != 0;
            rr[c] = hit ? ldsRes[acc & 0xffu] : 0;
            en[c] = hit ? 16u * PA[c] + er : 0u;
          }
        } else {
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            rr[c] = G.sX[c] >= firstAccept ? ldsRes[G.sX[c] & 0xffu] : 0;
            en[c] = lineLen;
          }
        }
        if (kStart) {
          uint32_t s3[2];
          StreamBook b3[2];
          uint64_t w0[2], w1[2];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            s3[c] = G.recS[c] & 0xffu;
            b3[c].acc = 0; b3[c].end = 0; b3[c].start = 0;
            w0[c] = __builtin_amdgcn_ballot_w64(s3[c] == init);  // no event "before" the piece
            w1[c] = w0[c];
          }
          streamWalk16<kSmFullStart, 0>(ps, s3, b3, w0, w1, firstAccept, init);
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            uint32_t pos = b3[c].start ? b3[c].start - 1 : 0;
            if (((w0[c] >> lane) & 1) && s3[c] != init) pos = 15;
            sv[c] = G.recS[c] ? 16u * PS[c] + pos : 0u;
          }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          if (ln[c] < oN) {
            __builtin_nontemporal_store(rr[c], oRes + ln[c]);
            if (oEnd) __builtin_nontemporal_store(rr[c] ? uint64_t(en[c]) : uint64_t(0), oEnd + ln[c]);
            if (kStart && oStart)
              __builtin_nontemporal_store(rr[c] ? uint64_t(sv[c]) : uint64_t(0), oStart + ln[c]);
          }
        }
      }
    }
    if (++r == R) {
      r = 0;
      tile += G;
      if (tile < nTiles) setWalkTile();
    }
  };

  auto walkBlock = [&](const BlockRegs<HALVES> (&blk)[CH]) {
    if constexpr (LEAN) {
      walkLean(blk);
    } else {
    if (r == 0) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        s[c] = init; accS[c] = 0; endv[c] = 0; startv[c] = 0;
        mA[c] = ~0ull; mB[c] = ~0ull;
      }
    }
#pragma unroll
    for (int h = 0; h < HALVES; ++h) {
      StreamBook b[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) { b[c].acc = accS[c]; b[c].end = 0; b[c].start = 0; }
      uint4 piece[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) piece[c] = blk[c].p[4 * h + 0];
      streamWalk16<MODE, 0>(piece, s, b, mA, mB, firstAccept, init);
#pragma unroll
      for (int c = 0; c < CH; ++c) piece[c] = blk[c].p[4 * h + 1];
      streamWalk16<MODE, 1>(piece, s, b, mA, mB, firstAccept, init);
#pragma unroll
      for (int c = 0; c < CH; ++c) piece[c] = blk[c].p[4 * h + 2];
      streamWalk16<MODE, 2>(piece, s, b, mA, mB, firstAccept, init);
#pragma unroll
      for (int c = 0; c < CH; ++c) piece[c] = blk[c].p[4 * h + 3];
      streamWalk16<MODE, 3>(piece, s, b, mA, mB, firstAccept, init);
      const uint32_t off = r * BLK + h * 64;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (kAcc) {
          accS[c] = b[c].acc;
          endv[c] = b[c].end ? off + b[c].end : endv[c];
          if (s[c] >= firstAccept) { accS[c] = s[c]; endv[c] = off + 64; }
          if (endv[c] <= m.ignoreAcceptUpTo) { endv[c] = 0; accS[c] = 0; }  // k_stream.h
        }
        if (kStart) {
          startv[c] = b[c].start ? off + b[c].start - 1 : startv[c];
          const bool wasInit63 = (mA[c] >> (threadIdx.x & 63)) & 1;
          if (wasInit63 && s[c] != init) startv[c] = off + 63;
        }
      }
    }
    // results after EVERY block, without a branch: lanes whose line ends here into the line's
    // slots, everyone else into the DFA's sink (k_stream.h: stores under a branch are vm
    // operations the compiler cannot count)
    const bool lineEnd = r + 1 == R;
    const uint64_t first = uint64_t(tile - wkLo) * LPT;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const uint64_t ln = first + uint64_t(c) * THREADS + threadIdx.x;
      const bool report = lineEnd && ln < oN;
      int32_t rr;
      uint32_t en;
      if (kAcc) {
        rr = ldsRes[accS[c] & 0xffu];
        rr = endv[c] ? rr : 0;
        en = endv[c];
      } else {
        rr = ldsRes[s[c] & 0xffu];
        rr = s[c] >= firstAccept ? rr : 0;
        en = lineLen;
      }
      int32_t *pr = report ? oRes + ln : reinterpret_cast<int32_t *>(d.sink);
      uint64_t *pe = report && oEnd ? oEnd + ln : reinterpret_cast<uint64_t *>(d.sink);
      uint64_t *ps = report && oStart ? oStart + ln : reinterpret_cast<uint64_t *>(d.sink);
      if ((NT & 2) && R == 1) {  // (longer lines: the sink must stay a cached line, k_stream.h)
        __builtin_nontemporal_store(rr, pr);
        __builtin_nontemporal_store(rr ? uint64_t(en) : uint64_t(0), pe);
        if (kStart) __builtin_nontemporal_store(rr ? uint64_t(startv[c]) : uint64_t(0), ps);
      } else {
        *pr = rr;
        *pe = rr ? uint64_t(en) : 0;
        if (kStart) *ps = rr ? uint64_t(startv[c]) : 0;
      }
    }
    if (++r == R) {
      r = 0;
      tile += G;
      if (tile < nTiles) setWalkTile();
    }
    }  // !LEAN
  };

  // rotated by one block, as k_stream: both ways into the loop head end in "block requested, a
  // block's results stored, block requested"
  issue(B);
  walkBlock(A);
  issue(A);
  for (uint64_t q = 1; q < Q; q += 2) {
    walkBlock(B);
    if (q + 1 >= Q) break;
    issue(B);
    walkBlock(A);
    issue(A);
  }
}

// tuning lab: REDGPU_MULTI_NT (see k_stream_multi; match<styLast> with start only),
// REDGPU_MULTI_WGS = workgroups per CU for 64-byte blocks (default 2: 16 waves per CU over two
// table copies - a single 64 MiB launch loses to it, 26.0 against 24.7 us, because both copies are
// staged before the first step; over many batches that head is paid once; the 128-byte form
// needs 161 VGPRs and stays at one), REDGPU_MULTI_SINGLE=1 sends single batches here too.
inline int multiLabInt(const char *name, int lo, int hi, int dflt) {
  const char *e = getenv(name);
  const int v = e ? atoi(e) : dflt;
  return v < lo || v > hi ? dflt : v;
}

template <int MODE, int NT, bool LEAN>
hipError_t launchStreamMultiN(const DevDfa &d, const MultiIo &m, const LaunchCfg &cfg,
                              hipStream_t stream) {
  static const int wgs = multiLabInt("REDGPU_MULTI_WGS", 1, 2, 2);
  const uint32_t tiles = m.tileStart[m.nb];
  const bool wide = m.lineLen % 128 == 0;
  const uint32_t want = uint32_t(cfg.numCUs) * uint32_t(wide ? 1 : wgs);
  const uint32_t blocks = tiles < want ? tiles : want;
  if (wide)
    hipLaunchKernelGGL((k_stream_multi<MODE, 2, kStreamThreads, NT, LEAN>), dim3(blocks),
                       dim3(kStreamThreads), 0, stream, d, m);
  else
    hipLaunchKernelGGL((k_stream_multi<MODE, 1, kStreamThreads, NT, LEAN>), dim3(blocks),
                       dim3(kStreamThreads), 0, stream, d, m);
  return hipGetLastError();
}

// the lean step with FOUR lines per lane (two chain groups), 64-byte blocks, one workgroup per
// CU: tiles of 2048 lines (the caller's MultiIo counts 1024-line tiles)
template <int MODE>
hipError_t launchStreamLean4(const DevDfa &d, const MultiIo &m, const LaunchCfg &cfg,
                             hipStream_t stream) {
  MultiIo m4 = m;
  const uint64_t lpt = uint64_t(kStreamThreads) * kStreamChains * 2;
  for (uint32_t k = 0; k < m.nb; ++k)
    m4.tileStart[k + 1] = m4.tileStart[k] + uint32_t((m.b[k].n + lpt - 1) / lpt);
  for (uint32_t k = m.nb; k < uint32_t(kMultiMax); ++k) m4.tileStart[k + 1] = m4.tileStart[m.nb];
  const uint32_t tiles = m4.tileStart[m4.nb];
  const uint32_t blocks = tiles < uint32_t(cfg.numCUs) ? tiles : uint32_t(cfg.numCUs);
  hipLaunchKernelGGL((k_stream_multi<MODE, 1, kStreamThreads, 2, true, 2>), dim3(blocks),
                     dim3(kStreamThreads), 0, stream, d, m4);
  return hipGetLastError();
}

// REDGPU_F_FORCE_LEAN (or the lab's REDGPU_LEAN=1): lines from kLeanMinLine bytes up take the
// deferred-bookkeeping step
inline bool leanWanted(const MultiIo &m, const LaunchCfg &cfg) {
  static const int lab = multiLabInt("REDGPU_LEAN", 0, 1, 0);
  return (lab || cfg.forceLean) && m.lineLen >= kLeanMinLine && m.lineLen < (1u << 27);
}

template <int MODE>
hipError_t launchStreamMultiT(const DevDfa &d, const MultiIo &m, const LaunchCfg &cfg,
                              hipStream_t stream) {
  if constexpr (MODE != kSmFull) {
    if (leanWanted(m, cfg)) {
      if (cfg.leanChains4) return launchStreamLean4<MODE>(d, m, cfg, stream);
      return launchStreamMultiN<MODE, 2, true>(d, m, cfg, stream);
    }
  }
  if constexpr (MODE == kSmLastStartEnd) {
    static const int nt = multiLabInt("REDGPU_MULTI_NT", 0, 3, 2);
    if (nt == 0) return launchStreamMultiN<MODE, 0, false>(d, m, cfg, stream);
    if (nt == 1) return launchStreamMultiN<MODE, 1, false>(d, m, cfg, stream);
    if (nt == 3) return launchStreamMultiN<MODE, 3, false>(d, m, cfg, stream);
  }
  return launchStreamMultiN<MODE, 2, false>(d, m, cfg, stream);
}
