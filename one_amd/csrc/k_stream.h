// k_stream.h - the streaming fixed-stride kernel (included by kernels.hip, inside its namespace).
//
// Hot path of BASELINE.json configs[1]/[2]: lines of a fixed length that is a multiple of 64
// bytes, DFA with at most 256 reachable states (fused [state][byte] u8 table, 64 KB, resident
// at LDS offset 0), styles Last and Full of check / match (include/Matcher.h:363-495).
//
// Shape, and why (every number below was measured on MI355X, tools/tune.hip):
//  * One workgroup of 512 threads per CU shares one table copy (8 waves x 2 lines = 16 dependent
//    chains per CU; 1024 threads saturate the LDS better in steady state but need twice the
//    data before their first step - on 64 MiB batches 512 wins); staging it costs ~1.6 us and is
//    done BEFORE any input is requested - input requests issued earlier sit in front of the
//    table's in the CU's memory queues and delay the table barrier to ~6 us.
//  * Each lane walks 2 lines at once (two dependent chains per lane).
//    Per line a ring of four 16-byte pieces lives in VGPRs; while piece p is walked, piece p+3
//    is requested into the slot piece p-1 vacated.  All requests are unconditional so the
//    compiler's in-order vmcnt counts are exact (a conditional request made it fall back to
//    vmcnt(0), which serialised every fourth piece behind a full HBM round trip).
//  * The byte step is inline asm: first the dependent chain for both lines
//    (v_perm_b32 forms (state << 8) | byte, ds_read_u8 fetches the next state), then the
//    previous state's bookkeeping under the LDS latency, then one s_waitcnt lgkmcnt(0).
//    6 VALU per byte per line for the full Outcome (2 v_cmp, 3 v_cndmask, 1 v_perm), no
//    zero-extension, no s_nop.  The asm clobbers SCC (s_andn2_b64) and says so.
//  * Bound: one ds_read_u8 per input byte.  On uniformly random bytes a 64-lane gather of the
//    64 KB table costs ~7 LDS cycles (2 + ~5 bank-conflict cycles, SQ_LDS_BANK_CONFLICT), i.e.
//    ~9 bytes/clk/CU: the LDS gather rate, not HBM, is this kernel's roof.
#pragma once

constexpr int kStreamThreads = 512;
constexpr int kStreamChains = 2;
constexpr uint32_t kStreamTabBytes = 65536;

enum StreamMode : int {
  kSmLastStartEnd = 0,  // match<styLast>  result + start + end
  kSmLastEnd = 1,       // match<styLast>  result + end; also check<styLast> (end not stored)
  kSmFullStart = 3,     // match<styFull>  result + start (+ end = line length)
  kSmFull = 4,          // check<styFull> / match<styFull> without start
};

struct StreamBook {
  uint32_t acc;    // last accepting state seen
  uint32_t end;    // block-relative position (1..63) of that accept, 0 = none in this block
  uint32_t start;  // block-relative position of the last "left the initial state", 0 = none
};

#define RS_PERM(c) "v_perm_b32 %[a" #c "], %[s" #c "], %[w" #c "], %[sel]\n\t"
#define RS_READ(c) "ds_read_u8 %[t" #c "], %[a" #c "]\n\t"
#define RS_CMPA(c) "v_cmp_le_u32_e64 %[m" #c "], %[T], %[s" #c "]\n\t"
#define RS_CMPI(c) "v_cmp_eq_u32_e64 %[i" #c "], %[init], %[s" #c "]\n\t"
#define RS_ACC(c) "v_cndmask_b32_e64 %[acc" #c "], %[acc" #c "], %[s" #c "], %[m" #c "]\n\t"
#define RS_END(c) "v_cndmask_b32_e64 %[e" #c "], %[e" #c "], %[idx], %[m" #c "]\n\t"
#define RS_LEAVE(c) "s_andn2_b64 %[l" #c "], %[was" #c "], %[i" #c "]\n\t"
#define RS_START(c) "v_cndmask_b32_e64 %[st" #c "], %[st" #c "], %[idx], %[l" #c "]\n\t"
#define RS_WAIT "s_waitcnt lgkmcnt(0)"

#define RS_O_CHAIN(c) [a##c] "=&v"(a[c]), [t##c] "=&v"(t[c])
#define RS_O_ACC(c) [m##c] "=&s"(m[c]), [acc##c] "+v"(b[c].acc)
#define RS_O_END(c) [e##c] "+v"(b[c].end)
#define RS_O_START(c) [i##c] "=&s"(isI[c]), [l##c] "=&s"(l[c]), [st##c] "+v"(b[c].start)
#define RS_I_CHAIN(c) [s##c] "v"(s[c]), [w##c] "v"(w[c])
#define RS_I_START(c) [was##c] "s"(wasI[c])

// One byte step for both chains.  IDX (0..63) = bytes of this 64-byte block already consumed =
// block-relative position of the state being book-kept.  `s` holds that state on entry and the
// next state on exit.  wasI / isI: lane masks "state == initial" before / after (ping-pong).
template <int MODE, int IDX>
__device__ __forceinline__ void streamStep(uint32_t (&s)[2], const uint32_t (&w)[2],
                                           StreamBook (&b)[2], const uint64_t (&wasI)[2],
                                           uint64_t (&isI)[2], uint32_t sel, uint32_t T,
                                           uint32_t init) {
  uint32_t a[2], t[2];
  uint64_t m[2], l[2];
  if constexpr (MODE == kSmLastStartEnd) {
    asm volatile(RS_PERM(0) RS_PERM(1) RS_READ(0) RS_READ(1)
                 RS_CMPA(0) RS_CMPA(1) RS_CMPI(0) RS_CMPI(1)
                 RS_ACC(0) RS_ACC(1) RS_END(0) RS_END(1)
                 RS_LEAVE(0) RS_LEAVE(1) RS_START(0) RS_START(1) RS_WAIT
                 : RS_O_CHAIN(0), RS_O_CHAIN(1), RS_O_ACC(0), RS_O_ACC(1), RS_O_END(0),
                   RS_O_END(1), RS_O_START(0), RS_O_START(1)
                 : RS_I_CHAIN(0), RS_I_CHAIN(1), RS_I_START(0), RS_I_START(1), [sel] "s"(sel),
                   [T] "s"(T), [init] "s"(init), [idx] "n"(IDX)
                 : "memory", "scc");
  } else if constexpr (MODE == kSmLastEnd) {
    asm volatile(RS_PERM(0) RS_PERM(1) RS_READ(0) RS_READ(1)
                 RS_CMPA(0) RS_CMPA(1) "s_nop 0\n\t" RS_ACC(0) RS_ACC(1) RS_END(0) RS_END(1) RS_WAIT
                 : RS_O_CHAIN(0), RS_O_CHAIN(1), RS_O_ACC(0), RS_O_ACC(1), RS_O_END(0), RS_O_END(1)
                 : RS_I_CHAIN(0), RS_I_CHAIN(1), [sel] "s"(sel), [T] "s"(T), [idx] "n"(IDX)
                 : "memory");
  } else if constexpr (MODE == kSmFullStart) {
    asm volatile(RS_PERM(0) RS_PERM(1) RS_READ(0) RS_READ(1)
                 RS_CMPI(0) RS_CMPI(1) "s_nop 0\n\t" RS_LEAVE(0) RS_LEAVE(1) RS_START(0) RS_START(1)
                 RS_WAIT
                 : RS_O_CHAIN(0), RS_O_CHAIN(1), RS_O_START(0), RS_O_START(1)
                 : RS_I_CHAIN(0), RS_I_CHAIN(1), RS_I_START(0), RS_I_START(1), [sel] "s"(sel),
                   [init] "s"(init), [idx] "n"(IDX)
                 : "memory", "scc");
  } else {
    asm volatile(RS_PERM(0) RS_PERM(1) RS_READ(0) RS_READ(1) RS_WAIT
                 : RS_O_CHAIN(0), RS_O_CHAIN(1)
                 : RS_I_CHAIN(0), RS_I_CHAIN(1), [sel] "s"(sel)
                 : "memory");
  }
  s[0] = t[0];
  s[1] = t[1];
}

// 16 bytes (one piece) of both chains; Q = which quarter of the 64-byte block this piece is.
template <int MODE, int Q>
__device__ __forceinline__ void streamWalk16(const uint4 (&piece)[2], uint32_t (&s)[2],
                                             StreamBook (&b)[2], uint64_t (&mA)[2],
                                             uint64_t (&mB)[2], uint32_t T, uint32_t init) {
  uint32_t w[2];
#define RS_WORD(K, FIELD)                                                          \
  w[0] = piece[0].FIELD;                                                           \
  w[1] = piece[1].FIELD;                                                           \
  streamStep<MODE, 16 * Q + 4 * K + 0>(s, w, b, mA, mB, 0x0c0c0400u, T, init);     \
  streamStep<MODE, 16 * Q + 4 * K + 1>(s, w, b, mB, mA, 0x0c0c0401u, T, init);     \
  streamStep<MODE, 16 * Q + 4 * K + 2>(s, w, b, mA, mB, 0x0c0c0402u, T, init);     \
  streamStep<MODE, 16 * Q + 4 * K + 3>(s, w, b, mB, mA, 0x0c0c0403u, T, init);
  RS_WORD(0, x) RS_WORD(1, y) RS_WORD(2, z) RS_WORD(3, w)
#undef RS_WORD
}

template <int MODE>
__global__ void __launch_bounds__(kStreamThreads)
k_stream(DevDfa d, Batch io) {
  constexpr int CH = kStreamChains;
  constexpr int THREADS = kStreamThreads;
  constexpr bool kAcc = MODE == kSmLastStartEnd || MODE == kSmLastEnd;
  constexpr bool kStart = MODE == kSmLastStartEnd || MODE == kSmFullStart;
  // the kernel's only LDS object: the table MUST sit at LDS offset 0 (asm addresses it so)
  __shared__ __align__(16) uint8_t lds[kStreamTabBytes + 1024];
  uint8_t *tab = lds;
  int32_t *ldsRes = reinterpret_cast<int32_t *>(lds + kStreamTabBytes);

  const uint32_t init = d.init, firstAccept = d.firstAccept;
  const uint32_t lineLen = uint32_t(io.stride);
  const uint32_t R = lineLen / 64;  // 64-byte blocks per line
  const uint64_t linesPerTile = uint64_t(THREADS) * CH;
  const uint64_t nTiles = (io.n + linesPerTile - 1) / linesPerTile;
  const uint64_t G = gridDim.x;
  if (blockIdx.x >= nTiles) return;
  const uint64_t myTiles = (nTiles - blockIdx.x + G - 1) / G;
  const uint64_t Q = myTiles * R;

  auto blockPtr = [&](uint64_t tile, uint32_t r, int c) {
    uint64_t ln = tile * linesPerTile + uint64_t(c) * THREADS + threadIdx.x;
    if (ln >= io.n) ln = io.n - 1;  // surplus lanes re-walk the last line; nothing is stored
    return io.data + ln * lineLen + r * 64;
  };
  auto ld = [](const uint8_t *p, int k) { return reinterpret_cast<const uint4 *>(p)[k]; };

  // ---- table first: 4 coalesced 16-byte pieces per thread (+ the result codes) ------------
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(d.table);
    const uint32_t n16 = d.tableBytes / 16;
    uint4 v[kStreamTabBytes / 16 / THREADS];
#pragma unroll
    for (uint32_t k = 0; k < kStreamTabBytes / 16 / THREADS; ++k) {
      const uint32_t i = k * THREADS + threadIdx.x;
      v[k] = i < n16 ? src[i] : make_uint4(0, 0, 0, 0);
    }
    const int32_t myRes = threadIdx.x < d.nStates ? d.result[threadIdx.x] : 0;
    uint4 *dst = reinterpret_cast<uint4 *>(tab);
#pragma unroll
    for (uint32_t k = 0; k < kStreamTabBytes / 16 / THREADS; ++k) dst[k * THREADS + threadIdx.x] = v[k];
    if (threadIdx.x < 256) ldsRes[threadIdx.x] = myRes;
  }
  __syncthreads();

  uint4 slot[4][CH];
  const uint8_t *cur[CH], *nxt[CH];
  uint64_t tile = blockIdx.x;
  uint32_t r = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) cur[c] = blockPtr(tile, 0, c);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int c = 0; c < CH; ++c) slot[k][c] = ld(cur[c], k);
  }

  uint32_t s[CH], accS[CH], endv[CH], startv[CH];
  uint64_t mA[CH], mB[CH];
  for (uint64_t q = 0; q < Q; ++q) {
    // where each chain's NEXT 64-byte block lives: further along the same lines, or the next
    // tile's lines; past the end of the work it is this block again (harmless re-read)
    const bool haveNext = q + 1 < Q;
    uint32_t nr = r + 1;
    uint64_t ntile = tile;
    if (nr == R) { nr = 0; ntile = tile + G; }
#pragma unroll
    for (int c = 0; c < CH; ++c) nxt[c] = blockPtr(haveNext ? ntile : tile, haveNext ? nr : r, c);
    if (r == 0) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        s[c] = init; accS[c] = 0; endv[c] = 0; startv[c] = 0;
        mA[c] = ~0ull; mB[c] = ~0ull;
      }
    }
    StreamBook b[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { b[c].acc = accS[c]; b[c].end = 0; b[c].start = 0; }

#pragma unroll
    for (int c = 0; c < CH; ++c) slot[3][c] = ld(cur[c], 3);
    streamWalk16<MODE, 0>(slot[0], s, b, mA, mB, firstAccept, init);
#pragma unroll
    for (int c = 0; c < CH; ++c) slot[0][c] = ld(nxt[c], 0);
    streamWalk16<MODE, 1>(slot[1], s, b, mA, mB, firstAccept, init);
#pragma unroll
    for (int c = 0; c < CH; ++c) slot[1][c] = ld(nxt[c], 1);
    streamWalk16<MODE, 2>(slot[2], s, b, mA, mB, firstAccept, init);
#pragma unroll
    for (int c = 0; c < CH; ++c) slot[2][c] = ld(nxt[c], 2);
    streamWalk16<MODE, 3>(slot[3], s, b, mA, mB, firstAccept, init);

    // Fold the block-relative events into absolute positions.  A recorded relative index k
    // (1..63) is "the state after k bytes": matchEnd = off + k (Matcher.h:463), matchStart =
    // off + k - 1 (the byte that left the initial state, Matcher.h:446-451).  Index 0 is the
    // carried-in state, already accounted for at the end of the previous block.  The state
    // after this block's 64th byte is handled here, once per block, in plain code.
    const uint32_t off = r * 64;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (kAcc) {
        accS[c] = b[c].acc;
        endv[c] = b[c].end ? off + b[c].end : endv[c];
        if (s[c] >= firstAccept) { accS[c] = s[c]; endv[c] = off + 64; }
      }
      if (kStart) {
        startv[c] = b[c].start ? off + b[c].start - 1 : startv[c];
        const bool wasInit63 = (mA[c] >> (threadIdx.x & 63)) & 1;  // written by the last step
        if (wasInit63 && s[c] != init) startv[c] = off + 63;
      }
    }
    if (++r == R) {
      r = 0;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const uint64_t ln = tile * linesPerTile + uint64_t(c) * THREADS + threadIdx.x;
        if (ln < io.n) {
          int32_t rr;
          uint32_t en;
          if (kAcc) {
            // endv != 0 <=> an accepting state was REACHED (an accepting initial state alone
            // does not count for non-empty input, Matcher.h:443-468)
            rr = endv[c] ? ldsRes[accS[c]] : 0;
            en = endv[c];
          } else {
            rr = s[c] >= firstAccept ? ldsRes[s[c]] : 0;  // styFull: the final state's result
            en = lineLen;
          }
          io.result[ln] = rr;
          if (io.end) io.end[ln] = rr ? uint64_t(en) : 0;
          if (kStart && io.start) io.start[ln] = rr ? uint64_t(startv[c]) : 0;
        }
      }
      tile += G;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) cur[c] = nxt[c];
  }
}

#undef RS_PERM
#undef RS_READ
#undef RS_CMPA
#undef RS_CMPI
#undef RS_ACC
#undef RS_END
#undef RS_LEAVE
#undef RS_START
#undef RS_WAIT
#undef RS_O_CHAIN
#undef RS_O_ACC
#undef RS_O_END
#undef RS_O_START
#undef RS_I_CHAIN
#undef RS_I_START

template <int MODE>
hipError_t launchStreamT(const DevDfa &d, const Batch &b, const LaunchCfg &cfg,
                         hipStream_t stream) {
  const uint64_t linesPerTile = uint64_t(kStreamThreads) * kStreamChains;
  const uint64_t tiles = (b.n + linesPerTile - 1) / linesPerTile;
  const uint64_t blocks = tiles < uint64_t(cfg.numCUs) ? tiles : uint64_t(cfg.numCUs);
  hipLaunchKernelGGL(k_stream<MODE>, dim3(uint32_t(blocks)), dim3(kStreamThreads), 0, stream, d, b);
  return hipGetLastError();
}
