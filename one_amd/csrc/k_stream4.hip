// k_stream4.hip - the fixed-stride hot path, second form: more dependent chains per CU.
//
// Same work as k_stream.h's k_stream<MODE, ., ., kTabFused> (styles Last / Full of check / match,
// include/Matcher.h:363-495 of /root/reference/quol/red, lines that are whole 64-byte blocks, fused
// [state][byte] u8 table of at most 256 states in LDS) with the things the round-1 counters asked
// for (SQ_LDS_IDX_ACTIVE only 40-50 % of the launch with 16 chains per CU):
//  * C = 4 (or 3) lines per lane: 32 (24) dependent chains per CU keep the LDS queue full - a
//    64-lane ds_read_u8 gather of the table costs ~7 LDS cycles on random bytes, so 16 chains
//    cover 112 cycles, less than one step's round trip (~64 cycles of LDS pipeline + perm + issue).
//  * the byte step is one asm statement per input DWORD (4 steps x C chains): inside it every wait
//    is counted - `s_waitcnt lgkmcnt(C-1)` before a chain's next lookup lets the C-1 younger
//    lookups of the other chains stay in flight - and only its end drains (form (i) of the
//    local guide, cdna_hip_programming.md 5.7: loads and their waits in ONE statement, so the
//    compiler never sees a register whose data has not landed).
//  * work is handed out per WAVE, not per workgroup: a wave-tile is 64 x C consecutive lines; a
//    workgroup owns the tiles blockIdx.x + gridDim.x * j and its 8 waves take them in order from
//    a ticket counter in LDS, so waves never wait for each other after the table barrier and a
//    batch that is not a multiple of the machine size is balanced to one wave-tile per CU.
//  * blocks are 64 bytes per chain in two register sets (next block streams in while this one is
//    walked): 2 x C x 16 data VGPRs.
// Bookkeeping, fold and result stores follow k_stream.h (positions are block-relative inline
// constants folded once per 64 bytes; stores are unconditional, lanes with nothing to report
// store to the DFA's sink, so the compiler's in-order vmcnt counts stay exact).
// gfx940/950 hazard honoured inside the strings: a VALU-written SGPR mask is read by a VALU
// no earlier than 2 instructions later (all compares of a step are issued before its selects).
#include "kernels.h"

#include <cstdlib>
#include <type_traits>

namespace redgpu {

namespace {

constexpr int kS4Threads = 512;
constexpr uint32_t kS4Tab = 65536;

// numeric values of k_stream.h's StreamMode
constexpr int kM4LastStartEnd = 0, kM4LastEnd = 1, kM4FullStart = 3, kM4Full = 4;

struct Book4 {
  uint32_t acc;    // last accepting state seen
  uint32_t end;    // block-relative position (1..63) of that accept, 0 = none in this block
  uint32_t start;  // block-relative position of the last "left the initial state", 0 = none
};

// ---- the asm pieces -----------------------------------------------------------------------
// chain c, step k: X = register holding the state on entry of the step, Y = the other one
// (receives address, then the next state); OLD / NEW = lane masks "state == initial" of the
// previous / this state.
#define S4_ISSUE(c, X, Y, k) \
  "v_perm_b32 %[" #Y #c "], %[" #X #c "], %[w" #c "], %[sel" #k "]\n\t" \
  "ds_read_u8 %[" #Y #c "], %[" #Y #c "]\n\t"
#define S4_CMPA(c, X) "v_cmp_le_u32_e64 %[m" #c "], %[T], %[" #X #c "]\n\t"
#define S4_CMPI(c, X, NEW) "v_cmp_eq_u32_e64 %[" #NEW #c "], %[init], %[" #X #c "]\n\t"
#define S4_ACC(c, X) "v_cndmask_b32_e64 %[acc" #c "], %[acc" #c "], %[" #X #c "], %[m" #c "]\n\t"
#define S4_END(c, k) "v_cndmask_b32_e64 %[e" #c "], %[e" #c "], %[p" #k "], %[m" #c "]\n\t"
#define S4_LEAVE(c, OLD, NEW) "s_andn2_b64 %[l" #c "], %[" #OLD #c "], %[" #NEW #c "]\n\t"
#define S4_START(c, k) "v_cndmask_b32_e64 %[st" #c "], %[st" #c "], %[p" #k "], %[l" #c "]\n\t"

// lists over the chains (C = 3 uses the first three)
#define S4_ALL3(M, ...) M(0, __VA_ARGS__) M(1, __VA_ARGS__) M(2, __VA_ARGS__)
#define S4_ALL4(M, ...) S4_ALL3(M, __VA_ARGS__) M(3, __VA_ARGS__)

// the issue half of step k >= 1: each chain waits for ITS previous lookup only
#define S4_WISSUE3(c, X, Y, k) "s_waitcnt lgkmcnt(2)\n\t" S4_ISSUE(c, X, Y, k)
#define S4_WISSUE4(c, X, Y, k) "s_waitcnt lgkmcnt(3)\n\t" S4_ISSUE(c, X, Y, k)

// bookkeeping of the state in X at block-relative position p<k>
#define S4_BOOK_LSE(ALL, X, OLD, NEW, k) \
  ALL(S4_CMPA, X) ALL(S4_CMPI, X, NEW) ALL(S4_ACC, X) ALL(S4_END, k) ALL(S4_LEAVE, OLD, NEW) \
  ALL(S4_START, k)
#define S4_BOOK_LE(ALL, X, OLD, NEW, k) ALL(S4_CMPA, X) ALL(S4_ACC, X) ALL(S4_END, k)
#define S4_BOOK_FS(ALL, X, OLD, NEW, k) ALL(S4_CMPI, X, NEW) "s_nop 0\n\t" ALL(S4_LEAVE, OLD, NEW) ALL(S4_START, k)
#define S4_BOOK_F(ALL, X, OLD, NEW, k)

// four steps (one input dword per chain); states end up in s<c>, masks in i<c>
#define S4_DWORD(ALL, WISSUE, BOOK)                                                     \
  ALL(S4_ISSUE, s, u, 0) BOOK(ALL, s, i, j, 0)                                          \
  ALL(WISSUE, u, s, 1) BOOK(ALL, u, j, i, 1)                                            \
  ALL(WISSUE, s, u, 2) BOOK(ALL, s, i, j, 2)                                            \
  ALL(WISSUE, u, s, 3) BOOK(ALL, u, j, i, 3)                                            \
  "s_waitcnt lgkmcnt(0)"

#define S4_O_CHAIN(c) [s##c] "+v"(s[c]), [u##c] "=&v"(u[c])
#define S4_O_ACC(c) [m##c] "=&s"(m[c]), [acc##c] "+v"(b[c].acc), [e##c] "+v"(b[c].end)
#define S4_O_START(c) [i##c] "+s"(was[c]), [j##c] "=&s"(jm[c]), [l##c] "=&s"(l[c]), [st##c] "+v"(b[c].start)
#define S4_I_W(c) [w##c] "v"(w[c])
#define S4_I_COMMON                                                                       \
  [sel0] "s"(0x0c0c0400u), [sel1] "s"(0x0c0c0401u), [sel2] "s"(0x0c0c0402u),              \
      [sel3] "s"(0x0c0c0403u), [p0] "n"(IDX), [p1] "n"(IDX + 1), [p2] "n"(IDX + 2),       \
      [p3] "n"(IDX + 3)

// One input dword of every chain: 4 byte steps.  IDX = block-relative position of the first.
template <int MODE, int C, int IDX>
__device__ __forceinline__ void stepDword(uint32_t (&s)[C], const uint32_t (&w)[C], Book4 (&b)[C],
                                          uint64_t (&was)[C], uint32_t T, uint32_t init) {
  uint32_t u[C];
  uint64_t m[C], l[C], jm[C];
  static_assert(C == 3 || C == 4, "3 or 4 chains");
  if constexpr (C == 4) {
    if constexpr (MODE == kM4LastStartEnd) {
      asm volatile(S4_DWORD(S4_ALL4, S4_WISSUE4, S4_BOOK_LSE)
                   : S4_O_CHAIN(0), S4_O_CHAIN(1), S4_O_CHAIN(2), S4_O_CHAIN(3), S4_O_ACC(0),
                     S4_O_ACC(1), S4_O_ACC(2), S4_O_ACC(3), S4_O_START(0), S4_O_START(1),
                     S4_O_START(2), S4_O_START(3)
                   : S4_I_W(0), S4_I_W(1), S4_I_W(2), S4_I_W(3), [T] "s"(T), [init] "s"(init),
                     S4_I_COMMON
                   : "memory", "scc");
    } else if constexpr (MODE == kM4LastEnd) {
      asm volatile(S4_DWORD(S4_ALL4, S4_WISSUE4, S4_BOOK_LE)
                   : S4_O_CHAIN(0), S4_O_CHAIN(1), S4_O_CHAIN(2), S4_O_CHAIN(3), S4_O_ACC(0),
                     S4_O_ACC(1), S4_O_ACC(2), S4_O_ACC(3)
                   : S4_I_W(0), S4_I_W(1), S4_I_W(2), S4_I_W(3), [T] "s"(T), S4_I_COMMON
                   : "memory");
    } else if constexpr (MODE == kM4FullStart) {
      asm volatile(S4_DWORD(S4_ALL4, S4_WISSUE4, S4_BOOK_FS)
                   : S4_O_CHAIN(0), S4_O_CHAIN(1), S4_O_CHAIN(2), S4_O_CHAIN(3), S4_O_START(0),
                     S4_O_START(1), S4_O_START(2), S4_O_START(3)
                   : S4_I_W(0), S4_I_W(1), S4_I_W(2), S4_I_W(3), [init] "s"(init), S4_I_COMMON
                   : "memory", "scc");
    } else {
      asm volatile(S4_DWORD(S4_ALL4, S4_WISSUE4, S4_BOOK_F)
                   : S4_O_CHAIN(0), S4_O_CHAIN(1), S4_O_CHAIN(2), S4_O_CHAIN(3)
                   : S4_I_W(0), S4_I_W(1), S4_I_W(2), S4_I_W(3), S4_I_COMMON
                   : "memory");
    }
  } else {
    if constexpr (MODE == kM4LastStartEnd) {
      asm volatile(S4_DWORD(S4_ALL3, S4_WISSUE3, S4_BOOK_LSE)
                   : S4_O_CHAIN(0), S4_O_CHAIN(1), S4_O_CHAIN(2), S4_O_ACC(0), S4_O_ACC(1),
                     S4_O_ACC(2), S4_O_START(0), S4_O_START(1), S4_O_START(2)
                   : S4_I_W(0), S4_I_W(1), S4_I_W(2), [T] "s"(T), [init] "s"(init), S4_I_COMMON
                   : "memory", "scc");
    } else if constexpr (MODE == kM4LastEnd) {
      asm volatile(S4_DWORD(S4_ALL3, S4_WISSUE3, S4_BOOK_LE)
                   : S4_O_CHAIN(0), S4_O_CHAIN(1), S4_O_CHAIN(2), S4_O_ACC(0), S4_O_ACC(1),
                     S4_O_ACC(2)
                   : S4_I_W(0), S4_I_W(1), S4_I_W(2), [T] "s"(T), S4_I_COMMON
                   : "memory");
    } else if constexpr (MODE == kM4FullStart) {
      asm volatile(S4_DWORD(S4_ALL3, S4_WISSUE3, S4_BOOK_FS)
                   : S4_O_CHAIN(0), S4_O_CHAIN(1), S4_O_CHAIN(2), S4_O_START(0), S4_O_START(1),
                     S4_O_START(2)
                   : S4_I_W(0), S4_I_W(1), S4_I_W(2), [init] "s"(init), S4_I_COMMON
                   : "memory", "scc");
    } else {
      asm volatile(S4_DWORD(S4_ALL3, S4_WISSUE3, S4_BOOK_F)
                   : S4_O_CHAIN(0), S4_O_CHAIN(1), S4_O_CHAIN(2)
                   : S4_I_W(0), S4_I_W(1), S4_I_W(2), S4_I_COMMON
                   : "memory");
    }
  }
}

template <int C>
struct Block4 {
  uint4 p[C][4];  // 64 bytes per chain
};

// PAIR (lines that are whole 128-byte cache lines): a lane asks for both 64-byte halves of a
// cache line back to back (the first misses, the second hits the line in L1) instead of coming
// back for the second half a block's walk later, when L1 has long dropped the line (64-byte
// blocks on 4 KiB lines: every line fetched from L2 twice, 3.6 against 4.4 TB/s).  Three
// 64-byte register slots per chain - the half being walked, and the two halves of the NEXT cache
// line - advance by register moves (48 v_mov per chain per 128 bytes), so the loop body stays
// two half-blocks long.
template <int MODE, int C, bool PAIR>
__global__ void __launch_bounds__(kS4Threads)
k_stream4(DevDfa d, Batch io) {
  constexpr bool kAcc = MODE == kM4LastStartEnd || MODE == kM4LastEnd;
  constexpr bool kStart = MODE == kM4LastStartEnd || MODE == kM4FullStart;
  __shared__ __align__(16) uint8_t lds[kS4Tab + 1024 + 16];  // table at LDS offset 0
  uint8_t *tab = lds;
  int32_t *ldsRes = reinterpret_cast<int32_t *>(lds + kS4Tab);
  uint32_t *ticket = reinterpret_cast<uint32_t *>(lds + kS4Tab + 1024);

  const uint32_t init = d.init, T = d.firstAccept;
  const uint32_t lineLen = uint32_t(io.stride);
  const uint32_t R = lineLen / 64;  // blocks per line
  constexpr uint64_t kTileLines = 64 * C;
  const uint64_t nTiles = (io.n + kTileLines - 1) / kTileLines;
  const uint64_t G = gridDim.x;
  if (blockIdx.x >= nTiles) return;
  const uint32_t myTiles = uint32_t((nTiles - blockIdx.x + G - 1) / G);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

  auto issue = [&](Block4<C> &blk, uint64_t tile, uint32_t r) {
    const uint8_t *p[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      uint64_t ln = tile * kTileLines + uint64_t(c) * 64 + lane;
      if (ln >= io.n) ln = io.n - 1;
      p[c] = io.data + ln * lineLen + uint64_t(r) * 64;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int c = 0; c < C; ++c) blk.p[c][k] = reinterpret_cast<const uint4 *>(p[c])[k];
    }
  };

  // both halves of a cache line, chain by chain: the eight requests of one line are adjacent
  // in the wave's instruction stream (the first misses, seven hit the line in L1) - requested
  // half by half they are 16 instructions = 128 KB of other lines apart and L1 has dropped it
  auto issuePair = [&](Block4<C> &h0, Block4<C> &h1, uint64_t tile, uint32_t r, uint32_t second) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      uint64_t ln = tile * kTileLines + uint64_t(c) * 64 + lane;
      if (ln >= io.n) ln = io.n - 1;
      const uint8_t *p = io.data + ln * lineLen + uint64_t(r) * 64;
      const uint8_t *q = p + second;
#pragma unroll
      for (int k = 0; k < 4; ++k) h0.p[c][k] = reinterpret_cast<const uint4 *>(p)[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) h1.p[c][k] = reinterpret_cast<const uint4 *>(q)[k];
    }
  };

  // table requests first, the first input block's right behind them, LDS stores and the barrier
  // after (k_stream.h measured both orders)
  const uint4 *tsrc = reinterpret_cast<const uint4 *>(d.table);
  const uint32_t n16 = d.tableBytes / 16;
  constexpr uint32_t kStagePieces = kS4Tab / 16 / kS4Threads;
  uint4 tv[kStagePieces];
#pragma unroll
  for (uint32_t k = 0; k < kStagePieces; ++k) {
    const uint32_t i = k * kS4Threads + threadIdx.x;
    tv[k] = i < n16 ? tsrc[i] : make_uint4(0, 0, 0, 0);
  }
  const int32_t myRes = threadIdx.x < d.nStates ? d.result[threadIdx.x] : 0;
  Block4<C> A, B;
  // the first tile of a wave is its own number (no ticket needed before the barrier)
  uint32_t j = wave;
  const bool any = j < myTiles;
  uint64_t tile = blockIdx.x + G * uint64_t(any ? j : 0);
  uint32_t r = 0;
  issue(A, tile, 0);
  {
    uint4 *dst = reinterpret_cast<uint4 *>(tab);
#pragma unroll
    for (uint32_t k = 0; k < kStagePieces; ++k) dst[k * kS4Threads + threadIdx.x] = tv[k];
    if (threadIdx.x < 256) ldsRes[threadIdx.x] = myRes;
    if (threadIdx.x == 0) *ticket = kS4Threads / 64;
  }
  // the byte steps read the table from inline asm only: make its address reach one
  asm volatile("" : : "v"(tab) : "memory");
  __syncthreads();
  if (!any) return;

  auto take = [&]() -> uint32_t {
    uint32_t v = 0;
    if (lane == 0) v = atomicAdd(ticket, 1u);
    return __builtin_amdgcn_readfirstlane(v);
  };
  uint32_t jn = take();

  uint32_t s[C], accS[C], endv[C], startv[C];
  uint64_t was[C];

  auto walk = [&](const Block4<C> &blk, auto storeTag) {
    constexpr bool STORE = decltype(storeTag)::value;
    if (r == 0) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        s[c] = init; accS[c] = 0; endv[c] = 0; startv[c] = 0;
        was[c] = ~0ull;
      }
    }
    Book4 b[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      b[c].acc = accS[c]; b[c].end = 0; b[c].start = 0;
      // lane masks carried around the loop feed "s" operands: pinned, or the compiler may park
      // them in VGPRs ("illegal VGPR to SGPR copy")
      const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(was[c]));
      const uint32_t hi = __builtin_amdgcn_readfirstlane(uint32_t(was[c] >> 32));
      was[c] = (uint64_t(hi) << 32) | lo;
    }
    uint32_t w[C];
#define S4_WORD(Q, K, FIELD)                                          \
  _Pragma("unroll") for (int c = 0; c < C; ++c) w[c] = blk.p[c][Q].FIELD; \
  stepDword<MODE, C, 16 * Q + 4 * K>(s, w, b, was, T, init);
#define S4_PIECE(Q) S4_WORD(Q, 0, x) S4_WORD(Q, 1, y) S4_WORD(Q, 2, z) S4_WORD(Q, 3, w)
    S4_PIECE(0) S4_PIECE(1) S4_PIECE(2) S4_PIECE(3)
#undef S4_PIECE
#undef S4_WORD
    const uint32_t off = r * 64;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      if (kAcc) {
        accS[c] = b[c].acc;
        endv[c] = b[c].end ? off + b[c].end : endv[c];
        if (s[c] >= T) { accS[c] = s[c]; endv[c] = off + 64; }
      }
      if (kStart) {
        startv[c] = b[c].start ? off + b[c].start - 1 : startv[c];
        const bool wasInit63 = (was[c] >> lane) & 1;
        if (wasInit63 && s[c] != init) startv[c] = off + 63;
      }
    }
    // results after EVERY block that can end a line, without a branch (k_stream.h): lanes whose
    // line ends here store into the line's slots, everyone else into the DFA's sink
    if constexpr (!STORE) return;
    const bool lineEnd = r + 1 == R;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const uint64_t ln = tile * kTileLines + uint64_t(c) * 64 + lane;
      const bool report = lineEnd && ln < io.n;
      int32_t rr;
      uint32_t en;
      if (kAcc) {
        rr = ldsRes[accS[c] & 0xffu];
        rr = endv[c] ? rr : 0;
        en = endv[c];
      } else {
        rr = ldsRes[s[c] & 0xffu];
        rr = s[c] >= T ? rr : 0;
        en = lineLen;
      }
      *(report ? io.result + ln : reinterpret_cast<int32_t *>(d.sink)) = rr;
      *(report && io.end ? io.end + ln : reinterpret_cast<uint64_t *>(d.sink)) = rr ? uint64_t(en) : 0;
      if (kStart)
        *(report && io.start ? io.start + ln : reinterpret_cast<uint64_t *>(d.sink)) =
            rr ? uint64_t(startv[c]) : 0;
    }
  };

  // next unit after (tile, r): the line's next block, else the first block of the next tile of
  // this wave; past the end of the work the current block is requested again (requests stay
  // unconditional so the compiler's vmcnt counts stay exact)
  uint64_t nTile;
  uint32_t nR;
  bool more;
  auto next = [&]() {
    if (r + 1 < R) { nTile = tile; nR = r + 1; more = true; }
    else if (jn < myTiles) { nTile = blockIdx.x + G * uint64_t(jn); nR = 0; more = true; }
    else { nTile = tile; nR = r; more = false; }
  };
  auto advance = [&]() {
    if (nR == 0) { j = jn; jn = take(); }
    tile = nTile; r = nR;
  };

  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;
  if constexpr (!PAIR) {
    for (;;) {
      next();
      issue(B, nTile, nR);
      walk(A, Yes{});
      if (!more) break;
      advance();
      next();
      issue(A, nTile, nR);
      walk(B, Yes{});
      if (!more) break;
      advance();
    }
  } else {
    // A = the half being walked, B / N2 = the halves that follow; (tile, r) = the half in A
    Block4<C> N2;
    issue(B, tile, 1);  // the first cache line's second half (its first is in A already)
    for (;;) {
      walk(A, No{});    // r even: no line ends here (R is even)
      r += 1;
      A = B;
      // the next cache line: this line's next pair, else the first pair of the wave's next tile
      next();           // from (tile, r odd): (tile, r + 1) or (next tile, 0) or a re-read
      issuePair(B, N2, nTile, nR, more ? 64u : 0u);
      walk(A, Yes{});   // r odd
      if (!more) break;
      advance();
      A = B;
      B = N2;
    }
  }
}

// ---- LDS gather calibration (bench.py: roofline.lds_roof) -----------------------------------
// The walk without its memory side: the DFA's 64 KB table in LDS (pseudo-random bytes when the
// DFA's primary table is not a fused u8 one), 4 chains per lane, 8 waves per CU, the 4-chain
// lookup-only step above (counted waits), input bytes from 16 registers per chain filled by a
// hash of (lane, chain, workgroup) - uniformly random bytes, as configs[1] feeds the real walk.
__global__ void __launch_bounds__(kS4Threads)
k_diag_lds(DevDfa d, uint32_t rounds, uint32_t *sink) {
  __shared__ __align__(16) uint8_t lds[kS4Tab];
  const bool real = d.tableKind == 1 && d.tableBytes == kS4Tab;
  for (uint32_t i = threadIdx.x; i < kS4Tab / 16; i += kS4Threads) {
    uint4 v;
    if (real) {
      v = reinterpret_cast<const uint4 *>(d.table)[i];
    } else {
      uint64_t z = (uint64_t(i) + 1) * 0x9E3779B97F4A7C15ull;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z ^= z >> 31;
      v = make_uint4(uint32_t(z), uint32_t(z >> 32), uint32_t(z * 3), uint32_t((z * 5) >> 32));
    }
    reinterpret_cast<uint4 *>(lds)[i] = v;
  }
  asm volatile("" : : "v"(lds) : "memory");
  __syncthreads();
  uint32_t w[16][4];
#pragma unroll
  for (int k = 0; k < 16; ++k)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      uint64_t z = (uint64_t(blockIdx.x) * kS4Threads + threadIdx.x) * 64 + k * 4 + c + 1;
      z *= 0x9E3779B97F4A7C15ull;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      w[k][c] = uint32_t(z ^ (z >> 31));
    }
  uint32_t s[4] = {d.init & 0xffu, (d.init + 1) & 0xffu, (d.init + 2) & 0xffu, (d.init + 3) & 0xffu};
  Book4 b[4];
  uint64_t was[4];
  for (uint32_t r = 0; r < rounds; ++r) {
#pragma unroll
    for (int k = 0; k < 16; ++k) stepDword<kM4Full, 4, 0>(s, w[k], b, was, 0u, 0u);
  }
  if ((s[0] ^ s[1] ^ s[2] ^ s[3]) == 0x12345u) atomicAdd(sink, 1u);  // keeps the walk alive
}

template <int MODE, int C>
hipError_t launchS4(const DevDfa &d, const Batch &b, const LaunchCfg &cfg, hipStream_t stream) {
  const uint64_t tiles = (b.n + 64 * C - 1) / (64 * C);
  const uint64_t blocks = tiles < uint64_t(cfg.numCUs) ? tiles : uint64_t(cfg.numCUs);
  if (b.stride % 128 == 0)
    hipLaunchKernelGGL((k_stream4<MODE, C, true>), dim3(uint32_t(blocks)), dim3(kS4Threads), 0,
                       stream, d, b);
  else
    hipLaunchKernelGGL((k_stream4<MODE, C, false>), dim3(uint32_t(blocks)), dim3(kS4Threads), 0,
                       stream, d, b);
  return hipGetLastError();
}

template <int C>
hipError_t launchS4M(int mode, const DevDfa &d, const Batch &b, const LaunchCfg &cfg,
                     hipStream_t stream) {
  switch (mode) {
  case kM4LastStartEnd: return launchS4<kM4LastStartEnd, C>(d, b, cfg, stream);
  case kM4LastEnd: return launchS4<kM4LastEnd, C>(d, b, cfg, stream);
  case kM4FullStart: return launchS4<kM4FullStart, C>(d, b, cfg, stream);
  default: return launchS4<kM4Full, C>(d, b, cfg, stream);
  }
}

}  // namespace

// chains per lane of the fixed-stride hot path: 0 = k_stream.h's two-chain kernel, the default.
// Measured on MI355X (scripts/lab_stream.py; profiles/r02_lab_stream_chains.log), SYN-256, full
// Outcome, 2 / 3 / 4 chains per lane:
//   2^20 x 64 B    25.3 / 30.0 / 30.7 us per launch (17.8 / 25.6 / 29.1 on three streams)
//   2^24 x 64 B    271 / 269 / 269 us: all three at the HBM read + write rate (84 B of traffic per
//                  64-byte line, 5.2 TB/s)
//   2^21 x 4 KiB   1.98 / 2.27 / 2.23 ms with the cache-line pairs below (2.29 / 2.41 ms with plain
//                  64-byte blocks); URI-D on text, where the table gathers do not conflict at all:
//                  1.96 / - / 2.23 ms
// So the round-1 reading "LDS only 40-50 % busy => chain-starved" does not hold: twice the chains
// fill the LDS queue and the launch gets slower.  What a wave has little of is ISSUE slots - every
// instruction of a wave's in-order stream costs it >= 4 cycles (MI355X_MICROARCH.md, issue cost
// row), ~9.5 instructions per byte step - and the same 8 waves per CU now carry twice the work per
// wave, need twice the data before their first step (128 KB per CU at 4 chains) and hold 230
// VGPRs.  The two-chain kernel sits within 20 % of three roofs at once (LDS gather 5.4 TB/s
// measured by k_diag_lds below, streaming read 5.7 TB/s, instruction issue ~7 TB/s); this file
// stays as the measured alternative and as the home of the calibration kernel.
int stream4Chains() {
  static const int chains = [] {
    const char *e = getenv("REDGPU_STREAM_CHAINS");
    const int v = e ? atoi(e) : 0;
    return (v == 3 || v == 4) ? v : 0;
  }();
  return chains;
}

bool stream4Eligible(const DevDfa &d, const Batch &b, const LaunchCfg &cfg) {
  if (cfg.streamChains == 2 || (!cfg.streamChains && !stream4Chains())) return false;
  // by default only with enough wave-tiles to give every wave of the machine one (few long
  // lines: k_stream.h's 256-thread form spreads them over more CUs)
  const uint64_t tiles = (b.n + 64 * 4 - 1) / (64 * 4);
  return d.tableKind == 1 /* REDGPU_TAB_LDS_FUSED_U8 */ && d.nStates <= 256 &&
         d.tableBytes <= kS4Tab &&
         (cfg.streamChains >= 3 || tiles >= uint64_t(cfg.numCUs) * (kS4Threads / 64)) &&
         tiles / uint64_t(cfg.numCUs) < (1ull << 31);
}

hipError_t launchDiagLds(const DevDfa &d, uint32_t rounds, uint32_t *sink, int numCUs,
                         hipStream_t stream, uint64_t *lookups) {
  hipLaunchKernelGGL(k_diag_lds, dim3(uint32_t(numCUs)), dim3(kS4Threads), 0, stream, d, rounds,
                     sink);
  if (lookups) *lookups = uint64_t(numCUs) * kS4Threads * 4 * 64 * rounds;
  return hipGetLastError();
}

hipError_t launchStream4(int mode, const DevDfa &d, const Batch &b, const LaunchCfg &cfg,
                         hipStream_t stream) {
  const int chains = cfg.streamChains >= 3 ? cfg.streamChains : stream4Chains();
  return chains == 3 ? launchS4M<3>(mode, d, b, cfg, stream) : launchS4M<4>(mode, d, b, cfg, stream);
}

}  // namespace redgpu
