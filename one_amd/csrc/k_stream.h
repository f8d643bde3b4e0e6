// k_stream.h - the streaming fixed-stride kernel (included by kernels.hip, inside its namespace).
//
// Hot path of BASELINE.json configs[1]/[2]: lines of a fixed length that is a multiple of 64
// bytes, DFA with at most 256 reachable states (fused [state][byte] u8 table, 64 KB, resident
// at LDS offset 0), styles Last and Full of check / match (include/Matcher.h:363-495).
//
// Shape, and why (every number below was measured on MI355X with the round-1 variant lab, tools/tune.hip in the history up to commit 151617f; scripts/lab_stream.py times the product kernels now):
//  * One workgroup of 512 threads per CU shares one table copy (8 waves x 2 lines = 16 dependent
//    chains per CU; 768 and 1024 threads - 3 and 4 waves per SIMD to hide the LDS latency a step
//    waits for - win only on huge batches of 64-byte lines (2^24 x 64 B: 3.84 against 3.69 TB/s)
//    and lose everywhere else (2^21 x 4 KiB: 3.92 / 3.45 against 4.25 TB/s; 2^20 x 64 B: 26.8
//    against 24.7 us)); staging it costs ~1.6 us and is
//    done BEFORE any input is requested - input requests issued earlier sit in front of the
//    table's in the CU's memory queues and delay the table barrier to ~6 us.
//  * Each lane walks 2 lines at once (two dependent chains per lane).  Input blocks are
//    requested unconditionally (past the end of the work the last block is re-read) so the
//    compiler's in-order vmcnt counts stay exact - a conditional request made it fall back to
//    vmcnt(0), which serialised the walk behind a full HBM round trip.
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
  kSmAdvance = 5,       // StatefulMatcher chunk: kSmFull's walk, state[] read at the first block
                        // of a line and written back after the last (include/Matcher.h:770-792)
  kSmChunk = 6,         // one CHUNK of a long line (k_chunk.h): kSmLastStartEnd's bookkeeping from
                        // an entry state (state[] in / out), raw records out instead of an Outcome:
                        // result[] = state of the last accept or -1, end[] = its chunk-relative
                        // end (0 = none), start[] = last "left the initial state" position + 1
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
  if constexpr ((MODE == kSmLastStartEnd || MODE == kSmChunk)) {
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

// ---- class-table form (TABK == kTabCls) ----------------------------------------------------
// DFAs of more than 256 states whose [state][class] u16 table fits 64 KB (343-state URI DFA:
// 17 KB).  LDS: [eq2: byte -> 2 x class, 256 B][rows at +256: entry = byte offset of the
// target's row].  The state register holds a ROW OFFSET, so a step is v_add_u32 (row + 2 x
// class) and ds_read_u16 ... offset:256; the accept / initial compares work on row offsets
// unchanged (offsets grow with the state index).  The class lookups do not depend on the
// state: the four of a word (per chain) are issued ahead by the compiler and only the row
// lookup sits on the dependent chain.  Two LDS gathers per byte instead of one: half the fused
// table's rate, four times k_generic's.
#define RC_ADD(c) "v_add_u32 %[a" #c "], %[s" #c "], %[k" #c "]\n\t"
#define RC_READ(c) "ds_read_u16 %[t" #c "], %[a" #c "] offset:256\n\t"
#define RC_I_CHAIN(c) [s##c] "v"(s[c]), [k##c] "v"(cls[c])
// index form (tables above 64 KB): row = state x rowBytes
#define RC_MAD(c) "v_mad_u32_u24 %[a" #c "], %[s" #c "], %[rowb], %[k" #c "]\n\t"

// ds_read_u16 at a given address + the step's bookkeeping (the index form's second half)
#define RC_I_READ(c) [s##c] "v"(s[c]), [a##c] "v"(a[c])
#define RC_O_READ(c) [t##c] "=&v"(t[c])
template <int MODE, int IDX>
__device__ __forceinline__ void streamStepRead(uint32_t (&s)[2], const uint32_t (&a)[2],
                                               uint32_t (&t)[2], StreamBook (&b)[2],
                                               const uint64_t (&wasI)[2], uint64_t (&isI)[2],
                                               uint32_t T, uint32_t init) {
  uint64_t m[2], l[2];
  if constexpr ((MODE == kSmLastStartEnd || MODE == kSmChunk)) {
    asm volatile(RC_READ(0) RC_READ(1)
                 RS_CMPA(0) RS_CMPA(1) RS_CMPI(0) RS_CMPI(1)
                 RS_ACC(0) RS_ACC(1) RS_END(0) RS_END(1)
                 RS_LEAVE(0) RS_LEAVE(1) RS_START(0) RS_START(1) RS_WAIT
                 : RC_O_READ(0), RC_O_READ(1), RS_O_ACC(0), RS_O_ACC(1), RS_O_END(0),
                   RS_O_END(1), RS_O_START(0), RS_O_START(1)
                 : RC_I_READ(0), RC_I_READ(1), RS_I_START(0), RS_I_START(1), [T] "s"(T),
                   [init] "s"(init), [idx] "n"(IDX)
                 : "memory", "scc");
  } else if constexpr (MODE == kSmLastEnd) {
    asm volatile(RC_READ(0) RC_READ(1)
                 RS_CMPA(0) RS_CMPA(1) "s_nop 0\n\t" RS_ACC(0) RS_ACC(1) RS_END(0) RS_END(1) RS_WAIT
                 : RC_O_READ(0), RC_O_READ(1), RS_O_ACC(0), RS_O_ACC(1), RS_O_END(0), RS_O_END(1)
                 : RC_I_READ(0), RC_I_READ(1), [T] "s"(T), [idx] "n"(IDX)
                 : "memory");
  } else if constexpr (MODE == kSmFullStart) {
    asm volatile(RC_READ(0) RC_READ(1)
                 RS_CMPI(0) RS_CMPI(1) "s_nop 0\n\t" RS_LEAVE(0) RS_LEAVE(1) RS_START(0) RS_START(1)
                 RS_WAIT
                 : RC_O_READ(0), RC_O_READ(1), RS_O_START(0), RS_O_START(1)
                 : RC_I_READ(0), RC_I_READ(1), RS_I_START(0), RS_I_START(1), [init] "s"(init),
                   [idx] "n"(IDX)
                 : "memory", "scc");
  } else {
    asm volatile(RC_READ(0) RC_READ(1) RS_WAIT
                 : RC_O_READ(0), RC_O_READ(1)
                 : RC_I_READ(0), RC_I_READ(1)
                 : "memory");
  }
}

template <int MODE, int IDX, bool BIG>
__device__ __forceinline__ void streamStepCls(uint32_t (&s)[2], const uint32_t (&cls)[2],
                                              StreamBook (&b)[2], const uint64_t (&wasI)[2],
                                              uint64_t (&isI)[2], uint32_t T, uint32_t init,
                                              uint32_t rowb) {
  uint32_t a[2], t[2];
  uint64_t m[2], l[2];
  if constexpr (BIG) {
    // the address first (its own statement: one more input), then the shared read + bookkeeping
    asm volatile(RC_MAD(0) RC_MAD(1)
                 : [a0] "=&v"(a[0]), [a1] "=&v"(a[1])
                 : [s0] "v"(s[0]), [s1] "v"(s[1]), [k0] "v"(cls[0]), [k1] "v"(cls[1]),
                   [rowb] "s"(rowb));
    streamStepRead<MODE, IDX>(s, a, t, b, wasI, isI, T, init);
    s[0] = t[0];
    s[1] = t[1];
    return;
  }
  if constexpr ((MODE == kSmLastStartEnd || MODE == kSmChunk)) {
    asm volatile(RC_ADD(0) RC_ADD(1) RC_READ(0) RC_READ(1)
                 RS_CMPA(0) RS_CMPA(1) RS_CMPI(0) RS_CMPI(1)
                 RS_ACC(0) RS_ACC(1) RS_END(0) RS_END(1)
                 RS_LEAVE(0) RS_LEAVE(1) RS_START(0) RS_START(1) RS_WAIT
                 : RS_O_CHAIN(0), RS_O_CHAIN(1), RS_O_ACC(0), RS_O_ACC(1), RS_O_END(0),
                   RS_O_END(1), RS_O_START(0), RS_O_START(1)
                 : RC_I_CHAIN(0), RC_I_CHAIN(1), RS_I_START(0), RS_I_START(1), [T] "s"(T),
                   [init] "s"(init), [idx] "n"(IDX)
                 : "memory", "scc");
  } else if constexpr (MODE == kSmLastEnd) {
    asm volatile(RC_ADD(0) RC_ADD(1) RC_READ(0) RC_READ(1)
                 RS_CMPA(0) RS_CMPA(1) "s_nop 0\n\t" RS_ACC(0) RS_ACC(1) RS_END(0) RS_END(1) RS_WAIT
                 : RS_O_CHAIN(0), RS_O_CHAIN(1), RS_O_ACC(0), RS_O_ACC(1), RS_O_END(0), RS_O_END(1)
                 : RC_I_CHAIN(0), RC_I_CHAIN(1), [T] "s"(T), [idx] "n"(IDX)
                 : "memory");
  } else if constexpr (MODE == kSmFullStart) {
    asm volatile(RC_ADD(0) RC_ADD(1) RC_READ(0) RC_READ(1)
                 RS_CMPI(0) RS_CMPI(1) "s_nop 0\n\t" RS_LEAVE(0) RS_LEAVE(1) RS_START(0) RS_START(1)
                 RS_WAIT
                 : RS_O_CHAIN(0), RS_O_CHAIN(1), RS_O_START(0), RS_O_START(1)
                 : RC_I_CHAIN(0), RC_I_CHAIN(1), RS_I_START(0), RS_I_START(1), [init] "s"(init),
                   [idx] "n"(IDX)
                 : "memory", "scc");
  } else {
    asm volatile(RC_ADD(0) RC_ADD(1) RC_READ(0) RC_READ(1) RS_WAIT
                 : RS_O_CHAIN(0), RS_O_CHAIN(1)
                 : RC_I_CHAIN(0), RC_I_CHAIN(1)
                 : "memory");
  }
  s[0] = t[0];
  s[1] = t[1];
}

template <int MODE, int Q, bool BIG>
__device__ __forceinline__ void streamWalk16Cls(const uint4 (&piece)[2], uint32_t (&s)[2],
                                                StreamBook (&b)[2], uint64_t (&mA)[2],
                                                uint64_t (&mB)[2], uint32_t T, uint32_t init,
                                                const uint8_t *eq2, uint32_t rowb) {
  uint32_t cl[4][2];
#define RC_WORD(K, FIELD)                                                                  \
  _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                          \
    cl[k][0] = eq2[(piece[0].FIELD >> (8 * k)) & 0xffu];                                   \
    cl[k][1] = eq2[(piece[1].FIELD >> (8 * k)) & 0xffu];                                   \
  }                                                                                        \
  streamStepCls<MODE, 16 * Q + 4 * K + 0, BIG>(s, cl[0], b, mA, mB, T, init, rowb);                   \
  streamStepCls<MODE, 16 * Q + 4 * K + 1, BIG>(s, cl[1], b, mB, mA, T, init, rowb);                   \
  streamStepCls<MODE, 16 * Q + 4 * K + 2, BIG>(s, cl[2], b, mA, mB, T, init, rowb);                   \
  streamStepCls<MODE, 16 * Q + 4 * K + 3, BIG>(s, cl[3], b, mB, mA, T, init, rowb);
  RC_WORD(0, x) RC_WORD(1, y) RC_WORD(2, z) RC_WORD(3, w)
#undef RC_WORD
}

constexpr int kTabFused = 0, kTabHot = 1, kTabCls = 2, kTabClsBig = 3;
constexpr uint32_t kStreamBigLds = 158720;  // the class-table form above 64 KB: one workgroup per CU

// =========================================================================================
// k_stream<MODE, HALVES>: each lane pulls a whole block of its line - 128 bytes (HALVES = 2,
// stride % 128 == 0: one full cache line) or 64 bytes (HALVES = 1) - with back-to-back 16-byte
// loads (the first misses, the rest hit the line in L1) into one of two register sets
// (ping-pong: the next block streams in while this one is walked).  2 lines per lane, 512
// threads, up to 128 data VGPRs.  A 16-byte "piece ring" (request piece p+3 while walking p)
// needs a quarter of the registers but touches every cache line 4-8 times: 2.6 TB/s at 4 KiB
// lines against 4.2 TB/s for this form, and no faster at 64 B either.
// =========================================================================================
template <int HALVES>
struct BlockRegs {
  uint4 p[4 * HALVES];
  uint32_t st;  // kSmAdvance: the line's entry state, requested with the block (else unused)
};

// HALVES = 2: 128-byte blocks (stride % 128 == 0); HALVES = 1: 64-byte blocks (stride % 64 == 0)
//
// HOT = true (REDGPU_TAB_HOT_ROWS DFAs): the LDS table is the 64 KB
// [hot index][byte] u8 table of dfa_image.h - entries are hot indices (index 0 = any pure dead
// end when hotShift is 1), 255 = "this transition leaves the hot set", row 255 an absorbing sink.  The fast walk is unchanged; after
// every 64-byte half-block the wave asks whether any of its lanes sits in the sink, and if so
// those lanes re-walk that half-block from its saved entry state in slowHalf() - hot steps
// through the LDS table, cold steps through the class table in L2 - and skip the fold.  Line-level
// states (accS, the carried cold state g) are GLOBAL device state ids in this mode.
constexpr uint32_t kNoState = 0xffffffffu;

struct SlowBook {  // by value in and out: nothing of the fast path has its address taken
  uint32_t st, accS, endv, startv;
};

template <int MODE>
__device__ __noinline__ SlowBook slowHalf(const DevDfa &d, const uint8_t *tab8, const uint8_t *eq,
                                          const uint8_t *p, uint32_t off, SlowBook in) {
  uint32_t st = in.st, accS = in.accS, endv = in.endv, startv = in.startv;
  constexpr bool kAcc = (MODE == kSmLastStartEnd || MODE == kSmChunk) || MODE == kSmLastEnd;
  constexpr bool kStart = (MODE == kSmLastStartEnd || MODE == kSmChunk) || MODE == kSmFullStart;
  const uint16_t *cls = reinterpret_cast<const uint16_t *>(d.table);
  // (eq: the class map's LDS copy - read from global memory a cold step was two dependent
  // round trips, the class and then the row)
  // the block comes back in four 16-byte requests (it is still in L1/L2), not 64 byte loads:
  // the lanes waiting on this one pay for every round trip
#pragma unroll 1
  for (uint32_t c4 = 0; c4 < 4; ++c4) {
    const uint4 v = *reinterpret_cast<const uint4 *>(p + 16 * c4);
#pragma unroll 1
    for (uint32_t wi = 0; wi < 4; ++wi) {
      const uint32_t word = wi == 0 ? v.x : wi == 1 ? v.y : wi == 2 ? v.z : v.w;
#pragma unroll 1
      for (uint32_t kb = 0; kb < 4; ++kb) {
        const uint32_t k = 16 * c4 + 4 * wi + kb;
        const uint32_t byte = (word >> (8 * kb)) & 0xffu;
        const uint32_t was = st;
        const uint32_t hr = st - d.hotLo;
        const uint32_t nx = hr < d.nHot ? uint32_t(tab8[((hr + d.hotShift) << 8) | byte]) : 255u;
        if (nx != 255u)
          st = (d.hotShift && nx == 0) ? 0u : d.hotLo + nx - d.hotShift;
        else
          st = cls[size_t(st) * d.nClasses + eq[byte]];
        if (kStart && was == d.init && st != was) startv = off + k;
        if (kAcc && st >= d.firstAccept) { accS = st; endv = off + k + 1; }
      }
    }
  }
  return SlowBook{st, accS, endv, startv};
}

template <int MODE, int HALVES, int THREADS, int TABK = kTabFused>
__global__ void __launch_bounds__(THREADS)
k_stream(DevDfa d, Batch io) {
  constexpr bool HOT = TABK == kTabHot;
  constexpr bool BIG = TABK == kTabClsBig;   // class table above 64 KB, index form
  constexpr bool CLS = TABK == kTabCls || BIG;
  constexpr bool IDXD = HOT || CLS;  // the walk's state values are not device state ids
  constexpr bool EARLY = !BIG;  // first input block requested before the table barrier (-2 %)
  constexpr uint32_t BLK = 64 * HALVES;
  constexpr int CH = kStreamChains;
  constexpr bool kAcc = (MODE == kSmLastStartEnd || MODE == kSmChunk) || MODE == kSmLastEnd;
  constexpr bool kStart = (MODE == kSmLastStartEnd || MODE == kSmChunk) || MODE == kSmFullStart;
  constexpr uint32_t kLdsBytes = BIG ? kStreamBigLds : kStreamTabBytes + 1024;
  __shared__ __align__(16) uint8_t lds[kLdsBytes];  // table at LDS offset 0
  uint8_t *tab = lds;
  int32_t *ldsRes = reinterpret_cast<int32_t *>(lds + kStreamTabBytes);

  // HOT: the walk runs in hot-index space (init / firstAccept relative to hotLo; a cold initial
  // state gets an index no lane can hold)
  const uint32_t init =
      HOT ? (d.init - d.hotLo < d.nHot ? d.init - d.hotLo + d.hotShift : 0x1ffu)
          : (CLS && !BIG) ? d.init * d.clsRowBytes : d.init;
  const uint32_t firstAccept = HOT ? d.firstAccept - d.hotLo + d.hotShift
                                   : (CLS && !BIG) ? d.firstAccept * d.clsRowBytes
                                                   : d.firstAccept;
  // HOT: global state id <-> hot index (255 = not hot; index 0 = dead when hotShift)
  auto toHot = [&](uint32_t st) -> uint32_t {
    if (BIG) return st;
    if (CLS) return st * d.clsRowBytes;
    if (d.hotShift && st < d.nPureDead) return 0u;
    return st - d.hotLo < d.nHot ? st - d.hotLo + d.hotShift : 255u;
  };
  auto toGlobal = [&](uint32_t idx) -> uint32_t {
    if (BIG) return idx;
    if (CLS) return idx / d.clsRowBytes;
    return (d.hotShift && idx == 0) ? 0u : d.hotLo + idx - d.hotShift;
  };
  const uint32_t lineLen = uint32_t(io.stride);
  const uint32_t R = lineLen / BLK;  // blocks per line
  const uint64_t linesPerTile = uint64_t(THREADS) * CH;
  const uint64_t nTiles = (io.n + linesPerTile - 1) / linesPerTile;
  const uint64_t G = gridDim.x;
  if (blockIdx.x >= nTiles) return;
  const uint64_t myTiles = (nTiles - blockIdx.x + G - 1) / G;
  const uint64_t Q = myTiles * R;

  // load cursor (runs one block ahead of the walk cursor)
  uint64_t ldTile = blockIdx.x;
  uint32_t ldR = 0;
  uint64_t ldQ = 0;
  auto issue = [&](BlockRegs<HALVES> (&b)[CH]) {
    // past the end of the work: re-read the last block (keeps requests unconditional)
    const uint64_t t = ldQ < Q ? ldTile : ldTile - (ldR == 0 ? G : 0);
    const uint32_t rr = ldQ < Q ? ldR : (ldR == 0 ? R - 1 : ldR - 1);
    const uint8_t *p[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      uint64_t ln = t * linesPerTile + uint64_t(c) * THREADS + threadIdx.x;
      if (ln >= io.n) ln = io.n - 1;
      p[c] = io.data + ln * lineLen + uint64_t(rr) * BLK;
    }
#pragma unroll
    for (int k = 0; k < 4 * HALVES; ++k) {
#pragma unroll
      for (int c = 0; c < CH; ++c) b[c].p[k] = reinterpret_cast<const uint4 *>(p[c])[k];
    }
    if (MODE == kSmAdvance || MODE == kSmChunk) {
      // unconditional like the data (every block re-requests its line's state; only the first
      // block of a line uses it) so the in-order vmcnt counts stay exact
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        uint64_t ln = t * linesPerTile + uint64_t(c) * THREADS + threadIdx.x;
        if (ln >= io.n) ln = io.n - 1;
        b[c].st = io.state[ln];
      }
    }
    if (ldQ < Q) {
      ++ldQ;
      if (++ldR == R) { ldR = 0; ldTile += G; }
    }
  };

  // table requests first, the first input block's right behind them (kStreamEarlyIssue), the
  // LDS stores and the barrier after: the table is at the head of the CU's memory queues, the
  // input no longer waits for the barrier before it is even requested
  const uint4 *tsrc =
      reinterpret_cast<const uint4 *>(d.table + (HOT ? d.hot8Off : CLS ? d.clsOff : 0u));
  const uint32_t n16 = HOT ? kStreamTabBytes / 16 : CLS ? d.clsBytes / 16 : d.tableBytes / 16;
  // CLS: 256 bytes of eq2 in front of a table of up to 64 KB - one more 16-byte piece per thread
  // 16-byte pieces per thread that cover the table (+ eq2's 256 B in the class-table form)
  constexpr uint32_t kStagePieces =
      BIG ? 1 : (kStreamTabBytes / 16 + (CLS ? 64 : 0) + THREADS - 1) / THREADS;
  uint4 tv[kStagePieces];
  if (BIG) {  // up to 155 KB: straight through, no register staging
    for (uint32_t i = threadIdx.x; i < n16; i += THREADS) reinterpret_cast<uint4 *>(tab)[i] = tsrc[i];
  }
#pragma unroll
  for (uint32_t k = 0; k < kStagePieces; ++k) {
    const uint32_t i = k * THREADS + threadIdx.x;
    tv[k] = (!BIG && i < n16) ? tsrc[i] : make_uint4(0, 0, 0, 0);
  }
  const int32_t myRes = IDXD ? 0 : threadIdx.x < d.nStates ? d.result[threadIdx.x] : 0;
  BlockRegs<HALVES> A[CH], B[CH];
  // (Tried, round 2: holding wave w's first requests back by w x 64..1024 cycles, so that the
  // waves' first blocks arrive one after the other instead of together at the end of the first
  // 32 MB: 25.5-26.5 us against 25.9 us - no effect.)
  if (EARLY) issue(A);
  {
    uint4 *dst = reinterpret_cast<uint4 *>(tab);
#pragma unroll
    for (uint32_t k = 0; k < kStagePieces; ++k) {
      const uint32_t i = k * THREADS + threadIdx.x;
      // (the result words behind a fused / hot table are written below, by other threads)
      if (!BIG && i < (kStreamTabBytes + (CLS ? 1024 : 0)) / 16) dst[i] = tv[k];
    }
    if (!CLS && threadIdx.x < 256) {
      // HOT keeps no results in LDS (its state values are hot indices): the kilobyte holds the
      // byte -> class map instead, for the cold steps of slowHalf()
      if (HOT) reinterpret_cast<uint8_t *>(ldsRes)[threadIdx.x] = d.equivLeader[threadIdx.x];
      else ldsRes[threadIdx.x] = myRes;
    }
  }
  // The byte steps read this table from inline asm only.  Unless its address visibly reaches an
  // asm statement the compiler may treat the array as never read and drop the stores above -
  // it did, in the one mode (kSmChunk) that has no C++ read of `lds` left.
  asm volatile("" : : "v"(tab) : "memory");
  __syncthreads();

  uint32_t s[CH], accS[CH], endv[CH], startv[CH];
  uint32_t g[CH];  // HOT: global id of the lane's state while it is outside the hot set
  uint64_t mA[CH], mB[CH];
  uint64_t tile = blockIdx.x;
  uint32_t r = 0;

  // walk one 128-byte block held in registers: two 64-byte halves, each folded on its own so
  // the block-relative positions stay inline constants (0..63)
  auto walkBlock = [&](const BlockRegs<HALVES> (&blk)[CH]) {
    if (r == 0) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        s[c] = init; accS[c] = 0; endv[c] = 0; startv[c] = 0;
        mA[c] = ~0ull; mB[c] = ~0ull;
        g[c] = kNoState;
        constexpr bool kStateIn = MODE == kSmAdvance || MODE == kSmChunk;
        if (MODE == kSmChunk) startv[c] = kNoState;  // "no such event in this chunk"
        if (kStateIn && !IDXD) s[c] = blk[c].st < d.nStates ? blk[c].st : init;
        if (kStateIn && CLS) s[c] = toHot(blk[c].st < d.nStates ? blk[c].st : d.init);
        if (kStateIn && HOT) {
          const uint32_t st = blk[c].st < d.nStates ? blk[c].st : d.init;
          s[c] = toHot(st);
          g[c] = s[c] != 255u ? kNoState : st;
        } else if (HOT && init == 0x1ffu) {
          s[c] = 255u;
          g[c] = d.init;
        }
        if (MODE == kSmChunk) {
          // the first step's "was in the initial state" is a fact about the ENTRY state here
          mA[c] = __builtin_amdgcn_ballot_w64(s[c] == init);
          mB[c] = mA[c];
        }
      }
    }
#pragma unroll
    for (int h = 0; h < HALVES; ++h) {
      StreamBook b[CH];
      uint32_t s0[CH];  // HOT: the half-block's entry state (hot index), for the re-walk
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        b[c].acc = IDXD ? 0u : accS[c]; b[c].end = 0; b[c].start = 0;
        s0[c] = s[c];
      }
      uint4 piece[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) piece[c] = blk[c].p[4 * h + 0];
      if constexpr (CLS) streamWalk16Cls<MODE, 0, BIG>(piece, s, b, mA, mB, firstAccept, init, tab, d.clsRowBytes);
      else streamWalk16<MODE, 0>(piece, s, b, mA, mB, firstAccept, init);
#pragma unroll
      for (int c = 0; c < CH; ++c) piece[c] = blk[c].p[4 * h + 1];
      if constexpr (CLS) streamWalk16Cls<MODE, 1, BIG>(piece, s, b, mA, mB, firstAccept, init, tab, d.clsRowBytes);
      else streamWalk16<MODE, 1>(piece, s, b, mA, mB, firstAccept, init);
#pragma unroll
      for (int c = 0; c < CH; ++c) piece[c] = blk[c].p[4 * h + 2];
      if constexpr (CLS) streamWalk16Cls<MODE, 2, BIG>(piece, s, b, mA, mB, firstAccept, init, tab, d.clsRowBytes);
      else streamWalk16<MODE, 2>(piece, s, b, mA, mB, firstAccept, init);
#pragma unroll
      for (int c = 0; c < CH; ++c) piece[c] = blk[c].p[4 * h + 3];
      if constexpr (CLS) streamWalk16Cls<MODE, 3, BIG>(piece, s, b, mA, mB, firstAccept, init, tab, d.clsRowBytes);
      else streamWalk16<MODE, 3>(piece, s, b, mA, mB, firstAccept, init);
      const uint32_t off = r * BLK + h * 64;
      bool redo[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) redo[c] = HOT && s[c] == 255u;
      if (HOT && __builtin_amdgcn_ballot_w64(redo[0] || redo[CH - 1])) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          if (redo[c]) {
            const uint64_t ln = tile * linesPerTile + uint64_t(c) * THREADS + threadIdx.x;
            uint32_t st = g[c] != kNoState ? g[c] : toGlobal(s0[c]);
            // lanes past the end of the batch walked a clamped line: nothing of theirs is kept
            if (ln < io.n) {
              const SlowBook o = slowHalf<MODE>(d, tab, reinterpret_cast<const uint8_t *>(ldsRes),
                                                io.data + ln * lineLen + off, off,
                                                SlowBook{st, accS[c], endv[c], startv[c]});
              st = o.st; accS[c] = o.accS; endv[c] = o.endv; startv[c] = o.startv;
            }
            s[c] = toHot(st);
            g[c] = s[c] != 255u ? kNoState : st;
          }
        }
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (HOT && redo[c]) continue;
        if (kAcc && !IDXD) {
          accS[c] = b[c].acc;
          endv[c] = b[c].end ? off + b[c].end : endv[c];
          if (s[c] >= firstAccept) { accS[c] = s[c]; endv[c] = off + 64; }
          // check<..., true> over a forced leader: what accepted up to the end of the leader is
          // not seen (Batch::ignoreAcceptUpTo; 0 otherwise, and an end is never 0)
          if (endv[c] <= io.ignoreAcceptUpTo) { endv[c] = 0; accS[c] = 0; }
        }
        if (kAcc && IDXD) {
          if (b[c].end) { accS[c] = toGlobal(b[c].acc); endv[c] = off + b[c].end; }
          if (s[c] >= firstAccept) { accS[c] = toGlobal(s[c]); endv[c] = off + 64; }
        }
        if (kStart) {
          startv[c] = b[c].start ? off + b[c].start - 1 : startv[c];
          const bool wasInit63 = (mA[c] >> (threadIdx.x & 63)) & 1;
          if (wasInit63 && s[c] != init) startv[c] = off + 63;
        }
      }
    }
    constexpr bool kPlainOut = !IDXD && MODE != kSmAdvance && MODE != kSmChunk;
    if constexpr (kPlainOut) {
      // Results are stored after EVERY block, without a branch: by lanes whose line ends here into
      // the line's slots, by everyone else into the DFA's sink.  Stores under a branch are vm
      // operations the compiler cannot count: the next walk's wait for its (older) input block
      // then has to assume none was issued and ends up sitting out the stores' acknowledgements -
      // once per tile, i.e. on every block of 64-byte lines.
      const bool lineEnd = r + 1 == R;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const uint64_t ln = tile * linesPerTile + uint64_t(c) * THREADS + threadIdx.x;
        const bool report = lineEnd && ln < io.n;
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
        int32_t *pr = report ? io.result + ln : reinterpret_cast<int32_t *>(d.sink);
        uint64_t *pe = report && io.end ? io.end + ln : reinterpret_cast<uint64_t *>(d.sink);
        uint64_t *ps = report && io.start ? io.start + ln : reinterpret_cast<uint64_t *>(d.sink);
        if (R == 1) {
          // one block per line, every store a line's Outcome - written once, not read by this
          // launch: non-temporal (round 3: configs[1]'s single launch 25.4 -> 23.7 us).  Both arms
          // issue the same number of stores, so the waits behind them stay exact.
          __builtin_nontemporal_store(rr, pr);
          __builtin_nontemporal_store(rr ? uint64_t(en) : uint64_t(0), pe);
          if (kStart) __builtin_nontemporal_store(rr ? uint64_t(startv[c]) : uint64_t(0), ps);
        } else {
          // longer lines: most of these stores rewrite the sink, which must stay a cached line
          // (non-temporal, every block's sink stores went out to HBM: 4 KiB lines 4.4 -> 3.1 TB/s)
          *pr = rr;
          *pe = rr ? uint64_t(en) : 0;
          if (kStart) *ps = rr ? uint64_t(startv[c]) : 0;
        }
      }
      if (++r == R) { r = 0; tile += G; }
      return;
    }
    if (++r == R) {
      r = 0;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const uint64_t ln = tile * linesPerTile + uint64_t(c) * THREADS + threadIdx.x;
        if (ln < io.n) {
          int32_t rr;
          uint32_t en;
          const uint32_t sG = !IDXD ? s[c] : g[c] != kNoState ? g[c] : toGlobal(s[c]);
          if (kAcc) {
            rr = endv[c] ? (IDXD ? d.result[accS[c]] : ldsRes[accS[c]]) : 0;
            en = endv[c];
          } else if (IDXD) {
            rr = sG >= d.firstAccept ? d.result[sG] : 0;
            en = lineLen;
          } else {
            rr = s[c] >= firstAccept ? ldsRes[s[c]] : 0;
            en = lineLen;
          }
          if (MODE == kSmChunk) {
            io.state[ln] = sG;
            io.result[ln] = endv[c] ? int32_t(accS[c]) : -1;
            io.end[ln] = endv[c];
            io.start[ln] = uint64_t(startv[c] + 1u);  // kNoState + 1 wraps to 0 = none
            continue;
          }
          io.result[ln] = rr;
          if (MODE == kSmAdvance) io.state[ln] = sG;
          if (io.end) io.end[ln] = rr ? uint64_t(en) : 0;
          if (kStart && io.start) io.start[ln] = rr ? uint64_t(startv[c]) : 0;
        }
      }
      tile += G;
    }
  };

  if (!EARLY) issue(A);
  // The loop is rotated by one block: both ways into its head (from here and around the back
  // edge) then end in "block requested, a block's results stored, block requested", and the
  // compiler's vmcnt for the head's walk - the minimum over the ways in - is exact.  Entered
  // straight after the first request it would have to assume no store was pending and would
  // sit out the previous block's result stores on every iteration.
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

// Few lines (fewer 1024-line tiles than CUs - BASELINE configs[4]: 65,536 x 64 KiB): 512-line
// tiles of 256 threads put the lines on twice as many CUs.  Such a batch is bound by the LDS
// latency of one line's dependent chain, not by throughput: 1.16 -> 1.63 TB/s, where the same
// bytes cut into 2^20 lines run at 3.5 TB/s.
inline bool fewLines(const Batch &b, const LaunchCfg &cfg) {
  return (b.n + uint64_t(kStreamThreads) * kStreamChains - 1) /
             (uint64_t(kStreamThreads) * kStreamChains) < uint64_t(cfg.numCUs);
}

// the HOT / CLS forms of the same launch
template <int MODE, int TABK, int THREADS>
hipError_t launchStreamHotT(const DevDfa &d, const Batch &b, const LaunchCfg &cfg,
                            hipStream_t stream) {
  const uint64_t linesPerTile = uint64_t(THREADS) * kStreamChains;
  const uint64_t tiles = (b.n + linesPerTile - 1) / linesPerTile;
  const uint64_t blocks = tiles < uint64_t(cfg.numCUs) ? tiles : uint64_t(cfg.numCUs);
  if (b.stride % 128 == 0)
    hipLaunchKernelGGL((k_stream<MODE, 2, THREADS, TABK>), dim3(uint32_t(blocks)), dim3(THREADS),
                       0, stream, d, b);
  else
    hipLaunchKernelGGL((k_stream<MODE, 1, THREADS, TABK>), dim3(uint32_t(blocks)), dim3(THREADS),
                       0, stream, d, b);
  return hipGetLastError();
}

template <int MODE, int TABK = kTabHot>
hipError_t launchStreamHot(const DevDfa &d, const Batch &b, const LaunchCfg &cfg,
                           hipStream_t stream) {
  if (fewLines(b, cfg)) return launchStreamHotT<MODE, TABK, 256>(d, b, cfg, stream);
  return launchStreamHotT<MODE, TABK, kStreamThreads>(d, b, cfg, stream);
}

template <int MODE, int THREADS>
hipError_t launchStreamTT(const DevDfa &d, const Batch &b, const LaunchCfg &cfg,
                          hipStream_t stream) {
  const uint64_t linesPerTile = uint64_t(THREADS) * kStreamChains;
  const uint64_t tiles = (b.n + linesPerTile - 1) / linesPerTile;
  // one workgroup per CU: two (32 chains per CU) measured 26.0 us against 24.7 us on configs[1]
  const uint64_t blocks = tiles < uint64_t(cfg.numCUs) ? tiles : uint64_t(cfg.numCUs);
  if (b.stride % 128 == 0)
    hipLaunchKernelGGL((k_stream<MODE, 2, THREADS>), dim3(uint32_t(blocks)), dim3(THREADS), 0,
                       stream, d, b);
  else
    hipLaunchKernelGGL((k_stream<MODE, 1, THREADS>), dim3(uint32_t(blocks)), dim3(THREADS), 0,
                       stream, d, b);
  return hipGetLastError();
}

template <int MODE>
hipError_t launchStreamT(const DevDfa &d, const Batch &b, const LaunchCfg &cfg,
                         hipStream_t stream) {
  // (a 1024-thread form existed behind REDGPU_STREAM_THREADS in round 1 and lost at every shape
  // but 2^24 x 64 B: 26.8 against 24.7 us on configs[1], 3.45 against 4.25 TB/s on 4 KiB lines)
  if (fewLines(b, cfg)) return launchStreamTT<MODE, 256>(d, b, cfg, stream);
  return launchStreamTT<MODE, 512>(d, b, cfg, stream);
}
