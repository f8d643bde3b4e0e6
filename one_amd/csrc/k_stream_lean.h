// k_stream_lean.h - the byte step of the streaming walk with its bookkeeping deferred (included
// by kernels.hip between k_stream.h and k_stream_multi.h, inside its namespace).
//
// k_stream's step keeps the Outcome exact at every byte: 2 v_cmp + 3 v_cndmask per byte and
// line on top of the lookup (v_perm + ds_read_u8), 6 VALU in all, and a workgroup of 8 waves
// issues nothing else - URI-D on 4 KiB lines runs at 4.4 TB/s where HBM gives a pure read
// 6.4 TB/s and the LDS gathers would allow 8.  matchCore's bookkeeping
// (include/Matcher.h:443-479) only ever asks two questions of a line - where is the LAST
// accepting state, where did the walk LAST leave the initial state - so per byte it is enough
// to know IN WHICH 16-byte piece the answer lies:
//
//   accept   the address of a lookup is (state << 8) | byte, and accepting states are the top
//            of the numbering: the running v_max3_u32 over two consecutive addresses (0.5 VALU
//            per byte) is >= firstAccept << 8 iff some state of the window accepts;
//   leave    one v_cmp_eq per byte gives the lane mask "state == initial"; "was & ~is",
//            OR-ed over the piece, stays in SGPRs (2 SALU per byte);
//   per piece (once per 16 bytes): v_cmp of the window's maximum, two v_cndmask that remember
//            the piece - its number and the state it was ENTERED in, one word - as the last
//            piece with an accept / with a leave.
//
// 2.75 VALU per byte and line instead of 6.  At the end of the line the two remembered pieces
// are read again (16 bytes each) and walked from their entry states with the exact step of
// k_stream.h (streamWalk16): 32 extra steps per line, which is why this form is for LONG lines
// (>= kLeanMinLine bytes); short ones keep the exact step.  Positions: a_k = address of the
// lookup of byte k = (state after k bytes << 8) | byte k; the accept window of piece q is
// a_{16q+1} .. a_{16q+16} (what a re-walk of bytes 16q .. 16q+15 can report as an end), its
// leave window the events at bytes 16q .. 16q+15; both close in the FIRST step of piece q + 1,
// which knows a_{16q+16} and whether state 16q+16 is the initial one, and that step's state is
// piece q + 1's entry state.  Carried across blocks and pieces: the state, a_{k-1}, the running
// maximum, the open piece's key, the two records, "state k-1 was initial", the leave mask.
#pragma once

constexpr uint32_t kLeanMinLine = 512;

struct LeanRegs {
  uint32_t sX[2];    // state after k bytes (k even between steps)
  uint32_t aO[2];    // address of the previous (odd) lookup
  uint32_t mx[2];    // running maximum of the open accept window
  uint32_t key[2];   // open piece: (piece number + 1) << 8 | entry state
  uint32_t recA[2];  // key of the last piece whose window held an accepting state, 0 = none
  uint32_t recS[2];  // key of the last piece in which the walk left the initial state, 0 = none
  uint64_t isO[2];   // lane mask: the state before the current (even) one was the initial state
  uint64_t any[2];   // lane mask: a leave event in the open piece so far
  uint32_t pbase;    // (open piece number + 1) << 8, uniform
};

#define LN_PERM(A, S, c) "v_perm_b32 %[" A #c "], %[" S #c "], %[w" #c "], %[sel]\n\t"
#define LN_READ(D, A, c) "ds_read_u8 %[" D #c "], %[" A #c "]\n\t"
#define LN_CMPI(M, S, c) "v_cmp_eq_u32_e64 %[" M #c "], %[init], %[" S #c "]\n\t"
#define LN_MAX3(c) "v_max3_u32 %[mx" #c "], %[mx" #c "], %[aO" #c "], %[aE" #c "]\n\t"
#define LN_MAX2(c) "v_max_u32_e32 %[mx" #c "], %[aO" #c "], %[aE" #c "]\n\t"
#define LN_LEAVE(WAS, IS, c) "s_andn2_b64 %[l" #c "], %[" WAS #c "], %[" IS #c "]\n\t"
#define LN_ANY(c) "s_or_b64 %[any" #c "], %[any" #c "], %[l" #c "]\n\t"
// the statement's closing wait, assembled only when operand [wt] is 1 (statements of the first
// of two chain groups leave their lookups in flight for the second group's statement to wait on)
#define LN_WAITIF ".if %[wt]\n\ts_waitcnt lgkmcnt(0)\n\t.endif"
// ... and its opening one: with two chain groups every step statement first waits for ITS OWN
// group's two lookups of the step before - issued one statement of the other group ago, so two
// younger lookups may stay in flight: s_waitcnt lgkmcnt(2) ([pre] = 2; -1 = no opening wait)
#define LN_PREIF ".if %[pre] >= 0\n\ts_waitcnt lgkmcnt(%[pre])\n\t.endif\n\t"

// even step k: state in sX, next state lands in sY.  FIRSTMAX: k % 16 == 2, the accept
// window's first pair (plain maximum instead of the running one).
// WAIT = false only for the step that opens a piece: its wait closes leanFlush's statement.
// (The wait sits INSIDE the step's statement: behind a statement of its own the compiler put an
// s_nop in front of every step - an issue slot per byte.)
template <bool kAcc, bool kStart, bool FIRSTMAX, bool WAIT = true, int PRE = -1>
__device__ __forceinline__ void leanEven(LeanRegs &L, uint32_t (&sY)[2], uint32_t (&aE)[2],
                                         uint64_t (&isE)[2], const uint32_t (&w)[2],
                                         uint32_t sel, uint32_t init) {
  uint64_t l[2];
#define LN_W LN_WAITIF
  constexpr int kWt = WAIT ? 1 : 0;
  if constexpr (kAcc && kStart) {
    if constexpr (FIRSTMAX)
      asm volatile(LN_PREIF LN_PERM("aE", "sX", 0) LN_PERM("aE", "sX", 1) LN_READ("sY", "aE", 0) LN_READ("sY", "aE", 1)
                   LN_CMPI("isE", "sX", 0) LN_CMPI("isE", "sX", 1) LN_MAX2(0) LN_MAX2(1)
                   LN_LEAVE("isO", "isE", 0) LN_LEAVE("isO", "isE", 1) LN_ANY(0) LN_ANY(1) LN_W
                   : [aE0] "=&v"(aE[0]), [aE1] "=&v"(aE[1]), [sY0] "=&v"(sY[0]), [sY1] "=&v"(sY[1]),
                     [isE0] "=&s"(isE[0]), [isE1] "=&s"(isE[1]), [l0] "=&s"(l[0]), [l1] "=&s"(l[1]),
                     [mx0] "=&v"(L.mx[0]), [mx1] "=&v"(L.mx[1]), [any0] "+s"(L.any[0]), [any1] "+s"(L.any[1])
                   : [sX0] "v"(L.sX[0]), [sX1] "v"(L.sX[1]), [w0] "v"(w[0]), [w1] "v"(w[1]),
                     [aO0] "v"(L.aO[0]), [aO1] "v"(L.aO[1]), [isO0] "s"(L.isO[0]), [isO1] "s"(L.isO[1]),
                     [init] "s"(init), [sel] "s"(sel), [wt] "n"(kWt), [pre] "n"(PRE)
                   : "memory", "scc");
    else
      asm volatile(LN_PREIF LN_PERM("aE", "sX", 0) LN_PERM("aE", "sX", 1) LN_READ("sY", "aE", 0) LN_READ("sY", "aE", 1)
                   LN_CMPI("isE", "sX", 0) LN_CMPI("isE", "sX", 1) LN_MAX3(0) LN_MAX3(1)
                   LN_LEAVE("isO", "isE", 0) LN_LEAVE("isO", "isE", 1) LN_ANY(0) LN_ANY(1) LN_W
                   : [aE0] "=&v"(aE[0]), [aE1] "=&v"(aE[1]), [sY0] "=&v"(sY[0]), [sY1] "=&v"(sY[1]),
                     [isE0] "=&s"(isE[0]), [isE1] "=&s"(isE[1]), [l0] "=&s"(l[0]), [l1] "=&s"(l[1]),
                     [mx0] "+v"(L.mx[0]), [mx1] "+v"(L.mx[1]), [any0] "+s"(L.any[0]), [any1] "+s"(L.any[1])
                   : [sX0] "v"(L.sX[0]), [sX1] "v"(L.sX[1]), [w0] "v"(w[0]), [w1] "v"(w[1]),
                     [aO0] "v"(L.aO[0]), [aO1] "v"(L.aO[1]), [isO0] "s"(L.isO[0]), [isO1] "s"(L.isO[1]),
                     [init] "s"(init), [sel] "s"(sel), [wt] "n"(kWt), [pre] "n"(PRE)
                   : "memory", "scc");
  } else if constexpr (kAcc) {
    if constexpr (FIRSTMAX)
      asm volatile(LN_PREIF LN_PERM("aE", "sX", 0) LN_PERM("aE", "sX", 1) LN_READ("sY", "aE", 0) LN_READ("sY", "aE", 1)
                   LN_MAX2(0) LN_MAX2(1) LN_W
                   : [aE0] "=&v"(aE[0]), [aE1] "=&v"(aE[1]), [sY0] "=&v"(sY[0]), [sY1] "=&v"(sY[1]),
                     [mx0] "=&v"(L.mx[0]), [mx1] "=&v"(L.mx[1])
                   : [sX0] "v"(L.sX[0]), [sX1] "v"(L.sX[1]), [w0] "v"(w[0]), [w1] "v"(w[1]),
                     [aO0] "v"(L.aO[0]), [aO1] "v"(L.aO[1]), [sel] "s"(sel), [wt] "n"(kWt), [pre] "n"(PRE)
                   : "memory");
    else
      asm volatile(LN_PREIF LN_PERM("aE", "sX", 0) LN_PERM("aE", "sX", 1) LN_READ("sY", "aE", 0) LN_READ("sY", "aE", 1)
                   LN_MAX3(0) LN_MAX3(1) LN_W
                   : [aE0] "=&v"(aE[0]), [aE1] "=&v"(aE[1]), [sY0] "=&v"(sY[0]), [sY1] "=&v"(sY[1]),
                     [mx0] "+v"(L.mx[0]), [mx1] "+v"(L.mx[1])
                   : [sX0] "v"(L.sX[0]), [sX1] "v"(L.sX[1]), [w0] "v"(w[0]), [w1] "v"(w[1]),
                     [aO0] "v"(L.aO[0]), [aO1] "v"(L.aO[1]), [sel] "s"(sel), [wt] "n"(kWt), [pre] "n"(PRE)
                   : "memory");
  } else {
    asm volatile(LN_PREIF LN_PERM("aE", "sX", 0) LN_PERM("aE", "sX", 1) LN_READ("sY", "aE", 0) LN_READ("sY", "aE", 1)
                 LN_CMPI("isE", "sX", 0) LN_CMPI("isE", "sX", 1) "s_nop 0\n\t"
                 LN_LEAVE("isO", "isE", 0) LN_LEAVE("isO", "isE", 1) LN_ANY(0) LN_ANY(1) LN_W
                 : [aE0] "=&v"(aE[0]), [aE1] "=&v"(aE[1]), [sY0] "=&v"(sY[0]), [sY1] "=&v"(sY[1]),
                   [isE0] "=&s"(isE[0]), [isE1] "=&s"(isE[1]), [l0] "=&s"(l[0]), [l1] "=&s"(l[1]),
                   [any0] "+s"(L.any[0]), [any1] "+s"(L.any[1])
                 : [sX0] "v"(L.sX[0]), [sX1] "v"(L.sX[1]), [w0] "v"(w[0]), [w1] "v"(w[1]),
                   [isO0] "s"(L.isO[0]), [isO1] "s"(L.isO[1]), [init] "s"(init), [sel] "s"(sel), [wt] "n"(kWt), [pre] "n"(PRE)
                 : "memory", "scc");
  }
}
#undef LN_W

// odd step k + 1: state in sY, next state lands in sX; its address stays in aO for the next
// even step's maximum
template <bool kAcc, bool kStart, bool WAIT = true, int PRE = -1>
__device__ __forceinline__ void leanOdd(LeanRegs &L, const uint32_t (&sY)[2],
                                        const uint64_t (&isE)[2], const uint32_t (&w)[2],
                                        uint32_t sel, uint32_t init) {
  uint64_t l[2];
  constexpr int kWt = WAIT ? 1 : 0;
  if constexpr (kStart) {
    asm volatile(LN_PREIF LN_PERM("aO", "sY", 0) LN_PERM("aO", "sY", 1) LN_READ("sX", "aO", 0) LN_READ("sX", "aO", 1)
                 LN_CMPI("isO", "sY", 0) LN_CMPI("isO", "sY", 1) "s_nop 0\n\t"
                 LN_LEAVE("isE", "isO", 0) LN_LEAVE("isE", "isO", 1) LN_ANY(0) LN_ANY(1) LN_WAITIF
                 : [aO0] "=&v"(L.aO[0]), [aO1] "=&v"(L.aO[1]), [sX0] "=&v"(L.sX[0]), [sX1] "=&v"(L.sX[1]),
                   [isO0] "=&s"(L.isO[0]), [isO1] "=&s"(L.isO[1]), [l0] "=&s"(l[0]), [l1] "=&s"(l[1]),
                   [any0] "+s"(L.any[0]), [any1] "+s"(L.any[1])
                 : [sY0] "v"(sY[0]), [sY1] "v"(sY[1]), [w0] "v"(w[0]), [w1] "v"(w[1]),
                   [isE0] "s"(isE[0]), [isE1] "s"(isE[1]), [init] "s"(init), [sel] "s"(sel), [wt] "n"(kWt), [pre] "n"(PRE)
                 : "memory", "scc");
  } else {
    asm volatile(LN_PREIF LN_PERM("aO", "sY", 0) LN_PERM("aO", "sY", 1) LN_READ("sX", "aO", 0) LN_READ("sX", "aO", 1) LN_WAITIF
                 : [aO0] "=&v"(L.aO[0]), [aO1] "=&v"(L.aO[1]), [sX0] "=&v"(L.sX[0]), [sX1] "=&v"(L.sX[1])
                 : [sY0] "v"(sY[0]), [sY1] "v"(sY[1]), [w0] "v"(w[0]), [w1] "v"(w[1]), [sel] "s"(sel), [wt] "n"(kWt), [pre] "n"(PRE)
                 : "memory");
  }
}

// behind the even step that opens a piece (k % 16 == 0), under its lookups' latency: the
// windows of the piece before are complete - record it where they fired, open the new piece;
// ends with that step's wait
template <bool kAcc, bool kStart, bool WAIT = true>
__device__ __forceinline__ void leanFlush(LeanRegs &L, uint32_t T8) {
  uint64_t am[2];
  constexpr int kWt = WAIT ? 1 : 0;
  if constexpr (kAcc && kStart) {
    asm volatile("v_cmp_le_u32_e64 %[am0], %[T8], %[mx0]\n\t"
                 "v_cmp_le_u32_e64 %[am1], %[T8], %[mx1]\n\t"
                 "v_cndmask_b32_e64 %[rS0], %[rS0], %[key0], %[any0]\n\t"
                 "v_cndmask_b32_e64 %[rS1], %[rS1], %[key1], %[any1]\n\t"
                 "v_cndmask_b32_e64 %[rA0], %[rA0], %[key0], %[am0]\n\t"
                 "v_cndmask_b32_e64 %[rA1], %[rA1], %[key1], %[am1]\n\t"
                 "s_mov_b64 %[any0], 0\n\t"
                 "s_mov_b64 %[any1], 0\n\t"
                 "v_or_b32_e32 %[key0], %[pb], %[sX0]\n\t"
                 "v_or_b32_e32 %[key1], %[pb], %[sX1]\n\t"
                 "s_add_u32 %[pb], %[pb], 0x100\n\t" LN_WAITIF
                 : [am0] "=&s"(am[0]), [am1] "=&s"(am[1]), [rS0] "+v"(L.recS[0]), [rS1] "+v"(L.recS[1]),
                   [rA0] "+v"(L.recA[0]), [rA1] "+v"(L.recA[1]), [key0] "+v"(L.key[0]),
                   [key1] "+v"(L.key[1]), [any0] "+s"(L.any[0]), [any1] "+s"(L.any[1]), [pb] "+s"(L.pbase)
                 : [mx0] "v"(L.mx[0]), [mx1] "v"(L.mx[1]), [sX0] "v"(L.sX[0]), [sX1] "v"(L.sX[1]),
                   [T8] "s"(T8), [wt] "n"(kWt)
                 : "memory", "scc");
  } else if constexpr (kAcc) {
    asm volatile("v_cmp_le_u32_e64 %[am0], %[T8], %[mx0]\n\t"
                 "v_cmp_le_u32_e64 %[am1], %[T8], %[mx1]\n\t"
                 "s_nop 1\n\t"
                 "v_cndmask_b32_e64 %[rA0], %[rA0], %[key0], %[am0]\n\t"
                 "v_cndmask_b32_e64 %[rA1], %[rA1], %[key1], %[am1]\n\t"
                 "v_or_b32_e32 %[key0], %[pb], %[sX0]\n\t"
                 "v_or_b32_e32 %[key1], %[pb], %[sX1]\n\t"
                 "s_add_u32 %[pb], %[pb], 0x100\n\t" LN_WAITIF
                 : [am0] "=&s"(am[0]), [am1] "=&s"(am[1]), [rA0] "+v"(L.recA[0]), [rA1] "+v"(L.recA[1]),
                   [key0] "+v"(L.key[0]), [key1] "+v"(L.key[1]), [pb] "+s"(L.pbase)
                 : [mx0] "v"(L.mx[0]), [mx1] "v"(L.mx[1]), [sX0] "v"(L.sX[0]), [sX1] "v"(L.sX[1]),
                   [T8] "s"(T8), [wt] "n"(kWt)
                 : "memory", "scc");
  } else {
    asm volatile("v_cndmask_b32_e64 %[rS0], %[rS0], %[key0], %[any0]\n\t"
                 "v_cndmask_b32_e64 %[rS1], %[rS1], %[key1], %[any1]\n\t"
                 "s_mov_b64 %[any0], 0\n\t"
                 "s_mov_b64 %[any1], 0\n\t"
                 "v_or_b32_e32 %[key0], %[pb], %[sX0]\n\t"
                 "v_or_b32_e32 %[key1], %[pb], %[sX1]\n\t"
                 "s_add_u32 %[pb], %[pb], 0x100\n\t" LN_WAITIF
                 : [rS0] "+v"(L.recS[0]), [rS1] "+v"(L.recS[1]), [key0] "+v"(L.key[0]),
                   [key1] "+v"(L.key[1]), [any0] "+s"(L.any[0]), [any1] "+s"(L.any[1]), [pb] "+s"(L.pbase)
                 : [sX0] "v"(L.sX[0]), [sX1] "v"(L.sX[1]), [wt] "n"(kWt)
                 : "memory", "scc");
  }
}

// 16 bytes (one piece) of every chain: GR groups of two chains.  With two groups the statements
// alternate between them and none waits at its end: each opens with s_waitcnt lgkmcnt(2) - its
// own group's lookups of the step before are done, the other group's two (issued since) may
// still be in flight - so the groups run half a step apart and a wave keeps four lookups in
// flight.  Between the statements nothing else may touch lgkmcnt: every statement clobbers
// "memory" (no load is moved in between) and the caller drains the counter before the first and
// after the last statement of a block (leanDrain).
template <bool kAcc, bool kStart, int GR>
__device__ __forceinline__ void leanWalk16(const uint4 (&piece)[2 * GR], LeanRegs (&L)[GR],
                                           uint32_t T8, uint32_t init) {
  uint32_t w[GR][2], sY[GR][2], aE[GR][2];
  uint64_t isE[GR][2];
#define LN_STEP(G, CALL) if constexpr (GR > 1) { CALL(G, false, 2) } else { CALL(G, true, -1) }
#define LN_E0F(G, W, P) leanEven<kAcc, kStart, false, false, P>(L[G], sY[G], aE[G], isE[G], w[G], 0x0c0c0400u, init); \
                        leanFlush<kAcc, kStart, W>(L[G], T8);
#define LN_E0(G, W, P) leanEven<kAcc, kStart, false, W, P>(L[G], sY[G], aE[G], isE[G], w[G], 0x0c0c0400u, init);
#define LN_O1(G, W, P) leanOdd<kAcc, kStart, W, P>(L[G], sY[G], isE[G], w[G], 0x0c0c0401u, init);
#define LN_E2F(G, W, P) leanEven<kAcc, kStart, true, W, P>(L[G], sY[G], aE[G], isE[G], w[G], 0x0c0c0402u, init);
#define LN_E2(G, W, P) leanEven<kAcc, kStart, false, W, P>(L[G], sY[G], aE[G], isE[G], w[G], 0x0c0c0402u, init);
#define LN_O3(G, W, P) leanOdd<kAcc, kStart, W, P>(L[G], sY[G], isE[G], w[G], 0x0c0c0403u, init);
#define LN_ALL(CALL) LN_STEP(0, CALL) if constexpr (GR > 1) { LN_STEP(GR - 1, CALL) }
#define LN_WORD(FIELD, FIRST)                                                    \
  _Pragma("unroll") for (int g = 0; g < GR; ++g) {                               \
    w[g][0] = piece[2 * g].FIELD;                                                \
    w[g][1] = piece[2 * g + 1].FIELD;                                            \
  }                                                                              \
  if (FIRST) { LN_ALL(LN_E0F) } else { LN_ALL(LN_E0) }                           \
  LN_ALL(LN_O1)                                                                  \
  if (FIRST) { LN_ALL(LN_E2F) } else { LN_ALL(LN_E2) }                           \
  LN_ALL(LN_O3)
  LN_WORD(x, true) LN_WORD(y, false) LN_WORD(z, false) LN_WORD(w, false)
#undef LN_WORD
#undef LN_ALL
#undef LN_O3
#undef LN_E2
#undef LN_E2F
#undef LN_O1
#undef LN_E0
#undef LN_E0F
#undef LN_STEP
}

__device__ __forceinline__ void leanDrain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// the lane masks and the piece counter are carried around the block loop in SGPRs and feed "s"
// asm operands: left alone the compiler may park them in VGPRs between blocks (the asm then does
// not assemble), so every block pins them again (no instruction where they already are scalar)
__device__ __forceinline__ uint64_t leanPin64(uint64_t v) {
  const uint32_t lo = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v))));
  const uint32_t hi = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v >> 32))));
  return (uint64_t(hi) << 32) | lo;
}
__device__ __forceinline__ void leanPin(LeanRegs &L) {
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    L.isO[c] = leanPin64(L.isO[c]);
    L.any[c] = leanPin64(L.any[c]);
  }
  L.pbase = uint32_t(__builtin_amdgcn_readfirstlane(int(L.pbase)));
}

// a line begins: nothing recorded, piece 0 opens with the first step's flush (the "piece before"
// has key 0 = none, so whatever its windows say is recorded as nothing)
__device__ __forceinline__ void leanBegin(LeanRegs &L, uint32_t init) {
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    L.sX[c] = init; L.aO[c] = 0; L.mx[c] = 0; L.key[c] = 0; L.recA[c] = 0; L.recS[c] = 0;
    L.isO[c] = ~0ull; L.any[c] = 0;
  }
  L.pbase = 0x100;
}

// a line ends after its last byte (state L.sX): close the last piece's windows with what the
// step behind the line would have seen - the final state's own "address" and whether it is the
// initial state
__device__ __forceinline__ void leanEnd(LeanRegs &L, uint32_t T8, uint32_t init, bool kAcc,
                                        bool kStart) {
  const uint32_t lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if (kAcc) {
      const uint32_t aL = L.sX[c] << 8;
      uint32_t mm = L.mx[c] > L.aO[c] ? L.mx[c] : L.aO[c];
      mm = mm > aL ? mm : aL;
      if (mm >= T8) L.recA[c] = L.key[c];
    }
    if (kStart) {
      const bool wasI = (L.isO[c] >> lane) & 1;
      const bool ev = ((L.any[c] >> lane) & 1) || (wasI && L.sX[c] != init);
      if (ev) L.recS[c] = L.key[c];
    }
  }
}
