// k_ragged.h - the streaming kernel for RAGGED lines (offsets[n+1]); included by kernels.hip
// inside its namespace, after k_stream.h (shares StreamMode / StreamBook).
//
// Same machinery as k_stream - fused u8 table at LDS offset 0, 2 lines per lane, whole 64-byte
// blocks per lane in ping-pong register sets, inline-asm byte step - plus what variable line
// lengths need:
//  * a lane's line may end anywhere: `rem` = bytes of the line left at the start of the block
//    (0..64).  Step IDX is VALID iff IDX < rem; an invalid step still issues its lookup
//    (uniform instruction stream) but the state is not advanced (one v_cndmask after the
//    wait), so a finished lane simply holds its final state.  A held state cannot fake a
//    "left the initial state" event (is-initial == was-initial) and re-recording it as the
//    last accepting state is harmless; only the END position needs the mask of the previous
//    step's validity.  8 VALU + 2 SALU per byte per line.
//  * a lane whose line ends takes the next line of its workgroup's range at the next block
//    boundary (see k_ragged below): no lane waits for the longest line of its wave.
//  * lines start at arbitrary byte offsets: 16-byte loads at unaligned addresses (the memory
//    pipeline splits them).  Block reads run up to 63 bytes past a line's end: a block that
//    would reach past the end of the input buffer is read from `pad` instead - a 192-byte
//    copy of the buffer's last 128 bytes followed by zeros, made by k_tail_pad just before.
//    (An earlier form left those lines to a one-lane generic walk: 25+ us of pure latency
//    for a single 256-byte line, serialised behind this kernel.)  Requests stay unconditional,
//    so the compiler's in-order vmcnt counts stay exact.
//  * an empty line reports the initial state's result (Matcher.h:379,437 with no iterations).
#pragma once

#define RG_PERM(c) "v_perm_b32 %[a" #c "], %[s" #c "], %[w" #c "], %[sel]\n\t"
#define RG_READ(c) "ds_read_u8 %[t" #c "], %[a" #c "]\n\t"
#define RG_VALID(c) "v_cmp_gt_u32_e64 %[v" #c "], %[rem" #c "], %[idx]\n\t"
#define RG_CMPA(c) "v_cmp_le_u32_e64 %[m" #c "], %[T], %[s" #c "]\n\t"
#define RG_CMPI(c) "v_cmp_eq_u32_e64 %[i" #c "], %[init], %[s" #c "]\n\t"
#define RG_ACC(c) "v_cndmask_b32_e64 %[acc" #c "], %[acc" #c "], %[s" #c "], %[m" #c "]\n\t"
#define RG_MEND(c) "s_and_b64 %[mr" #c "], %[m" #c "], %[rp" #c "]\n\t"
#define RG_END(c) "v_cndmask_b32_e64 %[e" #c "], %[e" #c "], %[idx], %[mr" #c "]\n\t"
#define RG_LEAVE(c) "s_andn2_b64 %[l" #c "], %[was" #c "], %[i" #c "]\n\t"
#define RG_START(c) "v_cndmask_b32_e64 %[st" #c "], %[st" #c "], %[idx], %[l" #c "]\n\t"
#define RG_WAIT "s_waitcnt lgkmcnt(0)\n\t"
#define RG_HOLD(c) "v_cndmask_b32_e64 %[t" #c "], %[s" #c "], %[t" #c "], %[v" #c "]\n\t"

#define RG_O_CHAIN(c) [a##c] "=&v"(a[c]), [t##c] "=&v"(t[c]), [v##c] "=&s"(validNow[c])
#define RG_O_ACC(c) [m##c] "=&s"(m[c]), [mr##c] "=&s"(mr[c]), [acc##c] "+v"(b[c].acc), [e##c] "+v"(b[c].end)
#define RG_O_START(c) [i##c] "=&s"(isI[c]), [l##c] "=&s"(l[c]), [st##c] "+v"(b[c].start)
#define RG_I_CHAIN(c) [s##c] "v"(s[c]), [w##c] "v"(w[c]), [rem##c] "v"(rem[c])
#define RG_I_ACC(c) [rp##c] "s"(validPrev[c])
#define RG_I_START(c) [was##c] "s"(wasI[c])

// validPrev: lane mask "the state being book-kept was reached by a real transition" (= the
// previous step was valid); validNow: written here for the next step.
template <int MODE, int IDX>
__device__ __forceinline__ void raggedStep(uint32_t (&s)[2], const uint32_t (&w)[2],
                                           const uint32_t (&rem)[2], StreamBook (&b)[2],
                                           const uint64_t (&wasI)[2], uint64_t (&isI)[2],
                                           const uint64_t (&validPrev)[2], uint64_t (&validNow)[2],
                                           uint32_t sel, uint32_t T, uint32_t init) {
  uint32_t a[2], t[2];
  uint64_t m[2], mr[2], l[2];
  if constexpr (MODE == kSmLastStartEnd) {
    asm volatile(RG_PERM(0) RG_PERM(1) RG_READ(0) RG_READ(1) RG_VALID(0) RG_VALID(1)
                 RG_CMPA(0) RG_CMPA(1) RG_CMPI(0) RG_CMPI(1) RG_ACC(0) RG_ACC(1)
                 RG_MEND(0) RG_MEND(1) RG_END(0) RG_END(1)
                 RG_LEAVE(0) RG_LEAVE(1) RG_START(0) RG_START(1) RG_WAIT RG_HOLD(0) RG_HOLD(1)
                 : RG_O_CHAIN(0), RG_O_CHAIN(1), RG_O_ACC(0), RG_O_ACC(1), RG_O_START(0),
                   RG_O_START(1)
                 : RG_I_CHAIN(0), RG_I_CHAIN(1), RG_I_ACC(0), RG_I_ACC(1), RG_I_START(0),
                   RG_I_START(1), [sel] "s"(sel), [T] "s"(T), [init] "s"(init), [idx] "n"(IDX)
                 : "memory", "scc");
  } else if constexpr (MODE == kSmLastEnd) {
    asm volatile(RG_PERM(0) RG_PERM(1) RG_READ(0) RG_READ(1) RG_VALID(0) RG_VALID(1)
                 RG_CMPA(0) RG_CMPA(1) "s_nop 0\n\t" RG_ACC(0) RG_ACC(1)
                 RG_MEND(0) RG_MEND(1) RG_END(0) RG_END(1) RG_WAIT RG_HOLD(0) RG_HOLD(1)
                 : RG_O_CHAIN(0), RG_O_CHAIN(1), RG_O_ACC(0), RG_O_ACC(1)
                 : RG_I_CHAIN(0), RG_I_CHAIN(1), RG_I_ACC(0), RG_I_ACC(1), [sel] "s"(sel),
                   [T] "s"(T), [idx] "n"(IDX)
                 : "memory", "scc");
  } else if constexpr (MODE == kSmFullStart) {
    asm volatile(RG_PERM(0) RG_PERM(1) RG_READ(0) RG_READ(1) RG_VALID(0) RG_VALID(1)
                 RG_CMPI(0) RG_CMPI(1) "s_nop 0\n\t" RG_LEAVE(0) RG_LEAVE(1)
                 RG_START(0) RG_START(1) RG_WAIT RG_HOLD(0) RG_HOLD(1)
                 : RG_O_CHAIN(0), RG_O_CHAIN(1), RG_O_START(0), RG_O_START(1)
                 : RG_I_CHAIN(0), RG_I_CHAIN(1), RG_I_START(0), RG_I_START(1), [sel] "s"(sel),
                   [init] "s"(init), [idx] "n"(IDX)
                 : "memory", "scc");
  } else {
    asm volatile(RG_PERM(0) RG_PERM(1) RG_READ(0) RG_READ(1) RG_VALID(0) RG_VALID(1)
                 RG_WAIT "s_nop 0\n\t" RG_HOLD(0) RG_HOLD(1)
                 : RG_O_CHAIN(0), RG_O_CHAIN(1)
                 : RG_I_CHAIN(0), RG_I_CHAIN(1), [sel] "s"(sel), [idx] "n"(IDX)
                 : "memory");
  }
  s[0] = t[0];
  s[1] = t[1];
}

template <int MODE, int Q>
__device__ __forceinline__ void raggedWalk16(const uint4 (&piece)[2], uint32_t (&s)[2],
                                             const uint32_t (&rem)[2], StreamBook (&b)[2],
                                             uint64_t (&mA)[2], uint64_t (&mB)[2],
                                             uint64_t (&vA)[2], uint64_t (&vB)[2], uint32_t T,
                                             uint32_t init) {
  uint32_t w[2];
#define RG_WORD(K, FIELD)                                                                     \
  w[0] = piece[0].FIELD;                                                                      \
  w[1] = piece[1].FIELD;                                                                      \
  raggedStep<MODE, 16 * Q + 4 * K + 0>(s, w, rem, b, mA, mB, vA, vB, 0x0c0c0400u, T, init);   \
  raggedStep<MODE, 16 * Q + 4 * K + 1>(s, w, rem, b, mB, mA, vB, vA, 0x0c0c0401u, T, init);   \
  raggedStep<MODE, 16 * Q + 4 * K + 2>(s, w, rem, b, mA, mB, vA, vB, 0x0c0c0402u, T, init);   \
  raggedStep<MODE, 16 * Q + 4 * K + 3>(s, w, rem, b, mB, mA, vB, vA, 0x0c0c0403u, T, init);
  RG_WORD(0, x) RG_WORD(1, y) RG_WORD(2, z) RG_WORD(3, w)
#undef RG_WORD
}

// class-table form of the step (k_stream.h, TABK == kTabCls): row + 2 x class, ds_read_u16
#define RGC_I_CHAIN(c) [s##c] "v"(s[c]), [k##c] "v"(cls[c]), [rem##c] "v"(rem[c])

// the index form (tables above 64 KB) computes state x rowBytes + 2 x class first (RC_MAD, its
// own statement) and enters the shared statement with that as the "class" operand and a plain
// move in place of the add
#define RGC_MOV(c) "v_mov_b32 %[a" #c "], %[k" #c "]\n\t"
#define RGC_STEP(ADDR0, ADDR1)                                                                   \
  if constexpr (MODE == kSmLastStartEnd) {                                                       \
    asm volatile(ADDR0 ADDR1 RC_READ(0) RC_READ(1) RG_VALID(0) RG_VALID(1)                       \
                 RG_CMPA(0) RG_CMPA(1) RG_CMPI(0) RG_CMPI(1) RG_ACC(0) RG_ACC(1)                 \
                 RG_MEND(0) RG_MEND(1) RG_END(0) RG_END(1)                                       \
                 RG_LEAVE(0) RG_LEAVE(1) RG_START(0) RG_START(1) RG_WAIT RG_HOLD(0) RG_HOLD(1)   \
                 : RG_O_CHAIN(0), RG_O_CHAIN(1), RG_O_ACC(0), RG_O_ACC(1), RG_O_START(0),        \
                   RG_O_START(1)                                                                 \
                 : RGC_I_CHAIN(0), RGC_I_CHAIN(1), RG_I_ACC(0), RG_I_ACC(1), RG_I_START(0),      \
                   RG_I_START(1), [T] "s"(T), [init] "s"(init), [idx] "n"(IDX)                   \
                 : "memory", "scc");                                                             \
  } else if constexpr (MODE == kSmLastEnd) {                                                     \
    asm volatile(ADDR0 ADDR1 RC_READ(0) RC_READ(1) RG_VALID(0) RG_VALID(1)                       \
                 RG_CMPA(0) RG_CMPA(1) "s_nop 0\n\t" RG_ACC(0) RG_ACC(1)                         \
                 RG_MEND(0) RG_MEND(1) RG_END(0) RG_END(1) RG_WAIT RG_HOLD(0) RG_HOLD(1)         \
                 : RG_O_CHAIN(0), RG_O_CHAIN(1), RG_O_ACC(0), RG_O_ACC(1)                        \
                 : RGC_I_CHAIN(0), RGC_I_CHAIN(1), RG_I_ACC(0), RG_I_ACC(1), [T] "s"(T),         \
                   [idx] "n"(IDX)                                                                \
                 : "memory", "scc");                                                             \
  } else if constexpr (MODE == kSmFullStart) {                                                   \
    asm volatile(ADDR0 ADDR1 RC_READ(0) RC_READ(1) RG_VALID(0) RG_VALID(1)                       \
                 RG_CMPI(0) RG_CMPI(1) "s_nop 0\n\t" RG_LEAVE(0) RG_LEAVE(1)                     \
                 RG_START(0) RG_START(1) RG_WAIT RG_HOLD(0) RG_HOLD(1)                           \
                 : RG_O_CHAIN(0), RG_O_CHAIN(1), RG_O_START(0), RG_O_START(1)                    \
                 : RGC_I_CHAIN(0), RGC_I_CHAIN(1), RG_I_START(0), RG_I_START(1),                 \
                   [init] "s"(init), [idx] "n"(IDX)                                              \
                 : "memory", "scc");                                                             \
  } else {                                                                                       \
    asm volatile(ADDR0 ADDR1 RC_READ(0) RC_READ(1) RG_VALID(0) RG_VALID(1)                       \
                 RG_WAIT "s_nop 0\n\t" RG_HOLD(0) RG_HOLD(1)                                     \
                 : RG_O_CHAIN(0), RG_O_CHAIN(1)                                                  \
                 : RGC_I_CHAIN(0), RGC_I_CHAIN(1), [idx] "n"(IDX)                                \
                 : "memory");                                                                    \
  }

template <int MODE, int IDX, bool BIG>
__device__ __forceinline__ void raggedStepCls(uint32_t (&s)[2], const uint32_t (&clsIn)[2],
                                              const uint32_t (&rem)[2], StreamBook (&b)[2],
                                              const uint64_t (&wasI)[2], uint64_t (&isI)[2],
                                              const uint64_t (&validPrev)[2],
                                              uint64_t (&validNow)[2], uint32_t T, uint32_t init,
                                              uint32_t rowb) {
  uint32_t a[2], t[2];
  uint64_t m[2], mr[2], l[2];
  uint32_t cls[2] = {clsIn[0], clsIn[1]};
  if constexpr (BIG) {
    asm volatile(RC_MAD(0) RC_MAD(1)
                 : [a0] "=&v"(cls[0]), [a1] "=&v"(cls[1])
                 : [s0] "v"(s[0]), [s1] "v"(s[1]), [k0] "v"(clsIn[0]), [k1] "v"(clsIn[1]),
                   [rowb] "s"(rowb));
    RGC_STEP(RGC_MOV(0), RGC_MOV(1))
  } else {
    RGC_STEP(RC_ADD(0), RC_ADD(1))
  }
  s[0] = t[0];
  s[1] = t[1];
}
#undef RGC_STEP

template <int MODE, int Q, bool BIG>
__device__ __forceinline__ void raggedWalk16Cls(const uint4 (&piece)[2], uint32_t (&s)[2],
                                                const uint32_t (&rem)[2], StreamBook (&b)[2],
                                                uint64_t (&mA)[2], uint64_t (&mB)[2],
                                                uint64_t (&vA)[2], uint64_t (&vB)[2], uint32_t T,
                                                uint32_t init, const uint8_t *eq2, uint32_t rowb) {
  uint32_t cl[4][2];
#define RGC_WORD(K, FIELD)                                                                   \
  _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                            \
    cl[k][0] = eq2[(piece[0].FIELD >> (8 * k)) & 0xffu];                                     \
    cl[k][1] = eq2[(piece[1].FIELD >> (8 * k)) & 0xffu];                                     \
  }                                                                                          \
  raggedStepCls<MODE, 16 * Q + 4 * K + 0, BIG>(s, cl[0], rem, b, mA, mB, vA, vB, T, init, rowb);        \
  raggedStepCls<MODE, 16 * Q + 4 * K + 1, BIG>(s, cl[1], rem, b, mB, mA, vB, vA, T, init, rowb);        \
  raggedStepCls<MODE, 16 * Q + 4 * K + 2, BIG>(s, cl[2], rem, b, mA, mB, vA, vB, T, init, rowb);        \
  raggedStepCls<MODE, 16 * Q + 4 * K + 3, BIG>(s, cl[3], rem, b, mB, mA, vB, vA, T, init, rowb);
  RGC_WORD(0, x) RGC_WORD(1, y) RGC_WORD(2, z) RGC_WORD(3, w)
#undef RGC_WORD
}

// lines of the batch: Batch::n, or what the device-side count says if that is fewer
__device__ __forceinline__ uint64_t raggedLineCount(uint64_t n, const uint64_t *nDev) {
  if (!nDev) return n;
  const uint64_t have = *nDev;
  return have < n ? have : n;
}

// a wave-uniform 64-bit value the compiler must keep in SGPRs (it feeds "s" asm operands)
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
  const uint32_t lo = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v))));
  const uint32_t hi = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v >> 32))));
  return (uint64_t(hi) << 32) | lo;
}

// HOT (REDGPU_TAB_HOT_ROWS DFAs; see k_stream.h): the lanes found in the sink after a 64-byte
// block re-walk the block's `rem` valid bytes from its saved entry state - hot steps through the
// LDS table, cold ones through the class table - with the bookkeeping in global state ids.
template <int MODE>
__device__ __noinline__ SlowBook slowRagged(const DevDfa &d, const uint8_t *tab8, const uint8_t *eq,
                                            const uint8_t *p, uint32_t off, uint32_t rem,
                                            SlowBook in) {
  uint32_t st = in.st, accS = in.accS, endv = in.endv, startv = in.startv;
  constexpr bool kAcc = MODE == kSmLastStartEnd || MODE == kSmLastEnd;
  constexpr bool kStart = MODE == kSmLastStartEnd || MODE == kSmFullStart;
  const uint16_t *cls = reinterpret_cast<const uint16_t *>(d.table);
  // four 16-byte requests (the whole block is readable: k_ragged read it from here), then
  // `rem` steps
#pragma unroll 1
  for (uint32_t c4 = 0; c4 < 4 && 16 * c4 < rem; ++c4) {
    const uint4 v = *reinterpret_cast<const uint4 *>(p + 16 * c4);
#pragma unroll 1
    for (uint32_t wi = 0; wi < 4; ++wi) {
      const uint32_t word = wi == 0 ? v.x : wi == 1 ? v.y : wi == 2 ? v.z : v.w;
#pragma unroll 1
      for (uint32_t kb = 0; kb < 4; ++kb) {
        const uint32_t k = 16 * c4 + 4 * wi + kb;
        if (k >= rem) break;
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

// =========================================================================================
// k_ragged<MODE, TABK>: the ragged walk with the lanes kept busy.  The first form gave a lane one
// line per tile and ran the wave for as many blocks as its longest line needed; here a lane whose
// line ends takes the next line of the workgroup's range at the next block boundary, so a lane
// idles only for the rest of its line's last block (uniform 32-256 B lines: 82 % of the steps are
// valid against 56 %; geometric lengths the same 82 % against ~25 %, with no sorting pre-pass).
//  * the workgroup owns the contiguous lines [n w / G, n (w + 1) / G); a cursor in LDS hands them
//    out in input order, one wave-aggregated ds_add per block boundary (lanes of a wave hold
//    neighbouring lines: reads and result stores stay local);
//  * every lane holds its current line and the NEXT one (index, offset, length); the line after
//    that is claimed, and its offsets requested, at the top of the block in which the current
//    line ends - a whole block's walk ahead of the first use, unconditionally (lanes that claim
//    nothing re-read offsets[0..1]) so the compiler's vmcnt counts stay exact;
//  * an empty line takes one block slot with no valid step;
//  * a launch cannot end before the line that was started last has been walked by its one lane,
//    a 64-byte block per turn: on skewed lengths (geometric text: the longest of 2^21 lines is
//    14 x the mean) that tail was longer than everything before it.  k_ragged_outliers lists the
//    lines of at least `outCtl[1]` bytes (a multiple of the batch's mean length, so they are few);
//    every workgroup takes its share of that list FIRST - slots [0, nOutW) of its cursor - and
//    then its contiguous range, in which those lines are passed over.  The long lines so run
//    beside the bulk instead of after it, and the tail is at most the threshold long;
//  * a huge line (8 x that threshold and more) is listed as PIECES that are walked at once, for
//    DFAs whose state a 64-byte lead-in predicts (see k_ragged_outliers): `kind`, `ent` below.
// =========================================================================================
template <int MODE, int TABK = kTabFused>
__global__ void __launch_bounds__(kStreamThreads)
k_ragged(DevDfa d, Batch io) {
  constexpr bool HOT = TABK == kTabHot;
  constexpr bool BIG = TABK == kTabClsBig;
  constexpr bool CLS = TABK == kTabCls || BIG;
  constexpr bool IDXD = HOT || CLS;
  constexpr int CH = kStreamChains;
  constexpr int THREADS = kStreamThreads;
  constexpr bool kAcc = MODE == kSmLastStartEnd || MODE == kSmLastEnd;
  constexpr bool kStart = MODE == kSmLastStartEnd || MODE == kSmFullStart;
  constexpr uint32_t kLdsBytes = BIG ? kStreamBigLds : kStreamTabBytes + 1024;
  __shared__ __align__(16) uint8_t lds[kLdsBytes];  // table at LDS offset 0
  __shared__ uint32_t cursor;                       // lines of this workgroup's range handed out
  uint8_t *tab = lds;
  int32_t *ldsRes = reinterpret_cast<int32_t *>(lds + kStreamTabBytes);

  const uint32_t init =
      HOT ? (d.init - d.hotLo < d.nHot ? d.init - d.hotLo + d.hotShift : 0x1ffu)
          : (CLS && !BIG) ? d.init * d.clsRowBytes : d.init;
  const uint32_t firstAccept = HOT ? d.firstAccept - d.hotLo + d.hotShift
                                   : (CLS && !BIG) ? d.firstAccept * d.clsRowBytes
                                                   : d.firstAccept;
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
  {
    const uint4 *src =
        reinterpret_cast<const uint4 *>(d.table + (HOT ? d.hot8Off : CLS ? d.clsOff : 0u));
    const uint32_t n16 = HOT ? kStreamTabBytes / 16 : CLS ? d.clsBytes / 16 : d.tableBytes / 16;
    constexpr uint32_t kStagePieces = BIG ? 1 : kStreamTabBytes / 16 / THREADS + (CLS ? 1 : 0);
    uint4 v[kStagePieces];
    if (BIG) {
      for (uint32_t i = threadIdx.x; i < n16; i += THREADS) reinterpret_cast<uint4 *>(tab)[i] = src[i];
    }
#pragma unroll
    for (uint32_t k = 0; k < kStagePieces; ++k) {
      const uint32_t i = k * THREADS + threadIdx.x;
      v[k] = (!BIG && i < n16) ? src[i] : make_uint4(0, 0, 0, 0);
    }
    const int32_t myRes = IDXD ? 0 : threadIdx.x < d.nStates ? d.result[threadIdx.x] : 0;
    uint4 *dst = reinterpret_cast<uint4 *>(tab);
#pragma unroll
    for (uint32_t k = 0; k < kStagePieces; ++k) {
      const uint32_t i = k * THREADS + threadIdx.x;
      if (!BIG && i < (kStreamTabBytes + 1024) / 16) dst[i] = v[k];
    }
    if (!CLS && threadIdx.x < 256) {
      // HOT: the kilobyte holds the byte -> class map for slowRagged()'s cold steps (k_stream.h)
      if (TABK == kTabHot) reinterpret_cast<uint8_t *>(ldsRes)[threadIdx.x] = d.equivLeader[threadIdx.x];
      else ldsRes[threadIdx.x] = myRes;
    }
    if (threadIdx.x == 0) cursor = 0;
  }
  asm volatile("" : : "v"(tab) : "memory");  // the table is read from inline asm: see k_stream.h
  __syncthreads();

  const uint64_t nLines = raggedLineCount(io.n, io.nDev);
  const uint64_t total = io.offsets[nLines];
  const uint64_t padStart = total >= 128 ? total - 128 : 0;  // pad[] = data[padStart, total) + 0s
  const int32_t initResult = IDXD ? (d.init >= d.firstAccept ? d.result[d.init] : 0)
                                 : (init >= firstAccept ? ldsRes[init] : 0);
  // this workgroup's lines: the contiguous range [lo, lo + range).  (Handing them out longest
  // first through the bucketing pass's permutation - so that a long line cannot start last -
  // measured slower at every shape, 591 against 706 GB/s on geometric lengths: the sort costs
  // 13 us, the lanes of a wave then read all over the buffer, and the permutation entry is a
  // dependent load in front of the offsets.  DESIGN.md section 7 has what to try instead.)
  const uint64_t lo = nLines * blockIdx.x / gridDim.x;
  const uint32_t range = uint32_t(nLines * (blockIdx.x + 1) / gridDim.x - lo);
  const uint32_t lane = threadIdx.x & 63u;
  // the long lines: this workgroup's share of the list, and the length that makes a line one
  const uint32_t nOut = io.outCtl ? io.outCtl[0] : 0u;
  const uint64_t longFrom = io.outCtl && io.outCtl[1] != 0xffffffffu ? uint64_t(io.outCtl[1]) : ~0ull;
  // (entry w, w + G, w + 2 G ... of the list: the pieces of one huge line are neighbours in it and
  // so spread over all workgroups)
  const uint32_t nOutW = nOut > blockIdx.x ? (nOut - blockIdx.x + gridDim.x - 1) / gridDim.x : 0u;
  const uint32_t slots = nOutW + range;

  // current line, next line, and the claim in flight (valid for the lanes that made it)
  bool have[CH], nHave[CH], dry[CH];
  uint32_t ln[CH], len[CH], done[CH], nLn[CH], nLen[CH];
  uint64_t lineOff[CH], nOff[CH];
  uint32_t tLn[CH];
  uint64_t tO[CH], tE[CH];
  bool tHave[CH], tOut[CH];
  // pieces (fused tables only): 0 = a whole line, 2 = a piece, 3 = a piece whose first block is
  // the 64 bytes in front of it, walked from the initial state to guess its entry state
  uint32_t kind[CH], nKind[CH], tKind[CH], ent[CH];

  // wave-aggregated claim of one line for every (lane, chain) with want[c]; requests its offsets
  auto claim = [&](const bool (&want)[CH]) {
    uint64_t need[CH];
    uint32_t k = 0, before[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      need[c] = __builtin_amdgcn_ballot_w64(want[c]);
      before[c] = k;
      k += uint32_t(__builtin_popcountll(need[c]));
    }
    uint32_t base = 0;
    if (k) {
      if (lane == 0) base = atomicAdd(&cursor, k);
      base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(need[c] >> 32),
                                __builtin_amdgcn_mbcnt_lo(uint32_t(need[c]), 0u));
      const uint32_t slot = base + before[c] + rank;
      tHave[c] = want[c] && slot < slots;
      if (want[c] && !tHave[c]) dry[c] = true;
      // unconditional requests (lanes that claimed nothing re-read entry 0): the offsets pair
      // from the list of long lines or from offsets[], the index from the list or the slot
      tOut[c] = tHave[c] && slot < nOutW;
      const uint64_t at = tHave[c] && !tOut[c] ? lo + (slot - nOutW) : 0;
      const uint64_t entry = uint64_t(blockIdx.x) + uint64_t(slot) * gridDim.x;
      const uint64_t *pair = tOut[c] ? io.outRec + 2 * entry : io.offsets + at;
      const uint32_t *lnAt = tOut[c] ? io.outLn + entry : io.outLn;
      tO[c] = pair[0];
      tE[c] = pair[1];
      tLn[c] = *lnAt;  // (always readable: the launcher points outLn at the pad without a list)
      if (!tOut[c]) tLn[c] = uint32_t(at);
    }
  };
  // a line of the contiguous range that is on the list is not this slot's to walk
  auto passedOver = [&](int c) -> bool { return !tOut[c] && tE[c] - tO[c] >= longFrom; };
  // the two top bits of a list entry's end say what it is (offsets stay far below 2^62)
  auto takeKind = [&](int c) {
    tKind[c] = uint32_t(tE[c] >> 62);
    tE[c] &= (1ull << 62) - 1;
  };
  auto lengthOf = [&](uint64_t o, uint64_t e) -> uint32_t {
    // Batch::stride doubles as "trailing delimiter bytes per line" for ragged lines
    return e - o >= io.stride ? uint32_t(e - o - io.stride) : 0u;
  };

  uint32_t s[CH], accS[CH], endv[CH], startv[CH];
  uint32_t g[CH];  // HOT: global id of the lane's state while it is outside the hot set
  uint64_t mA[CH], mB[CH], vA[CH], vB[CH];
  auto freshLine = [&](int c) {
    s[c] = init; accS[c] = 0; endv[c] = 0; startv[c] = 0; g[c] = kNoState; ent[c] = d.init;
    if (HOT && init == 0x1ffu) { s[c] = 255u; g[c] = d.init; }
  };
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    dry[c] = false;
    mA[c] = ~0ull; mB[c] = ~0ull; vA[c] = ~0ull; vB[c] = ~0ull;
    done[c] = 0;
    freshLine(c);
  }
  {
    const bool all[CH] = {true, true};
    claim(all);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      takeKind(c);
      have[c] = tHave[c] && !passedOver(c); ln[c] = tLn[c]; lineOff[c] = have[c] ? tO[c] : 0;
      len[c] = have[c] ? lengthOf(tO[c], tE[c]) : 0;
      kind[c] = tHave[c] ? tKind[c] : 0u;
    }
    claim(all);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      takeKind(c);
      nHave[c] = tHave[c] && !passedOver(c); nLn[c] = tLn[c]; nOff[c] = tO[c];
      nLen[c] = lengthOf(tO[c], tE[c]);
      nKind[c] = tHave[c] ? tKind[c] : 0u;
    }
  }

  // (Tried and dropped: reading ALIGNED 64-byte sectors and cutting the walk's window out of two
  // of them in registers - a 4-stage barrel shift + v_alignbyte, ~95 VALU per window.  It removes
  // the re-fetching of cache lines that unaligned blocks cause (TCC misses x 128 B = 3.3 x the
  // input on 32-256 B lines) but this kernel is bound by VALU issue and step latency, not by
  // memory: 2.21 against 2.73 TB/s on 256-byte lines, 1.42 against 1.54 TB/s on 32-256 B.)
  BlockRegs<1> A[CH], B[CH];
  auto blockPtr = [&](uint64_t bo) -> const uint8_t * {
    return bo + 64 <= total ? io.data + bo : io.pad + (bo - padStart);
  };
  auto issueAt = [&](BlockRegs<1> (&blk)[CH], const uint64_t (&bo)[CH]) {
    const uint8_t *src[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) src[c] = blockPtr(bo[c]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int c = 0; c < CH; ++c) blk[c].p[k] = *reinterpret_cast<const uint4 *>(src[c] + 16 * k);
    }
  };

  // one block of every lane's current line: X holds it, Y receives the block after it
  auto turn = [&](const BlockRegs<1> (&X)[CH], BlockRegs<1> (&Y)[CH]) {
    bool ends[CH];
    // claim first, then request the block: the compiler moves the claimed offsets out of their
    // load registers somewhere in the middle of the walk, and the wait it puts there must not
    // cover the block requests (requested after the offsets they are not waited for: vmcnt(8))
    {
      bool want[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        ends[c] = !have[c] || done[c] + 64u >= len[c];
        want[c] = ends[c] && !dry[c];
      }
      claim(want);
    }
    {
      uint64_t follow[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c)
        follow[c] = ends[c] ? (nHave[c] ? nOff[c] : 0) : lineOff[c] + done[c] + 64u;
      issueAt(Y, follow);
    }
    // loop-carried lane masks: pinned to SGPRs where the asm steps take them
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      mA[c] = uniform64(mA[c]);
      vA[c] = uniform64(vA[c]);
    }

    uint32_t rem[CH];
    StreamBook b[CH];
    uint32_t s0[CH];  // HOT: the block's entry state (hot index), for the re-walk
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const uint32_t left = have[c] ? len[c] - done[c] : 0u;
      rem[c] = left > 64u ? 64u : left;
      b[c].acc = IDXD ? 0u : accS[c]; b[c].end = 0; b[c].start = 0;
      s0[c] = s[c];
    }
    uint4 piece[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) piece[c] = X[c].p[0];
    if constexpr (CLS) raggedWalk16Cls<MODE, 0, BIG>(piece, s, rem, b, mA, mB, vA, vB, firstAccept, init, tab, d.clsRowBytes);
    else raggedWalk16<MODE, 0>(piece, s, rem, b, mA, mB, vA, vB, firstAccept, init);
#pragma unroll
    for (int c = 0; c < CH; ++c) piece[c] = X[c].p[1];
    if constexpr (CLS) raggedWalk16Cls<MODE, 1, BIG>(piece, s, rem, b, mA, mB, vA, vB, firstAccept, init, tab, d.clsRowBytes);
    else raggedWalk16<MODE, 1>(piece, s, rem, b, mA, mB, vA, vB, firstAccept, init);
#pragma unroll
    for (int c = 0; c < CH; ++c) piece[c] = X[c].p[2];
    if constexpr (CLS) raggedWalk16Cls<MODE, 2, BIG>(piece, s, rem, b, mA, mB, vA, vB, firstAccept, init, tab, d.clsRowBytes);
    else raggedWalk16<MODE, 2>(piece, s, rem, b, mA, mB, vA, vB, firstAccept, init);
#pragma unroll
    for (int c = 0; c < CH; ++c) piece[c] = X[c].p[3];
    if constexpr (CLS) raggedWalk16Cls<MODE, 3, BIG>(piece, s, rem, b, mA, mB, vA, vB, firstAccept, init, tab, d.clsRowBytes);
    else raggedWalk16<MODE, 3>(piece, s, rem, b, mA, mB, vA, vB, firstAccept, init);

    // fold (as k_ragged), with the block's offset in the line per lane
    bool redo[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) redo[c] = HOT && rem[c] != 0 && s[c] == 255u;
    if (HOT && __builtin_amdgcn_ballot_w64(redo[0] || redo[CH - 1])) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (redo[c]) {
          const SlowBook o = slowRagged<MODE>(
              d, tab, reinterpret_cast<const uint8_t *>(ldsRes), blockPtr(lineOff[c] + done[c]),
              done[c], rem[c],
              SlowBook{g[c] != kNoState ? g[c] : toGlobal(s0[c]), accS[c], endv[c], startv[c]});
          accS[c] = o.accS; endv[c] = o.endv; startv[c] = o.startv;
          s[c] = toHot(o.st);
          g[c] = s[c] != 255u ? kNoState : o.st;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (HOT && redo[c]) continue;
      const uint32_t off = done[c];
      const bool full = rem[c] == 64;
      if (kAcc && !IDXD) {
        accS[c] = b[c].acc;
        endv[c] = b[c].end ? off + b[c].end : endv[c];
        if (full && s[c] >= firstAccept) { accS[c] = s[c]; endv[c] = off + 64; }
      }
      if (kAcc && IDXD) {
        if (b[c].end) { accS[c] = toGlobal(b[c].acc); endv[c] = off + b[c].end; }
        if (full && s[c] >= firstAccept && (CLS || s[c] != 255u)) {
          accS[c] = toGlobal(s[c]);
          endv[c] = off + 64;
        }
      }
      if (kStart) {
        startv[c] = b[c].start ? off + b[c].start - 1 : startv[c];
        const bool wasInit63 = (mA[c] >> lane) & 1;
        if (full && wasInit63 && s[c] != init) startv[c] = off + 63;
      }
    }
    // a piece's lead-in block is behind it: what it arrived in (as a global state id) is the guess
    // of its entry state, and nothing seen on the way belongs to the piece
    auto globalOf = [&](int c) -> uint32_t {
      return HOT ? (g[c] != kNoState ? g[c] : toGlobal(s[c])) : CLS ? toGlobal(s[c]) : s[c];
    };
    // (pieces are rare: what concerns them sits behind one wave-uniform test per turn - done per
    // lane on every turn it cost equal lines 4 % of their rate)
    const bool anyPiece = __builtin_amdgcn_ballot_w64((kind[0] | kind[CH - 1]) != 0u) != 0;
    if (anyPiece) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if ((kind[c] & 1u) && done[c] == 0) {
          ent[c] = globalOf(c);
          accS[c] = 0; endv[c] = 0; startv[c] = 0;
        }
      }
    }

    // The claimed lines' offsets (requested at the top of the turn) are taken BEFORE this turn's
    // results are stored: vmcnt counts stores too, and conditional ones cannot be counted
    // exactly, so a wait placed after them becomes vmcnt(0) and sits out their acknowledgements
    // (+1.5-2 us per turn whenever some lane of the wave finishes a line, i.e. on every turn of
    // mixed-length lines).
    uint64_t pOff[CH];
    uint32_t pLen[CH];
    bool pHave[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      // (the empty asm keeps the compiler from hoisting this wait into the walk above)
      asm volatile("" : "+v"(tO[c]), "+v"(tE[c]), "+v"(tLn[c]) : : "memory");
      takeKind(c);
      pOff[c] = tO[c];
      pLen[c] = lengthOf(tO[c], tE[c]);
      pHave[c] = tHave[c] && !passedOver(c);
      asm volatile("" : "+v"(pOff[c]), "+v"(pLen[c]) : : "memory");
    }

    // lines that ended in this block report; their lanes move on to the next line
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      {
        // Every lane stores on every turn - the ones with nothing to report into a dummy slot
        // behind the tail pad: stores under a branch are vm operations the compiler cannot
        // count, and the next turn's waits for its input block degrade to vmcnt(0).
        const bool report = have[c] && ends[c];
        int32_t rr;
        uint32_t en;
        uint32_t st = startv[c];
        const uint32_t accAt = report && endv[c] ? accS[c] : 0u;
        if (len[c] == 0) {
          rr = initResult;  // no byte walked: the start state's own result, positions 0
          en = 0;
          st = 0;
        } else if (kAcc) {
          rr = IDXD ? d.result[accAt] : ldsRes[accAt];
          rr = endv[c] ? rr : 0;
          en = endv[c];
        } else if (IDXD) {
          const uint32_t sG = g[c] != kNoState ? g[c] : toGlobal(s[c]);
          const bool accepting = report && sG >= d.firstAccept && sG < d.nStates;
          rr = d.result[accepting ? sG : 0u];
          rr = accepting ? rr : 0;
          en = len[c];
        } else {
          rr = ldsRes[s[c] & 0xffu];
          rr = s[c] >= firstAccept ? rr : 0;
          en = len[c];
        }
        const uint64_t at = ln[c];
        uint8_t *dummy = const_cast<uint8_t *>(io.pad) + 192;
        // a piece leaves its record where a line leaves its Outcome: the same three stores
        // (record: last accepting state | accepted << 31; end | exit state << 32 | entry guess
        // << 48, the states as global ids - at most 16 bits, the launcher sees to that; start)
        int32_t v32 = rr;
        uint64_t vEnd = rr ? uint64_t(en) : 0, vStart = rr ? uint64_t(st) : 0;
        int32_t *resTo = io.result;
        uint64_t *endTo = io.end, *startTo = io.start;
        if (anyPiece && (kind[c] & 2u)) {
          const uint32_t exitG = globalOf(c);
          v32 = int32_t((endv[c] ? accS[c] : 0u) | (endv[c] ? 1u << 31 : 0u));
          vEnd = uint64_t(endv[c]) | (uint64_t(exitG & 0xffffu) << 32) |
                 (uint64_t(ent[c] & 0xffffu) << 48);
          vStart = startv[c];
          resTo = io.pieceRes; endTo = io.pieceEnd; startTo = io.pieceStart;
        }
        int32_t *rp = report ? resTo + at : reinterpret_cast<int32_t *>(dummy);
        *rp = v32;
        uint64_t *ep = report && endTo ? endTo + at : reinterpret_cast<uint64_t *>(dummy);
        *ep = vEnd;
        if (kStart) {
          uint64_t *sp = report && startTo ? startTo + at : reinterpret_cast<uint64_t *>(dummy);
          *sp = vStart;
        }
      }
      if (ends[c]) {
        have[c] = nHave[c]; ln[c] = nLn[c]; lineOff[c] = nOff[c]; len[c] = nLen[c];
        kind[c] = nKind[c];
        nKind[c] = tHave[c] ? tKind[c] : 0u;
        done[c] = 0;
        freshLine(c);
        nHave[c] = pHave[c]; nLn[c] = tLn[c]; nOff[c] = pOff[c]; nLen[c] = pLen[c];

      } else {
        done[c] += 64u;
      }
      // a fresh line starts in the initial state, reached by a "valid" step
      const uint64_t fresh = __builtin_amdgcn_ballot_w64(ends[c]);
      mA[c] = uniform64(mA[c] | fresh);
      vA[c] = uniform64(vA[c] | fresh);
    }
  };

  {
    uint64_t first[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) first[c] = lineOff[c];
    issueAt(A, first);
  }
  // until no lane holds a line and none can claim one (a lane whose current and next line were
  // both passed over holds nothing and is not dry)
  auto busy = [&]() -> bool {
    bool any = false;
#pragma unroll
    for (int c = 0; c < CH; ++c) any = any || have[c] || nHave[c] || !dry[c];
    return __builtin_amdgcn_ballot_w64(any) != 0;
  };
  while (true) {
    if (!busy()) break;
    turn(A, B);
    if (!busy()) break;
    turn(B, A);
  }
}

#undef RG_PERM
#undef RG_READ
#undef RG_VALID
#undef RG_CMPA
#undef RG_CMPI
#undef RG_ACC
#undef RG_MEND
#undef RG_END
#undef RG_LEAVE
#undef RG_START
#undef RG_WAIT
#undef RG_HOLD
#undef RG_O_CHAIN
#undef RG_O_ACC
#undef RG_O_START
#undef RG_I_CHAIN
#undef RG_I_ACC
#undef RG_I_START

// ---- length bucketing ---------------------------------------------------------------------
// A wave of k_ragged runs for as many 64-byte blocks as its LONGEST line needs; on text whose
// line lengths are skewed (geometric: the longest of a wave's 128 lines is ~5x the mean) most
// lanes idle.  A counting sort of the line indices by block count (descending, so the heavy
// tiles run first and the tail of the launch is made of light ones) puts lines of like length
// in the same wave.  Three small kernels over offsets[] only (8 bytes per line); each workgroup
// owns a contiguous range of lines so that lines of one bucket keep their memory order.
constexpr int kBucketCount = 64;
constexpr int kBucketThreads = 256;

__device__ __forceinline__ uint32_t bucketOf(const uint64_t *offsets, uint64_t line, uint64_t trim) {
  uint64_t len = offsets[line + 1] - offsets[line];
  len = len >= trim ? len - trim : 0;
  const uint64_t nb = (len + 63) >> 6;
  // descending: bucket 0 = the longest lines (63+ blocks), bucket 63 = empty lines
  return uint32_t(kBucketCount - 1 - (nb < kBucketCount - 1 ? nb : kBucketCount - 1));
}

// pass 1: per-workgroup length histograms (and, by workgroup 0, the tail pad of k_tail_pad)
__global__ void __launch_bounds__(kBucketThreads)
k_bucket_hist(const uint8_t *data, const uint64_t *offsets, uint64_t n, uint64_t trim,
              uint64_t perBlock, uint32_t *hist /* [bucket][block] */, uint8_t *pad) {
  __shared__ uint32_t h[kBucketCount];
  if (threadIdx.x < kBucketCount) h[threadIdx.x] = 0;
  if (blockIdx.x == 0 && threadIdx.x < 192) {
    const uint64_t total = offsets[n];
    const uint64_t i = (total >= 128 ? total - 128 : 0) + threadIdx.x;
    pad[threadIdx.x] = i < total ? data[i] : uint8_t(0);
  }
  __syncthreads();
  const uint64_t lo = uint64_t(blockIdx.x) * perBlock;
  const uint64_t hi = lo + perBlock < n ? lo + perBlock : n;
  for (uint64_t i = lo + threadIdx.x; i < hi; i += kBucketThreads)
    atomicAdd(&h[bucketOf(offsets, i, trim)], 1u);
  __syncthreads();
  if (threadIdx.x < kBucketCount) hist[uint64_t(threadIdx.x) * gridDim.x + blockIdx.x] = h[threadIdx.x];
}

// pass 2: every workgroup derives its own scatter bases from the whole histogram table (64 x
// <= 256 entries: cheaper than a separate scan kernel and its dispatch), reaches the same
// verdict, and scatters its lines' indices.  Verdict: a wave of 128 lines drawn at random from
// the length histogram spends 128 x E[longest] blocks where its lines need 128 x mean;
// bucketing is applied when that ratio exceeds 1.5 (uniform 32..256-byte lines: 1.48;
// geometric lengths: ~5).  Workgroup 0 publishes the verdict for k_ragged.
__global__ void __launch_bounds__(kBucketThreads)
k_bucket_scatter(const uint64_t *offsets, uint64_t n, uint64_t trim, uint64_t perBlock,
                 const uint32_t *hist, uint32_t *perm, uint32_t *usePerm) {
  __shared__ uint32_t rowTot[kBucketCount], rowBefore[kBucketCount], cur[kBucketCount];
  __shared__ uint32_t verdict;
  const uint32_t nBlocks = gridDim.x;
  {
    // 4 threads per bucket row
    const uint32_t k = threadIdx.x >> 2, part = threadIdx.x & 3;
    uint32_t tot = 0, before = 0;
    for (uint32_t blk = part; blk < nBlocks; blk += 4) {
      const uint32_t v = hist[uint64_t(k) * nBlocks + blk];
      tot += v;
      before += blk < blockIdx.x ? v : 0;
    }
    tot += __shfl_xor(tot, 1); tot += __shfl_xor(tot, 2);
    before += __shfl_xor(before, 1); before += __shfl_xor(before, 2);
    if (part == 0) { rowTot[k] = tot; rowBefore[k] = before; }
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    // lane k = bucket k (descending length: bucket k holds lines of 63 - k blocks)
    const uint32_t k = threadIdx.x;
    uint32_t incl = rowTot[k];
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = __shfl_up(incl, o);
      if (k >= uint32_t(o)) incl += u;
    }
    cur[k] = incl - rowTot[k] + rowBefore[k];
    const float total = float(__shfl(incl, 63));
    // P(length <= 63 - k blocks) = (lines in buckets >= k) / total
    const float geq = total > 0.f ? (total - float(incl - rowTot[k])) / total : 0.f;
    float g = geq;                       // ^128 by squaring
    for (int q = 0; q < 7; ++q) g *= g;
    const float gNext = k < 63 ? __shfl_down(g, 1) : 0.f;   // P(max <= 62 - k)
    const float nbk = float(63 - k);
    float emax = nbk * (g - (k < 63 ? gNext : 0.f));
    float mean = total > 0.f ? nbk * float(rowTot[k]) / total : 0.f;
    for (int o = 32; o; o >>= 1) { emax += __shfl_xor(emax, o); mean += __shfl_xor(mean, o); }
    if (k == 0) {
      verdict = (mean > 0.f && emax > 1.5f * mean) ? 1u : 0u;
      if (blockIdx.x == 0) *usePerm = verdict;
    }
  }
  __syncthreads();
  if (!verdict) return;
  const uint64_t lo = uint64_t(blockIdx.x) * perBlock;
  const uint64_t hi = lo + perBlock < n ? lo + perBlock : n;
  for (uint64_t i = lo + threadIdx.x; i < hi; i += kBucketThreads)
    perm[atomicAdd(&cur[bucketOf(offsets, i, trim)], 1u)] = uint32_t(i);
}

constexpr uint64_t kBucketMinLines = 16384;

// Scratch for the ragged launch (tail pad, permutation, histograms): the process-wide pool of
// host_stage.cpp, keyed by (device, stream) - work queued on one stream runs in order, so the next
// call on that stream may reuse the buffer the previous one used; another stream gets its own.
// hipMallocAsync/hipFreeAsync per call cost ~12 us of host time and a bubble on the stream.
inline hipError_t raggedScratch(hipStream_t stream, size_t bytes, void **out) {
  return scratchFor(stream, bytes, out);
}

// pad[0..192) = data[padStart, total) followed by zeros (see the header comment)
__global__ void __launch_bounds__(192)
k_tail_pad(const uint8_t *data, const uint64_t *offsets, uint64_t nMax, uint8_t *pad,
           const uint64_t *nDev = nullptr) {
  const uint64_t total = offsets[raggedLineCount(nMax, nDev)];
  const uint64_t padStart = total >= 128 ? total - 128 : 0;
  const uint64_t i = padStart + threadIdx.x;
  pad[threadIdx.x] = i < total ? data[i] : uint8_t(0);
}

// The pre-pass shared by k_ragged and the ragged form of k_generic: rb = b plus the tail pad
// (wantPad) and, for >= 16384 lines unless REDGPU_F_NO_BUCKETING, the length-bucketed
// permutation with its verdict behind it.  Scratch layout (cached per thread and stream, see
// raggedScratch): [pad 192 -> 256][perm u32[n]][usePerm u32][pad to 16][hist u32[64 * nb]].
inline hipError_t prepareRagged(const Batch &b, const LaunchCfg &cfg, hipStream_t stream,
                                bool wantPad, Batch &rb) {
  rb = b;
  const bool bucket = b.n >= kBucketMinLines && b.n < (1ull << 32) && !cfg.noBucketing;
  if (!bucket && !wantPad) return hipSuccess;
  const uint32_t nb = bucket ? uint32_t(b.n / 4096 < 256 ? (b.n + 4095) / 4096 : 256) : 0;
  const size_t permBytes = bucket ? (size_t(b.n + 1) * 4 + 15) & ~size_t(15) : 0;
  void *scratch = nullptr;
  hipError_t e = raggedScratch(stream, 256 + permBytes + size_t(kBucketCount) * nb * 4, &scratch);
  if (e != hipSuccess) return e;
  uint8_t *pad = static_cast<uint8_t *>(scratch);
  if (wantPad) rb.pad = pad;
  if (bucket) {
    const uint64_t perBlock = (b.n + nb - 1) / nb;
    uint32_t *perm = reinterpret_cast<uint32_t *>(pad + 256);
    uint32_t *hist = reinterpret_cast<uint32_t *>(pad + 256 + permBytes);
    hipLaunchKernelGGL(k_bucket_hist, dim3(nb), dim3(kBucketThreads), 0, stream, b.data, b.offsets,
                       b.n, b.stride, perBlock, hist, pad);
    hipLaunchKernelGGL(k_bucket_scatter, dim3(nb), dim3(kBucketThreads), 0, stream, b.offsets, b.n,
                       b.stride, perBlock, hist, perm, perm + b.n);
    rb.perm = perm;
  } else {
    hipLaunchKernelGGL(k_tail_pad, dim3(1), dim3(192), 0, stream, b.data, b.offsets, b.n, pad, nullptr);
  }
  return hipGetLastError();
}

#include "k_ragged_long.h"

template <int MODE, int TABK = kTabFused>
hipError_t launchRaggedT(const DevDfa &d, const Batch &b, const LaunchCfg &cfg,
                         hipStream_t stream) {
  const uint64_t linesPerTile = uint64_t(kStreamThreads) * kStreamChains;
  const uint64_t tiles = (b.n + linesPerTile - 1) / linesPerTile;
  // one workgroup per CU.  (Two - the fused-table form fits, 119-122 VGPRs and 66 KB of LDS each -
  // were measured in round 3: equal lines +6-8 % (2^23 x 256 B 2.70 -> 2.93 TB/s), mixed lengths
  // -3 % (uniform 32-256: 1.60 -> 1.56), a 2^20-line batch of geometric lengths -18 %.)
  const uint64_t blocks = tiles < uint64_t(cfg.numCUs) ? tiles : uint64_t(cfg.numCUs);
  Batch rb = b;
  // lines are handed out in input order, the long ones of a large batch first
  // (REDGPU_F_NO_BUCKETING: plain input order), the huge ones of a forgetful DFA in pieces.
  // Scratch: [pad 256][32 unused][outLn u32[capE]][outRec u64[2 capE]][hugeLn, hugeFirst u32[capH]]
  // [pieceRes i32[capP]][pieceEnd, pieceStart u64[capP]], every part 16-byte aligned.
  const uint32_t factor = raggedLongFactor();
  static const uint64_t minLines = [] {
    const char *e = getenv("REDGPU_RAGGED_LONG_MIN");  // lab: smallest batch that gets the list
    const long v = e ? atol(e) : long(kLongFirstMinLines);
    return uint64_t(v < 64 ? 64 : v);
  }();
  const bool longFirst = factor && !cfg.noBucketing && b.n >= minLines;
  static const bool piecesOn = [] { const char *e = getenv("REDGPU_RAGGED_PIECES"); return !e || atoi(e) != 0; }();
  // (records keep states as 16-bit global ids)
  const bool pieces = longFirst && piecesOn && (d.forgetful || cfg.forcePieces) && d.nStates <= 65535;
  const uint64_t perX = longFirst ? (b.n + factor - 1) / factor : 0;
  auto pad16 = [](size_t v) { return (v + 15) & ~size_t(15); };
  static const uint32_t hugeX = [] {
    const char *e = getenv("REDGPU_RAGGED_HUGE_X");  // lab: pieces from hugeX x T bytes on
    const int v = e ? atoi(e) : 8;
    return uint32_t(v < 2 ? 2 : v > 1024 ? 1024 : v);
  }();
  const uint64_t capH = pieces ? perX / hugeX + 64 : 0;
  const uint64_t capP = pieces ? perX + perX / hugeX + 64 : 0;
  const uint64_t capE = longFirst ? perX + capP + 64 : 0;
  const size_t offLn = 256 + 32, offRec = offLn + pad16(capE * 4), offHuge = offRec + capE * 16;
  const size_t offPRes = offHuge + 2 * pad16(capH * 4), offPEnd = offPRes + pad16(capP * 4);
  const size_t bytes = offPEnd + 2 * capP * 8;
  if (capE >= (1ull << 32)) return hipErrorInvalidValue;  // (n < 2^32: cannot happen)
  void *scratch = nullptr;
  hipError_t e = raggedScratch(stream, bytes, &scratch);
  if (e != hipSuccess) return e;
  uint8_t *pad = static_cast<uint8_t *>(scratch);
  rb.pad = pad;
  rb.outLn = reinterpret_cast<const uint32_t *>(pad);
  if (longFirst) {
    OutlierBufs ob;
    // (the control words are not in the scratch buffer, which every launch family of the thread
    // and stream overwrites, but beside it: this call's slot arrives zeroed - by the pre-pass of
    // the call before - and the sequence needs no memset in front)
    e = scratchCtlFor(stream, &ob.ctl, &ob.ctlNext);
    if (e != hipSuccess) return e;
    ob.outLn = reinterpret_cast<uint32_t *>(pad + offLn);
    ob.outRec = reinterpret_cast<uint64_t *>(pad + offRec);
    ob.hugeLn = reinterpret_cast<uint32_t *>(pad + offHuge);
    ob.hugeFirst = reinterpret_cast<uint32_t *>(pad + offHuge + pad16(capH * 4));
    ob.capE = uint32_t(capE); ob.capH = uint32_t(capH); ob.capP = uint32_t(capP);
    // (two workgroups per CU at most; 8 and 32 measured the same)
    const uint64_t want = (b.n + 4095) / 4096;
    const uint32_t nb = uint32_t(want < 2ull * uint64_t(cfg.numCUs) ? want : 2ull * uint64_t(cfg.numCUs));
    hipLaunchKernelGGL(k_ragged_outliers, dim3(nb), dim3(kOutlierThreads), 0, stream, b.data,
                       b.offsets, b.n, b.nDev, factor, uint32_t(b.stride), pieces ? hugeX : 0u, pad, ob);
    rb.outCtl = ob.ctl;
    rb.outLn = ob.outLn;
    rb.outRec = ob.outRec;
    if (pieces) {
      rb.hugeLn = ob.hugeLn;
      rb.hugeFirst = ob.hugeFirst;
      rb.pieceRes = reinterpret_cast<int32_t *>(pad + offPRes);
      rb.pieceEnd = reinterpret_cast<uint64_t *>(pad + offPEnd);
      rb.pieceStart = rb.pieceEnd + capP;
    }
  } else {
    hipLaunchKernelGGL(k_tail_pad, dim3(1), dim3(192), 0, stream, b.data, b.offsets, b.n, pad,
                       b.nDev);
  }
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_ragged<MODE, TABK>), dim3(uint32_t(blocks)), dim3(kStreamThreads), 0,
                     stream, d, rb);
  e = hipGetLastError();
  if (e != hipSuccess || !pieces) return e;
  constexpr bool kAcc = MODE == kSmLastStartEnd || MODE == kSmLastEnd;
  constexpr bool kStart = MODE == kSmLastStartEnd || MODE == kSmFullStart;
  // (dynamic LDS = the fused table, at most 64 KB: no attribute to raise)
  // (a small grid: a batch has a few huge lines per million at most.  One without any pays this
  // launch for nothing: 4-5 us by the rocprofv3 trace whatever the grid - the dispatch with its
  // 54 KB of LDS and one dependent load of the count.)
  const uint64_t foldWant = (capH + kFoldLines - 1) / kFoldLines;
  const uint32_t foldBlocks = uint32_t(foldWant < 64 ? foldWant : 64);
  hipLaunchKernelGGL(k_ragged_pieces_fold, dim3(foldBlocks), dim3(kFoldThreads),
                     TABK == kTabFused ? d.tableBytes : 0, stream, d, rb, kAcc ? 1 : 0,
                     kStart ? 1 : 0, TABK);
  return hipGetLastError();
}
