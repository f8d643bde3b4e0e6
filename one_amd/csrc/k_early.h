// k_early.h - k_early<KIND, WALK, ...>: match / check over early-death DFAs (BASELINE configs[3]) - probe
// every line for 16 bytes, park the survivors in LDS, drain them densely
// (included by kernels.hip inside namespace redgpu { namespace { ... } }; see its file map).
#pragma once

// =========================================================================================
// k_early<KIND>: match<style,doLeader> for EARLY-DEATH DFAs - anchored patterns and signature
// sets on arbitrary lines (BASELINE configs[3]: LOG-100, matchLong over 8 M ragged lines), where
// most lines are in a pure dead end after a byte or two and the others walk a whole signature.
// k_generic gives a lane a line: a wave then holds 64 lines until its slowest one is done (half
// of its lanes idle on configs[3]) and pays the line's memory round trips - offsets, first
// bytes, next trip - one behind the other.  Here a workgroup takes LPL lines per lane at a time
// and
//   1. PROBES them: offsets and the first PC x 16 bytes of all of them are requested together,
//      then each is walked through those bytes from registers (eight at a time; a wave moves on
//      once none of its lanes is alive).  A line that is done by then (pure dead end, an
//      early-exit style, end of line) stores its Outcome; a survivor's loop state (MatchWalk) is
//      parked in an LDS queue (one wave-aggregated atomic per wave and line slot);
//   2. DRAINS the queue: the survivors, now dense, are dealt out again - every lane resumes one
//      behind the bytes the probe held and walks it to its end.
// Same lane code as matchLane (MatchWalk::step / finish), so the results are the reference's
// for every style; the table kinds are the LDS-resident ones.
// =========================================================================================
// bytes of a line (>= 16 long) the probe holds in registers: whole 16-byte pieces, at most PC
template <int PC>
__device__ __forceinline__ uint32_t earlyHave(uint64_t n) {
  const uint64_t pieces = n >> 4;
  return 16u * uint32_t(pieces < uint64_t(PC) ? pieces : uint64_t(PC));
}

// LPL = lines per lane and round; WPS = waves per SIMD the register allocation must allow: LDS
// decides how many workgroups share a CU, and occupancy is what this kernel lives on.  Measured on
// configs[3] (2^23 lines, LOG-100; scripts/gpu_run10.sh): 4 lines per lane, 2 workgroups per CU
// 443 us; 2 lines per lane, 3 workgroups per CU 415 us; 1 line, 3 workgroups 437 us.  Requesting
// the next round's offsets and first bytes a phase ahead, and draining two survivors per lane with
// their next 64 bytes requested together, both made it slower (446-569 us: more registers, and
// the launch moves ~2.2 GB through L2 - nearly every cache line of the input is touched by a line
// start, and again when a survivor is drained - so it sits near the memory system's rate for
// scattered 128-byte requests, not on the latency of any one of them).
// PC = 16-byte pieces of a line the probe holds: all 16 bytes of the first piece walked in the
// probe (it was 8: the lines that die between byte 8 and 16 no longer pay a queue slot and a
// reload) 415 -> 345 us; two or four pieces (fewer reloads: 1.6 GB instead of 2.1 GB missing L2)
// 346 / 377 us - no faster; 1024-thread workgroups (32 waves per CU) 363 us; the drain's next
// two pieces requested together 360 us (scripts/gpu_run17.sh).
template <int KIND, class WALK, int LPL, int WPS, int PC, int THREADS, bool LEAN_DRAIN>
__global__ void __launch_bounds__(THREADS, WPS)
k_early(DevDfa d, Batch b, int style, int lead) {
  constexpr uint32_t kEarlyChunk = THREADS * LPL;
  constexpr uint32_t L = LPL;
  extern __shared__ __align__(16) uint8_t lds[];
  // (no LDS copy of the result table: this kernel reads it once per line, and the space buys a
  // third workgroup per CU)
  const Tab<KIND> tab = stageTab<KIND, THREADS, false>(d, lds);
  uint4 *queue = reinterpret_cast<uint4 *>(lds + 512 + ((tableOnlyBytes<KIND>(d) + 15) & ~size_t(15)));
  __shared__ uint32_t qCount;
  LaneCtx c;
  c.eq = lds;
  c.leader = lds + 256;
  c.res = d.result;
  c.init = d.init; c.leaderNext = d.leaderNext; c.nPureDead = d.nPureDead;
  c.firstAccept = d.firstAccept; c.leaderLen = d.leaderLen;
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t nChunks = (b.n + kEarlyChunk - 1) / kEarlyChunk;

  // line -> (byte offset, length); lines past the end of the batch read the last line (never
  // stored: `valid` below)
  auto spanOf = [&](uint64_t line, uint64_t &o, uint64_t &n) {
    const uint64_t ln = line < b.n ? line : b.n - 1;
    if (b.offsets) {
      o = b.offsets[ln];
      const uint64_t e = b.offsets[ln + 1];
      n = e - o >= b.stride ? e - o - b.stride : 0;  // stride = trailing bytes to drop (ragged)
    } else {
      o = ln * b.stride;
      n = b.stride;
    }
  };
  auto store = [&](uint64_t line, WALK &w) {
    uint64_t st, en;
    b.result[line] = w.finish(c, style, st, en);
    if (b.start) b.start[line] = st;
    if (b.end) b.end[line] = en;
  };

  for (uint64_t chunk = blockIdx.x; chunk < nChunks; chunk += gridDim.x) {
    if (threadIdx.x == 0) qCount = 0;
    __syncthreads();
    // ---- 1. probe: offsets and first bytes of all the lane's lines requested together --------
    uint64_t o[L], n[L];
    uint4 head[L][PC];
#pragma unroll
    for (uint32_t k = 0; k < L; ++k)
      spanOf(chunk * kEarlyChunk + uint64_t(k) * THREADS + threadIdx.x, o[k], n[k]);
#pragma unroll
    for (uint32_t k = 0; k < L; ++k)
#pragma unroll
      for (uint32_t j = 0; j < uint32_t(PC); ++j)
        head[k][j] = n[k] >= 16 * (j + 1) ? *reinterpret_cast<const uint4 *>(b.data + o[k] + 16 * j)
                                          : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (uint32_t k = 0; k < L; ++k) {
      const uint64_t line = chunk * kEarlyChunk + uint64_t(k) * THREADS + threadIdx.x;
      const bool valid = line < b.n;
      const uint8_t *p = b.data + o[k];
      WALK w;
      w.begin(c);
      bool alive = valid;
      if (valid && lead && !lookingAt(c, p, 0, n[k])) {
        b.result[line] = 0;
        if (b.start) b.start[line] = 0;
        if (b.end) b.end[line] = 0;
        alive = false;
      }
      if (alive) {
        if (n[k] >= 16) {
          const uint32_t have = earlyHave<PC>(n[k]);
#pragma unroll
          for (uint32_t g = 0; g < 2 * uint32_t(PC); ++g) {
            // past the probe proper only the survivors walk on, from the bytes already in registers
            if (g > 0 && !__builtin_amdgcn_ballot_w64(alive && 8 * g < have)) break;
            const uint4 &h = head[k][g >> 1];
            const uint32_t words[2] = {g & 1 ? h.z : h.x, g & 1 ? h.w : h.y};
#pragma unroll
            for (uint32_t i = 0; i < 8; ++i)
              if (alive && 8 * g < have)
                alive = w.step(tab, c, style, (words[i >> 2] >> (8 * (i & 3))) & 0xffu, 8 * g + i);
          }
          if (!alive || have == n[k]) {  // done within the probe; the others have bytes left
            store(line, w);
            alive = false;
          }
        } else {  // a short line: all of it, byte by byte
          for (uint64_t i = 0; i < n[k] && alive; ++i) alive = w.step(tab, c, style, p[i], i);
          store(line, w);
          alive = false;
        }
      }
      // survivors: one queue slot each, claimed per wave
      const uint64_t mask = __builtin_amdgcn_ballot_w64(alive);
      if (mask) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&qCount, uint32_t(__builtin_popcountll(mask)));
        base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
        if (alive) {
          const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32),
                                    __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
          queue[base + rank] = w.pack(uint32_t(k * THREADS + threadIdx.x));
        }
      }
    }
    __syncthreads();
    // ---- 2. drain: the survivors, dense again, walked to their end ---------------------------
    const uint32_t qn = qCount;
    for (uint32_t q = threadIdx.x; q < qn; q += THREADS) {
      const uint4 en = queue[q];
      WALK w;
      w.unpack(en);
      const uint64_t ln = chunk * kEarlyChunk + (en.x & 0xfffu);
      uint64_t oo, nl;
      spanOf(ln, oo, nl);
      if constexpr (LEAN_DRAIN) {
        // the lean walk without a branch per byte: a piece's 16 steps are selects under the
        // lane's `alive` flag (a lane that has met its pure dead end changes nothing any more),
        // the loop asks once per piece; positions in 32 bits (longer lines: the walk below)
        if (nl < (1ull << 32)) {
          uint32_t st = w.s, accS = w.accS, ms = uint32_t(w.matchStart), me = uint32_t(w.matchEnd);
          uint32_t pos = earlyHave<PC>(nl);
          const uint32_t n32 = uint32_t(nl);
          const uint8_t *p = b.data + oo;
          bool alive = true;
          auto lean = [&](uint32_t byte, uint32_t idx) {
            const uint32_t s2 = tab.next(st, byte);
            const bool leaves = st == c.init && s2 != st;
            const bool acc = s2 >= c.firstAccept;
            ms = alive && leaves ? idx : ms;
            accS = alive && acc ? s2 : accS;
            me = alive && acc ? idx + 1 : me;
            st = alive ? s2 : st;
            alive = alive && (acc || s2 >= c.nPureDead);
          };
          while (alive && pos + 16 <= n32) {
            const uint4 v = *reinterpret_cast<const uint4 *>(p + pos);
            const uint32_t words[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t k = 0; k < 16; ++k) lean((words[k >> 2] >> (8 * (k & 3))) & 0xffu, pos + k);
            pos += 16;
          }
          for (; alive && pos < n32; ++pos) lean(uint32_t(p[pos]), pos);
          w.s = st; w.accS = accS; w.matchStart = ms; w.matchEnd = me;
          store(ln, w);
          continue;
        }
      }
      walkBytes(b.data + oo, earlyHave<PC>(nl), nl,
                [&](uint32_t byte, uint64_t idx) { return w.step(tab, c, style, byte, idx); });
      store(ln, w);
    }
    __syncthreads();
  }
}
