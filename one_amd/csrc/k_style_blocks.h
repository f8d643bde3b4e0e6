// k_style_blocks.h - k_style_blocks<KIND, THREADS, W, POS>: check / match with the early-exit styles (and the
// whole-line styles on odd strides) over the block walk of k_lists.h
// (included by kernels.hip inside namespace redgpu { namespace { ... } }; see its file map).
#pragma once

// =========================================================================================
// k_style_blocks: check / match with the EARLY-EXIT styles (styInstant, styFirst, styTangent;
// include/Matcher.h:382-403, :443-479), no leader, over the same two phases as k_matchall_blocks.
// What those styles report is decided by the FIRST run of accepting positions:
//   styInstant : the first accepting position a0 - result of its state, end = a0 + 1;
//   styTangent : the run of consecutive accepting positions from a0 - result of its last state,
//                end behind it (the loop leaves at the first non-accepting position behind one);
//   styFirst   : the same run cut where the result changes (:457-460) - result of a0's state;
//   start      : the last "left the initial state" position up to and including the position
//                at which the loop left (the update at :446-451 precedes the tests).
// Phase A (mabWalkBlock) walks a block without looking; phase B reads the masks: first set bit,
// first clear bit behind it, and only for styFirst the results along the run.  A lane whose loop
// has left stops taking blocks, a wave whose lanes all have stops reading: a dense DFA is done
// within its first block, where k_fixed walked every line to its end.  Requires what
// k_matchall_blocks requires (absorbing pure dead ends: nothing accepts past one).
// =========================================================================================
template <int KIND, int THREADS, int W, bool POS>
__global__ void __launch_bounds__(THREADS)
k_style_blocks(DevDfa d, Batch b, int style) {
  constexpr uint32_t kPos = 64 / W;
  constexpr uint32_t kPerWord = 4 / W;
  extern __shared__ __align__(16) uint8_t lds[];
  const Tab<KIND> tab = stageTab<KIND, THREADS>(d, lds);
  LaneCtx c{lds, lds + 256, resOf<KIND>(d, lds), d.init, d.leaderNext, d.nPureDead, d.firstAccept,
            d.leaderLen};
  uint32_t *stage = reinterpret_cast<uint32_t *>(lds + 512 + ((ldsTableBytes<KIND>(d) + 15) & ~size_t(15)));
  const uint8_t *stageBytes = reinterpret_cast<const uint8_t *>(stage);
  const int32_t *ldsRes = reinterpret_cast<const int32_t *>(lds + 512 + ((tableOnlyBytes<KIND>(d) + 15) & ~size_t(15)));
  const bool tableAt512 = uint32_t(reinterpret_cast<uintptr_t>(lds)) == 0u;
  const uint8_t *bufEnd = b.data + (b.offsets ? b.offsets[b.n] : b.n * b.stride);
  const int32_t initRes = d.init >= d.firstAccept ? ldsRes[d.init] : 0;
  const uint64_t step = uint64_t(gridDim.x) * THREADS;
  for (uint64_t line = uint64_t(blockIdx.x) * THREADS + threadIdx.x; line < b.n; line += step) {
    const uint8_t *p;
    uint64_t n;
    if (b.offsets) {
      const uint64_t o = b.offsets[line];
      p = b.data + o;
      n = b.offsets[line + 1] - o;
      n = n >= b.stride ? n - b.stride : 0;  // stride = trailing bytes to drop (ragged)
    } else {
      p = b.data + line * b.stride;
      n = b.stride;
    }
    uint32_t s = c.init;
    // mode 0: no accepting position yet; 1: inside the first run; 2: the loop has left
    uint32_t mode = 0;
    int32_t result = n ? 0 : initRes;  // an accepting initial state only counts for empty input
    int32_t r0 = 0, prevR = 0;
    uint64_t matchStart = 0, startOut = 0, curEnd = 0;
    for (uint64_t base = 0; base < n && mode != 2; base += kPos) {
      const uint64_t wasInit = s == c.init ? 1u : 0u;
      uint64_t acc = 0, ini = 0;
      const uint32_t cnt = mabWalkBlock<KIND, THREADS, W>(tab, c, p + base, n - base,
                                                          bufEnd - (p + base), s, stage,
                                                          tableAt512, acc, ini);
      const uint64_t valid = cnt >= 64 ? ~0ull : (1ull << cnt) - 1;
      const uint64_t esc = (((ini << 1) | wasInit) & ~ini) & valid;
      auto resultAt = [&](uint32_t i) -> int32_t {
        const uint8_t *slot = stageBytes + (((i / kPerWord) * THREADS + threadIdx.x) << 2) +
                              W * (i % kPerWord);
        const uint32_t si = W == 1 ? uint32_t(*slot) : uint32_t(*reinterpret_cast<const uint16_t *>(slot));
        return ldsRes[si];
      };
      // the last escape at or below position i of this block, else the one carried in
      auto startAt = [&](uint32_t i) -> uint64_t {
        const uint64_t m = esc & ((2ull << i) - 1);
        return m ? base + 63 - uint32_t(__builtin_clzll(m)) : matchStart;
      };
      if (style == kStyLast || style == kStyFull) {
        // the whole-line styles (fixed strides the streaming kernels do not take): styLast wants the
        // LAST accepting position - the top bit of the mask, its state read back while the block is
        // still staged; styFull only the final state
        if (style == kStyLast && acc) {
          const uint32_t i = 63 - uint32_t(__builtin_clzll(acc));
          prevR = resultAt(i);
          curEnd = base + i + 1;
        }
        if (esc) matchStart = base + 63 - uint32_t(__builtin_clzll(esc));
        continue;
      }
      uint32_t q = 0;  // first position of this block the run still has to look at
      if (mode == 0 && acc) {
        const uint32_t a0 = uint32_t(__builtin_ctzll(acc));
        r0 = prevR = resultAt(a0);
        curEnd = base + a0 + 1;
        if (style == kStyInstant) {
          result = r0;
          if (POS) startOut = startAt(a0);
          mode = 2;
        } else {
          mode = 1;
          q = a0 + 1;
        }
      }
      if (mode == 1) {
        // the run goes on over accepting positions from q; zf = the first one that is not
        const uint64_t clear = ~acc & valid & (q >= 64 ? 0ull : ~0ull << q);
        const uint32_t zf = clear ? uint32_t(__builtin_ctzll(clear)) : cnt;
        uint32_t stop = 0xffffffffu;
        if (style == kStyFirst) {
          uint32_t i = q;
          for (; i < zf; ++i) {
            if (resultAt(i) != r0) break;  // another result: the loop leaves, keeping the first (:457-460)
            curEnd = base + i + 1;
          }
          if (i < zf) stop = i;
          else if (zf < cnt) stop = zf;
          prevR = r0;
        } else {  // styTangent: the result of the run's last accepting position
          if (zf > q) {
            prevR = resultAt(zf - 1);
            curEnd = base + zf;
          }
          if (zf < cnt) stop = zf;
        }
        if (stop != 0xffffffffu) {
          result = prevR;
          if (POS) startOut = startAt(stop);
          mode = 2;
        }
      }
      if (esc) matchStart = base + 63 - uint32_t(__builtin_clzll(esc));
    }
    if (style == kStyLast) {
      if (n) result = prevR;
      startOut = matchStart;
    } else if (style == kStyFull) {
      if (n) result = s >= d.firstAccept ? ldsRes[s] : 0;
      startOut = matchStart;
      curEnd = n;  // end is the line length when the final state accepts (Matcher.h:463)
    } else if (mode == 1) {  // the line ended inside the run
      result = prevR;
      startOut = matchStart;
    }
    b.result[line] = result;
    if (POS) {
      if (b.start) b.start[line] = result ? startOut : 0;
      if (b.end) b.end[line] = result ? curEnd : 0;
    }
  }
}
