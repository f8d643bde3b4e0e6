// k_misc.h - k_advance (StatefulMatcher chunks), k_replace + the exclusive scan of its lengths, k_visits
// (redgpu_dfa_tune's histogram), k_walked (bytes an early-exit walk reads)
// (included by kernels.hip inside namespace redgpu { namespace { ... } }; see its file map).
#pragma once

// StatefulMatcher::advance (include/Matcher.h:770-792, lib/Matcher.cpp:106-158) over a whole
// chunk per line: state[line] is the matcher's state_ (a device state index; REDGPU_STATE_INITIAL
// = a freshly constructed matcher, lib/Matcher.cpp:113-136), advanced by every byte of the
// chunk with no early exit and no style rules, then stored back; result[line] = result() after
// the last byte (= the state's result; for an empty chunk the current state's).
template <int KIND, int kThreads>
__global__ void __launch_bounds__(kThreads)
k_advance(DevDfa d, Batch b, uint32_t *state) {
  extern __shared__ __align__(16) uint8_t lds[];
  uint8_t *eq = lds;
  uint8_t *leader = lds + 256;
  const Tab<KIND> tab = stageTab<KIND, kThreads>(d, lds);
  LaneCtx c{eq, leader, resOf<KIND>(d, lds), d.init, d.leaderNext, d.nPureDead, d.firstAccept,
            d.leaderLen};
  c.startWord[0] = d.startFreeWord; c.startCount[0] = d.startFreeCount;
  c.startWord[1] = d.startLeadWord; c.startCount[1] = d.startLeadCount;
  c.start2Word[0] = d.start2FreeWord; c.start2Count[0] = d.start2FreeCount;
  c.start2Word[1] = d.start2LeadWord; c.start2Count[1] = d.start2LeadCount;
  c.suffixClosed = d.suffixClosed;
  const uint64_t step = uint64_t(gridDim.x) * kThreads;
  for (uint64_t line = uint64_t(blockIdx.x) * kThreads + threadIdx.x; line < b.n; line += step) {
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
    uint32_t s = state[line];
    if (s >= d.nStates) s = d.init;  // REDGPU_STATE_INITIAL (and any token that is not ours)
    if (d.deadAbsorbing && d.earlyDeath) {
      // an absorbing dead end stays put: stop reading (anchored patterns die in their first bytes)
      walkBytes(p, 0, n, [&](uint32_t byte, uint64_t) {
        s = tab.next(s, byte);
        return s >= d.nPureDead;
      });
    } else {
      walkAllBytes(p, n, [&](uint32_t byte, uint64_t) { s = tab.next(s, byte); });
    }
    state[line] = s;
    b.result[line] = c.resultOf(s);
  }
}

// include/Matcher.h:643-706 replaceCore.  out == nullptr: only count and measure.
// Returns the number of replacements; outLen = length of the rewritten line.
template <class T>
__device__ uint64_t replaceLane(const T &tab, const LaneCtx &c, const uint8_t *p, uint64_t n,
                                int style, bool lead, const uint8_t *repl, uint64_t replLen,
                                uint64_t max, uint8_t *out, uint64_t &outLen) {
  uint64_t cnt = 0, w = 0, in = 0;
  while (in < n) {
    if (cnt >= max) {
      if (out)
        for (uint64_t k = in; k < n; ++k) out[w + (k - in)] = p[k];
      w += n - in;
      break;
    }
    uint64_t found = ~0ull;
    bool toEnd = false;  // the attempt read p[in..n) to its end
    if (!lead || lookingAt(c, p, in, n)) {
      uint32_t s = c.init;
      int32_t prev = 0;
      uint64_t q = in;
      for (; q < n; ++q) {
        s = tab.next(s, p[q]);
        if (s >= c.firstAccept) {
          const int32_t r = c.res[s];
          if (style == kStyFirst) {
            if (prev && r != prev) break;
            prev = r;
          }
          found = q;
          if (style == kStyInstant) break;
        } else {
          if (style == kStyFull) found = ~0ull;
          if (((style == kStyFirst || style == kStyTangent) && found != ~0ull) ||
              s < c.nPureDead)
            break;
        }
      }
      toEnd = q == n;
    }
    if (found == ~0ull && toEnd && c.suffixClosed && !lead) {
      // L = SIGMA* L: nothing matched on p[in..n), so nothing can at any later position (they read
      // suffixes of it) - the rest of the line is copied as the reference's loop would, byte by byte
      if (out)
        for (uint64_t k = in; k < n; ++k) out[w + (k - in)] = p[k];
      w += n - in;
      break;
    }
    if (found != ~0ull) {
      if (out)
        for (uint64_t k = 0; k < replLen; ++k) out[w + k] = repl[k];
      w += replLen;
      in = found + 1;
      ++cnt;
    } else {
      if (out) out[w] = p[in];
      ++w;
      ++in;
    }
  }
  outLen = w;
  return cnt;
}

// pass 1 (out == nullptr): counts[line], outLens[line].  pass 2: writes line i's rewritten
// bytes at out + outOffsets[i] when outOffsets[i + 1] <= outCap.
template <int KIND, int kThreads>
__global__ void __launch_bounds__(kThreads)
k_replace(DevDfa d, Batch b, int style, int lead, const uint8_t *repl, uint64_t replLen,
          uint64_t max, uint64_t *counts, uint64_t *outLens, const uint64_t *outOffsets,
          uint8_t *out, uint64_t outCap) {
  extern __shared__ __align__(16) uint8_t lds[];
  uint8_t *eq = lds;
  uint8_t *leader = lds + 256;
  const Tab<KIND> tab = stageTab<KIND, kThreads>(d, lds);
  LaneCtx c{eq, leader, resOf<KIND>(d, lds), d.init, d.leaderNext, d.nPureDead, d.firstAccept,
            d.leaderLen};
  c.startWord[0] = d.startFreeWord; c.startCount[0] = d.startFreeCount;
  c.startWord[1] = d.startLeadWord; c.startCount[1] = d.startLeadCount;
  c.start2Word[0] = d.start2FreeWord; c.start2Count[0] = d.start2FreeCount;
  c.start2Word[1] = d.start2LeadWord; c.start2Count[1] = d.start2LeadCount;
  c.suffixClosed = d.suffixClosed;
  const uint64_t step = uint64_t(gridDim.x) * kThreads;
  for (uint64_t line = uint64_t(blockIdx.x) * kThreads + threadIdx.x; line < b.n; line += step) {
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
    uint64_t len = 0;
    if (!out) {
      counts[line] = replaceLane(tab, c, p, n, style, lead != 0, repl, replLen, max, nullptr, len);
      outLens[line] = len;
    } else if (outOffsets[line + 1] <= outCap) {
      replaceLane(tab, c, p, n, style, lead != 0, repl, replLen, max, out + outOffsets[line], len);
    }
  }
}

// exclusive scan of lens[n] into offs[n + 1] (offs[n] = total): per-1024 partial sums, one
// workgroup over the partials, then the fill
__global__ void __launch_bounds__(256)
k_scan_partials(const uint64_t *lens, uint64_t n, uint64_t *partials) {
  __shared__ uint64_t ws[4];
  const uint64_t base = uint64_t(blockIdx.x) * 1024;
  uint64_t v = 0;
  for (uint32_t k = 0; k < 4; ++k) {
    const uint64_t i = base + k * 256 + threadIdx.x;
    v += i < n ? lens[i] : 0;
  }
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ void __launch_bounds__(1024)
k_scan_tops(uint64_t *partials, uint64_t nPart) {
  __shared__ uint64_t part[1024];
  const uint64_t per = (nPart + 1023) / 1024;
  const uint64_t lo = uint64_t(threadIdx.x) * per;
  const uint64_t hi = lo + per < nPart ? lo + per : nPart;
  uint64_t sum = 0;
  for (uint64_t i = lo; i < hi; ++i) sum += partials[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t run = 0;
    for (int i = 0; i < 1024; ++i) { const uint64_t v = part[i]; part[i] = run; run += v; }
  }
  __syncthreads();
  uint64_t run = part[threadIdx.x];
  for (uint64_t i = lo; i < hi; ++i) { const uint64_t v = partials[i]; partials[i] = run; run += v; }
}

__global__ void __launch_bounds__(256)
k_scan_fill(const uint64_t *lens, uint64_t n, const uint64_t *partials, uint64_t *offs) {
  // one wave per 256 elements would do; keep it simple: thread 0 of each 64-lane group walks
  // its 64 elements after a wave-level prefix
  const uint64_t base = uint64_t(blockIdx.x) * 1024;
  __shared__ uint64_t ws[4];
  uint64_t carry = partials[blockIdx.x];
  for (uint32_t k = 0; k < 4; ++k) {
    const uint64_t i = base + k * 256 + threadIdx.x;
    const uint64_t v = i < n ? lens[i] : 0;
    uint64_t incl = v;
    for (int o = 1; o < 64; o <<= 1) {
      const uint64_t u = __shfl_up(incl, o);
      if ((threadIdx.x & 63) >= uint32_t(o)) incl += u;
    }
    if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t wb = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) wb += ws[w];
    if (i < n) offs[i] = carry + wb + incl - v;
    if (i + 1 == n) offs[n] = carry + wb + incl;
    carry += ws[0] + ws[1] + ws[2] + ws[3];
    __syncthreads();
  }
}

// Visit histogram for redgpu_dfa_tune: the anchored walk of match<styLast,false> over every
// line of a SAMPLE, hist[state] += 1 per byte consumed.  A profiling pass, not a hot path:
// plain global atomics.
template <int KIND, int kThreads>
__global__ void __launch_bounds__(kThreads)
k_visits(DevDfa d, Batch b, uint32_t *hist) {
  extern __shared__ __align__(16) uint8_t lds[];
  const Tab<KIND> tab = stageTab<KIND, kThreads>(d, lds);
  const uint64_t step = uint64_t(gridDim.x) * kThreads;
  for (uint64_t line = uint64_t(blockIdx.x) * kThreads + threadIdx.x; line < b.n; line += step) {
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
    uint32_t s = d.init;
    walkBytes(p, 0, n, [&](uint32_t byte, uint64_t) {
      s = tab.next(s, byte);
      atomicAdd(&hist[s], 1u);
      return s >= d.nPureDead;
    });
  }
}

// bench.py's "bytes actually walked": what the loop of match<styLast,lead> (include/Matcher.h:
// 424-479) consumes per line - nothing when the leader peek fails, else every byte up to and
// including the one that reaches a pure dead end.  One atomic per wave.
template <int KIND, int kThreads>
__global__ void __launch_bounds__(kThreads)
k_walked(DevDfa d, Batch b, int lead, unsigned long long *walked) {
  extern __shared__ __align__(16) uint8_t lds[];
  const Tab<KIND> tab = stageTab<KIND, kThreads>(d, lds);
  LaneCtx c;
  c.eq = lds;
  c.leader = lds + 256;
  c.res = resOf<KIND>(d, lds);
  c.init = d.init; c.leaderNext = d.leaderNext; c.nPureDead = d.nPureDead;
  c.firstAccept = d.firstAccept; c.leaderLen = d.leaderLen;
  const uint64_t step = uint64_t(gridDim.x) * kThreads;
  unsigned long long mine = 0;
  for (uint64_t line = uint64_t(blockIdx.x) * kThreads + threadIdx.x; line < b.n; line += step) {
    const uint8_t *p;
    uint64_t n;
    if (b.offsets) {
      const uint64_t o = b.offsets[line];
      p = b.data + o;
      n = b.offsets[line + 1] - o;
      n = n >= b.stride ? n - b.stride : 0;
    } else {
      p = b.data + line * b.stride;
      n = b.stride;
    }
    if (lead && !lookingAt(c, p, 0, n)) continue;
    uint32_t s = d.init;
    walkBytes(p, 0, n, [&](uint32_t byte, uint64_t) {
      s = tab.next(s, byte);
      ++mine;
      return s >= d.nPureDead;
    });
  }
  for (int o = 32; o; o >>= 1) mine += __shfl_xor(mine, o);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(walked, mine);
}
