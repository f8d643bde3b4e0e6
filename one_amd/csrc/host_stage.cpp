// host_stage.cpp - per-thread staging for the host-buffer entry points, and the per-thread
// scratch pool of the ragged / chunked launches (see host_stage.h, kernels.h).
#include "host_stage.h"

#include <atomic>
#include <cstring>
#include <memory>
#include <vector>

#include "kernels.h"

namespace redgpu {

// ---- scratch pool ---------------------------------------------------------------------------
// Device scratch of the ragged / chunked / replace launches (tail pad, permutation, histograms,
// chunk records).  Those are multi-kernel sequences that carry state in the buffer from one
// launch to the next, so a buffer must belong to ONE caller: the pool is per HOST THREAD, keyed
// by (device, stream) - work a thread queues on one stream runs in order, so its next call on
// that stream may reuse the buffer; another stream, or another thread on the same stream, gets
// its own (round 2 shared one buffer per (device, stream) process-wide: two threads on the
// default stream overwrote each other's tail pads, and growing the buffer freed a pointer the
// other thread had not launched with yet).  Bounded per thread (least recently used entry
// freed beyond kMaxPerThread), freed at thread exit and by redgpu_thread_release(); entries of
// a stream this library created are dropped with the stream.
namespace {

struct ScratchEntry {
  int dev;
  hipStream_t stream;
  void *ptr;
  size_t bytes;
  uint64_t stamp;
  uint32_t *ctl = nullptr;  // two slots of 8 words (scratchCtlFor)
  uint32_t ctlCalls = 0;
};
constexpr size_t kMaxPerThread = 48;
std::atomic<size_t> gScratchCount{0};  // entries alive in all threads (redgpu_scratch_entries)

void freeEntry(ScratchEntry &en) {
  if (!en.ptr) return;
  int cur = -1;
  const bool sw = hipGetDevice(&cur) == hipSuccess && cur != en.dev &&
                  hipSetDevice(en.dev) == hipSuccess;
  // the buffer may still be read by kernels queued on its stream (which may itself be gone by
  // now, so it cannot be asked): wait for the device, then free.  Only on the rare paths - a
  // buffer outgrown, an entry evicted, a thread leaving.  (Round 2 relied on hipFree alone to
  // wait; with evictions no longer rare a kernel faulted on its freed tail pad.)
  (void)hipDeviceSynchronize();
  (void)hipFree(en.ptr);
  if (sw) (void)hipSetDevice(cur);
  en.ptr = nullptr;
  en.bytes = 0;
}

// (with the entry itself: eviction, a dropped stream, the thread leaving)
void freeCtl(ScratchEntry &en) {
  if (!en.ctl) return;
  int cur = -1;
  const bool sw = hipGetDevice(&cur) == hipSuccess && cur != en.dev &&
                  hipSetDevice(en.dev) == hipSuccess;
  (void)hipDeviceSynchronize();
  (void)hipFree(en.ctl);
  if (sw) (void)hipSetDevice(cur);
  en.ctl = nullptr;
}

struct ThreadScratch {
  std::vector<ScratchEntry> v;
  uint64_t stamp = 0;
  void clear() {
    for (auto &en : v) { freeEntry(en); freeCtl(en); }
    gScratchCount -= v.size();
    v.clear();
  }
  ~ThreadScratch() { clear(); }
};
thread_local ThreadScratch tlsScratch;

}  // namespace

hipError_t scratchFor(hipStream_t stream, size_t bytes, void **out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  ThreadScratch &ts = tlsScratch;
  ScratchEntry *slot = nullptr;
  for (auto &en : ts.v)
    if (en.dev == dev && en.stream == stream) slot = &en;
  if (slot && slot->bytes >= bytes) {
    slot->stamp = ++ts.stamp;
    *out = slot->ptr;
    return hipSuccess;
  }
  if (!slot) {
    if (ts.v.size() >= kMaxPerThread) {
      size_t lru = 0;
      for (size_t i = 1; i < ts.v.size(); ++i)
        if (ts.v[i].stamp < ts.v[lru].stamp) lru = i;
      freeEntry(ts.v[lru]);
      freeCtl(ts.v[lru]);
      ts.v.erase(ts.v.begin() + long(lru));
      --gScratchCount;
    }
    ts.v.push_back(ScratchEntry{dev, stream, nullptr, 0, 0});
    ++gScratchCount;
    slot = &ts.v.back();
  }
  freeEntry(*slot);
  const size_t want = bytes + bytes / 2 + 4096;
  e = hipMalloc(&slot->ptr, want);
  if (e != hipSuccess) {
    slot->ptr = nullptr;
    freeCtl(*slot);
    ts.v.erase(ts.v.begin() + (slot - ts.v.data()));
    --gScratchCount;
    return e;
  }
  slot->bytes = want;
  slot->stamp = ++ts.stamp;
  *out = slot->ptr;
  return hipSuccess;
}

hipError_t scratchCtlFor(hipStream_t stream, uint32_t **out, uint32_t **next) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  ThreadScratch &ts = tlsScratch;
  ScratchEntry *slot = nullptr;
  for (auto &en : ts.v)
    if (en.dev == dev && en.stream == stream) slot = &en;
  if (!slot) return hipErrorInvalidValue;  // (scratchFor first: it makes the entry)
  if (!slot->ctl) {
    e = hipMalloc(reinterpret_cast<void **>(&slot->ctl), 64);
    if (e != hipSuccess) { slot->ctl = nullptr; return e; }
    // once per entry, ON THE ENTRY'S STREAM: a hipMemset goes to the null stream, which the
    // caller's non-blocking stream does not wait for - the pre-pass ran ahead of it once in a
    // few hundred first calls, the memset then wiped what it had written (the threshold, so
    // every line was "passed over"), and one worker of the 8-thread test got stale answers
    e = hipMemsetAsync(slot->ctl, 0, 64, stream);
    if (e != hipSuccess) return e;
    slot->ctlCalls = 0;
  }
  const uint32_t k = slot->ctlCalls++ & 1u;
  *out = slot->ctl + 8 * k;
  *next = slot->ctl + 8 * (k ^ 1u);
  return hipSuccess;
}

void scratchDrop(int device, hipStream_t stream) {
  ThreadScratch &ts = tlsScratch;
  for (size_t i = 0; i < ts.v.size();) {
    if (ts.v[i].dev == device && ts.v[i].stream == stream) {
      freeEntry(ts.v[i]);
      freeCtl(ts.v[i]);
      ts.v.erase(ts.v.begin() + long(i));
      --gScratchCount;
    } else {
      ++i;
    }
  }
}

void scratchReleaseThread() { tlsScratch.clear(); }

size_t scratchEntries() { return gScratchCount.load(); }

// ---- per-thread stages -------------------------------------------------------------------
hipError_t HostStage::get(int slot, size_t bytes, void **out) {
  Buf &b = bufs_[slot];
  if (b.cap >= bytes + 16 && b.p) {
    *out = b.p;
    return hipSuccess;
  }
  hipError_t e = sync();
  if (e != hipSuccess) return e;
  if (b.p) {
    (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
  }
  const size_t want = bytes + bytes / 4 + 4096;
  e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    b.p = nullptr;
    return e;
  }
  b.cap = want;
  *out = b.p;
  return hipSuccess;
}

hipError_t HostStage::sync() {
  for (hipStream_t s : streams) {
    if (!s) continue;
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
  }
  flush(-1);
  bounceUsed_ = 0;  // nothing of the arena is in flight any more, either way
  return hipSuccess;
}

hipError_t HostStage::syncStream(int idx) {
  hipError_t e = hipStreamSynchronize(streams[idx]);
  if (e != hipSuccess) return e;
  flush(idx);
  return hipSuccess;
}

void HostStage::flush(int idx) {
  size_t keep = 0;
  for (size_t i = 0; i < pending_.size(); ++i) {
    const Pending &p = pending_[i];
    if (idx < 0 || p.idx == idx) memcpy(p.dst, p.src, p.bytes);
    else pending_[keep++] = p;
  }
  pending_.resize(keep);
  keep = 0;
  for (size_t i = 0; i < tempPins_.size(); ++i) {
    if (idx < 0 || tempPins_[i].idx == idx) (void)hipHostUnregister(tempPins_[i].p);
    else tempPins_[keep++] = tempPins_[i];
  }
  tempPins_.resize(keep);
}

// is p host memory the runtime knows as pinned (hipHostMalloc / hipHostRegister)?
bool isPinnedHost(const void *p) {
  if (!p) return false;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // plain pageable memory: not an error of the call
    return false;
  }
  return a.type == hipMemoryTypeHost;
}

bool HostStage::pinForCall(const void *p, size_t bytes, int idx) {
  if (hipHostRegister(const_cast<void *>(p), bytes, hipHostRegisterDefault) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  tempPins_.push_back(TempPin{const_cast<void *>(p), idx});
  return true;
}

// `bytes` of the pinned arena (64-byte aligned), or nullptr when the transfer should go direct.
// A full arena waits for what is in flight and starts over; one that is too small is replaced
// (only ever with nothing in flight).
void *HostStage::bounceTake(size_t bytes) {
  const size_t need = (bytes + 63) & ~size_t(63);
  if (bounceUsed_ + need > bounceCap_) {
    if (bounceUsed_ && sync() != hipSuccess) return nullptr;  // (sync() empties the arena)
    bounceUsed_ = 0;
    if (need > bounceCap_) {
      if (bounce_) (void)hipHostFree(bounce_);
      bounce_ = nullptr;
      bounceCap_ = 0;
      size_t want = need + need / 2;
      if (want < (size_t(1) << 20)) want = size_t(1) << 20;
      if (hipHostMalloc(&bounce_, want, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        bounce_ = nullptr;
        return nullptr;
      }
      bounceCap_ = want;
    }
  }
  void *p = static_cast<uint8_t *>(bounce_) + bounceUsed_;
  bounceUsed_ += need;
  return p;
}

// how a transfer of caller memory goes: 0 = as it is (pinned, or nothing else is possible),
// otherwise through `*slice` of the arena
std::atomic<uint64_t> gRoute[4];  // transfers: caller-pinned, through the arena, registered, pageable

void hostRouteCounts(uint64_t out[4]) {
  for (int i = 0; i < 4; ++i) out[i] = gRoute[i].load();
}

// One transfer of caller memory, either direction (toDevice: dev <- host).
//  * memory the caller pinned (`direct`, or the runtime says so for both ends of the range): as it is;
//  * a small call (beginCall) or a small transfer: through the arena;
//  * otherwise the WHOLE PAGES inside the range are registered for the call and copied as they
//    are, and the partial pages at its two ends go through the arena.  Registration is by page:
//    two buffers that share a page - numpy's result / start / end arrays sit back to back in the
//    heap - registered one after the other and released one after the other left a page
//    unmapped under a copy that still needed it (the "Memory access fault ... on address
//    0x5a95..." of round 3, before and after the explicit registration).  Registering only what
//    a buffer owns alone cannot collide;
//  * what the runtime refuses to register goes through the arena piece by piece.  Nothing is
//    ever handed over pageable.
hipError_t HostStage::move(void *dev, void *host, size_t bytes, int idx, bool direct, bool toDevice) {
  if (bytes == 0) return hipSuccess;
  hipStream_t s = streams[idx];
  auto plain = [&](void *d, void *h, size_t n) {
    return toDevice ? hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s)
                    : hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s);
  };
  // through the arena, in pieces the arena can hold (bounceTake waits and starts over when full)
  auto bounced = [&](uint8_t *d, uint8_t *h, size_t n) -> hipError_t {
    constexpr size_t kPiece = size_t(4) << 20;
    for (size_t at = 0; at < n; at += kPiece) {
      const size_t m = n - at < kPiece ? n - at : kPiece;
      void *p = bounceTake(m);
      if (!p) return hipErrorOutOfMemory;
      hipError_t e;
      if (toDevice) {
        memcpy(p, h + at, m);
        e = hipMemcpyAsync(d + at, p, m, hipMemcpyHostToDevice, s);
      } else {
        e = hipMemcpyAsync(p, d + at, m, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) pending_.push_back(Pending{h + at, p, m, idx});
      }
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  };
  uint8_t *d8 = static_cast<uint8_t *>(dev), *h8 = static_cast<uint8_t *>(host);
  if (!direct && !callDirect_ && bytes <= kBounceMax) { ++gRoute[1]; return bounced(d8, h8, bytes); }
  if (direct || (isPinnedHost(h8) && isPinnedHost(h8 + bytes - 1))) { ++gRoute[0]; return plain(dev, host, bytes); }
  constexpr uintptr_t kPage = 4096;
  const uintptr_t a = reinterpret_cast<uintptr_t>(h8);
  const uintptr_t lo = (a + kPage - 1) & ~(kPage - 1), hi = (a + bytes) & ~(kPage - 1);
  if (hi > lo && hi - lo >= 16 * kPage && pinForCall(reinterpret_cast<void *>(lo), hi - lo, idx)) {
    ++gRoute[2];
    const size_t head = lo - a, tail = a + bytes - hi;
    hipError_t e = plain(d8 + head, reinterpret_cast<void *>(lo), hi - lo);
    if (e == hipSuccess && head) e = bounced(d8, h8, head);
    if (e == hipSuccess && tail) e = bounced(d8 + (bytes - tail), h8 + (bytes - tail), tail);
    return e;
  }
  ++gRoute[1];
  return bounced(d8, h8, bytes);
}

hipError_t HostStage::copyIn(void *dDst, const void *hSrc, size_t bytes, int idx, bool direct) {
  return move(dDst, const_cast<void *>(hSrc), bytes, idx, direct, true);
}

hipError_t HostStage::copyOut(void *hDst, const void *dSrc, size_t bytes, int idx, bool direct) {
  return move(const_cast<void *>(dSrc), hDst, bytes, idx, direct, false);
}

void HostStage::release() {
  if (device < 0) return;
  int prev = -1;
  const bool sw = hipGetDevice(&prev) == hipSuccess && prev != device &&
                  hipSetDevice(device) == hipSuccess;
  (void)sync();
  for (Buf &b : bufs_) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
  }
  if (bounce_) (void)hipHostFree(bounce_);
  bounce_ = nullptr;
  bounceCap_ = bounceUsed_ = 0;
  pending_.clear();
  for (const TempPin &t : tempPins_) (void)hipHostUnregister(t.p);
  tempPins_.clear();
  for (hipStream_t &s : streams) {
    if (s) {
      scratchDrop(device, s);
      (void)hipStreamDestroy(s);
    }
    s = nullptr;
  }
  if (ready) (void)hipEventDestroy(ready);
  ready = nullptr;
  if (sw) (void)hipSetDevice(prev);
  device = -1;
}

namespace {
struct ThreadStages {
  std::vector<std::unique_ptr<HostStage>> v;
};
thread_local ThreadStages tlsStages;
}  // namespace

hipError_t hostStage(int device, HostStage **out) {
  for (auto &s : tlsStages.v)
    if (s->device == device) {
      *out = s.get();
      return hipSuccess;
    }
  auto st = std::make_unique<HostStage>();
  for (hipStream_t &s : st->streams) {
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) {
      st->device = device;  // so that release() destroys what exists
      return e;
    }
  }
  hipError_t e = hipEventCreateWithFlags(&st->ready, hipEventDisableTiming);
  st->device = device;
  if (e != hipSuccess) return e;
  *out = st.get();
  tlsStages.v.push_back(std::move(st));
  return hipSuccess;
}

void hostStageReleaseThread() {
  tlsStages.v.clear();
  scratchReleaseThread();
}

ScopedPin::ScopedPin(const void *ptr, size_t bytes, bool enable) {
  if (!enable || !ptr || !bytes) return;
  if (hipHostRegister(const_cast<void *>(ptr), bytes, hipHostRegisterDefault) == hipSuccess)
    p = const_cast<void *>(ptr);
  else
    (void)hipGetLastError();  // not an error of the call: the copies just stay synchronous
}

ScopedPin::~ScopedPin() {
  if (p) (void)hipHostUnregister(p);
}

}  // namespace redgpu
