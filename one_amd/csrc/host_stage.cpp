// host_stage.cpp - per-thread staging for the host-buffer entry points, and the process-wide
// scratch pool of the ragged / chunked launches (see host_stage.h, kernels.h).
#include "host_stage.h"

#include <memory>
#include <mutex>
#include <vector>

#include "kernels.h"

namespace redgpu {

// ---- scratch pool ---------------------------------------------------------------------------
// Device scratch of the ragged launches (tail pad, permutation, histograms, chunk records),
// keyed by (device, stream): work queued on one stream runs in order, so the next launch on that
// stream may reuse the buffer the previous one used; another stream gets its own.  Round 1 kept
// these in thread_local slots that were never freed at thread exit; the pool is
// process-wide, bounded (least recently used entry freed beyond kMaxEntries) and entries of a
// stream this library created are dropped with that stream.
namespace {

struct ScratchEntry {
  int dev;
  hipStream_t stream;
  void *ptr;
  size_t bytes;
  uint64_t stamp;
};
constexpr size_t kMaxEntries = 64;
std::mutex gScratchMutex;
std::vector<ScratchEntry> gScratch;
uint64_t gStamp = 0;

}  // namespace

hipError_t scratchFor(hipStream_t stream, size_t bytes, void **out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(gScratchMutex);
  ScratchEntry *slot = nullptr;
  for (auto &en : gScratch)
    if (en.dev == dev && en.stream == stream) slot = &en;
  if (slot && slot->bytes >= bytes) {
    slot->stamp = ++gStamp;
    *out = slot->ptr;
    return hipSuccess;
  }
  if (!slot) {
    if (gScratch.size() >= kMaxEntries) {
      size_t lru = 0;
      for (size_t i = 1; i < gScratch.size(); ++i)
        if (gScratch[i].stamp < gScratch[lru].stamp) lru = i;
      if (gScratch[lru].dev == dev) {
        (void)hipFree(gScratch[lru].ptr);  // waits for the work that may still be using it
      } else {
        int prev = dev;
        (void)hipSetDevice(gScratch[lru].dev);
        (void)hipFree(gScratch[lru].ptr);
        (void)hipSetDevice(prev);
      }
      gScratch.erase(gScratch.begin() + long(lru));
    }
    gScratch.push_back(ScratchEntry{dev, stream, nullptr, 0, 0});
    slot = &gScratch.back();
  }
  if (slot->ptr) {
    (void)hipFree(slot->ptr);
    slot->ptr = nullptr;
    slot->bytes = 0;
  }
  const size_t want = bytes + bytes / 2 + 4096;
  e = hipMalloc(&slot->ptr, want);
  if (e != hipSuccess) {
    slot->ptr = nullptr;
    gScratch.erase(gScratch.begin() + (slot - gScratch.data()));
    return e;
  }
  slot->bytes = want;
  slot->stamp = ++gStamp;
  *out = slot->ptr;
  return hipSuccess;
}

void scratchDrop(int device, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(gScratchMutex);
  for (size_t i = 0; i < gScratch.size();) {
    if (gScratch[i].dev == device && gScratch[i].stream == stream) {
      (void)hipFree(gScratch[i].ptr);
      gScratch.erase(gScratch.begin() + long(i));
    } else {
      ++i;
    }
  }
}

size_t scratchEntries() {
  std::lock_guard<std::mutex> lock(gScratchMutex);
  return gScratch.size();
}

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
  return hipSuccess;
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

void hostStageReleaseThread() { tlsStages.v.clear(); }

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
