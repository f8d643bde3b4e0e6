// host_stage.h - what the HOST-buffer entry points of include/redgpu.h stage through.
//
// The reference's matchers take caller memory and touch nothing else
// (/root/reference/quol/red/doc/Performance.md:81-84: read-only, lock-free, re-entrant; N threads
// over one Executable in tools/thr_red.cpp:84-91).  A device needs staging, so each host thread
// keeps, per device it has used: two private non-blocking streams and a small set of grow-only
// device buffers.  Nothing is allocated or freed per call once the buffers have reached the
// thread's batch size (round 1 paid five hipMalloc + five hipFree - each hipFree a device-wide
// synchronisation that stalled every other thread's streams - and a stream create/destroy per
// call).  The cache dies with its thread (thread_local destructor) or on redgpu_thread_release().
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

#include <hip/hip_runtime.h>

namespace redgpu {

class HostStage {
 public:
  static constexpr int kBufs = 14;
  int device = -1;
  hipStream_t streams[2] = {nullptr, nullptr};
  hipEvent_t ready = nullptr;  // "the shared uploads (offsets, replacement) have landed"

  // grow-only device buffer `slot` of at least `bytes` (+16: 16-byte loads of a last line never
  // leave the allocation).  Growing waits for this thread's own streams, then frees and allocates.
  hipError_t get(int slot, size_t bytes, void **out);
  hipError_t sync();  // both streams; completes the copyOut()s behind them
  void release();
  ~HostStage() { release(); }

  // Caller memory <-> device, asynchronously on streams[idx].  Pageable caller memory of up to
  // kBounceMax bytes per transfer, in calls whose input is below kDirectFrom bytes, goes through
  // this thread's own PINNED arena (a CPU copy in front of the upload / behind the download; a
  // call of 8 MiB and more is a throughput call and keeps the runtime's own path - the copy in
  // the middle cost 2^20 x 64-byte lines 40.9 -> 25.8 GB/s): handed to hipMemcpyAsync as it is, pageable
  // memory above ~1 MiB is pinned by the runtime on the fly and the pinning cached by address -
  // when the caller's allocator has meanwhile returned part of that range to the system (a
  // heap that shrank) the next copy over the cached pin faults on the GPU (seen in round 3 as
  // an intermittent "Memory access fault" at a host heap address in the test suite, whose numpy
  // buffers come and go).  Larger transfers and memory the caller pinned go direct.
  // A throughput call's pageable buffers are REGISTERED for the duration of the call instead
  // (hipHostRegister now, hipHostUnregister behind the stream's next drain) - the whole pages
  // inside each buffer only, its partial end pages go through the arena (HostStage::move has the
  // why) - so the runtime's own on-the-fly path is never taken.  A range the runtime refuses to
  // register goes through the arena in pieces.
  // copyOut()'s bytes are in the caller's buffer after syncStream(idx) / sync().
  static constexpr size_t kBounceMax = size_t(16) << 20;
  static constexpr size_t kDirectFrom = size_t(8) << 20;
  hipError_t copyIn(void *dDst, const void *hSrc, size_t bytes, int idx, bool direct = false);
  hipError_t copyOut(void *hDst, const void *dSrc, size_t bytes, int idx, bool direct = false);
  hipError_t syncStream(int idx);
  // at the top of a host-buffer entry point (the one before left with its streams drained);
  // inputBytes = what the call uploads
  void beginCall(size_t inputBytes) {
    if (pending_.empty()) bounceUsed_ = 0;
    callDirect_ = inputBytes >= kDirectFrom;
  }
  bool callDirect() const { return callDirect_; }

 private:
  void *bounceTake(size_t bytes);
  hipError_t move(void *dev, void *host, size_t bytes, int idx, bool direct, bool toDevice);
  bool pinForCall(const void *p, size_t bytes, int idx);
  struct TempPin {
    void *p;
    int idx;
  };
  std::vector<TempPin> tempPins_;
  void flush(int idx);  // the finished downloads of stream idx (-1: all) into the caller's buffers
  bool callDirect_ = false;
  void *bounce_ = nullptr;
  size_t bounceCap_ = 0, bounceUsed_ = 0;
  struct Pending {
    void *dst;
    const void *src;
    size_t bytes;
    int idx;
  };
  std::vector<Pending> pending_;

  struct Buf {
    void *p = nullptr;
    size_t cap = 0;
  } bufs_[kBufs];
  friend hipError_t hostStage(int, HostStage **);
};

// is p host memory the runtime knows as pinned (hipHostMalloc / hipHostRegister)?
bool isPinnedHost(const void *p);
// transfers of caller memory so far, by route: pinned by the caller, through a pinned arena,
// registered for the call, handed over pageable (redgpu_host_route_counts)
void hostRouteCounts(uint64_t out[4]);

// The calling thread's stage for `device` (created on first use; the device must be current).
hipError_t hostStage(int device, HostStage **out);
// Frees every stage the calling thread holds.
void hostStageReleaseThread();

// Pins caller memory for the duration of one call so that copies on two streams overlap
// (H2D of one chunk beside D2H of the previous one); silently does nothing if the runtime
// refuses the range (read-only mappings, already registered memory).
struct ScopedPin {
  void *p = nullptr;
  ScopedPin(const void *ptr, size_t bytes, bool enable);
  ~ScopedPin();
  bool pinned() const { return p != nullptr; }
};

}  // namespace redgpu
