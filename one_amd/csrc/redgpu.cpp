// redgpu.cpp - the C-ABI of include/redgpu.h over the gfx950 kernels.  Host C++ (built with
// hipcc for the HIP runtime API).  There is no CPU compute path in this file or behind it:
// every batch entry point either launches a kernel or fails.
#include "../../include/redgpu.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include <hip/hip_runtime.h>

#include "dfa_image.h"
#include "host_stage.h"
#include "kernels.h"
#include "redgpu_internal.h"

using namespace redgpu;

namespace {

int checkStyle(int style) {
  if (style < REDGPU_STY_INSTANT || style > REDGPU_STY_FULL)
    return fail(REDGPU_EEXEC, "unsupported style");  // lib/Matcher.cpp:45
  return REDGPU_OK;
}

LaunchCfg cfgOfFlags(const redgpu_dfa *dfa, uint32_t extraFlags) {
  return LaunchCfg{dfa->numCUs, ((dfa->flags | extraFlags) & REDGPU_F_FORCE_GENERIC) ? 1 : 0,
                   (dfa->flags & REDGPU_F_NO_BUCKETING) ? 1 : 0,
                   (dfa->flags & REDGPU_F_FORCE_STREAM) ? 1 : 0,
                   (dfa->flags & REDGPU_F_NO_CHUNKING) ? 1 : 0,
                   (dfa->flags & REDGPU_F_FORCE_CHUNKING) ? 1 : 0,
                   (dfa->flags & REDGPU_F_FORCE_EARLY) ? 1 : 0,
                   (dfa->flags & REDGPU_F_FORCE_LEAN) ? 1 : 0,
                   (dfa->flags & REDGPU_F_LEAN_CHAINS_4) ? 1 : 0,
                   (dfa->flags & REDGPU_F_STREAM_CHAINS_2) ? 2
                   : (dfa->flags & REDGPU_F_STREAM_CHAINS_4) ? 4 : 0};
}

LaunchCfg cfgOf(const redgpu_dfa *dfa, uint32_t extraFlags = 0) {
  LaunchCfg cfg = cfgOfFlags(dfa, extraFlags);
  cfg.forcePieces = (dfa->flags & REDGPU_F_FORCE_PIECES) ? 1 : 0;
  return cfg;
}

int runDev(const redgpu_dfa *dfa, int verb, int style, int doLeader, const uint8_t *data,
           const uint64_t *offsets, uint64_t stride, uint64_t n, int32_t *result,
           uint64_t *start, uint64_t *end, hipStream_t stream, uint32_t extraFlags = 0) {
  if (!dfa) return fail(REDGPU_EAPI, "null dfa handle");
  if (dfa->im->device < 0) return fail(REDGPU_EAPI, "dfa handle has no device image");
  if (int rc = checkStyle(style)) return rc;
  if (n == 0) return REDGPU_OK;
  if (!result) return fail(REDGPU_EAPI, "null result buffer");
  if (!data && (offsets || stride)) return fail(REDGPU_EAPI, "null data buffer");
  if (!offsets && stride >= (1ull << 40)) return fail(REDGPU_ELIMIT, "stride too large");
  if (offsets && stride > 16) return fail(REDGPU_EAPI, "with offsets, stride is the number of "
                                                       "trailing bytes to drop per line (0..16)");
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  Batch b{data, offsets, stride, n, result, start, end};
  const LaunchCfg cfg = cfgOf(dfa, extraFlags);
  const char *name = "";
  hipError_t e = launchBatch(dfa->im->dev, b, verb, style, doLeader ? 1 : 0, cfg, stream, &name);
  tlsKernel = name;
  if (e != hipSuccess) return failHip(e, "kernel launch");
  return REDGPU_OK;
}

// K batches, in order, on one stream (redgpu_*_batches_dev): each validated as runDev validates
// its one batch, then handed to launchBatches, which folds runs of streaming-kernel batches
// into single launches.
int runDevMany(const redgpu_dfa *dfa, int verb, int style, int doLeader, const redgpu_batch *bs,
               uint32_t nb, hipStream_t stream) {
  if (!dfa) return fail(REDGPU_EAPI, "null dfa handle");
  if (dfa->im->device < 0) return fail(REDGPU_EAPI, "dfa handle has no device image");
  if (int rc = checkStyle(style)) return rc;
  if (nb == 0) return REDGPU_OK;
  if (!bs) return fail(REDGPU_EAPI, "null batch descriptors");
  std::vector<Batch> v;
  v.reserve(nb);
  for (uint32_t k = 0; k < nb; ++k) {
    const redgpu_batch &b = bs[k];
    if (b.n == 0) continue;
    if (!b.result) return fail(REDGPU_EAPI, "null result buffer");
    if (!b.data && (b.offsets || b.stride)) return fail(REDGPU_EAPI, "null data buffer");
    if (!b.offsets && b.stride >= (1ull << 40)) return fail(REDGPU_ELIMIT, "stride too large");
    if (b.offsets && b.stride > 16)
      return fail(REDGPU_EAPI, "with offsets, stride is the number of trailing bytes to drop "
                               "per line (0..16)");
    const bool pos = verb == kMatch || verb == kSearch;
    v.push_back(Batch{b.data, b.offsets, b.stride, b.n, b.result, pos ? b.start : nullptr,
                      pos ? b.end : nullptr});
  }
  if (v.empty()) return REDGPU_OK;
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  const LaunchCfg cfg = cfgOf(dfa);
  const char *name = "";
  hipError_t e = launchBatches(dfa->im->dev, v.data(), uint32_t(v.size()), verb, style,
                               doLeader ? 1 : 0, cfg, stream, &name);
  tlsKernel = name;
  if (e != hipSuccess) return failHip(e, "kernel launch");
  return REDGPU_OK;
}

// offsets[0..n] of a host-buffer call: monotone (a decreasing pair would underflow a line length
// on the device and send the walk far outside the buffer)
int checkOffsets(const uint64_t *offsets, uint64_t n, uint64_t *maxLen = nullptr) {
  uint64_t m = 0;
  for (uint64_t i = 0; i < n; ++i) {
    if (offsets[i] > offsets[i + 1]) return fail(REDGPU_EAPI, "offsets not monotonic");
    const uint64_t l = offsets[i + 1] - offsets[i];
    m = l > m ? l : m;
  }
  if (maxLen) *maxLen = m;
  return REDGPU_OK;
}

// The calling thread's staging for the handle's device (host_stage.h); the device scope must
// be held by the caller.
int stageOf(const redgpu_dfa *dfa, HostStage **st) {
  hipError_t e = hostStage(dfa->im->device, st);
  if (e != hipSuccess) return failHip(e, "host staging (streams)");
  return REDGPU_OK;
}

#define STAGE_TRY(expr, what)                                  \
  do {                                                         \
    hipError_t e_ = (expr);                                    \
    if (e_ != hipSuccess) {                                    \
      (void)st->sync();                                        \
      return failHip(e_, what);                                \
    }                                                          \
  } while (0)

// device buffer slots of a HostStage
enum StageSlot : int {
  kSlData = 0,   // +parity
  kSlRes = 2,    // +parity
  kSlStart = 4,  // +parity
  kSlEnd = 6,    // +parity
  kSlOff = 8,
  kSlAux0 = 9,   // counts / state / replacement ...
  kSlAux1 = 10,
  kSlAux2 = 11,
  kSlAux3 = 12,
};

// chunk size of the host-buffer pipeline (REDGPU_HOST_CHUNK_MB overrides: tuning)
static uint64_t hostChunkBytes() {
  static const uint64_t v = [] {
    const char *e = getenv("REDGPU_HOST_CHUNK_MB");
    const long mb = e ? atol(e) : 0;
    return uint64_t(mb >= 1 && mb <= 1024 ? mb : 32) << 20;
  }();
  return v;
}
#define kHostChunkBytes hostChunkBytes()

// offsets of a chunk that does not start at byte 0, rebased to the chunk's own buffer: the
// kernels may read data[0, offsets[n]) anywhere (lanes without a line re-read block 0), so a
// chunk is always presented as a batch of its own
__global__ void __launch_bounds__(256)
k_rebase(const uint64_t *in, uint64_t n1, uint64_t base, uint64_t *out) {
  const uint64_t step = uint64_t(gridDim.x) * 256;
  for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n1; i += step)
    out[i] = in[i] - base;
}

// host-buffer form: the batch is cut into chunks of ~32 MiB of input that alternate between the
// thread's two private streams - copy in, kernel, copy out per chunk - so that with pinned
// caller memory (registered for the duration of the call when the batch has several chunks)
// the upload of one chunk runs beside the download of the previous one.  No allocation, no
// stream creation and no device-wide synchronisation per call (host_stage.h).
int runHost(const redgpu_dfa *dfa, int verb, int style, int doLeader, const uint8_t *data,
            const uint64_t *offsets, uint64_t stride, uint64_t n, int32_t *result,
            uint64_t *start, uint64_t *end) {
  if (!dfa) return fail(REDGPU_EAPI, "null dfa handle");
  if (dfa->im->device < 0) return fail(REDGPU_EAPI, "dfa handle has no device image");
  if (int rc = checkStyle(style)) return rc;
  if (n == 0) return REDGPU_OK;
  if (!result) return fail(REDGPU_EAPI, "null result buffer");
  if (!offsets && stride >= (1ull << 40)) return fail(REDGPU_ELIMIT, "stride too large");
  uint64_t maxLen = 0;
  if (offsets)
    if (int rc = checkOffsets(offsets, n, &maxLen)) return rc;
  // the block-wise ragged kernels keep line positions in 32 bits: a line of 4 GiB or more takes
  // the general kernel (64-bit positions) - here, where the offsets can be read
  const uint32_t extra = maxLen >= (1ull << 32) - 256 ? REDGPU_F_FORCE_GENERIC : 0u;
  const uint64_t total = offsets ? offsets[n] : stride * n;
  if (total && !data) return fail(REDGPU_EAPI, "null data buffer");
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  HostStage *st = nullptr;
  if (int rc = stageOf(dfa, &st)) return rc;
  st->beginCall(total);

  // Caller memory that is already pinned (redgpu_host_register, hipHostMalloc, torch's
  // pin_memory): its copies are asynchronous as they stand, so even a batch of a few MiB is worth
  // cutting - chunks of 8 MiB alternate between the two streams and the upload of one runs beside
  // the walk and the download of the one before.  Pageable memory keeps the 32 MiB chunks and is
  // only cut (and pinned for the call) above 64 MiB: below that the pinning costs more than the
  // overlap returns.
  const bool callerPinned = isPinnedHost(data) && isPinnedHost(result) &&
                            (!start || isPinnedHost(start)) && (!end || isPinnedHost(end));
  const uint64_t chunkBytes = callerPinned ? (8ull << 20) : kHostChunkBytes;
  // chunk plan: cuts[c] .. cuts[c + 1] are the lines of chunk c
  std::vector<uint64_t> cuts{0};
  const uint64_t base0 = offsets ? offsets[0] : 0;
  if (total - base0 > 2 * chunkBytes && n >= 4096) {
    if (!offsets) {
      uint64_t per = (chunkBytes / (stride ? stride : 1)) & ~uint64_t(1023);
      if (per < 1024) per = 1024;
      for (uint64_t lo = per; lo < n; lo += per) cuts.push_back(lo);
    } else {
      uint64_t lo = 0;
      while (lo < n) {
        // first line whose start is >= chunkBytes past this chunk's start
        const uint64_t target = offsets[lo] + chunkBytes;
        uint64_t a = lo + 1, b = n;
        while (a < b) {
          const uint64_t mid = (a + b) / 2;
          if (offsets[mid] >= target) b = mid; else a = mid + 1;
        }
        lo = a;
        if (lo < n) cuts.push_back(lo);
      }
    }
  }
  cuts.push_back(n);
  const size_t nChunks = cuts.size() - 1;
  const bool multi = nChunks > 1;
  uint64_t maxBytes = 0, maxLines = 0;
  for (size_t c = 0; c < nChunks; ++c) {
    const uint64_t lo = cuts[c], hi = cuts[c + 1];
    const uint64_t bytes = offsets ? offsets[hi] - offsets[lo] : (hi - lo) * stride;
    if (bytes > maxBytes) maxBytes = bytes;
    if (hi - lo > maxLines) maxLines = hi - lo;
  }
  // every buffer before the first copy: growing one waits for the streams
  uint8_t *dData[2] = {nullptr, nullptr};
  int32_t *dRes[2] = {nullptr, nullptr};
  uint64_t *dStart[2] = {nullptr, nullptr}, *dEnd[2] = {nullptr, nullptr}, *dOff = nullptr;
  for (int k = 0; k < (multi ? 2 : 1); ++k) {
    STAGE_TRY(st->get(kSlData + k, maxBytes, reinterpret_cast<void **>(&dData[k])), "hipMalloc data");
    STAGE_TRY(st->get(kSlRes + k, maxLines * 4, reinterpret_cast<void **>(&dRes[k])), "hipMalloc result");
    if (start)
      STAGE_TRY(st->get(kSlStart + k, maxLines * 8, reinterpret_cast<void **>(&dStart[k])), "hipMalloc start");
    if (end)
      STAGE_TRY(st->get(kSlEnd + k, maxLines * 8, reinterpret_cast<void **>(&dEnd[k])), "hipMalloc end");
  }
  uint64_t *dReb[2] = {nullptr, nullptr};
  if (offsets) {
    STAGE_TRY(st->get(kSlOff, (n + 1) * 8, reinterpret_cast<void **>(&dOff)), "hipMalloc offsets");
    if (multi || base0)
      for (int k = 0; k < (multi ? 2 : 1); ++k)
        STAGE_TRY(st->get(kSlAux0 + k, (maxLines + 1) * 8, reinterpret_cast<void **>(&dReb[k])),
                  "hipMalloc chunk offsets");
  }

  // pinned caller memory makes the copies truly asynchronous (only worth its price when there
  // is something to overlap)
  // the caller's memory is pinned and copied as it is; otherwise HostStage::copyIn / copyOut
  // choose per transfer (a small call's pageable memory through the thread's pinned arena, a
  // large one's whole pages registered for the call - per chunk when there are several)
  const bool direct = callerPinned;

  if (offsets) {
    STAGE_TRY(st->copyIn(dOff, offsets, (n + 1) * 8, 0, direct), "copy offsets");
    if (multi) {
      STAGE_TRY(hipEventRecord(st->ready, st->streams[0]), "hipEventRecord");
      STAGE_TRY(hipStreamWaitEvent(st->streams[1], st->ready, 0), "hipStreamWaitEvent");
    }
  }
  for (size_t c = 0; c < nChunks; ++c) {
    const int k = int(c & 1);
    hipStream_t s = st->streams[k];
    const uint64_t lo = cuts[c], hi = cuts[c + 1], nl = hi - lo;
    const uint64_t byteLo = offsets ? offsets[lo] : lo * stride;
    const uint64_t bytes = offsets ? offsets[hi] - byteLo : nl * stride;
    if (bytes)
      STAGE_TRY(st->copyIn(dData[k], data + byteLo, bytes, k, direct), "copy data");
    const uint64_t *chunkOff = offsets ? dOff + lo : nullptr;
    if (offsets && byteLo) {
      const uint32_t blocks = uint32_t((nl + 256) / 256 < 1024 ? (nl + 256) / 256 : 1024);
      hipLaunchKernelGGL(k_rebase, dim3(blocks), dim3(256), 0, s, dOff + lo, nl + 1, byteLo,
                         dReb[k]);
      chunkOff = dReb[k];
    }
    const int rc = runDev(dfa, verb, style, doLeader, dData[k], chunkOff, stride, nl, dRes[k],
                          dStart[k], dEnd[k], s, extra);
    if (rc != REDGPU_OK) {
      (void)st->sync();
      return rc;
    }
    STAGE_TRY(st->copyOut(result + lo, dRes[k], nl * 4, k, direct), "copy result");
    if (start) STAGE_TRY(st->copyOut(start + lo, dStart[k], nl * 8, k, direct), "copy start");
    if (end) STAGE_TRY(st->copyOut(end + lo, dEnd[k], nl * 8, k, direct), "copy end");
  }
  STAGE_TRY(st->sync(), "hipStreamSynchronize");
  return REDGPU_OK;
}

// the two verbs that emit a variable-length record list per line
enum ListVerb : int { kListCollect = 0, kListMatchAll = 1, kListMatchAllLeader = 2 };

int collectDev(const redgpu_dfa *dfa, int listVerb, const uint8_t *data, const uint64_t *offsets,
               uint64_t stride, uint64_t n, uint64_t cap, uint64_t *counts, int32_t *result,
               uint64_t *start, uint64_t *end, hipStream_t stream) {
  if (!dfa) return fail(REDGPU_EAPI, "null dfa handle");
  if (dfa->im->device < 0) return fail(REDGPU_EAPI, "dfa handle has no device image");
  if (n == 0) return REDGPU_OK;
  if (!counts) return fail(REDGPU_EAPI, "null counts buffer");
  if (cap && !result) return fail(REDGPU_EAPI, "null result buffer");
  if (!data && (offsets || stride)) return fail(REDGPU_EAPI, "null data buffer");
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  Batch b{data, offsets, stride, n, result, start, end};
  LaunchCfg cfg{dfa->numCUs, (dfa->flags & REDGPU_F_FORCE_GENERIC) ? 1 : 0};
  hipError_t e = listVerb == kListCollect
                     ? launchCollect(dfa->im->dev, b, cap, counts, cfg, stream)
                     : launchMatchAll(dfa->im->dev, b, cap, counts, listVerb == kListMatchAllLeader,
                                      cfg, stream);
  tlsKernel = listVerb == kListCollect ? "k_collect" : "k_matchall";
  if (e != hipSuccess) return failHip(e, "kernel launch");
  return REDGPU_OK;
}

// Uploads im->img to im->device.  Caller holds the device scope.
int uploadImage(SharedImage *im) {
  const DfaImage &img = im->img;
  uint8_t eqLead[1024];
  std::memcpy(eqLead, img.equiv, 256);
  std::memcpy(eqLead + 256, img.leader, 256);
  std::memcpy(eqLead + 512, img.startFlags[0], 256);
  std::memcpy(eqLead + 768, img.startFlags[1], 256);
  const size_t tabBytes = (img.table.size() + 15) & ~size_t(15);
  hipError_t e;
  if ((e = hipMalloc(&im->dTable, tabBytes + 16)) != hipSuccess) return failHip(e, "hipMalloc table");
  if ((e = hipMalloc(&im->dResult, img.nStates * sizeof(int32_t) + 16)) != hipSuccess)
    return failHip(e, "hipMalloc result");
  if ((e = hipMalloc(&im->dEquivLeader, 1024 + 64)) != hipSuccess) return failHip(e, "hipMalloc equiv");
  if ((e = hipMemset(im->dTable, 0, tabBytes + 16)) != hipSuccess) return failHip(e, "hipMemset");
  // The table and the result array are heap memory of this process (std::vector): uploaded
  // THROUGH A PINNED BUFFER of the library's own, never as they are - above ~1 MiB the runtime
  // would pin the vector's pages on the fly, by address, and a GPU fault at a heap address was
  // seen three times exactly here, under the 2 MiB table of SYN-4K, when the heap around it
  // had just been registered and released for a host-buffer call.
  auto upload = [](void *dst, const void *src, size_t bytes) -> hipError_t {
    if (bytes == 0) return hipSuccess;
    void *pin = nullptr;
    hipError_t e2 = hipHostMalloc(&pin, bytes, hipHostMallocDefault);
    if (e2 != hipSuccess) return e2;
    std::memcpy(pin, src, bytes);
    e2 = hipMemcpy(dst, pin, bytes, hipMemcpyHostToDevice);
    (void)hipHostFree(pin);
    return e2;
  };
  if ((e = upload(im->dTable, img.table.data(), img.table.size())) != hipSuccess)
    return failHip(e, "upload table");
  if ((e = upload(im->dResult, img.result.data(), img.nStates * sizeof(int32_t))) != hipSuccess)
    return failHip(e, "upload result");
  if ((e = hipMemcpy(im->dEquivLeader, eqLead, 1024, hipMemcpyHostToDevice)) != hipSuccess)
    return failHip(e, "upload equiv");

  DevDfa &d = im->dev;
  d.table = static_cast<const uint8_t *>(im->dTable);
  d.result = static_cast<const int32_t *>(im->dResult);
  d.equivLeader = static_cast<const uint8_t *>(im->dEquivLeader);
  d.sink = static_cast<uint8_t *>(im->dEquivLeader) + 512;
  d.tableKind = img.tableKind;
  d.tableBytes = (img.primaryBytes + 15u) & ~15u;  // the tableKind table only (LDS staging size)
  d.nStates = img.nStates;
  d.nClasses = img.nClasses;
  d.init = img.init;
  d.leaderNext = img.leaderNext;
  d.nPureDead = img.nPureDead;
  d.firstAccept = img.firstAccept;
  d.leaderLen = img.leaderLen;
  d.deadAbsorbing = img.deadAbsorbing ? 1 : 0;
  d.hotLo = img.hotLo;
  d.nHot = img.nHot;
  d.hot8Off = img.hot8Off;
  d.hotShift = img.hotShift;
  d.earlyDeath = img.earlyDeath ? 1 : 0;
  d.clsOff = img.clsOff;
  d.clsRowBytes = img.clsRowBytes;
  d.clsBytes = img.clsBytes;
  d.clsIndexForm = img.clsIndexForm ? 1 : 0;
  d.sparseCombOff = img.sparseCombOff;
  d.sparseDefault = img.sparseDefault;
  d.tuned = img.tuned ? 1 : 0;
  d.forgetful = img.forgetful ? 1 : 0;
  d.suffixClosed = img.suffixClosed ? 1 : 0;
  d.uniformResult = img.uniformResult ? 1 : 0;
  d.leaderForced = img.leaderForced ? 1 : 0;
  {
    const char *e = getenv("REDGPU_GATHER_NT");
    d.gatherNt = e && e[0] == '1';
  }
  d.startLeadWord = img.startLeadWord;
  d.startLeadCount = img.startLeadCount;
  d.startFreeWord = img.startFreeWord;
  d.startFreeCount = img.startFreeCount;
  d.start2LeadWord = img.start2LeadWord;
  d.start2LeadCount = img.start2LeadCount;
  d.start2FreeWord = img.start2FreeWord;
  d.start2FreeCount = img.start2FreeCount;
  for (int k = 0; k < 2; ++k) {
    d.startTotal[k] = img.startTotal[k];
    d.startFollow[k] = img.startFollow[k] ? 1u : 0u;
  }
  return REDGPU_OK;
}

// Loader cache (the reference's load path, lib/Serializer.cpp:257-267, hands
// every caller its own Executable): a service that creates many handles from the same blob pays
// for validation, repack and upload once per (blob, device, build options).  Entries are weak:
// the image goes away with its last handle.
struct CacheKey {
  uint32_t checksum;
  size_t len;
  int device;
  uint32_t buildFlags, ldsTableMax;
  bool operator==(const CacheKey &o) const {
    return checksum == o.checksum && len == o.len && device == o.device &&
           buildFlags == o.buildFlags && ldsTableMax == o.ldsTableMax;
  }
};
struct CacheKeyHash {
  size_t operator()(const CacheKey &k) const {
    uint64_t h = k.checksum;
    h = h * 1099511628211ull ^ k.len;
    h = h * 1099511628211ull ^ uint64_t(uint32_t(k.device));
    h = h * 1099511628211ull ^ k.buildFlags;
    h = h * 1099511628211ull ^ k.ldsTableMax;
    return size_t(h);
  }
};
std::mutex gCacheMutex;
std::unordered_map<CacheKey, std::weak_ptr<SharedImage>, CacheKeyHash> gCache;
constexpr uint32_t kBuildFlagMask = REDGPU_F_FORCE_GLOBAL | REDGPU_F_FORCE_HOT;

} // namespace

extern "C" {

int redgpu_version(void) { return 100; }

const char *redgpu_last_error(void) { return tlsError.c_str(); }

const char *redgpu_last_kernel(void) { return tlsKernel; }

int redgpu_reda_check(const void *reda, size_t len, const char **msg) {
  const char *m = (reda && len) ? checkHeader(reda, len) : "serialized dfa is empty";
  if (msg) *msg = m;
  if (m) return fail(REDGPU_EAPI, m);
  return REDGPU_OK;
}

int redgpu_dfa_create(const void *reda, size_t len, const redgpu_opts *opts, redgpu_dfa **out) {
  if (!out) return fail(REDGPU_EAPI, "null output handle");
  *out = nullptr;
  redgpu_opts o{};
  o.device = REDGPU_DEVICE_CURRENT;
  if (opts) o = *opts;
  if (!reda || len == 0) return fail(REDGPU_EAPI, "serialized dfa is empty");
  if (const char *m = checkHeader(reda, len)) return fail(REDGPU_EAPI, m);

  int dev = o.device;
  int numCUs = 0;
  if (dev != REDGPU_DEVICE_NONE) {
    if (dev == REDGPU_DEVICE_CURRENT) {
      hipError_t e = hipGetDevice(&dev);
      if (e != hipSuccess) return failHip(e, "hipGetDevice");
    }
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return failHip(e, "hipGetDeviceProperties");
    numCUs = prop.multiProcessorCount;
  }

  redgpu_dfa *h = new (std::nothrow) redgpu_dfa();
  if (!h) return fail(REDGPU_ELIMIT, "out of host memory");
  h->flags = o.flags;
  h->ldsTableMax = o.lds_table_max;
  h->numCUs = numCUs;

  // the header is valid, so the checksum is the FNV-1a-32 of the payload: the cache key
  const CacheKey key{calcChecksum(reda, len), len, dev, o.flags & kBuildFlagMask, o.lds_table_max};
  {
    std::lock_guard<std::mutex> lock(gCacheMutex);
    auto it = gCache.find(key);
    if (it != gCache.end()) {
      if (std::shared_ptr<SharedImage> hit = it->second.lock()) {
        if (hit->blob.size() == len && std::memcmp(hit->blob.data(), reda, len) == 0) {
          h->im = std::move(hit);
          *out = h;
          return REDGPU_OK;
        }
      } else {
        gCache.erase(it);
      }
    }
  }

  auto im = std::make_shared<SharedImage>();
  int code = REDGPU_OK;
  std::string err = buildImage(reda, len, o.lds_table_max, (o.flags & REDGPU_F_FORCE_GLOBAL) != 0,
                               im->img, code, (o.flags & REDGPU_F_FORCE_HOT) != 0);
  if (!err.empty()) {
    delete h;
    return fail(code, err);
  }
  im->blob.assign(static_cast<const uint8_t *>(reda), static_cast<const uint8_t *>(reda) + len);
  im->buildFlags = key.buildFlags;
  im->ldsTableMax = o.lds_table_max;
  im->device = dev;
  if (dev != REDGPU_DEVICE_NONE) {
    DeviceScope scope(dev);
    if (scope.err != hipSuccess) { delete h; return failHip(scope.err, "hipSetDevice"); }
    if (int rc = uploadImage(im.get())) { delete h; return rc; }
  }
  h->im = im;
  {
    std::lock_guard<std::mutex> lock(gCacheMutex);
    gCache[key] = im;
  }
  *out = h;
  return REDGPU_OK;
}

void redgpu_dfa_destroy(redgpu_dfa *h) {
  delete h;  // the shared image goes with its last handle (~SharedImage frees the device side)
}

int redgpu_dfa_info(const redgpu_dfa *h, redgpu_info *out) {
  if (!h || !out) return fail(REDGPU_EAPI, "null argument");
  const DfaImage &img = h->im->img;
  out->format = img.format;
  out->n_classes = img.nClasses;
  out->leader_len = img.leaderLen;
  out->states_total = img.statesTotal;
  out->states_used = img.nStates;
  out->n_pure_dead = img.nPureDead;
  out->first_accept = img.firstAccept;
  out->table_kind = img.tableKind;
  out->table_bytes = img.primaryBytes;
  out->max_result = img.maxResult;
  out->device = h->im->device;
  out->checksum = img.checksum;
  out->fast_path = (img.tableKind == REDGPU_TAB_LDS_FUSED_U8 && img.deadAbsorbing &&
                    !(h->flags & REDGPU_F_FORCE_GENERIC) &&
                    (!img.earlyDeath || (h->flags & REDGPU_F_FORCE_STREAM))) ? 1 : 0;
  out->n_hot = img.nHot;
  out->hot_lo = img.hotLo;
  out->hot_coverage_ppm = img.hotCoveragePpm;
  out->early_death = img.earlyDeath ? 1 : 0;
  out->forgetful = img.forgetful ? 1 : 0;
  out->suffix_closed = img.suffixClosed ? 1 : 0;
  out->image_refs = uint32_t(h->im.use_count());
  return REDGPU_OK;
}

int redgpu_dfa_serialized(const redgpu_dfa *h, const void **reda, size_t *len) {
  if (!h || !reda || !len) return fail(REDGPU_EAPI, "null argument");
  *reda = h->im->blob.data();
  *len = h->im->blob.size();
  return REDGPU_OK;
}

int redgpu_check_batch(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                       const uint64_t *offsets, uint64_t stride, uint64_t n, int32_t *result) {
  return runHost(dfa, kCheck, style, do_leader, data, offsets, stride, n, result, nullptr,
                 nullptr);
}

int redgpu_match_batch(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                       const uint64_t *offsets, uint64_t stride, uint64_t n, int32_t *result,
                       uint64_t *start, uint64_t *end) {
  return runHost(dfa, kMatch, style, do_leader, data, offsets, stride, n, result, start, end);
}

int redgpu_scan_batch(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                      const uint64_t *offsets, uint64_t stride, uint64_t n, int32_t *result) {
  return runHost(dfa, kScan, style, do_leader, data, offsets, stride, n, result, nullptr,
                 nullptr);
}

int redgpu_search_batch(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                        const uint64_t *offsets, uint64_t stride, uint64_t n, int32_t *result,
                        uint64_t *start, uint64_t *end) {
  return runHost(dfa, kSearch, style, do_leader, data, offsets, stride, n, result, start, end);
}

int redgpu_search_batch_dev(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                            const uint64_t *offsets, uint64_t stride, uint64_t n,
                            int32_t *result, uint64_t *start, uint64_t *end, void *stream) {
  return runDev(dfa, kSearch, style, do_leader, data, offsets, stride, n, result, start, end,
                static_cast<hipStream_t>(stream));
}

int redgpu_collect_batch_dev(const redgpu_dfa *dfa, const uint8_t *data, const uint64_t *offsets,
                             uint64_t stride, uint64_t n, uint64_t cap, uint64_t *counts,
                             int32_t *result, uint64_t *start, uint64_t *end, void *stream) {
  return collectDev(dfa, kListCollect, data, offsets, stride, n, cap, counts, result, start, end,
                    static_cast<hipStream_t>(stream));
}

int redgpu_match_all_batch_dev(const redgpu_dfa *dfa, int do_leader, const uint8_t *data,
                               const uint64_t *offsets, uint64_t stride, uint64_t n,
                               uint64_t cap, uint64_t *counts, int32_t *result, uint64_t *start,
                               uint64_t *end, void *stream) {
  return collectDev(dfa, do_leader ? kListMatchAllLeader : kListMatchAll, data, offsets, stride, n,
                    cap, counts, result, start, end, static_cast<hipStream_t>(stream));
}

static int listHost(const redgpu_dfa *dfa, int listVerb, const uint8_t *data,
                    const uint64_t *offsets, uint64_t stride, uint64_t n, uint64_t cap,
                    uint64_t *counts, int32_t *result, uint64_t *start, uint64_t *end);

int redgpu_collect_batch(const redgpu_dfa *dfa, const uint8_t *data, const uint64_t *offsets,
                         uint64_t stride, uint64_t n, uint64_t cap, uint64_t *counts,
                         int32_t *result, uint64_t *start, uint64_t *end) {
  return listHost(dfa, kListCollect, data, offsets, stride, n, cap, counts, result, start, end);
}

int redgpu_match_all_batch(const redgpu_dfa *dfa, int do_leader, const uint8_t *data,
                           const uint64_t *offsets, uint64_t stride, uint64_t n, uint64_t cap,
                           uint64_t *counts, int32_t *result, uint64_t *start, uint64_t *end) {
  return listHost(dfa, do_leader ? kListMatchAllLeader : kListMatchAll, data, offsets, stride, n,
                  cap, counts, result, start, end);
}

static int listHost(const redgpu_dfa *dfa, int listVerb, const uint8_t *data,
                    const uint64_t *offsets, uint64_t stride, uint64_t n, uint64_t cap,
                    uint64_t *counts, int32_t *result, uint64_t *start, uint64_t *end) {
  if (!dfa) return fail(REDGPU_EAPI, "null dfa handle");
  if (dfa->im->device < 0) return fail(REDGPU_EAPI, "dfa handle has no device image");
  if (n == 0) return REDGPU_OK;
  if (!counts) return fail(REDGPU_EAPI, "null counts buffer");
  if (cap && !result) return fail(REDGPU_EAPI, "null result buffer");
  if (offsets)
    if (int rc = checkOffsets(offsets, n)) return rc;
  if (!offsets && stride >= (1ull << 40)) return fail(REDGPU_ELIMIT, "stride too large");
  if (cap && n > (~0ull / 16) / cap) return fail(REDGPU_ELIMIT, "n * cap too large");
  const uint64_t total = offsets ? offsets[n] : stride * n;
  if (total && !data) return fail(REDGPU_EAPI, "null data buffer");
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  HostStage *st = nullptr;
  if (int rc = stageOf(dfa, &st)) return rc;
  st->beginCall(total);
  hipStream_t s = st->streams[0];
  uint8_t *dData = nullptr;
  uint64_t *dOff = nullptr, *dCnt = nullptr, *dStart = nullptr, *dEnd = nullptr;
  int32_t *dRes = nullptr;
  const uint64_t slots = n * cap;
  STAGE_TRY(st->get(kSlData, total, reinterpret_cast<void **>(&dData)), "hipMalloc data");
  STAGE_TRY(st->get(kSlAux0, n * 8, reinterpret_cast<void **>(&dCnt)), "hipMalloc counts");
  STAGE_TRY(st->get(kSlRes, (slots + 1) * 4, reinterpret_cast<void **>(&dRes)), "hipMalloc result");
  if (start)
    STAGE_TRY(st->get(kSlStart, (slots + 1) * 8, reinterpret_cast<void **>(&dStart)), "hipMalloc start");
  if (end)
    STAGE_TRY(st->get(kSlEnd, (slots + 1) * 8, reinterpret_cast<void **>(&dEnd)), "hipMalloc end");
  if (offsets) {
    STAGE_TRY(st->get(kSlOff, (n + 1) * 8, reinterpret_cast<void **>(&dOff)), "hipMalloc offsets");
    STAGE_TRY(st->copyIn(dOff, offsets, (n + 1) * 8, 0), "copy offsets");
  }
  if (total) STAGE_TRY(st->copyIn(dData, data, total, 0), "copy data");
  const int rc = collectDev(dfa, listVerb, dData, dOff, stride, n, cap, dCnt, dRes, dStart, dEnd, s);
  if (rc != REDGPU_OK) {
    (void)st->sync();
    return rc;
  }
  STAGE_TRY(st->copyOut(counts, dCnt, n * 8, 0), "copy counts");
  if (slots) {
    STAGE_TRY(st->copyOut(result, dRes, slots * 4, 0), "copy result");
    if (start) STAGE_TRY(st->copyOut(start, dStart, slots * 8, 0), "copy start");
    if (end) STAGE_TRY(st->copyOut(end, dEnd, slots * 8, 0), "copy end");
  }
  STAGE_TRY(st->syncStream(0), "hipStreamSynchronize");
  return REDGPU_OK;
}

int redgpu_replace_batch_dev(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                             const uint64_t *offsets, uint64_t stride, uint64_t n,
                             const uint8_t *repl, uint64_t repl_len, uint64_t max_count,
                             uint64_t *counts, uint64_t *out_offsets, uint8_t *out,
                             uint64_t out_cap, void *stream) {
  if (!dfa) return fail(REDGPU_EAPI, "null dfa handle");
  if (dfa->im->device < 0) return fail(REDGPU_EAPI, "dfa handle has no device image");
  if (int rc = checkStyle(style)) return rc;
  if (n == 0) return REDGPU_OK;
  if (!counts || !out_offsets) return fail(REDGPU_EAPI, "null output buffer");
  if (!data && (offsets || stride)) return fail(REDGPU_EAPI, "null data buffer");
  if (repl_len && !repl) return fail(REDGPU_EAPI, "null replacement");
  if (offsets && stride > 16) return fail(REDGPU_EAPI, "with offsets, stride is the number of "
                                                       "trailing bytes to drop per line (0..16)");
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  Batch b{data, offsets, stride, n, nullptr, nullptr, nullptr};
  LaunchCfg cfg{dfa->numCUs, 0};
  hipError_t e = launchReplace(dfa->im->dev, b, style, do_leader ? 1 : 0, repl, repl_len, max_count,
                               counts, out_offsets, out, out_cap, cfg,
                               static_cast<hipStream_t>(stream));
  tlsKernel = "k_replace";
  if (e != hipSuccess) return failHip(e, "kernel launch");
  return REDGPU_OK;
}

int redgpu_replace_batch(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                         const uint64_t *offsets, uint64_t stride, uint64_t n, const uint8_t *repl,
                         uint64_t repl_len, uint64_t max_count, uint64_t *counts,
                         uint64_t *out_offsets, uint8_t *out, uint64_t out_cap) {
  if (!dfa) return fail(REDGPU_EAPI, "null dfa handle");
  if (dfa->im->device < 0) return fail(REDGPU_EAPI, "dfa handle has no device image");
  if (int rc = checkStyle(style)) return rc;
  if (n == 0) return REDGPU_OK;
  if (!counts || !out_offsets) return fail(REDGPU_EAPI, "null output buffer");
  if (offsets)
    if (int rc = checkOffsets(offsets, n)) return rc;
  if (!offsets && stride >= (1ull << 40)) return fail(REDGPU_ELIMIT, "stride too large");
  const uint64_t total = offsets ? offsets[n] : stride * n;
  if (total && !data) return fail(REDGPU_EAPI, "null data buffer");
  if (repl_len && !repl) return fail(REDGPU_EAPI, "null replacement");
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  HostStage *st = nullptr;
  if (int rc = stageOf(dfa, &st)) return rc;
  st->beginCall(total);
  hipStream_t s = st->streams[0];
  uint8_t *dData = nullptr, *dRepl = nullptr, *dOut = nullptr;
  uint64_t *dOff = nullptr, *dCnt = nullptr, *dOutOff = nullptr;
  STAGE_TRY(st->get(kSlData, total, reinterpret_cast<void **>(&dData)), "hipMalloc data");
  STAGE_TRY(st->get(kSlAux0, repl_len, reinterpret_cast<void **>(&dRepl)), "hipMalloc repl");
  STAGE_TRY(st->get(kSlAux1, n * 8, reinterpret_cast<void **>(&dCnt)), "hipMalloc counts");
  STAGE_TRY(st->get(kSlAux2, (n + 1) * 8, reinterpret_cast<void **>(&dOutOff)), "hipMalloc out offsets");
  if (out && out_cap)
    STAGE_TRY(st->get(kSlAux3, out_cap, reinterpret_cast<void **>(&dOut)), "hipMalloc out");
  if (offsets) {
    STAGE_TRY(st->get(kSlOff, (n + 1) * 8, reinterpret_cast<void **>(&dOff)), "hipMalloc offsets");
    STAGE_TRY(st->copyIn(dOff, offsets, (n + 1) * 8, 0), "copy offsets");
  }
  if (total) STAGE_TRY(st->copyIn(dData, data, total, 0), "copy data");
  if (repl_len) STAGE_TRY(st->copyIn(dRepl, repl, repl_len, 0), "copy repl");
  const int rc = redgpu_replace_batch_dev(dfa, style, do_leader, dData, dOff, stride, n, dRepl,
                                          repl_len, max_count, dCnt, dOutOff, dOut,
                                          dOut ? out_cap : 0, s);
  if (rc != REDGPU_OK) {
    (void)st->sync();
    return rc;
  }
  STAGE_TRY(st->copyOut(counts, dCnt, n * 8, 0), "copy counts");
  STAGE_TRY(st->copyOut(out_offsets, dOutOff, (n + 1) * 8, 0),
            "copy out offsets");
  STAGE_TRY(st->syncStream(0), "hipStreamSynchronize");
  if (dOut) {
    // the lines that fit are a prefix (offsets are monotone): copy up to the last one that does
    uint64_t lo = 0, hi = n;  // largest k with out_offsets[k] <= out_cap
    while (lo < hi) {
      const uint64_t mid = (lo + hi + 1) / 2;
      if (out_offsets[mid] <= out_cap) lo = mid; else hi = mid - 1;
    }
    if (out_offsets[lo]) {
      STAGE_TRY(st->copyOut(out, dOut, out_offsets[lo], 0), "copy out");
      STAGE_TRY(st->syncStream(0), "hipStreamSynchronize");
    }
  }
  return REDGPU_OK;
}

int redgpu_split_lines_dev(const redgpu_dfa *dfa, const uint8_t *data, uint64_t len, uint8_t delim,
                           uint64_t *offsets, uint64_t cap, uint64_t *n_lines, void *stream) {
  if (!dfa) return fail(REDGPU_EAPI, "null dfa handle");
  if (dfa->im->device < 0) return fail(REDGPU_EAPI, "dfa handle has no device image");
  if (!offsets || !n_lines) return fail(REDGPU_EAPI, "null output buffer");
  if (len && !data) return fail(REDGPU_EAPI, "null data buffer");
  if (splitChunks(len) >= (1ull << 31)) return fail(REDGPU_ELIMIT, "buffer too large");
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const uint64_t nChunks = splitChunks(len);
  void *scratch = nullptr;
  // counts u32[nChunks], bases u64[nChunks], then the delimiter masks (2 bytes per 16 of input):
  // the calling thread's scratch for this stream (kernels.h) - what the next launch on the
  // stream does with the same buffer comes behind these kernels.  (Up to round 3 this was
  // hipMallocAsync / hipFreeAsync per call: ~12 us of host time each, and the only use of the
  // stream-ordered allocator in the library.)
  const size_t countBytes = (size_t(nChunks) * 4 + 15) & ~size_t(15);
  const size_t headBytes = countBytes + size_t(nChunks) * 8 + 16;
  HIP_TRY(scratchFor(s, headBytes + splitMaskBytes(len) + 16, &scratch), "hipMalloc scratch");
  uint32_t *counts = static_cast<uint32_t *>(scratch);
  uint64_t *bases = reinterpret_cast<uint64_t *>(static_cast<uint8_t *>(scratch) + countBytes);
  uint16_t *masks = reinterpret_cast<uint16_t *>(static_cast<uint8_t *>(scratch) + headBytes);
  hipError_t e = launchSplitLines(data, len, delim, offsets, cap, n_lines, counts, bases, masks, s);
  tlsKernel = "k_split_scatter";
  if (e != hipSuccess) return failHip(e, "kernel launch");
  return REDGPU_OK;
}

// raw text -> lines -> check / match, one call, everything on `stream`
static int textDev(const redgpu_dfa *dfa, int verb, int style, int doLeader, const uint8_t *data,
                   uint64_t len, uint8_t delim, uint64_t *offsets, uint64_t cap, uint64_t *n_lines,
                   int32_t *result, uint64_t *start, uint64_t *end, void *stream) {
  if (!dfa) return fail(REDGPU_EAPI, "null dfa handle");
  if (int rc = checkStyle(style)) return rc;
  if (cap && !result) return fail(REDGPU_EAPI, "null result buffer");
  if (int rc = redgpu_split_lines_dev(dfa, data, len, delim, offsets, cap, n_lines, stream)) return rc;
  if (cap > len) cap = len;  // a buffer holds no more lines than bytes
  if (cap == 0) return REDGPU_OK;
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  hipStream_t s = static_cast<hipStream_t>(stream);
  Batch b{data, offsets, 1, cap, result, start, end};
  b.nDev = n_lines;
  const LaunchCfg cfg = cfgOf(dfa);
  const char *name = "";
  hipError_t e = launchBatch(dfa->im->dev, b, verb, style, doLeader ? 1 : 0, cfg, s, &name);
  if (e == hipSuccess) {
    tlsKernel = name;
    return REDGPU_OK;
  }
  if (e != hipErrorNotSupported) return failHip(e, "kernel launch");
  (void)hipGetLastError();
  // a kernel family that takes its line count from the host: wait for the split
  uint64_t n = 0;
  HIP_TRY(hipMemcpyAsync(&n, n_lines, 8, hipMemcpyDeviceToHost, s), "copy line count");
  HIP_TRY(hipStreamSynchronize(s), "hipStreamSynchronize");
  return runDev(dfa, verb, style, doLeader, data, offsets, 1, n < cap ? n : cap, result, start,
                end, s);
}

int redgpu_check_text_dev(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                          uint64_t len, uint8_t delim, uint64_t *offsets, uint64_t cap,
                          uint64_t *n_lines, int32_t *result, void *stream) {
  return textDev(dfa, kCheck, style, do_leader, data, len, delim, offsets, cap, n_lines, result,
                 nullptr, nullptr, stream);
}

int redgpu_match_text_dev(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                          uint64_t len, uint8_t delim, uint64_t *offsets, uint64_t cap,
                          uint64_t *n_lines, int32_t *result, uint64_t *start, uint64_t *end,
                          void *stream) {
  return textDev(dfa, kMatch, style, do_leader, data, len, delim, offsets, cap, n_lines, result,
                 start, end, stream);
}

// host-buffer form: one upload, split + verb on the device, the filled prefixes back
int redgpu_match_text(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                      uint64_t len, uint8_t delim, uint64_t *offsets, uint64_t cap,
                      uint64_t *n_lines, int32_t *result, uint64_t *start, uint64_t *end) {
  if (!dfa) return fail(REDGPU_EAPI, "null dfa handle");
  if (dfa->im->device < 0) return fail(REDGPU_EAPI, "dfa handle has no device image");
  if (!n_lines) return fail(REDGPU_EAPI, "null output buffer");
  if (cap && (!offsets || !result)) return fail(REDGPU_EAPI, "null output buffer");
  if (len && !data) return fail(REDGPU_EAPI, "null data buffer");
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  HostStage *st = nullptr;
  if (int rc = stageOf(dfa, &st)) return rc;
  st->beginCall(len);
  hipStream_t s = st->streams[0];
  if (cap > len) cap = len;
  uint8_t *dData = nullptr;
  uint64_t *dOff = nullptr, *dN = nullptr, *dStart = nullptr, *dEnd = nullptr;
  int32_t *dRes = nullptr;
  STAGE_TRY(st->get(kSlData, len, reinterpret_cast<void **>(&dData)), "hipMalloc data");
  STAGE_TRY(st->get(kSlAux0, 8, reinterpret_cast<void **>(&dN)), "hipMalloc count");
  STAGE_TRY(st->get(kSlOff, (cap + 1) * 8, reinterpret_cast<void **>(&dOff)), "hipMalloc offsets");
  STAGE_TRY(st->get(kSlRes, cap * 4, reinterpret_cast<void **>(&dRes)), "hipMalloc result");
  if (start) STAGE_TRY(st->get(kSlStart, cap * 8, reinterpret_cast<void **>(&dStart)), "hipMalloc start");
  if (end) STAGE_TRY(st->get(kSlEnd, cap * 8, reinterpret_cast<void **>(&dEnd)), "hipMalloc end");
  if (len) STAGE_TRY(st->copyIn(dData, data, len, 0), "copy data");
  const int rc = textDev(dfa, start || end ? kMatch : kCheck, style, do_leader, dData, len, delim,
                         dOff, cap, dN, dRes, dStart, dEnd, s);
  if (rc != REDGPU_OK) {
    (void)st->sync();
    return rc;
  }
  STAGE_TRY(st->copyOut(n_lines, dN, 8, 0), "copy count");
  STAGE_TRY(st->syncStream(0), "hipStreamSynchronize");
  const uint64_t got = *n_lines < cap ? *n_lines : cap;
  if (cap) STAGE_TRY(st->copyOut(offsets, dOff, (got + 1) * 8, 0), "copy offsets");
  if (got) {
    STAGE_TRY(st->copyOut(result, dRes, got * 4, 0), "copy result");
    if (start) STAGE_TRY(st->copyOut(start, dStart, got * 8, 0), "copy start");
    if (end) STAGE_TRY(st->copyOut(end, dEnd, got * 8, 0), "copy end");
  }
  STAGE_TRY(st->syncStream(0), "hipStreamSynchronize");
  return REDGPU_OK;
}

int redgpu_split_lines(const redgpu_dfa *dfa, const uint8_t *data, uint64_t len, uint8_t delim,
                       uint64_t *offsets, uint64_t cap, uint64_t *n_lines) {
  if (!dfa) return fail(REDGPU_EAPI, "null dfa handle");
  if (dfa->im->device < 0) return fail(REDGPU_EAPI, "dfa handle has no device image");
  if (!offsets || !n_lines) return fail(REDGPU_EAPI, "null output buffer");
  if (len && !data) return fail(REDGPU_EAPI, "null data buffer");
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  HostStage *st = nullptr;
  if (int rc = stageOf(dfa, &st)) return rc;
  st->beginCall(len);
  hipStream_t s = st->streams[0];
  uint8_t *dData = nullptr;
  uint64_t *dOff = nullptr, *dN = nullptr;
  STAGE_TRY(st->get(kSlData, len, reinterpret_cast<void **>(&dData)), "hipMalloc data");
  STAGE_TRY(st->get(kSlAux0, 8, reinterpret_cast<void **>(&dN)), "hipMalloc count");
  // count first (room for no line at all), then size the device offsets to what will be kept
  STAGE_TRY(st->get(kSlOff, 8, reinterpret_cast<void **>(&dOff)), "hipMalloc offsets");
  if (len) STAGE_TRY(st->copyIn(dData, data, len, 0), "copy data");
  int rc = redgpu_split_lines_dev(dfa, dData, len, delim, dOff, 0, dN, s);
  if (rc != REDGPU_OK) {
    (void)st->sync();
    return rc;
  }
  STAGE_TRY(st->copyOut(n_lines, dN, 8, 0), "copy count");
  STAGE_TRY(st->syncStream(0), "hipStreamSynchronize");
  const uint64_t got = *n_lines < cap ? *n_lines : cap;
  if (got) {
    STAGE_TRY(st->get(kSlOff, (got + 1) * 8, reinterpret_cast<void **>(&dOff)), "hipMalloc offsets");
    rc = redgpu_split_lines_dev(dfa, dData, len, delim, dOff, got, dN, s);
    if (rc != REDGPU_OK) {
      (void)st->sync();
      return rc;
    }
  }
  STAGE_TRY(st->copyOut(offsets, dOff, (got + 1) * 8, 0), "copy offsets");
  STAGE_TRY(st->syncStream(0), "hipStreamSynchronize");
  return REDGPU_OK;
}

int redgpu_diag_read_dev(const redgpu_dfa *dfa, const void *data, uint64_t bytes, uint32_t *sink,
                         void *stream) {
  if (!dfa) return fail(REDGPU_EAPI, "null dfa handle");
  if (dfa->im->device < 0) return fail(REDGPU_EAPI, "dfa handle has no device image");
  if (!data || !sink) return fail(REDGPU_EAPI, "null buffer");
  if (reinterpret_cast<uintptr_t>(data) % 16) return fail(REDGPU_EAPI, "buffer not 16-byte aligned");
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  hipError_t e = launchDiagRead(data, bytes, sink, dfa->numCUs, static_cast<hipStream_t>(stream));
  tlsKernel = "k_diag_read";
  if (e != hipSuccess) return failHip(e, "kernel launch");
  return REDGPU_OK;
}

int redgpu_diag_lines_dev(const redgpu_dfa *dfa, const uint8_t *data, uint64_t n_lines,
                          uint64_t line_bytes, int32_t *result, uint64_t *start, uint64_t *end,
                          uint32_t *sink, void *stream) {
  if (!dfa) return fail(REDGPU_EAPI, "null dfa handle");
  if (dfa->im->device < 0) return fail(REDGPU_EAPI, "dfa handle has no device image");
  if (line_bytes != 64 && (line_bytes % 128 || line_bytes == 0 || line_bytes >= (1ull << 31)))
    return fail(REDGPU_EAPI, "line_bytes must be 64 or a multiple of 128");
  if (!data || !sink || (line_bytes == 64 && (!result || !start || !end)))
    return fail(REDGPU_EAPI, "null buffer");
  if (reinterpret_cast<uintptr_t>(data) % 16) return fail(REDGPU_EAPI, "buffer not 16-byte aligned");
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  hipError_t e = launchDiagLines(data, n_lines, uint32_t(line_bytes), result, start, end, sink,
                                 dfa->numCUs, static_cast<hipStream_t>(stream));
  tlsKernel = "k_diag_lines";
  if (e != hipSuccess) return failHip(e, "kernel launch");
  return REDGPU_OK;
}

int redgpu_diag_lds_dev(const redgpu_dfa *dfa, uint32_t rounds, uint32_t *sink, uint64_t *lookups,
                        void *stream) {
  if (!dfa) return fail(REDGPU_EAPI, "null dfa handle");
  if (dfa->im->device < 0) return fail(REDGPU_EAPI, "dfa handle has no device image");
  if (!sink) return fail(REDGPU_EAPI, "null buffer");
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  hipError_t e = launchDiagLds(dfa->im->dev, rounds, sink, dfa->numCUs,
                               static_cast<hipStream_t>(stream), lookups);
  tlsKernel = "k_diag_lds";
  if (e != hipSuccess) return failHip(e, "kernel launch");
  return REDGPU_OK;
}

int redgpu_diag_l2_dev(const redgpu_dfa *dfa, const uint16_t *table, uint32_t rounds, uint32_t *sink,
                       uint64_t *lookups, void *stream) {
  if (!dfa) return fail(REDGPU_EAPI, "null dfa handle");
  if (dfa->im->device < 0) return fail(REDGPU_EAPI, "dfa handle has no device image");
  if (!table || !sink) return fail(REDGPU_EAPI, "null buffer");
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  hipError_t e = launchDiagL2(table, rounds, sink, dfa->numCUs, static_cast<hipStream_t>(stream),
                              lookups);
  tlsKernel = "k_diag_l2";
  if (e != hipSuccess) return failHip(e, "kernel launch");
  return REDGPU_OK;
}

int redgpu_diag_walked_dev(const redgpu_dfa *dfa, int do_leader, const uint8_t *data,
                           const uint64_t *offsets, uint64_t stride, uint64_t n, uint64_t *walked,
                           void *stream) {
  if (!dfa) return fail(REDGPU_EAPI, "null dfa handle");
  if (dfa->im->device < 0) return fail(REDGPU_EAPI, "dfa handle has no device image");
  if (n == 0) return REDGPU_OK;
  if (!walked) return fail(REDGPU_EAPI, "null buffer");
  if (!data && (offsets || stride)) return fail(REDGPU_EAPI, "null data buffer");
  if (offsets && stride > 16) return fail(REDGPU_EAPI, "with offsets, stride is the number of "
                                                       "trailing bytes to drop per line (0..16)");
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  Batch b{data, offsets, stride, n, nullptr, nullptr, nullptr};
  LaunchCfg cfg{dfa->numCUs, 0};
  hipError_t e = launchWalked(dfa->im->dev, b, do_leader ? 1 : 0,
                              reinterpret_cast<unsigned long long *>(walked), cfg,
                              static_cast<hipStream_t>(stream));
  tlsKernel = "k_walked";
  if (e != hipSuccess) return failHip(e, "kernel launch");
  return REDGPU_OK;
}

int redgpu_host_register(void *ptr, size_t bytes) {
  if (!ptr || !bytes) return fail(REDGPU_EAPI, "null buffer");
  hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
  if (e != hipSuccess) return failHip(e, "hipHostRegister");
  return REDGPU_OK;
}

int redgpu_host_unregister(void *ptr) {
  if (!ptr) return fail(REDGPU_EAPI, "null buffer");
  hipError_t e = hipHostUnregister(ptr);
  if (e != hipSuccess) return failHip(e, "hipHostUnregister");
  return REDGPU_OK;
}

void redgpu_thread_release(void) { hostStageReleaseThread(); }

void redgpu_host_route_counts(uint64_t counts[4]) { if (counts) hostRouteCounts(counts); }

uint64_t redgpu_scratch_entries(void) { return scratchEntries(); }

int redgpu_dfa_tune_dev(redgpu_dfa *dfa, const uint8_t *data, const uint64_t *offsets,
                        uint64_t stride, uint64_t n, void *stream) {
  if (!dfa) return fail(REDGPU_EAPI, "null dfa handle");
  if (dfa->im->device < 0) return fail(REDGPU_EAPI, "dfa handle has no device image");
  if (n == 0) return REDGPU_OK;
  if (!data && (offsets || stride)) return fail(REDGPU_EAPI, "null data buffer");
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  // a fused u8 table in LDS (<= 256 states) already takes the streaming kernels
  if (dfa->im->img.tableKind == REDGPU_TAB_LDS_FUSED_U8) return REDGPU_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const uint32_t nStates = dfa->im->img.nStates;
  uint32_t *dHist = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dHist), size_t(nStates) * 4), "hipMalloc histogram");
  auto done = [&](int rc) { (void)hipFree(dHist); return rc; };
  hipError_t e = hipMemsetAsync(dHist, 0, size_t(nStates) * 4, s);
  if (e != hipSuccess) return done(failHip(e, "hipMemsetAsync"));
  Batch b{data, offsets, stride, n, nullptr, nullptr, nullptr};
  LaunchCfg cfg{dfa->numCUs, 0};
  e = launchVisits(dfa->im->dev, b, dHist, cfg, s);
  tlsKernel = "k_visits";
  if (e != hipSuccess) return done(failHip(e, "kernel launch"));
  std::vector<uint32_t> hist(nStates);
  e = hipMemcpyAsync(hist.data(), dHist, size_t(nStates) * 4, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) return done(failHip(e, "histogram copy"));
  (void)hipFree(dHist);
  dHist = nullptr;

  // device index -> blob state id, then the same builder with the observed visits
  std::vector<double> measured(dfa->im->img.statesTotal, 0.0);
  for (uint32_t i = 0; i < nStates; ++i) measured[dfa->im->img.rawOf[i]] = double(hist[i]);
  DfaImage img;
  int code = REDGPU_OK;
  std::string err = buildImage(dfa->im->blob.data(), dfa->im->blob.size(), dfa->ldsTableMax,
                               (dfa->flags & REDGPU_F_FORCE_GLOBAL) != 0, img, code,
                               (dfa->flags & REDGPU_F_FORCE_HOT) != 0, &measured);
  if (!err.empty()) return fail(code, err);
  // all work queued on the device so far may still read the old tables
  e = hipDeviceSynchronize();
  if (e != hipSuccess) return failHip(e, "hipDeviceSynchronize");
  // copy on tune: other handles that share the image keep the one they were created with
  auto im = std::make_shared<SharedImage>();
  im->blob = dfa->im->blob;
  im->img = std::move(img);
  im->device = dfa->im->device;
  im->buildFlags = dfa->im->buildFlags;
  im->ldsTableMax = dfa->im->ldsTableMax;
  if (int rc = uploadImage(im.get())) return rc;
  dfa->im = std::move(im);
  return REDGPU_OK;
}

int redgpu_dfa_tune(redgpu_dfa *dfa, const uint8_t *data, const uint64_t *offsets, uint64_t stride,
                    uint64_t n) {
  if (!dfa) return fail(REDGPU_EAPI, "null dfa handle");
  if (dfa->im->device < 0) return fail(REDGPU_EAPI, "dfa handle has no device image");
  if (n == 0) return REDGPU_OK;
  if (offsets)
    if (int rc = checkOffsets(offsets, n)) return rc;
  const uint64_t total = offsets ? offsets[n] : stride * n;
  if (total && !data) return fail(REDGPU_EAPI, "null data buffer");
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  HostStage *st = nullptr;
  if (int rc = stageOf(dfa, &st)) return rc;
  st->beginCall(total);
  hipStream_t s = st->streams[0];
  uint8_t *dData = nullptr;
  uint64_t *dOff = nullptr;
  STAGE_TRY(st->get(kSlData, total, reinterpret_cast<void **>(&dData)), "hipMalloc data");
  if (offsets) {
    STAGE_TRY(st->get(kSlOff, (n + 1) * 8, reinterpret_cast<void **>(&dOff)), "hipMalloc offsets");
    STAGE_TRY(st->copyIn(dOff, offsets, (n + 1) * 8, 0), "copy offsets");
  }
  if (total) STAGE_TRY(st->copyIn(dData, data, total, 0), "copy data");
  return redgpu_dfa_tune_dev(dfa, dData, dOff, stride, n, s);  // synchronises
}

int redgpu_advance_batch_dev(const redgpu_dfa *dfa, const uint8_t *data, const uint64_t *offsets,
                             uint64_t stride, uint64_t n, uint32_t *state, int32_t *result,
                             void *stream) {
  if (!dfa) return fail(REDGPU_EAPI, "null dfa handle");
  if (dfa->im->device < 0) return fail(REDGPU_EAPI, "dfa handle has no device image");
  if (n == 0) return REDGPU_OK;
  if (!state) return fail(REDGPU_EAPI, "null state buffer");
  if (!result) return fail(REDGPU_EAPI, "null result buffer");
  if (!data && (offsets || stride)) return fail(REDGPU_EAPI, "null data buffer");
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  Batch b{data, offsets, stride, n, result, nullptr, nullptr};
  LaunchCfg cfg{dfa->numCUs, (dfa->flags & REDGPU_F_FORCE_GENERIC) ? 1 : 0};
  const char *name = "";
  hipError_t e = launchAdvance(dfa->im->dev, b, state, cfg, static_cast<hipStream_t>(stream), &name);
  tlsKernel = name;
  if (e != hipSuccess) return failHip(e, "kernel launch");
  return REDGPU_OK;
}

int redgpu_advance_batch(const redgpu_dfa *dfa, const uint8_t *data, const uint64_t *offsets,
                         uint64_t stride, uint64_t n, uint32_t *state, int32_t *result) {
  if (!dfa) return fail(REDGPU_EAPI, "null dfa handle");
  if (dfa->im->device < 0) return fail(REDGPU_EAPI, "dfa handle has no device image");
  if (n == 0) return REDGPU_OK;
  if (!state) return fail(REDGPU_EAPI, "null state buffer");
  if (!result) return fail(REDGPU_EAPI, "null result buffer");
  if (offsets)
    if (int rc = checkOffsets(offsets, n)) return rc;
  const uint64_t total = offsets ? offsets[n] : stride * n;
  if (total && !data) return fail(REDGPU_EAPI, "null data buffer");
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  HostStage *st = nullptr;
  if (int rc = stageOf(dfa, &st)) return rc;
  st->beginCall(total);
  hipStream_t s = st->streams[0];
  uint8_t *dData = nullptr;
  uint64_t *dOff = nullptr;
  uint32_t *dState = nullptr;
  int32_t *dRes = nullptr;
  STAGE_TRY(st->get(kSlData, total, reinterpret_cast<void **>(&dData)), "hipMalloc data");
  STAGE_TRY(st->get(kSlAux0, n * 4, reinterpret_cast<void **>(&dState)), "hipMalloc state");
  STAGE_TRY(st->get(kSlRes, n * 4, reinterpret_cast<void **>(&dRes)), "hipMalloc result");
  if (offsets) {
    STAGE_TRY(st->get(kSlOff, (n + 1) * 8, reinterpret_cast<void **>(&dOff)), "hipMalloc offsets");
    STAGE_TRY(st->copyIn(dOff, offsets, (n + 1) * 8, 0), "copy offsets");
  }
  if (total) STAGE_TRY(st->copyIn(dData, data, total, 0), "copy data");
  STAGE_TRY(st->copyIn(dState, state, n * 4, 0), "copy state");
  const int rc = redgpu_advance_batch_dev(dfa, dData, dOff, stride, n, dState, dRes, s);
  if (rc != REDGPU_OK) {
    (void)st->sync();
    return rc;
  }
  STAGE_TRY(st->copyOut(state, dState, n * 4, 0), "copy state back");
  STAGE_TRY(st->copyOut(result, dRes, n * 4, 0), "copy result");
  STAGE_TRY(st->syncStream(0), "hipStreamSynchronize");
  return REDGPU_OK;
}

int redgpu_check_batch_dev(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                           const uint64_t *offsets, uint64_t stride, uint64_t n,
                           int32_t *result, void *stream) {
  return runDev(dfa, kCheck, style, do_leader, data, offsets, stride, n, result, nullptr,
                nullptr, static_cast<hipStream_t>(stream));
}

int redgpu_match_batch_dev(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                           const uint64_t *offsets, uint64_t stride, uint64_t n,
                           int32_t *result, uint64_t *start, uint64_t *end, void *stream) {
  return runDev(dfa, kMatch, style, do_leader, data, offsets, stride, n, result, start, end,
                static_cast<hipStream_t>(stream));
}

int redgpu_check_batches_dev(const redgpu_dfa *dfa, int style, int do_leader,
                             const redgpu_batch *batches, uint32_t n_batches, void *stream) {
  return runDevMany(dfa, kCheck, style, do_leader, batches, n_batches,
                    static_cast<hipStream_t>(stream));
}

int redgpu_match_batches_dev(const redgpu_dfa *dfa, int style, int do_leader,
                             const redgpu_batch *batches, uint32_t n_batches, void *stream) {
  return runDevMany(dfa, kMatch, style, do_leader, batches, n_batches,
                    static_cast<hipStream_t>(stream));
}

int redgpu_scan_batch_dev(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                          const uint64_t *offsets, uint64_t stride, uint64_t n, int32_t *result,
                          void *stream) {
  return runDev(dfa, kScan, style, do_leader, data, offsets, stride, n, result, nullptr,
                nullptr, static_cast<hipStream_t>(stream));
}

} // extern "C"
