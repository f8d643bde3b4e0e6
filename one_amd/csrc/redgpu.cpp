// redgpu.cpp - the C-ABI of include/redgpu.h over the gfx950 kernels.  Host C++ (built with
// hipcc for the HIP runtime API).  There is no CPU compute path in this file or behind it:
// every batch entry point either launches a kernel or fails.
#include "../../include/redgpu.h"

#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include <hip/hip_runtime.h>

#include "dfa_image.h"
#include "kernels.h"

using namespace redgpu;

// What is expensive to make - the validated blob copy, the repacked image and its device
// allocations - is immutable once built and shared between handles created from the same blob
// with the same options on the same device (the loader cache below).
struct SharedImage {
  std::vector<uint8_t> blob;  // our own copy (Executable(gCopyTag,..) semantics)
  DfaImage img;
  int device = REDGPU_DEVICE_NONE;
  uint32_t buildFlags = 0;    // the flags / LDS budget the image was built with (cache key)
  uint32_t ldsTableMax = 0;
  void *dTable = nullptr;
  void *dResult = nullptr;
  void *dEquivLeader = nullptr;
  DevDfa dev{};
  ~SharedImage();
};

SharedImage::~SharedImage() {
  if (device < 0) return;
  int prev = -1;
  const bool sw = hipGetDevice(&prev) == hipSuccess && prev != device &&
                  hipSetDevice(device) == hipSuccess;
  if (dTable) (void)hipFree(dTable);
  if (dResult) (void)hipFree(dResult);
  if (dEquivLeader) (void)hipFree(dEquivLeader);
  if (sw) (void)hipSetDevice(prev);
}

struct redgpu_dfa {
  std::shared_ptr<SharedImage> im;
  int numCUs = 0;
  uint32_t flags = 0;
  uint32_t ldsTableMax = 0;
};

namespace {

thread_local std::string tlsError;
thread_local const char *tlsKernel = "";

int fail(int code, const std::string &msg) {
  tlsError = msg;
  return code;
}

int failHip(hipError_t e, const char *what) {
  tlsError = std::string(what) + ": " + hipGetErrorString(e);
  return REDGPU_EHIP;
}

#define HIP_TRY(expr, what)                          \
  do {                                               \
    hipError_t e_ = (expr);                          \
    if (e_ != hipSuccess) return failHip(e_, what);  \
  } while (0)

// RAII: run on the handle's device, restore the caller's current device afterwards
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceScope(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) {
      err = hipSetDevice(dev);
      switched = (err == hipSuccess);
    }
  }
  ~DeviceScope() {
    if (switched) (void)hipSetDevice(prev);
  }
};

int checkStyle(int style) {
  if (style < REDGPU_STY_INSTANT || style > REDGPU_STY_FULL)
    return fail(REDGPU_EEXEC, "unsupported style");  // lib/Matcher.cpp:45
  return REDGPU_OK;
}

int runDev(const redgpu_dfa *dfa, int verb, int style, int doLeader, const uint8_t *data,
           const uint64_t *offsets, uint64_t stride, uint64_t n, int32_t *result,
           uint64_t *start, uint64_t *end, hipStream_t stream) {
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
  LaunchCfg cfg{dfa->numCUs, (dfa->flags & REDGPU_F_FORCE_GENERIC) ? 1 : 0,
                (dfa->flags & REDGPU_F_NO_BUCKETING) ? 1 : 0,
                (dfa->flags & REDGPU_F_FORCE_STREAM) ? 1 : 0,
                (dfa->flags & REDGPU_F_NO_CHUNKING) ? 1 : 0,
                (dfa->flags & REDGPU_F_FORCE_CHUNKING) ? 1 : 0,
                (dfa->flags & REDGPU_F_STREAM_CHAINS_2) ? 2
                : (dfa->flags & REDGPU_F_STREAM_CHAINS_4) ? 4 : 0};
  const char *name = "";
  hipError_t e = launchBatch(dfa->im->dev, b, verb, style, doLeader ? 1 : 0, cfg, stream, &name);
  tlsKernel = name;
  if (e != hipSuccess) return failHip(e, "kernel launch");
  return REDGPU_OK;
}

// host-buffer form: stage through device memory on a private stream
int runHost(const redgpu_dfa *dfa, int verb, int style, int doLeader, const uint8_t *data,
            const uint64_t *offsets, uint64_t stride, uint64_t n, int32_t *result,
            uint64_t *start, uint64_t *end) {
  if (!dfa) return fail(REDGPU_EAPI, "null dfa handle");
  if (dfa->im->device < 0) return fail(REDGPU_EAPI, "dfa handle has no device image");
  if (int rc = checkStyle(style)) return rc;
  if (n == 0) return REDGPU_OK;
  if (!result) return fail(REDGPU_EAPI, "null result buffer");
  uint64_t total = offsets ? offsets[n] : stride * n;
  if (offsets) {
    for (uint64_t i = 0; i < n; ++i)
      if (offsets[i] > offsets[i + 1]) return fail(REDGPU_EAPI, "offsets not monotonic");
  }
  if (total && !data) return fail(REDGPU_EAPI, "null data buffer");
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");

  hipStream_t s = nullptr;
  uint8_t *dData = nullptr;
  uint64_t *dOff = nullptr, *dStart = nullptr, *dEnd = nullptr;
  int32_t *dRes = nullptr;
  int rc = REDGPU_OK;
  auto cleanup = [&]() {
    if (dData) (void)hipFree(dData);
    if (dOff) (void)hipFree(dOff);
    if (dRes) (void)hipFree(dRes);
    if (dStart) (void)hipFree(dStart);
    if (dEnd) (void)hipFree(dEnd);
    if (s) (void)hipStreamDestroy(s);
  };
#define HOST_TRY(expr, what)                                         \
  do {                                                               \
    hipError_t e_ = (expr);                                          \
    if (e_ != hipSuccess) { rc = failHip(e_, what); cleanup(); return rc; } \
  } while (0)
  HOST_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");
  // +16 so that 16-byte loads of the last line never leave the allocation
  HOST_TRY(hipMalloc(reinterpret_cast<void **>(&dData), total + 16), "hipMalloc data");
  HOST_TRY(hipMalloc(reinterpret_cast<void **>(&dRes), n * sizeof(int32_t)), "hipMalloc result");
  if (offsets) {
    HOST_TRY(hipMalloc(reinterpret_cast<void **>(&dOff), (n + 1) * sizeof(uint64_t)),
             "hipMalloc offsets");
    HOST_TRY(hipMemcpyAsync(dOff, offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s),
             "copy offsets");
  }
  if (start) HOST_TRY(hipMalloc(reinterpret_cast<void **>(&dStart), n * 8), "hipMalloc start");
  if (end) HOST_TRY(hipMalloc(reinterpret_cast<void **>(&dEnd), n * 8), "hipMalloc end");
  if (total)
    HOST_TRY(hipMemcpyAsync(dData, data, total, hipMemcpyHostToDevice, s), "copy data");
  rc = runDev(dfa, verb, style, doLeader, dData, dOff, stride, n, dRes, dStart, dEnd, s);
  if (rc != REDGPU_OK) { cleanup(); return rc; }
  HOST_TRY(hipMemcpyAsync(result, dRes, n * sizeof(int32_t), hipMemcpyDeviceToHost, s),
           "copy result");
  if (start) HOST_TRY(hipMemcpyAsync(start, dStart, n * 8, hipMemcpyDeviceToHost, s), "copy start");
  if (end) HOST_TRY(hipMemcpyAsync(end, dEnd, n * 8, hipMemcpyDeviceToHost, s), "copy end");
  HOST_TRY(hipStreamSynchronize(s), "hipStreamSynchronize");
#undef HOST_TRY
  cleanup();
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
  LaunchCfg cfg{dfa->numCUs, 0};
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
  uint8_t eqLead[512];
  std::memcpy(eqLead, img.equiv, 256);
  std::memcpy(eqLead + 256, img.leader, 256);
  const size_t tabBytes = (img.table.size() + 15) & ~size_t(15);
  hipError_t e;
  if ((e = hipMalloc(&im->dTable, tabBytes + 16)) != hipSuccess) return failHip(e, "hipMalloc table");
  if ((e = hipMalloc(&im->dResult, img.nStates * sizeof(int32_t) + 16)) != hipSuccess)
    return failHip(e, "hipMalloc result");
  if ((e = hipMalloc(&im->dEquivLeader, 512 + 64)) != hipSuccess) return failHip(e, "hipMalloc equiv");
  if ((e = hipMemset(im->dTable, 0, tabBytes + 16)) != hipSuccess) return failHip(e, "hipMemset");
  if ((e = hipMemcpy(im->dTable, img.table.data(), img.table.size(), hipMemcpyHostToDevice)) !=
      hipSuccess)
    return failHip(e, "upload table");
  if ((e = hipMemcpy(im->dResult, img.result.data(), img.nStates * sizeof(int32_t),
                     hipMemcpyHostToDevice)) != hipSuccess)
    return failHip(e, "upload result");
  if ((e = hipMemcpy(im->dEquivLeader, eqLead, 512, hipMemcpyHostToDevice)) != hipSuccess)
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
  d.startLeadWord = img.startLeadWord;
  d.startLeadCount = img.startLeadCount;
  d.startFreeWord = img.startFreeWord;
  d.startFreeCount = img.startFreeCount;
  d.start2LeadWord = img.start2LeadWord;
  d.start2LeadCount = img.start2LeadCount;
  d.start2FreeWord = img.start2FreeWord;
  d.start2FreeCount = img.start2FreeCount;
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
  const uint64_t total = offsets ? offsets[n] : stride * n;
  if (total && !data) return fail(REDGPU_EAPI, "null data buffer");
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  uint8_t *dData = nullptr;
  uint64_t *dOff = nullptr, *dCnt = nullptr, *dStart = nullptr, *dEnd = nullptr;
  int32_t *dRes = nullptr;
  auto cleanup = [&]() {
    for (void *q : {(void *)dData, (void *)dOff, (void *)dCnt, (void *)dRes, (void *)dStart, (void *)dEnd})
      if (q) (void)hipFree(q);
  };
  int rc = REDGPU_OK;
#define CH_TRY(expr, what)                                                    \
  do {                                                                        \
    hipError_t e_ = (expr);                                                   \
    if (e_ != hipSuccess) { rc = failHip(e_, what); cleanup(); return rc; }   \
  } while (0)
  const uint64_t slots = n * cap;
  CH_TRY(hipMalloc(reinterpret_cast<void **>(&dData), total + 16), "hipMalloc data");
  CH_TRY(hipMalloc(reinterpret_cast<void **>(&dCnt), n * 8), "hipMalloc counts");
  CH_TRY(hipMalloc(reinterpret_cast<void **>(&dRes), (slots + 1) * 4), "hipMalloc result");
  if (start) CH_TRY(hipMalloc(reinterpret_cast<void **>(&dStart), (slots + 1) * 8), "hipMalloc start");
  if (end) CH_TRY(hipMalloc(reinterpret_cast<void **>(&dEnd), (slots + 1) * 8), "hipMalloc end");
  if (offsets) {
    CH_TRY(hipMalloc(reinterpret_cast<void **>(&dOff), (n + 1) * 8), "hipMalloc offsets");
    CH_TRY(hipMemcpy(dOff, offsets, (n + 1) * 8, hipMemcpyHostToDevice), "copy offsets");
  }
  if (total) CH_TRY(hipMemcpy(dData, data, total, hipMemcpyHostToDevice), "copy data");
  rc = collectDev(dfa, listVerb, dData, dOff, stride, n, cap, dCnt, dRes, dStart, dEnd, nullptr);
  if (rc != REDGPU_OK) { cleanup(); return rc; }
  CH_TRY(hipDeviceSynchronize(), "hipDeviceSynchronize");
  CH_TRY(hipMemcpy(counts, dCnt, n * 8, hipMemcpyDeviceToHost), "copy counts");
  if (slots) {
    CH_TRY(hipMemcpy(result, dRes, slots * 4, hipMemcpyDeviceToHost), "copy result");
    if (start) CH_TRY(hipMemcpy(start, dStart, slots * 8, hipMemcpyDeviceToHost), "copy start");
    if (end) CH_TRY(hipMemcpy(end, dEnd, slots * 8, hipMemcpyDeviceToHost), "copy end");
  }
#undef CH_TRY
  cleanup();
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
  const uint64_t total = offsets ? offsets[n] : stride * n;
  if (offsets) {
    for (uint64_t i = 0; i < n; ++i)
      if (offsets[i] > offsets[i + 1]) return fail(REDGPU_EAPI, "offsets not monotonic");
  }
  if (total && !data) return fail(REDGPU_EAPI, "null data buffer");
  if (repl_len && !repl) return fail(REDGPU_EAPI, "null replacement");
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  uint8_t *dData = nullptr, *dRepl = nullptr, *dOut = nullptr;
  uint64_t *dOff = nullptr, *dCnt = nullptr, *dOutOff = nullptr;
  auto cleanup = [&]() {
    for (void *q : {(void *)dData, (void *)dRepl, (void *)dOut, (void *)dOff, (void *)dCnt,
                    (void *)dOutOff})
      if (q) (void)hipFree(q);
  };
  int rc = REDGPU_OK;
#define RP_TRY(expr, what)                                                    \
  do {                                                                        \
    hipError_t e_ = (expr);                                                   \
    if (e_ != hipSuccess) { rc = failHip(e_, what); cleanup(); return rc; }   \
  } while (0)
  RP_TRY(hipMalloc(reinterpret_cast<void **>(&dData), total + 16), "hipMalloc data");
  RP_TRY(hipMalloc(reinterpret_cast<void **>(&dRepl), repl_len + 16), "hipMalloc repl");
  RP_TRY(hipMalloc(reinterpret_cast<void **>(&dCnt), n * 8), "hipMalloc counts");
  RP_TRY(hipMalloc(reinterpret_cast<void **>(&dOutOff), (n + 1) * 8), "hipMalloc out offsets");
  if (out && out_cap) RP_TRY(hipMalloc(reinterpret_cast<void **>(&dOut), out_cap), "hipMalloc out");
  if (offsets) {
    RP_TRY(hipMalloc(reinterpret_cast<void **>(&dOff), (n + 1) * 8), "hipMalloc offsets");
    RP_TRY(hipMemcpy(dOff, offsets, (n + 1) * 8, hipMemcpyHostToDevice), "copy offsets");
  }
  if (total) RP_TRY(hipMemcpy(dData, data, total, hipMemcpyHostToDevice), "copy data");
  if (repl_len) RP_TRY(hipMemcpy(dRepl, repl, repl_len, hipMemcpyHostToDevice), "copy repl");
  rc = redgpu_replace_batch_dev(dfa, style, do_leader, dData, dOff, stride, n, dRepl, repl_len,
                                max_count, dCnt, dOutOff, dOut, dOut ? out_cap : 0, nullptr);
  if (rc != REDGPU_OK) { cleanup(); return rc; }
  RP_TRY(hipDeviceSynchronize(), "hipDeviceSynchronize");
  RP_TRY(hipMemcpy(counts, dCnt, n * 8, hipMemcpyDeviceToHost), "copy counts");
  RP_TRY(hipMemcpy(out_offsets, dOutOff, (n + 1) * 8, hipMemcpyDeviceToHost), "copy out offsets");
  if (dOut) {
    // the lines that fit are a prefix (offsets are monotone): copy up to the last one that does
    uint64_t lo = 0, hi = n;  // largest k with out_offsets[k] <= out_cap
    while (lo < hi) {
      const uint64_t mid = (lo + hi + 1) / 2;
      if (out_offsets[mid] <= out_cap) lo = mid; else hi = mid - 1;
    }
    if (out_offsets[lo])
      RP_TRY(hipMemcpy(out, dOut, out_offsets[lo], hipMemcpyDeviceToHost), "copy out");
  }
#undef RP_TRY
  cleanup();
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
  // counts u32[nChunks] then bases u64[nChunks], stream-ordered so the call stays asynchronous
  const size_t countBytes = (size_t(nChunks) * 4 + 15) & ~size_t(15);
  HIP_TRY(hipMallocAsync(&scratch, countBytes + size_t(nChunks) * 8 + 16, s), "hipMallocAsync");
  uint32_t *counts = static_cast<uint32_t *>(scratch);
  uint64_t *bases = reinterpret_cast<uint64_t *>(static_cast<uint8_t *>(scratch) + countBytes);
  hipError_t e = launchSplitLines(data, len, delim, offsets, cap, n_lines, counts, bases, s);
  tlsKernel = "k_split_scatter";
  hipError_t e2 = hipFreeAsync(scratch, s);
  if (e != hipSuccess) return failHip(e, "kernel launch");
  if (e2 != hipSuccess) return failHip(e2, "hipFreeAsync");
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
  uint8_t *dData = nullptr;
  uint64_t *dOff = nullptr, *dN = nullptr;
  auto cleanup = [&]() {
    for (void *q : {(void *)dData, (void *)dOff, (void *)dN})
      if (q) (void)hipFree(q);
  };
  int rc = REDGPU_OK;
#define SP_TRY(expr, what)                                                    \
  do {                                                                        \
    hipError_t e_ = (expr);                                                   \
    if (e_ != hipSuccess) { rc = failHip(e_, what); cleanup(); return rc; }   \
  } while (0)
  SP_TRY(hipMalloc(reinterpret_cast<void **>(&dData), len + 16), "hipMalloc data");
  SP_TRY(hipMalloc(reinterpret_cast<void **>(&dN), 8), "hipMalloc count");
  if (len) SP_TRY(hipMemcpy(dData, data, len, hipMemcpyHostToDevice), "copy data");
  // count first (room for no line at all), then size the device offsets to what will be kept
  SP_TRY(hipMalloc(reinterpret_cast<void **>(&dOff), 8), "hipMalloc offsets");
  rc = redgpu_split_lines_dev(dfa, dData, len, delim, dOff, 0, dN, nullptr);
  if (rc != REDGPU_OK) { cleanup(); return rc; }
  SP_TRY(hipDeviceSynchronize(), "hipDeviceSynchronize");
  SP_TRY(hipMemcpy(n_lines, dN, 8, hipMemcpyDeviceToHost), "copy count");
  const uint64_t got = *n_lines < cap ? *n_lines : cap;
  if (got) {
    (void)hipFree(dOff);
    dOff = nullptr;
    SP_TRY(hipMalloc(reinterpret_cast<void **>(&dOff), (got + 1) * 8), "hipMalloc offsets");
    rc = redgpu_split_lines_dev(dfa, dData, len, delim, dOff, got, dN, nullptr);
    if (rc != REDGPU_OK) { cleanup(); return rc; }
    SP_TRY(hipDeviceSynchronize(), "hipDeviceSynchronize");
  }
  SP_TRY(hipMemcpy(offsets, dOff, (got + 1) * 8, hipMemcpyDeviceToHost), "copy offsets");
#undef SP_TRY
  cleanup();
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
  const uint64_t total = offsets ? offsets[n] : stride * n;
  if (offsets) {
    for (uint64_t i = 0; i < n; ++i)
      if (offsets[i] > offsets[i + 1]) return fail(REDGPU_EAPI, "offsets not monotonic");
  }
  if (total && !data) return fail(REDGPU_EAPI, "null data buffer");
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  uint8_t *dData = nullptr;
  uint64_t *dOff = nullptr;
  auto cleanup = [&]() {
    if (dData) (void)hipFree(dData);
    if (dOff) (void)hipFree(dOff);
  };
  int rc = REDGPU_OK;
#define TU_TRY(expr, what)                                                    \
  do {                                                                        \
    hipError_t e_ = (expr);                                                   \
    if (e_ != hipSuccess) { rc = failHip(e_, what); cleanup(); return rc; }   \
  } while (0)
  TU_TRY(hipMalloc(reinterpret_cast<void **>(&dData), total + 16), "hipMalloc data");
  if (offsets) {
    TU_TRY(hipMalloc(reinterpret_cast<void **>(&dOff), (n + 1) * 8), "hipMalloc offsets");
    TU_TRY(hipMemcpy(dOff, offsets, (n + 1) * 8, hipMemcpyHostToDevice), "copy offsets");
  }
  if (total) TU_TRY(hipMemcpy(dData, data, total, hipMemcpyHostToDevice), "copy data");
#undef TU_TRY
  rc = redgpu_dfa_tune_dev(dfa, dData, dOff, stride, n, nullptr);
  cleanup();
  return rc;
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
  const uint64_t total = offsets ? offsets[n] : stride * n;
  if (offsets) {
    for (uint64_t i = 0; i < n; ++i)
      if (offsets[i] > offsets[i + 1]) return fail(REDGPU_EAPI, "offsets not monotonic");
  }
  if (total && !data) return fail(REDGPU_EAPI, "null data buffer");
  DeviceScope scope(dfa->im->device);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  uint8_t *dData = nullptr;
  uint64_t *dOff = nullptr;
  uint32_t *dState = nullptr;
  int32_t *dRes = nullptr;
  auto cleanup = [&]() {
    for (void *q : {(void *)dData, (void *)dOff, (void *)dState, (void *)dRes})
      if (q) (void)hipFree(q);
  };
  int rc = REDGPU_OK;
#define AD_TRY(expr, what)                                                    \
  do {                                                                        \
    hipError_t e_ = (expr);                                                   \
    if (e_ != hipSuccess) { rc = failHip(e_, what); cleanup(); return rc; }   \
  } while (0)
  AD_TRY(hipMalloc(reinterpret_cast<void **>(&dData), total + 16), "hipMalloc data");
  AD_TRY(hipMalloc(reinterpret_cast<void **>(&dState), n * 4), "hipMalloc state");
  AD_TRY(hipMalloc(reinterpret_cast<void **>(&dRes), n * 4), "hipMalloc result");
  if (offsets) {
    AD_TRY(hipMalloc(reinterpret_cast<void **>(&dOff), (n + 1) * 8), "hipMalloc offsets");
    AD_TRY(hipMemcpy(dOff, offsets, (n + 1) * 8, hipMemcpyHostToDevice), "copy offsets");
  }
  if (total) AD_TRY(hipMemcpy(dData, data, total, hipMemcpyHostToDevice), "copy data");
  AD_TRY(hipMemcpy(dState, state, n * 4, hipMemcpyHostToDevice), "copy state");
  rc = redgpu_advance_batch_dev(dfa, dData, dOff, stride, n, dState, dRes, nullptr);
  if (rc != REDGPU_OK) { cleanup(); return rc; }
  AD_TRY(hipDeviceSynchronize(), "hipDeviceSynchronize");
  AD_TRY(hipMemcpy(state, dState, n * 4, hipMemcpyDeviceToHost), "copy state back");
  AD_TRY(hipMemcpy(result, dRes, n * 4, hipMemcpyDeviceToHost), "copy result");
#undef AD_TRY
  cleanup();
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

int redgpu_scan_batch_dev(const redgpu_dfa *dfa, int style, int do_leader, const uint8_t *data,
                          const uint64_t *offsets, uint64_t stride, uint64_t n, int32_t *result,
                          void *stream) {
  return runDev(dfa, kScan, style, do_leader, data, offsets, stride, n, result, nullptr,
                nullptr, static_cast<hipStream_t>(stream));
}

} // extern "C"
