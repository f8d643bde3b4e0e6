// redgpu_internal.h - what the translation units behind include/redgpu.h share: the handle
// types, the thread-local error slot and the device scope.
#pragma once

#include <memory>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/redgpu.h"
#include "dfa_image.h"
#include "kernels.h"

// What is expensive to make - the validated blob copy, the repacked image and its device
// allocations - is immutable once built and shared between handles created from the same blob
// with the same options on the same device (the loader cache below).
struct SharedImage {
  std::vector<uint8_t> blob;  // our own copy (Executable(gCopyTag,..) semantics)
  redgpu::DfaImage img;
  int device = REDGPU_DEVICE_NONE;
  uint32_t buildFlags = 0;    // the flags / LDS budget the image was built with (cache key)
  uint32_t ldsTableMax = 0;
  void *dTable = nullptr;
  void *dResult = nullptr;
  void *dEquivLeader = nullptr;
  redgpu::DevDfa dev{};
  ~SharedImage();
};

inline SharedImage::~SharedImage() {
  if (device < 0) return;
  int prev = -1;
  const bool sw = hipGetDevice(&prev) == hipSuccess && prev != device &&
                  hipSetDevice(device) == hipSuccess;
  // kernels queued with this image (asynchronous _dev calls the caller has not waited for) may
  // still be reading its tables: drain the device before the tables go (the reference's rule is
  // "exec must outlive the matcher", include/Matcher.h:772 - here breaking it is merely slow)
  (void)hipDeviceSynchronize();
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

namespace redgpu {

inline thread_local std::string tlsError;
inline thread_local const char *tlsKernel = "";

inline int fail(int code, const std::string &msg) {
  tlsError = msg;
  return code;
}

inline int failHip(hipError_t e, const char *what) {
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


}  // namespace redgpu
