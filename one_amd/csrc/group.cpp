// group.cpp - several GPUs of one node behind the C-ABI (include/redgpu.h, "several GPUs").
//
// The reference scales by N host threads over one shared read-only Red
// (/root/reference/quol/red/tools/thr_red.cpp:84-91).  Here a group holds one image of the same
// blob per device; a batch is cut into contiguous shards (equal lines for a fixed stride, equal
// bytes for ragged lines), every device scans its shard with no data-path exchange, and only the
// per-line results travel: to the caller's arrays (host form), or as compact records over xGMI
// to the root device (device form; peer copies or RCCL send / receive), widened there.
// Built with hipcc as HIP (the two pack / unpack kernels live here).
#include <dlfcn.h>

#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>

#include "redgpu_internal.h"

using namespace redgpu;

namespace {

// ---- compact records: three planes per shard [result n*rw][start n*pw][end n*pw] ---------------
__device__ __forceinline__ void putN(uint8_t *p, uint64_t i, int w, uint64_t v) {
  if (w == 1) p[i] = uint8_t(v);
  else if (w == 2) reinterpret_cast<uint16_t *>(p)[i] = uint16_t(v);
  else if (w == 4) reinterpret_cast<uint32_t *>(p)[i] = uint32_t(v);
  else reinterpret_cast<uint64_t *>(p)[i] = v;
}
__device__ __forceinline__ uint64_t getN(const uint8_t *p, uint64_t i, int w) {
  if (w == 1) return p[i];
  if (w == 2) return reinterpret_cast<const uint16_t *>(p)[i];
  if (w == 4) return reinterpret_cast<const uint32_t *>(p)[i];
  return reinterpret_cast<const uint64_t *>(p)[i];
}
// planes start on 16-byte boundaries
__host__ __device__ inline uint64_t planeBytes(uint64_t n, int w) { return (n * uint64_t(w) + 15) & ~15ull; }

__global__ void __launch_bounds__(256)
k_pack(const int32_t *res, const uint64_t *start, const uint64_t *end, uint64_t n, int rw, int pw,
       uint8_t *rec) {
  uint8_t *pr = rec;
  uint8_t *ps = pr + planeBytes(n, rw);
  uint8_t *pe = ps + (start ? planeBytes(n, pw) : 0);
  const uint64_t step = uint64_t(gridDim.x) * 256;
  for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += step) {
    putN(pr, i, rw, uint64_t(uint32_t(res[i])));
    if (start) putN(ps, i, pw, start[i]);
    if (end) putN(pe, i, pw, end[i]);
  }
}

__global__ void __launch_bounds__(256)
k_unpack(const uint8_t *rec, uint64_t n, int rw, int pw, int32_t *res, uint64_t *start,
         uint64_t *end) {
  const uint8_t *pr = rec;
  const uint8_t *ps = pr + planeBytes(n, rw);
  const uint8_t *pe = ps + (start ? planeBytes(n, pw) : 0);
  const uint64_t step = uint64_t(gridDim.x) * 256;
  for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += step) {
    res[i] = int32_t(uint32_t(getN(pr, i, rw)));  // results are >= 0 (include/Types.h:22)
    if (start) start[i] = getN(ps, i, pw);
    if (end) end[i] = getN(pe, i, pw);
  }
}

int widthFor(uint64_t maxValue) {
  return maxValue <= 0xffull ? 1 : maxValue <= 0xffffull ? 2 : maxValue <= 0xffffffffull ? 4 : 8;
}

// ---- RCCL, loaded on first use (the library must not need it to scan on one GPU) ----------------
struct Rccl {
  void *so = nullptr;
  int (*CommInitAll)(void **, int, const int *) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool ok = false;
};

Rccl &rccl() {
  static Rccl r = [] {
    Rccl x;
    for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      x.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (x.so) break;
    }
    if (!x.so) return x;
#define RCCL_SYM(field, sym) x.field = reinterpret_cast<decltype(x.field)>(dlsym(x.so, sym))
    RCCL_SYM(CommInitAll, "ncclCommInitAll");
    RCCL_SYM(CommDestroy, "ncclCommDestroy");
    RCCL_SYM(GroupStart, "ncclGroupStart");
    RCCL_SYM(GroupEnd, "ncclGroupEnd");
    RCCL_SYM(Send, "ncclSend");
    RCCL_SYM(Recv, "ncclRecv");
    RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RCCL_SYM
    x.ok = x.CommInitAll && x.CommDestroy && x.GroupStart && x.GroupEnd && x.Send && x.Recv;
    return x;
  }();
  return r;
}
constexpr int kNcclUint8 = 1;  // ncclDataType_t: ncclInt8 0, ncclUint8 1

int failRccl(int rc, const char *what) {
  const Rccl &r = rccl();
  tlsError = std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error");
  return REDGPU_ERCCL;
}

struct Member {
  redgpu_dfa *dfa = nullptr;
  int device = -1;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
  hipEvent_t ready = nullptr;  // "the shard's producer stream has reached the call"
  // grow-only working buffers of the device form
  int32_t *res = nullptr;
  uint64_t *start = nullptr, *end = nullptr;
  uint8_t *rec = nullptr;
  size_t resCap = 0, startCap = 0, endCap = 0, recCap = 0;
  void *comm = nullptr;  // ncclComm_t
};

hipError_t grow(void **p, size_t *cap, size_t bytes) {
  if (*p && *cap >= bytes) return hipSuccess;
  if (*p) {
    (void)hipDeviceSynchronize();  // an earlier call's pack / copy may still be reading it
    (void)hipFree(*p);
  }
  *p = nullptr;
  *cap = 0;
  const size_t want = bytes + bytes / 4 + 256;
  hipError_t e = hipMalloc(p, want);
  if (e == hipSuccess) *cap = want;
  return e;
}

}  // namespace

// One long-lived host thread per shard > 0 of the HOST form (the caller's thread takes shard 0):
// a worker keeps its per-thread staging (host_stage.h) from call to call, as one of thr_red's
// workers would - threads made per call would allocate and free it every time.
struct Worker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<void()> job;
  bool pending = false, stop = false, done = true;
  void loop() {
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      cv.wait(lk, [&] { return pending || stop; });
      if (stop) return;
      pending = false;
      lk.unlock();
      job();
      lk.lock();
      done = true;
      cv.notify_all();
    }
  }
  void post(std::function<void()> f) {
    std::lock_guard<std::mutex> lk(m);
    job = std::move(f);
    pending = true;
    done = false;
    cv.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return done; });
  }
};

struct redgpu_group {
  std::vector<Member> m;
  std::vector<std::unique_ptr<Worker>> workers;  // [k - 1] serves shard k of the host form
  std::mutex hostMu;        // the host form: one call at a time per group
  std::mutex mu;            // the device form: one call at a time
  uint8_t *rootRec = nullptr;
  size_t rootRecCap = 0;
  hipEvent_t consumed = nullptr;  // root: the previous call's records have been widened
  bool consumedValid = false;
  bool distinct = true;           // no device named twice (RCCL needs that)
  bool commsUp = false;
};

extern "C" {

int redgpu_group_create(const void *reda, size_t len, const redgpu_opts *opts,
                        const int32_t *devices, uint32_t n_devices, redgpu_group **out) {
  if (!out) return fail(REDGPU_EAPI, "null out pointer");
  *out = nullptr;
  if (!devices || n_devices == 0) return fail(REDGPU_EAPI, "empty device list");
  if (n_devices > 64) return fail(REDGPU_ELIMIT, "too many devices");
  auto g = std::make_unique<redgpu_group>();
  g->m.resize(n_devices);
  auto undo = [&]() {
    for (Member &mb : g->m)
      if (mb.dfa) redgpu_dfa_destroy(mb.dfa);
  };
  for (uint32_t i = 0; i < n_devices; ++i) {
    if (devices[i] < 0) { undo(); return fail(REDGPU_EAPI, "group devices must be HIP ordinals"); }
    for (uint32_t k = 0; k < i; ++k)
      if (devices[k] == devices[i]) g->distinct = false;
    redgpu_opts o{};
    if (opts) o = *opts;
    o.device = devices[i];
    const int rc = redgpu_dfa_create(reda, len, &o, &g->m[i].dfa);
    if (rc != REDGPU_OK) { undo(); return rc; }
    g->m[i].device = devices[i];
  }
  // streams, events and peer access (best effort: without it peer copies are staged)
  for (uint32_t i = 0; i < n_devices; ++i) {
    DeviceScope scope(devices[i]);
    hipError_t e = scope.err;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->m[i].stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->m[i].done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->m[i].ready, hipEventDisableTiming);
    if (e == hipSuccess && i == 0) e = hipEventCreateWithFlags(&g->consumed, hipEventDisableTiming);
    if (e != hipSuccess) {
      redgpu_group *raw = g.release();
      redgpu_group_destroy(raw);
      return failHip(e, "group streams");
    }
    if (devices[i] != devices[0]) {
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, devices[i], devices[0]) == hipSuccess && can)
        if (hipDeviceEnablePeerAccess(devices[0], 0) != hipSuccess) (void)hipGetLastError();
    }
  }
  for (uint32_t i = 1; i < n_devices; ++i) {
    g->workers.emplace_back(new Worker());
    Worker *w = g->workers.back().get();
    w->th = std::thread([w] { w->loop(); });
  }
  *out = g.release();
  return REDGPU_OK;
}

void redgpu_group_destroy(redgpu_group *g) {
  if (!g) return;
  for (auto &w : g->workers) {
    {
      std::lock_guard<std::mutex> lk(w->m);
      w->stop = true;
      w->cv.notify_all();
    }
    if (w->th.joinable()) w->th.join();  // its staging goes with the thread
  }
  for (Member &mb : g->m) {
    if (mb.device >= 0) {
      DeviceScope scope(mb.device);
      if (mb.stream) (void)hipStreamSynchronize(mb.stream);
      if (mb.comm && rccl().ok) (void)rccl().CommDestroy(mb.comm);
      for (void *p : {(void *)mb.res, (void *)mb.start, (void *)mb.end, (void *)mb.rec})
        if (p) (void)hipFree(p);
      if (mb.stream) {
        scratchDrop(mb.device, mb.stream);
        (void)hipStreamDestroy(mb.stream);
      }
      if (mb.done) (void)hipEventDestroy(mb.done);
      if (mb.ready) (void)hipEventDestroy(mb.ready);
    }
    if (mb.dfa) redgpu_dfa_destroy(mb.dfa);
  }
  if (!g->m.empty() && g->m[0].device >= 0) {
    DeviceScope scope(g->m[0].device);
    if (g->rootRec) (void)hipFree(g->rootRec);
    if (g->consumed) (void)hipEventDestroy(g->consumed);
  }
  delete g;
}

uint32_t redgpu_group_size(const redgpu_group *g) { return g ? uint32_t(g->m.size()) : 0; }

// ---- the record format on its own (one process per GPU: one_amd/sharding.py) -----------------
static bool widthsOk(int rw, int pw) {
  return (rw == 1 || rw == 2 || rw == 4) && (pw == 1 || pw == 2 || pw == 4 || pw == 8);
}

uint64_t redgpu_records_bytes(uint64_t n, int rw, int pw, int withStart) {
  if (!widthsOk(rw, pw)) return 0;
  return planeBytes(n, rw) + (withStart ? planeBytes(n, pw) : 0) + planeBytes(n, pw);
}

static int recordsLaunch(int32_t device, bool pack, const void *rec, uint64_t n, int rw, int pw,
                         const int32_t *res, const uint64_t *start, const uint64_t *end,
                         void *stream) {
  if (!widthsOk(rw, pw)) return fail(REDGPU_EAPI, "record widths must be 1/2/4 (result) and 1/2/4/8 (positions)");
  if (n == 0) return REDGPU_OK;
  if (!rec || !res || !end) return fail(REDGPU_EAPI, "null records / result / end buffer");
  int dev = device;
  if (dev == REDGPU_DEVICE_CURRENT) {
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return failHip(e, "hipGetDevice");
  }
  DeviceScope scope(dev);
  if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
  const uint64_t want = (n + 255) / 256;
  const uint32_t blocks = uint32_t(want < 4096 ? want : 4096);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (pack)
    hipLaunchKernelGGL(k_pack, dim3(blocks), dim3(256), 0, st, res, start, end, n, rw, pw,
                       static_cast<uint8_t *>(const_cast<void *>(rec)));
  else
    hipLaunchKernelGGL(k_unpack, dim3(blocks), dim3(256), 0, st, static_cast<const uint8_t *>(rec),
                       n, rw, pw, const_cast<int32_t *>(res), const_cast<uint64_t *>(start),
                       const_cast<uint64_t *>(end));
  hipError_t e = hipGetLastError();
  tlsKernel = pack ? "k_pack" : "k_unpack";
  if (e != hipSuccess) return failHip(e, "kernel launch");
  return REDGPU_OK;
}

int redgpu_records_pack_dev(int32_t device, const int32_t *result, const uint64_t *start,
                            const uint64_t *end, uint64_t n, int rw, int pw, void *records,
                            void *stream) {
  return recordsLaunch(device, true, records, n, rw, pw, result, start, end, stream);
}

int redgpu_records_unpack_dev(int32_t device, const void *records, uint64_t n, int rw, int pw,
                              int32_t *result, uint64_t *start, uint64_t *end, void *stream) {
  return recordsLaunch(device, false, records, n, rw, pw, result, start, end, stream);
}

const redgpu_dfa *redgpu_group_member(const redgpu_group *g, uint32_t i) {
  return g && i < g->m.size() ? g->m[i].dfa : nullptr;
}

int redgpu_group_plan(const redgpu_group *g, const uint64_t *offsets, uint64_t stride, uint64_t n,
                      uint64_t *cuts) {
  if (!g || !cuts) return fail(REDGPU_EAPI, "null argument");
  const uint64_t G = g->m.size();
  cuts[0] = 0;
  if (!offsets) {
    // equal line counts (sizes differ by at most one line)
    const uint64_t base = n / G, extra = n % G;
    for (uint64_t k = 1; k <= G; ++k) cuts[k] = k * base + (k < extra ? k : extra);
    return REDGPU_OK;
  }
  for (uint64_t i = 0; i < n; ++i)
    if (offsets[i] > offsets[i + 1]) return fail(REDGPU_EAPI, "offsets not monotonic");
  // equal BYTES: the first line that starts at or after the k-th share of the bytes
  const uint64_t lo = offsets[0], total = offsets[n] - lo;
  for (uint64_t k = 1; k < G; ++k) {
    const uint64_t target = lo + uint64_t((__uint128_t)total * k / G);
    uint64_t a = cuts[k - 1], b = n;
    while (a < b) {
      const uint64_t mid = (a + b) / 2;
      if (offsets[mid] >= target) b = mid; else a = mid + 1;
    }
    cuts[k] = a;
  }
  cuts[G] = n;
  return REDGPU_OK;
}

int redgpu_group_batch(const redgpu_group *g, int verb, int style, int do_leader,
                       const uint8_t *data, const uint64_t *offsets, uint64_t stride, uint64_t n,
                       int32_t *result, uint64_t *start, uint64_t *end) {
  if (!g) return fail(REDGPU_EAPI, "null group handle");
  if (verb < REDGPU_VERB_CHECK || verb > REDGPU_VERB_SEARCH) return fail(REDGPU_EAPI, "bad verb");
  if (n == 0) return REDGPU_OK;
  if (!result) return fail(REDGPU_EAPI, "null result buffer");
  const size_t G = g->m.size();
  std::vector<uint64_t> cuts(G + 1);
  if (int rc = redgpu_group_plan(g, offsets, stride, n, cuts.data())) return rc;
  std::vector<int> rcs(G, REDGPU_OK);
  std::vector<std::string> msgs(G);
  auto work = [&](size_t k) {
    const uint64_t lo = cuts[k], nl = cuts[k + 1] - lo;
    if (!nl) return;
    const redgpu_dfa *d = g->m[k].dfa;
    // ragged: the shard's slice of the offsets against the unshifted data pointer (the host
    // entry point copies [offsets[lo], offsets[hi]) and rebases)
    const uint8_t *p = offsets ? data : data + lo * stride;
    const uint64_t *o = offsets ? offsets + lo : nullptr;
    int rc;
    switch (verb) {
    case REDGPU_VERB_CHECK: rc = redgpu_check_batch(d, style, do_leader, p, o, stride, nl, result + lo); break;
    case REDGPU_VERB_SCAN: rc = redgpu_scan_batch(d, style, do_leader, p, o, stride, nl, result + lo); break;
    case REDGPU_VERB_SEARCH:
      rc = redgpu_search_batch(d, style, do_leader, p, o, stride, nl, result + lo,
                               start ? start + lo : nullptr, end ? end + lo : nullptr);
      break;
    default:
      rc = redgpu_match_batch(d, style, do_leader, p, o, stride, nl, result + lo,
                              start ? start + lo : nullptr, end ? end + lo : nullptr);
    }
    rcs[k] = rc;
    if (rc != REDGPU_OK) msgs[k] = redgpu_last_error();
  };
  // one host thread per device, as thr_red.cpp runs one per core; the caller's thread takes shard 0
  {
    std::lock_guard<std::mutex> lock(const_cast<redgpu_group *>(g)->hostMu);
    for (size_t k = 1; k < G; ++k) g->workers[k - 1]->post([&work, k] { work(k); });
    work(0);
    for (size_t k = 1; k < G; ++k) g->workers[k - 1]->wait();
  }
  for (size_t k = 0; k < G; ++k)
    if (rcs[k] != REDGPU_OK) return fail(rcs[k], "device shard " + std::to_string(k) + ": " + msgs[k]);
  return REDGPU_OK;
}

int redgpu_group_batch_dev(redgpu_group *g, int verb, int style, int do_leader,
                           const uint8_t *const *data, const uint64_t *const *offsets,
                           uint64_t stride, const uint64_t *n, int32_t *result, uint64_t *start,
                           uint64_t *end, int gather, void *const *shard_streams,
                           void *root_stream) {
  if (!g) return fail(REDGPU_EAPI, "null group handle");
  if (verb < REDGPU_VERB_CHECK || verb > REDGPU_VERB_SEARCH) return fail(REDGPU_EAPI, "bad verb");
  if (!data || !n) return fail(REDGPU_EAPI, "null shard arrays");
  if (!result) return fail(REDGPU_EAPI, "null result buffer");
  if (gather != REDGPU_GATHER_PEER && gather != REDGPU_GATHER_RCCL)
    return fail(REDGPU_EAPI, "bad gather mode");
  const bool positions = verb == REDGPU_VERB_MATCH || verb == REDGPU_VERB_SEARCH;
  if (!positions) { start = nullptr; end = nullptr; }
  const size_t G = g->m.size();
  std::lock_guard<std::mutex> lock(g->mu);
  const int rootDev = g->m[0].device;
  hipStream_t rootStream = static_cast<hipStream_t>(root_stream);

  if (gather == REDGPU_GATHER_RCCL && G > 1) {
    if (!g->distinct) return fail(REDGPU_EAPI, "the RCCL gather needs distinct devices");
    Rccl &r = rccl();
    if (!r.ok) return fail(REDGPU_ERCCL, "librccl.so could not be loaded");
    if (!g->commsUp) {
      std::vector<void *> comms(G, nullptr);
      std::vector<int> devs(G);
      for (size_t k = 0; k < G; ++k) devs[k] = g->m[k].device;
      const int rc = r.CommInitAll(comms.data(), int(G), devs.data());
      if (rc != 0) return failRccl(rc, "ncclCommInitAll");
      for (size_t k = 0; k < G; ++k) g->m[k].comm = comms[k];
      g->commsUp = true;
    }
  }

  // record geometry
  const int rw = widthFor(uint64_t(g->m[0].dfa->im->img.maxResult > 0 ? g->m[0].dfa->im->img.maxResult : 0));
  std::vector<uint64_t> lineBase(G + 1, 0), recBytes(G, 0), recBase(G + 1, 0);
  std::vector<int> pws(G, 8);
  for (size_t k = 0; k < G; ++k) lineBase[k + 1] = lineBase[k] + n[k];

  // 1. every device: scan its shard on its own stream
  for (size_t k = 0; k < G; ++k) {
    Member &mb = g->m[k];
    if (!n[k]) continue;
    DeviceScope scope(mb.device);
    if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
    HIP_TRY(grow(reinterpret_cast<void **>(&mb.res), &mb.resCap, n[k] * 4), "hipMalloc shard result");
    if (start) HIP_TRY(grow(reinterpret_cast<void **>(&mb.start), &mb.startCap, n[k] * 8), "hipMalloc shard start");
    if (end) HIP_TRY(grow(reinterpret_cast<void **>(&mb.end), &mb.endCap, n[k] * 8), "hipMalloc shard end");
    const uint64_t *off = offsets ? offsets[k] : nullptr;
    if (shard_streams) {
      // the shard's inputs are complete where its producer stream stands now: the scan waits there
      HIP_TRY(hipEventRecord(mb.ready, static_cast<hipStream_t>(shard_streams[k])), "hipEventRecord");
      HIP_TRY(hipStreamWaitEvent(mb.stream, mb.ready, 0), "hipStreamWaitEvent");
    }
    int rc;
    switch (verb) {
    case REDGPU_VERB_CHECK: rc = redgpu_check_batch_dev(mb.dfa, style, do_leader, data[k], off, stride, n[k], mb.res, mb.stream); break;
    case REDGPU_VERB_SCAN: rc = redgpu_scan_batch_dev(mb.dfa, style, do_leader, data[k], off, stride, n[k], mb.res, mb.stream); break;
    case REDGPU_VERB_SEARCH:
      rc = redgpu_search_batch_dev(mb.dfa, style, do_leader, data[k], off, stride, n[k], mb.res,
                                   start ? mb.start : nullptr, end ? mb.end : nullptr, mb.stream);
      break;
    default:
      rc = redgpu_match_batch_dev(mb.dfa, style, do_leader, data[k], off, stride, n[k], mb.res,
                                  start ? mb.start : nullptr, end ? mb.end : nullptr, mb.stream);
    }
    if (rc != REDGPU_OK) return rc;
  }
  // 2. record widths: positions never exceed the longest line.  Fixed stride: that is the stride.
  // Ragged lines: nothing on the host knows the offsets, and reading the longest line back would
  // put a host synchronisation into a call that promises none (round 2 did) - 4-byte positions,
  // which is what the block-wise kernels compute in anyway (a handle made with
  // REDGPU_F_FORCE_GENERIC, the only way to lines of 4 GiB and more, gets 8).
  for (size_t k = 0; k < G; ++k) {
    if (!n[k]) continue;
    Member &mb = g->m[k];
    uint64_t maxPos = stride;
    if (positions && offsets && offsets[k])
      maxPos = (mb.dfa->flags & REDGPU_F_FORCE_GENERIC) ? ~0ull : 0xffffffffull;
    pws[k] = widthFor(maxPos);
    recBytes[k] = planeBytes(n[k], rw) + (start ? planeBytes(n[k], pws[k]) : 0) +
                  (end ? planeBytes(n[k], pws[k]) : 0);
  }
  for (size_t k = 0; k < G; ++k) recBase[k + 1] = recBase[k] + recBytes[k];

  // 3. root buffer for every shard's records (the previous call's must have been widened)
  {
    DeviceScope scope(rootDev);
    if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
    if (g->rootRecCap < recBase[G]) {
      if (g->consumedValid) HIP_TRY(hipEventSynchronize(g->consumed), "hipEventSynchronize");
      HIP_TRY(grow(reinterpret_cast<void **>(&g->rootRec), &g->rootRecCap, recBase[G]), "hipMalloc root records");
    }
  }
  // 4. pack on every device, move to the root
  const bool useRccl = gather == REDGPU_GATHER_RCCL && G > 1;
  for (size_t k = 0; k < G; ++k) {
    if (!n[k]) continue;
    Member &mb = g->m[k];
    DeviceScope scope(mb.device);
    if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
    HIP_TRY(grow(reinterpret_cast<void **>(&mb.rec), &mb.recCap, recBytes[k]), "hipMalloc records");
    const uint32_t blocks = uint32_t((n[k] + 255) / 256 < 2048 ? (n[k] + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_pack, dim3(blocks), dim3(256), 0, mb.stream, mb.res,
                       start ? mb.start : nullptr, end ? mb.end : nullptr, n[k], rw, pws[k], mb.rec);
    if (g->consumedValid) HIP_TRY(hipStreamWaitEvent(mb.stream, g->consumed, 0), "hipStreamWaitEvent");
    if (!useRccl || k == 0) {
      if (mb.device == rootDev)
        HIP_TRY(hipMemcpyAsync(g->rootRec + recBase[k], mb.rec, recBytes[k], hipMemcpyDeviceToDevice, mb.stream), "copy records");
      else
        HIP_TRY(hipMemcpyPeerAsync(g->rootRec + recBase[k], rootDev, mb.rec, mb.device, recBytes[k], mb.stream), "peer copy records");
    }
  }
  if (useRccl) {
    Rccl &r = rccl();
    if (g->consumedValid && !n[0]) {  // an empty root shard skipped the wait above; its stream receives
      DeviceScope scope(rootDev);
      HIP_TRY(hipStreamWaitEvent(g->m[0].stream, g->consumed, 0), "hipStreamWaitEvent");
    }
    int rc = r.GroupStart();
    if (rc != 0) return failRccl(rc, "ncclGroupStart");
    for (size_t k = 1; k < G && rc == 0; ++k) {
      if (!n[k]) continue;
      rc = r.Send(g->m[k].rec, recBytes[k], kNcclUint8, 0, g->m[k].comm, g->m[k].stream);
      if (rc == 0)
        rc = r.Recv(g->rootRec + recBase[k], recBytes[k], kNcclUint8, int(k), g->m[0].comm, g->m[0].stream);
    }
    const int rc2 = r.GroupEnd();
    if (rc != 0) return failRccl(rc, "ncclSend / ncclRecv");
    if (rc2 != 0) return failRccl(rc2, "ncclGroupEnd");
  }
  for (size_t k = 0; k < G; ++k) {
    if (!n[k] && k != 0) continue;
    DeviceScope scope(g->m[k].device);
    HIP_TRY(hipEventRecord(g->m[k].done, g->m[k].stream), "hipEventRecord");
    // ... and the shard's own stream may not touch (or free) its inputs before the scan has read them
    if (shard_streams && n[k])
      HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(shard_streams[k]), g->m[k].done, 0),
              "hipStreamWaitEvent");
  }
  // 5. root: widen every shard's records into the caller's arrays, on the caller's stream
  {
    DeviceScope scope(rootDev);
    if (scope.err != hipSuccess) return failHip(scope.err, "hipSetDevice");
    for (size_t k = 0; k < G; ++k)
      if (n[k] || k == 0) HIP_TRY(hipStreamWaitEvent(rootStream, g->m[k].done, 0), "hipStreamWaitEvent");
    for (size_t k = 0; k < G; ++k) {
      if (!n[k]) continue;
      const uint32_t blocks = uint32_t((n[k] + 255) / 256 < 2048 ? (n[k] + 255) / 256 : 2048);
      hipLaunchKernelGGL(k_unpack, dim3(blocks), dim3(256), 0, rootStream, g->rootRec + recBase[k],
                         n[k], rw, pws[k], result + lineBase[k], start ? start + lineBase[k] : nullptr,
                         end ? end + lineBase[k] : nullptr);
    }
    HIP_TRY(hipGetLastError(), "kernel launch");
    HIP_TRY(hipEventRecord(g->consumed, rootStream), "hipEventRecord");
    g->consumedValid = true;
  }
  return REDGPU_OK;
}

}  // extern "C"
