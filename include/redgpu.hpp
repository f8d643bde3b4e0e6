// redgpu.hpp - C++ mirror of RED's matcher interface over the C-ABI of redgpu.h.
//
// Same names, argument meaning and error behaviour as the reference for the hot path
// (citations relative to /root/reference/quol/red/):
//   Executable        include/Executable.h:28-76   (validated serialized DFA; move-only)
//   Style             include/Matcher.h:67-74
//   Result / Outcome  include/Types.h:22, include/Outcome.h:32-51
//   check/match/scan  include/Matcher.h:79-92 (run-time style, doLeader = true) and
//                     :133-169 (template <Style, bool doLeader>)
//   RedExcept*        include/Except.h:30-107
// plus the batch forms (checkBatch / matchBatch / scanBatch) that replace the callers'
// per-input loops (tools/bench.cpp:60-71, tools/thr_red.cpp:36-47).  Header-only; link with
// one_amd/libredgpu.so.  Single-input calls are batches of one ON THE GPU - there is no CPU
// matcher behind this header.
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

#include "redgpu.h"

namespace redgpu {

typedef uint8_t Byte;
typedef int32_t Result;

enum Style {
  styInvalid = 0,
  styInstant = 1,
  styFirst = 2,
  styTangent = 3,
  styLast = 4,
  styFull = 5,
};

struct Outcome {
  Result result_;
  size_t start_;
  size_t end_;
  bool operator==(const Outcome &rhs) const {
    return result_ == rhs.result_ && start_ == rhs.start_ && end_ == rhs.end_;
  }
  explicit operator bool() const { return result_ > 0; }
  static Outcome fail() { return Outcome{0, 0, 0}; }
};

class RedExcept : public std::runtime_error { using std::runtime_error::runtime_error; };
class RedExceptInternal : public RedExcept { using RedExcept::RedExcept; };
class RedExceptUser : public RedExcept { using RedExcept::RedExcept; };
class RedExceptLimit : public RedExcept { using RedExcept::RedExcept; };
class RedExceptExec : public RedExceptInternal { using RedExceptInternal::RedExceptInternal; };
class RedExceptApi : public RedExceptUser { using RedExceptUser::RedExceptUser; };
class RedExceptHip : public RedExceptExec { using RedExceptExec::RedExceptExec; };  // new

inline void throwOnError(int rc) {
  if (rc == REDGPU_OK) return;
  const std::string msg = redgpu_last_error();
  switch (rc) {
  case REDGPU_EAPI: throw RedExceptApi(msg);
  case REDGPU_EEXEC: throw RedExceptExec(msg);
  case REDGPU_ELIMIT: throw RedExceptLimit(msg);
  case REDGPU_EHIP: throw RedExceptHip(msg);
  default: throw RedExcept(msg);
  }
}

// tag types kept for source compatibility (include/Types.h:43-53); every constructor COPIES
struct CopyTag {};
constexpr CopyTag gCopyTag;
struct UnownedTag {};
constexpr UnownedTag gUnownedTag;

class Executable {
public:
  Executable() : h_(nullptr) {}
  Executable(Executable &&other) : h_(std::exchange(other.h_, nullptr)) {}
  explicit Executable(std::string &&buf, int device = REDGPU_DEVICE_CURRENT) : h_(nullptr) {
    if (buf.empty()) throw RedExceptApi("serialized dfa move-string is empty");
    init(buf.data(), buf.size(), device);
  }
  Executable(const CopyTag &, std::string_view sv, int device = REDGPU_DEVICE_CURRENT)
      : h_(nullptr) {
    if (sv.empty()) throw RedExceptApi("serialized dfa string_view is empty");
    init(sv.data(), sv.size(), device);
  }
  Executable(const UnownedTag &, std::string_view sv, int device = REDGPU_DEVICE_CURRENT)
      : h_(nullptr) {
    if (!sv.data()) throw RedExceptApi("serialized dfa unowned-view is empty");
    init(sv.data(), sv.size(), device);
  }
  ~Executable() { redgpu_dfa_destroy(h_); }
  Executable &operator=(Executable &&rhs) {
    if (this != &rhs) {
      redgpu_dfa_destroy(h_);
      h_ = std::exchange(rhs.h_, nullptr);
    }
    return *this;
  }
  Executable(const Executable &) = delete;
  Executable &operator=(const Executable &) = delete;

  std::string_view serialized() const {
    const void *p = nullptr;
    size_t n = 0;
    throwOnError(redgpu_dfa_serialized(h_, &p, &n));
    return std::string_view(static_cast<const char *>(p), n);
  }
  redgpu_info info() const {
    redgpu_info i;
    throwOnError(redgpu_dfa_info(h_, &i));
    return i;
  }
  const redgpu_dfa *handle() const { return h_; }

private:
  void init(const void *p, size_t n, int device) {
    redgpu_opts o{};
    o.device = device;
    throwOnError(redgpu_dfa_create(p, n, &o, &h_));
  }
  redgpu_dfa *h_;
};

// ---- batch forms: line i = data[offsets[i], offsets[i+1]) (offsets has n+1 entries), or with
// offsets == nullptr, data[i*stride, (i+1)*stride).  Host buffers. -----------------------------
template <Style style, bool doLeader>
void checkBatch(const Executable &exec, const Byte *data, const uint64_t *offsets, uint64_t stride,
                uint64_t n, Result *result) {
  throwOnError(redgpu_check_batch(exec.handle(), style, doLeader, data, offsets, stride, n, result));
}

template <Style style, bool doLeader>
void matchBatch(const Executable &exec, const Byte *data, const uint64_t *offsets, uint64_t stride,
                uint64_t n, Result *result, uint64_t *start, uint64_t *end) {
  throwOnError(redgpu_match_batch(exec.handle(), style, doLeader, data, offsets, stride, n, result,
                                  start, end));
}

template <Style style, bool doLeader>
void scanBatch(const Executable &exec, const Byte *data, const uint64_t *offsets, uint64_t stride,
               uint64_t n, Result *result) {
  throwOnError(redgpu_scan_batch(exec.handle(), style, doLeader, data, offsets, stride, n, result));
}

inline std::vector<Outcome> matchBatch(const Executable &exec, const std::vector<std::string_view> &lines,
                                       Style style, bool doLeader = true) {
  std::string flat;
  std::vector<uint64_t> off(lines.size() + 1, 0);
  for (size_t i = 0; i < lines.size(); ++i) {
    flat.append(lines[i]);
    off[i + 1] = flat.size();
  }
  std::vector<Result> r(lines.size());
  std::vector<uint64_t> s(lines.size()), e(lines.size());
  throwOnError(redgpu_match_batch(exec.handle(), style, doLeader,
                                  reinterpret_cast<const Byte *>(flat.data()), off.data(), 0,
                                  lines.size(), r.data(), s.data(), e.data()));
  std::vector<Outcome> out(lines.size());
  for (size_t i = 0; i < lines.size(); ++i) out[i] = Outcome{r[i], size_t(s[i]), size_t(e[i])};
  return out;
}

// The lines of a text blob (lib/Util.cpp:109-130's rule: [start, delimiter), bytes behind the last
// delimiter are not a line), each matched - the loop of tools/skim_red.cpp:36-46 in one call.
// lineStarts (optional) receives where each line begins in `text`; Outcome positions are relative
// to that, as if the line had been handed to match() on its own.
inline std::vector<Outcome> matchText(const Executable &exec, std::string_view text, Style style,
                                      bool doLeader = true, char delim = '\n',
                                      std::vector<size_t> *lineStarts = nullptr) {
  const Byte *p = reinterpret_cast<const Byte *>(text.data());
  uint64_t n = 0, none = 0;
  throwOnError(redgpu_split_lines(exec.handle(), p, text.size(), Byte(delim), &none, 0, &n));
  std::vector<uint64_t> off(n + 1, 0), s(n), e(n);
  std::vector<Result> r(n);
  throwOnError(redgpu_match_text(exec.handle(), style, doLeader, p, text.size(), Byte(delim),
                                 off.data(), n, &n, r.data(), s.data(), e.data()));
  std::vector<Outcome> out(r.size());
  for (size_t i = 0; i < out.size(); ++i) out[i] = Outcome{r[i], size_t(s[i]), size_t(e[i])};
  if (lineStarts) lineStarts->assign(off.begin(), off.begin() + out.size());
  return out;
}

// ---- several GPUs of one node: one image per device, contiguous shards, results in the caller's
// arrays - the device form of tools/thr_red.cpp:84-91 (N workers over one shared Red).  devices
// may name a device more than once (the shards then share it). ----------------------------------
class Group {
public:
  Group(std::string_view serialized, const std::vector<int32_t> &devices) : g_(nullptr) {
    if (serialized.empty()) throw RedExceptApi("serialized dfa string_view is empty");
    throwOnError(redgpu_group_create(serialized.data(), serialized.size(), nullptr, devices.data(),
                                     uint32_t(devices.size()), &g_));
  }
  ~Group() { redgpu_group_destroy(g_); }
  Group(const Group &) = delete;
  Group &operator=(const Group &) = delete;
  uint32_t size() const { return redgpu_group_size(g_); }
  const redgpu_dfa *member(uint32_t i) const { return redgpu_group_member(g_, i); }
  // shard g = lines [cuts[g], cuts[g + 1]): equal lines, or equal bytes when offsets are given
  std::vector<uint64_t> plan(const uint64_t *offsets, uint64_t stride, uint64_t n) const {
    std::vector<uint64_t> cuts(size() + 1);
    throwOnError(redgpu_group_plan(g_, offsets, stride, n, cuts.data()));
    return cuts;
  }
  template <Style style, bool doLeader>
  void matchBatch(const Byte *data, const uint64_t *offsets, uint64_t stride, uint64_t n,
                  Result *result, uint64_t *start, uint64_t *end) const {
    throwOnError(redgpu_group_batch(g_, REDGPU_VERB_MATCH, style, doLeader, data, offsets, stride, n,
                                    result, start, end));
  }
  template <Style style, bool doLeader>
  void checkBatch(const Byte *data, const uint64_t *offsets, uint64_t stride, uint64_t n,
                  Result *result) const {
    throwOnError(redgpu_group_batch(g_, REDGPU_VERB_CHECK, style, doLeader, data, offsets, stride, n,
                                    result, nullptr, nullptr));
  }
  redgpu_group *handle() const { return g_; }

private:
  redgpu_group *g_;
};

// ---- single-input forms with the reference's signatures (a batch of one on the GPU) ---------
namespace detail {
inline Result one(int (*fn)(const redgpu_dfa *, int, int, const uint8_t *, const uint64_t *,
                            uint64_t, uint64_t, int32_t *),
                  const Executable &exec, const void *ptr, size_t len, int style, bool lead) {
  const uint64_t off[2] = {0, len};
  Result r = 0;
  throwOnError(fn(exec.handle(), style, lead, static_cast<const Byte *>(ptr), off, 0, 1, &r));
  return r;
}
} // namespace detail

template <Style style, bool doLeader>
Result check(const Executable &exec, const void *ptr, size_t len) {
  return detail::one(redgpu_check_batch, exec, ptr, len, style, doLeader);
}
template <Style style, bool doLeader>
Result check(const Executable &exec, std::string_view sv) {
  return check<style, doLeader>(exec, sv.data(), sv.size());
}
template <Style style, bool doLeader>
Result scan(const Executable &exec, const void *ptr, size_t len) {
  return detail::one(redgpu_scan_batch, exec, ptr, len, style, doLeader);
}
template <Style style, bool doLeader>
Result scan(const Executable &exec, std::string_view sv) {
  return scan<style, doLeader>(exec, sv.data(), sv.size());
}
template <Style style, bool doLeader>
Outcome match(const Executable &exec, const void *ptr, size_t len) {
  const uint64_t off[2] = {0, len};
  Result r = 0;
  uint64_t s = 0, e = 0;
  throwOnError(redgpu_match_batch(exec.handle(), style, doLeader, static_cast<const Byte *>(ptr),
                                  off, 0, 1, &r, &s, &e));
  return Outcome{r, size_t(s), size_t(e)};
}
template <Style style, bool doLeader>
Outcome match(const Executable &exec, std::string_view sv) {
  return match<style, doLeader>(exec, sv.data(), sv.size());
}

// search<style,doLeader>: include/Matcher.h:172-182,557-640 (sliding-window match)
template <Style style, bool doLeader>
Outcome search(const Executable &exec, std::string_view sv) {
  const uint64_t off[2] = {0, sv.size()};
  Result r = 0;
  uint64_t s = 0, e = 0;
  throwOnError(redgpu_search_batch(exec.handle(), style, doLeader,
                                   reinterpret_cast<const Byte *>(sv.data()), off, 0, 1, &r, &s, &e));
  return Outcome{r, size_t(s), size_t(e)};
}
inline Outcome search(const Executable &exec, std::string_view sv, Style style) {
  const uint64_t off[2] = {0, sv.size()};
  Result r = 0;
  uint64_t s = 0, e = 0;
  throwOnError(redgpu_search_batch(exec.handle(), style, 1,
                                   reinterpret_cast<const Byte *>(sv.data()), off, 0, 1, &r, &s, &e));
  return Outcome{r, size_t(s), size_t(e)};
}
template <Style style, bool doLeader>
void searchBatch(const Executable &exec, const Byte *data, const uint64_t *offsets, uint64_t stride,
                 uint64_t n, Result *result, uint64_t *start, uint64_t *end) {
  throwOnError(redgpu_search_batch(exec.handle(), style, doLeader, data, offsets, stride, n, result,
                                   start, end));
}

namespace detail {
// the two list verbs: first try with room for 16 records, retry once with the exact count
template <class F>
size_t listOne(F call, const Executable &exec, std::string_view sv, std::vector<Outcome> &out) {
  const uint64_t off[2] = {0, sv.size()};
  uint64_t cap = 16, found = 0;
  std::vector<Result> r;
  std::vector<uint64_t> s, e;
  for (int pass = 0; pass < 2; ++pass) {
    r.assign(cap, 0);
    s.assign(cap, 0);
    e.assign(cap, 0);
    throwOnError(call(exec.handle(), reinterpret_cast<const Byte *>(sv.data()), off, cap, &found,
                      r.data(), s.data(), e.data()));
    if (found <= cap) break;
    cap = found;
  }
  out.clear();
  for (uint64_t i = 0; i < found; ++i) out.push_back(Outcome{r[i], size_t(s[i]), size_t(e[i])});
  return out.size();
}
} // namespace detail

// matchAll(exec, sv, out): include/Matcher.h:127, lib/Matcher.cpp:97-102 (doLeader = true)
inline size_t matchAll(const Executable &exec, std::string_view sv, std::vector<Outcome> &out) {
  return detail::listOne(
      [](const redgpu_dfa *h, const Byte *p, const uint64_t *off, uint64_t cap, uint64_t *cnt,
         Result *r, uint64_t *s, uint64_t *e) {
        return redgpu_match_all_batch(h, 1, p, off, 0, 1, cap, cnt, r, s, e);
      },
      exec, sv, out);
}

// Red::collect(text, out): include/Red.h:115, lib/Red.cpp:103-116
inline size_t collect(const Executable &exec, std::string_view sv, std::vector<Outcome> &out) {
  return detail::listOne(
      [](const redgpu_dfa *h, const Byte *p, const uint64_t *off, uint64_t cap, uint64_t *cnt,
         Result *r, uint64_t *s, uint64_t *e) {
        return redgpu_collect_batch(h, p, off, 0, 1, cap, cnt, r, s, e);
      },
      exec, sv, out);
}

// replace(exec, sv, repl, out, max, style): include/Matcher.h:119-124, lib/Matcher.cpp:72-92
// (run-time style, doLeader = true); the templates of Matcher.h:186-211 below it
namespace detail {
inline size_t replaceOne(const Executable &exec, std::string_view sv, std::string_view repl,
                         std::string &out, size_t max, int style, int doLeader) {
  const uint64_t off[2] = {0, sv.size()};
  uint64_t cnt = 0, ooff[2] = {0, 0};
  out.assign(sv.size() + 64, '\0');
  for (int pass = 0; pass < 2; ++pass) {
    throwOnError(redgpu_replace_batch(exec.handle(), style, doLeader,
                                      reinterpret_cast<const Byte *>(sv.data()), off, 0, 1,
                                      reinterpret_cast<const Byte *>(repl.data()), repl.size(), max,
                                      &cnt, ooff, reinterpret_cast<Byte *>(out.data()),
                                      out.size()));
    if (ooff[1] <= out.size()) break;
    out.assign(size_t(ooff[1]), '\0');
  }
  out.resize(size_t(ooff[1]));
  return size_t(cnt);
}
} // namespace detail

inline size_t replace(const Executable &exec, std::string_view sv, std::string_view repl,
                      std::string &out, size_t max, Style style) {
  return detail::replaceOne(exec, sv, repl, out, max, style, 1);
}
template <Style style, bool doLeader>
size_t replace(const Executable &exec, std::string_view sv, std::string_view repl, std::string &out,
               size_t max) {
  return detail::replaceOne(exec, sv, repl, out, max, style, doLeader);
}

// StatefulMatcher: include/Matcher.h:770-792.  advance(Byte) as in the reference, plus
// advance(ptr, len) for a whole chunk per launch.  exec must outlive this object.
class StatefulMatcher {
public:
  explicit StatefulMatcher(const Executable &exec) : exec_(exec) { advance(nullptr, 0); }

  Result advance(Byte input) { return advance(&input, 1); }
  Result advance(const Byte *p, size_t len) {
    const uint64_t off[2] = {0, len};
    const Byte dummy = 0;
    throwOnError(redgpu_advance_batch(exec_.handle(), p ? p : &dummy, off, 0, 1, &state_,
                                      &result_));
    return result_;
  }
  Result result() const { return result_; }

private:
  const Executable &exec_;
  uint32_t state_ = REDGPU_STATE_INITIAL;
  Result result_ = 0;
};

// run-time style: doLeader = true, unknown style -> RedExceptExec("unsupported style")
// (lib/Matcher.cpp:37-67)
inline Result check(const Executable &exec, std::string_view sv, Style style) {
  return detail::one(redgpu_check_batch, exec, sv.data(), sv.size(), style, true);
}
inline Result scan(const Executable &exec, std::string_view sv, Style style) {
  return detail::one(redgpu_scan_batch, exec, sv.data(), sv.size(), style, true);
}
inline Outcome match(const Executable &exec, std::string_view sv, Style style) {
  const uint64_t off[2] = {0, sv.size()};
  Result r = 0;
  uint64_t s = 0, e = 0;
  throwOnError(redgpu_match_batch(exec.handle(), style, 1,
                                  reinterpret_cast<const Byte *>(sv.data()), off, 0, 1, &r, &s, &e));
  return Outcome{r, size_t(s), size_t(e)};
}

} // namespace redgpu
