// matcher_cabi_test.cpp - reads like the reference's test/matcher.cpp, but through
// include/redgpu.hpp (C++ mirror) -> include/redgpu.h (C-ABI) -> the gfx950 kernels.
// Usage: matcher_cabi_test <golden-dir>   (blobs compiled by the reference: tests/golden/dfas)
#include <cstdio>
#include <fstream>
#include <iterator>
#include <string>

#include "redgpu.hpp"

using namespace redgpu;

static int failures = 0;
#define EXPECT_EQ(a, b)                                                              \
  do {                                                                               \
    auto va = (a); auto vb = (b);                                                    \
    if (!(va == vb)) { ++failures; std::printf("FAIL %s:%d  %s != %s\n", __FILE__, __LINE__, #a, #b); } \
  } while (0)

static std::string slurp(const std::string &path) {
  std::ifstream f(path, std::ios::binary);
  return std::string(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
}

int main(int argc, char **argv) {
  const std::string dir = argc > 1 ? argv[1] : "tests/golden/dfas";

  // test/matcher.cpp:316-363 verifyLast ([0-9]+ -> 1, [0-9]+a -> 2, [0-9]+abcd -> 3)
  {
    Executable rex(slurp(dir + "/num3.reda"));
    Outcome oc = match(rex, "123abcd", styLast);
    EXPECT_EQ(check(rex, "123abcd", styLast), oc.result_);
    EXPECT_EQ(3, oc.result_);
    EXPECT_EQ(size_t(0), oc.start_);
    EXPECT_EQ(size_t(7), oc.end_);
    oc = match(rex, "123abcde", styFull);  // :366-413
    EXPECT_EQ(0, oc.result_);
    EXPECT_EQ(size_t(0), oc.end_);
    EXPECT_EQ(1, scan(rex, ".,_123", styInstant));  // :417-460
    Outcome so = search(rex, ".,_123abcde", styLast);  // :555-598 searchLast
    EXPECT_EQ(3, so.result_);
    EXPECT_EQ(size_t(3), so.start_);
    EXPECT_EQ(size_t(10), so.end_);
    EXPECT_EQ(2, (match<styTangent, true>(rex, "123abcd").result_));  // :266-313
  }
  // test/matcher.cpp:149-163 matchLast
  {
    Executable rex(gCopyTag, slurp(dir + "/newyork.reda"));
    Outcome oc = match(rex, "I love New York.", styLast);
    EXPECT_EQ(2, oc.result_);
    EXPECT_EQ(size_t(7), oc.start_);
    EXPECT_EQ(size_t(15), oc.end_);
    std::vector<std::string_view> lines = {"New", "nothing here", "I love New York.", ""};
    std::vector<Outcome> ocs = matchBatch(rex, lines, styLast);
    EXPECT_EQ(1, ocs[0].result_);
    EXPECT_EQ(0, ocs[1].result_);
    EXPECT_EQ(2, ocs[2].result_);
    EXPECT_EQ(size_t(15), ocs[2].end_);
    EXPECT_EQ(0, ocs[3].result_);
  }
  // errors: test/red.cpp:128-130 (non-REDA -> RedExceptApi), lib/Matcher.cpp:45 (bad style)
  {
    bool threw = false;
    try { Executable bad(std::string(1024, '\0')); } catch (const RedExceptApi &) { threw = true; }
    EXPECT_EQ(true, threw);
    Executable rex(slurp(dir + "/err.reda"));
    threw = false;
    try { check(rex, "error", static_cast<Style>(9)); } catch (const RedExceptExec &) { threw = true; }
    EXPECT_EQ(true, threw);
    Executable moved(std::move(rex));  // test/executable.cpp:55-63 move semantics
    EXPECT_EQ(1, check(moved, "error", styFull));
  }
  std::printf(failures ? "%d FAILURES\n" : "all C++ mirror checks passed\n", failures);
  return failures ? 1 : 0;
}
