// matcher_cabi_test.cpp - reads like the reference's test/matcher.cpp, but through
// include/redgpu.hpp (C++ mirror) -> include/redgpu.h (C-ABI) -> the gfx950 kernels.
// Usage: matcher_cabi_test <golden-dir>   (blobs compiled by the reference: tests/golden/dfas)
#include <cstdio>
#include <fstream>
#include <iterator>
#include <string>
#include <thread>
#include <vector>

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
    // the same lines as one text blob (tools/skim_red.cpp:36-46; the tail without a delimiter
    // is not a line: lib/Util.cpp:109-130)
    std::vector<size_t> starts;
    std::vector<Outcome> tx = matchText(rex, "New\nnothing here\nI love New York.\n\nNew York", styLast,
                                        true, '\n', &starts);
    EXPECT_EQ(size_t(4), tx.size());
    if (tx.size() == 4 && tx[0].result_ != 1) {  // what came back, for the log
      std::fprintf(stderr, "matchText: results");
      for (const Outcome &o : tx) std::fprintf(stderr, " (%d %zu %zu)", o.result_, o.start_, o.end_);
      std::fprintf(stderr, " starts");
      for (size_t v : starts) std::fprintf(stderr, " %zu", v);
      std::fprintf(stderr, " kernel %s\n", redgpu_last_kernel());
    }
    EXPECT_EQ(1, tx[0].result_);
    EXPECT_EQ(0, tx[1].result_);
    EXPECT_EQ(2, tx[2].result_);
    EXPECT_EQ(size_t(7), tx[2].start_);
    EXPECT_EQ(size_t(15), tx[2].end_);
    EXPECT_EQ(0, tx[3].result_);
    EXPECT_EQ(size_t(17), starts[2]);
    EXPECT_EQ(size_t(0), matchText(rex, "no delimiter at all", styLast).size());
  }
  // test/matcher.cpp:695-723 matchAll
  {
    Executable rex(slurp(dir + "/set5.reda"));
    std::vector<Outcome> vec;
    EXPECT_EQ(size_t(4), matchAll(rex, "0123456789", vec));
    EXPECT_EQ(size_t(4), vec.size());
    EXPECT_EQ(1, vec[0].result_);
    EXPECT_EQ(size_t(0), vec[0].start_);
    EXPECT_EQ(size_t(1), vec[0].end_);
    EXPECT_EQ(3, vec[1].result_);
    EXPECT_EQ(size_t(3), vec[1].end_);
    EXPECT_EQ(2, vec[2].result_);
    EXPECT_EQ(size_t(4), vec[2].end_);
    EXPECT_EQ(5, vec[3].result_);
    EXPECT_EQ(size_t(0), vec[3].start_);
    EXPECT_EQ(size_t(6), vec[3].end_);
  }
  // test/matcher.cpp:725-745 matchAllLoose
  {
    Executable rex(slurp(dir + "/loose2.reda"));
    std::vector<Outcome> vec;
    EXPECT_EQ(size_t(6), matchAll(rex, ".aa..b.bb..abba.", vec));
    const int expR[6] = {1, 2, 2, 1, 2, 1};
    const size_t expE[6] = {3, 6, 9, 12, 14, 15};
    for (int i = 0; i < 6 && i < int(vec.size()); ++i) {
      EXPECT_EQ(expR[i], vec[i].result_);
      EXPECT_EQ(expE[i], vec[i].end_);
    }
    // more records than the first pass has room for
    std::string many;
    for (int i = 0; i < 40; ++i) many += "ab";
    EXPECT_EQ(size_t(80), matchAll(rex, many, vec));
    EXPECT_EQ(size_t(80), vec.back().end_);
  }
  // test/matcher.cpp:800-818 charByChar
  {
    Executable rex(slurp(dir + "/ale.reda"));
    StatefulMatcher sm(rex);
    EXPECT_EQ(0, sm.result());
    EXPECT_EQ(0, sm.advance('a'));
    EXPECT_EQ(0, sm.advance('l'));
    EXPECT_EQ(1, sm.advance('e'));
    EXPECT_EQ(1, sm.advance('e'));
    EXPECT_EQ(2, sm.advance('x'));
    EXPECT_EQ(2, sm.result());
  }
  // test/matcher.cpp:667-691 replaceStyles
  {
    Executable rex(slurp(dir + "/num3defg.reda"));
    std::string s;
    EXPECT_EQ(size_t(1), replace(rex, "#123defg!", "xyz", s, 1, styInstant));
    EXPECT_EQ(std::string("#xyz23defg!"), s);
    EXPECT_EQ(size_t(3), replace(rex, "#123defg!", "xyz", s, 9999, styInstant));
    EXPECT_EQ(std::string("#xyzxyzxyzdefg!"), s);
    EXPECT_EQ(size_t(1), replace(rex, "#123defg!", "xyz", s, 9999, styFirst));
    EXPECT_EQ(std::string("#xyzdefg!"), s);
    EXPECT_EQ(size_t(1), replace(rex, "#123defg!", "xyz", s, 9999, styTangent));
    EXPECT_EQ(std::string("#xyzefg!"), s);
    EXPECT_EQ(size_t(1), replace(rex, "#123defg!", "xyz", s, 9999, styLast));
    EXPECT_EQ(std::string("#xyz!"), s);
    EXPECT_EQ(size_t(0), replace(rex, "#123defg!", "xyz", s, 9999, styFull));
    EXPECT_EQ(std::string("#123defg!"), s);
    EXPECT_EQ(size_t(1), replace(rex, "#123defg", "xyz", s, 9999, styFull));
    EXPECT_EQ(std::string("#xyz"), s);
    std::string longRepl(300, 'Z');   // result longer than the first pass's buffer
    EXPECT_EQ(size_t(3), (replace<styInstant, true>(rex, "#123defg!", longRepl, s, 9999)));
    EXPECT_EQ(size_t(1 + 900 + 5), s.size());
  }
  // test/red.cpp:190-221 collect
  {
    Executable rex(slurp(dir + "/newyork4.reda"));
    std::vector<Outcome> out;
    EXPECT_EQ(size_t(5), collect(rex, "in new york12345, a new 6789 york city", out));
    const int expR[5] = {1, 4, 2, 4, 3};
    const size_t expS[5] = {3, 11, 20, 24, 29}, expE[5] = {11, 16, 23, 28, 33};
    for (int i = 0; i < 5 && i < int(out.size()); ++i) {
      EXPECT_EQ(expR[i], out[i].result_);
      EXPECT_EQ(expS[i], out[i].start_);
      EXPECT_EQ(expE[i], out[i].end_);
    }
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
  // tools/thr_red.cpp:84-91: N std::threads over ONE shared read-only matcher - here over one
  // redgpu_dfa (host-buffer batches, staged per thread), then the same lines through a Group of
  // three shards on device 0; every worker's answers equal the single-threaded ones
  {
    Executable rex(slurp(dir + "/uri.reda"));
    const std::string url = "see https://ab-c.example.com:8080/p/x.y?q=1#frag ok ";
    std::string flat;
    std::vector<uint64_t> off{0};
    for (int i = 0; i < 6000; ++i) {
      flat += (i % 3 == 0) ? url : std::string(size_t(5 + i % 40), char('a' + i % 7));
      off.push_back(flat.size());
    }
    const uint64_t n = off.size() - 1;
    std::vector<Result> r0(n);
    std::vector<uint64_t> s0(n), e0(n);
    matchBatch<styLast, false>(rex, reinterpret_cast<const Byte *>(flat.data()), off.data(), 0, n,
                               r0.data(), s0.data(), e0.data());
    int hits = 0;
    for (uint64_t i = 0; i < n; ++i) hits += r0[i] > 0;
    EXPECT_EQ(2000, hits);
    std::vector<int> bad(8, 0);
    std::vector<std::thread> workers;
    for (int t = 0; t < 8; ++t)
      workers.emplace_back([&, t] {
        for (int rep = 0; rep < 5; ++rep) {
          std::vector<Result> r(n);
          std::vector<uint64_t> s(n), e(n);
          matchBatch<styLast, false>(rex, reinterpret_cast<const Byte *>(flat.data()), off.data(), 0,
                                     n, r.data(), s.data(), e.data());
          if (r != r0 || s != s0 || e != e0) ++bad[t];
        }
        redgpu_thread_release();
      });
    for (auto &w : workers) w.join();
    for (int t = 0; t < 8; ++t) EXPECT_EQ(0, bad[t]);
    Group grp(rex.serialized(), {0, 0, 0});
    EXPECT_EQ(uint32_t(3), grp.size());
    const std::vector<uint64_t> cuts = grp.plan(off.data(), 0, n);
    EXPECT_EQ(uint64_t(0), cuts[0]);
    EXPECT_EQ(n, cuts[3]);
    std::vector<Result> rg(n);
    std::vector<uint64_t> sg(n), eg(n);
    grp.matchBatch<styLast, false>(reinterpret_cast<const Byte *>(flat.data()), off.data(), 0, n,
                                   rg.data(), sg.data(), eg.data());
    EXPECT_EQ(true, rg == r0 && sg == s0 && eg == e0);
  }
  std::printf(failures ? "%d FAILURES\n" : "all C++ mirror checks passed\n", failures);
  return failures ? 1 : 0;
}
