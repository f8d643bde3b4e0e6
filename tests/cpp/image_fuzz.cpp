// image_fuzz.cpp - robustness of the REDA reader (one_amd/csrc/dfa_image.cpp) on corrupt blobs.
// Built with -fsanitize=address,undefined by tests/test_image_fuzz.py (CPU only): golden blobs
// are mutated (bytes, header fields, truncation), their checksum is FIXED UP so that validation
// gets past lib/Serializer.cpp:270-298's checks and into the offset walking, and buildImage must
// either refuse cleanly or produce an image whose every index is in range - the GPU kernels
// chase these indices without further checks.
// usage: image_fuzz <iterations> <blob>...
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../../include/redgpu.h"
#include "../../one_amd/csrc/dfa_image.h"

using namespace redgpu;

static uint64_t rngState = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
  uint64_t z = (rngState += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static void fixChecksum(std::vector<uint8_t> &b) {
  if (b.size() < kHeaderBytes) return;
  const uint32_t c = calcChecksum(b.data(), b.size());
  std::memcpy(&b[8], &c, 4);
}

static int checkImage(const DfaImage &img) {
  const uint32_t n = img.nStates;
  if (n == 0 || img.init >= n || img.leaderNext >= n) return 1;
  if (img.nPureDead > img.firstAccept || img.firstAccept > n) return 2;
  if (img.next.size() != size_t(n) * img.nClasses || img.result.size() != n) return 3;
  for (uint32_t v : img.next)
    if (v >= n) return 4;
  for (uint32_t i = 0; i < n; ++i)
    if ((img.result[i] > 0) != (i >= img.firstAccept)) return 5;
  for (uint32_t c = 0; c < 256; ++c)
    if (img.equiv[c] >= img.nClasses) return 6;
  for (uint32_t k = 0; k < img.leaderLen; ++k)
    if (img.leader[k] >= img.nClasses) return 7;
  if (img.tableKind == REDGPU_TAB_HOT_ROWS) {
    if (img.nHot == 0 || img.nHot > 254 || img.hotLo + img.nHot > n) return 8;
    if (img.hot8Off + 65536u > img.table.size()) return 9;
    const uint8_t *t8 = &img.table[img.hot8Off];
    for (uint32_t i = 0; i < 65536; ++i)
      if (t8[i] != 0xff && t8[i] >= img.nHot + img.hotShift) return 10;
  }
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  const long iters = std::atol(argv[1]);
  long accepted = 0, refused = 0;
  for (int a = 2; a < argc; ++a) {
    std::ifstream f(argv[a], std::ios::binary);
    const std::vector<uint8_t> good((std::istreambuf_iterator<char>(f)),
                                    std::istreambuf_iterator<char>());
    {
      DfaImage img;
      int code = 0;
      const std::string err = buildImage(good.data(), good.size(), 0, false, img, code);
      if (!err.empty() || checkImage(img)) {
        std::printf("FAIL pristine %s: %s / %d\n", argv[a], err.c_str(), checkImage(img));
        return 1;
      }
    }
    for (long it = 0; it < iters; ++it) {
      std::vector<uint8_t> b = good;
      const int kind = int(rnd() % 6);
      if (kind == 0) {  // a few random bytes anywhere
        for (int k = 0, m = 1 + int(rnd() % 4); k < m; ++k) b[rnd() % b.size()] = uint8_t(rnd());
      } else if (kind == 1) {  // a header field
        const size_t off[] = {12, 13, 14, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27};
        b[off[rnd() % 15]] = uint8_t(rnd());
      } else if (kind == 2) {  // truncate / extend
        const size_t n = rnd() % (b.size() + 64);
        b.resize(n, uint8_t(rnd()));
      } else if (kind == 3) {  // a row entry
        if (b.size() > kHeaderBytes + 8)
          b[kHeaderBytes + rnd() % (b.size() - kHeaderBytes)] = uint8_t(rnd());
      } else if (kind == 4) {  // equivalence map entry
        b[32 + rnd() % 256] = uint8_t(rnd());
      } else {  // a 32-bit field blown up
        const uint32_t big = uint32_t(rnd());
        std::memcpy(&b[16 + 4 * (rnd() % 3)], &big, 4);
      }
      if (rnd() % 8) fixChecksum(b);  // mostly get past the checksum
      DfaImage img;
      int code = 0;
      const uint32_t budget = (rnd() % 3 == 0) ? uint32_t(rnd() % 200000) : 0u;
      const std::string err = buildImage(b.data(), b.size(), budget, rnd() % 5 == 0, img, code,
                                         rnd() % 5 == 0);
      if (err.empty()) {
        const int bad = checkImage(img);
        if (bad) {
          std::printf("FAIL %s iteration %ld kind %d: accepted image breaks invariant %d\n",
                      argv[a], it, kind, bad);
          return 1;
        }
        ++accepted;
      } else {
        if (code != REDGPU_EAPI && code != REDGPU_ELIMIT) {
          std::printf("FAIL %s iteration %ld: refusal with code %d\n", argv[a], it, code);
          return 1;
        }
        ++refused;
      }
    }
  }
  std::printf("image fuzz ok: %ld accepted, %ld refused\n", accepted, refused);
  return 0;
}
