// dfa_image.cpp - see dfa_image.h.  Host-only C++.
#include "dfa_image.h"

#include <algorithm>
#include <cstring>
#include <deque>

#include "../../include/redgpu.h"

namespace redgpu {

namespace {

// include/Serializer.h:42-59 - field offsets of the packed little-endian FileHeader
enum : size_t {
  kOffMajVer = 4, kOffMinVer = 6, kOffChecksum = 8, kOffFormat = 12, kOffMaxChar = 13,
  kOffLeaderLen = 14, kOffStateCnt = 16, kOffInitialOff = 20, kOffLeaderOff = 24,
  kOffEquivMap = 32,
};

inline uint16_t rd16(const uint8_t *p) { uint16_t v; std::memcpy(&v, p, 2); return v; }
inline uint32_t rd32(const uint8_t *p) { uint32_t v; std::memcpy(&v, p, 4); return v; }

inline uint32_t rdValue(const uint8_t *p, uint32_t vsz) {
  return vsz == 1 ? *p : vsz == 2 ? rd16(p) : rd32(p);
}

} // namespace

uint32_t fnv1a32(const void *ptr, size_t len) {
  const uint8_t *b = static_cast<const uint8_t *>(ptr);
  uint32_t h = 0x811c9dc5u;
  for (size_t i = 0; i < len; ++i) {
    h ^= b[i];
    h *= 0x01000193u;
  }
  return h;
}

uint32_t calcChecksum(const void *ptr, size_t len) {
  // everything from format_ to the end (lib/Serializer.cpp:301-306)
  return fnv1a32(static_cast<const uint8_t *>(ptr) + kOffFormat, len - kOffFormat);
}

const char *checkHeader(const void *ptr, size_t len) {
  const uint8_t *h = static_cast<const uint8_t *>(ptr);
  if (len < kHeaderBytes)
    return "Serialized DFA: header too short";
  if (h[0] != 'R' || h[1] != 'E' || h[2] != 'D' || h[3] != 'A')
    return "Serialized DFA: bad magic number";
  if (rd16(h + kOffMajVer) != 1 || rd16(h + kOffMinVer) != 0)
    return "Serialized DFA: unrecognized version";
  const uint32_t sum = calcChecksum(ptr, len);
  const uint32_t want = rd32(h + kOffChecksum);
  if (want != sum) {
    if (want == __builtin_bswap32(sum))
      return "serialized DFA: foreign endian-ness";
    return "serialized DFA: checksum mismatch";
  }
  const uint8_t fmt = h[kOffFormat];
  if (fmt != 1 && fmt != 2 && fmt != 4)
    return "Serialized DFA: unsupported format";
  return nullptr;
}

std::string buildImage(const void *reda, size_t len, uint32_t ldsTableMax, bool forceGlobal,
                       DfaImage &img, int &errCode) {
  errCode = REDGPU_EAPI;
  if (!reda || len == 0)
    return "serialized dfa is empty";
  if (const char *msg = checkHeader(reda, len))
    return msg;

  const uint8_t *h = static_cast<const uint8_t *>(reda);
  const uint32_t vsz = h[kOffFormat];
  const uint32_t nCls = uint32_t(h[kOffMaxChar]) + 1;
  const uint32_t leaderLen = h[kOffLeaderLen];
  const uint32_t stateCnt = rd32(h + kOffStateCnt);
  const uint32_t initialOff = rd32(h + kOffInitialOff);
  const uint32_t leaderOff = rd32(h + kOffLeaderOff);
  const size_t pad = (size_t(leaderLen) + 7u) & ~size_t(7); // lib/Executable.cpp:166
  const size_t rowBytes = size_t(nCls + 1) * vsz;            // include/Proxy.h:170-172
  const size_t baseOff = kHeaderBytes + pad;

  // The reference trusts the checksum; a GPU walk must not, so every offset is checked.
  if (stateCnt == 0)
    return "Serialized DFA: no states";
  if (baseOff > len || size_t(stateCnt) * rowBytes > len - baseOff)
    return "Serialized DFA: truncated state table";
  if (initialOff % rowBytes || initialOff / rowBytes >= stateCnt)
    return "Serialized DFA: initial offset out of range";
  if (leaderOff % rowBytes || leaderOff / rowBytes >= stateCnt)
    return "Serialized DFA: leader offset out of range";
  const uint8_t *base = h + baseOff;
  const uint32_t rowVals = nCls + 1;
  for (uint32_t c = 0; c < 256; ++c)
    if (h[kOffEquivMap + c] >= nCls)
      return "Serialized DFA: equivalence class out of range";
  for (uint32_t i = 0; i < leaderLen; ++i)
    if (h[kHeaderBytes + i] >= nCls)
      return "Serialized DFA: leader class out of range";

  img = DfaImage();
  img.format = vsz;
  img.nClasses = nCls;
  img.leaderLen = leaderLen;
  img.statesTotal = stateCnt;
  img.checksum = rd32(h + kOffChecksum);
  std::memcpy(img.equiv, h + kOffEquivMap, 256);
  std::memcpy(img.leader, h + kHeaderBytes, leaderLen);

  // raw automaton in the blob's own numbering (state id = row index)
  const uint32_t resultMask = vsz == 1 ? 0x7fu : vsz == 2 ? 0x7fffu : 0x7fffffffu;
  const uint32_t deadBit = 1u << (vsz * 8 - 1);
  auto rowOf = [&](uint32_t s) { return base + size_t(s) * rowBytes; };
  auto targetOf = [&](uint32_t s, uint32_t c, uint32_t &t) -> bool {
    // entry = target row's byte offset / sizeof(Value)  (include/Proxy.h:143-145,174-180)
    const uint32_t e = rdValue(rowOf(s) + size_t(1 + c) * vsz, vsz);
    if (e % rowVals)
      return false;
    t = e / rowVals;
    return t < stateCnt;
  };

  const uint32_t rawInit = uint32_t(initialOff / rowBytes);
  const uint32_t rawLead = uint32_t(leaderOff / rowBytes);

  // reachable set, breadth first from the initial state (and the post-leader state)
  std::vector<uint8_t> seen(stateCnt, 0);
  std::deque<uint32_t> todo;
  seen[rawInit] = 1;
  todo.push_back(rawInit);
  if (!seen[rawLead]) {
    seen[rawLead] = 1;
    todo.push_back(rawLead);
  }
  std::vector<uint32_t> reach;
  while (!todo.empty()) {
    const uint32_t s = todo.front();
    todo.pop_front();
    reach.push_back(s);
    for (uint32_t c = 0; c < nCls; ++c) {
      uint32_t t;
      if (!targetOf(s, c, t))
        return "Serialized DFA: transition out of range";
      if (!seen[t]) {
        seen[t] = 1;
        todo.push_back(t);
      }
    }
  }
  std::sort(reach.begin(), reach.end());

  // order: pure dead ends | other non-accepting | accepting   (stable in blob order)
  auto headOf = [&](uint32_t s) { return rdValue(rowOf(s), vsz); };
  auto klass = [&](uint32_t s) {
    const uint32_t hd = headOf(s);
    if (hd == deadBit) return 0;              // result 0 and dead-end flag: Proxy.h:139-141
    return (hd & resultMask) ? 2 : 1;
  };
  std::vector<uint32_t> order;
  order.reserve(reach.size());
  for (int k = 0; k < 3; ++k) {
    if (k == 1) img.nPureDead = uint32_t(order.size());
    if (k == 2) img.firstAccept = uint32_t(order.size());
    for (uint32_t s : reach)
      if (klass(s) == k)
        order.push_back(s);
  }
  img.nStates = uint32_t(order.size());
  std::vector<uint32_t> newId(stateCnt, 0xffffffffu);
  for (uint32_t i = 0; i < img.nStates; ++i)
    newId[order[i]] = i;
  img.init = newId[rawInit];
  img.leaderNext = newId[rawLead];

  img.result.resize(img.nStates);
  img.next.resize(size_t(img.nStates) * nCls);
  for (uint32_t i = 0; i < img.nStates; ++i) {
    const uint32_t s = order[i];
    const int32_t r = int32_t(headOf(s) & resultMask);
    img.result[i] = r;
    img.maxResult = std::max(img.maxResult, r);
    for (uint32_t c = 0; c < nCls; ++c) {
      uint32_t t = 0;
      targetOf(s, c, t);
      img.next[size_t(i) * nCls + c] = newId[t];
      if (i < img.nPureDead && t != s)
        img.deadAbsorbing = false;
    }
  }

  // table placement
  if (ldsTableMax == 0)
    ldsTableMax = 144u * 1024u;
  const uint64_t fused8 = uint64_t(img.nStates) * 256u;
  const uint64_t fused16 = uint64_t(img.nStates) * 512u;
  const uint64_t class16 = uint64_t(img.nStates) * nCls * 2u;
  if (!forceGlobal && img.nStates <= 256 && fused8 <= ldsTableMax)
    img.tableKind = REDGPU_TAB_LDS_FUSED_U8;
  else if (!forceGlobal && img.nStates <= 65536 && fused16 <= ldsTableMax)
    img.tableKind = REDGPU_TAB_LDS_FUSED_U16;
  else if (!forceGlobal && img.nStates <= 65536 && class16 <= ldsTableMax)
    img.tableKind = REDGPU_TAB_LDS_CLASS_U16;
  else if (img.nStates <= 65536)
    img.tableKind = REDGPU_TAB_GLOBAL_U16;
  else
    img.tableKind = REDGPU_TAB_GLOBAL_U32;

  auto put = [&](size_t idx, uint32_t v, uint32_t width) {
    if (width == 1) img.table[idx] = uint8_t(v);
    else if (width == 2) { uint16_t x = uint16_t(v); std::memcpy(&img.table[idx * 2], &x, 2); }
    else std::memcpy(&img.table[idx * 4], &v, 4);
  };
  switch (img.tableKind) {
  case REDGPU_TAB_LDS_FUSED_U8:
  case REDGPU_TAB_LDS_FUSED_U16: {
    // fused [state][byte]: the equivalence map is folded in, one lookup per input byte
    const uint32_t w = img.tableKind == REDGPU_TAB_LDS_FUSED_U8 ? 1 : 2;
    img.table.assign(size_t(img.nStates) * 256 * w, 0);
    for (uint32_t i = 0; i < img.nStates; ++i)
      for (uint32_t b = 0; b < 256; ++b)
        put(size_t(i) * 256 + b, img.next[size_t(i) * nCls + img.equiv[b]], w);
    break;
  }
  default: {
    const uint32_t w = img.tableKind == REDGPU_TAB_GLOBAL_U32 ? 4 : 2;
    img.table.assign(size_t(img.nStates) * nCls * w, 0);
    for (size_t k = 0; k < img.next.size(); ++k)
      put(k, img.next[k], w);
  }
  }
  errCode = REDGPU_OK;
  return std::string();
}

} // namespace redgpu
