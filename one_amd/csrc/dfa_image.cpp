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
                       DfaImage &img, int &errCode, bool forceHot,
                       const std::vector<double> *measured) {
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
  std::vector<uint32_t> bfsRank(stateCnt, 0xffffffffu);
  for (uint32_t i = 0; i < reach.size(); ++i)
    bfsRank[reach[i]] = i;
  std::sort(reach.begin(), reach.end());

  auto headOf = [&](uint32_t s) { return rdValue(rowOf(s), vsz); };
  auto klass = [&](uint32_t s) {
    const uint32_t hd = headOf(s);
    if (hd == deadBit) return 0;              // result 0 and dead-end flag: Proxy.h:139-141
    return (hd & resultMask) ? 2 : 1;
  };

  // start bytes (see dfa_image.h)
  {
    auto pack = [&](auto pred, uint32_t &word, uint32_t &count) {
      uint32_t n = 0, w = 0;
      for (uint32_t b = 0; b < 256; ++b)
        if (pred(b)) {
          if (n < 4) w |= b << (8 * n);
          ++n;
        }
      // unused slots repeat the first member, so a 4-way test needs no count
      for (uint32_t k = n; k < 4 && n > 0; ++k) w |= (w & 0xffu) << (8 * k);
      word = w;
      count = n <= 4 ? n : 0xffu;
    };
    if (leaderLen > 0)
      pack([&](uint32_t b) { return h[kOffEquivMap + b] == h[kHeaderBytes]; }, img.startLeadWord,
           img.startLeadCount);
    pack([&](uint32_t b) {
      uint32_t t = 0;
      targetOf(rawInit, h[kOffEquivMap + b], t);
      return klass(t) != 0;
    }, img.startFreeWord, img.startFreeCount);
    if (leaderLen > 1)
      pack([&](uint32_t b) { return h[kOffEquivMap + b] == h[kHeaderBytes + 1]; },
           img.start2LeadWord, img.start2LeadCount);
    // second step without the leader: only meaningful when no first step accepts
    std::vector<uint32_t> firsts;  // non-dead targets of the initial state
    bool firstAccepts = false;
    for (uint32_t c = 0; c < nCls; ++c) {
      uint32_t t = 0;
      targetOf(rawInit, c, t);
      if (klass(t) == 0) continue;
      if (klass(t) == 2) firstAccepts = true;
      firsts.push_back(t);
    }
    if (!firstAccepts)
      pack([&](uint32_t b) {
        for (uint32_t f1 : firsts) {
          uint32_t t = 0;
          targetOf(f1, h[kOffEquivMap + b], t);
          if (klass(t) != 0) return true;
        }
        return false;
      }, img.start2FreeWord, img.start2FreeCount);
  }

  // ... and the full flag tables
  {
    std::vector<uint32_t> firsts;
    bool firstAccepts = false;
    for (uint32_t c = 0; c < nCls; ++c) {
      uint32_t t = 0;
      targetOf(rawInit, c, t);
      if (klass(t) == 0) continue;
      if (klass(t) == 2) firstAccepts = true;
      firsts.push_back(t);
    }
    img.startTotal[0] = img.startTotal[1] = 0;
    img.startFollow[0] = !firstAccepts;
    img.startFollow[1] = leaderLen > 1;
    for (uint32_t b = 0; b < 256; ++b) {
      const uint8_t cls = h[kOffEquivMap + b];
      uint32_t t = 0;
      targetOf(rawInit, cls, t);
      uint8_t f0 = klass(t) != 0 ? 1 : 0;
      if (img.startFollow[0])
        for (uint32_t f1 : firsts) {
          uint32_t t2 = 0;
          targetOf(f1, cls, t2);
          if (klass(t2) != 0) { f0 |= 2; break; }
        }
      img.startFlags[0][b] = f0;
      img.startTotal[0] += f0 & 1u;
      uint8_t fl = 0;
      if (leaderLen > 0 && cls == h[kHeaderBytes]) fl |= 1;
      if (leaderLen > 1 && cls == h[kHeaderBytes + 1]) fl |= 2;
      img.startFlags[1][b] = fl;
      img.startTotal[1] += fl & 1u;
    }
  }

  // table placement (decided before the renumbering: the hot-row kind orders states its own way)
  if (ldsTableMax == 0)
    ldsTableMax = 144u * 1024u;
  {
    const uint64_t nR = reach.size();
    const uint64_t fused8 = nR * 256u, fused16 = nR * 512u, class16 = nR * nCls * 2u;
    if (!forceGlobal && nR <= 256 && fused8 <= ldsTableMax)
      img.tableKind = REDGPU_TAB_LDS_FUSED_U8;
    else if (!forceGlobal && nR <= 65536 && fused16 <= ldsTableMax)
      img.tableKind = REDGPU_TAB_LDS_FUSED_U16;
    else if (!forceGlobal && nR <= 65536 && class16 <= ldsTableMax)
      img.tableKind = REDGPU_TAB_LDS_CLASS_U16;
    else if (!forceGlobal && nR <= 65536 && ldsTableMax >= 8u * 256u)
      img.tableKind = REDGPU_TAB_HOT_ROWS;
    else if (nR <= 65536)
      img.tableKind = REDGPU_TAB_GLOBAL_U16;
    else
      img.tableKind = REDGPU_TAB_GLOBAL_U32;
  }

  // Hot rows: which states deserve an LDS row?  Model: expected visits of a walk from the
  // initial state over 64 bytes drawn half from all 256 values and half from printable ASCII
  // (power iteration over the class table, class weights = bytes per class); ties - states the
  // model never reaches, e.g. deep inside a signature - go by breadth-first distance from
  // the initial state.  Pure dead ends are never looked up (the walk stops there).
  // The model is also what flags "early death" for every placement (it is skipped for tables so
  // large that the iteration itself would take long).
  std::vector<uint8_t> isHot(stateCnt, 0);
  // Hot rows are for tables that do not fit LDS - and, once visits have been OBSERVED
  // (redgpu_dfa_tune), also for a table of more than 256 states that would fit: when
  // practically all of the real walk is hot the streaming kernels (one-byte hot index) run it
  // ~3x faster than k_generic does from an LDS class table.  On the model alone that switch is
  // not made: a cold excursion per match costs more than the LDS class table saves (343-state
  // URI DFA, URL in every 8th 64-byte line, untuned: 163 us against 119 us).
  const uint32_t fitKind = img.tableKind;
  const bool haveMeasured = measured && measured->size() == stateCnt;
  // a class table of <= 64 KB (<= 127 classes) has its own streaming form (k_stream cls: two
  // lookups per byte, no cold path, 2.3 TB/s on that same DFA and input) and stays where it is
  const bool clsStream = nCls <= 127 && uint64_t(reach.size()) * nCls * 2u <= 158720u - 256u;
  const bool wantHot = !forceGlobal && reach.size() > 256 && reach.size() <= 65536 &&
                       ldsTableMax >= 8u * 256u &&
                       (fitKind == REDGPU_TAB_HOT_ROWS ||
                        (((haveMeasured && !clsStream) || forceHot) &&
                         (fitKind == REDGPU_TAB_LDS_FUSED_U16 ||
                          fitKind == REDGPU_TAB_LDS_CLASS_U16)));
  if (wantHot || uint64_t(reach.size()) * nCls <= (8ull << 20)) {
    std::vector<double> w(nCls, 0.0);
    for (uint32_t b = 0; b < 256; ++b) {
      double wb = 0.5 / 256.0;
      if ((b >= 0x20 && b <= 0x7e) || b == 0x09) wb += 0.5 / 96.0;
      w[h[kOffEquivMap + b]] += wb;
    }
    std::vector<double> p(stateCnt, 0.0), q(stateCnt, 0.0), visits(stateCnt, 0.0);
    p[rawInit] = 1.0;
    double died = 0.0;  // mass that has reached a pure dead end so far
    for (int step = 0; step < 64; ++step) {
      std::fill(q.begin(), q.end(), 0.0);
      for (uint32_t s : reach) {
        if (p[s] < 1e-15 || klass(s) == 0) continue;
        for (uint32_t c = 0; c < nCls; ++c) {
          uint32_t t = 0;
          targetOf(s, c, t);
          q[t] += p[s] * w[c];
        }
      }
      for (uint32_t s : reach) {
        visits[s] += q[s];
        if (klass(s) == 0) died += q[s];
      }
      if (step == 15) img.earlyDeath = died > 0.5;
      p.swap(q);
    }
    img.forgetful = p[rawInit] >= 0.7;
    if (measured && measured->size() == stateCnt) {
      img.tuned = true;
      // observed visits decide; the model (scaled far below one observed visit) breaks ties
      double msum = 0.0, vsum = 0.0;
      for (uint32_t s : reach) { msum += (*measured)[s]; vsum += visits[s]; }
      if (msum > 0.0)
        for (uint32_t s : reach)
          visits[s] = (*measured)[s] / msum + 1e-9 * (vsum > 0.0 ? visits[s] / vsum : 0.0);
    }
    if (wantHot) {
    std::vector<uint32_t> cand;
    double total = 0.0;
    for (uint32_t s : reach)
      if (klass(s) != 0) { cand.push_back(s); total += visits[s]; }
    std::sort(cand.begin(), cand.end(), [&](uint32_t a, uint32_t b) {
      if (visits[a] != visits[b]) return visits[a] > visits[b];
      return bfsRank[a] < bfsRank[b];
    });
    // as many rows as the budget holds, never more than 254: hot states are indexed with one
    // byte, 255 means "not hot" and index 0 may stand for the pure dead ends
    const uint32_t maxRows =
        std::min<uint64_t>(254u, std::min<uint64_t>(cand.size(), ldsTableMax / 256u));
    auto coverage = [&](uint32_t rows) {
      double c = 0.0;
      for (uint32_t i = 0; i < rows; ++i) c += visits[cand[i]];
      return total > 0.0 ? c / total : 1.0;
    };
    const uint32_t rows = maxRows;
    // a table that fits LDS whole gives way only to a hot set that covers practically every
    // visit; one that does not fit takes hot rows unless there is no locality to exploit (a
    // dense random DFA), in which case the whole table is left to L2
    const double need = fitKind == REDGPU_TAB_HOT_ROWS ? 0.5 : 0.999;
    if (coverage(rows) < need && !forceHot) {
      img.tableKind = fitKind == REDGPU_TAB_HOT_ROWS ? uint32_t(REDGPU_TAB_GLOBAL_U16) : fitKind;
    } else {
      img.tableKind = REDGPU_TAB_HOT_ROWS;
      for (uint32_t i = 0; i < rows; ++i) isHot[cand[i]] = 1;
      if (!isHot[rawInit] && klass(rawInit) != 0 && rows) {
        isHot[cand[rows - 1]] = 0;  // the initial state always has a row
        isHot[rawInit] = 1;
      }
      img.nHot = rows;
      img.hotCoveragePpm = uint32_t(coverage(rows) * 1e6);
    }
    }  // wantHot
  }

  // Sparse form (REDGPU_TAB_LDS_SPARSE).  A signature set - anchored patterns that share little -
  // has a class table far too big for LDS of which almost every entry is the dead state
  // (LOG-100: 3150 states x 40 classes = 252 KB, 5.9 % of the entries are anything else).  Its
  // walks die within a few bytes on ordinary lines and run through one signature's private
  // states on matching ones: no set of 254 hot rows covers those, every step of a match is an
  // L2 round trip.  Row displacement (Tarjan & Yao's comb) packs the exceptions into one slot
  // array - each row shifted until its exceptions fall on free slots, a slot remembering its
  // owner - and the whole DFA sits in LDS: base[state], then slot[base + class].
  // Taken only for early-death DFAs (they run on k_generic whatever the table) whose class
  // table does not fit; a loose-start DFA keeps the hot-row streaming kernels.
  std::vector<uint32_t> sparseBase;   // per device state (plain order)
  std::vector<uint32_t> sparseSlot;   // (owner << 16) | target, 0xffff0000 = free
  if (!forceGlobal && !forceHot && img.earlyDeath && reach.size() < 65535 &&
      (fitKind == REDGPU_TAB_HOT_ROWS || fitKind == REDGPU_TAB_GLOBAL_U16)) {
    std::vector<uint32_t> plain;
    plain.reserve(reach.size());
    for (int k = 0; k < 3; ++k)
      for (uint32_t s : reach)
        if (klass(s) == k) plain.push_back(s);
    std::vector<uint32_t> idOf(stateCnt, 0xffffffffu);
    for (uint32_t i = 0; i < plain.size(); ++i) idOf[plain[i]] = i;
    const uint32_t nS = uint32_t(plain.size());
    std::vector<uint32_t> tgt(size_t(nS) * nCls);
    std::vector<uint32_t> freq(nS, 0);
    for (uint32_t i = 0; i < nS; ++i)
      for (uint32_t c = 0; c < nCls; ++c) {
        uint32_t t = 0;
        targetOf(plain[i], c, t);
        tgt[size_t(i) * nCls + c] = idOf[t];
        ++freq[idOf[t]];
      }
    const uint32_t dflt = uint32_t(std::max_element(freq.begin(), freq.end()) - freq.begin());
    std::vector<uint32_t> rows(nS), cnt(nS, 0);
    for (uint32_t i = 0; i < nS; ++i) {
      rows[i] = i;
      for (uint32_t c = 0; c < nCls; ++c) cnt[i] += tgt[size_t(i) * nCls + c] != dflt;
    }
    std::stable_sort(rows.begin(), rows.end(), [&](uint32_t a, uint32_t b) { return cnt[a] > cnt[b]; });
    const uint64_t budgetSlots = ldsTableMax / 4u;
    std::vector<uint8_t> used;
    std::vector<uint32_t> base(nS, 0);
    uint32_t firstFree = 0, top = 0;
    bool ok = true;
    for (uint32_t i : rows) {
      if (cnt[i] == 0) continue;  // base 0: every lookup finds somebody else's slot or a free one
      uint32_t b = firstFree;
      for (;; ++b) {
        if (uint64_t(b) + nCls > budgetSlots || b > 0xffffu - nCls) { ok = false; break; }
        if (used.size() < size_t(b) + nCls) used.resize(size_t(b) + nCls, 0);
        bool fits = true;
        for (uint32_t c = 0; c < nCls && fits; ++c)
          fits = tgt[size_t(i) * nCls + c] == dflt || !used[b + c];
        if (fits) break;
      }
      if (!ok) break;
      base[i] = b;
      for (uint32_t c = 0; c < nCls; ++c)
        if (tgt[size_t(i) * nCls + c] != dflt) { used[b + c] = 1; top = std::max(top, b + c + 1); }
      while (firstFree < used.size() && used[firstFree]) ++firstFree;
    }
    const uint64_t baseBytes = (uint64_t(nS) * 2u + 15u) & ~uint64_t(15);
    const uint64_t slots = uint64_t(top) + nCls;  // lookups of rows that own nothing up there stay inside
    if (ok && baseBytes + slots * 4u <= ldsTableMax) {
      img.tableKind = REDGPU_TAB_LDS_SPARSE;
      img.nHot = 0;
      img.hotCoveragePpm = 0;
      std::fill(isHot.begin(), isHot.end(), 0);
      img.sparseDefault = dflt;
      sparseBase = base;
      sparseSlot.assign(size_t(slots), 0xffff0000u);
      for (uint32_t i = 0; i < nS; ++i)
        for (uint32_t c = 0; c < nCls; ++c)
          if (tgt[size_t(i) * nCls + c] != dflt)
            sparseSlot[base[i] + c] = (i << 16) | tgt[size_t(i) * nCls + c];
    }
  }

  // order: pure dead ends | non-accepting | accepting (stable in blob order); with hot rows
  // the two middle groups are split cold | hot and hot | cold so the hot set is contiguous
  std::vector<uint32_t> order;
  order.reserve(reach.size());
  auto take = [&](int k, int hot) {
    for (uint32_t s : reach)
      if (klass(s) == k && (hot < 0 || int(isHot[s]) == hot))
        order.push_back(s);
  };
  take(0, -1);
  img.nPureDead = uint32_t(order.size());
  if (img.tableKind == REDGPU_TAB_HOT_ROWS) {
    take(1, 0);
    img.hotLo = uint32_t(order.size());
    take(1, 1);
    img.firstAccept = uint32_t(order.size());
    take(2, 1);
    take(2, 0);
  } else {
    take(1, -1);
    img.firstAccept = uint32_t(order.size());
    take(2, -1);
  }
  img.nStates = uint32_t(order.size());
  std::vector<uint32_t> newId(stateCnt, 0xffffffffu);
  for (uint32_t i = 0; i < img.nStates; ++i)
    newId[order[i]] = i;
  img.rawOf = order;
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

  // is the leader the forced chain fixedPrefix would derive? (dfa_image.h: leaderForced)
  if (img.leaderLen > 0 && img.deadAbsorbing) {
    bool forced = true;
    uint32_t s = img.init;
    for (uint32_t i = 0; i < img.leaderLen && forced; ++i) {
      if (s >= img.firstAccept || s < img.nPureDead) { forced = false; break; }
      const uint32_t want = img.leader[i];
      if (want >= nCls) { forced = false; break; }
      for (uint32_t c = 0; c < nCls; ++c) {
        const uint32_t t = img.next[size_t(s) * nCls + c];
        if (c != want && t >= img.nPureDead) forced = false;
      }
      s = img.next[size_t(s) * nCls + want];
    }
    img.leaderForced = forced && s == img.leaderNext && s >= img.nPureDead;
  }

  // L = SIGMA* L ?  (dfa_image.h: suffixClosed)  For every class c the language of the initial
  // state must be included in that of next(init, c): walk the product from (init, next(init, c)),
  // fail on a pair whose left accepts and whose right does not.  Pairs (x, x) hold trivially.
  // Bounded: a DFA that needs more than 1 M pair visits is left unflagged.
  {
    const uint32_t n = img.nStates;
    bool closed = img.maxResult > 0 && n <= 8192;
    std::vector<uint64_t> seen;
    std::vector<std::pair<uint32_t, uint32_t>> todo;
    uint64_t visits = 0;
    if (closed) seen.assign((size_t(n) * n + 63) / 64, 0);
    auto push = [&](uint32_t x, uint32_t y) {
      if (x == y) return;
      const size_t bit = size_t(x) * n + y;
      if (seen[bit >> 6] >> (bit & 63) & 1u) return;
      seen[bit >> 6] |= 1ull << (bit & 63);
      todo.emplace_back(x, y);
    };
    for (uint32_t c = 0; closed && c < nCls; ++c) push(img.init, img.next[size_t(img.init) * nCls + c]);
    while (closed && !todo.empty()) {
      const auto [x, y] = todo.back();
      todo.pop_back();
      if (++visits > (1u << 20)) { closed = false; break; }
      if (img.result[x] > 0 && !(img.result[y] > 0)) { closed = false; break; }
      for (uint32_t c = 0; c < nCls; ++c)
        push(img.next[size_t(x) * nCls + c], img.next[size_t(y) * nCls + c]);
    }
    img.suffixClosed = closed;
    int32_t one = 0;
    bool uniform = true;
    for (uint32_t i = 0; i < n && uniform; ++i)
      if (img.result[i] > 0) {
        if (one == 0) one = img.result[i];
        uniform = img.result[i] == one;
      }
    img.uniformResult = uniform && one > 0;
  }

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
  case REDGPU_TAB_LDS_SPARSE: {
    // (device ids are the plain order the slots were packed in)
    img.sparseCombOff = uint32_t((size_t(img.nStates) * 2u + 15u) & ~size_t(15));
    img.table.assign(size_t(img.sparseCombOff) + sparseSlot.size() * 4u, 0);
    for (uint32_t i = 0; i < img.nStates; ++i) {
      const uint16_t b = uint16_t(sparseBase[i]);
      std::memcpy(&img.table[size_t(i) * 2u], &b, 2);
    }
    std::memcpy(&img.table[img.sparseCombOff], sparseSlot.data(), sparseSlot.size() * 4u);
    break;
  }
  default: {
    const uint32_t w = img.tableKind == REDGPU_TAB_GLOBAL_U32 ? 4 : 2;
    img.table.assign(size_t(img.nStates) * nCls * w, 0);
    for (size_t k = 0; k < img.next.size(); ++k)
      put(k, img.next[k], w);
    img.primaryBytes = uint32_t(img.table.size());
    if (img.tableKind == REDGPU_TAB_HOT_ROWS) {
      // [hot index][byte] u8 behind the class table, 16-byte aligned; unused rows and the
      // sink row 255 are all 255; with reachable (absorbing) pure dead ends hot index 0 is
      // their shared all-zero row
      img.hotShift = (img.nPureDead > 0 && img.deadAbsorbing) ? 1 : 0;
      img.hot8Off = uint32_t((img.table.size() + 15u) & ~size_t(15));
      img.table.resize(size_t(img.hot8Off) + 65536u, 0xff);
      uint8_t *t8 = &img.table[img.hot8Off];
      if (img.hotShift) std::memset(t8, 0, 256);
      for (uint32_t hr = 0; hr < img.nHot; ++hr)
        for (uint32_t b = 0; b < 256; ++b) {
          const uint32_t t = img.next[size_t(img.hotLo + hr) * nCls + img.equiv[b]];
          uint8_t v = 0xff;
          if (img.hotShift && t < img.nPureDead) v = 0;
          else if (t - img.hotLo < img.nHot) v = uint8_t(t - img.hotLo + img.hotShift);
          t8[size_t(hr + img.hotShift) * 256 + b] = v;
        }
    }
  }
  }
  if (img.primaryBytes == 0) img.primaryBytes = uint32_t(img.table.size());
  constexpr uint64_t kClsBigMax = 158720u - 256u;  // k_stream's big LDS array minus eq2
  if ((img.tableKind == REDGPU_TAB_LDS_CLASS_U16 || img.tableKind == REDGPU_TAB_LDS_FUSED_U16) &&
      nCls <= 127 && uint64_t(img.nStates) * nCls * 2u <= kClsBigMax) {
    img.clsIndexForm = uint64_t(img.nStates) * nCls * 2u > 65536u;
    img.clsRowBytes = nCls * 2u;
    img.clsOff = uint32_t((img.table.size() + 15u) & ~size_t(15));
    const uint32_t rows = img.nStates * img.clsRowBytes;
    img.clsBytes = (256u + rows + 15u) & ~15u;
    img.table.resize(size_t(img.clsOff) + img.clsBytes, 0);
    uint8_t *base = &img.table[img.clsOff];
    for (uint32_t b = 0; b < 256; ++b) base[b] = uint8_t(2u * img.equiv[b]);
    for (uint32_t i = 0; i < img.nStates; ++i)
      for (uint32_t c = 0; c < nCls; ++c) {
        const uint32_t t = img.next[size_t(i) * nCls + c];
        const uint16_t off = uint16_t(img.clsIndexForm ? t : t * img.clsRowBytes);
        std::memcpy(base + 256 + size_t(i) * img.clsRowBytes + 2 * c, &off, 2);
      }
  }
  errCode = REDGPU_OK;
  return std::string();
}

} // namespace redgpu
