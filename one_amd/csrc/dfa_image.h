// dfa_image.h - host-side reader of RED's serialized DFA ("REDA") and its repack into the
// layout the gfx950 kernels walk.  Pure host C++ (no HIP) so it is testable without a GPU.
//
// Reference layout being read (citations relative to /root/reference/quol/red/):
//   FileHeader            include/Serializer.h:42-59   (288 bytes, little-endian, packed)
//   StateDirect1/2/4 rows include/Serializer.h:62-77, lib/Serializer.cpp:39-53
//   base / leader / map   lib/Executable.cpp:159-170
//   header validation     lib/Serializer.cpp:270-306, include/Fnv.h:36-66
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace redgpu {

constexpr size_t kHeaderBytes = 288;

// What lives in HBM/LDS.  States are RENUMBERED: only states reachable from the initial
// state are kept, ordered so that the two per-byte predicates of the reference's loops become
// integer compares on the state index instead of a second table lookup:
//     s <  nPureDead    <=>  DfaProxy::pureDeadEnd()   (include/Proxy.h:139-141)
//     s >= firstAccept  <=>  DfaProxy::result() > 0    (include/Proxy.h:131-133)
struct DfaImage {
  // from the header
  uint32_t format = 0;       // 1 / 2 / 4
  uint32_t nClasses = 0;     // maxChar_ + 1
  uint32_t leaderLen = 0;
  uint32_t statesTotal = 0;  // stateCnt_
  uint32_t checksum = 0;
  uint8_t  equiv[256] = {};  // byte -> class
  uint8_t  leader[256] = {}; // class-space fixed prefix
  // renumbered automaton
  uint32_t nStates = 0;      // reachable
  uint32_t init = 0;         // device index of the initial state (initialOff_)
  uint32_t leaderNext = 0;   // device index of the state after the leader (leaderOff_)
  uint32_t nPureDead = 0;
  uint32_t firstAccept = 0;
  int32_t  maxResult = 0;
  bool     deadAbsorbing = true;  // every pure dead end self-loops on every class
  // The leader is what DfaObj::fixedPrefix (lib/Dfa.cpp:146-168) derives from THIS table: walking
  // its classes from the initial state reaches leaderNext, every state on the way is non-accepting
  // and has no other way out than the leader's class (anything else falls into an absorbing pure
  // dead end).  Then check<style,true> (include/Matcher.h:370-377: compare the leader, start in
  // leaderOff_) is the plain walk from the initial state with ONE difference - the reference
  // never looks at the result of the post-leader state itself unless the input ends there - and
  // the streaming kernels can run it (k_stream.h: Batch::ignoreAcceptUpTo).
  bool     leaderForced = false;
  // "Start bytes" for scan / search: the only input bytes at which an attempt can survive its
  // first step - with the leader, the bytes of the leader's first class; without, the bytes
  // whose transition out of the initial state is not a pure dead end.  Up to 4 are listed
  // (packed, one per byte of the word) so that the kernels can test a whole input word at once;
  // count 0xff = more than 4 (no word filter).
  uint32_t startLeadWord = 0, startLeadCount = 0xff;
  uint32_t startFreeWord = 0, startFreeCount = 0xff;
  // ...and the bytes that may FOLLOW a start byte in an attempt that survives its second step
  // (leader: the second class; else: not a pure dead end after some start byte, and no start
  // byte accepts at once).  Same packing; 0xff = no second filter.
  uint32_t start2LeadWord = 0, start2LeadCount = 0xff;
  uint32_t start2FreeWord = 0, start2FreeCount = 0xff;
  // The same two sets in full, one flag byte per input byte ([0] without the leader, [1] with
  // it): bit 0 = start byte, bit 1 = may follow one.  For DFAs with more than 4 start bytes
  // (a pattern that begins with a character class) k_scan_marked tests bytes against this table.
  uint8_t  startFlags[2][256] = {};
  uint32_t startTotal[2] = {256, 256};   // how many byte values are start bytes
  bool     startFollow[2] = {false, false};  // bit 1 is meaningful (else: anything may follow)
  std::vector<int32_t>  result;   // [nStates]
  std::vector<uint32_t> next;     // [nStates][nClasses], device indices
  std::vector<uint32_t> rawOf;    // [nStates] device index -> state id in the blob
  // table chosen for the device
  uint32_t tableKind = 0;         // REDGPU_TAB_*
  std::vector<uint8_t> table;     // packed bytes of that table (+ the appended forms below)
  uint32_t primaryBytes = 0;      // size of the tableKind table itself, before anything appended
  // REDGPU_TAB_HOT_ROWS: the class table (u16) is followed, at table[hot8Off], by a 64 KB u8
  // table [hot index][byte] -> hot index of the target, 255 = target outside the hot set (row
  // 255 is an absorbing sink).  Hot index of device state s in [hotLo, hotLo + nHot) is
  // s - hotLo + hotShift; with hotShift = 1 hot index 0 stands for every pure dead end (one
  // absorbing all-zero row).  The order is then
  //   pure dead | cold non-accepting | hot non-accepting | hot accepting | cold accepting
  // so that the hot set is ONE index range that straddles firstAccept.
  uint32_t hotLo = 0, nHot = 0, hot8Off = 0, hotShift = 0;
  // LDS_CLASS_U16 / LDS_FUSED_U16 DFAs whose class table is at most 64 KB (and <= 127 classes)
  // also carry the streaming kernel's form of it at table[clsOff]: 256 bytes eq2 (byte -> 2 x
  // class), then nStates rows of nClasses u16 = byte offset of the target's row (clsRowBytes =
  // 2 x nClasses); clsBytes = 256 + rows, rounded up to 16.  0 = not built.
  uint32_t clsOff = 0, clsRowBytes = 0, clsBytes = 0;
  // ... and above 64 KB (up to what LDS holds) the same blob in INDEX form: entries are plain
  // state indices and the kernel multiplies (v_mad_u32_u24) instead of adding
  bool clsIndexForm = false;
  // REDGPU_TAB_LDS_SPARSE: table = [base u16[nStates], padded to 16][slot u32[...]]; a slot is
  // (owner state << 16) | target, 0xffff0000 when free; lookups of (state, class) read slot
  // base[state] + class and fall back to sparseDefault when the owner is somebody else
  uint32_t sparseCombOff = 0, sparseDefault = 0;
  uint32_t hotCoveragePpm = 0;    // modelled share of visits landing on hot rows
  // L = SIGMA* L: whatever the DFA accepts it also accepts behind any prefix (a pattern added
  // with a loose start).  Then an attempt of scan / search that walked to the end of the line
  // without an accepting state proves that no later start position can accept either (the
  // text it would read is a suffix of what this attempt read), and the sliding loop can stop.
  // Decided by language inclusion L(init) <= L(next(init, c)) for every class c (product walk).
  bool     suffixClosed = false;
  // every accepting state reports the same result (a single pattern, or several under one id):
  // check then answers the same for every style but styFull - whichever accepting state a style
  // picks, its result is that one value
  bool     uniformResult = false;
  bool     forgetful = false;     // the model's walk is back in the initial state most of the
                                  // time (>= 70 % of its mass after 64 bytes): chunks of a line
                                  // may be walked from the initial state as a guess (k_chunk.h)
  bool     tuned = false;         // the hot rows were ranked by observed visits (redgpu_dfa_tune)
  bool     earlyDeath = false;    // the model's walk is in a pure dead end within 16 bytes
                                  // more often than not (anchored DFA on arbitrary text)
};

// lib/Serializer.cpp:270-298, message text verbatim; nullptr when the header is good.
const char *checkHeader(const void *ptr, size_t len);

// include/Fnv.h:36-66 (32-bit), lib/Serializer.cpp:301-306
uint32_t fnv1a32(const void *ptr, size_t len);
uint32_t calcChecksum(const void *ptr, size_t len);

// Parses + bounds-checks + renumbers.  Returns "" on success, else the error message
// (code: REDGPU_EAPI for a bad blob, REDGPU_ELIMIT for capacity).
// `measured` (optional, one entry per state of the BLOB, stateCnt_ of them): visit counts
// observed on real input (redgpu_dfa_tune); they then rank the hot rows, the model only
// breaking ties.
std::string buildImage(const void *reda, size_t len, uint32_t ldsTableMax, bool forceGlobal,
                       DfaImage &img, int &errCode, bool forceHot = false,
                       const std::vector<double> *measured = nullptr);

} // namespace redgpu
