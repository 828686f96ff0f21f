// adjacency.h -- signature adjacency lists of the lexicon image ("ball lists"), built once per model on the host.
//
// find_nearest_anahashes (/root/reference/src/lib.rs:1143-1308) returns the classes within anagram distance k of the focus; the scan
// kernel's exact per-class test (kernels_scan.hpp) decides that, and the group-sum signature only PRUNES: a class whose signature is
// further than k (L1) from the query's cannot pass.  Until round 4 every scan tile enumerated the signatures of that L1 ball itself
// (<= 575 hash probes for 7 groups, k = 3), staged the runs it found and gathered their 16-byte scan records.  The ball of a
// signature depends on the lexicon only, so it is now an INDEX: for every signature u of the lexicon and of its neighbourhood
// (signatures within `closure` of a lexicon signature: where the queries of a spelling corrector fall) the records of all lexicon
// signatures within kAdjRadius of u, as rows of 64 records, ordered by record length so that a row has ONE length (the tile's
// threshold and its "fails the length test of the DL" flag are wave-uniform per row), padded with never-matching records.
// A tile whose signature has a list streams it (coalesced 8 + 4 bytes per record); the others keep the probe walk.
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>

namespace anx {

struct LexiconImage;

constexpr int kAdjRadius = 3;             // lists hold the ball of this radius; tiles with k <= kAdjRadius use the sections |lc - lq| <= k
constexpr int kAdjSections = 2 * kAdjRadius + 1;
constexpr int kAdjMaxClosure = 2;         // lists are built for signatures within 0..kAdjMaxClosure of a lexicon signature: ONE clamp for the switch parser,
                                          // the host builder (the device builder's test reference) and the device builder
constexpr uint32_t kAdjRow = 64;          // records per row (one wave)

struct AdjHdr {        // 32 bytes, read with scalar loads
  uint32_t row0;       // first row of the list
  uint32_t cum[kAdjSections];  // cum[i] = rows of sections 0..i; section i holds the records of length L - kAdjRadius + i (L = the signature's length)
};
struct AdjSlot {       // open-addressing table signature -> list, every key within 16 slots of its home; hdr1 == 0: empty slot
  uint32_t lo, hi;     // the signature (group sums as bytes)
  uint32_t hdr1;       // header index + 1
  uint32_t rows;       // rows of the list (cost estimates of the tile builders)
};
struct AdjPlanes { uint32_t p1, p2; };  // thermometer planes 1 and 2 of the record's class (planes 3 / 4 are gathered by entry id when a tile needs them)

struct AdjIndex {
  std::vector<AdjSlot> hash;
  uint32_t hash_mask = 0;
  std::vector<AdjHdr> hdr;
  AdjPlanes* planes = nullptr;   // [rows * 64] (malloc: filled by the builder's threads, not value-initialised)
  uint32_t* ids = nullptr;       // [rows * 64] entry ids; padding = nentries (the never-matching scan record)
  uint64_t rows = 0;
  // statistics
  uint32_t nsig_lexicon = 0, nsig_closure = 0, nsig_kept = 0;
  uint64_t records = 0;          // without padding
  uint64_t rows_wanted = 0;      // rows of every list of the closure (what an unlimited budget would keep)
  // per length L: records in the ball of a lexicon entry's signature, averaged over the entries of that length (0: no entry) -- what
  // the scan tests per query of that length when the queries resemble the lexicon: the prior of the length-partitioned split (capi.cpp)
  double len_records[256] = {};
  // the same by (length, group sum 0, group sum 1) for lengths < 64, sums clipped to 31: index length * 1024 + sum0 * 32 + sum1 -- the
  // classes the split orders the inputs by; within one length the dense signatures (common letters) meet several times the records
  std::vector<float> class_records;
  std::vector<float> class_nsig;   // lexicon signatures per class: with the inputs of a class in a batch, how full its scan tiles get
  double build_ms = 0.0;
  AdjIndex() = default;
  AdjIndex(const AdjIndex&) = delete;
  AdjIndex& operator=(const AdjIndex&) = delete;
  ~AdjIndex();
  // header index + 1 of the signature's list, 0 = none
  uint32_t find(uint32_t lo, uint32_t hi) const;
};

// closure: lists are built for every signature within this L1 distance of a lexicon signature (0..2); budget_bytes caps the
// planes + ids arrays (lists are kept by (distance from the lexicon, size) ascending until the budget is used up)
void build_adjacency(const LexiconImage& img, int closure, size_t budget_bytes, unsigned threads, AdjIndex& out);

}  // namespace anx
