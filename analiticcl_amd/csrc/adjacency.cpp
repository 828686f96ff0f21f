// adjacency.cpp -- builds the signature adjacency lists (adjacency.h) from the lexicon image; host only, threaded.
#include "adjacency.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>

#include "host_model.h"
#include "sig_hash.h"

namespace anx {

AdjIndex::~AdjIndex() {
  free(planes);
  free(ids);
}

uint32_t AdjIndex::find(uint32_t lo, uint32_t hi) const {
  if (hash.empty()) return 0;
  uint32_t h = sig_hash(lo, hi) & hash_mask;
  for (int p = 0; p < 17; ++p) {  // every key within 16 slots of its home
    const AdjSlot& s = hash[h];
    if (!s.hdr1) return 0;
    if (s.lo == lo && s.hi == hi) return s.hdr1;
    h = (h + 1) & hash_mask;
  }
  return 0;
}

namespace {

struct Offset {
  int8_t d[8];
  int8_t sum;  // sum of d = length difference
  int8_t l1;   // sum of |d|
};

std::vector<Offset> ball_offsets(int ngroups, int radius) {
  std::vector<Offset> out;
  Offset cur{};
  std::function<void(int, int)> rec = [&](int g, int left) {
    if (g == ngroups) {
      Offset o = cur;
      int s = 0, a = 0;
      for (int i = 0; i < 8; ++i) { s += o.d[i]; a += o.d[i] < 0 ? -o.d[i] : o.d[i]; }
      o.sum = (int8_t)s;
      o.l1 = (int8_t)a;
      out.push_back(o);
      return;
    }
    for (int x = -left; x <= left; ++x) {
      cur.d[g] = (int8_t)x;
      rec(g + 1, left - (x < 0 ? -x : x));
    }
    cur.d[g] = 0;
  };
  rec(0, radius);
  return out;
}

// sig + offset; false when a group sum leaves [0, 255] or the length leaves [1, kMaxSymbols]
inline bool apply_offset(uint64_t sig, const Offset& o, uint64_t* out) {
  uint64_t r = 0;
  int len = 0;
  for (int g = 0; g < 8; ++g) {
    const int v = (int)((sig >> (8 * g)) & 0xFFu) + o.d[g];
    if (v < 0 || v > 255) return false;
    len += v;
    r |= (uint64_t)v << (8 * g);
  }
  if (len < 1 || len > kMaxSymbols) return false;
  *out = r;
  return true;
}

struct SigTable {  // lexicon signature -> index of its run
  std::vector<uint32_t> slot;  // index + 1, 0 = empty
  uint32_t mask = 0;
  const LexiconImage* img = nullptr;
  void build(const LexiconImage& im) {
    img = &im;
    uint32_t n = 64;
    while (n < 2 * im.nsigs) n <<= 1;
    mask = n - 1;
    slot.assign(n, 0u);
    for (uint32_t i = 0; i < im.nsigs; ++i) {
      uint32_t h = sig_hash(im.sig_lo[i], im.sig_hi[i]) & mask;
      while (slot[h]) h = (h + 1) & mask;
      slot[h] = i + 1;
    }
  }
  int find(uint64_t sig) const {
    const uint32_t lo = (uint32_t)sig, hi = (uint32_t)(sig >> 32);
    uint32_t h = sig_hash(lo, hi) & mask;
    while (uint32_t s = slot[h]) {
      if (img->sig_lo[s - 1] == lo && img->sig_hi[s - 1] == hi) return (int)(s - 1);
      h = (h + 1) & mask;
    }
    return -1;
  }
};

template <typename F>
void parallel_chunks(size_t n, size_t chunk, unsigned threads, F&& f) {
  std::atomic<size_t> next{0};
  auto work = [&]() {
    for (;;) {
      const size_t lo = next.fetch_add(chunk);
      if (lo >= n) return;
      f(lo, std::min(n, lo + chunk));
    }
  };
  if (threads <= 1 || n <= chunk) { work(); return; }
  std::vector<std::thread> th;
  for (unsigned t = 0; t < threads; ++t) th.emplace_back(work);
  for (auto& x : th) x.join();
}

}  // namespace

void build_adjacency(const LexiconImage& img, int closure, size_t budget_bytes, unsigned threads, AdjIndex& out) {
  const auto t0 = std::chrono::steady_clock::now();
  const bool timing = getenv("ANX_ADJ_TIMING") != nullptr;
  auto tl = t0;
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[anx adjacency] %-24s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t - tl).count());
    tl = t;
  };
  out.hash.clear(); out.hdr.clear();
  free(out.planes); free(out.ids);
  out.planes = nullptr; out.ids = nullptr;
  out.rows = 0; out.records = 0; out.nsig_closure = out.nsig_kept = 0;
  out.nsig_lexicon = img.nsigs;
  if (img.nsym > 32 || img.nsigs == 0 || img.cls_bits.empty()) return;  // the bit-plane scan only
  closure = std::max(0, std::min(closure, kAdjMaxClosure));
  threads = std::max(1u, threads);
  int ng = 1;
  for (uint8_t g : img.sym_group) ng = std::max(ng, (int)g + 1);
  SigTable lexsig;
  lexsig.build(img);
  const std::vector<Offset> ball = ball_offsets(ng, kAdjRadius);
  const std::vector<Offset> near = ball_offsets(ng, closure);

  // ---- the signatures that get a list: the lexicon's and everything within `closure` of one -------------------------------------
  struct Key { uint64_t sig; uint32_t tier; };
  std::vector<Key> keys;
  {
    std::vector<std::vector<Key>> part(threads);
    std::atomic<unsigned> tid{0};
    std::vector<std::thread> th;
    auto gen = [&](unsigned t) {
      std::vector<Key>& v = part[t];
      const size_t lo = (size_t)img.nsigs * t / threads, hi = (size_t)img.nsigs * (t + 1) / threads;
      v.reserve((hi - lo) * near.size());
      for (size_t i = lo; i < hi; ++i) {
        const uint64_t s = (uint64_t)img.sig_lo[i] | (uint64_t)img.sig_hi[i] << 32;
        for (const Offset& o : near) {
          uint64_t u;
          if (apply_offset(s, o, &u)) v.push_back(Key{u, (uint32_t)o.l1});
        }
      }
      std::sort(v.begin(), v.end(), [](const Key& a, const Key& b) { return a.sig != b.sig ? a.sig < b.sig : a.tier < b.tier; });
      v.erase(std::unique(v.begin(), v.end(), [](const Key& a, const Key& b) { return a.sig == b.sig; }), v.end());
    };
    if (threads == 1) gen(0);
    else {
      for (unsigned t = 0; t < threads; ++t) th.emplace_back(gen, t);
      for (auto& x : th) x.join();
    }
    (void)tid;
    size_t total = 0;
    for (auto& v : part) total += v.size();
    keys.reserve(total);
    for (auto& v : part) { keys.insert(keys.end(), v.begin(), v.end()); std::vector<Key>().swap(v); }
    std::sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) { return a.sig != b.sig ? a.sig < b.sig : a.tier < b.tier; });
    keys.erase(std::unique(keys.begin(), keys.end(), [](const Key& a, const Key& b) { return a.sig == b.sig; }), keys.end());
  }
  lap("closure");
  const size_t nk = keys.size();
  out.nsig_closure = (uint32_t)nk;

  // ---- pass 1: rows per section of every list ------------------------------------------------------------------------------------
  auto entries_of_run = [&](int i) { return img.cls_off[std::min(img.sig_cbeg[i + 1], img.nclasses)] - img.cls_off[std::min(img.sig_cbeg[i], img.nclasses)]; };
  std::vector<AdjHdr> hdr(nk);
  std::vector<uint32_t> nrec(nk, 0);
  parallel_chunks(nk, 256, threads, [&](size_t lo, size_t hi) {
    for (size_t x = lo; x < hi; ++x) {
      uint32_t cnt[kAdjSections] = {};
      for (const Offset& o : ball) {
        uint64_t v;
        if (!apply_offset(keys[x].sig, o, &v)) continue;
        const int i = lexsig.find(v);
        if (i >= 0) cnt[o.sum + kAdjRadius] += entries_of_run(i);
      }
      uint32_t rows = 0, recs = 0;
      for (int s = 0; s < kAdjSections; ++s) {
        rows += (cnt[s] + kAdjRow - 1) / kAdjRow;
        recs += cnt[s];
        hdr[x].cum[s] = rows;
      }
      hdr[x].row0 = 0;
      nrec[x] = recs;
    }
  });

  {  // expected records per query of every length (queries drawn like lexicon entries)
    double num[256] = {}, den[256] = {};
    std::vector<double> cnum(64 * 1024, 0.0), cden(64 * 1024, 0.0);
    out.class_nsig.assign(64 * 1024, 0.0f);
    for (size_t x = 0; x < nk; ++x) {
      if (keys[x].tier != 0) continue;
      const int i = lexsig.find(keys[x].sig);
      if (i < 0) continue;
      int len = 0;
      for (int g = 0; g < 8; ++g) len += (int)((keys[x].sig >> (8 * g)) & 0xFFu);
      const double e = (double)entries_of_run(i);
      num[std::min(len, 255)] += e * (double)nrec[x];
      den[std::min(len, 255)] += e;
      if (len < 64) {
        const size_t c = (size_t)len * 1024 + std::min<size_t>(keys[x].sig & 0xFFu, 31) * 32 + std::min<size_t>((keys[x].sig >> 8) & 0xFFu, 31);
        cnum[c] += e * (double)nrec[x];
        cden[c] += e;
        out.class_nsig[c] += 1.0f;
      }
    }
    for (int L = 0; L < 256; ++L) out.len_records[L] = den[L] > 0.0 ? num[L] / den[L] : 0.0;
    out.class_records.assign(64 * 1024, 0.0f);
    for (size_t c = 0; c < cnum.size(); ++c) out.class_records[c] = cden[c] > 0.0 ? (float)(cnum[c] / cden[c]) : 0.0f;
  }
  lap("pass 1 (rows)");
  // ---- which lists fit the budget: by (distance from the lexicon, rows) ascending ---------------------------------------------
  const uint64_t row_bytes = (uint64_t)kAdjRow * (sizeof(AdjPlanes) + sizeof(uint32_t));
  std::vector<uint8_t> keep(nk, 1);
  {
    uint64_t total = 0;
    for (size_t x = 0; x < nk; ++x) total += hdr[x].cum[kAdjSections - 1];
    out.rows_wanted = total;
    if (total * row_bytes > budget_bytes) {
      std::vector<uint32_t> ord(nk);
      for (size_t x = 0; x < nk; ++x) ord[x] = (uint32_t)x;
      std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {
        if (keys[a].tier != keys[b].tier) return keys[a].tier < keys[b].tier;
        const uint32_t ra = hdr[a].cum[kAdjSections - 1], rb = hdr[b].cum[kAdjSections - 1];
        return ra != rb ? ra < rb : a < b;
      });
      uint64_t used = 0;
      for (uint32_t x : ord) {
        const uint64_t need = (uint64_t)hdr[x].cum[kAdjSections - 1] * row_bytes;
        if (used + need <= budget_bytes) used += need;
        else keep[x] = 0;
      }
    }
  }
  std::vector<uint32_t> kept;
  kept.reserve(nk);
  uint64_t rows = 0;
  for (size_t x = 0; x < nk; ++x)
    if (keep[x]) {
      if (rows + hdr[x].cum[kAdjSections - 1] >= 0xFFFFFFFFull) { keep[x] = 0; continue; }  // row numbers are 32-bit
      hdr[x].row0 = (uint32_t)rows;
      rows += hdr[x].cum[kAdjSections - 1];
      out.records += nrec[x];
      kept.push_back((uint32_t)x);
    }
  out.rows = rows;
  out.nsig_kept = (uint32_t)kept.size();
  out.planes = static_cast<AdjPlanes*>(malloc(std::max<size_t>(rows * kAdjRow * sizeof(AdjPlanes), 16)));
  out.ids = static_cast<uint32_t*>(malloc(std::max<size_t>(rows * kAdjRow * sizeof(uint32_t), 16)));

  lap("budget + alloc");
  // ---- pass 2: the records, per section in entry order, every section padded to whole rows ---------------------------------------
  const uint32_t* plane1 = img.cls_bits.data();
  const uint32_t* plane2 = img.cls_bits.data() + img.cstride;
  parallel_chunks(kept.size(), 64, threads, [&](size_t lo, size_t hi) {
    struct Run { uint32_t sec, c0, c1; };
    std::vector<Run> runs;
    for (size_t y = lo; y < hi; ++y) {
      const size_t x = kept[y];
      runs.clear();
      for (const Offset& o : ball) {
        uint64_t v;
        if (!apply_offset(keys[x].sig, o, &v)) continue;
        const int i = lexsig.find(v);
        if (i >= 0) runs.push_back(Run{(uint32_t)(o.sum + kAdjRadius), std::min(img.sig_cbeg[i], img.nclasses), std::min(img.sig_cbeg[i + 1], img.nclasses)});
      }
      std::sort(runs.begin(), runs.end(), [](const Run& a, const Run& b) { return a.sec != b.sec ? a.sec < b.sec : a.c0 < b.c0; });
      size_t r = 0;
      for (uint32_t s = 0; s < (uint32_t)kAdjSections; ++s) {
        uint64_t p = ((uint64_t)hdr[x].row0 + (s ? hdr[x].cum[s - 1] : 0u)) * kAdjRow;
        const uint64_t pend = ((uint64_t)hdr[x].row0 + hdr[x].cum[s]) * kAdjRow;
        for (; r < runs.size() && runs[r].sec == s; ++r)
          for (uint32_t c = runs[r].c0; c < runs[r].c1; ++c)
            for (uint32_t e = img.cls_off[c]; e < img.cls_off[c + 1]; ++e, ++p) {
              out.planes[p] = AdjPlanes{plane1[c], plane2[c]};
              out.ids[p] = e;
            }
        for (; p < pend; ++p) {
          out.planes[p] = AdjPlanes{0u, 0u};   // shares no symbol with anything: never a hit
          out.ids[p] = img.nentries;           // the padding scan record
        }
      }
    }
  });

  lap("pass 2 (records)");
  // ---- the table signature -> list ---------------------------------------------------------------------------------------------------
  out.hdr.resize(kept.size());
  for (size_t y = 0; y < kept.size(); ++y) out.hdr[y] = hdr[kept[y]];
  uint32_t hsize = 64;
  while (hsize < 2 * (uint32_t)kept.size()) hsize <<= 1;
  for (;; hsize <<= 1) {
    out.hash.assign(hsize, AdjSlot{0u, 0u, 0u, 0u});
    bool ok = true;
    for (size_t y = 0; y < kept.size() && ok; ++y) {
      const uint32_t lo = (uint32_t)keys[kept[y]].sig, hi = (uint32_t)(keys[kept[y]].sig >> 32);
      uint32_t h = sig_hash(lo, hi) & (hsize - 1);
      int probes = 0;
      while (out.hash[h].hdr1 && probes < 16) { h = (h + 1) & (hsize - 1); ++probes; }
      if (probes == 16) { ok = false; break; }
      out.hash[h] = AdjSlot{lo, hi, (uint32_t)y + 1u, out.hdr[y].cum[kAdjSections - 1]};
    }
    if (ok) break;
  }
  out.hash_mask = hsize - 1;
  lap("hash table");
  out.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace anx
