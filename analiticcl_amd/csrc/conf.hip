// conf.hip -- confusable weighting of the ranked lists ON THE DEVICE (gfx950 / CDNA4).
//
// Replaces, for a whole batch at once, compute_confusable_weight (/root/reference/src/lib.rs:1733-1756) and the rescoring steps
// around it: late (default, :1591-1595: weight the cropped list, re-rank, then the cutoff :1598-1622) and early
// (set_confusables_before_pruning, :1505-1508: weight every candidate before the crop).  The edit script
// (sesdiff::shortest_edit_script = dissimilar / diff-match-patch, restated) and the pattern matcher (src/confusables.rs:47-127) are
// the SAME source the host compiles (confusables_core.hpp): one lane per row, fixed-capacity working memory per lane in HBM, the
// memories of a wave's 64 lanes interleaved word by word.
//   k_conf_screen  : one lane per row slot: decode nothing yet -- ASCII presence bits of the input (from its UTF-8 bytes) against
//                    the precomputed bits of the vocabulary item; rows no pattern can match get weight 1 (62 % on BASELINE
//                    configs[2]), the others are appended to a dense list
//   k_conf_script  : one lane per listed row: decode the input, edit script, found_in over the patterns -> weight
//   k_conf_apply   : early: multiply the candidate rows' scores;  late: multiply, stable re-sort (rank_cmp, src/types.rs:344-365),
//                    cutoff, new row count
// A row the fixed-size context cannot hold (strings beyond 64 code points, arena / diff overflow) raises a flag and the host
// repeats the batch with the host-side weighting (capi.cpp): rare, and it keeps the device path free of "approximately" results.
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "engine_internal.h"

namespace anx {

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      err = std::string(#expr) + ": " + hipGetErrorString(_e);                                 \
      return ANX_ENODEVICE;                                                                    \
    }                                                                                          \
  } while (0)

#include "kernels_common.hpp"

struct DeviceConf {
  int device = 0;
  cdiff::FlatConf* conf = nullptr;
  cdiff::FlatOp* ops = nullptr;
  cdiff::FlatOpt* opts = nullptr;
  uint32_t* pool = nullptr;
  uint32_t nconf = 0;
  uint32_t* v_pool = nullptr;       // vocabulary: code points, offsets [V + 1], ASCII presence bits [V]
  uint32_t* v_off = nullptr;
  cdiff::CharSet* v_cs = nullptr;
  uint32_t nvocab = 0;
  uint32_t (*alpha)[2] = nullptr;   // char::is_alphabetic ranges
  uint32_t nalpha = 0;
  size_t built_patterns = 0, built_vocab = 0;
};

// per-lane working memory (k_conf_script): words of the lane memory of confusables_core.hpp, the 64 lanes of a wave interleaved
// (word w of lane l = block[w * 64 + l]): lanes that work on the same word of their own rows share cache lines
constexpr uint32_t CF_MAXCP = 64;      // code points per string the device handles
constexpr uint32_t CF_ARENA = 768;     // code points
constexpr uint32_t CF_DIFFS = 48;
constexpr uint32_t CF_V = 4 * (CF_MAXCP + 2) + 8;
constexpr uint32_t CF_FRAMES = 16;
constexpr uint32_t CF_IN_OFF = 0, CF_CAND_OFF = CF_MAXCP, CF_ARENA_OFF = 2 * CF_MAXCP, CF_D_OFF = CF_ARENA_OFF + CF_ARENA,
                   CF_V_OFF = CF_D_OFF + CF_DIFFS * cdiff::DIFF_WORDS, CF_F_OFF = CF_V_OFF + CF_V,
                   CF_WORDS = CF_F_OFF + CF_FRAMES * cdiff::FRAME_WORDS;
constexpr uint32_t CF_BLOCKS = 4096, CF_THREADS = 64;  // lanes in flight = working sets of a batch (262 k x 5.8 KB = 1.5 GB of HBM; 8192 blocks measured no faster)
#ifndef ANX_CF_G
#define ANX_CF_G 4
#endif
constexpr uint32_t CF_G = ANX_CF_G;  // words of a lane that stay together
typedef cdiff::Core<64, CF_G> DC;

struct ConfArgs {
  uint32_t nq, row_cap;            // row_cap: slots the row buffers hold (a run that needs more is repeated by the host)
  int early;                       // rows = c_rows (all candidates) / r_rows (ranked, r_count per query)
  const uint32_t* soff;            // [nq + 1] row segment of sorted query s
  const uint32_t* r_count;         // late
  SurvRow* c_rows;                 // early
  DevRow* r_rows;                  // late
  const uint32_t* q_orig;          // sorted query -> input index
  const uint8_t* text;             // the inputs' bytes
  const uint32_t* textoff;         // [n + 1]; input i = text[textoff[i] .. textoff[i + 1] - 1)
  double* weight;                  // per row slot
  uint2* need;                     // list of (row slot, sorted query) that need an edit script
  uint32_t* ctr;                   // [0] list length, [1] fallback rows
  cdiff::Patterns P;
  const uint32_t* v_pool; const uint32_t* v_off; const cdiff::CharSet* v_cs; uint32_t nvocab;
  const uint32_t (*alpha)[2]; uint32_t nalpha;
  uint32_t* work;                  // [CF_BLOCKS][CF_WORDS][64]
  uint32_t* key;                   // [row_cap] shape key per list entry (k_conf_key), 0xFF behind the list
  const uint32_t* order;           // [row_cap] list positions sorted by shape key: k_conf_script takes them in this order
  double cutoff_threshold; float freq_weight;
};

// the vocabulary item whose text a row is compared with: the variant itself for rows reached through a variant list
__device__ inline uint32_t row_item(const ConfArgs& a, uint32_t slot) {
  if (a.early) {
    const SurvRow r = a.c_rows[slot];
    return r.via != 0xFFFFFFFFu ? r.via : r.vocab;
  }
  const DevRow r = a.r_rows[slot];
  return r.via != 0xFFFFFFFFu ? r.via : r.vocab_id;
}

// One lane per query: the ASCII presence bits of the input once, then its rows against the patterns' presence bits.
__global__ __launch_bounds__(256) void k_conf_screen(ConfArgs a) {
  const uint32_t s = blockIdx.x * 256 + threadIdx.x;
  if (a.soff[a.nq] > a.row_cap) return;  // (uniform)
  // every lane stays until the end: the list positions are reserved once per wave (one atomic on the list counter per row and
  // lane, or even per wave and round, is what this kernel's time was: ~80 k returning atomics on one word per million queries)
  const bool live = s < a.nq;
  const uint32_t base = live ? a.soff[s] : 0u, n = !live ? 0u : a.early ? a.soff[s + 1] - base : a.r_count[s];
  const uint32_t i = n ? a.q_orig[s] : 0u, t0 = n ? a.textoff[i] : 0u, t1 = n ? a.textoff[i + 1] - 1u : 0u;
  unsigned long long needmask = 0;  // rows 0..63 of this query that need an edit script
  cdiff::CharSet ins;
  ins.w[0] = ins.w[1] = 0;
  ins.other = 0;
  for (uint32_t p = t0; p < t1; ++p) {
    const uint32_t ch = a.text[p];
    if (ch < 128u) ins.w[ch >> 6] |= 1ull << (ch & 63u);
    else ins.other = 1;
  }
  for (uint32_t k = 0; k < n; ++k) {
    const uint32_t slot = base + k, id = row_item(a, slot);
    bool need = false;
    if (id < a.nvocab) {
      const cdiff::CharSet cs = a.v_cs[id];
      // the screen of confusables_core.hpp may_match on the presence bits alone: a pattern whose every instruction is `simple` is
      // decided here; one with a multi-character or non-ASCII option, or a `$` tail, is kept for the exact test in k_conf_script
      for (uint32_t j = 0; j < a.P.nconf && !need; ++j) {
        const cdiff::FlatConf& cf = a.P.conf[j];
        bool may = true;
        for (uint32_t o_ = 0; o_ < cf.nops && may; ++o_) {
          const cdiff::FlatOp& o = a.P.ops[cf.op_begin + o_];
          const bool tail = cf.strictend && o_ == cf.nops - 1 && o.op != '=';
          if (o.simple && !tail) {
            uint64_t h0 = o.bits[0], h1 = o.bits[1];
            if (o.op != '+') { h0 &= ins.w[0]; h1 &= ins.w[1]; }
            if (o.op != '-') { h0 &= cs.w[0]; h1 &= cs.w[1]; }
            if (!(h0 | h1)) may = false;
          }
        }
        need = may;
      }
      if (need) {
        // Second look, at the MIDDLES.  Every deletion's text is made of characters of the input between the common prefix and the
        // common suffix of the two strings, every insertion's of the candidate's: diff-match-patch strips both first, bisects
        // the middles, and its clean-ups only regroup those characters (an equality that is dissolved lies between edits, so
        // inside the middles; the lossless shifts rotate an edit through the neighbouring equality, same characters).  So a `-`
        // instruction needs one of its characters in the input's middle, a `+` instruction in the candidate's (a `$` tail too:
        // its text is an insertion or deletion like any other).  Prefix and suffix are taken on bytes and stop at the first
        // non-ASCII character: shorter than the real ones at worst, which only makes the test weaker.  On BASELINE configs[2]
        // this leaves about half of the rows the presence bits of the whole strings let through (1.9 M -> 1.0 M edit scripts).
        const uint32_t nin = t1 - t0, c0 = a.v_off[id], ncand = a.v_off[id + 1] - c0, nmin = nin < ncand ? nin : ncand;
        uint32_t p = 0;
        while (p < nmin) {
          const uint32_t x = a.text[t0 + p];
          if (x >= 128u || x != a.v_pool[c0 + p]) break;
          ++p;
        }
        uint32_t sfx = 0;
        while (sfx < nmin - p) {
          const uint32_t x = a.text[t1 - 1u - sfx];
          if (x >= 128u || x != a.v_pool[c0 + ncand - 1u - sfx]) break;
          ++sfx;
        }
        uint64_t im0 = 0, im1 = 0, cm0 = 0, cm1 = 0;
        for (uint32_t x = p; x < nin - sfx; ++x) {
          const uint32_t ch = a.text[t0 + x];
          if (ch < 64u) im0 |= 1ull << ch; else if (ch < 128u) im1 |= 1ull << (ch - 64u);
        }
        for (uint32_t x = p; x < ncand - sfx; ++x) {
          const uint32_t ch = a.v_pool[c0 + x];
          if (ch < 64u) cm0 |= 1ull << ch; else if (ch < 128u) cm1 |= 1ull << (ch - 64u);
        }
        need = false;
        for (uint32_t j = 0; j < a.P.nconf && !need; ++j) {
          const cdiff::FlatConf& cf = a.P.conf[j];
          bool may = true;
          for (uint32_t o_ = 0; o_ < cf.nops && may; ++o_) {
            const cdiff::FlatOp& o = a.P.ops[cf.op_begin + o_];
            if (!o.simple) continue;
            uint64_t h0 = o.bits[0], h1 = o.bits[1];
            if (o.op == '-') { h0 &= im0; h1 &= im1; }
            else if (o.op == '+') { h0 &= cm0; h1 &= cm1; }
            else { h0 &= ins.w[0] & cs.w[0]; h1 &= ins.w[1] & cs.w[1]; }
            if (!(h0 | h1)) may = false;
          }
          need = may;
        }
      }
    }
    a.weight[slot] = 1.0;
    if (need) {
      if (k < 64u) needmask |= 1ull << k;
      else a.need[atomicAdd(&a.ctr[0], 1u)] = make_uint2(slot, s);  // (early mode, long candidate lists)
    }
  }
  const uint32_t cnt = (uint32_t)__popcll(needmask), lane = threadIdx.x & 63u;
  uint32_t incl = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
    if ((int)lane >= o) incl += up;
  }
  const uint32_t total = (uint32_t)__shfl((int)incl, 63);
  if (total == 0) return;  // (wave-uniform)
  uint32_t first = 0;
  if (lane == 0) first = atomicAdd(&a.ctr[0], total);
  uint32_t pos = (uint32_t)__shfl((int)first, 0) + incl - cnt;
  for (unsigned long long m = needmask; m; m &= m - 1ull) a.need[pos++] = make_uint2(base + (uint32_t)__ffsll((long long)m) - 1u, s);
}

__global__ void k_conf_iota(uint32_t* p, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = i;
}
// One lane per list entry: the SHAPE of its edit script as a sort key.  k_conf_script runs one row per lane through a branchy
// algorithm; rows whose scripts take different routes serialise (measured: the lanes of a wave ran practically one after the
// other).  What decides the route is what remains of the two strings once their common prefix and suffix are gone -- the lengths
// of the two middles (0/1: a single insertion or deletion; 1/1: a substitution; ...) -- and whether there is a prefix / suffix at
// all (the clean-up passes walk the diff list).  Sorting the list by that key puts rows of one route side by side in the waves.
// Only an ordering hint: computed on the input's BYTES against the candidate's code points, it stops at the first non-ASCII
// character, and nothing but the order in which rows are processed depends on it.
__global__ __launch_bounds__(256) void k_conf_key(ConfArgs a) {
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= a.row_cap) return;
  uint32_t key = 0x1FFFu;  // behind the list: sorts last
  if (k < a.ctr[0]) {
    const uint32_t slot = a.need[k].x, s = a.need[k].y, id = row_item(a, slot);
    const uint32_t i = a.q_orig[s], t0 = a.textoff[i], nin = a.textoff[i + 1] - 1u - t0;
    const uint32_t c0 = a.v_off[id], ncand = a.v_off[id + 1] - c0;
    const uint32_t nmin = nin < ncand ? nin : ncand;
    uint32_t p = 0;
    while (p < nmin) {
      const uint32_t x = a.text[t0 + p];
      if (x >= 128u || x != a.v_pool[c0 + p]) break;
      ++p;
    }
    uint32_t sfx = 0;
    while (sfx < nmin - p) {
      const uint32_t x = a.text[t0 + nin - 1u - sfx];
      if (x >= 128u || x != a.v_pool[c0 + ncand - 1u - sfx]) break;
      ++sfx;
    }
    const uint32_t ma = nin - p - sfx, mb = ncand - p - sfx;
    key = ((ma < 6u ? ma : 6u) << 5) | ((mb < 6u ? mb : 6u) << 2) | (p ? 2u : 0u) | (sfx ? 1u : 0u);  // < 0xFF
    key = key << 5 | (nin < 31u ? nin : 31u);  // then by length: the loops of a route run as long as the strings
  }
  a.key[k] = key;
}

__global__ __launch_bounds__(CF_THREADS) void k_conf_script(ConfArgs a) {
  const uint32_t n = a.ctr[0];
  static_assert(CF_THREADS == 64, "one wave per block: the lane memories of a wave interleave");
  cdiff::Ctx c;
  c.mem = a.work + (size_t)blockIdx.x * CF_WORDS * 64u + threadIdx.x * CF_G;
  c.arena_off = CF_ARENA_OFF; c.arena_cap = CF_ARENA; c.arena_used = 0;
  c.d_off = CF_D_OFF; c.d_cap = CF_DIFFS; c.nd = 0;
  c.v_off = CF_V_OFF; c.v_cap = CF_V;
  c.f_off = CF_F_OFF; c.frame_cap = CF_FRAMES;
  c.alpha = a.alpha; c.nalpha = a.nalpha;
  c.overflow = false;
  for (uint32_t k = blockIdx.x * CF_THREADS + threadIdx.x; k < n; k += gridDim.x * CF_THREADS) {
    const uint2 nd = a.need[a.order[k]];
    const uint32_t slot = nd.x, s = nd.y, id = row_item(a, slot);
    const uint32_t i = a.q_orig[s], t0 = a.textoff[i], t1 = a.textoff[i + 1] - 1u;
    // UTF-8 -> scalar values, as host_model.cpp utf8_decode_at does (invalid bytes decode to themselves, one at a time)
    uint32_t nin = 0;
    bool fits = true;
    for (uint32_t p = t0; p < t1;) {
      const uint32_t b0 = a.text[p];
      uint32_t cp = b0, len = 1;
      const uint32_t avail = t1 - p;
      if (b0 >= 0xC0u && b0 < 0xE0u && avail >= 2 && (a.text[p + 1] & 0xC0u) == 0x80u) { cp = ((b0 & 0x1Fu) << 6) | (a.text[p + 1] & 0x3Fu); len = 2; }
      else if (b0 >= 0xE0u && b0 < 0xF0u && avail >= 3 && (a.text[p + 1] & 0xC0u) == 0x80u && (a.text[p + 2] & 0xC0u) == 0x80u) {
        cp = ((b0 & 0x0Fu) << 12) | ((a.text[p + 1] & 0x3Fu) << 6) | (a.text[p + 2] & 0x3Fu); len = 3;
      } else if (b0 >= 0xF0u && b0 < 0xF8u && avail >= 4 && (a.text[p + 1] & 0xC0u) == 0x80u && (a.text[p + 2] & 0xC0u) == 0x80u && (a.text[p + 3] & 0xC0u) == 0x80u) {
        cp = ((b0 & 0x07u) << 18) | ((a.text[p + 1] & 0x3Fu) << 12) | ((a.text[p + 2] & 0x3Fu) << 6) | (a.text[p + 3] & 0x3Fu); len = 4;
      }
      if (nin >= CF_MAXCP) { fits = false; break; }
      DC::W(c, CF_IN_OFF + nin++) = cp;
      p += len;
    }
    const uint32_t c0 = a.v_off[id], ncand = a.v_off[id + 1] - c0;
    double w = 1.0;
    bool ok = fits && ncand <= CF_MAXCP;
    if (ok) {
      for (uint32_t x = 0; x < ncand; ++x) DC::W(c, CF_CAND_OFF + x) = a.v_pool[c0 + x];  // the candidate joins the lane memory
      const cdiff::View in = cdiff::mk(CF_IN_OFF, nin), cand = cdiff::mk(CF_CAND_OFF, ncand);
      ok = DC::confusable_weight(c, a.P, in, DC::charset_of(c, in), cand, a.v_cs[id], &w);
    }
    if (!ok) atomicAdd(&a.ctr[1], 1u);
    a.weight[slot] = w;
  }
}

// early mode: the candidate rows' scores times their weights (k_rank then ranks, crops and cuts off as always)
__global__ __launch_bounds__(256) void k_conf_apply_early(ConfArgs a) {
  const uint32_t slot = blockIdx.x * 256 + threadIdx.x;
  if (slot >= a.soff[a.nq] || a.soff[a.nq] > a.row_cap) return;
  const double w = a.weight[slot];
  if (w != 1.0) a.c_rows[slot].score *= w;
}

__device__ inline double conf_vr_score(const DevRow& r, float fw) {  // src/types.rs:335-341
  if (fw == 0.0f) return r.dist_score;
  return (r.dist_score + ((double)fw * r.freq_score)) / (1.0 + (double)fw);
}
__device__ inline bool conf_before(const DevRow& x, const DevRow& y, float fw) {  // rank_cmp, src/types.rs:344-365: x strictly before y
  if (fw > 0.0f) return conf_vr_score(x, fw) > conf_vr_score(y, fw);
  if (x.dist_score != y.dist_score) return x.dist_score > y.dist_score;
  return x.freq_score > y.freq_score;
}
// late mode: one lane per query: weight the ranked rows, stable re-sort, cutoff (src/lib.rs:1591-1622), new count
__global__ __launch_bounds__(256) void k_conf_apply_late(ConfArgs a, uint32_t* __restrict__ r_count) {
  const uint32_t s = blockIdx.x * 256 + threadIdx.x;
  if (s >= a.nq || a.soff[a.nq] > a.row_cap) return;
  const uint32_t n = r_count[s], base = a.soff[s];
  if (n == 0) return;
  DevRow* __restrict__ v = a.r_rows + base;
  bool any = false;
  for (uint32_t k = 0; k < n; ++k) {
    const double w = a.weight[base + k];
    if (w != 1.0) { v[k].dist_score *= w; any = true; }
  }
  if (any)  // stable insertion sort (the lists hold max_matches + 1 rows)
    for (uint32_t k = 1; k < n; ++k) {
      const DevRow x = v[k];
      uint32_t j = k;
      while (j > 0 && conf_before(x, v[j - 1], a.freq_weight)) { v[j] = v[j - 1]; --j; }
      v[j] = x;
    }
  uint32_t len = n;
  if (a.cutoff_threshold >= 1.0) {
    const double best = conf_vr_score(v[0], a.freq_weight);
    for (uint32_t k = 1; k < n; ++k)
      if (conf_vr_score(v[k], a.freq_weight) <= best / a.cutoff_threshold) { len = k; break; }
  }
  r_count[s] = len;
}

// ---- host driver ----------------------------------------------------------------------------------------------------------
namespace {
std::mutex g_conf_mu;
template <typename T>
int cf_upload(T** dst, const void* src, size_t count, std::string& err) {
  if (*dst) { pool_free(*dst); *dst = nullptr; }
  HIP_TRY(pool_malloc(reinterpret_cast<void**>(dst), std::max<size_t>(count * sizeof(T), 16)));
  if (count) HIP_TRY(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
  return ANX_OK;
}
}  // namespace

void conf_free(DeviceConf* dc) {
  if (!dc) return;
  (void)hipSetDevice(dc->device);
  for (void* p : {(void*)dc->conf, (void*)dc->ops, (void*)dc->opts, (void*)dc->pool, (void*)dc->v_pool, (void*)dc->v_off, (void*)dc->v_cs, (void*)dc->alpha})
    if (p) pool_free(p);
  delete dc;
}

// The replica's copy of the pattern tables and of the vocabulary (code points + presence bits): built on first use, rebuilt when
// patterns or vocabulary items were added since.
static int conf_ensure(const HostModel& m, const DeviceLexicon* dl, std::string& err) {
  std::lock_guard<std::mutex> g(g_conf_mu);
  DeviceConf*& dc = dl->dconf;
  if (!dc) { dc = new DeviceConf(); dc->device = dl->device; }
  int rc;
  if (dc->built_patterns != m.confusables.size() || !dc->conf) {
    const HostModel::ConfTables& t = m.conf_tables;
    if ((rc = cf_upload(&dc->conf, t.conf.data(), t.conf.size(), err)) || (rc = cf_upload(&dc->ops, t.ops.data(), t.ops.size(), err)) ||
        (rc = cf_upload(&dc->opts, t.opts.data(), t.opts.size(), err)) || (rc = cf_upload(&dc->pool, t.pool.data(), t.pool.size(), err)))
      return rc;
    dc->nconf = (uint32_t)t.conf.size();
    dc->built_patterns = m.confusables.size();
  }
  if (dc->built_vocab != m.decoder.size() || !dc->v_cs) {
    const uint32_t *pool, *off;
    const void* cs;
    size_t npool, n;
    m.conf_vocab_arrays(&pool, &npool, &off, &cs, &n);
    if ((rc = cf_upload(&dc->v_pool, pool, npool, err)) || (rc = cf_upload(&dc->v_off, off, n + 1, err)) ||
        (rc = cf_upload(&dc->v_cs, cs, n, err)))
      return rc;
    dc->nvocab = (uint32_t)n;
    dc->built_vocab = m.decoder.size();
  }
  if (!dc->alpha) {
    uint32_t na = 0;
    const uint32_t(*al)[2] = alphabetic_ranges(&na);
    uint32_t* p = nullptr;
    if ((rc = cf_upload(&p, al, (size_t)na * 2, err))) return rc;
    dc->alpha = reinterpret_cast<uint32_t(*)[2]>(p);
    dc->nalpha = na;
  }
  return ANX_OK;
}

// Enqueues the weighting of the batch's rows on `st`.  early: before k_rank, on the candidate rows; late: after k_rank, on the
// ranked rows, followed by the re-rank + cutoff.  The working memory of k_conf_script belongs to the batch.
int conf_launch(const HostModel& m, const DeviceLexicon* dl, Batch* b, hipStream_t st, bool early, uint32_t row_cap, std::string& err) {
  int rc = conf_ensure(m, dl, err);
  if (rc) return rc;
  b->conf_skipped = false;
  const DeviceConf* dc = dl->dconf;
  const uint32_t nq = (uint32_t)b->nq;
  if (!b->cf_weight || b->cf_cap < row_cap) {
    for (void* p : {(void*)b->cf_weight, (void*)b->cf_need, (void*)b->cf_sort, b->cf_sort_tmp})
      if (p) pool_free(p);
    b->cf_weight = nullptr; b->cf_need = nullptr; b->cf_sort = nullptr; b->cf_sort_tmp = nullptr;
    HIP_TRY(pool_malloc(reinterpret_cast<void**>(&b->cf_weight), std::max<size_t>((size_t)row_cap * sizeof(double), 16)));
    HIP_TRY(pool_malloc(reinterpret_cast<void**>(&b->cf_need), std::max<size_t>((size_t)row_cap * sizeof(uint2), 16)));
    b->cf_cap = row_cap;
  }
  if (!b->cf_sort) {  // (freed with cf_weight above when the capacity grows)
    HIP_TRY(pool_malloc(reinterpret_cast<void**>(&b->cf_sort), std::max<size_t>(4 * (size_t)row_cap * sizeof(uint32_t), 16)));
    size_t bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, bytes, b->cf_sort, b->cf_sort, b->cf_sort, b->cf_sort, (size_t)row_cap, 0u, 13u, st));
    HIP_TRY(pool_malloc(&b->cf_sort_tmp, std::max<size_t>(bytes, 16)));
    b->cf_sort_tmp_bytes = bytes;
  }
  uint32_t* sort_buf = b->cf_sort;
  if (!b->cf_ctr) HIP_TRY(pool_malloc(reinterpret_cast<void**>(&b->cf_ctr), 16));
  // one wave per 64 rows that may need a script, at most CF_BLOCKS of them in flight: the working set (5.8 KB per lane) follows the
  // batch's rows, so a one-line query or a small n-gram order of a search call does not claim the 1.5 GB a million queries use
  const uint32_t cf_blocks = std::min<uint32_t>(CF_BLOCKS, std::max<uint32_t>(1u, (row_cap + 63u) / 64u));
  if (b->cf_work && b->cf_work_blocks < cf_blocks) { pool_free(b->cf_work); b->cf_work = nullptr; }
  if (!b->cf_work) {
    if (pool_malloc(&b->cf_work, (size_t)cf_blocks * CF_WORDS * 64u * sizeof(uint32_t)) != hipSuccess) {
      b->cf_work = nullptr;
      (void)hipGetLastError();
      b->conf_skipped = true;  // no room for the working set: the batch is redone with the host-side weighting (same results)
      return ANX_OK;
    }
    b->cf_work_blocks = cf_blocks;
  }
  HIP_TRY(hipMemsetAsync(b->cf_ctr, 0, 16, st));
  ConfArgs a;
  a.nq = nq; a.row_cap = row_cap; a.early = early ? 1 : 0; a.soff = b->soff; a.r_count = b->r_count; a.c_rows = b->c_rows; a.r_rows = b->r_rows;
  a.q_orig = b->q_orig; a.text = b->d_text; a.textoff = b->d_textoff; a.weight = b->cf_weight; a.need = reinterpret_cast<uint2*>(b->cf_need); a.ctr = b->cf_ctr;
  a.P.conf = dc->conf; a.P.nconf = dc->nconf; a.P.ops = dc->ops; a.P.opts = dc->opts; a.P.pool = dc->pool;
  a.v_pool = dc->v_pool; a.v_off = dc->v_off; a.v_cs = dc->v_cs; a.nvocab = dc->nvocab;
  a.alpha = dc->alpha; a.nalpha = dc->nalpha; a.work = static_cast<uint32_t*>(b->cf_work);
  a.key = sort_buf; a.order = sort_buf + 3 * (size_t)row_cap;
  a.cutoff_threshold = b->params.cutoff_threshold; a.freq_weight = b->params.freq_weight;
  // the row slots: every candidate row (early) / every slot of the ranked segments (late); the kernels bound-check against soff[nq]
  const uint32_t nblk = (row_cap + 255u) / 256u;
  hipLaunchKernelGGL(k_conf_screen, dim3((nq + 255u) / 256u), dim3(256), 0, st, a);
  if (nblk) {
    // the list in the order of its rows' script shapes: 8-bit keys, one pass of the device radix sort over the list's capacity
    // (entries behind the list carry the largest key; the list length stays on the device)
    hipLaunchKernelGGL(k_conf_key, dim3(nblk), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_conf_iota, dim3(nblk), dim3(256), 0, st, sort_buf + (size_t)row_cap, row_cap);
    HIP_TRY(rocprim::radix_sort_pairs(b->cf_sort_tmp, b->cf_sort_tmp_bytes, sort_buf, sort_buf + 2 * (size_t)row_cap, sort_buf + (size_t)row_cap,
                                      sort_buf + 3 * (size_t)row_cap, (size_t)row_cap, 0u, 13u, st));
  }
  const int kt = ktimer_begin("k_conf_script", st);
  hipLaunchKernelGGL(k_conf_script, dim3(b->cf_work_blocks), dim3(CF_THREADS), 0, st, a);
  ktimer_end(kt, st);
  if (early) { if (nblk) hipLaunchKernelGGL(k_conf_apply_early, dim3(nblk), dim3(256), 0, st, a); }
  else hipLaunchKernelGGL(k_conf_apply_late, dim3((nq + 255u) / 256u), dim3(256), 0, st, a, b->r_count);
  HIP_TRY(hipGetLastError());
  return ANX_OK;
}

}  // namespace anx
