// engine.hip -- the HIP (gfx950 / CDNA4) variant-query pipeline of the anx engine.
//
// Replaces, for a whole batch of queries at once, the reference's
//   find_nearest_anahashes  (/root/reference/src/lib.rs:1143-1308)  -> k_scan_bits / k_scan_sad
//   gather_instances        (src/lib.rs:1311-1402, src/distance.rs)  -> k_score_pairs
//   score_and_rank          (src/lib.rs:1405-1653, src/types.rs:334-365) -> k_score_pairs + k_compact + k_rank
// Integer work only: no MFMA.  Wave = 64 lanes everywhere.  See DESIGN.md for layout and rooflines.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>
#include <numeric>
#include <thread>
#include <type_traits>

#include "engine.h"

namespace anx {

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      err = std::string(#expr) + ": " + hipGetErrorString(_e);                                 \
      return ANX_ENODEVICE;                                                                    \
    }                                                                                          \
  } while (0)

// ------------------------------------------------------------------------------------------------
// Device structures
// ------------------------------------------------------------------------------------------------
struct Tile {           // <= SCAN_TQ queries of one scan kind, one length and one signature
  uint32_t q0, nq;      // query range (queries are sorted by (scan kind, length, signature))
  uint32_t s0, s1;      // signature range [s0, s1) of the +-k charcount window
  uint32_t k;           // clamped anagram distance for this length
  uint32_t lq;          // query length in symbols
  uint32_t sig_lo, sig_hi;  // the tile's signature (per-group symbol counts, one byte each)
  uint32_t kind;        // 0 = SAD body, 1..NBITPLANES = bit-plane body with T = kind
};

constexpr int NBITPLANES = 4;            // thermometer planes stored per class / query
constexpr uint32_t SCAN_TQ = 64;         // queries per tile (= per wave), compared in passes of 32 (one hit-mask bit each)
constexpr uint32_t SCAN_CHUNK = 256;     // pair slots a wave reserves per global atomic
constexpr uint32_t SCAN_REGIONS = 64;     // pair-list regions with one reservation counter each
constexpr uint32_t RC_STRIDE = 32;        // uint32 words per region counter block (128 B)
constexpr uint32_t RAW_INVALID = 0xFFFFFFFFu;
constexpr uint32_t META_SKIPPED = 0xFFFFFFFFu;

struct EntRec {   // per-entry attributes k_compact needs, one 16-B gather
  uint32_t vocab, freq, order, meta;
};

struct DeviceLexicon {
  int device = 0;
  int nplanes = 0;      // count-vector dwords (SAD path)
  int nsym = 0;
  uint32_t nclasses = 0, nentries = 0, cstride = 0, max_len = 0;
  uint32_t* cls_planes = nullptr;  // [nplanes][cstride] packed u8 counts
  uint32_t* cls_bits = nullptr;    // [NBITPLANES][cstride] thermometer planes (bit s of plane t: count_s > t), nsym <= 32
  uint8_t* cls_len = nullptr;      // [cstride]
  uint32_t* cls_off = nullptr;
  uint2* sig = nullptr;            // [nsig_pad] signature table (see LexiconImage), lo/hi interleaved
  uint32_t* sig_cbeg = nullptr;    // [nsig_pad+1]
  uint32_t* ent_vocab = nullptr;
  uint32_t* ent_freq = nullptr;
  uint32_t* ent_meta = nullptr;
  uint32_t* ent_rowoff = nullptr;
  uint32_t* ent_order = nullptr;
  EntRec* ent_rec = nullptr;           // {vocab, freq, order, meta} per entry
  uint32_t* ent_var_off = nullptr;     // CSR entry -> VariantOf references (variant lists, src/lib.rs:1677-1727)
  uint32_t* var_target = nullptr;      // vocab id of the reference item
  uint32_t* var_target_freq = nullptr;
  double* var_score = nullptr;
  int any_variants = 0;
  uint4* rows = nullptr;
  size_t bytes = 0;
};

enum { CTR_SKIPPED = 2, CTR_N = 8 };

struct SurvRow {  // one candidate result row of a query (k_compact -> k_rank); 32 B, written / read as two 16-B words
  double score;            // dist_score (times the variant score for expanded rows)
  unsigned long long ord;  // enumeration-order key: ent_order << 20 | position inside the expansion
  uint32_t vocab, freq;    // vocab id, absolute frequency of the row
  uint32_t via, pad;       // vocab id of the variant the row was reached through, 0xFFFFFFFF = none
};
struct SurvRec {   // one scored pair that passed the score threshold (k_score_* -> k_compact), appended per wave
  uint32_t q, e;
  double score;
};
struct DevRow {   // one ranked result row (device) for download / gather
  uint32_t vocab_id, via;
  double dist_score, freq_score;
};

struct Batch {
  int device = 0;
  size_t nq = 0;            // encoded queries
  anx_params params;
  // host side
  std::vector<uint32_t> order;     // sorted position -> original index
  std::vector<int32_t> status;     // per original query: 0 ok, ANX_EEMPTY, ANX_ELIMIT
  size_t n_input = 0;
  std::vector<Tile> tiles;         // in launch order: bit-plane kinds, then the SAD kind; each by decreasing cost
  uint32_t n_sad_tiles = 0;
  uint32_t qw = 1;                 // uint4 words per query row
  uint32_t dmax = 0;
  uint64_t n_class_tests = 0;
  uint64_t n_tests_kind[NBITPLANES + 1] = {};
  // device: queries
  uint32_t* q_cv = nullptr;        // [nq][nplanes]
  uint32_t* q_bits = nullptr;      // [nq][NBITPLANES]
  uint4* q_rows = nullptr;         // [nq][qw]
  uint32_t* q_meta = nullptr;      // len | k<<8 | d<<16 | first_is_lower<<24
  uint32_t* q_orig = nullptr;      // original index
  Tile* d_tiles = nullptr;
  // device: pipeline
  uint32_t* counters = nullptr;
  uint32_t* rctr = nullptr;        // [SCAN_REGIONS][RC_STRIDE] per-region reservation / statistics counters
  uint32_t region_shift = 0;       // log2(slots per region); raw_cap = SCAN_REGIONS << region_shift
  uint32_t region_fill[SCAN_REGIONS] = {};  // host copy of rctr[r][RC_RAW] after the last run
  uint32_t* qexact = nullptr;      // per query: its exact-anagram class, 0xFFFFFFFF = none (StopAtExactMatch; host lookup)
  uint32_t* qsurv = nullptr;       // per query: pairs with score >= threshold
  uint32_t* soff = nullptr;        // nq+1, exclusive scan of qsurv
  uint32_t* qcur = nullptr;
  uint32_t* qmaxfreq = nullptr;
  uint32_t* scan_tmp = nullptr;
  uint2* raw = nullptr;            // flat pair list (query, entry | exact<<31), in wave chunks
  double* p_score = nullptr;       // per pair-list slot: score of the pairs that went through a DL kernel
  uint32_t* p_meta = nullptr;      // per pair-list slot: skipped / rejected / ld | samecase<<7 | lcs<<8 | prefix<<16 | suffix<<24
  uint32_t* list8 = nullptr;       // slot lists of the selected pairs the fused kernel leaves to k_score_fast8 / k_score_pairs
  uint32_t* listg = nullptr;
  uint32_t* lctr = nullptr;        // [2][SCAN_REGIONS][RC_STRIDE] their fills
  size_t list_cap = 0;             // slots per region in list8 / listg
  size_t raw_cap = 0;
  double* quot = nullptr;          // table of IEEE quotients x / L (ScoreArgs::quot)
  SurvRec* surv = nullptr;         // survivor records in SCAN_REGIONS regions of surv_region_cap (order arbitrary)
  uint32_t* sctr = nullptr;        // [SCAN_REGIONS][RC_STRIDE] fill of every survivor region
  size_t surv_region_cap = 0;
  SurvRow* c_rows = nullptr;       // candidate result rows grouped by query (survivors, expanded by variant lists)
  uint32_t* qexpand = nullptr;     // per query: some DL survivor has variant references (has_expandable_variants)
  DevRow* r_rows = nullptr;        // ranked rows, per query at soff[q] .. soff[q] + r_count[q]
  double* t_key = nullptr;
  size_t surv_cap = 0;
  uint32_t* r_count = nullptr;
  uint32_t* r_off = nullptr;       // nq+1
  uint32_t n_raw = 0;
  uint64_t n_sel = 0;
  uint64_t n_pairs = 0, n_surv = 0, n_results = 0;
  bool ran = false;
  hipEvent_t ev[6] = {};
  hipEvent_t ev_scan0 = nullptr;   // just before the scan kernels (after the counter memsets)
  anx_batch_stats stats = {};
};

typedef const __attribute__((address_space(4))) uint32_t* cptr_u32;  // constant address space: s_load

// ------------------------------------------------------------------------------------------------
// K1: signature-pruned anagram scan.
//   Spec: the set returned by find_nearest_anahashes (src/lib.rs:1143-1308) equals
//     { class c : L1(cv_q, cv_c) <= k, |len_c - len_q| <= k, cv_q and cv_c share a symbol }
//   (the bigint `cand % av == 0` containment test of src/anahash.rs:165-171 is
//   multiset inclusion, i.e. a statement about the prime-exponent = count vectors).
//   Pruning: sig(x) = per-group sums of the count vector (LexiconImage::sym_group); summing is a contraction of
//   L1, so L1(sig_q, sig_c) > k excludes c.  Queries are sorted by (kind, length, signature) and a tile holds
//   <= 32 queries of ONE signature; classes are stored in (charcount, signature) order, one run per signature.
//   One WAVE owns one tile: it tests the tile's signature against the signature table of the +-k charcount
//   window (64 signatures per step, 2 v_sad_u8 each), copies the class ids of the compatible runs to an LDS stage
//   and, whenever 64*CPL classes are staged, compares them (lane = class, gathered planes in registers) with
//   every query of the tile (query planes broadcast from LDS).  On eng.aspell k<=3 this leaves 4.6 k of the
//   68 k class tests per query that the plain charcount window needs.
//   The query loop is branch-free: every lane keeps one hit bit per (class, query) in registers; after the loop
//   the hits of the chunk (1-2 % of the remaining tests) are expanded through the class -> entries CSR into
//   wave-private 256-slot chunks of the pair list.  The pair list is split into SCAN_REGIONS regions with one
//   reservation counter each (128 B apart): a single contended counter word sustains only ~88 M atomics/s, which
//   at ~1.5 ms per million queries would be the bottleneck.
// ------------------------------------------------------------------------------------------------
enum { RC_RAW = 0, RC_VALID = 1, RC_TESTS = 4 /* u64 per scan kind at 4 + 2*kind */ };

struct WaveOut {
  uint32_t base, left;  // unused part of the current chunk of the pair list (wave-uniform)
  uint32_t emitted;     // pairs appended by this wave (wave-uniform)
  uint32_t nbase;       // chunk reserved by the last wave_reserve when the appended run spills over
  uint32_t split;       // run indices < split go to [base..), the rest to [nbase..)
  uint32_t rbase, rend; // this wave's region of the pair list: slots [rbase, rend)
  uint32_t* ctr;        // the region's counter block
};
// Wave-wide exclusive prefix sum of ntot + chunk reservation.  Returns this lane's first index g in the
// wave's appended run; wave_slot(g) maps run indices to pair-list slots.  A run that does not fit in the
// rest of the current chunk fills it up and continues in a freshly reserved chunk (ONE atomic).
__device__ inline uint32_t wave_reserve(WaveOut& w, uint32_t ntot, uint32_t lane, uint32_t* total_out) {
  uint32_t incl = ntot;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t u = __shfl_up(incl, o);
    if (lane >= (uint32_t)o) incl += u;
  }
  const uint32_t total = __shfl(incl, 63);
  *total_out = total;
  w.split = w.left;
  if (total > w.left) {
    const uint32_t rest = total - w.left;
    const uint32_t need = rest > SCAN_CHUNK ? rest : SCAN_CHUNK;
    uint32_t b = 0;
    if (lane == 0) b = atomicAdd(&w.ctr[RC_RAW], need);
    w.nbase = w.rbase + __shfl(b, 0);
  }
  return incl - ntot;
}
__device__ inline uint32_t wave_slot(const WaveOut& w, uint32_t g) {
  return g < w.split ? w.base + g : w.nbase + (g - w.split);
}
__device__ inline void wave_commit(WaveOut& w, uint32_t total) {
  if (total > w.left) {
    const uint32_t rest = total - w.left;
    const uint32_t need = rest > SCAN_CHUNK ? rest : SCAN_CHUNK;
    w.base = w.nbase + rest;
    w.left = need - rest;
  } else {
    w.base += total;
    w.left -= total;
  }
  w.emitted += total;
}
__device__ inline void wave_close(const WaveOut& w, uint32_t lane, uint2* __restrict__ raw) {
  for (uint32_t i = lane; i < w.left; i += 64)
    if (w.base + i < w.rend) raw[w.base + i] = make_uint2(RAW_INVALID, 0u);
  if (lane == 0 && w.emitted) atomicAdd(&w.ctr[RC_VALID], w.emitted);  // one atomic per wave
}

struct ScanArgs {
  const Tile* tiles;
  uint32_t ntiles;
  const uint32_t* q_bits;
  const uint32_t* q_cv;
  const uint32_t* cls_bits;
  const uint32_t* cls_planes;
  uint32_t cstride;
  uint32_t pad_class;   // a never-matching padding class (bits 0, counts 0xFF, len 255)
  const uint8_t* cls_len;
  const uint32_t* cls_off;
  const uint2* sig;         // signature table: (groups 0-3, groups 4-7) packed as bytes
  const uint32_t* sig_cbeg;
  uint2* raw;
  uint32_t region_cap;  // pair-list slots per region
  uint32_t* rctr;       // [SCAN_REGIONS][RC_STRIDE]
  const uint32_t* qexact;  // per query: class id of its exact anagram class (0xFFFFFFFF = none); stop mode only
  int want_exact;
  int dbg;  // ANX_SCAN_DBG (timing experiments only; results are wrong when set): 1 skip the query loop, 2 skip process(), 4 skip the expansion
};

__device__ inline int32_t bcnt_acc(uint32_t x, int32_t acc) {  // acc + popcount(x) in one v_bcnt_u32_b32
  int32_t r;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
  return r;
}

// T >= 1: thermometer bit planes.  common(q,c) = sum_t popc(Q_t & C_t) is exact when every symbol of the query
//   occurs at most T times (min(a,b) only needs a's planes; class planes saturate at NBITPLANES).
//   L1 = len_q + len_c - 2 common, so  hit <=> common >= max(1, ceil((len_q + len_c - k) / 2))   (>= 1: the
//   classes share a symbol, src/iterators.rs:177).  2 ops per plane: v_and_b32 + accumulating v_bcnt_u32_b32.
// T == 0: general path (any alphabet size / multiplicity): packed u8 count vectors, NP x v_sad_u8;
//   hit <=> L1 <= k and L1 < len_q + len_c.
template <int T, int NP>
__device__ inline void scan_tile(const ScanArgs& A, const Tile& t, uint32_t item, uint32_t* __restrict__ stage,
                                 uint32_t* __restrict__ qlds) {
  constexpr bool BITS = T > 0;
  constexpr int CPL = BITS ? 4 : (NP <= 8 ? 4 : NP <= 16 ? 2 : 1);  // classes per lane
  constexpr int W = BITS ? T : NP;                                   // dwords compared per class
  constexpr int QSTRIDE = BITS ? NBITPLANES : NP;
  constexpr uint32_t CHUNK = 64u * CPL;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t* __restrict__ cls_words = BITS ? A.cls_bits : A.cls_planes;
  const uint8_t* __restrict__ cls_len = A.cls_len;
  const uint32_t* __restrict__ cls_off = A.cls_off;
  uint2* __restrict__ raw = A.raw;
  const uint32_t cstride = A.cstride;
  const uint32_t region = item % SCAN_REGIONS;
  WaveOut wo{0, 0, 0, 0, 0, region * A.region_cap, (region + 1) * A.region_cap, A.rctr + region * RC_STRIDE};
  uint32_t ns = 0;  // staged class ids (wave-uniform)
  uint32_t nchunks = 0;
  {  // the tile's query words -> LDS: the comparison loop reads them back as broadcasts into VGPRs
    const uint32_t* __restrict__ src = (BITS ? A.q_bits : A.q_cv) + (size_t)t.q0 * QSTRIDE;
    for (uint32_t i = lane; i < t.nq * QSTRIDE; i += 64) qlds[i] = src[i];
  }

  // compares the first CHUNK staged classes (padded with the never-matching class) with every query of the tile, 32
  // queries per pass; bit (npass-1-qi) of hm[j] = query qi of the pass hits class j of this lane.  The query loop is
  // branch-free: 2 ops per plane + ONE v_alignbit_b32 per test (it shifts the sign bit of acc = "miss" into the mask).
  auto process = [&]() {
    ++nchunks;
    if (A.dbg & 2) return;
    uint32_t cid[CPL], cw[CPL][W];
    int32_t thr[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const uint32_t idx = (uint32_t)j * 64u + lane;
      cid[j] = idx < ns ? stage[idx] : A.pad_class;
#pragma unroll
      for (int p = 0; p < W; ++p) cw[j][p] = cls_words[(size_t)p * cstride + cid[j]];
      const int32_t lc = (int32_t)cls_len[cid[j]];
      if (BITS) {
        const int32_t need = ((int32_t)t.lq - (int32_t)t.k + lc + 1) >> 1;  // ceil((lq + lc - k) / 2)
        thr[j] = -(need < 1 ? 1 : need);
      } else {
        const int32_t share = (int32_t)t.lq + lc - 1;  // L1 < lq + lc: shares a symbol (src/iterators.rs:177, src/lib.rs:1205)
        thr[j] = share < (int32_t)t.k ? share : (int32_t)t.k;
      }
    }
    for (uint32_t qb = 0; qb < t.nq; qb += 32) {
      const uint32_t npass = (A.dbg & 1) ? 1u : (t.nq - qb < 32u ? t.nq - qb : 32u);
      uint32_t hm[CPL];
#pragma unroll
      for (int j = 0; j < CPL; ++j) hm[j] = 0xFFFFFFFFu;  // miss bits
      for (uint32_t qi = 0; qi < npass; ++qi) {
        uint32_t qreg[W];
#pragma unroll
        for (int p = 0; p < W; ++p) qreg[p] = qlds[(qb + qi) * QSTRIDE + p];
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          int32_t acc;
          if (BITS) {
            acc = thr[j];  // common - threshold: negative = miss
#pragma unroll
            for (int p = 0; p < W; ++p) acc = bcnt_acc(qreg[p] & cw[j][p], acc);
          } else {
            uint32_t sad = 0;
#pragma unroll
            for (int p = 0; p < W; ++p) sad = __builtin_amdgcn_sad_u8(qreg[p], cw[j][p], sad);
            acc = thr[j] - (int32_t)sad;  // threshold - L1: negative = miss
          }
          hm[j] = __builtin_amdgcn_alignbit(hm[j], (uint32_t)acc, 31);  // (hm << 1) | sign(acc)
        }
      }
      // expand the hits of this pass into (query, entry) pairs
      const uint32_t valid = npass >= 32u ? 0xFFFFFFFFu : ((1u << npass) - 1u);
      uint32_t any = 0;
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        hm[j] = ~hm[j] & valid;
        any |= hm[j];
      }
      if (__ballot(any != 0) == 0ull) continue;  // wave-uniform
      uint32_t e0[CPL], ne[CPL], cnt = 0;
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        e0[j] = 0;
        ne[j] = 0;
        if (hm[j]) {
          e0[j] = cls_off[cid[j]];
          ne[j] = cls_off[cid[j] + 1] - e0[j];
          cnt += (uint32_t)__popc(hm[j]) * ne[j];
        }
      }
      uint32_t total;
      uint32_t g = wave_reserve(wo, cnt, lane, &total);
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        uint32_t m = hm[j];
        while (m) {
          const uint32_t bit = 31u - (uint32_t)__clz((int)m);
          m &= ~(1u << bit);
          const uint32_t q = t.q0 + qb + (npass - 1u - bit);
          // the exact anagram class (StopAtExactMatch, src/lib.rs:1164-1173)
          const uint32_t exact = (A.want_exact && A.qexact[q] == cid[j]) ? 0x80000000u : 0u;
          for (uint32_t i = 0; i < ne[j]; ++i, ++g) {
            const uint32_t pos = wave_slot(wo, g);
            if (pos < wo.rend) raw[pos] = make_uint2(q, (e0[j] + i) | exact);
          }
        }
      }
      wave_commit(wo, total);
    }
  };

  // No bounds test on s: signatures outside [s0, s1) belong to other charcounts, so their L1 distance to the tile's
  // signature is at least the length difference > k, and the table is padded with never-matching entries.
  const uint2* __restrict__ sigp = A.sig + t.s0 + lane;
  for (uint32_t sb = t.s0; sb < t.s1; sb += 64, sigp += 64) {
    const uint32_t s = sb + lane;
    const uint2 sg = *sigp;
    const bool ok = __builtin_amdgcn_sad_u8(sg.x, t.sig_lo, __builtin_amdgcn_sad_u8(sg.y, t.sig_hi, 0u)) <= t.k;
    unsigned long long m = __ballot(ok);
    if (!m || (A.dbg & 4)) continue;
    uint32_t cb = 0, n = 0;
    if (ok) {
      cb = A.sig_cbeg[s];
      n = A.sig_cbeg[s + 1] - cb;
    }
    while (m) {  // scalar loop over the compatible signatures of this step
      const int i = __ffsll((long long)m) - 1;
      m &= m - 1;
      uint32_t cbi = (uint32_t)__builtin_amdgcn_readlane((int)cb, i), ni = (uint32_t)__builtin_amdgcn_readlane((int)n, i);
      while (ni) {
        const uint32_t take = ni < 64u ? ni : 64u;  // ns < CHUNK here, the stage holds CHUNK + 128 ids
        stage[ns + lane] = cbi + lane;              // all 64 lanes write; only the first `take` ids count
        ns += take;
        cbi += take;
        ni -= take;
        if (ns >= CHUNK) {
          process();
          const uint32_t rem = ns - CHUNK;
          uint32_t v = 0;
          if (lane < rem) v = stage[CHUNK + lane];
          if (lane < rem) stage[lane] = v;
          ns = rem;
        }
      }
    }
  }
  if (ns) process();
  wave_close(wo, lane, raw);
  if (lane == 0 && nchunks)
    atomicAdd(reinterpret_cast<unsigned long long*>(wo.ctr + RC_TESTS + 2 * T), (unsigned long long)nchunks * CHUNK * t.nq);
}

// Every wave takes one tile; tiles are ordered by decreasing cost.  The bit-plane tiles (wave-uniform switch over
// T) and the count-vector tiles run as two launches so that the rarely used wide SAD body does not set the register
// budget (= occupancy) of the common one.
constexpr uint32_t SCAN_STAGE = 64 * 4 + 128;
template <int NP, bool BITS>
__device__ inline void scan_wave(const ScanArgs& A) {
  constexpr int QWORDS = SCAN_TQ * (BITS ? NBITPLANES : NP);
  __shared__ uint32_t s_qlds[4][QWORDS];
  __shared__ uint32_t s_stage[4][SCAN_STAGE];
  const uint32_t wid = threadIdx.x >> 6;
  const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + wid));
  if (item >= A.ntiles) return;
  const cptr_u32 tp = (cptr_u32)(A.tiles + item);
  Tile t;
  t.q0 = tp[0]; t.nq = tp[1]; t.s0 = tp[2]; t.s1 = tp[3]; t.k = tp[4]; t.lq = tp[5]; t.sig_lo = tp[6]; t.sig_hi = tp[7]; t.kind = tp[8];
  uint32_t* qlds = s_qlds[wid];
  uint32_t* stage = s_stage[wid];
  if (BITS) {
    switch (t.kind) {
      case 1: scan_tile<1, NP>(A, t, item, stage, qlds); break;
      case 2: scan_tile<2, NP>(A, t, item, stage, qlds); break;
      case 3: scan_tile<3, NP>(A, t, item, stage, qlds); break;
      default: scan_tile<4, NP>(A, t, item, stage, qlds); break;
    }
  } else {
    scan_tile<0, NP>(A, t, item, stage, qlds);
  }
}
// <= 80 VGPRs = 6 waves per SIMD for the bit-plane kernel (measured: unconstrained 85 VGPRs -> 2.33 ms, 80 -> 2.20 ms,
// 64 with spills -> 2.60 ms)
template <int NP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_scan_bits(ScanArgs A) { scan_wave<NP, true>(A); }
template <int NP>
__global__ __launch_bounds__(256) void k_scan_sad(ScanArgs A) { scan_wave<NP, false>(A); }

// ------------------------------------------------------------------------------------------------
// Exclusive prefix sum (u32), three small kernels.  out has n+1 entries.
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 8, SCAN_THREADS = 256, SCAN_TILE = SCAN_ITEMS * SCAN_THREADS;

__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t* total) {
  __shared__ uint32_t wsum[SCAN_THREADS / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t u = __shfl_up(inc, o);
    if (lane >= o) inc += u;
  }
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int i = 0; i < SCAN_THREADS / 64; ++i) {
    if (i < wid) base += wsum[i];
    tot += wsum[i];
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_local(const uint32_t* __restrict__ in, uint32_t n,
                                                             uint32_t* __restrict__ out,
                                                             uint32_t* __restrict__ blocksum) {
  const uint32_t i0 = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS], s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    v[i] = i0 + i < n ? in[i0 + i] : 0;
    s += v[i];
  }
  uint32_t tot;
  uint32_t ex = block_exclusive_scan(s, &tot);
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    if (i0 + i < n) out[i0 + i] = ex;
    ex += v[i];
  }
  if (threadIdx.x == 0) blocksum[blockIdx.x] = tot;
}
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_sums(uint32_t* __restrict__ blocksum, uint32_t nb) {
  uint32_t carry = 0;
  for (uint32_t b0 = 0; b0 < nb; b0 += SCAN_THREADS) {
    const uint32_t i = b0 + threadIdx.x;
    const uint32_t v = i < nb ? blocksum[i] : 0;
    uint32_t tot;
    const uint32_t ex = block_exclusive_scan(v, &tot);
    if (i < nb) blocksum[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) blocksum[nb] = carry;
}
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_add(uint32_t* __restrict__ out, uint32_t n,
                                                           const uint32_t* __restrict__ blocksum, uint32_t nb) {
  const uint32_t i0 = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  const uint32_t add = blocksum[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i)
    if (i0 + i < n) out[i0 + i] += add;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = blocksum[nb];
}

// ------------------------------------------------------------------------------------------------
// K3: score one (query, candidate) pair per lane, straight off the flat pair list.
//   damerau_levenshtein (src/distance.rs:101-179) in its band-limited saturating form:
//   cells with |i-j| > d are d+1, every value saturates at d+1, the transposition term only looks back
//   d rows / d columns (farther ones cost > d).  Identical to the reference for every outcome <= d.
//   Per-lane state lives in LDS: query row, candidate row, a ring of d+2 band rows.
//   longest_common_substring_length / common_prefix_length / common_suffix_length: src/distance.rs:181-231.
//   Score: src/lib.rs:1433-1452 (f64, same association, no FMA contraction).
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// K2: prefilter + selection.  A necessary condition for damerau_levenshtein(q, c) <= d (src/distance.rs:101-179):
// every optimal edit script matches all but <= d symbols of q (and of c) to an EQUAL symbol of the other string
// at an offset within +-d (each unmatched symbol costs one deletion/insertion/substitution; transposed symbols
// are equal symbols within the offset bound).  So count the positions of q that have no equal symbol of c in
// [i-d, i+d] (and vice versa); more than d of them => the reference returns None.  Pure register SWAR over the
// two 16-byte rows (7 byte-shifts with v_alignbyte_b32, zero-byte detection), no LDS, no DP.  On config 2 it
// rejects ~2/3 of the pairs; the banded DP then runs only on the selected third.  Strings longer than 16
// symbols or d > 3 are passed through unfiltered.
// ------------------------------------------------------------------------------------------------
#define PAIR_NONE 0x7Fu

__device__ inline uint32_t nonzero_bytes(uint32_t x) {  // bit 7 of every byte that is non-zero
  return ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;
}
__device__ inline uint32_t len_mask(int len, int k) {  // 0x80 in every byte position (4k..4k+3) below len
  const int n = len - 4 * k;
  return n >= 4 ? 0x80808080u : n <= 0 ? 0u : (0x80808080u & ((1u << (8 * n)) - 1u));
}

template <int DELTA, int NW>
__device__ inline void filter_shift(const uint32_t (&q)[NW], const uint32_t (&c)[NW + 2], bool enabled, uint32_t (&nmA)[NW],
                                    uint32_t (&nmB)[NW]) {
  uint32_t nz[NW + 2];
  nz[0] = 0xFFFFFFFFu;
  nz[NW + 1] = 0xFFFFFFFFu;
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    uint32_t cs;  // bytes C[4k + DELTA ..]
    if (DELTA == 0) cs = c[k + 1];
    else if (DELTA > 0) cs = __builtin_amdgcn_alignbyte(c[k + 2], c[k + 1], DELTA);
    else cs = __builtin_amdgcn_alignbyte(c[k + 1], c[k], 4 + DELTA);
    const uint32_t v = nonzero_bytes(q[k] ^ cs);  // bit7 set where q[i] != c[i + DELTA]
    nz[k + 1] = enabled ? v : 0xFFFFFFFFu;
    nmA[k] &= nz[k + 1];
  }
#pragma unroll
  for (int k = 0; k < NW; ++k) {  // the same comparisons seen from c: position j pairs with i = j - DELTA
    uint32_t b;
    if (DELTA == 0) b = nz[k + 1];
    else if (DELTA > 0) b = __builtin_amdgcn_alignbyte(nz[k + 1], nz[k], 4 - DELTA);
    else b = __builtin_amdgcn_alignbyte(nz[k + 2], nz[k + 1], -DELTA);
    nmB[k] &= b;
  }
}
// band-match bound: a symbol with no equal symbol of the other string within +-d positions costs at least one edit
template <int NW>
__device__ inline bool band_bound_rejects(const uint32_t (&q)[NW], const uint32_t (&c)[NW + 2], bool filt, int d, int lq, int lc) {
  uint32_t nmA[NW], nmB[NW];
#pragma unroll
  for (int k = 0; k < NW; ++k) { nmA[k] = 0xFFFFFFFFu; nmB[k] = 0xFFFFFFFFu; }
  filter_shift<0, NW>(q, c, true, nmA, nmB);
  if (__any(filt && d >= 1)) { filter_shift<1, NW>(q, c, d >= 1, nmA, nmB); filter_shift<-1, NW>(q, c, d >= 1, nmA, nmB); }
  if (__any(filt && d >= 2)) { filter_shift<2, NW>(q, c, d >= 2, nmA, nmB); filter_shift<-2, NW>(q, c, d >= 2, nmA, nmB); }
  if (__any(filt && d >= 3)) { filter_shift<3, NW>(q, c, d >= 3, nmA, nmB); filter_shift<-3, NW>(q, c, d >= 3, nmA, nmB); }
  int unA = 0, unB = 0;
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    unA += __popc(nmA[k] & len_mask(lq, k));
    unB += __popc(nmB[k] & len_mask(lc, k));
  }
  return filt && (unA > d || unB > d);
}

struct ScoreArgs {
  const double* quot;  // [33][33] quot[x*33+L] = (double)x / (double)L computed on the host, or nullptr
  int dbg;  // ANX_SCORE_DBG (timing experiments only): 1 skip LCS, 2 skip everything after DL
  double w_ld, w_lcs, w_prefix, w_suffix, w_case, w_sum;
  double score_threshold;
  int have_freq, any_variants;
  uint32_t lqp, lcp;   // bytes reserved per lane for the query / candidate row (multiples of 16)
  uint32_t stride;     // bytes per lane (odd number of dwords: conflict-free ds access)
  uint32_t qw;
};

// Appends the wave's survivors to the survivor list: one atomic per wave on the counter of the given region (the
// pair-list region the pairs come from, so a region holds at most as many survivors as that region has slots).
struct SurvOut {
  SurvRec* list;
  uint32_t* ctr;          // [SCAN_REGIONS][RC_STRIDE]
  uint32_t region_cap;
};
__device__ inline void surv_append(const SurvOut& o, uint32_t region, bool keep, uint32_t q, uint32_t e, double score) {
  const unsigned long long km = __ballot(keep);
  if (!km) return;  // wave-uniform
  const uint32_t lane = threadIdx.x & 63;
  uint32_t base = 0;
  if (lane == (uint32_t)__ffsll((long long)km) - 1u) base = atomicAdd(&o.ctr[region * RC_STRIDE], (uint32_t)__popcll(km));
  base = (uint32_t)__builtin_amdgcn_readlane((int)base, __ffsll((long long)km) - 1);
  const uint32_t pos = base + (uint32_t)__popcll(km & ((1ull << lane) - 1ull));
  if (keep && pos < o.region_cap) o.list[(size_t)region * o.region_cap + pos] = SurvRec{q, e, score};
}

// The part of gather_instances / score_and_rank that follows a successful Damerau-Levenshtein (ld <= d):
// LCS, prefix, suffix, case (src/lib.rs:1352-1377), the f64 score (:1433-1452), max_freq and the survivor count.
__device__ inline double score_tail(const uint8_t* S, const uint8_t* T, int lq, int lc, uint32_t ld, uint32_t qm, uint32_t em,
                                    uint32_t q, uint32_t e, const ScoreArgs& a, const uint32_t* __restrict__ ent_freq,
                                    const uint32_t* __restrict__ ent_var_off, uint32_t* __restrict__ qmaxfreq,
                                    uint32_t* __restrict__ qsurv, uint32_t* __restrict__ qexpand, uint32_t& lcs,
                                    uint32_t& pre, uint32_t& suf, uint32_t& samecase, bool& keep) {
  if (a.w_lcs > 0.0 && !(a.dbg & 1)) {
    // longest common substring (src/lib.rs:1352-1356, src/distance.rs:181-205) = longest run of equal symbols on
    // any diagonal.  Diagonals are visited from the main one outwards (0, +1, -1, +2, ...): the overlap of a diagonal
    // only shrinks with |delta|, so the walk stops as soon as neither side can beat the best run found so far.
    uint32_t best = 0;
    for (int r = 0; r < max(lq, lc); ++r) {
      bool open = false;
      for (int side = 0; side < (r ? 2 : 1); ++side) {
        const int delta = side ? -r : r;
        const int i0 = delta < 0 ? -delta : 0;
        const int i1 = min(lq, lc - delta);
        if (i1 - i0 <= (int)best) continue;
        open = true;
        uint32_t run = 0;
        for (int i = i0; i < i1; ++i) {
          run = S[i] == T[i + delta] ? run + 1 : 0;
          best = max(best, run);
        }
      }
      if (!open) break;
    }
    lcs = best;
  }
  const int m = min(lq, lc);
  if (a.w_prefix > 0.0) {
    int n = 0;
    while (n < m && S[n] == T[n]) ++n;
    pre = n;
  }
  if (a.w_suffix > 0.0) {
    int n = 0;
    while (n < m && S[lq - 1 - n] == T[lc - 1 - n]) ++n;
    suf = n;
  }
  if (a.w_case > 0.0) samecase = ((qm >> 24) & 1u) == ((em >> 8) & 1u);  // src/lib.rs:1367-1377
  // x / L for integers x <= L <= 32 comes from a table of host-computed IEEE quotients (identical bits, no f64 divide)
  const double L = (double)lq;
  const bool tab = a.quot && lq <= 32;
  auto over_L = [&](uint32_t x) { return (tab && x <= 32u) ? a.quot[x * 33u + (uint32_t)lq] : (double)x / L; };
  const double distance_score = (int)ld > lq ? 0.0 : 1.0 - over_L(ld);
  const double lcs_score = over_L(lcs);
  const double prefix_score = over_L(pre);
  const double suffix_score = over_L(suf);
  const double num = a.w_ld * distance_score + a.w_lcs * lcs_score + a.w_prefix * prefix_score +
                     a.w_suffix * suffix_score + (samecase ? a.w_case : 0.0);
  const double score = a.w_sum == 1.0 ? num : num / a.w_sum;  // x / 1.0 == x
  // max_freq over every DL-surviving instance, before the threshold test (src/lib.rs:1454-1462)
  atomicMax(&qmaxfreq[q], a.have_freq ? ent_freq[e] : 1u);
  uint32_t nrows = 1;
  if (a.any_variants) {  // variant lists loaded (src/lib.rs:1464-1466, 1510, 1677-1727)
    if (em & 0x200u) qexpand[q] = 1;  // benign race: every writer stores 1
    nrows = (ent_var_off[e + 1] - ent_var_off[e]) + ((em & 0x400u) ? 0u : 1u);  // transparent: references only
  }
  keep = score >= a.score_threshold && nrows;  // src/lib.rs:1475
  if (keep) atomicAdd(&qsurv[q], nrows);
  return score;
}

// ------------------------------------------------------------------------------------------------
// K3 fast path: pairs with both strings <= 16 symbols and d <= 3 (every pair of BASELINE configs 1-2).
// The banded unrestricted Damerau-Levenshtein runs entirely in registers: both strings are 4 dwords, the row loop
// is fully unrolled (row number, band column and matrix column are compile-time constants, lanes whose query is
// shorter are masked), band rows live in a ring of D+2 register rows, and values are NOT saturated: every cell is
// >= the true distance and exact along any path of cost <= D, cells outside the band read as D+1 (their true value
// is >= D+1, so everything derived from them is > D), which gives the same outcome for every result <= d
// (SURVEY.md appendix A.3).  The transposition term of src/distance.rs:157-162 in band form: with
// l = i-1-a the last earlier row whose symbol equals t[j-1] and db = j-1-b the last earlier column of this row
// that matches s[i-1], T = D[l-1][db-1] + a + b + 1, only needed for a + b <= D - 1.
// ------------------------------------------------------------------------------------------------

template <int NW>
__device__ inline uint32_t byte_of(const uint32_t (&w)[NW], int idx) { return (w[idx >> 2] >> (8 * (idx & 3))) & 0xFFu; }

template <int D, int NW>
__device__ inline uint32_t dl_band(const uint32_t (&S)[NW], const uint32_t (&T)[NW], int lq, int lc, int lqmax) {
  constexpr int BW = 2 * D + 1, NR = D + 2, MAXLEN = 4 * NW;
  constexpr uint32_t CAP = D + 1;
  uint32_t row[NR][BW];
  // T padded with D+1 never-matching bytes in front: the band window of row i is bytes [i, i+2D] of tp
  uint32_t tp[NW + 3];
  {
    constexpr int SH = D + 1;  // 2..4 bytes
    const uint32_t fill = 0xFFFFFFFFu;
    if (SH == 4) {
      tp[0] = fill;
#pragma unroll
      for (int w = 0; w < NW; ++w) tp[w + 1] = T[w];
      tp[NW + 1] = fill;
      tp[NW + 2] = fill;
    } else {
      tp[0] = __builtin_amdgcn_alignbyte(T[0], fill, 4 - SH);
#pragma unroll
      for (int w = 1; w < NW; ++w) tp[w] = __builtin_amdgcn_alignbyte(T[w], T[w - 1], 4 - SH);
      tp[NW] = __builtin_amdgcn_alignbyte(fill, T[NW - 1], 4 - SH);
      tp[NW + 1] = fill;
      tp[NW + 2] = fill;
    }
  }
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int c = 0; c < BW; ++c) row[r][c] = CAP;
#pragma unroll
  for (int c = 0; c < BW; ++c) row[0][c] = c - D >= 0 ? (uint32_t)(c - D) : CAP;
#pragma unroll
  for (int i = 1; i <= MAXLEN; ++i) {
    if (i <= lqmax) {    // wave-uniform
      if (i <= lq) {     // lanes with shorter queries keep their last row
        const uint32_t sc = byte_of<NW>(S, i - 1);
        const uint32_t wlo = __builtin_amdgcn_alignbyte(tp[(i >> 2) + 1], tp[i >> 2], i & 3);
        const uint32_t whi = __builtin_amdgcn_alignbyte(tp[(i >> 2) + 2], tp[(i >> 2) + 1], i & 3);
        uint32_t (&cur)[BW] = row[i % NR];
        const uint32_t (&prev)[BW] = row[(i - 1) % NR];
        bool mt[BW];
        uint32_t nv[BW];
#pragma unroll
        for (int c = 0; c < BW; ++c) {
          const int j = i + c - D;
          mt[c] = false;
          nv[c] = CAP;
          if (j == 0) nv[c] = (uint32_t)i;
          else if (j >= 1 && j <= MAXLEN) {
            const uint32_t tc = ((c < 4 ? wlo : whi) >> (8 * (c & 3))) & 0xFFu;
            mt[c] = sc == tc;
            const uint32_t up = c + 1 < BW ? prev[c + 1] : CAP;
            const uint32_t left = c > 0 ? nv[c - 1] : CAP;
            uint32_t v = min(min(left, up) + 1u, prev[c] + (mt[c] ? 0u : 1u));
            // transposition
            bool eqs[D], any_eqs = false, any_mt = false;
#pragma unroll
            for (int a = 0; a < D; ++a) {
              eqs[a] = i - 2 - a >= 0 ? byte_of<NW>(S, i - 2 - a >= 0 ? i - 2 - a : 0) == tc : false;
              any_eqs |= eqs[a];
            }
#pragma unroll
            for (int b = 0; b < D; ++b)
              if (c - 1 - b >= 0) any_mt |= mt[c - 1 - b];
            if (__builtin_amdgcn_ballot_w64(any_eqs && any_mt)) {  // wave-uniform: some lane has a transposition candidate
              bool a_open = true;  // no closer row matched yet
#pragma unroll
              for (int a = 0; a < D; ++a) {
                if (i - 2 - a < 0) break;
                bool b_open = true;  // no closer column matched yet
#pragma unroll
                for (int b = 0; a + b < D; ++b) {
                  const int cb = c - 1 - b, x = c + a - b;
                  if (cb < 0 || j - 1 - b < 1) break;
                  if (x >= 0 && x < BW) {
                    const bool cond = a_open && eqs[a] && b_open && mt[cb];
                    const uint32_t tv = row[(i - 2 - a) % NR][x] + (uint32_t)(a + b + 1);
                    v = cond ? min(v, tv) : v;
                  }
                  b_open = b_open && !mt[cb];
                }
                a_open = a_open && !eqs[a];
              }
            }
            nv[c] = v;
          }
        }
#pragma unroll
        for (int c = 0; c < BW; ++c) cur[c] = nv[c];
      }
    }
  }
  // D[lq][lc]: ring row lq % NR, band column lc - lq + D
  uint32_t res = CAP;
  const int rsel = lq % NR, csel = lc - lq + D;
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int c = 0; c < BW; ++c) res = (rsel == r && csel == c) ? row[r][c] : res;
  return res;
}

// A list of pair-list slots per region (the selected pairs a later kernel has to score), appended per wave.
struct SlotList {
  uint32_t* list;      // [SCAN_REGIONS][region_cap]
  uint32_t* ctr;       // [SCAN_REGIONS][RC_STRIDE]
  uint32_t region_cap;
};
__device__ inline void slot_append(const SlotList& o, uint32_t region, bool put, uint32_t slot) {
  const unsigned long long km = __ballot(put);
  if (!km) return;  // wave-uniform
  const uint32_t lane = threadIdx.x & 63;
  const int first = __ffsll((long long)km) - 1;
  uint32_t base = 0;
  if ((int)lane == first) base = atomicAdd(&o.ctr[region * RC_STRIDE], (uint32_t)__popcll(km));
  base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
  const uint32_t pos = base + (uint32_t)__popcll(km & ((1ull << lane) - 1ull));
  if (put && pos < o.region_cap) o.list[(size_t)region * o.region_cap + pos] = slot;
}

struct PairArgs {  // what every scoring kernel reads / writes
  const uint2* raw;
  const uint32_t* q_meta;
  const uint4* q_rows;
  const uint32_t* ent_meta;
  const uint32_t* ent_rowoff;
  const uint4* rows;
  const uint32_t* ent_freq;
  const uint32_t* ent_var_off;
  double* p_score;      // per pair-list slot
  uint32_t* p_meta;     // per pair-list slot: ld | samecase<<7 | lcs<<8 | prefix<<16 | suffix<<24, or skipped / rejected
  uint32_t* qmaxfreq;
  uint32_t* qsurv;
  uint32_t* qexpand;
};

// Scores the pair in slot p with the register-resident DL of NW words (all lanes of the wave call this; lanes with
// !active only take part in the wave-wide steps).  lds: per-lane staging of both strings for the byte-wise tail.
template <int D, int NW>
__device__ inline void score_fast_pair(uint32_t p, bool active, const PairArgs& A, const ScoreArgs& a, const SurvOut& so,
                                       uint32_t surv_region, uint32_t* __restrict__ lds) {
  uint32_t q = 0, e = 0, qm = 0, em = 0;
  int lq = 0, lc = 0, d = 0;
  uint32_t S[NW], T[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) { S[w] = 0xFEFEFEFEu; T[w] = 0xFFFFFFFFu; }
  if (active) {
    const uint2 rp = A.raw[p];
    q = rp.x;
    e = rp.y & 0x7FFFFFFFu;
    qm = A.q_meta[q];
    em = A.ent_meta[e];
    lq = qm & 0xFF; d = (qm >> 16) & 0xFF; lc = em & 0xFF;
    const uint4* qr = A.q_rows + (size_t)q * a.qw;
    const uint4* cr = A.rows + A.ent_rowoff[e];
#pragma unroll
    for (int w = 0; w < NW / 4; ++w) {
      if (w * 16 < lq) { const uint4 Q = qr[w]; S[4 * w] = Q.x; S[4 * w + 1] = Q.y; S[4 * w + 2] = Q.z; S[4 * w + 3] = Q.w; }
      if (w * 16 < lc) { const uint4 C = cr[w]; T[4 * w] = C.x; T[4 * w + 1] = C.y; T[4 * w + 2] = C.z; T[4 * w + 3] = C.w; }
    }
  }
  int lqmax = active ? lq : 0;
#pragma unroll
  for (int o = 32; o; o >>= 1) lqmax = max(lqmax, __shfl_xor(lqmax, o));
  lqmax = __builtin_amdgcn_readfirstlane(lqmax);
  const uint32_t res = dl_band<D, NW>(S, T, active ? lq : 0, lc, lqmax);
  uint32_t ld = PAIR_NONE, lcs = 0, pre = 0, suf = 0, samecase = 1;
  double score = __builtin_nan("");
  bool keep = false;
  const int diff = lq > lc ? lq - lc : lc - lq;
  if (active && diff <= d && res <= (uint32_t)d && !(a.dbg & 2)) {  // src/distance.rs:109-130, 173-178
    uint32_t* mine = lds + (threadIdx.x & 255) * (2 * NW + 1);
#pragma unroll
    for (int w = 0; w < NW; ++w) { mine[w] = S[w]; mine[NW + w] = T[w]; }
    ld = res;
    score = score_tail(reinterpret_cast<const uint8_t*>(mine), reinterpret_cast<const uint8_t*>(mine + NW), lq, lc, ld, qm, em,
                       q, e, a, A.ent_freq, A.ent_var_off, A.qmaxfreq, A.qsurv, A.qexpand, lcs, pre, suf, samecase, keep);
  }
  surv_append(so, surv_region, keep, q, e, score);
  if (active) {
    A.p_score[p] = score;
    A.p_meta[p] = ld | (samecase << 7) | (lcs << 8) | (pre << 16) | (suf << 24);
  }
}

// K2+K3 fused: prefilter of every pair-list slot and register-resident DL of the selected pairs, one block per
// FS_BLK consecutive slots of a region.  Phase 1 (FS_BLK / 256 rounds): length test |lq - lc| <= d
// (src/distance.rs:109-130), StopAtExactMatch drop (src/lib.rs:1164-1173) and the SWAR band-match bound; selected
// pairs of <= 16 symbols with d <= D are queued in LDS, longer ones go to the slot lists of the 8-word / general
// kernels.  Phase 2: the queue is scored 256 pairs at a time, so the DL lanes are dense although only ~1/3 of the
// slots survive phase 1 (no global compaction pass, no index list).  D = 0: no inline DL (d > 3), everything selected
// goes to the general kernel's list.
constexpr uint32_t FS_BLK = 4096;
struct FilterArgs {
  uint32_t region_shift;
  const uint32_t* rctr;     // region fills of the pair list
  const uint32_t* qexact;
  int stop, enable;
  int use_nw8;              // selected pairs of 17..32 symbols go to list8 (else to the general list)
  uint32_t* counters;
  uint32_t* stat_ctr;       // [SCAN_REGIONS][RC_STRIDE], word 1: selected pairs
};
template <int D>
__global__ __launch_bounds__(256) void k_filter_score(FilterArgs f, PairArgs A, ScoreArgs a, SurvOut so, SlotList list8, SlotList listg) {
  __shared__ uint16_t s_q[FS_BLK];  // queued pairs as offsets from the block's first slot
  __shared__ uint32_t s_n;
  __shared__ uint32_t s_str[256 * 9];
  // 1-D grid, region fastest: blocks that run at the same time append to different regions' counters (a single
  // counter word sustains only ~88 M atomics/s)
  const uint32_t region = blockIdx.x % SCAN_REGIONS, fill = f.rctr[region * RC_STRIDE + RC_RAW], base = (blockIdx.x / SCAN_REGIONS) * FS_BLK;
  if (base >= fill) return;  // block-uniform
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63;
  uint32_t nselected = 0;  // wave-uniform
  for (uint32_t r = 0; r < FS_BLK / 256; ++r) {
    const uint32_t idx = base + r * 256 + threadIdx.x;
    if (base + r * 256 >= fill) break;  // block-uniform
    const uint32_t p = (region << f.region_shift) + idx;
    const bool live = idx < fill;
    bool selected = false, stop_skipped = false, invalid = false;
    int d = 0, lq = 0, lc = 0;
    // words that are not loaded keep the row padding (query 0xFE, candidate 0xFF: never equal to anything)
    uint32_t q8[8] = {0, 0, 0, 0, 0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu};
    uint32_t c10[10] = {0xFFFFFFFFu, 0, 0, 0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    bool filt = false, wide = false;
    if (live) {
      const uint2 rp = A.raw[p];
      const uint32_t q = rp.x, e = rp.y & 0x7FFFFFFFu;
      invalid = q == RAW_INVALID;
      // unused chunk tail, or (StopAtExactMatch) a non-exact class of a query that has an exact one
      const bool skip = invalid || (f.stop && !(rp.y & 0x80000000u) && f.qexact[q] != 0xFFFFFFFFu);
      stop_skipped = skip && !invalid;
      if (!skip) {
        const uint32_t qm = A.q_meta[q], em = A.ent_meta[e];
        lq = qm & 0xFF; d = (qm >> 16) & 0xFF; lc = em & 0xFF;
        const int diff = lq > lc ? lq - lc : lc - lq;
        selected = diff <= d;
        filt = selected && f.enable && d <= 3 && lq <= 32 && lc <= 32;
        wide = filt && (lq > 16 || lc > 16);
        if (filt) {
          const uint4* qr = A.q_rows + (size_t)q * a.qw;
          const uint4* cr = A.rows + A.ent_rowoff[e];
          const uint4 Q = qr[0], C = cr[0];
          q8[0] = Q.x; q8[1] = Q.y; q8[2] = Q.z; q8[3] = Q.w;
          c10[1] = C.x; c10[2] = C.y; c10[3] = C.z; c10[4] = C.w;
          if (lq > 16) { const uint4 Q1 = qr[1]; q8[4] = Q1.x; q8[5] = Q1.y; q8[6] = Q1.z; q8[7] = Q1.w; }
          if (lc > 16) { const uint4 C1 = cr[1]; c10[5] = C1.x; c10[6] = C1.y; c10[7] = C1.z; c10[8] = C1.w; }
        }
      }
    }
    if (__any(wide)) {  // wave-uniform: some pair of the wave has a string of 17..32 symbols
      if (band_bound_rejects<8>(q8, c10, filt, d, lq, lc)) selected = false;
    } else if (__any(filt)) {
      const uint32_t q4[4] = {q8[0], q8[1], q8[2], q8[3]}, c6[6] = {0xFFFFFFFFu, c10[1], c10[2], c10[3], c10[4], 0xFFFFFFFFu};
      if (band_bound_rejects<4>(q4, c6, filt, d, lq, lc)) selected = false;
    }
    if (live && !selected)  // skipped (tail / StopAtExactMatch) or rejected: ld = None, samecase = true
      A.p_meta[p] = (invalid || stop_skipped) ? META_SKIPPED : (PAIR_NONE | (1u << 7));
    const bool inl = selected && D > 0 && lq <= 16 && lc <= 16 && d <= D;
    const bool to8 = selected && !inl && f.use_nw8 && D > 0 && lq <= 32 && lc <= 32 && d <= D;
    const bool tog = selected && !inl && !to8;
    const unsigned long long mi = __ballot(inl);
    if (mi) {  // wave-uniform: queue the inline pairs
      const int first = __ffsll((long long)mi) - 1;
      uint32_t qb = 0;
      if ((int)lane == first) qb = atomicAdd(&s_n, (uint32_t)__popcll(mi));
      qb = (uint32_t)__builtin_amdgcn_readlane((int)qb, first);
      if (inl) s_q[qb + (uint32_t)__popcll(mi & ((1ull << lane) - 1ull))] = (uint16_t)(r * 256 + threadIdx.x);
    }
    slot_append(list8, region, to8, p);
    slot_append(listg, region, tog, p);
    nselected += (uint32_t)__popcll(__ballot(selected));
    if (f.stop) {  // scored pairs = pairs emitted by the scan minus the ones StopAtExactMatch drops
      const unsigned long long ms = __ballot(stop_skipped);
      if (lane == 0 && ms) atomicAdd(&f.counters[CTR_SKIPPED], (uint32_t)__popcll(ms));
    }
  }
  if (lane == 0 && nselected) atomicAdd(&f.stat_ctr[region * RC_STRIDE + 1], nselected);
  __syncthreads();
  if (D > 0) {
    const uint32_t n = s_n;
    for (uint32_t r0 = 0; r0 < n; r0 += 256) {  // block-uniform trip count
      const uint32_t i = r0 + threadIdx.x;
      const bool active = i < n;
      score_fast_pair<(D > 0 ? D : 1), 4>(active ? (region << f.region_shift) + base + s_q[i] : 0u, active, A, a, so, region, s_str);
    }
  }
}

// the selected pairs with a string of 17..32 symbols (list8 of k_filter_score)
template <int D>
__global__ __launch_bounds__(256) void k_score_fast8(SlotList in, PairArgs A, ScoreArgs a, SurvOut so) {
  __shared__ uint32_t s_str[256 * 17];
  const uint32_t region = blockIdx.x % SCAN_REGIONS, blk = blockIdx.x / SCAN_REGIONS, i = blk * 256 + threadIdx.x, n = in.ctr[region * RC_STRIDE];
  if (blk * 256 >= n) return;  // block-uniform
  const bool active = i < n;
  score_fast_pair<D, 8>(active ? in.list[(size_t)region * in.region_cap + i] : 0u, active, A, a, so, region, s_str);
}

__global__ void k_score_pairs(SlotList in, PairArgs A, ScoreArgs a, SurvOut so) {
  extern __shared__ uint32_t lds32[];
  const uint2* __restrict__ raw = A.raw;
  const uint32_t* __restrict__ q_meta = A.q_meta;
  const uint4* __restrict__ q_rows = A.q_rows;
  const uint32_t* __restrict__ ent_meta = A.ent_meta;
  const uint32_t* __restrict__ ent_rowoff = A.ent_rowoff;
  const uint4* __restrict__ rows = A.rows;
  const uint32_t* __restrict__ ent_freq = A.ent_freq;
  const uint32_t* __restrict__ ent_var_off = A.ent_var_off;
  uint32_t* __restrict__ qmaxfreq = A.qmaxfreq;
  uint32_t* __restrict__ qsurv = A.qsurv;
  uint32_t* __restrict__ qexpand = A.qexpand;
  const uint32_t region = blockIdx.x % SCAN_REGIONS, blk = blockIdx.x / SCAN_REGIONS, i_sel = blk * blockDim.x + threadIdx.x, nsel = in.ctr[region * RC_STRIDE];
  if (blk * blockDim.x >= nsel) return;  // block-uniform
  bool keep = false;
  uint32_t kq = 0, ke = 0;
  double kscore = 0.0;
  if (i_sel < nsel) {
    const uint32_t p = in.list[(size_t)region * in.region_cap + i_sel];
    const uint2 rp = raw[p];
    const uint32_t q = rp.x, e = rp.y & 0x7FFFFFFFu;
    uint32_t ld = PAIR_NONE, lcs = 0, pre = 0, suf = 0, samecase = 1;
    double score = __builtin_nan("");
    {
      uint8_t* S = reinterpret_cast<uint8_t*>(lds32) + (size_t)threadIdx.x * a.stride;
      uint8_t* T = S + a.lqp;
      uint8_t* R = T + a.lcp;
      const uint32_t qm = q_meta[q], em = ent_meta[e];
      const int lq = qm & 0xFF, d = (qm >> 16) & 0xFF, lc = em & 0xFF;
      const int diff = lq > lc ? lq - lc : lc - lq;
      if (diff <= d) {  // src/distance.rs:109-130 (both lengths > 0 here)
        {
          uint32_t* S32 = reinterpret_cast<uint32_t*>(S);
          const uint4* qr = q_rows + (size_t)q * a.qw;
          for (int wq = 0; wq * 16 < lq; ++wq) {
            const uint4 v = qr[wq];
            S32[wq * 4 + 0] = v.x; S32[wq * 4 + 1] = v.y; S32[wq * 4 + 2] = v.z; S32[wq * 4 + 3] = v.w;
          }
          uint32_t* T32 = reinterpret_cast<uint32_t*>(T);
          const uint4* cr = rows + ent_rowoff[e];
          for (int wc = 0; wc * 16 < lc; ++wc) {
            const uint4 v = cr[wc];
            T32[wc * 4 + 0] = v.x; T32[wc * 4 + 1] = v.y; T32[wc * 4 + 2] = v.z; T32[wc * 4 + 3] = v.w;
          }
        }
        // ---- banded unrestricted Damerau-Levenshtein ------------------------------------------------
        const int cap = d + 1, W = 2 * d + 3, NR = d + 2;
        // row i is stored at R[(i % NR) * W + col], col = j - i + d + 1 in [1, 2d+1]; cols 0, 2d+2 are guards
        for (int col = 0; col < W; ++col) {
          const int j = col - d - 1;
          R[col] = (uint8_t)((j >= 0 && j <= lc && col >= 1 && col <= 2 * d + 1) ? (j < cap ? j : cap) : cap);
        }
        for (int i = 1; i <= lq; ++i) {
          uint8_t* cur = R + (i % NR) * W;
          const uint8_t* prev = R + ((i - 1) % NR) * W;
          const uint32_t sc = S[i - 1];
          int db = 0;
          cur[0] = (uint8_t)cap;
          for (int col = 1; col <= 2 * d + 1; ++col) {
            const int j = i + col - d - 1;
            uint32_t v;
            if (j < 0 || j > lc) v = cap;
            else if (j == 0) v = i < cap ? i : cap;
            else {
              const uint32_t tc = T[j - 1];
              const uint32_t cost = sc != tc;
              v = min(min((uint32_t)cur[col - 1] + 1u, (uint32_t)prev[col + 1] + 1u), (uint32_t)prev[col] + cost);
              if (db > 0) {
                // l = last row i' < i with s[i'-1] == t[j-1] (char_map, src/distance.rs:146,154,170), looking
                // back at most d rows: farther rows make the term exceed d
                for (int back = 0; back < d; ++back) {
                  const int l = i - 1 - back;
                  if (l < 1) break;
                  if (S[l - 1] == tc) {
                    const int colx = db - l + d + 1;  // column of D[l-1][db-1] in row l-1
                    if (colx >= 1 && colx <= 2 * d + 1) {
                      const uint32_t tv = (uint32_t)R[((l - 1) % NR) * W + colx] + (uint32_t)(i - l - 1) + 1u +
                                          (uint32_t)(j - db - 1);  // src/distance.rs:161
                      v = min(v, tv);
                    }
                    break;
                  }
                }
              }
              v = min(v, (uint32_t)cap);
              if (cost == 0) db = j;  // src/distance.rs:165-167
            }
            cur[col] = (uint8_t)v;
          }
          cur[2 * d + 2] = (uint8_t)cap;
        }
        const uint32_t res = R[(lq % NR) * W + (lc - lq + d + 1)];
        if (res <= (uint32_t)d && !(a.dbg & 2)) {  // src/distance.rs:173-178
          ld = res;
          score = score_tail(S, T, lq, lc, ld, qm, em, q, e, a, ent_freq, ent_var_off, qmaxfreq, qsurv, qexpand, lcs, pre, suf, samecase, keep);
          kq = q; ke = e; kscore = score;
        }
      }
    }
    A.p_score[p] = score;
    A.p_meta[p] = ld | (samecase << 7) | (lcs << 8) | (pre << 16) | (suf << 24);
  }
  surv_append(so, region, keep, kq, ke, kscore);
}

// K3b: gather the survivors (score >= threshold) into per-query segments of result rows.  With variant lists a
// survivor contributes one row per VariantOf reference (expand_variants, src/lib.rs:1677-1727: score * variant
// score, min(reference frequency, own frequency), via = itself) and itself unless it is TRANSPARENT.
// Order inside a query is arbitrary; ranking uses a total order whose last key is c_ord (= reference order).
struct CompactArgs {
  int have_freq, any_variants;
};
__global__ __launch_bounds__(256) void k_compact(const SurvRec* __restrict__ surv, const uint32_t* __restrict__ sctr,
                                                 uint32_t region_cap, CompactArgs a, const uint32_t* __restrict__ soff,
                                                 uint32_t* __restrict__ qcur, const EntRec* __restrict__ ent_rec,
                                                 const uint32_t* __restrict__ ent_var_off,
                                                 const uint32_t* __restrict__ var_target,
                                                 const uint32_t* __restrict__ var_target_freq,
                                                 const double* __restrict__ var_score, SurvRow* __restrict__ c_rows) {
  const uint32_t region = blockIdx.x % SCAN_REGIONS, i = (blockIdx.x / SCAN_REGIONS) * 256 + threadIdx.x;  // 1-D grid, region fastest
  if (i >= sctr[region * RC_STRIDE]) return;
  const SurvRec sr = surv[(size_t)region * region_cap + i];
  const double s = sr.score;
  const uint32_t e = sr.e, q = sr.q;
  const EntRec er = ent_rec[e];
  const uint32_t f = a.have_freq ? er.freq : 1u;
  const unsigned long long ord = (unsigned long long)er.order << 20;
  uint32_t v0 = 0, v1 = 0, self = 1;
  if (a.any_variants) {
    v0 = ent_var_off[e];
    v1 = ent_var_off[e + 1];
    self = (er.meta & 0x400u) ? 0u : 1u;
  }
  const uint32_t nrows = (v1 - v0) + self;
  uint32_t pos = soff[q] + atomicAdd(&qcur[q], nrows);
  for (uint32_t j = v0; j < v1; ++j, ++pos) {  // references first, then the item itself (src/lib.rs:1689-1717)
    const uint32_t tf = var_target_freq[j];
    // min(target frequency, own freq_score)
    c_rows[pos] = SurvRow{s * var_score[j], ord | (unsigned long long)(j - v0), var_target[j],
                          a.have_freq ? (tf < f ? tf : f) : (tf < 1u ? tf : 1u), er.vocab, 0u};
  }
  if (self) c_rows[pos] = SurvRow{s, ord | (unsigned long long)(v1 - v0), er.vocab, f, 0xFFFFFFFFu, 0u};
}

// ------------------------------------------------------------------------------------------------
// K4: rank.  One wave per query over its survivors.  freq normalisation (src/lib.rs:1521-1525),
// stable sort by rank_cmp (src/types.rs:344-365) realised as a total order with ent_order as last key,
// crop with the tie rule (:1536-1589), cutoff (:1598-1622).
// ------------------------------------------------------------------------------------------------
struct RankArgs {
  double cutoff_threshold;
  uint64_t max_matches;
  float freq_weight;
  int have_freq, any_variants;
};
constexpr int RANK_LCAP = 128;  // rows per query staged in LDS by the 64-lane path; longer lists spill to t_key / global reads

__device__ inline double result_score(double dist, double freq, float fw) {  // src/types.rs:335-341
  if (fw == 0.0f) return dist;
  return (dist + ((double)fw * freq)) / (1.0 + (double)fw);
}

// One group of G lanes ranks one query, every candidate row taking part (rows beyond LCAP through t_key / global
// reads).  Every lane of the wave calls this (ballots are wave-wide, sliced per group).
template <int G, int LCAP>
__device__ inline void rank_query_all(uint32_t q, bool valid, int gl, int gshift, double* __restrict__ s_key,
                                  unsigned long long* __restrict__ s_ord, uint32_t* __restrict__ s_freq,
                                  double* __restrict__ s_sdist, double* __restrict__ s_sfreq, uint32_t seg0, uint32_t n,
                                  uint32_t maxf, uint32_t qex, const SurvRow* __restrict__ c_rows, const RankArgs& a,
                                  double* __restrict__ t_key, DevRow* __restrict__ r_rows, uint32_t* __restrict__ r_count) {
  const unsigned long long gmask = G >= 64 ? ~0ull : ((1ull << (G & 63)) - 1ull);
  if (!valid) n = 0;
  if (valid && n == 0 && gl == 0) r_count[q] = 0;
  // expanded rows never raise max_freq: their frequency is a min() with the expanding item's (src/lib.rs:1512-1517)
  const double max_freq = a.have_freq ? (double)maxf : (maxf ? 1.0 : 0.0);
  const bool sort_weighted = a.freq_weight > 0.0f;    // rank_cmp's branch
  const bool score_weighted = a.freq_weight != 0.0f;  // score()'s branch
  const bool expanded = n && a.any_variants && qex != 0;  // has_expandable_variants
  // ---- sort keys ------------------------------------------------------------------------------------
  SurvRow mine{0.0, 0ull, 0u, 0u, 0u, 0u};  // row gl stays in registers (most lists are shorter than the group)
  for (uint32_t i = gl; i < n; i += G) {
    const SurvRow r = c_rows[seg0 + i];
    if (i == (uint32_t)gl) mine = r;
    double key = r.score;
    if (sort_weighted) {
      const double fs = max_freq > 0.0 ? (double)r.freq / max_freq : (double)r.freq;
      key = result_score(key, fs, a.freq_weight);
    }
    if (i < (uint32_t)LCAP) { s_key[i] = key; s_freq[i] = r.freq; s_ord[i] = r.ord; }
    else t_key[seg0 + i] = key;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  // ---- rank by counting -------------------------------------------------------------------------------
  const bool full = score_weighted || a.max_matches == 0 || expanded;
  const uint32_t M = full ? n : (uint32_t)min((uint64_t)n, a.max_matches + 1);
  for (uint32_t i = gl; i < n; i += G) {
    double ki; uint32_t fi; unsigned long long oi;
    if (i < (uint32_t)LCAP) { ki = s_key[i]; fi = s_freq[i]; oi = s_ord[i]; }
    else { ki = t_key[seg0 + i]; fi = c_rows[seg0 + i].freq; oi = c_rows[seg0 + i].ord; }
    uint32_t rank = 0;
    for (uint32_t j = 0; j < n; ++j) {
      double kj; uint32_t fj; unsigned long long oj;
      if (j < (uint32_t)LCAP) { kj = s_key[j]; fj = s_freq[j]; oj = s_ord[j]; }
      else { kj = t_key[seg0 + j]; fj = c_rows[seg0 + j].freq; oj = c_rows[seg0 + j].ord; }
      bool before;
      if (sort_weighted) before = kj > ki || (kj == ki && oj < oi);
      else before = kj > ki || (kj == ki && (fj > fi || (fj == fi && oj < oi)));
      rank += before;
    }
    if (rank < M) {
      const SurvRow r = i == (uint32_t)gl ? mine : c_rows[seg0 + i];
      const double ff = max_freq > 0.0 ? (double)fi / max_freq : (double)fi;
      r_rows[seg0 + rank] = DevRow{r.vocab, a.any_variants ? r.via : 0xFFFFFFFFu, r.score, ff};
      if (rank < (uint32_t)G) { s_sdist[rank] = r.score; s_sfreq[rank] = ff; }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  const bool parallel_tail = n && !expanded && M <= (uint32_t)G;
  {
    // ---- crop + cutoff, group-parallel (same rules as the serial code below; lane i holds ranked row i) --
    const float fw = a.freq_weight;
    const bool have = parallel_tail && (uint32_t)gl < M;
    const double di = have ? s_sdist[gl] : 0.0;
    const double si = have ? result_score(di, s_sfreq[gl], fw) : 0.0;
    uint32_t len = n;
    const uint64_t mm = a.max_matches;
    const bool crop = parallel_tail && mm > 0 && (uint64_t)n > mm;
    double last = 0.0, cropped = 0.0;
    if (crop) {
      last = result_score(s_sdist[mm - 1], s_sfreq[mm - 1], fw);
      cropped = result_score(s_sdist[mm], s_sfreq[mm], fw);
    }
    // wave-wide ballots (every lane participates), sliced per group
    const unsigned long long lt = (__ballot(have && crop && di < cropped) >> gshift) & gmask;
    const uint32_t stop_at = lt ? (uint32_t)__ffsll((long long)lt) - 1 : (uint32_t)G;  // the loop breaks at the first smaller row
    const unsigned long long eq = (__ballot(have && crop && gl >= 1 && (uint32_t)gl <= stop_at && di == cropped) >> gshift) & gmask;
    if (crop) {
      if (cropped < last) len = (uint32_t)mm;
      else {
        const uint32_t early = eq ? (uint32_t)__ffsll((long long)eq) - 1 : 0;
        const uint32_t late = lt ? stop_at : 0;
        if (early > 0) len = early + 1;
        else if (late > 0) len = late + 1;
      }
    }
    const bool docut = parallel_tail && a.cutoff_threshold >= 1.0;
    const double best = docut ? result_score(s_sdist[0], s_sfreq[0], fw) : 0.0;
    const unsigned long long cut = (__ballot(have && docut && gl >= 1 && (uint32_t)gl < len && si <= best / a.cutoff_threshold) >> gshift) & gmask;
    if (cut) len = (uint32_t)__ffsll((long long)cut) - 1;
    if (parallel_tail && gl == 0) r_count[q] = len;
  }
  // ---- general case: dedup + crop + cutoff, literally, by one lane -------------------------------------
  if (n && !parallel_tail && gl == 0) {
    const float fw = a.freq_weight;
    DevRow* rr = r_rows + seg0;
    uint32_t len = n, avail = M;
    if (expanded) {  // results.dedup_by_key(|x| x.vocab_id): consecutive duplicates, first kept (src/lib.rs:1530-1533)
      uint32_t w = 0;
      for (uint32_t i = 0; i < n; ++i)
        if (w == 0 || rr[w - 1].vocab_id != rr[i].vocab_id) {
          rr[w] = rr[i];
          ++w;
        }
      len = w;
      avail = w;
    }
    const uint64_t mm = a.max_matches;
    if (mm > 0 && (uint64_t)len > mm) {
      const double last = result_score(rr[mm - 1].dist_score, rr[mm - 1].freq_score, fw);
      const double cropped = result_score(rr[mm].dist_score, rr[mm].freq_score, fw);
      if (cropped < last) len = (uint32_t)mm;
      else {
        uint32_t early = 0, late = 0;
        for (uint32_t i = 0; i < avail; ++i) {
          if (rr[i].dist_score == cropped && early == 0) early = i;
          if (rr[i].dist_score < cropped) { late = i; break; }
        }
        if (early > 0) len = early + 1;
        else if (late > 0) len = late + 1;
      }
    }
    uint32_t cutoff = 0;
    if (a.cutoff_threshold >= 1.0) {
      bool have = false;
      double best = 0.0;
      for (uint32_t i = 0; i < len; ++i) {
        const double sc = result_score(rr[i].dist_score, rr[i].freq_score, fw);
        if (have) {
          if (sc <= best / a.cutoff_threshold) { cutoff = i; break; }
        } else { best = sc; have = true; }
      }
    }
    if (cutoff > 0) len = cutoff;
    r_count[q] = len;
  }
}

// Same result, but rows that the cutoff rule (src/lib.rs:1598-1622) is certain to drop are discarded BEFORE the
// O(n^2) rank-by-counting.  The list is sorted by the very key the cutoff tests, so every row with
// key <= best / cutoff_threshold (and key < best) lies behind the first such row and is cut; the crop rule
// (:1536-1589) only ever looks at rows before that point or yields a length beyond it (then the cutoff wins).
// On config 2 the survivors per query are heavy-tailed (mean 10, 2 % above 64 carry half of sum n^2) and most of a
// long list is below half the best score.  nloop: wave-uniform upper bound of n (ballot count must match).
template <int G, int LCAP>
__device__ inline void rank_query(uint32_t q, bool valid, int gl, int gshift, uint32_t nloop, double* __restrict__ s_key,
                                  unsigned long long* __restrict__ s_ord, uint32_t* __restrict__ s_freq,
                                  uint16_t* __restrict__ s_src, double* __restrict__ s_sdist, double* __restrict__ s_sfreq,
                                  uint32_t seg0, uint32_t n, uint32_t maxf, uint32_t qex,
                                  const SurvRow* __restrict__ c_rows, const RankArgs& a, double* __restrict__ t_key,
                                  DevRow* __restrict__ r_rows, uint32_t* __restrict__ r_count) {
  const unsigned long long gmask = G >= 64 ? ~0ull : ((1ull << (G & 63)) - 1ull);
  if (!valid) n = 0;
  const double max_freq = a.have_freq ? (double)maxf : (maxf ? 1.0 : 0.0);
  const bool sort_weighted = a.freq_weight > 0.0f;    // rank_cmp's branch
  const bool score_weighted = a.freq_weight != 0.0f;  // score()'s branch
  const bool expanded = n && a.any_variants && qex != 0;  // has_expandable_variants
  const bool prune = a.cutoff_threshold >= 1.0 && !expanded && (!score_weighted || sort_weighted) && n <= 0xFFFFu;
  auto key_of = [&](const SurvRow& r) {
    if (!sort_weighted) return r.score;
    const double fs = max_freq > 0.0 ? (double)r.freq / max_freq : (double)r.freq;
    return result_score(r.score, fs, a.freq_weight);
  };
  // ---- best key of the group ----------------------------------------------------------------------------
  SurvRow mine{0.0, 0ull, 0u, 0u, 0u, 0u};
  double best = -1.0;
  for (uint32_t i = gl; i < n; i += G) {
    const SurvRow r = c_rows[seg0 + i];
    if (i == (uint32_t)gl) mine = r;
    best = fmax(best, key_of(r));
  }
#pragma unroll
  for (int o = G / 2; o; o >>= 1) best = fmax(best, __shfl_xor(best, o));
  const double thr = best / a.cutoff_threshold;
  // ---- long lists, only max_matches + 1 ranks wanted: tau = the (max_matches+1)-th largest key, by quickselect ----
  // A row with key < tau has at least max_matches+1 rows before it, so it can neither be returned nor influence the
  // crop / cutoff rules (they only look at the first max_matches+1 ranked rows).  Counting is ballot + popcount over
  // the wave; the pivot is the first surviving key strictly inside the current bracket (the list is unsorted, so
  // that is a random pivot).  Needed for d = 3 / long words, where the cutoff rule prunes little (config 3: 1 % of
  // the queries have more than 128 rows and carry 40 % of sum n^2).
  double tau = -1.0;  // keys are >= 0
  if (G == 64 && n > 32 && !(score_weighted || a.max_matches == 0 || expanded)) {
    const uint32_t want = (uint32_t)a.max_matches + 1u;
    double lo = -1.0, hi = __builtin_inf();
    for (int round = 0; round < 96; ++round) {
      double pivot = 0.0;
      bool found = false;
      for (uint32_t base = 0; base < n && !found; base += G) {  // n is wave-uniform in the 64-lane path
        const uint32_t i = base + (uint32_t)gl;
        SurvRow r = mine;
        if (base && i < n) r = c_rows[seg0 + i];
        const double key = key_of(r);
        const bool inr = i < n && !(prune && key <= thr && key < best) && key > lo && key < hi;
        const unsigned long long m = __ballot(inr);
        if (m) {
          const int src = (round & 1) ? 63 - __clzll((long long)m) : __ffsll((long long)m) - 1;  // alternate ends
          pivot = __shfl(key, src);
          found = true;
        }
      }
      if (!found) break;  // nothing strictly inside the bracket
      uint32_t cgt = 0, cge = 0;
      for (uint32_t base = 0; base < n; base += G) {
        const uint32_t i = base + (uint32_t)gl;
        SurvRow r = mine;
        if (base && i < n) r = c_rows[seg0 + i];
        const double key = key_of(r);
        const bool pa = i < n && !(prune && key <= thr && key < best);
        cgt += (uint32_t)__popcll(__ballot(pa && key > pivot));
        cge += (uint32_t)__popcll(__ballot(pa && key >= pivot));
      }
      if (cgt < want && want <= cge) { tau = pivot; break; }
      if (cgt >= want) lo = pivot;
      else hi = pivot;
    }
    // tau, if it exists, always lies strictly inside (lo, hi): an empty bracket means fewer than `want` rows -> keep all
  }
  // ---- keep the rows neither rule can drop, compacted into LDS --------------------------------------------
  uint32_t kept = 0;  // group-uniform
  for (uint32_t base = 0; base < nloop; base += G) {
    const uint32_t i = base + (uint32_t)gl;
    SurvRow r = mine;
    if (base && i < n) r = c_rows[seg0 + i];
    const double key = key_of(r);
    const bool keep = i < n && !(prune && key <= thr && key < best) && key >= tau;
    const unsigned long long m = (__ballot(keep) >> gshift) & gmask;
    const uint32_t pos = kept + (uint32_t)__popcll(m & ((1ull << gl) - 1ull));
    if (keep && pos < (uint32_t)LCAP) { s_key[pos] = key; s_freq[pos] = r.freq; s_ord[pos] = r.ord; s_src[pos] = (uint16_t)i; }
    kept += (uint32_t)__popcll(m);
  }
  if (kept > (uint32_t)LCAP) {  // group-uniform; only the 64-lane path can get here (wave-uniform there)
    rank_query_all<G, LCAP>(q, valid, gl, gshift, s_key, s_ord, s_freq, s_sdist, s_sfreq, seg0, n, maxf, qex, c_rows, a, t_key,
                            r_rows, r_count);
    return;
  }
  n = kept;
  if (valid && n == 0 && gl == 0) r_count[q] = 0;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  // ---- rank by counting -------------------------------------------------------------------------------
  const bool full = score_weighted || a.max_matches == 0 || expanded;
  const uint32_t M = full ? n : (uint32_t)min((uint64_t)n, a.max_matches + 1);
  for (uint32_t i = gl; i < n; i += G) {
    const double ki = s_key[i];
    const uint32_t fi = s_freq[i];
    const unsigned long long oi = s_ord[i];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < n; ++j) {
      const double kj = s_key[j];
      const uint32_t fj = s_freq[j];
      const unsigned long long oj = s_ord[j];
      bool before;
      if (sort_weighted) before = kj > ki || (kj == ki && oj < oi);
      else before = kj > ki || (kj == ki && (fj > fi || (fj == fi && oj < oi)));
      rank += before;
    }
    if (rank < M) {
      const SurvRow r = c_rows[seg0 + s_src[i]];
      const double ff = max_freq > 0.0 ? (double)fi / max_freq : (double)fi;
      r_rows[seg0 + rank] = DevRow{r.vocab, a.any_variants ? r.via : 0xFFFFFFFFu, r.score, ff};
      if (rank < (uint32_t)G) { s_sdist[rank] = r.score; s_sfreq[rank] = ff; }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  const bool parallel_tail = n && !expanded && M <= (uint32_t)G;
  {
    // ---- crop + cutoff, group-parallel (same rules as the serial code below; lane i holds ranked row i) --
    const float fw = a.freq_weight;
    const bool have = parallel_tail && (uint32_t)gl < M;
    const double di = have ? s_sdist[gl] : 0.0;
    const double si = have ? result_score(di, s_sfreq[gl], fw) : 0.0;
    uint32_t len = n;
    const uint64_t mm = a.max_matches;
    const bool crop = parallel_tail && mm > 0 && (uint64_t)n > mm;
    double last = 0.0, cropped = 0.0;
    if (crop) {
      last = result_score(s_sdist[mm - 1], s_sfreq[mm - 1], fw);
      cropped = result_score(s_sdist[mm], s_sfreq[mm], fw);
    }
    // wave-wide ballots (every lane participates), sliced per group
    const unsigned long long lt = (__ballot(have && crop && di < cropped) >> gshift) & gmask;
    const uint32_t stop_at = lt ? (uint32_t)__ffsll((long long)lt) - 1 : (uint32_t)G;  // the loop breaks at the first smaller row
    const unsigned long long eq = (__ballot(have && crop && gl >= 1 && (uint32_t)gl <= stop_at && di == cropped) >> gshift) & gmask;
    if (crop) {
      if (cropped < last) len = (uint32_t)mm;
      else {
        const uint32_t early = eq ? (uint32_t)__ffsll((long long)eq) - 1 : 0;
        const uint32_t late = lt ? stop_at : 0;
        if (early > 0) len = early + 1;
        else if (late > 0) len = late + 1;
      }
    }
    const bool docut = parallel_tail && a.cutoff_threshold >= 1.0;
    const double best = docut ? result_score(s_sdist[0], s_sfreq[0], fw) : 0.0;
    const unsigned long long cut = (__ballot(have && docut && gl >= 1 && (uint32_t)gl < len && si <= best / a.cutoff_threshold) >> gshift) & gmask;
    if (cut) len = (uint32_t)__ffsll((long long)cut) - 1;
    if (parallel_tail && gl == 0) r_count[q] = len;
  }
  // ---- general case: dedup + crop + cutoff, literally, by one lane -------------------------------------
  if (n && !parallel_tail && gl == 0) {
    const float fw = a.freq_weight;
    DevRow* rr = r_rows + seg0;
    uint32_t len = n, avail = M;
    if (expanded) {  // results.dedup_by_key(|x| x.vocab_id): consecutive duplicates, first kept (src/lib.rs:1530-1533)
      uint32_t w = 0;
      for (uint32_t i = 0; i < n; ++i)
        if (w == 0 || rr[w - 1].vocab_id != rr[i].vocab_id) {
          rr[w] = rr[i];
          ++w;
        }
      len = w;
      avail = w;
    }
    const uint64_t mm = a.max_matches;
    if (mm > 0 && (uint64_t)len > mm) {
      const double last = result_score(rr[mm - 1].dist_score, rr[mm - 1].freq_score, fw);
      const double cropped = result_score(rr[mm].dist_score, rr[mm].freq_score, fw);
      if (cropped < last) len = (uint32_t)mm;
      else {
        uint32_t early = 0, late = 0;
        for (uint32_t i = 0; i < avail; ++i) {
          if (rr[i].dist_score == cropped && early == 0) early = i;
          if (rr[i].dist_score < cropped) { late = i; break; }
        }
        if (early > 0) len = early + 1;
        else if (late > 0) len = late + 1;
      }
    }
    uint32_t cutoff = 0;
    if (a.cutoff_threshold >= 1.0) {
      bool have = false;
      double best = 0.0;
      for (uint32_t i = 0; i < len; ++i) {
        const double sc = result_score(rr[i].dist_score, rr[i].freq_score, fw);
        if (have) {
          if (sc <= best / a.cutoff_threshold) { cutoff = i; break; }
        } else { best = sc; have = true; }
      }
    }
    if (cutoff > 0) len = cutoff;
    r_count[q] = len;
  }
}

// A wave owns 4 consecutive queries.  If none of them has more than 16 candidate rows (the common case: ~10 per
// query on config 2) the four are ranked side by side by 16 lanes each; otherwise one after the other by the whole
// wave (lists up to RANK_LCAP rows in LDS, longer ones through t_key).  1M one-query waves were latency-bound.
constexpr int RANK_QPW = 4;                                    // queries per wave
constexpr int RANK_WAVE_BYTES = RANK_LCAP * 22 + 64 * 16;      // LDS per wave: keys, order keys, freqs, source rows + ranked heads
__global__ __launch_bounds__(256) void k_rank(uint32_t nq, const uint32_t* __restrict__ soff,
                                              const SurvRow* __restrict__ c_rows,
                                              const uint32_t* __restrict__ qmaxfreq,
                                              const uint32_t* __restrict__ qexpand, RankArgs a,
                                              double* __restrict__ t_key, DevRow* __restrict__ r_rows,
                                              uint32_t* __restrict__ r_count) {
  __shared__ __attribute__((aligned(16))) uint8_t s_raw[4 * RANK_WAVE_BYTES];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint8_t* wl = s_raw + wid * RANK_WAVE_BYTES;
  const uint32_t qbase = (blockIdx.x * 4 + wid) * RANK_QPW;
  // lanes 0..3 fetch the four segments; everybody reads them back with shuffles
  uint32_t my_seg0 = 0, my_n = 0, my_maxf = 0, my_qex = 0;
  if (lane < RANK_QPW && qbase + lane < nq) {
    my_seg0 = soff[qbase + lane];
    my_n = soff[qbase + lane + 1] - my_seg0;
    my_maxf = qmaxfreq[qbase + lane];
    if (a.any_variants) my_qex = qexpand[qbase + lane];
  }
  uint32_t nmax = my_n;
  nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, 1));
  nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, 2));
  nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmax);
  if (nmax <= 16) {
    const int grp = lane >> 4, gl = lane & 15;
    // per group: 16 keys (8 B), 16 order keys (8 B), 16 freqs (4 B), 16 source rows (2 B), 16 + 16 ranked heads (8 B) = 608 B
    uint8_t* gb = wl + grp * 608;
    rank_query<16, 16>(qbase + grp, qbase + grp < nq, gl, grp * 16, 16u, reinterpret_cast<double*>(gb),
                       reinterpret_cast<unsigned long long*>(gb + 128), reinterpret_cast<uint32_t*>(gb + 256),
                       reinterpret_cast<uint16_t*>(gb + 320), reinterpret_cast<double*>(gb + 352), reinterpret_cast<double*>(gb + 480),
                       (uint32_t)__shfl((int)my_seg0, grp), (uint32_t)__shfl((int)my_n, grp), (uint32_t)__shfl((int)my_maxf, grp),
                       (uint32_t)__shfl((int)my_qex, grp), c_rows, a, t_key, r_rows, r_count);
  } else {
    for (int k = 0; k < RANK_QPW; ++k) {
      if (qbase + k >= nq) break;  // wave-uniform
      const uint32_t nk = (uint32_t)__builtin_amdgcn_readlane((int)my_n, k);
      rank_query<64, RANK_LCAP>(qbase + k, true, lane, 0, nk, reinterpret_cast<double*>(wl),
                                reinterpret_cast<unsigned long long*>(wl + RANK_LCAP * 8),
                                reinterpret_cast<uint32_t*>(wl + RANK_LCAP * 16), reinterpret_cast<uint16_t*>(wl + RANK_LCAP * 20),
                                reinterpret_cast<double*>(wl + RANK_LCAP * 22), reinterpret_cast<double*>(wl + RANK_LCAP * 22 + 512),
                                (uint32_t)__shfl((int)my_seg0, k), nk, (uint32_t)__shfl((int)my_maxf, k), (uint32_t)__shfl((int)my_qex, k),
                                c_rows, a, t_key, r_rows, r_count);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
  }
}

// dense result rows (device) for download / gather
__global__ __launch_bounds__(256) void k_pack_rows(uint32_t nq, const uint32_t* __restrict__ soff,
                                                   const uint32_t* __restrict__ r_off,
                                                   const uint32_t* __restrict__ r_count,
                                                   const DevRow* __restrict__ r_rows, DevRow* __restrict__ out) {
  const uint32_t q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  const uint32_t n = r_count[q], src = soff[q], dst = r_off[q];
  for (uint32_t i = 0; i < n; ++i) out[dst + i] = r_rows[src + i];
}
__global__ __launch_bounds__(256) void k_export_topk(uint32_t nq, uint32_t stride, const uint32_t* __restrict__ soff,
                                                     const uint32_t* __restrict__ r_count,
                                                     const DevRow* __restrict__ r_rows,
                                                     const uint32_t* __restrict__ q_orig,
                                                     anx_topk_record* __restrict__ out) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (uint64_t)nq * stride) return;
  const uint32_t q = (uint32_t)(t / stride), i = (uint32_t)(t % stride);
  anx_topk_record r;
  r.vocab_id = 0xFFFFFFFFu;
  r.freq_score = 0.0f;
  r.dist_score = 0.0;
  if (i < r_count[q]) {
    const DevRow d = r_rows[soff[q] + i];
    r.vocab_id = d.vocab_id;
    r.freq_score = (float)d.freq_score;
    r.dist_score = d.dist_score;
  }
  out[(size_t)q_orig[q] * stride + i] = r;
}

// ------------------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------------------
int device_count(std::string& err) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    err = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
    return 0;
  }
  return n;
}

// Device memory pool.  A batch needs several GB of scratch (pair list, per-slot arrays, survivor rows); hipMalloc /
// hipFree of that size cost 100-300 ms per call, far more than the 6 ms the kernels take for a million queries, and
// search mode issues one batch per n-gram order.  Freed blocks are kept per device and handed out again (best fit,
// at most 2x the request); lexicon_free() of the last lexicon on a device returns them to the driver.
namespace {
struct DevPool {
  std::mutex mu;
  std::multimap<size_t, void*> free_blocks;        // size -> block
  std::unordered_map<void*, size_t> size_of;       // every live or cached block of this pool
  size_t cached = 0;
  int lexicons = 0;
};
constexpr size_t POOL_CACHE_LIMIT = (size_t)96 << 30;  // bytes kept per device (MI355X: 288 GB HBM)
DevPool& pool_of(int device) {
  static DevPool pools[64];
  return pools[device >= 0 && device < 64 ? device : 0];
}
hipError_t pool_malloc(void** p, size_t bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  DevPool& pl = pool_of(dev);
  bytes = (bytes + 255) & ~(size_t)255;
  {
    std::lock_guard<std::mutex> g(pl.mu);
    auto it = pl.free_blocks.lower_bound(bytes);
    if (it != pl.free_blocks.end() && it->first <= 2 * bytes + (1u << 20)) {
      *p = it->second;
      pl.cached -= it->first;
      pl.free_blocks.erase(it);
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) {  // out of memory: give the cache back and retry once
    std::vector<void*> drop;
    {
      std::lock_guard<std::mutex> g(pl.mu);
      for (auto& kv : pl.free_blocks) { drop.push_back(kv.second); pl.size_of.erase(kv.second); }
      pl.free_blocks.clear();
      pl.cached = 0;
    }
    for (void* d : drop) (void)hipFree(d);
    (void)hipGetLastError();
    e = hipMalloc(p, bytes);
    if (e != hipSuccess) return e;
  }
  std::lock_guard<std::mutex> g(pl.mu);
  pl.size_of[*p] = bytes;
  return hipSuccess;
}
void pool_free(void* p) {
  if (!p) return;
  int dev = 0;
  (void)hipGetDevice(&dev);
  DevPool& pl = pool_of(dev);
  {
    std::lock_guard<std::mutex> g(pl.mu);
    auto it = pl.size_of.find(p);
    if (it != pl.size_of.end() && pl.cached + it->second <= POOL_CACHE_LIMIT) {
      pl.free_blocks.emplace(it->second, p);
      pl.cached += it->second;
      return;
    }
    if (it != pl.size_of.end()) pl.size_of.erase(it);
  }
  (void)hipFree(p);
}
void pool_trim(int device) {
  DevPool& pl = pool_of(device);
  std::vector<void*> drop;
  {
    std::lock_guard<std::mutex> g(pl.mu);
    for (auto& kv : pl.free_blocks) { drop.push_back(kv.second); pl.size_of.erase(kv.second); }
    pl.free_blocks.clear();
    pl.cached = 0;
  }
  for (void* d : drop) (void)hipFree(d);
}
}  // namespace

template <typename T>
static int upload(T** dst, const void* src, size_t count, std::string& err, size_t* total) {
  const size_t bytes = std::max<size_t>(count * sizeof(T), 16);
  HIP_TRY(pool_malloc(reinterpret_cast<void**>(dst), bytes));
  if (count) HIP_TRY(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
  if (total) *total += bytes;
  return ANX_OK;
}
template <typename T>
static int dalloc(T** dst, size_t count, std::string& err) {
  HIP_TRY(pool_malloc(reinterpret_cast<void**>(dst), std::max<size_t>(count * sizeof(T), 16)));
  return ANX_OK;
}

DeviceLexicon* lexicon_upload(const LexiconImage& img, int device, std::string& err) {
  int n = device_count(err);
  if (n <= 0) {
    if (err.empty()) err = "no HIP device available";
    return nullptr;
  }
  if (device < 0 || device >= n) { err = "invalid device ordinal"; return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { err = "hipSetDevice failed"; return nullptr; }
  DeviceLexicon* d = new DeviceLexicon();
  d->device = device;
  { DevPool& pl = pool_of(device); std::lock_guard<std::mutex> g(pl.mu); ++pl.lexicons; }
  d->nplanes = img.nplanes;
  d->nsym = img.nsym;
  d->nclasses = img.nclasses;
  d->nentries = img.nentries;
  d->cstride = img.cstride;
  d->any_variants = img.any_variants ? 1 : 0;
  uint32_t maxlen = 1;
  for (uint32_t m : img.ent_meta) maxlen = std::max(maxlen, m & 0xFFu);
  d->max_len = maxlen;
  int rc = ANX_OK;
  std::vector<EntRec> rec(img.ent_vocab.size());
  for (size_t e = 0; e < rec.size(); ++e) rec[e] = EntRec{img.ent_vocab[e], img.ent_freq[e], img.ent_order[e], img.ent_meta[e]};
  std::vector<uint2> sig2(img.sig_lo.size());
  for (size_t i = 0; i < sig2.size(); ++i) sig2[i] = make_uint2(img.sig_lo[i], img.sig_hi[i]);
  std::vector<uint32_t> off = img.cls_off;
  if (off.empty()) off.push_back(0);
  if ((rc = upload(&d->cls_planes, img.cls_planes.data(), img.cls_planes.size(), err, &d->bytes)) ||
      (rc = upload(&d->cls_bits, img.cls_bits.data(), img.cls_bits.size(), err, &d->bytes)) ||
      (rc = upload(&d->cls_len, img.cls_len.data(), img.cls_len.size(), err, &d->bytes)) ||
      (rc = upload(&d->cls_off, off.data(), off.size(), err, &d->bytes)) ||
      (rc = upload(&d->sig, sig2.data(), sig2.size(), err, &d->bytes)) ||
      (rc = upload(&d->sig_cbeg, img.sig_cbeg.data(), img.sig_cbeg.size(), err, &d->bytes)) ||
      (rc = upload(&d->ent_vocab, img.ent_vocab.data(), img.ent_vocab.size(), err, &d->bytes)) ||
      (rc = upload(&d->ent_freq, img.ent_freq.data(), img.ent_freq.size(), err, &d->bytes)) ||
      (rc = upload(&d->ent_meta, img.ent_meta.data(), img.ent_meta.size(), err, &d->bytes)) ||
      (rc = upload(&d->ent_rowoff, img.ent_rowoff.data(), img.ent_rowoff.size(), err, &d->bytes)) ||
      (rc = upload(&d->ent_order, img.ent_order.data(), img.ent_order.size(), err, &d->bytes)) ||
      (rc = upload(&d->ent_rec, rec.data(), rec.size(), err, &d->bytes)) ||
      (rc = upload(&d->ent_var_off, img.ent_var_off.data(), img.ent_var_off.size(), err, &d->bytes)) ||
      (rc = upload(&d->var_target, img.var_target.data(), img.var_target.size(), err, &d->bytes)) ||
      (rc = upload(&d->var_target_freq, img.var_target_freq.data(), img.var_target_freq.size(), err, &d->bytes)) ||
      (rc = upload(&d->var_score, img.var_score.data(), img.var_score.size(), err, &d->bytes)) ||
      (rc = upload(reinterpret_cast<uint8_t**>(&d->rows), img.rows.data(), img.rows.size(), err, &d->bytes))) {
    lexicon_free(d);
    return nullptr;
  }
  return d;
}

void lexicon_free(DeviceLexicon* d) {
  if (!d) return;
  (void)hipSetDevice(d->device);
  for (void* p : {(void*)d->cls_planes, (void*)d->cls_bits, (void*)d->cls_len, (void*)d->cls_off, (void*)d->sig, (void*)d->sig_cbeg, (void*)d->ent_vocab,
                  (void*)d->ent_freq, (void*)d->ent_meta, (void*)d->ent_rowoff, (void*)d->ent_order, (void*)d->ent_rec, (void*)d->ent_var_off,
                  (void*)d->var_target, (void*)d->var_target_freq, (void*)d->var_score, (void*)d->rows})
    if (p) pool_free(p);
  bool last;
  { DevPool& pl = pool_of(d->device); std::lock_guard<std::mutex> g(pl.mu); last = --pl.lexicons <= 0; }
  if (last) pool_trim(d->device);  // the last model of this device: hand the cached blocks back to the driver
  delete d;
}

static int scan_mode() {  // ANX_SCAN=sad forces the general count-vector kernel (A/B testing)
  const char* e = getenv("ANX_SCAN");
  return (e && strcmp(e, "sad") == 0) ? 1 : 0;
}

Batch* batch_encode(const HostModel& m, const DeviceLexicon* dl, const char* const* utf8, size_t n,
                    const anx_params& p, std::string& err, int* code) {
  *code = ANX_OK;
  if (!dl) { err = "model is not resident on a device (no HIP device / anx_model_to_device not called)"; *code = ANX_ENODEVICE; return nullptr; }
  if (hipSetDevice(dl->device) != hipSuccess) { err = "hipSetDevice failed"; *code = ANX_ENODEVICE; return nullptr; }
  static const bool timing = getenv("ANX_ENCODE_TIMING") != nullptr;
  auto tnow = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_prev = tnow();
  auto lap = [&](const char* what) { if (timing) { const double t = tnow(); fprintf(stderr, "[anx encode] %-28s %8.2f ms\n", what, (t - t_prev) * 1e3); t_prev = t; } };
  Batch* b = new Batch();
  b->device = dl->device;
  b->params = p;
  b->n_input = n;
  b->status.assign(n, 0);
  const int NP = dl->nplanes;
  const bool bits_ok = dl->nsym <= 32 && !scan_mode();
  // ---- host encoding, threaded: normalisation (src/anahash.rs:50-80), count vector, threshold clamps -------
  struct Enc {
    uint32_t meta;   // len | k<<8 | d<<16 | first_is_lower<<24 ; 0 = not encodable
    uint32_t key;    // kind*256 + len : bucket for the counting sort
    uint32_t off;    // offset of the norm string in its thread's arena
    uint16_t thread;
    uint64_t sig;    // per-group symbol counts (LexiconImage::sym_group)
  };
  std::vector<Enc> enc(n);
  unsigned nthreads = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
  if (n < 4096) nthreads = 1;
  std::vector<std::vector<uint8_t>> arena(nthreads);
  const int A = m.alphabet.size();
  const size_t cvbytes = (size_t)NP * 4;
  std::vector<uint8_t> cv_all(n * cvbytes, 0);
  auto encode_range = [&](unsigned tid, size_t lo, size_t hi) {
    std::vector<uint8_t>& ar = arena[tid];
    ar.reserve((hi - lo) * 12);
    int16_t codes[kMaxSymbols];
    for (size_t i = lo; i < hi; ++i) {
      Enc& e = enc[i];
      e.meta = 0; e.key = 0; e.off = 0; e.thread = (uint16_t)tid; e.sig = 0;
      const int len = utf8[i] ? m.alphabet.scan_into(utf8[i], strlen(utf8[i]), codes, kMaxSymbols) : -1;
      if (len < 0) { b->status[i] = ANX_ELIMIT; continue; }
      if (len == 0) { b->status[i] = ANX_EEMPTY; continue; }
      uint8_t* cv = &cv_all[i * cvbytes];
      e.off = (uint32_t)ar.size();
      uint32_t maxcount = 0;
      for (int s = 0; s < len; ++s) {
        ar.push_back((uint8_t)(codes[s] >= 0 ? codes[s] : A + 1));                     // src/anahash.rs:76
        const uint8_t c = ++cv[(size_t)(codes[s] >= 0 ? codes[s] : A)];                // src/anahash.rs:42
        maxcount = std::max<uint32_t>(maxcount, c);
      }
      const int k = clamp_threshold(p.max_anagram_distance, len, kMaxAnagramDistance);
      const int d = clamp_threshold(p.max_edit_distance, len, kMaxEditDistance);
      e.meta = (uint32_t)len | ((uint32_t)k << 8) | ((uint32_t)d << 16) |
               (first_char_is_lowercase(utf8[i]) ? 1u << 24 : 0u);
      const uint32_t kind = (bits_ok && maxcount <= (uint32_t)NBITPLANES) ? maxcount : 0;
      e.key = kind * 256 + (uint32_t)len;
      e.sig = signature_of(cv, cvbytes, m.lex.sym_group);
    }
  };
  {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthreads; ++t) {
      const size_t lo = n * t / nthreads, hi = n * (t + 1) / nthreads;
      if (nthreads == 1) encode_range(0, lo, hi);
      else th.emplace_back(encode_range, t, lo, hi);
    }
    for (auto& x : th) x.join();
  }
  lap("normalise + count vectors");
  // (kernel kind, length)-bucketed order, stable (counting sort): a bucket shares k, d and the class window
  constexpr uint32_t NKEYS = (NBITPLANES + 1) * 256;
  std::vector<size_t> kstart(NKEYS + 1, 0);
  size_t maxlen = 1;
  for (size_t i = 0; i < n; ++i)
    if (enc[i].meta) {
      kstart[enc[i].key + 1]++;
      maxlen = std::max<size_t>(maxlen, enc[i].meta & 0xFF);
      b->dmax = std::max<uint32_t>(b->dmax, (enc[i].meta >> 16) & 0xFF);
    }
  for (uint32_t kx = 0; kx < NKEYS; ++kx) kstart[kx + 1] += kstart[kx];
  const size_t nq = kstart[NKEYS];
  b->nq = nq;
  b->qw = (uint32_t)((maxlen + 15) / 16);
  std::vector<uint32_t> h_cv(nq * (size_t)NP, 0), h_bits(nq * (size_t)NBITPLANES, 0), h_meta(nq), h_orig(nq), h_kind(nq);
  std::vector<uint64_t> h_sig(nq);
  std::vector<uint8_t> h_rows(nq * (size_t)b->qw * 16, 0xFE);
  b->order.resize(nq);
  {
    std::vector<size_t> cursor(kstart.begin(), kstart.end() - 1);
    for (size_t i = 0; i < n; ++i)
      if (enc[i].meta) b->order[cursor[enc[i].key]++] = (uint32_t)i;
  }
  lap("counting sort");
  {  // inside a (kind, length) bucket: by signature, stable; buckets are independent -> threads take them round-robin
    auto sort_buckets = [&](unsigned tid) {
      for (uint32_t kx = tid; kx < NKEYS; kx += nthreads)
        if (kstart[kx + 1] - kstart[kx] > 1)
          std::stable_sort(b->order.begin() + (ptrdiff_t)kstart[kx], b->order.begin() + (ptrdiff_t)kstart[kx + 1],
                           [&](uint32_t x, uint32_t y) { return enc[x].sig < enc[y].sig; });
    };
    std::vector<std::thread> th;
    if (nthreads == 1) sort_buckets(0);
    else
      for (unsigned t = 0; t < nthreads; ++t) th.emplace_back(sort_buckets, t);
    for (auto& x : th) x.join();
  }
  lap("signature sort");
  auto fill_range = [&](size_t lo, size_t hi) {
    for (size_t s = lo; s < hi; ++s) {
      const size_t i = b->order[s];
      const Enc& e = enc[i];
      const uint8_t* cv = &cv_all[i * cvbytes];
      memcpy(&h_cv[s * (size_t)NP], cv, cvbytes);
      for (size_t sym = 0; sym < cvbytes && sym < 32; ++sym)
        for (uint32_t tp = 0; tp < (uint32_t)NBITPLANES; ++tp)
          if (cv[sym] > tp) h_bits[s * NBITPLANES + tp] |= 1u << sym;
      memcpy(&h_rows[s * (size_t)b->qw * 16], &arena[e.thread][e.off], e.meta & 0xFF);
      h_meta[s] = e.meta;
      h_orig[s] = (uint32_t)i;
      h_kind[s] = e.key >> 8;
      h_sig[s] = e.sig;
    }
  };
  {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthreads; ++t) {
      const size_t lo = nq * t / nthreads, hi = nq * (t + 1) / nthreads;
      if (nthreads == 1) fill_range(lo, hi);
      else th.emplace_back(fill_range, lo, hi);
    }
    for (auto& x : th) x.join();
  }
  lap("fill device images");
  // tiles: <= SCAN_TQ queries of one kind, length and signature; one wave of k_scan each
  for (size_t i = 0; i < nq;) {
    size_t j = i;
    while (j < nq && h_kind[j] == h_kind[i] && (h_meta[j] & 0xFF) == (h_meta[i] & 0xFF) && h_sig[j] == h_sig[i]) ++j;
    const uint32_t kind = h_kind[i], lq = h_meta[i] & 0xFF, k = (h_meta[i] >> 8) & 0xFF;
    const int lo = std::max<int>(1, (int)lq - (int)k), hi = std::min<int>(kMaxSymbols, (int)lq + (int)k);
    const uint32_t s0 = m.lex.siglen_begin[lo], s1 = m.lex.siglen_begin[hi + 1];
    static const uint32_t tq = []() { const char* e = getenv("ANX_SCAN_TQ"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= (int)SCAN_TQ ? (uint32_t)v : SCAN_TQ; }();
    for (size_t s = i; s < j; s += tq)
      b->tiles.push_back(Tile{(uint32_t)s, (uint32_t)std::min<size_t>(tq, j - s), s0, s1, k, lq, (uint32_t)h_sig[i],
                              (uint32_t)(h_sig[i] >> 32), kind});
    i = j;
  }
  // longest-processing-time-first: cost ~ queries (the compatible classes per query vary little inside a length)
  // bit-plane tiles first, the SAD tiles (kind 0) after them: two launches
  std::stable_sort(b->tiles.begin(), b->tiles.end(), [](const Tile& x, const Tile& y) {
    if ((x.kind == 0) != (y.kind == 0)) return y.kind == 0;
    return (uint64_t)x.nq * (x.s1 - x.s0 + 64) > (uint64_t)y.nq * (y.s1 - y.s0 + 64);
  });
  for (const Tile& t : b->tiles) b->n_sad_tiles += t.kind == 0;
  lap("tiles + LPT order");
  std::vector<uint32_t> h_xcls(nq, 0xFFFFFFFFu);
  if (p.stop_at_exact_match) {  // the exact anagram class of every query (the index lookup of src/lib.rs:1164-1173)
    auto lookup_range = [&](size_t lo, size_t hi) {
      std::string key(cvbytes, '\0');
      for (size_t s = lo; s < hi; ++s) {
        memcpy(&key[0], &cv_all[(size_t)b->order[s] * cvbytes], cvbytes);
        auto it = m.class_of_cv.find(key);
        if (it != m.class_of_cv.end()) h_xcls[s] = it->second;
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthreads; ++t) {
      const size_t lo = nq * t / nthreads, hi = nq * (t + 1) / nthreads;
      if (nthreads == 1) lookup_range(lo, hi);
      else th.emplace_back(lookup_range, lo, hi);
    }
    for (auto& x : th) x.join();
  }
  int rc;
  if ((rc = upload(&b->qexact, h_xcls.data(), nq, err, nullptr)) ||
      (rc = upload(&b->q_cv, h_cv.data(), h_cv.size(), err, nullptr)) ||
      (rc = upload(&b->q_bits, h_bits.data(), h_bits.size(), err, nullptr)) ||
      (rc = upload(reinterpret_cast<uint8_t**>(&b->q_rows), h_rows.data(), h_rows.size(), err, nullptr)) ||
      (rc = upload(&b->q_meta, h_meta.data(), nq, err, nullptr)) || (rc = upload(&b->q_orig, h_orig.data(), nq, err, nullptr)) ||
      (rc = upload(&b->d_tiles, b->tiles.data(), b->tiles.size(), err, nullptr))) {
    *code = rc;
    batch_free(b);
    return nullptr;
  }
  lap("uploads");
  const size_t nblk = (nq + SCAN_TILE - 1) / SCAN_TILE + 2;
  if ((rc = dalloc(&b->counters, CTR_N, err)) || (rc = dalloc(&b->rctr, SCAN_REGIONS * RC_STRIDE, err)) || (rc = dalloc(&b->sctr, SCAN_REGIONS * RC_STRIDE, err)) || (rc = dalloc(&b->lctr, 2 * SCAN_REGIONS * RC_STRIDE, err)) || (rc = dalloc(&b->quot, 33 * 33, err)) || (rc = dalloc(&b->qexpand, nq, err)) || (rc = dalloc(&b->qsurv, nq, err)) ||
      (rc = dalloc(&b->soff, nq + 1, err)) || (rc = dalloc(&b->qcur, nq, err)) || (rc = dalloc(&b->qmaxfreq, nq, err)) ||
      (rc = dalloc(&b->scan_tmp, nblk, err)) || (rc = dalloc(&b->r_count, nq, err)) || (rc = dalloc(&b->r_off, nq + 1, err))) {
    *code = rc;
    batch_free(b);
    return nullptr;
  }
  {
    std::vector<double> quot(33 * 33, 0.0);
    for (int x = 0; x <= 32; ++x)
      for (int L = 1; L <= 32; ++L) {
        volatile double num = (double)x, den = (double)L;  // a real division at run time, as the reference does
        quot[(size_t)x * 33 + L] = num / den;
      }
    if (hipMemcpy(b->quot, quot.data(), quot.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { err = "hipMemcpy failed"; *code = ANX_ENODEVICE; batch_free(b); return nullptr; }
  }
  for (auto& e : b->ev)
    if (hipEventCreate(&e) != hipSuccess) { err = "hipEventCreate failed"; *code = ANX_ENODEVICE; batch_free(b); return nullptr; }
  if (hipEventCreate(&b->ev_scan0) != hipSuccess) { err = "hipEventCreate failed"; *code = ANX_ENODEVICE; batch_free(b); return nullptr; }
  lap("device allocations");
  return b;
}

static int exclusive_scan(const uint32_t* in, uint32_t n, uint32_t* out, uint32_t* tmp, hipStream_t st) {
  const uint32_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nb == 0) return hipMemsetAsync(out, 0, sizeof(uint32_t), st) == hipSuccess ? 0 : -1;
  hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(SCAN_THREADS), 0, st, in, n, out, tmp);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_THREADS), 0, st, tmp, nb);
  hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(SCAN_THREADS), 0, st, out, n, tmp, nb);
  return 0;
}

template <int NP>
static void launch_scan(ScanArgs A, uint32_t nbits, uint32_t nsad, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {  // tiles: [bit-plane kinds | SAD kind]
  (void)hipEventRecord(e0, st);  // e0 .. e1 = k_scan_bits alone (anx_batch_stats.ms_scan_kernel)
  if (nbits) {
    A.ntiles = nbits;
    hipLaunchKernelGGL((k_scan_bits<NP>), dim3((nbits + 3) / 4), dim3(256), 0, st, A);
  }
  (void)hipEventRecord(e1, st);
  if (nsad) {
    A.tiles += nbits;
    A.ntiles = nsad;
    hipLaunchKernelGGL((k_scan_sad<NP>), dim3((nsad + 3) / 4), dim3(256), 0, st, A);
  }
}

static int ensure_raw(Batch* b, size_t slots_per_region, std::string& err) {
  uint32_t shift = 10;
  while (((size_t)1 << shift) < slots_per_region) ++shift;
  if (shift > 25) { err = "pair list exceeds 2^31 slots: split the batch"; return ANX_ELIMIT; }
  const size_t cap = (size_t)SCAN_REGIONS << shift;
  if (cap <= b->raw_cap) return ANX_OK;
  for (void* p : {(void*)b->raw, (void*)b->p_score, (void*)b->p_meta})
    if (p) pool_free(p);
  b->raw = nullptr; b->p_score = nullptr; b->p_meta = nullptr; b->raw_cap = 0;
  int rc;
  if ((rc = dalloc(&b->raw, cap, err)) || (rc = dalloc(&b->p_score, cap, err)) || (rc = dalloc(&b->p_meta, cap, err))) return rc;
  b->raw_cap = cap;
  b->region_shift = shift;
  return ANX_OK;
}
static int ensure_surv(Batch* b, size_t cap, std::string& err) {
  if (cap <= b->surv_cap) return ANX_OK;
  for (void* p : {(void*)b->c_rows, (void*)b->r_rows, (void*)b->t_key})
    if (p) pool_free(p);
  b->c_rows = nullptr;
  b->r_rows = nullptr;
  b->t_key = nullptr;
  b->surv_cap = 0;
  int rc;
  if ((rc = dalloc(&b->c_rows, cap, err)) || (rc = dalloc(&b->r_rows, cap, err)) || (rc = dalloc(&b->t_key, cap, err))) return rc;
  b->surv_cap = cap;
  return ANX_OK;
}

int batch_run(const HostModel& m, const DeviceLexicon* dl, Batch* b, void* stream, std::string& err) {
  if (!dl) { err = "model is not resident on a device"; return ANX_ENODEVICE; }
  HIP_TRY(hipSetDevice(dl->device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const uint32_t nq = (uint32_t)b->nq;
  b->ran = false;
  b->n_pairs = b->n_results = b->n_surv = 0;
  b->n_raw = 0;
  if (nq == 0) { b->ran = true; return ANX_OK; }
  const int stop = b->params.stop_at_exact_match ? 1 : 0;
  int rc;
  if (b->raw_cap == 0 && (rc = ensure_raw(b, (nq * (size_t)140 + b->tiles.size() * SCAN_CHUNK) / SCAN_REGIONS + 4096, err)))
    return rc;
  uint32_t h_counters[CTR_N];
  std::vector<uint32_t> h_rctr(SCAN_REGIONS * RC_STRIDE);
  HIP_TRY(hipEventRecord(b->ev[0], st));
  // ---- scan ------------------------------------------------------------------------------------------
  uint32_t maxfill = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    HIP_TRY(hipMemsetAsync(b->counters, 0, CTR_N * sizeof(uint32_t), st));
    HIP_TRY(hipMemsetAsync(b->rctr, 0, SCAN_REGIONS * RC_STRIDE * sizeof(uint32_t), st));
    if (b->tiles.empty()) { HIP_TRY(hipEventRecord(b->ev_scan0, st)); HIP_TRY(hipEventRecord(b->ev[5], st)); }
    if (!b->tiles.empty()) {
      ScanArgs A;
      A.tiles = b->d_tiles; A.ntiles = (uint32_t)b->tiles.size(); A.q_bits = b->q_bits; A.q_cv = b->q_cv;
      A.cls_bits = dl->cls_bits; A.cls_planes = dl->cls_planes; A.cstride = dl->cstride; A.pad_class = dl->nclasses;
      A.cls_len = dl->cls_len; A.cls_off = dl->cls_off; A.sig = dl->sig; A.sig_cbeg = dl->sig_cbeg;
      A.raw = b->raw; A.region_cap = 1u << b->region_shift; A.rctr = b->rctr; A.qexact = b->qexact; A.want_exact = stop;
      { static const int dbg = []() { const char* e = getenv("ANX_SCAN_DBG"); return e ? atoi(e) : 0; }(); A.dbg = dbg; }
      const uint32_t nsad = b->n_sad_tiles, nbits = A.ntiles - nsad;
      switch (dl->nplanes) {
        case 8: launch_scan<8>(A, nbits, nsad, st, b->ev_scan0, b->ev[5]); break;
        case 16: launch_scan<16>(A, nbits, nsad, st, b->ev_scan0, b->ev[5]); break;
        case 24: launch_scan<24>(A, nbits, nsad, st, b->ev_scan0, b->ev[5]); break;
        case 32: launch_scan<32>(A, nbits, nsad, st, b->ev_scan0, b->ev[5]); break;
        default: launch_scan<42>(A, nbits, nsad, st, b->ev_scan0, b->ev[5]); break;
      }
    }
    HIP_TRY(hipMemcpyAsync(h_rctr.data(), b->rctr, h_rctr.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    maxfill = 0;
    for (uint32_t r = 0; r < SCAN_REGIONS; ++r) maxfill = std::max(maxfill, h_rctr[r * RC_STRIDE + RC_RAW]);
    if (maxfill <= (1u << b->region_shift)) break;
    if (attempt == 1) { err = "pair list overflow after regrow"; return ANX_ENODEVICE; }
    if ((rc = ensure_raw(b, (size_t)maxfill + (maxfill >> 3) + 4096, err))) return rc;
  }
  HIP_TRY(hipEventRecord(b->ev[1], st));
  uint64_t n_valid = 0, n_slots = 0;
  b->n_class_tests = 0;
  for (int i = 0; i <= NBITPLANES; ++i) b->n_tests_kind[i] = 0;
  for (uint32_t r = 0; r < SCAN_REGIONS; ++r) {
    const uint32_t* c = &h_rctr[r * RC_STRIDE];
    b->region_fill[r] = c[RC_RAW];
    n_slots += c[RC_RAW];
    n_valid += c[RC_VALID];
    for (int i = 0; i <= NBITPLANES; ++i) {
      uint64_t v;
      memcpy(&v, c + RC_TESTS + 2 * i, sizeof v);
      b->n_tests_kind[i] += v;
      b->n_class_tests += v;
    }
  }
  const uint32_t nraw = maxfill ? (uint32_t)(SCAN_REGIONS << b->region_shift) : 0;  // slot space (regions are sparse)
  b->n_raw = nraw;
  // ---- score -----------------------------------------------------------------------------------------
  HIP_TRY(hipMemsetAsync(b->qsurv, 0, nq * sizeof(uint32_t), st));
  HIP_TRY(hipMemsetAsync(b->qcur, 0, nq * sizeof(uint32_t), st));
  HIP_TRY(hipMemsetAsync(b->qmaxfreq, 0, nq * sizeof(uint32_t), st));
  if (dl->any_variants) HIP_TRY(hipMemsetAsync(b->qexpand, 0, nq * sizeof(uint32_t), st));
  ScoreArgs sa;
  { static const int dbg = []() { const char* e = getenv("ANX_SCORE_DBG"); return e ? atoi(e) : 0; }(); sa.dbg = dbg; }
  sa.quot = b->quot;
  sa.w_ld = m.weights.ld; sa.w_lcs = m.weights.lcs; sa.w_prefix = m.weights.prefix; sa.w_suffix = m.weights.suffix;
  sa.w_case = m.weights.casew;
  sa.w_sum = m.weights.ld + m.weights.lcs + m.weights.prefix + m.weights.suffix + m.weights.casew;  // src/types.rs:69-73
  sa.score_threshold = b->params.score_threshold;
  sa.have_freq = m.have_freq ? 1 : 0;
  sa.any_variants = dl->any_variants;
  sa.lqp = b->qw * 16;
  sa.lcp = (dl->max_len + 15) / 16 * 16;
  const uint32_t d = b->dmax;
  uint32_t stride = sa.lqp + sa.lcp + (d + 2) * (2 * d + 3);
  stride = (stride + 3) / 4;
  if ((stride & 1) == 0) stride++;
  sa.stride = stride * 4;
  sa.qw = b->qw;
  uint32_t threads = 256;
  while (threads > 64 && (size_t)threads * sa.stride > 64 * 1024) threads >>= 1;
  if ((size_t)threads * sa.stride > 64 * 1024) { err = "per-lane scoring state exceeds the LDS budget"; return ANX_ELIMIT; }
  // fused prefilter + register DL (ANX_PREFILTER=0 disables the filter: every length-compatible pair goes to the DL)
  SurvOut so{nullptr, b->sctr, 0};
  const bool have_long_q = b->qw > 1;
  if (nraw) {
    static const int enable_filter = []() { const char* e = getenv("ANX_PREFILTER"); return (e && e[0] == '0') ? 0 : 1; }();
    static const int enable_fast = []() { const char* e = getenv("ANX_SCORE_FAST"); return (e && e[0] == '0') ? 0 : 1; }();
    const int fastD = (enable_fast && d >= 1 && d <= 3) ? (int)d : 0;
    // survivor records: region r of the survivor list takes the survivors of region r of the pair list (<= maxfill)
    if ((size_t)maxfill > b->surv_region_cap) {
      if (b->surv) pool_free(b->surv);
      b->surv = nullptr;
      b->surv_region_cap = 0;
      const size_t need = (size_t)maxfill + (maxfill >> 3) + 256;
      if ((rc = dalloc(&b->surv, need * SCAN_REGIONS, err))) return rc;
      b->surv_region_cap = need;
    }
    so.list = b->surv;
    so.region_cap = (uint32_t)b->surv_region_cap;
    // slot lists for the pairs the fused kernel cannot score inline: strings of 17..32 symbols (8-word kernel, when the
    // batch has such queries) and everything else (longer strings, d > 3, or the few long candidates of a short-query
    // batch: for those the general kernel is cheaper than the 8-word one, measured 0.10 vs 0.22 ms on config 2)
    const bool need_lists = !fastD || have_long_q || dl->max_len > 16;
    if (need_lists && (size_t)maxfill > b->list_cap) {
      for (void* p : {(void*)b->list8, (void*)b->listg})
        if (p) pool_free(p);
      b->list8 = b->listg = nullptr;
      b->list_cap = 0;
      const size_t need = (size_t)maxfill + (maxfill >> 3) + 256;
      if ((rc = dalloc(&b->list8, need * SCAN_REGIONS, err)) || (rc = dalloc(&b->listg, need * SCAN_REGIONS, err))) return rc;
      b->list_cap = need;
    }
    HIP_TRY(hipMemsetAsync(b->sctr, 0, SCAN_REGIONS * RC_STRIDE * sizeof(uint32_t), st));
    HIP_TRY(hipMemsetAsync(b->lctr, 0, 2 * SCAN_REGIONS * RC_STRIDE * sizeof(uint32_t), st));
    const SlotList l8{b->list8, b->lctr, (uint32_t)b->list_cap}, lg{b->listg, b->lctr + SCAN_REGIONS * RC_STRIDE, (uint32_t)b->list_cap};
    const PairArgs pa{b->raw, b->q_meta, b->q_rows, dl->ent_meta, dl->ent_rowoff, dl->rows, dl->ent_freq, dl->ent_var_off,
                      b->p_score, b->p_meta, b->qmaxfreq, b->qsurv, b->qexpand};
    FilterArgs fa;
    fa.region_shift = b->region_shift; fa.rctr = b->rctr; fa.qexact = b->qexact; fa.stop = stop; fa.enable = enable_filter;
    fa.use_nw8 = have_long_q ? 1 : 0; fa.counters = b->counters; fa.stat_ctr = b->sctr;
    const dim3 fgrid(((maxfill + FS_BLK - 1) / FS_BLK) * SCAN_REGIONS);
    if (fastD == 1) hipLaunchKernelGGL(k_filter_score<1>, fgrid, dim3(256), 0, st, fa, pa, sa, so, l8, lg);
    else if (fastD == 2) hipLaunchKernelGGL(k_filter_score<2>, fgrid, dim3(256), 0, st, fa, pa, sa, so, l8, lg);
    else if (fastD == 3) hipLaunchKernelGGL(k_filter_score<3>, fgrid, dim3(256), 0, st, fa, pa, sa, so, l8, lg);
    else hipLaunchKernelGGL(k_filter_score<0>, fgrid, dim3(256), 0, st, fa, pa, sa, so, l8, lg);
    if (need_lists) {  // the list fills are only known on the device: grids cover the fullest pair-list region
      const dim3 lgrid(((maxfill + 255) / 256) * SCAN_REGIONS);
      if (fastD && have_long_q) {
        if (fastD == 1) hipLaunchKernelGGL(k_score_fast8<1>, lgrid, dim3(256), 0, st, l8, pa, sa, so);
        else if (fastD == 2) hipLaunchKernelGGL(k_score_fast8<2>, lgrid, dim3(256), 0, st, l8, pa, sa, so);
        else hipLaunchKernelGGL(k_score_fast8<3>, lgrid, dim3(256), 0, st, l8, pa, sa, so);
      }
      hipLaunchKernelGGL(k_score_pairs, dim3(((maxfill + threads - 1) / threads) * SCAN_REGIONS), dim3(threads), threads * sa.stride, st,
                         lg, pa, sa, so);
    }
  }
  HIP_TRY(hipEventRecord(b->ev[2], st));
  // ---- compact survivors -----------------------------------------------------------------------------
  exclusive_scan(b->qsurv, nq, b->soff, b->scan_tmp, st);
  uint32_t total_surv = 0;
  HIP_TRY(hipMemcpyAsync(&total_surv, b->soff + nq, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(h_counters, b->counters, sizeof h_counters, hipMemcpyDeviceToHost, st));
  if (nraw) HIP_TRY(hipMemcpyAsync(h_rctr.data(), b->sctr, h_rctr.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  b->n_pairs = n_valid - h_counters[CTR_SKIPPED];
  b->n_surv = total_surv;
  if ((rc = ensure_surv(b, (size_t)total_surv + (total_surv >> 2) + 1024, err))) return rc;
  uint32_t surv_fill = 0;
  uint64_t nsel = 0;
  if (nraw)
    for (uint32_t r = 0; r < SCAN_REGIONS; ++r) {
      surv_fill = std::max(surv_fill, h_rctr[r * RC_STRIDE]);
      nsel += h_rctr[r * RC_STRIDE + 1];
    }
  b->n_sel = nsel;
  if (surv_fill > b->surv_region_cap) { err = "survivor region overflow"; return ANX_ENODEVICE; }  // cannot happen: see the sizing above
  if (surv_fill) {
    CompactArgs ca{m.have_freq ? 1 : 0, dl->any_variants};
    hipLaunchKernelGGL(k_compact, dim3(((surv_fill + 255) / 256) * SCAN_REGIONS), dim3(256), 0, st, b->surv, b->sctr,
                       (uint32_t)b->surv_region_cap, ca, b->soff, b->qcur, dl->ent_rec, dl->ent_var_off, dl->var_target,
                       dl->var_target_freq, dl->var_score, b->c_rows);
  }
  HIP_TRY(hipEventRecord(b->ev[3], st));
  // ---- rank ------------------------------------------------------------------------------------------
  RankArgs ra;
  ra.cutoff_threshold = b->params.cutoff_threshold;
  ra.max_matches = b->params.max_matches;
  ra.freq_weight = b->params.freq_weight;
  ra.have_freq = m.have_freq ? 1 : 0;
  ra.any_variants = dl->any_variants;
  hipLaunchKernelGGL(k_rank, dim3((nq + 4 * RANK_QPW - 1) / (4 * RANK_QPW)), dim3(256), 0, st, nq, b->soff, b->c_rows, b->qmaxfreq,
                     b->qexpand, ra, b->t_key, b->r_rows, b->r_count);
  exclusive_scan(b->r_count, nq, b->r_off, b->scan_tmp, st);
  HIP_TRY(hipEventRecord(b->ev[4], st));
  uint32_t total_results = 0;
  HIP_TRY(hipMemcpyAsync(&total_results, b->r_off + nq, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  b->n_results = total_results;
  b->ran = true;
  anx_batch_stats& s = b->stats;
  s.n_queries = nq;
  s.n_pairs = b->n_pairs;
  s.n_class_tests = b->n_class_tests;
  s.n_results = total_results;
  s.n_scan_blocks = b->tiles.size();
  for (int i = 0; i <= NBITPLANES; ++i) s.n_tests_kind[i] = b->n_tests_kind[i];
  s.n_pair_slots = n_slots;
  s.n_survivors = total_surv;
  s.n_selected = b->n_sel;
  (void)hipEventElapsedTime(&s.ms_scan, b->ev[0], b->ev[1]);
  (void)hipEventElapsedTime(&s.ms_score, b->ev[1], b->ev[2]);
  (void)hipEventElapsedTime(&s.ms_group, b->ev[2], b->ev[3]);
  (void)hipEventElapsedTime(&s.ms_rank, b->ev[3], b->ev[4]);
  (void)hipEventElapsedTime(&s.ms_total, b->ev[0], b->ev[4]);
  (void)hipEventElapsedTime(&s.ms_scan_kernel, b->ev_scan0, b->ev[5]);
  return ANX_OK;
}

int batch_fetch(const HostModel& m, const DeviceLexicon* dl, const Batch* b, anx_result** rows, size_t** offs,
                std::string& err) {
  (void)m;
  (void)dl;
  if (!b->ran) { err = "batch has not been run"; return ANX_EINVAL; }
  HIP_TRY(hipSetDevice(b->device));
  const size_t n = b->n_input;
  size_t* off = static_cast<size_t*>(calloc(n + 1, sizeof(size_t)));
  anx_result* out = static_cast<anx_result*>(malloc(std::max<size_t>(1, b->n_results) * sizeof(anx_result)));
  if (!off || !out) { free(off); free(out); err = "out of memory"; return ANX_EINVAL; }
  if (b->nq && b->n_results) {
    DevRow* d_rows = nullptr;
    HIP_TRY(pool_malloc(reinterpret_cast<void**>(&d_rows), b->n_results * sizeof(DevRow)));
    hipLaunchKernelGGL(k_pack_rows, dim3(((uint32_t)b->nq + 255) / 256), dim3(256), 0, 0, (uint32_t)b->nq, b->soff,
                       b->r_off, b->r_count, b->r_rows, d_rows);
    std::vector<DevRow> h(b->n_results);
    std::vector<uint32_t> h_cnt(b->nq);
    HIP_TRY(hipMemcpy(h.data(), d_rows, b->n_results * sizeof(DevRow), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h_cnt.data(), b->r_count, b->nq * sizeof(uint32_t), hipMemcpyDeviceToHost));
    pool_free(d_rows);
    for (size_t s = 0; s < b->nq; ++s) off[b->order[s] + 1] = h_cnt[s];
    for (size_t i = 0; i < n; ++i) off[i + 1] += off[i];
    size_t src = 0;
    for (size_t s = 0; s < b->nq; ++s) {
      size_t dst = off[b->order[s]];
      for (uint32_t i = 0; i < h_cnt[s]; ++i, ++src, ++dst) {
        out[dst].vocab_id = h[src].vocab_id;
        out[dst].dist_score = h[src].dist_score;
        out[dst].freq_score = h[src].freq_score;
        out[dst].via = h[src].via == 0xFFFFFFFFu ? ANX_NO_VIA : (uint64_t)h[src].via;
      }
    }
  }
  *rows = out;
  *offs = off;
  return ANX_OK;
}

int batch_fetch_pairs(const HostModel& m, const DeviceLexicon* dl, const Batch* b, anx_pair** out, size_t* n,
                      std::string& err) {
  (void)m;
  if (!b->ran) { err = "batch has not been run"; return ANX_EINVAL; }
  HIP_TRY(hipSetDevice(b->device));
  const size_t R = b->n_raw;
  anx_pair* res = static_cast<anx_pair*>(malloc(std::max<size_t>(1, (size_t)b->n_pairs) * sizeof(anx_pair)));
  if (!res) { err = "out of memory"; return ANX_EINVAL; }
  size_t w = 0;
  if (R) {
    std::vector<uint2> pr(R);
    std::vector<uint32_t> pm(R), ev(dl->nentries);
    std::vector<double> ps(R);
    HIP_TRY(hipMemcpy(pr.data(), b->raw, R * sizeof(uint2), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pm.data(), b->p_meta, R * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ps.data(), b->p_score, R * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ev.data(), dl->ent_vocab, (size_t)dl->nentries * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < R; ++i) {
      if ((i & (((size_t)1 << b->region_shift) - 1)) >= b->region_fill[i >> b->region_shift]) continue;  // beyond the region's fill
      if (pm[i] == META_SKIPPED || w >= b->n_pairs) continue;
      anx_pair& r = res[w++];
      r.query = b->order[pr[i].x];
      r.vocab_id = ev[pr[i].y & 0x7FFFFFFFu];
      const uint32_t ld = pm[i] & 0x7F;
      r.ld = ld == PAIR_NONE ? (int16_t)-1 : (int16_t)ld;
      r.samecase = (pm[i] >> 7) & 1;
      r.lcs = (pm[i] >> 8) & 0xFF;
      r.prefixlen = (pm[i] >> 16) & 0xFF;
      r.suffixlen = (pm[i] >> 24) & 0xFF;
      r._pad = 0;
      r.score = ld == PAIR_NONE ? 0.0 : ps[i];
    }
  }
  *out = res;
  *n = w;
  return ANX_OK;
}

int batch_export_topk(const DeviceLexicon* dl, const Batch* b, void* dst, uint32_t stride, void* stream,
                      std::string& err) {
  (void)dl;
  if (!b->ran) { err = "batch has not been run"; return ANX_EINVAL; }
  if (!dst || stride == 0) { err = "bad export arguments"; return ANX_EINVAL; }
  HIP_TRY(hipSetDevice(b->device));
  const uint64_t total = (uint64_t)b->nq * stride;
  if (total)
    hipLaunchKernelGGL(k_export_topk, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), (uint32_t)b->nq, stride, b->soff, b->r_count,
                       b->r_rows, b->q_orig, static_cast<anx_topk_record*>(dst));
  HIP_TRY(hipGetLastError());
  return ANX_OK;
}

void batch_stats(const Batch* b, anx_batch_stats* s) { *s = b->stats; }

void batch_free(Batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  for (void* p : {(void*)b->q_cv, (void*)b->q_bits, (void*)b->q_rows, (void*)b->q_meta, (void*)b->q_orig, (void*)b->d_tiles, (void*)b->rctr, (void*)b->sctr, (void*)b->surv, (void*)b->quot,
                  (void*)b->counters, (void*)b->qexact, (void*)b->qsurv, (void*)b->soff, (void*)b->qcur,
                  (void*)b->qmaxfreq, (void*)b->scan_tmp, (void*)b->raw, (void*)b->p_score, (void*)b->p_meta, (void*)b->list8, (void*)b->listg, (void*)b->lctr,
                  (void*)b->c_rows, (void*)b->qexpand, (void*)b->r_rows, (void*)b->t_key, (void*)b->r_count, (void*)b->r_off})
    if (p) pool_free(p);
  for (auto& e : b->ev)
    if (e) (void)hipEventDestroy(e);
  if (b->ev_scan0) (void)hipEventDestroy(b->ev_scan0);
  delete b;
}

}  // namespace anx
