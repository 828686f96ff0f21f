// engine.hip -- the HIP (gfx950 / CDNA4) variant-query pipeline of the anx engine.
//
// Replaces, for a whole batch of queries at once, the reference's
//   find_nearest_anahashes  (/root/reference/src/lib.rs:1143-1308)  -> k_anagram_scan
//   gather_instances        (src/lib.rs:1311-1402, src/distance.rs)  -> k_group_pairs + k_score_pairs
//   score_and_rank          (src/lib.rs:1405-1653, src/types.rs:334-365) -> k_score_pairs + k_rank
// Integer work only: no MFMA.  Wave = 64 lanes everywhere.  See DESIGN.md for layout and rooflines.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
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
struct Tile {           // TQ queries of one length against the +-k charcount window of classes
  uint32_t q0, nq;      // query range (queries are sorted by length)
  uint32_t c0, c1;      // class-rank range [c0, c1)
  uint32_t k;           // clamped anagram distance for this length
  uint32_t lq;          // query length in symbols
};

struct DeviceLexicon {
  int device = 0;
  int nplanes = 0;
  uint32_t nclasses = 0, nentries = 0, cstride = 0, max_len = 0;
  uint32_t* cls_planes = nullptr;
  uint8_t* cls_len = nullptr;
  uint32_t* cls_off = nullptr;
  uint32_t* ent_vocab = nullptr;
  uint32_t* ent_freq = nullptr;
  uint32_t* ent_meta = nullptr;
  uint32_t* ent_rowoff = nullptr;
  uint32_t* ent_order = nullptr;
  uint4* rows = nullptr;
  size_t bytes = 0;
};

enum { CTR_RAW = 0, CTR_TESTS_LO = 1, CTR_TESTS_HI = 2, CTR_RESULTS = 3, CTR_N = 8 };

struct Batch {
  int device = 0;
  size_t nq = 0;            // encoded queries
  anx_params params;
  // host side
  std::vector<uint32_t> order;     // sorted position -> original index
  std::vector<int32_t> status;     // per original query: 0 ok, ANX_EEMPTY, ANX_ELIMIT
  size_t n_input = 0;
  std::vector<Tile> tiles;
  uint32_t qw = 1;                 // uint4 words per query row
  uint32_t dmax = 0;
  uint64_t n_class_tests = 0;
  // device: queries
  uint32_t* q_cv = nullptr;        // [nq][nplanes]
  uint4* q_rows = nullptr;         // [nq][qw]
  uint32_t* q_meta = nullptr;      // len | k<<8 | d<<16 | first_is_lower<<24
  uint32_t* q_orig = nullptr;      // original index
  Tile* d_tiles = nullptr;
  // device: pipeline
  uint32_t* counters = nullptr;
  uint32_t* qcount = nullptr;
  uint32_t* qexact = nullptr;
  uint32_t* qoff = nullptr;        // nq+1
  uint32_t* qcur = nullptr;
  uint32_t* qmaxfreq = nullptr;
  uint32_t* scan_tmp = nullptr;
  uint2* raw = nullptr;
  size_t raw_cap = 0;
  uint32_t* pair_q = nullptr;
  uint32_t* pair_e = nullptr;
  double* p_score = nullptr;
  uint32_t* p_meta = nullptr;
  uint32_t* r_entry = nullptr;
  double* r_dist = nullptr;
  double* r_freq = nullptr;
  double* t_key = nullptr;
  uint32_t* t_pos = nullptr;
  uint32_t* r_count = nullptr;
  uint32_t* r_off = nullptr;       // nq+1
  size_t pair_cap = 0;
  uint64_t n_pairs = 0, n_results = 0;
  bool ran = false;
  hipEvent_t ev[6] = {};
  anx_batch_stats stats = {};
};

// ------------------------------------------------------------------------------------------------
// K1: anagram window scan.
//   Spec: the set returned by find_nearest_anahashes (src/lib.rs:1143-1308) equals
//     { class c : L1(cv_q, cv_c) <= k, |len_c - len_q| <= k, cv_q and cv_c share a symbol }
//   (the bigint `cand % av == 0` containment test of src/anahash.rs:165-171 is
//   multiset inclusion, i.e. a statement about the prime-exponent = count vectors).
//   Each lane keeps CPL classes' count vectors in VGPRs (coalesced plane loads); the tile's queries are
//   streamed through SGPRs (wave-uniform scalar loads), 4 symbols per v_sad_u8.
// ------------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(4))) uint32_t* cptr_u32;  // constant address space: s_load

constexpr uint32_t SCAN_TQ = 256;       // queries per tile (= per workgroup)
constexpr uint32_t SCAN_CHUNK = 1024;   // pair slots a wave reserves per global atomic
constexpr uint32_t RAW_INVALID = 0xFFFFFFFFu;

// One workgroup owns one query tile and streams every class chunk of the tile's charcount window past it.
// Per-query pair counts live in LDS (no global atomics); each wave appends its hits to wave-private
// 1024-slot chunks of the flat pair list, reserved with ONE global atomic per chunk (a single contended
// counter word sustains only ~88 M atomics/s on this chip, MI355X_MICROARCH.md "dequeue").
template <int NP, int CPL>
__global__ __launch_bounds__(256) void k_anagram_scan(const Tile* __restrict__ tiles,
                                                      const uint32_t* __restrict__ q_cv,
                                                      const uint32_t* __restrict__ planes, uint32_t cstride,
                                                      const uint8_t* __restrict__ cls_len,
                                                      const uint32_t* __restrict__ cls_off, uint2* __restrict__ raw,
                                                      uint32_t raw_cap, uint32_t* __restrict__ counters,
                                                      uint32_t* __restrict__ qcount, uint32_t* __restrict__ qexact) {
  __shared__ uint32_t s_cnt[SCAN_TQ], s_exact[SCAN_TQ];
  const Tile t = tiles[blockIdx.x];
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t i = threadIdx.x; i < SCAN_TQ; i += 256) { s_cnt[i] = 0; s_exact[i] = 0; }
  __syncthreads();
  uint32_t w_base = 0, w_left = 0;  // this wave's current chunk of the pair list (wave-uniform)
  const uint32_t k = t.k;
  cptr_u32 qbase = (cptr_u32)(q_cv + (size_t)t.q0 * NP);
  for (uint32_t cb = t.c0; cb < t.c1; cb += 256 * CPL) {
    // Unguarded loads: the plane arrays are padded by a full chunk of 0xFF classes, and real classes
    // beyond c1 lie outside the +-k charcount window, so they can never satisfy L1 <= k.
    uint32_t cv[CPL][NP];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const uint32_t c = cb + j * 256 + threadIdx.x;
#pragma unroll
      for (int p = 0; p < NP; ++p) cv[j][p] = planes[(size_t)p * cstride + c];
    }
    uint32_t qnext[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) qnext[p] = qbase[p];
    for (uint32_t qi = 0; qi < t.nq; ++qi) {
      uint32_t qreg[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) qreg[p] = qnext[p];
      // prefetch the next query's count vector into SGPRs while this one is compared
      cptr_u32 qv = qbase + (size_t)(qi + 1 < t.nq ? qi + 1 : qi) * NP;
#pragma unroll
      for (int p = 0; p < NP; ++p) qnext[p] = qv[p];
      uint32_t dist[CPL];
      bool any = false;
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        uint32_t acc = 0;
#pragma unroll
        for (int p = 0; p < NP; ++p) acc = __builtin_amdgcn_sad_u8(qreg[p], cv[j][p], acc);
        dist[j] = acc;
        any |= acc <= k;
      }
      if (__ballot(any) != 0ull) {  // wave-uniform; ~0.1 % of class tests hit
        uint32_t e0[CPL], n[CPL], ntot = 0, nex = 0;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          n[j] = 0;
          e0[j] = 0;
          if (dist[j] <= k) {
            const uint32_t c = cb + j * 256 + threadIdx.x;
            // shares at least one symbol <=> L1 < len_q + len_c (deleting all of q is never enumerated:
            // RecurseDeletionIterator with empty_leaves=false, src/iterators.rs:177, src/lib.rs:1205)
            if (dist[j] < t.lq + (uint32_t)cls_len[c]) {
              e0[j] = cls_off[c];
              n[j] = cls_off[c + 1] - e0[j];
              ntot += n[j];
              if (dist[j] == 0) nex += n[j];
            }
          }
        }
        // wave-wide exclusive prefix sum of the per-lane pair counts
        uint32_t incl = ntot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t u = __shfl_up(incl, o);
          if (lane >= (uint32_t)o) incl += u;
        }
        const uint32_t total = __shfl(incl, 63);
        if (total) {
          if (total > w_left) {  // wave-uniform: close the current chunk, reserve a new one
            for (uint32_t i = lane; i < w_left; i += 64)
              if (w_base + i < raw_cap) raw[w_base + i] = make_uint2(RAW_INVALID, 0u);
            const uint32_t need = total > SCAN_CHUNK ? total : SCAN_CHUNK;
            uint32_t b = 0;
            if (lane == 0) b = atomicAdd(&counters[CTR_RAW], need);
            w_base = __shfl(b, 0);
            w_left = need;
          }
          if (ntot) {
            const uint32_t q = t.q0 + qi;
            atomicAdd(&s_cnt[qi], ntot);
            if (nex) atomicAdd(&s_exact[qi], nex);
            uint32_t pos = w_base + incl - ntot;
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
              const uint32_t exact = dist[j] == 0 ? 0x80000000u : 0u;
              for (uint32_t i = 0; i < n[j]; ++i, ++pos)
                if (pos < raw_cap) raw[pos] = make_uint2(q, (e0[j] + i) | exact);
            }
          }
          w_base += total;
          w_left -= total;
        }
      }
    }
  }
  for (uint32_t i = lane; i < w_left; i += 64)
    if (w_base + i < raw_cap) raw[w_base + i] = make_uint2(RAW_INVALID, 0u);
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < t.nq; i += 256) {
    qcount[t.q0 + i] = s_cnt[i];
    qexact[t.q0 + i] = s_exact[i];
  }
}

// StopCriterion::StopAtExactMatch (src/lib.rs:1164-1173): only the exact class survives.
__global__ void k_stop_fixup(uint32_t n, uint32_t* __restrict__ qcount, const uint32_t* __restrict__ qexact) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && qexact[i] > 0) qcount[i] = qexact[i];
}

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
// K2: group the flat pair list by query (counting-sort scatter).  Order inside a query is arbitrary;
// ranking uses a total order whose last key is ent_order (= reference enumeration order).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_group_pairs(const uint2* __restrict__ raw, uint32_t nraw,
                                                     const uint32_t* __restrict__ qoff,
                                                     const uint32_t* __restrict__ qexact, int stop,
                                                     uint32_t* __restrict__ qcur, uint32_t* __restrict__ pair_q,
                                                     uint32_t* __restrict__ pair_e) {
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  if (r >= nraw) return;
  const uint2 v = raw[r];
  const uint32_t q = v.x;
  if (q == RAW_INVALID) return;  // unused tail of a wave's chunk
  if (stop && qexact[q] > 0 && !(v.y & 0x80000000u)) return;
  const uint32_t pos = qoff[q] + atomicAdd(&qcur[q], 1u);
  pair_q[pos] = q;
  pair_e[pos] = v.y & 0x7FFFFFFFu;
}

// ------------------------------------------------------------------------------------------------
// K3: score one (query, candidate) pair per lane.
//   damerau_levenshtein (src/distance.rs:101-179) in its band-limited saturating form:
//   cells with |i-j| > d are d+1, every value saturates at d+1, the transposition term only looks back
//   d rows / d columns (farther ones cost > d).  Identical to the reference for every outcome <= d.
//   Per-lane state lives in LDS: query row, candidate row, a ring of d+2 band rows.
//   longest_common_substring_length / common_prefix_length / common_suffix_length: src/distance.rs:181-231.
//   Score: src/lib.rs:1433-1452 (f64, same association, no FMA contraction).
// ------------------------------------------------------------------------------------------------
struct ScoreArgs {
  double w_ld, w_lcs, w_prefix, w_suffix, w_case, w_sum;
  int have_freq;
  uint32_t lqp, lcp;   // bytes reserved per lane for the query / candidate row (multiples of 16)
  uint32_t stride;     // bytes per lane (odd number of dwords: conflict-free ds access)
  uint32_t qw;
};
#define PAIR_NONE 0x7Fu

__global__ void k_score_pairs(uint32_t P, const uint32_t* __restrict__ pair_q, const uint32_t* __restrict__ pair_e,
                              const uint32_t* __restrict__ q_meta, const uint4* __restrict__ q_rows,
                              const uint32_t* __restrict__ ent_meta, const uint32_t* __restrict__ ent_rowoff,
                              const uint4* __restrict__ rows, const uint32_t* __restrict__ ent_freq, ScoreArgs a,
                              double* __restrict__ p_score, uint32_t* __restrict__ p_meta,
                              uint32_t* __restrict__ qmaxfreq) {
  extern __shared__ uint32_t lds32[];
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  uint8_t* S = reinterpret_cast<uint8_t*>(lds32) + (size_t)threadIdx.x * a.stride;
  uint8_t* T = S + a.lqp;
  uint8_t* R = T + a.lcp;
  const uint32_t q = pair_q[p], e = pair_e[p];
  const uint32_t qm = q_meta[q], em = ent_meta[e];
  const int lq = qm & 0xFF, d = (qm >> 16) & 0xFF, lc = em & 0xFF;
  const int diff = lq > lc ? lq - lc : lc - lq;
  uint32_t ld = PAIR_NONE, lcs = 0, pre = 0, suf = 0;
  uint32_t samecase = 1;
  double score = __builtin_nan("");
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
    // ---- banded unrestricted Damerau-Levenshtein --------------------------------------------------
    const int cap = d + 1, W = 2 * d + 3, NR = d + 2;
    // row i is stored at R[(i % NR) * W + col], col = j - i + d + 1 in [1, 2d+1]; cols 0 and 2d+2 are guards
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
    if (res <= (uint32_t)d) {  // src/distance.rs:173-178
      ld = res;
      if (a.w_lcs > 0.0) {  // src/lib.rs:1352-1356; diagonal walk == the reference's naive scan
        uint32_t best = 0;
        for (int delta = -(lq - 1); delta <= lc - 1; ++delta) {
          const int i0 = delta < 0 ? -delta : 0;
          const int i1 = min(lq, lc - delta);
          if ((uint32_t)(i1 - i0) <= best) continue;
          uint32_t run = 0;
          for (int i = i0; i < i1; ++i) {
            run = S[i] == T[i + delta] ? run + 1 : 0;
            best = max(best, run);
          }
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
      const double L = (double)lq;
      const double distance_score = (int)ld > lq ? 0.0 : 1.0 - ((double)ld / L);
      const double lcs_score = (double)lcs / L;
      const double prefix_score = (double)pre / L;
      const double suffix_score = (double)suf / L;
      score = (a.w_ld * distance_score + a.w_lcs * lcs_score + a.w_prefix * prefix_score +
               a.w_suffix * suffix_score + (samecase ? a.w_case : 0.0)) /
              a.w_sum;
      // max_freq over every DL-surviving instance, before the threshold test (src/lib.rs:1454-1462)
      if (a.have_freq) atomicMax(&qmaxfreq[q], ent_freq[e]);
      else atomicMax(&qmaxfreq[q], 1u);
    }
  }
  p_score[p] = score;
  p_meta[p] = ld | (samecase << 7) | (lcs << 8) | (pre << 16) | (suf << 24);
}

// ------------------------------------------------------------------------------------------------
// K4: rank.  One wave per query.  score threshold (src/lib.rs:1475), freq normalisation (:1521-1525),
// stable sort by rank_cmp (src/types.rs:344-365) realised as a total order with the entry index as last
// key, crop with the tie rule (:1536-1589), cutoff (:1598-1622).
// ------------------------------------------------------------------------------------------------
struct RankArgs {
  double score_threshold, cutoff_threshold;
  uint64_t max_matches;
  float freq_weight;
  int have_freq;
};
constexpr int RANK_LCAP = 256;

__device__ inline double result_score(double dist, double freq, float fw) {  // src/types.rs:335-341
  if (fw == 0.0f) return dist;
  return (dist + ((double)fw * freq)) / (1.0 + (double)fw);
}

__global__ __launch_bounds__(256) void k_rank(uint32_t nq, const uint32_t* __restrict__ qoff,
                                              const uint32_t* __restrict__ pair_e,
                                              const double* __restrict__ p_score,
                                              const uint32_t* __restrict__ ent_freq,
                                              const uint32_t* __restrict__ ent_order,
                                              const uint32_t* __restrict__ qmaxfreq, RankArgs a,
                                              double* __restrict__ t_key, uint32_t* __restrict__ t_pos,
                                              uint32_t* __restrict__ r_entry, double* __restrict__ r_dist,
                                              double* __restrict__ r_freq, uint32_t* __restrict__ r_count) {
  __shared__ double s_key[4][RANK_LCAP];
  __shared__ uint32_t s_freq[4][RANK_LCAP], s_entry[4][RANK_LCAP], s_pos[4][RANK_LCAP], s_ord[4][RANK_LCAP];
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t q = blockIdx.x * 4 + wid;
  if (q >= nq) return;
  const uint32_t seg0 = qoff[q], seg1 = qoff[q + 1];
  const uint32_t maxf = qmaxfreq[q];
  const double max_freq = a.have_freq ? (double)maxf : (maxf ? 1.0 : 0.0);
  const bool sort_weighted = a.freq_weight > 0.0f;    // rank_cmp's branch
  const bool score_weighted = a.freq_weight != 0.0f;  // score()'s branch
  // ---- compact the survivors (score >= threshold) ---------------------------------------------------
  uint32_t n = 0;
  for (uint32_t base = seg0; base < seg1; base += 64) {
    const uint32_t pos = base + lane;
    double sc = __builtin_nan("");
    if (pos < seg1) sc = p_score[pos];
    const bool keep = sc >= a.score_threshold;  // NaN (pruned by DL) compares false
    const unsigned long long mask = __ballot(keep);
    if (keep) {
      const uint32_t idx = n + __popcll(mask & ((1ull << lane) - 1ull));
      const uint32_t e = pair_e[pos];
      const uint32_t f = a.have_freq ? ent_freq[e] : 1u;
      double key = sc;
      if (sort_weighted) {
        const double fs = max_freq > 0.0 ? (double)f / max_freq : (double)f;
        key = result_score(sc, fs, a.freq_weight);
      }
      if (idx < RANK_LCAP) {
        s_key[wid][idx] = key; s_freq[wid][idx] = f; s_entry[wid][idx] = e; s_pos[wid][idx] = pos;
        s_ord[wid][idx] = ent_order[e];
      } else {
        t_key[seg0 + idx] = key; t_pos[seg0 + idx] = pos;
      }
    }
    n += __popcll(mask);
  }
  if (n > RANK_LCAP) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  // ---- rank by counting -------------------------------------------------------------------------------
  const bool full = score_weighted || a.max_matches == 0;
  const uint32_t M = full ? n : (uint32_t)min((uint64_t)n, a.max_matches + 1);
  for (uint32_t i = lane; i < n; i += 64) {
    double ki; uint32_t fi, ei, pi, oi;
    if (i < RANK_LCAP) { ki = s_key[wid][i]; fi = s_freq[wid][i]; ei = s_entry[wid][i]; pi = s_pos[wid][i]; oi = s_ord[wid][i]; }
    else { ki = t_key[seg0 + i]; pi = t_pos[seg0 + i]; ei = pair_e[pi]; fi = a.have_freq ? ent_freq[ei] : 1u; oi = ent_order[ei]; }
    uint32_t rank = 0;
    for (uint32_t j = 0; j < n; ++j) {
      double kj; uint32_t fj, oj;
      if (j < RANK_LCAP) { kj = s_key[wid][j]; fj = s_freq[wid][j]; oj = s_ord[wid][j]; }
      else { kj = t_key[seg0 + j]; const uint32_t ej = pair_e[t_pos[seg0 + j]]; fj = a.have_freq ? ent_freq[ej] : 1u; oj = ent_order[ej]; }
      bool before;
      if (sort_weighted) before = kj > ki || (kj == ki && oj < oi);
      else before = kj > ki || (kj == ki && (fj > fi || (fj == fi && oj < oi)));
      rank += before;
    }
    if (rank < M) {
      r_entry[seg0 + rank] = ei;
      r_dist[seg0 + rank] = p_score[pi];
      r_freq[seg0 + rank] = max_freq > 0.0 ? (double)fi / max_freq : (double)fi;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  // ---- crop + cutoff, literally, by one lane ---------------------------------------------------------
  if (lane == 0) {
    const float fw = a.freq_weight;
    const double* rd = r_dist + seg0;
    const double* rf = r_freq + seg0;
    uint32_t len = n;
    const uint64_t mm = a.max_matches;
    if (mm > 0 && (uint64_t)len > mm) {
      const double last = result_score(rd[mm - 1], rf[mm - 1], fw);
      const double cropped = result_score(rd[mm], rf[mm], fw);
      if (cropped < last) len = (uint32_t)mm;
      else {
        uint32_t early = 0, late = 0;
        for (uint32_t i = 0; i < M; ++i) {
          if (rd[i] == cropped && early == 0) early = i;
          if (rd[i] < cropped) { late = i; break; }
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
        const double s = result_score(rd[i], rf[i], fw);
        if (have) {
          if (s <= best / a.cutoff_threshold) { cutoff = i; break; }
        } else { best = s; have = true; }
      }
    }
    if (cutoff > 0) len = cutoff;
    r_count[q] = len;
  }
}

// dense result rows (device) for download / gather
struct DevRow {
  uint32_t vocab_id, query;
  double dist_score, freq_score;
};
__global__ __launch_bounds__(256) void k_pack_rows(uint32_t nq, const uint32_t* __restrict__ qoff,
                                                   const uint32_t* __restrict__ r_off,
                                                   const uint32_t* __restrict__ r_count,
                                                   const uint32_t* __restrict__ r_entry,
                                                   const double* __restrict__ r_dist,
                                                   const double* __restrict__ r_freq,
                                                   const uint32_t* __restrict__ ent_vocab,
                                                   const uint32_t* __restrict__ q_orig, DevRow* __restrict__ out) {
  const uint32_t q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  const uint32_t n = r_count[q], src = qoff[q], dst = r_off[q];
  for (uint32_t i = 0; i < n; ++i) {
    DevRow r;
    r.vocab_id = ent_vocab[r_entry[src + i]];
    r.query = q_orig[q];
    r.dist_score = r_dist[src + i];
    r.freq_score = r_freq[src + i];
    out[dst + i] = r;
  }
}
__global__ __launch_bounds__(256) void k_export_topk(uint32_t nq, uint32_t stride, const uint32_t* __restrict__ qoff,
                                                     const uint32_t* __restrict__ r_count,
                                                     const uint32_t* __restrict__ r_entry,
                                                     const double* __restrict__ r_dist,
                                                     const double* __restrict__ r_freq,
                                                     const uint32_t* __restrict__ ent_vocab,
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
    const uint32_t src = qoff[q] + i;
    r.vocab_id = ent_vocab[r_entry[src]];
    r.freq_score = (float)r_freq[src];
    r.dist_score = r_dist[src];
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

template <typename T>
static int upload(T** dst, const void* src, size_t count, std::string& err, size_t* total) {
  const size_t bytes = std::max<size_t>(count * sizeof(T), 16);
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(dst), bytes));
  if (count) HIP_TRY(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
  if (total) *total += bytes;
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
  d->nplanes = img.nplanes;
  d->nclasses = img.nclasses;
  d->nentries = img.nentries;
  d->cstride = img.cstride;
  uint32_t maxlen = 1;
  for (uint32_t m : img.ent_meta) maxlen = std::max(maxlen, m & 0xFFu);
  d->max_len = maxlen;
  int rc = ANX_OK;
  std::vector<uint32_t> off = img.cls_off;
  if (off.empty()) off.push_back(0);
  if ((rc = upload(&d->cls_planes, img.cls_planes.data(), img.cls_planes.size(), err, &d->bytes)) ||
      (rc = upload(&d->cls_len, img.cls_len.data(), img.cls_len.size(), err, &d->bytes)) ||
      (rc = upload(&d->cls_off, off.data(), off.size(), err, &d->bytes)) ||
      (rc = upload(&d->ent_vocab, img.ent_vocab.data(), img.ent_vocab.size(), err, &d->bytes)) ||
      (rc = upload(&d->ent_freq, img.ent_freq.data(), img.ent_freq.size(), err, &d->bytes)) ||
      (rc = upload(&d->ent_meta, img.ent_meta.data(), img.ent_meta.size(), err, &d->bytes)) ||
      (rc = upload(&d->ent_rowoff, img.ent_rowoff.data(), img.ent_rowoff.size(), err, &d->bytes)) ||
      (rc = upload(&d->ent_order, img.ent_order.data(), img.ent_order.size(), err, &d->bytes)) ||
      (rc = upload(reinterpret_cast<uint8_t**>(&d->rows), img.rows.data(), img.rows.size(), err, &d->bytes))) {
    lexicon_free(d);
    return nullptr;
  }
  return d;
}

void lexicon_free(DeviceLexicon* d) {
  if (!d) return;
  (void)hipSetDevice(d->device);
  for (void* p : {(void*)d->cls_planes, (void*)d->cls_len, (void*)d->cls_off, (void*)d->ent_vocab,
                  (void*)d->ent_freq, (void*)d->ent_meta, (void*)d->ent_rowoff, (void*)d->ent_order, (void*)d->rows})
    if (p) (void)hipFree(p);
  delete d;
}


Batch* batch_encode(const HostModel& m, const DeviceLexicon* dl, const char* const* utf8, size_t n,
                    const anx_params& p, std::string& err, int* code) {
  *code = ANX_OK;
  if (!dl) { err = "model is not resident on a device (no HIP device / anx_model_to_device not called)"; *code = ANX_ENODEVICE; return nullptr; }
  if (hipSetDevice(dl->device) != hipSuccess) { err = "hipSetDevice failed"; *code = ANX_ENODEVICE; return nullptr; }
  Batch* b = new Batch();
  b->device = dl->device;
  b->params = p;
  b->n_input = n;
  b->status.assign(n, 0);
  const int NP = dl->nplanes;
  struct Enc {
    std::vector<uint8_t> norm, cv;
    uint32_t orig;
    uint32_t meta;
  };
  std::vector<Enc> enc;
  enc.reserve(n);
  size_t maxlen = 1;
  for (size_t i = 0; i < n; ++i) {
    Enc e;
    e.orig = (uint32_t)i;
    if (!utf8[i] || !m.encode(utf8[i], e.norm, e.cv)) { b->status[i] = ANX_ELIMIT; continue; }
    if (e.norm.empty()) { b->status[i] = ANX_EEMPTY; continue; }
    const int len = (int)e.norm.size();
    const int k = clamp_threshold(p.max_anagram_distance, len, kMaxAnagramDistance);
    const int d = clamp_threshold(p.max_edit_distance, len, kMaxEditDistance);
    e.meta = (uint32_t)len | ((uint32_t)k << 8) | ((uint32_t)d << 16) |
             (first_char_is_lowercase(utf8[i]) ? 1u << 24 : 0u);
    maxlen = std::max(maxlen, e.norm.size());
    b->dmax = std::max<uint32_t>(b->dmax, (uint32_t)d);
    enc.push_back(std::move(e));
  }
  // length-bucketed order (stable): queries of one length share k, d and the class window
  std::stable_sort(enc.begin(), enc.end(), [](const Enc& x, const Enc& y) { return (x.meta & 0xFF) < (y.meta & 0xFF); });
  const size_t nq = enc.size();
  b->nq = nq;
  b->qw = (uint32_t)((maxlen + 15) / 16);
  std::vector<uint32_t> h_cv(nq * (size_t)NP, 0), h_meta(nq), h_orig(nq);
  std::vector<uint8_t> h_rows(nq * (size_t)b->qw * 16, 0xFE);
  b->order.resize(nq);
  for (size_t i = 0; i < nq; ++i) {
    memcpy(&h_cv[i * (size_t)NP], enc[i].cv.data(), std::min(enc[i].cv.size(), (size_t)NP * 4));
    memcpy(&h_rows[i * (size_t)b->qw * 16], enc[i].norm.data(), enc[i].norm.size());
    h_meta[i] = enc[i].meta;
    h_orig[i] = enc[i].orig;
    b->order[i] = enc[i].orig;
  }
  // tiles + work list
  const uint32_t TQ = SCAN_TQ;
  for (size_t i = 0; i < nq;) {
    size_t j = i;
    while (j < nq && (h_meta[j] & 0xFFFF) == (h_meta[i] & 0xFFFF)) ++j;
    const uint32_t lq = h_meta[i] & 0xFF, k = (h_meta[i] >> 8) & 0xFF;
    const int lo = std::max<int>(1, (int)lq - (int)k), hi = std::min<int>(kMaxSymbols, (int)lq + (int)k);
    const uint32_t c0 = m.lex.bucket_begin[lo], c1 = m.lex.bucket_begin[hi + 1];
    for (size_t s = i; s < j; s += TQ) {
      Tile t{(uint32_t)s, (uint32_t)std::min<size_t>(TQ, j - s), c0, c1, k, lq};
      b->tiles.push_back(t);
      b->n_class_tests += (uint64_t)t.nq * (c1 - c0);
    }
    i = j;
  }
  auto up = [&](auto** dst, const void* src, size_t count) { return upload(dst, src, count, err, nullptr); };
  int rc;
  if ((rc = up(&b->q_cv, h_cv.data(), h_cv.size())) || (rc = up(reinterpret_cast<uint8_t**>(&b->q_rows), h_rows.data(), h_rows.size())) ||
      (rc = up(&b->q_meta, h_meta.data(), nq)) || (rc = up(&b->q_orig, h_orig.data(), nq)) ||
      (rc = up(&b->d_tiles, b->tiles.data(), b->tiles.size()))) {
    *code = rc;
    batch_free(b);
    return nullptr;
  }
  auto dalloc = [&](auto** dst, size_t count) -> int {
    using TT = std::remove_pointer_t<std::remove_pointer_t<decltype(dst)>>;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(dst), std::max<size_t>(count * sizeof(TT), 16));
    if (e != hipSuccess) { err = std::string("hipMalloc: ") + hipGetErrorString(e); return ANX_ENODEVICE; }
    return ANX_OK;
  };
  const size_t nblk = (nq + SCAN_TILE - 1) / SCAN_TILE + 2;
  if ((rc = dalloc(&b->counters, CTR_N)) || (rc = dalloc(&b->qcount, nq)) || (rc = dalloc(&b->qexact, nq)) ||
      (rc = dalloc(&b->qoff, nq + 1)) || (rc = dalloc(&b->qcur, nq)) || (rc = dalloc(&b->qmaxfreq, nq)) ||
      (rc = dalloc(&b->scan_tmp, nblk)) || (rc = dalloc(&b->r_count, nq)) || (rc = dalloc(&b->r_off, nq + 1))) {
    *code = rc;
    batch_free(b);
    return nullptr;
  }
  b->raw_cap = nq * 192 + (size_t)b->tiles.size() * 4 * SCAN_CHUNK + (1u << 16);
  if ((rc = dalloc(&b->raw, b->raw_cap))) { *code = rc; batch_free(b); return nullptr; }
  for (auto& e : b->ev)
    if (hipEventCreate(&e) != hipSuccess) { err = "hipEventCreate failed"; *code = ANX_ENODEVICE; batch_free(b); return nullptr; }
  return b;
}

template <int NP, int CPL>
static void launch_scan(const DeviceLexicon* dl, Batch* b, hipStream_t st) {
  hipLaunchKernelGGL((k_anagram_scan<NP, CPL>), dim3((uint32_t)b->tiles.size()), dim3(256), 0, st, b->d_tiles,
                     b->q_cv, dl->cls_planes, dl->cstride, dl->cls_len, dl->cls_off, b->raw,
                     (uint32_t)std::min<size_t>(b->raw_cap, 0xFFFFFFFFu), b->counters, b->qcount, b->qexact);
}

static int exclusive_scan(const uint32_t* in, uint32_t n, uint32_t* out, uint32_t* tmp, hipStream_t st) {
  const uint32_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nb == 0) return hipMemsetAsync(out, 0, sizeof(uint32_t), st) == hipSuccess ? 0 : -1;
  hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(SCAN_THREADS), 0, st, in, n, out, tmp);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_THREADS), 0, st, tmp, nb);
  hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(SCAN_THREADS), 0, st, out, n, tmp, nb);
  return 0;
}

static void free_pair_buffers(Batch* b) {
  for (void* p : {(void*)b->pair_q, (void*)b->pair_e, (void*)b->p_score, (void*)b->p_meta, (void*)b->r_entry,
                  (void*)b->r_dist, (void*)b->r_freq, (void*)b->t_key, (void*)b->t_pos})
    if (p) (void)hipFree(p);
  b->pair_q = b->pair_e = b->p_meta = b->r_entry = b->t_pos = nullptr;
  b->p_score = b->r_dist = b->r_freq = b->t_key = nullptr;
  b->pair_cap = 0;
}

int batch_run(const HostModel& m, const DeviceLexicon* dl, Batch* b, void* stream, std::string& err) {
  if (!dl) { err = "model is not resident on a device"; return ANX_ENODEVICE; }
  HIP_TRY(hipSetDevice(dl->device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const uint32_t nq = (uint32_t)b->nq;
  b->ran = false;
  b->n_pairs = b->n_results = 0;
  if (nq == 0) { b->ran = true; return ANX_OK; }
  const int stop = b->params.stop_at_exact_match ? 1 : 0;
  uint32_t h_counters[CTR_N];
  uint32_t total_pairs = 0;
  HIP_TRY(hipEventRecord(b->ev[0], st));
  for (int attempt = 0; attempt < 2; ++attempt) {
    HIP_TRY(hipMemsetAsync(b->counters, 0, CTR_N * sizeof(uint32_t), st));
    HIP_TRY(hipMemsetAsync(b->qcur, 0, nq * sizeof(uint32_t), st));
    HIP_TRY(hipMemsetAsync(b->qmaxfreq, 0, nq * sizeof(uint32_t), st));
    if (!b->tiles.empty()) {
      switch (dl->nplanes) {
        case 8: launch_scan<8, 4>(dl, b, st); break;
        case 16: launch_scan<16, 2>(dl, b, st); break;
        case 24: launch_scan<24, 1>(dl, b, st); break;
        case 32: launch_scan<32, 1>(dl, b, st); break;
        default: launch_scan<42, 1>(dl, b, st); break;
      }
    }
    if (stop) hipLaunchKernelGGL(k_stop_fixup, dim3((nq + 255) / 256), dim3(256), 0, st, nq, b->qcount, b->qexact);
    if (attempt == 0) HIP_TRY(hipEventRecord(b->ev[1], st));
    exclusive_scan(b->qcount, nq, b->qoff, b->scan_tmp, st);
    HIP_TRY(hipMemcpyAsync(h_counters, b->counters, sizeof h_counters, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&total_pairs, b->qoff + nq, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if ((size_t)h_counters[CTR_RAW] <= b->raw_cap) break;
    if (attempt == 1) { err = "pair list overflow after regrow"; return ANX_ENODEVICE; }
    (void)hipFree(b->raw);
    b->raw = nullptr;
    b->raw_cap = (size_t)h_counters[CTR_RAW] + (size_t)b->tiles.size() * 4 * SCAN_CHUNK + 1024;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->raw), b->raw_cap * sizeof(uint2)));
  }
  const uint32_t nraw = h_counters[CTR_RAW];
  const uint32_t P = total_pairs;
  b->n_pairs = P;
  if ((size_t)P > b->pair_cap) {
    free_pair_buffers(b);
    const size_t cap = (size_t)P + (P >> 3) + 1024;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->pair_q), cap * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->pair_e), cap * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->p_score), cap * 8));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->p_meta), cap * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->r_entry), cap * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->r_dist), cap * 8));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->r_freq), cap * 8));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->t_key), cap * 8));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->t_pos), cap * 4));
    b->pair_cap = cap;
  }
  if (nraw)
    hipLaunchKernelGGL(k_group_pairs, dim3((nraw + 255) / 256), dim3(256), 0, st, b->raw, nraw, b->qoff, b->qexact,
                       stop, b->qcur, b->pair_q, b->pair_e);
  HIP_TRY(hipEventRecord(b->ev[2], st));
  // ---- score ---------------------------------------------------------------------------------------------
  ScoreArgs sa;
  sa.w_ld = m.weights.ld; sa.w_lcs = m.weights.lcs; sa.w_prefix = m.weights.prefix; sa.w_suffix = m.weights.suffix;
  sa.w_case = m.weights.casew;
  sa.w_sum = m.weights.ld + m.weights.lcs + m.weights.prefix + m.weights.suffix + m.weights.casew;  // src/types.rs:69-73
  sa.have_freq = m.have_freq ? 1 : 0;
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
  if (P)
    hipLaunchKernelGGL(k_score_pairs, dim3((P + threads - 1) / threads), dim3(threads), threads * sa.stride, st, P,
                       b->pair_q, b->pair_e, b->q_meta, b->q_rows, dl->ent_meta, dl->ent_rowoff, dl->rows,
                       dl->ent_freq, sa, b->p_score, b->p_meta, b->qmaxfreq);
  HIP_TRY(hipEventRecord(b->ev[3], st));
  // ---- rank ----------------------------------------------------------------------------------------------
  RankArgs ra;
  ra.score_threshold = b->params.score_threshold;
  ra.cutoff_threshold = b->params.cutoff_threshold;
  ra.max_matches = b->params.max_matches;
  ra.freq_weight = b->params.freq_weight;
  ra.have_freq = m.have_freq ? 1 : 0;
  hipLaunchKernelGGL(k_rank, dim3((nq + 3) / 4), dim3(256), 0, st, nq, b->qoff, b->pair_e, b->p_score, dl->ent_freq,
                     dl->ent_order, b->qmaxfreq, ra, b->t_key, b->t_pos, b->r_entry, b->r_dist, b->r_freq, b->r_count);
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
  s.n_pairs = P;
  s.n_class_tests = b->n_class_tests;
  s.n_results = total_results;
  s.n_scan_blocks = b->tiles.size();
  (void)hipEventElapsedTime(&s.ms_scan, b->ev[0], b->ev[1]);
  (void)hipEventElapsedTime(&s.ms_group, b->ev[1], b->ev[2]);
  (void)hipEventElapsedTime(&s.ms_score, b->ev[2], b->ev[3]);
  (void)hipEventElapsedTime(&s.ms_rank, b->ev[3], b->ev[4]);
  (void)hipEventElapsedTime(&s.ms_total, b->ev[0], b->ev[4]);
  return ANX_OK;
}

int batch_fetch(const HostModel& m, const DeviceLexicon* dl, const Batch* b, anx_result** rows, size_t** offs,
                std::string& err) {
  (void)m;
  if (!b->ran) { err = "batch has not been run"; return ANX_EINVAL; }
  HIP_TRY(hipSetDevice(b->device));
  const size_t n = b->n_input;
  size_t* off = static_cast<size_t*>(calloc(n + 1, sizeof(size_t)));
  anx_result* out = static_cast<anx_result*>(malloc(std::max<size_t>(1, b->n_results) * sizeof(anx_result)));
  if (!off || !out) { free(off); free(out); err = "out of memory"; return ANX_EINVAL; }
  if (b->nq && b->n_results) {
    DevRow* d_rows = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_rows), b->n_results * sizeof(DevRow)));
    hipLaunchKernelGGL(k_pack_rows, dim3(((uint32_t)b->nq + 255) / 256), dim3(256), 0, 0, (uint32_t)b->nq, b->qoff,
                       b->r_off, b->r_count, b->r_entry, b->r_dist, b->r_freq, dl->ent_vocab, b->q_orig, d_rows);
    std::vector<DevRow> h(b->n_results);
    std::vector<uint32_t> h_cnt(b->nq);
    HIP_TRY(hipMemcpy(h.data(), d_rows, b->n_results * sizeof(DevRow), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h_cnt.data(), b->r_count, b->nq * sizeof(uint32_t), hipMemcpyDeviceToHost));
    (void)hipFree(d_rows);
    for (size_t s = 0; s < b->nq; ++s) off[b->order[s] + 1] = h_cnt[s];
    for (size_t i = 0; i < n; ++i) off[i + 1] += off[i];
    size_t src = 0;
    for (size_t s = 0; s < b->nq; ++s) {
      size_t dst = off[b->order[s]];
      for (uint32_t i = 0; i < h_cnt[s]; ++i, ++src, ++dst) {
        out[dst].vocab_id = h[src].vocab_id;
        out[dst].dist_score = h[src].dist_score;
        out[dst].freq_score = h[src].freq_score;
        out[dst].via = ANX_NO_VIA;
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
  const size_t P = b->n_pairs;
  anx_pair* res = static_cast<anx_pair*>(malloc(std::max<size_t>(1, P) * sizeof(anx_pair)));
  if (!res) { err = "out of memory"; return ANX_EINVAL; }
  if (P) {
    std::vector<uint32_t> pq(P), pe(P), pm(P), ev(dl->nentries);
    std::vector<double> ps(P);
    HIP_TRY(hipMemcpy(pq.data(), b->pair_q, P * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pe.data(), b->pair_e, P * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pm.data(), b->p_meta, P * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ps.data(), b->p_score, P * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ev.data(), dl->ent_vocab, (size_t)dl->nentries * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < P; ++i) {
      anx_pair& r = res[i];
      r.query = b->order[pq[i]];
      r.vocab_id = ev[pe[i]];
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
  *n = P;
  return ANX_OK;
}

int batch_export_topk(const DeviceLexicon* dl, const Batch* b, void* dst, uint32_t stride, void* stream,
                      std::string& err) {
  if (!b->ran) { err = "batch has not been run"; return ANX_EINVAL; }
  if (!dst || stride == 0) { err = "bad export arguments"; return ANX_EINVAL; }
  HIP_TRY(hipSetDevice(b->device));
  const uint64_t total = (uint64_t)b->nq * stride;
  if (total)
    hipLaunchKernelGGL(k_export_topk, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), (uint32_t)b->nq, stride, b->qoff, b->r_count,
                       b->r_entry, b->r_dist, b->r_freq, dl->ent_vocab, b->q_orig,
                       static_cast<anx_topk_record*>(dst));
  HIP_TRY(hipGetLastError());
  return ANX_OK;
}

void batch_stats(const Batch* b, anx_batch_stats* s) { *s = b->stats; }

void batch_free(Batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  free_pair_buffers(b);
  for (void* p : {(void*)b->q_cv, (void*)b->q_rows, (void*)b->q_meta, (void*)b->q_orig, (void*)b->d_tiles,
                  (void*)b->counters, (void*)b->qcount, (void*)b->qexact, (void*)b->qoff,
                  (void*)b->qcur, (void*)b->qmaxfreq, (void*)b->scan_tmp, (void*)b->raw, (void*)b->r_count,
                  (void*)b->r_off})
    if (p) (void)hipFree(p);
  for (auto& e : b->ev)
    if (e) (void)hipEventDestroy(e);
  delete b;
}

}  // namespace anx
