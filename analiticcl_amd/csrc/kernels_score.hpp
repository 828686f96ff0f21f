// kernels_score.hpp -- K2+K3: prefilter, register-resident and general Damerau-Levenshtein, LCS / prefix / suffix, score (k_filter_score, k_score_fast8, k_score_pairs)
// Part of the single translation unit engine.hip (included inside namespace anx); gfx950 only.
#pragma once

// ------------------------------------------------------------------------------------------------
// K3: score one (query, candidate) pair per lane, straight off the flat pair list.
//   damerau_levenshtein (src/distance.rs:101-179) in its band-limited saturating form:
//   cells with |i-j| > d are d+1, every value saturates at d+1, the transposition term only looks back
//   d rows / d columns (farther ones cost > d).  Identical to the reference for every outcome <= d.
//   Per-lane state lives in LDS: query row, candidate row, a ring of d+2 band rows.
//   longest_common_substring_length / common_prefix_length / common_suffix_length: src/distance.rs:181-231.
//   Score: src/lib.rs:1433-1452 (f64, same association, no FMA contraction).
// ------------------------------------------------------------------------------------------------
struct ScoreArgs {
  const double* quot;  // [33][33] quot[x*33+L] = (double)x / (double)L computed on the host, or nullptr
  int dbg;  // ANX_SCORE_DBG (timing experiments only): 1 skip LCS, 2 skip everything after DL, 4 skip the DL, 8 skip the short round's row gathers
  int store_pairs;  // write the per-slot outputs p_meta / p_score (only the debug view anx_batch_fetch_pairs reads them)
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
// LCS, prefix, suffix (src/lib.rs:1352-1366; score_tail: byte-wise from LDS rows) and case (:1367-1377), the f64 score (:1433-1452),
// max_freq and the survivor count (score_finish).
// freq: the entry's frequency (ent_freq[e]); quot17: nullptr, or the part of a.quot for x, L <= 16 as an LDS table [17][17]
__device__ inline double score_finish(int lq, uint32_t ld, uint32_t lcs, uint32_t pre, uint32_t suf, uint32_t qm, uint32_t em, uint32_t q, uint32_t e,
                                      const ScoreArgs& a, uint32_t freq, const uint32_t* __restrict__ ent_var_off,
                                      uint32_t* __restrict__ qmaxfreq, uint32_t* __restrict__ qsurv, uint32_t* __restrict__ qexpand,
                                      uint32_t& samecase, bool& keep, const double* quot17 = nullptr, uint32_t* rows_out = nullptr);
__device__ inline double score_tail(const uint8_t* S, const uint8_t* T, int lq, int lc, uint32_t ld, uint32_t qm, uint32_t em,
                                    uint32_t q, uint32_t e, const ScoreArgs& a, const uint32_t* __restrict__ ent_freq,
                                    const uint32_t* __restrict__ ent_var_off, uint32_t* __restrict__ qmaxfreq,
                                    uint32_t* __restrict__ qsurv, uint32_t* __restrict__ qexpand, uint32_t& lcs,
                                    uint32_t& pre, uint32_t& suf, uint32_t& samecase, bool& keep) {
  if (a.w_lcs > 0.0 && !(ANX_DBG(a.dbg) & 1)) {
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
  return score_finish(lq, ld, lcs, pre, suf, qm, em, q, e, a, a.have_freq ? ent_freq[e] : 1u, ent_var_off, qmaxfreq, qsurv, qexpand, samecase, keep);
}
__device__ inline double score_finish(int lq, uint32_t ld, uint32_t lcs, uint32_t pre, uint32_t suf, uint32_t qm, uint32_t em, uint32_t q, uint32_t e,
                                      const ScoreArgs& a, uint32_t freq, const uint32_t* __restrict__ ent_var_off,
                                      uint32_t* __restrict__ qmaxfreq, uint32_t* __restrict__ qsurv, uint32_t* __restrict__ qexpand,
                                      uint32_t& samecase, bool& keep, const double* quot17, uint32_t* rows_out) {
  if (a.w_case > 0.0) samecase = ((qm >> 24) & 1u) == ((em >> 8) & 1u);  // src/lib.rs:1367-1377
  // x / L for integers x <= L <= 32 comes from a table of host-computed IEEE quotients (identical bits, no f64 divide)
  const double L = (double)lq;
  const bool tab = a.quot && lq <= 32;
  auto over_L = [&](uint32_t x) {
    if (quot17 && lq <= 16 && x <= 16u) return quot17[x * 17u + (uint32_t)lq];
    return (tab && x <= 32u) ? a.quot[x * 33u + (uint32_t)lq] : (double)x / L;
  };
  const double distance_score = (int)ld > lq ? 0.0 : 1.0 - over_L(ld);
  const double lcs_score = over_L(lcs);
  const double prefix_score = over_L(pre);
  const double suffix_score = over_L(suf);
  const double num = a.w_ld * distance_score + a.w_lcs * lcs_score + a.w_prefix * prefix_score +
                     a.w_suffix * suffix_score + (samecase ? a.w_case : 0.0);
  const double score = a.w_sum == 1.0 ? num : num / a.w_sum;  // x / 1.0 == x
  // max_freq over every DL-surviving instance, before the threshold test (src/lib.rs:1454-1462)
  // (rows_out: the caller adds the two per-query counters itself, see survivor_counts)
  if (!rows_out && !(ANX_DBG(a.dbg) & 16)) atomicMax(&qmaxfreq[q], freq);
  uint32_t nrows = 1;
  if (a.any_variants) {  // variant lists loaded (src/lib.rs:1464-1466, 1510, 1677-1727)
    if (em & 0x200u) qexpand[q] = 1;  // benign race: every writer stores 1
    nrows = (ent_var_off[e + 1] - ent_var_off[e]) + ((em & 0x400u) ? 0u : 1u);  // transparent: references only
  }
  keep = score >= a.score_threshold && nrows;  // src/lib.rs:1475
  if (rows_out) *rows_out = keep ? nrows : 0u;
  else if (keep && !(ANX_DBG(a.dbg) & 16)) atomicAdd(&qsurv[q], nrows);
  return score;
}
// The per-query counters of score_finish for a whole wave (all lanes call this): the DL survivors of a query are neighbours in the
// pair list and in k_filter_score's queue, so ONE lane per run of equal queries adds the run's rows and its largest frequency --
// 64 lanes' atomics on a few words of one cache line were 0.14 of the kernel's 0.63 ms.
__device__ __forceinline__ void survivor_counts(bool has, uint32_t q, uint32_t freq, uint32_t nrows, const ScoreArgs& a,
                                                uint32_t* __restrict__ qmaxfreq, uint32_t* __restrict__ qsurv) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t key = has ? q : 0xFFFFFFFFu;
  const uint32_t kprev = (uint32_t)__shfl_up((int)key, 1);
  const bool head = lane == 0u || kprev != key;
  const unsigned long long hm = __ballot(head);
  const uint32_t start = 63u - (uint32_t)__clzll((long long)(hm & ((2ull << lane) - 1ull)));   // (lane 0 is a head)
  const unsigned long long above = lane == 63u ? 0ull : hm & ~((2ull << lane) - 1ull);
  const bool last = above ? (uint32_t)__ffsll((long long)above) - 1u == lane + 1u : lane == 63u;   // the run's last lane
  // inclusive segmented scans over the run: rows (sum), frequency (max)
  uint32_t rows = has ? nrows : 0u, mf = has ? freq : 0u;
  const bool scan_rows = a.any_variants != 0, scan_freq = a.have_freq != 0;   // (else: one row per kept lane, frequency 1)
  if (!scan_rows) {
    const unsigned long long km = __ballot(rows != 0u);
    const unsigned long long upto = (2ull << lane) - 1ull, from = ~((1ull << start) - 1ull);
    rows = (uint32_t)__popcll(km & upto & from);
  }
  if (scan_rows || scan_freq) {
#pragma unroll
    for (uint32_t off = 1; off < 64u; off <<= 1) {
      const uint32_t ur = (uint32_t)__shfl_up((int)rows, off), uf = (uint32_t)__shfl_up((int)mf, off);
      const bool in = lane >= start + off;
      if (scan_rows) rows += in ? ur : 0u;
      if (scan_freq) mf = in ? max(mf, uf) : mf;
    }
  }
  if (last && has && !(ANX_DBG(a.dbg) & 16)) {
    atomicMax(&qmaxfreq[q], mf);
    if (rows) atomicAdd(&qsurv[q], rows);
  }
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

// (round 6, measured and dropped: taking the transposition term's "an earlier query symbol equals t[j-1]" tests from the match bits of
// the D rows above -- column c of row i and column c+1+a of row i-1-a are the same candidate symbol -- instead of comparing again:
// v_cmp 415 -> 368 in the ISA of k_filter_score<2>, but the masks kept across the exec-masked rows cost more in merges than the
// compares saved: 2028 -> 2094 VALU, 1138 -> 1573 SALU instructions)
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

// ------------------------------------------------------------------------------------------------
// K3, round 6: the same distance by DIAGONALS (Ukkonen / Landau-Vishkin furthest-reaching form) for pairs of <= 16 symbols.
//   L[e][k] = the furthest row i on diagonal k (column j = i + k) with D[i][j] <= e.  D is non-decreasing along a diagonal, so
//   L[e][k] = slide(max(L[e-1][k-1], L[e-1][k] + 1, L[e-1][k+1] + 1, transposition terms)) where slide() follows the equal symbols.
//   The transposition term of src/distance.rs:157-162 (cost (i-l-1) + 1 + (j-db-1): a symbols of s deleted and b symbols of t inserted
//   between the two swapped ones, x = 1 + a + b) only has to be tried at the FURTHEST point r = L[e-x][k'] of its source diagonal
//   k' = k - b + a: from any earlier point of that diagonal the landing row r' + 2 + a is reached by one substitution + a deletions +
//   b insertions from the furthest point as well (r + 1 + a >= r' + 2 + a), at the same cost.  It applies iff s[r] == t[c + 1 + b] and
//   s[r + 1 + a] == t[c] (c = r + k'), and lands on row r + 2 + a; every term so formed is a real edit sequence of the unrestricted
//   distance, so the minimum is the reference's value for every outcome <= D (the reference keeps the LAST matching row / column only,
//   which Lowrance-Wagner prove sufficient).  Result = the smallest e with L[e][lc - lq] >= lq.
//   All of it runs on 16-bit MISMATCH MASKS, one per diagonal of the band: bit i of M[k] = (s[i] != t[i + k]), built from the register
//   words by XOR, a zero-byte test and a 4 x 8-bit dot product that gathers the four flags of a word (v_dot4_u32_u8 with the weights
//   1, 2, 4, 8 / 16, 32, 64, 128); slide(r, k) = r + ctz(M[k] >> r), the symbol tests are single bits of ~M.  No row loop: (D+1)^2 cells
//   instead of (2D+1) x lq, the same instruction count for every length, no exec masks.  Rows beyond a string's end need no clamp:
//   the paddings equal nothing (bits set), every step off a diagonal costs 1 whatever the row, and the final test is ">= lq".
//   Python model + 720 k random pairs against the twin's damerau_levenshtein (alphabets of 2..26 symbols, D = 1..3): HISTORY.md section 15.
// ------------------------------------------------------------------------------------------------
template <bool B7>
__device__ __forceinline__ uint32_t nonzero_byte_flags(uint32_t x) {  // 0x80 in every byte of x that is not zero (B7: every byte of x < 0x80)
  if (B7) return (x + 0x7F7F7F7Fu) & 0x80808080u;
  return (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
}
// bit i (i < 16) = s[i] != t[i + K]; bits 16..31 set.  pad = four bytes that equal nothing of s
template <int K, bool B7>
__device__ __forceinline__ uint32_t diag_mismatch_mask(const uint32_t (&S)[4], const uint32_t (&T)[4], uint32_t pad) {
  static_assert(K >= -15 && K <= 15, "16 symbols");
  constexpr int KM = (K + 16) / 4 - 4, KS = (K + 16) % 4;   // K = 4 * KM + KS, KS in 0..3
  auto word = [&](int i) { return i >= 0 && i < 4 ? T[i >= 0 && i < 4 ? i : 0] : pad; };
  uint32_t tw[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) tw[w] = KS == 0 ? word(w + KM) : __builtin_amdgcn_alignbyte(word(w + KM + 1), word(w + KM), KS);
  const uint32_t lo = __builtin_amdgcn_udot4(nonzero_byte_flags<B7>(S[1] ^ tw[1]), 0x80402010u,
                                             __builtin_amdgcn_udot4(nonzero_byte_flags<B7>(S[0] ^ tw[0]), 0x08040201u, 0u, false), false);
  const uint32_t hi = __builtin_amdgcn_udot4(nonzero_byte_flags<B7>(S[3] ^ tw[3]), 0x80402010u,
                                             __builtin_amdgcn_udot4(nonzero_byte_flags<B7>(S[2] ^ tw[2]), 0x08040201u, 0u, false), false);
  return (lo >> 7) | (hi << 1) | 0xFFFF0000u;   // (the flags are 0x80: the sums come out shifted by 7)
}
template <int D, bool B7>
struct DiagMasks {
  static constexpr uint32_t PM = B7 ? 0x7F7F7F7Fu : 0xFFFFFFFFu;  // B7: symbols < 0x7E, the paddings become 0x7E / 0x7F
  uint32_t M[2 * D + 1];   // [k + D]
  uint32_t S[4], T[4];     // the words the masks are made from
  template <int K>
  __device__ __forceinline__ void fill() {
    M[K + D] = diag_mismatch_mask<K, B7>(S, T, PM);
    if constexpr (K < D) fill<K + 1>();
  }
  // S0: query words padded 0xFE, T0: candidate words padded 0xFF (load_pair)
  __device__ __forceinline__ void build(const uint32_t (&S0)[4], const uint32_t (&T0)[4]) {
#pragma unroll
    for (int w = 0; w < 4; ++w) { S[w] = S0[w] & PM; T[w] = T0[w] & PM; }
    fill<-D>();
  }
  template <int K>
  __device__ __forceinline__ uint32_t far() const { return diag_mismatch_mask<K, B7>(S, T, PM); }   // a diagonal beyond the band
};
// The same masks from the strings' symbol PLANES (kernels_common.hpp symbol_planes16: plane b, bit i = bit b of code + 1, zero from the
// string's end on): M[k] = OR_b (S_b ^ T_b >> k) | (all positions from lq on).  A position of s inside the string against one outside t
// (shifted-in zeros or t's own) differs in some plane, because code + 1 is never zero.  2 x 6 + 1 instructions a diagonal, all at the
// full issue rate (v_lshrrev / v_bitop3): ~30 cycles where the byte rows take ~68 (4 v_alignbyte + 4 v_xad + 4 v_and + 4 v_dot4 + 3).
template <int D>
struct PlaneMasks : PlaneRows {   // (kernels_swar.hpp)
  uint32_t M[2 * D + 1];   // [k + D]
  template <int K>
  __device__ __forceinline__ void fill() {
    M[K + D] = this->template far<K>();
    if constexpr (K < D) fill<K + 1>();
  }
  // qp = q_rec[q][1], cp = e_planes[e]; lq = 0 for a lane without a pair (nothing matches)
  __device__ __forceinline__ void build(const uint4& qp, const uint4& cp, int lq) {
    this->load(qp, cp, lq);
    fill<-D>();
  }
};
// ------------------------------------------------------------------------------------------------
// The tail's string measures from the same masks (pairs of <= 16 symbols, |lq - lc| <= D):
//   common_prefix_length (src/distance.rs:207-218) = the first set bit of M[0];
//   common_suffix_length (:220-231) = the clear bits of M[lc - lq] downwards from bit lq - 1;
//   longest_common_substring_length (:181-205) = the longest run of clear bits on ANY diagonal: the band's 2D + 1 masks two to a
//   register (z & z >> 1 until nothing is left; the upper half holds a negative diagonal, whose bit 0 is never a match, so nothing
//   leaks into the lower half), then the diagonals +-r beyond the band, r = D + 1, ... while some lane of the wave has an overlap
//   min(lq, lc - r) or min(lq - r, lc) above its best run so far (the overlaps only shrink with r: same walk and same stop as
//   score_tail's byte loop).  On configs[1]'s survivors a lane needs r = 3 in 24 %, r = 4 in 7 %, r >= 5 in 2 % of the cases, and a far
//   diagonal holds the longest run 4 times in 28 644.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t runs_of_packed(uint32_t p) {  // longest run of set bits in either half of p (see above), wave-uniform loop
  uint32_t n = 0;
  while (__any(p != 0u)) {
    n += p != 0u;
    p &= p >> 1;
  }
  return n;
}
template <int R, int D, class MT>
__device__ __forceinline__ void lcs_far_diagonals(const MT& dm, int lq, int lc, uint32_t& best) {
  const int ov = max(min(lq, lc - R), min(lq - R, lc));
  if (!__any(ov > (int)best)) return;   // wave-uniform
  const uint32_t zp = ~dm.template far<R>(), zm = ~dm.template far<-R>();
  best = max(best, runs_of_packed(zp | (zm << 16)));
  if constexpr (R < 15) lcs_far_diagonals<R + 1, D, MT>(dm, lq, lc, best);
}
template <int D, class MT>
__device__ __forceinline__ void measures16(const MT& dm, int lq, int lc, const ScoreArgs& a, uint32_t& lcs, uint32_t& pre, uint32_t& suf) {
  if (a.w_prefix > 0.0) pre = (uint32_t)__builtin_ctz(dm.M[D]);
  if (a.w_suffix > 0.0) {
    const int kf = lc - lq;
    uint32_t sel = dm.M[0];
#pragma unroll
    for (int k = -D + 1; k <= D; ++k) {
      sel = kf == k ? dm.M[k + D] : sel;
      asm("" : "+v"(sel));   // (keeps the selects: the compiler otherwise turns the chain into an indexed load of M[] from scratch memory)
    }
    suf = (uint32_t)__builtin_clz(~(~sel << (32 - lq)));   // (1 <= lq <= 16: the low bits of the shifted word are clear)
  }
  if (a.w_lcs > 0.0 && !(ANX_DBG(a.dbg) & 1)) {
    uint32_t best = 0;
    {
      uint32_t p[D + 1];
#pragma unroll
      for (int k = 0; k <= D; ++k) p[k] = ~dm.M[k + D] | (k + 1 <= D ? ~dm.M[D - (k + 1)] << 16 : 0u);
      while (true) {   // wave-uniform
        uint32_t any = 0;
#pragma unroll
        for (int k = 0; k <= D; ++k) any |= p[k];
        if (!__any(any != 0u)) break;
        best += any != 0u;
#pragma unroll
        for (int k = 0; k <= D; ++k) p[k] &= p[k] >> 1;
      }
    }
    lcs_far_diagonals<D + 1, D, MT>(dm, lq, lc, best);
    lcs = best;
  }
}

template <int D, class MT>
__device__ __forceinline__ uint32_t dl_diag(const MT& dm, int lq, int lc) {
  constexpr int BW = 2 * D + 1;
  uint32_t N[BW];
#pragma unroll
  for (int c = 0; c < BW; ++c) N[c] = ~dm.M[c];
  auto slide = [&](uint32_t r, int k) { return r + (uint32_t)__builtin_ctz(dm.M[k + D] >> r); };  // (r <= 16 + D; bits 16.. are set)
  uint32_t L[D + 1][BW];
#pragma unroll
  for (int e = 0; e <= D; ++e)
#pragma unroll
    for (int c = 0; c < BW; ++c) L[e][c] = 0;
  L[0][D] = slide(0u, 0);
#pragma unroll
  for (int e = 1; e <= D; ++e) {
#pragma unroll
    for (int k = -e; k <= e; ++k) {
      uint32_t v = 0;
      if (k - 1 >= -(e - 1) && k - 1 <= e - 1) v = max(v, L[e - 1][k - 1 + D]);          // insertion: same row, next column
      if (k + 1 >= -(e - 1) && k + 1 <= e - 1) v = max(v, L[e - 1][k + 1 + D] + 1u);     // deletion
#pragma unroll
      for (int a = 0; a < e; ++a) {
#pragma unroll
        for (int b = 0; a + b < e; ++b) {
          const int x = 1 + a + b, kp = k - b + a, lim = e - x;
          if (kp < -lim || kp > lim) continue;
          // substitution + a deletions + b insertions from the furthest point of (e - x, kp), one row more if the transposition applies
          // (a = b = 0: the plain substitution term)
          const uint32_t r = L[e - x][kp + D];
          const uint32_t both = N[kp + 1 + b + D] & (N[kp - 1 - a + D] >> (1 + a));
          v = max(v, r + (uint32_t)(1 + a) + ((both >> r) & 1u));
        }
      }
      L[e][k + D] = slide(v, k);
    }
  }
  const int kf = lc - lq;
  uint32_t res = D + 1;
#pragma unroll
  for (int e = D; e >= 0; --e)
#pragma unroll
    for (int k = -e; k <= e; ++k) res = (kf == k && L[e][k + D] >= (uint32_t)lq) ? (uint32_t)e : res;
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
  const uint4* q_rec;       // [Q][2]: {first 16 symbols of the query} {meta, -, -, -}: one 32-B gather instead of two
  const uint4* e_rec;       // [E][2]: {first 16 symbols of the entry} {meta, row offset, freq, -}
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
  const uint4* e_planes;    // [E] {meta, symbol planes}; the queries' planes are q_rec[q][1].yzw
};

// Loads the pair of slot p into registers (NW words per string); returns false for !active.
template <int NW>
struct PairRegs {
  uint32_t q = 0, e = 0, qm = 0, em = 0, freq = 1;
  int lq = 0, lc = 0, d = 0;
  uint32_t S[NW], T[NW];
};
template <int NW>
__device__ inline void load_pair_qe(uint32_t q, uint32_t e, bool active, const PairArgs& A, const ScoreArgs& a, PairRegs<NW>& r);
template <int NW>
__device__ inline void load_pair(uint32_t p, bool active, const PairArgs& A, const ScoreArgs& a, PairRegs<NW>& r) {
  uint2 rp = make_uint2(0u, 0u);
  if (active) rp = A.raw[p];
  load_pair_qe<NW>(rp.x, rp.y & RAW_ENTRY_MASK, active, A, a, r);
}
template <int NW>
__device__ inline void load_pair_qe(uint32_t q, uint32_t e, bool active, const PairArgs& A, const ScoreArgs& a, PairRegs<NW>& r) {
#pragma unroll
  for (int w = 0; w < NW; ++w) { r.S[w] = 0xFEFEFEFEu; r.T[w] = 0xFFFFFFFFu; }
  if (!active) return;
  r.q = q;
  r.e = e;
  const uint4 Q0 = rec32(A.q_rec, r.q)[0], QM = rec32(A.q_rec, r.q)[1];
  const uint4 C0 = rec32(A.e_rec, r.e)[0], CM = rec32(A.e_rec, r.e)[1];
  r.qm = QM.x;
  r.em = CM.x;
  r.freq = a.have_freq ? CM.z : 1u;   // (e_rec[e][1] = {meta, row offset, ent_freq[e], -})
  r.lq = r.qm & 0xFF; r.d = (r.qm >> 16) & 0xFF; r.lc = r.em & 0xFF;
  r.S[0] = Q0.x; r.S[1] = Q0.y; r.S[2] = Q0.z; r.S[3] = Q0.w;
  r.T[0] = C0.x; r.T[1] = C0.y; r.T[2] = C0.z; r.T[3] = C0.w;
  const uint4* qr = A.q_rows + (size_t)r.q * a.qw;
  const uint4* cr = A.rows + CM.y;
#pragma unroll
  for (int w = 1; w < NW / 4; ++w) {
    if (w * 16 < r.lq) { const uint4 Q = qr[w]; r.S[4 * w] = Q.x; r.S[4 * w + 1] = Q.y; r.S[4 * w + 2] = Q.z; r.S[4 * w + 3] = Q.w; }
    if (w * 16 < r.lc) { const uint4 C = cr[w]; r.T[4 * w] = C.x; r.T[4 * w + 1] = C.y; r.T[4 * w + 2] = C.z; r.T[4 * w + 3] = C.w; }
  }
}
// DL of the loaded pair in the row form (the 8-word kernel; all lanes of the wave call this); PAIR_NONE if above the pair's d
template <int D, int NW>
__device__ inline uint32_t dl_of_pair(const PairRegs<NW>& r, bool active) {
  int lqmax = active ? r.lq : 0;
#pragma unroll
  for (int o = 32; o; o >>= 1) lqmax = max(lqmax, __shfl_xor(lqmax, o));
  lqmax = __builtin_amdgcn_readfirstlane(lqmax);
  const uint32_t res = dl_band<D, NW>(r.S, r.T, active ? r.lq : 0, r.lc, lqmax);
  const int diff = r.lq > r.lc ? r.lq - r.lc : r.lc - r.lq;
  return (active && diff <= r.d && res <= (uint32_t)r.d) ? res : PAIR_NONE;  // src/distance.rs:109-130, 173-178
}
// tail of a DL survivor (ld != PAIR_NONE for has) + outputs; all lanes of the wave call this
template <int NW>
__device__ inline void tail_of_pair(uint32_t p, bool has, uint32_t ld, const PairRegs<NW>& r, const PairArgs& A, const ScoreArgs& a,
                                    const SurvOut& so, uint32_t surv_region, uint32_t* __restrict__ lds) {
  uint32_t lcs = 0, pre = 0, suf = 0, samecase = 1;
  double score = __builtin_nan("");
  bool keep = false;
  if (has && !(ANX_DBG(a.dbg) & 2)) {
    uint32_t* mine = lds + (threadIdx.x & 255) * (2 * NW + 1);
#pragma unroll
    for (int w = 0; w < NW; ++w) { mine[w] = r.S[w]; mine[NW + w] = r.T[w]; }
    score = score_tail(reinterpret_cast<const uint8_t*>(mine), reinterpret_cast<const uint8_t*>(mine + NW), r.lq, r.lc, ld, r.qm,
                       r.em, r.q, r.e, a, A.ent_freq, A.ent_var_off, A.qmaxfreq, A.qsurv, A.qexpand, lcs, pre, suf, samecase, keep);
  }
  surv_append(so, surv_region, keep, r.q, r.e, score);
  if (has) {
    if (a.store_pairs) {
      A.p_score[p] = score;
      A.p_meta[p] = ld | (samecase << 7) | (lcs << 8) | (pre << 16) | (suf << 24);
    }
  }
}

// tail_of_pair for pairs of <= 16 symbols whose |lq - lc| <= D: the measures from the diagonal masks, no LDS rows
// (returns keep: the pair goes to the survivor list with `score`; the caller appends it.  Wave-wide: lanes without a pair (has = false)
// hold masks in which nothing matches.  dm: the pair's band masks, DiagMasks or PlaneMasks)
template <int D, class MT>
__device__ __forceinline__ bool tail16(uint32_t p, bool has, uint32_t ld, const MT& dm, int lq, int lc, uint32_t qm, uint32_t em, uint32_t q, uint32_t e, uint32_t freq,
                                       const PairArgs& A, const ScoreArgs& a, const double* quot17, double& score) {
  uint32_t lcs = 0, pre = 0, suf = 0, samecase = 1;
  score = __builtin_nan("");
  bool keep = false;
  if (!(ANX_DBG(a.dbg) & 2)) {
    measures16<D>(dm, has ? lq : 1, has ? lc : 1, a, lcs, pre, suf);
    uint32_t nrows = 0;
    if (has) score = score_finish(lq, ld, lcs, pre, suf, qm, em, q, e, a, freq, A.ent_var_off, A.qmaxfreq, A.qsurv, A.qexpand, samecase, keep, quot17, &nrows);
    survivor_counts(has, q, freq, nrows, a, A.qmaxfreq, A.qsurv);
  }
  if (has && a.store_pairs) {
    A.p_score[p] = score;
    A.p_meta[p] = ld | (samecase << 7) | (lcs << 8) | (pre << 16) | (suf << 24);
  }
  return keep;
}

// Scores the pair in slot p with the register-resident DL of NW words (all lanes of the wave call this; lanes with
// !active only take part in the wave-wide steps).  lds: per-lane staging of both strings for the byte-wise tail.
template <int D, int NW>
__device__ inline void score_fast_pair(uint32_t p, bool active, const PairArgs& A, const ScoreArgs& a, const SurvOut& so,
                                       uint32_t surv_region, uint32_t* __restrict__ lds) {
  PairRegs<NW> r;
  load_pair<NW>(p, active, A, a, r);
  const uint32_t ld = dl_of_pair<D, NW>(r, active);
  if (a.store_pairs && active && ld == PAIR_NONE) A.p_meta[p] = PAIR_NONE | (1u << 7);  // ld = None, samecase = true
  tail_of_pair<NW>(p, ld != PAIR_NONE, ld, r, A, a, so, surv_region, lds);
}

// K2+K3 fused: prefilter of every pair-list slot, the DL of the selected pairs of <= 16 symbols and the tail of its survivors, one
// block per FS_BLK consecutive slots of a region, 256 slots per round (the next round's slots are requested a round ahead).
//   A wave whose slots all carry RAW_PREFILTERED (the scan applied the length test and the band-match bound; both strings <= 16 symbols,
//   d <= the batch's largest d = D) gathers 16 + 16 bytes per pair -- the two symbol rows -- and nothing else: the lengths are read off
//   the rows' paddings, and the query's own d is tested with the survivors.  Any other wave takes the general round: length test
//   |lq - lc| <= d (src/distance.rs:109-130), StopAtExactMatch drop (src/lib.rs:1164-1173), the SWAR band-match bound; selected pairs
//   of <= 16 symbols with d <= D join the inline DL, longer ones go to the slot lists of the 8-word / general kernels.
//   The DL runs on the slots as they lie (dl_diag: ~185 instructions a wave whatever the lengths; 3/4 of a fused tile's slots hold a
//   pair).  Its survivors (~29 %) are queued in LDS -- (query, entry | ld << 26), a queue per WAVE -- and a wave drains its queue in dense
//   rounds of 64 whenever another round could overflow it: both records of the pair, the query's d, LCS / prefix / suffix from the
//   diagonal masks, the f64 score with the x / L quotients from an LDS copy of the host's table, survivor record.  No barrier after
//   the table is loaded: the waves of a block drift apart, one's gathers under another's DL.
//   Round 6: until then every selected pair was queued first (a u16 slot offset), gathered through raw[] + two 32-byte records for the
//   DL in dense rounds and gathered AGAIN the same way for the tail: 5 + 5 vector loads with 64 different lines each per pair-wave,
//   which is what the kernel's time was once the DL itself had become cheap (the texture path takes about a cycle per line).
//   D = 0: no inline DL (d > 3), everything selected goes to the general kernel's list.
constexpr uint32_t FS_BLK = 4096;
constexpr uint32_t FS_SURV = 256;    // entries of a wave's LDS survivor queue
static_assert(FS_BLK <= 65536 && FS_BLK % 256 == 0, "slot offsets inside a block are 16-bit, a round is 256 slots");
static_assert(FS_SURV >= 128 && FS_SURV % 64 == 0, "a round adds at most 64 entries to a queue that is drained above FS_SURV - 64");
struct FilterArgs {
  uint32_t region_shift;
  const uint32_t* rctr;     // region fills of the pair list
  const uint32_t* qexact;
  int stop, enable;
  int use_nw8;              // selected pairs of 17..32 symbols go to list8 (else to the general list)
  uint32_t* counters;
  uint32_t* stat_ctr;       // [SCAN_REGIONS][RC_STRIDE], word 1: selected pairs
  uint32_t fill_cap;        // slots per region the grid covers (a region filled beyond it makes batch_finish repeat the run)
  uint32_t blk;             // slots per block: FS_BLK in the batch path; the small call takes smaller blocks (a multiple of 256, <= FS_BLK): more blocks share its few pairs
};
// WIDE = true: the 8-word prefilter of pairs with a string of 17..32 symbols runs inline (batches with such queries: many
// wide pairs).  WIDE = false (every query <= 16 symbols, so only the few pairs with a 17..19-symbol candidate are wide):
// wide pairs are appended unfiltered to listw and k_filter_wide prefilters them -- the 8-word SWAR state is what sets the
// register count of this kernel.
// arguments the prefilter rounds hardly touch (score weights, survivor / slot lists) live in device memory and are read where
// they are used: as by-value kernel arguments they stayed in SGPRs over the unrolled rounds and pushed 20 SGPRs into VGPR
// lanes (a v_readlane per use)
struct FsCold {
  ScoreArgs a;
  SurvOut so;
  SlotList list8, listg, listw;
};
// number of symbols of a row of 16 whose unused bytes hold `pad` (x = the row's words XOR pad: zero bytes = unused)
template <bool B7>
__device__ __forceinline__ int row_length16(const uint32_t (&x)[4]) {
  const uint32_t lo = __builtin_amdgcn_udot4(nonzero_byte_flags<B7>(x[1]), 0x80402010u, __builtin_amdgcn_udot4(nonzero_byte_flags<B7>(x[0]), 0x08040201u, 0u, false), false);
  const uint32_t hi = __builtin_amdgcn_udot4(nonzero_byte_flags<B7>(x[3]), 0x80402010u, __builtin_amdgcn_udot4(nonzero_byte_flags<B7>(x[2]), 0x08040201u, 0u, false), false);
  return __builtin_ctz(~((lo >> 7) | (hi << 1)));   // (bits 16.. of the sum are clear: at most 16)
}
// MODE: 0 = byte rows, any alphabet; 1 = byte rows, B7 (symbol codes below 0x7E: the one-add zero-byte test); 2 = symbol planes (B7 and
// A <= kSymbolPlanesMaxA): the short round gathers q_rec[q][1] + e_planes[e], the lengths and d come with them
template <int D, bool WIDE, int MODE>
__global__ __launch_bounds__(256) void k_filter_score(FilterArgs f, PairArgs A, const FsCold* __restrict__ cold) {
  constexpr bool B7 = MODE >= 1, PL = MODE == 2;
  const ScoreArgs& a = cold->a;
  const SurvOut& so = cold->so;
  const SlotList &list8 = cold->list8, &listg = cold->listg, &listw = cold->listw;
  constexpr int DD = D > 0 ? D : 1;
  __shared__ uint4 s_sv_all[D > 0 ? 4 * FS_SURV : 1];         // survivors of the inline DL: (query, entry | ld << 26, -, -), after their tail: (query, entry, f64 score) = SurvRec
  __shared__ uint16_t s_svoff_all[D > 0 ? 4 * FS_SURV : 1];   // their slots as offsets from the block's first one (the per-slot debug outputs)
  __shared__ double s_quot[D > 0 ? 17 * 17 : 1];              // a.quot for x, L <= 16
  uint4* const s_sv = s_sv_all + (D > 0 ? (threadIdx.x >> 6) * FS_SURV : 0u);
  uint16_t* const s_svoff = s_svoff_all + (D > 0 ? (threadIdx.x >> 6) * FS_SURV : 0u);
  uint32_t nsv = 0;   // entries in this wave's queue (wave-uniform)
  // 1-D grid, region fastest: blocks that run at the same time append to different regions' counters (a single
  // counter word sustains only ~88 M atomics/s)
  const uint32_t region = blockIdx.x % SCAN_REGIONS, fill = f.rctr[region * RC_STRIDE + RC_RAW], base = (blockIdx.x / SCAN_REGIONS) * f.blk;
  if (base >= fill) return;  // block-uniform
  const uint32_t lim = min(fill, base + f.blk);  // this block's slots: [base, lim)
  const uint32_t p0 = (region << f.region_shift) + base;
  const bool have_quot = D > 0 && a.quot != nullptr;
  if (have_quot)
    for (uint32_t i = threadIdx.x; i < 17u * 17u; i += 256u) s_quot[i] = a.quot[(i / 17u) * 33u + i % 17u];
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63;
  // the wave's queue in dense rounds: records, the query's own d, tail
  auto drain = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the wave's own LDS writes; LDS operations of a wave complete in order)
    const uint32_t m = nsv;
    uint32_t nk = 0;   // entries that go to the survivor list: written back to the head of the queue (never beyond the entry just read)
    for (uint32_t r0 = 0; r0 < m; r0 += 64) {
      const uint32_t i = r0 + lane;
      const bool active = i < m;
      const uint2 ent = active ? *reinterpret_cast<const uint2*>(&s_sv[i]) : make_uint2(0u, 0u);
      const uint32_t p = p0 + (active ? (uint32_t)s_svoff[i] : 0u), ld = (ent.y >> 26) & 3u;
      const uint32_t q = ent.x, e = ent.y & 0x3FFFFFFu;
      double score;
      bool keep;
      if (PL) {
        uint4 qp = make_uint4(0u, 0u, 0u, 0u), cp = qp;
        uint32_t freq = 1u;
        if (active) {
          qp = rec32(A.q_rec, q)[1];
          cp = A.e_planes[e];
          if (a.have_freq) freq = A.ent_freq[e];
        }
        const int lq = qp.x & 0xFF, d = (qp.x >> 16) & 0xFF, lc = cp.x & 0xFF;
        const bool has = active && ld <= (uint32_t)d;   // (|lq - lc| <= ld)
        if (a.store_pairs && active && !has) A.p_meta[p] = PAIR_NONE | (1u << 7);  // ld = None, samecase = true
        PlaneMasks<DD> dm;
        dm.build(qp, cp, has ? lq : 0);
        keep = tail16<DD>(p, has, ld, dm, lq, lc, qp.x, cp.x, q, e, freq, A, a, have_quot ? s_quot : nullptr, score);
      } else {
        PairRegs<4> r;
        load_pair_qe<4>(q, e, active && !(ANX_DBG(a.dbg) & 64), A, a, r);
        if (ANX_DBG(a.dbg) & 64) { r.lq = 8; r.lc = 8; r.d = 2; }
        const bool has = active && ld <= (uint32_t)r.d;   // (|lq - lc| <= ld)
        if (a.store_pairs && active && !has) A.p_meta[p] = PAIR_NONE | (1u << 7);  // ld = None, samecase = true
        DiagMasks<DD, B7> dm;
        dm.build(r.S, r.T);   // (a lane without a pair holds paddings: nothing matches)
        keep = tail16<DD>(p, has, ld, dm, r.lq, r.lc, r.qm, r.em, q, e, r.freq, A, a, have_quot ? s_quot : nullptr, score);
      }
      const unsigned long long km = __ballot(keep);
      if (keep) {
        const unsigned long long sb = (unsigned long long)__double_as_longlong(score);
        s_sv[nk + (uint32_t)__popcll(km & ((1ull << lane) - 1ull))] = make_uint4(q, e, (uint32_t)sb, (uint32_t)(sb >> 32));
      }
      nk += (uint32_t)__popcll(km);
    }
    nsv = 0;
    // one reservation in the region's survivor list for the whole queue (a returning atomic: one wait per drain, not per round),
    // and the records leave as consecutive 16-byte pieces
    if (nk && !(ANX_DBG(a.dbg) & 32)) {  // wave-uniform
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      uint32_t sbase = 0;
      if (lane == 0) sbase = atomicAdd(&so.ctr[region * RC_STRIDE], nk);
      sbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)sbase);
      uint4* const out = reinterpret_cast<uint4*>(so.list) + (size_t)region * so.region_cap;
      for (uint32_t i = lane; i < nk; i += 64)
        if (sbase + i < so.region_cap) out[sbase + i] = s_sv[i];
    }
  };
  uint32_t nselected = 0;  // wave-uniform
  // Straight-line loads: a lane without a pair (beyond the region's fill, unused chunk tail) reads the block's first slot
  uint2 rp_next = A.raw[base + threadIdx.x < lim ? p0 + threadIdx.x : p0];
  for (uint32_t r = 0; r < FS_BLK / 256; ++r) {
    if (base + r * 256 >= lim) break;  // block-uniform
    const uint32_t idx = base + r * 256 + threadIdx.x;
    const uint32_t p = p0 + r * 256 + threadIdx.x;
    const bool live = idx < lim;
    const uint2 rp = rp_next;
    if (base + (r + 1) * 256 < lim) rp_next = A.raw[idx + 256 < lim ? p + 256 : p0];
    const bool invalid = !live || rp.x == RAW_INVALID;  // unused chunk tail
    // RAW_PREFILTERED: a chunk comes from one tile, so whole waves take the short round (wave-uniform test)
    const bool pre = !invalid && (rp.y & RAW_PREFILTERED);
    bool selected = pre, wide = false, inl = false;
    int lq = 0, lc = 0, d = D;
    uint32_t S[4] = {0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu}, T[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};   // MODE 0 / 1
    uint4 qpl = make_uint4(0u, 0u, 0u, 0u), cpl = qpl;   // MODE 2: the pair's plane records
    bool stop_skipped = false;
    const bool general = __any(!invalid && !pre);   // wave-uniform
    if (general) {
      bool skip = invalid;
      if (f.stop)  // StopAtExactMatch: a non-exact class of a query that has an exact one (wave-uniform branch)
        skip = skip || (!pre && !(rp.y & 0x80000000u) && f.qexact[skip ? 0u : rp.x] != 0xFFFFFFFFu);
      stop_skipped = skip && !invalid;
      const uint32_t q = skip ? 0u : rp.x, e = skip ? 0u : (rp.y & RAW_ENTRY_MASK);
      // the first 16 symbols of both strings come with the records (rows are padded with bytes that equal nothing:
      // query 0xFE, candidate 0xFF); a lane without a pair reads record 0, its verdict is masked by `selected`
      const uint4 Q = rec32(A.q_rec, q)[0];  // 32-B records: one line each
      const uint4 QM = rec32(A.q_rec, q)[1];
      const uint4 C = rec32(A.e_rec, e)[0];
      const uint4 CM = rec32(A.e_rec, e)[1];
      const uint32_t crow = CM.y;
      lq = QM.x & 0xFF; d = (QM.x >> 16) & 0xFF; lc = CM.x & 0xFF;
      const int diff = lq > lc ? lq - lc : lc - lq;
      selected = pre || (!skip && diff <= d);
      const bool filt = selected && !pre && f.enable && d <= 3 && lq <= 32 && lc <= 32;
      wide = filt && (lq > 16 || lc > 16);
      if (WIDE && __any(wide)) {  // wave-uniform, rare: some pair of the wave has a string of 17..32 symbols
        uint32_t q8[8] = {Q.x, Q.y, Q.z, Q.w, 0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu};
        uint32_t c10[10] = {0xFFFFFFFFu, C.x, C.y, C.z, C.w, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        if (wide && lq > 16) { const uint4 Q1 = A.q_rows[(size_t)q * a.qw + 1]; q8[4] = Q1.x; q8[5] = Q1.y; q8[6] = Q1.z; q8[7] = Q1.w; }
        if (wide && lc > 16) { const uint4 C1 = A.rows[crow + 1]; c10[5] = C1.x; c10[6] = C1.y; c10[7] = C1.z; c10[8] = C1.w; }
        if (band_bound_rejects<8>(q8, c10, filt, d, lq, lc)) selected = false;
      } else if (__any(filt && !wide)) {
        // as many 4-symbol words as the longest string of the WAVE needs (its 64 slots come from one scan chunk, i.e. one tile
        // and one query length)
        const bool f4 = filt && !wide;
        const int ml = lq > lc ? lq : lc;
        // B7 (alphabets of <= 124 classes): symbols < 0x7E, the paddings 0xFE / 0xFF masked down to 0x7E / 0x7F
        constexpr uint32_t M = B7 ? 0x7F7F7F7Fu : 0xFFFFFFFFu;
        if (__any(f4 && ml > 12)) {
          const uint32_t q4[4] = {Q.x & M, Q.y & M, Q.z & M, Q.w & M}, c6[6] = {M, C.x & M, C.y & M, C.z & M, C.w & M, M};
          if (band_bound_rejects<4, B7>(q4, c6, f4, d, lq, lc)) selected = false;
        } else if (__any(f4 && ml > 8)) {
          const uint32_t q3[3] = {Q.x & M, Q.y & M, Q.z & M}, c5[5] = {M, C.x & M, C.y & M, C.z & M, M};
          if (band_bound_rejects<3, B7>(q3, c5, f4, d, lq, lc)) selected = false;
        } else {
          const uint32_t q2[2] = {Q.x & M, Q.y & M}, c4[4] = {M, C.x & M, C.y & M, M};
          if (band_bound_rejects<2, B7>(q2, c4, f4, d, lq, lc)) selected = false;
        }
      }
      inl = selected && D > 0 && (pre || (lq <= 16 && lc <= 16 && d <= D));   // (pre: d <= the batch's largest d = D)
      if (inl && !PL) { S[0] = Q.x; S[1] = Q.y; S[2] = Q.z; S[3] = Q.w; T[0] = C.x; T[1] = C.y; T[2] = C.z; T[3] = C.w; }
      if (inl && PL) { qpl = QM; cpl = A.e_planes[e]; }
    } else if (D > 0 && __any(pre)) {
      inl = pre;
      if (PL) {
        if (pre && !(ANX_DBG(a.dbg) & 8)) { qpl = rec32(A.q_rec, rp.x)[1]; cpl = A.e_planes[rp.y & 0x3FFFFFFu]; }
        lq = qpl.x & 0xFF; d = pre ? (int)((qpl.x >> 16) & 0xFF) : D; lc = cpl.x & 0xFF;
      } else {
        if (pre && !(ANX_DBG(a.dbg) & 8)) {
          const uint4 Q = rec32(A.q_rec, rp.x)[0], C = rec32(A.e_rec, rp.y & 0x3FFFFFFu)[0];
          S[0] = Q.x; S[1] = Q.y; S[2] = Q.z; S[3] = Q.w; T[0] = C.x; T[1] = C.y; T[2] = C.z; T[3] = C.w;
        }
        const uint32_t xs[4] = {(S[0] ^ 0xFEFEFEFEu), (S[1] ^ 0xFEFEFEFEu), (S[2] ^ 0xFEFEFEFEu), (S[3] ^ 0xFEFEFEFEu)};
        const uint32_t xt[4] = {~T[0], ~T[1], ~T[2], ~T[3]};
        lq = row_length16<false>(xs);   // (lanes without a pair: 0)
        lc = row_length16<false>(xt);
      }
    }
    const bool tow = !WIDE && wide;  // prefiltered later by k_filter_wide (which also counts it as selected if it passes)
    if (a.store_pairs && live && !selected && !tow)  // skipped (tail / StopAtExactMatch) or rejected: ld = None, samecase = true
      A.p_meta[p] = (invalid || stop_skipped) ? META_SKIPPED : (PAIR_NONE | (1u << 7));
    const bool to8 = selected && !inl && !tow && f.use_nw8 && D > 0 && lq <= 32 && lc <= 32 && d <= D;
    const bool tog = selected && !inl && !tow && !to8;
    if (general || D == 0) {   // (a short round with the inline DL lists nothing: every pair it has is inline)
      slot_append(list8, region, to8, p);
      slot_append(listg, region, tog, p);
      if (!WIDE) slot_append(listw, region, tow, p);
      if (f.stop) {  // scored pairs = pairs emitted by the scan minus the ones StopAtExactMatch drops
        const unsigned long long ms = __ballot(stop_skipped);
        if (lane == 0 && ms) atomicAdd(&f.counters[CTR_SKIPPED], (uint32_t)__popcll(ms));
      }
    }
    nselected += (uint32_t)__popcll(__ballot(selected && !tow));
    if (D > 0) {
      if (ANX_DBG(a.dbg) & 4) {   // timing: the rows are consumed, no DL
        asm volatile("" :: "v"(S[0]), "v"(S[1]), "v"(S[2]), "v"(S[3]), "v"(T[0]), "v"(T[1]), "v"(T[2]), "v"(T[3]), "v"(lq), "v"(lc), "v"(qpl.y), "v"(qpl.z), "v"(qpl.w), "v"(cpl.y), "v"(cpl.z), "v"(cpl.w));
      } else if (__any(inl)) {  // wave-uniform
        uint32_t res;
        if (PL) {
          PlaneMasks<DD> dm;
          dm.build(qpl, cpl, inl ? lq : 0);
          res = dl_diag<DD>(dm, lq, lc);
        } else {
          DiagMasks<DD, B7> dm;
          dm.build(S, T);
          res = dl_diag<DD>(dm, lq, lc);
        }
        const int diff = lq > lc ? lq - lc : lc - lq;
        // src/distance.rs:109-130, 173-178 (short round: d = D here, the query's own d is tested when the queue is drained)
        const bool surv = inl && diff <= d && res <= (uint32_t)d && !(ANX_DBG(a.dbg) & 2);
        if (a.store_pairs && inl && !surv) A.p_meta[p] = PAIR_NONE | (1u << 7);  // ld = None, samecase = true
        const unsigned long long ms = __ballot(surv);
        if (ms) {  // wave-uniform
          if (surv) {
            const uint32_t pos = nsv + (uint32_t)__popcll(ms & ((1ull << lane) - 1ull));
            *reinterpret_cast<uint2*>(&s_sv[pos]) = make_uint2(rp.x, (rp.y & 0x3FFFFFFu) | (res << 26));
            s_svoff[pos] = (uint16_t)(r * 256 + threadIdx.x);
          }
          nsv += (uint32_t)__popcll(ms);
        }
      }
      if (nsv > FS_SURV - 64u || base + (r + 1) * 256 >= lim) drain();  // wave-uniform: another round could overflow the queue / the last round
    }
  }
  if (lane == 0 && nselected) atomicAdd(&f.stat_ctr[region * RC_STRIDE + 1], nselected);
}

constexpr uint32_t LIST_P = 64;  // blocks per region of the slot-list kernels (the kernels stride by gridDim.x / SCAN_REGIONS: the small path launches fewer)
// The 8-word band-match prefilter of the wide pairs k_filter_score<D, false> deferred (listw): survivors go to the slot
// list of the 8-word kernel (fastD > 0 and the batch uses it) or of the general kernel.
// (body + launch wrapper: the small call runs the three slot-list kernels as ONE launch, k_small_lists; region = the list's region, the
// block takes rounds blk0, blk0 + blkstep, ... of 256 entries)
__device__ inline void filter_wide_body(const SlotList& in, const FilterArgs& f, const PairArgs& A, const ScoreArgs& a, int fastD, const SlotList& list8, const SlotList& listg,
                                        uint32_t region, uint32_t blk0, uint32_t blkstep) {
  const uint32_t n = min(in.ctr[region * RC_STRIDE], in.region_cap), lane = threadIdx.x & 63;  // (a list filled beyond its capacity makes the host repeat the run)
  uint32_t nselected = 0;  // wave-uniform
  for (uint32_t blk = blk0; blk * 256 < n; blk += blkstep) {  // block-uniform
    const uint32_t i = blk * 256 + threadIdx.x;
    const bool active = i < n;
    const uint32_t p = active ? in.list[(size_t)region * in.region_cap + i] : 0u;
    uint32_t q8[8] = {0u, 0u, 0u, 0u, 0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu};
    uint32_t c10[10] = {0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    int d = 0, lq = 0, lc = 0;
    if (active) {
      const uint2 rp = A.raw[p];
      const uint32_t q = rp.x, e = rp.y & RAW_ENTRY_MASK;
      const uint4 Q = rec32(A.q_rec, q)[0], QM = rec32(A.q_rec, q)[1], C = rec32(A.e_rec, e)[0], CM = rec32(A.e_rec, e)[1];
      lq = QM.x & 0xFF; d = (QM.x >> 16) & 0xFF; lc = CM.x & 0xFF;
      q8[0] = Q.x; q8[1] = Q.y; q8[2] = Q.z; q8[3] = Q.w;
      c10[1] = C.x; c10[2] = C.y; c10[3] = C.z; c10[4] = C.w;
      if (lq > 16) { const uint4 Q1 = A.q_rows[(size_t)q * a.qw + 1]; q8[4] = Q1.x; q8[5] = Q1.y; q8[6] = Q1.z; q8[7] = Q1.w; }
      if (lc > 16) { const uint4 C1 = A.rows[CM.y + 1]; c10[5] = C1.x; c10[6] = C1.y; c10[7] = C1.z; c10[8] = C1.w; }
    }
    const bool selected = active && !band_bound_rejects<8>(q8, c10, active, d, lq, lc);
    if (a.store_pairs && active && !selected) A.p_meta[p] = PAIR_NONE | (1u << 7);  // rejected: ld = None, samecase = true
    const bool to8 = selected && f.use_nw8 && fastD > 0 && lq <= 32 && lc <= 32 && d <= fastD;
    slot_append(list8, region, to8, p);
    slot_append(listg, region, selected && !to8, p);
    nselected += (uint32_t)__popcll(__ballot(selected));
  }
  if (lane == 0 && nselected) atomicAdd(&f.stat_ctr[region * RC_STRIDE + 1], nselected);
}
__global__ __launch_bounds__(256) void k_filter_wide(SlotList in, FilterArgs f, PairArgs A, ScoreArgs a, int fastD, SlotList list8, SlotList listg) {
  filter_wide_body(in, f, A, a, fastD, list8, listg, blockIdx.x % SCAN_REGIONS, blockIdx.x / SCAN_REGIONS, gridDim.x / SCAN_REGIONS);
}
// the selected pairs with a string of 17..32 symbols (list8 of k_filter_score)
template <int D>
__device__ inline void score_fast8_body(const SlotList& in, const PairArgs& A, const ScoreArgs& a, const SurvOut& so, uint32_t* __restrict__ s_str, uint32_t region, uint32_t blk0, uint32_t blkstep) {
  // LIST_P blocks per region walk the region's slot list in strides: the list fills are only known on the device, and a
  // grid sized for the fullest possible list would consist of ~400 k empty blocks (0.08 ms of dispatch on config 2)
  const uint32_t n = min(in.ctr[region * RC_STRIDE], in.region_cap);
  for (uint32_t blk = blk0; blk * 256 < n; blk += blkstep) {  // block-uniform
    const uint32_t i = blk * 256 + threadIdx.x;
    const bool active = i < n;
    score_fast_pair<D, 8>(active ? in.list[(size_t)region * in.region_cap + i] : 0u, active, A, a, so, region, s_str);
    __syncthreads();  // s_str is reused by the next round
  }
}
template <int D>
__global__ __launch_bounds__(256) void k_score_fast8(SlotList in, PairArgs A, ScoreArgs a, SurvOut so) {
  __shared__ uint32_t s_str[256 * 17];
  score_fast8_body<D>(in, A, a, so, s_str, blockIdx.x % SCAN_REGIONS, blockIdx.x / SCAN_REGIONS, gridDim.x / SCAN_REGIONS);
}

__device__ inline void score_pairs_body(const SlotList& in, const PairArgs& A, const ScoreArgs& a, const SurvOut& so, uint32_t* __restrict__ lds32, uint32_t region, uint32_t blk0, uint32_t blkstep) {
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
  const uint32_t nsel = min(in.ctr[region * RC_STRIDE], in.region_cap);
  for (uint32_t blk = blk0; blk * blockDim.x < nsel; blk += blkstep) {  // block-uniform; see k_score_fast8
  const uint32_t i_sel = blk * blockDim.x + threadIdx.x;
  bool keep = false;
  uint32_t kq = 0, ke = 0;
  double kscore = 0.0;
  if (i_sel < nsel) {
    const uint32_t p = in.list[(size_t)region * in.region_cap + i_sel];
    const uint2 rp = raw[p];
    const uint32_t q = rp.x, e = rp.y & RAW_ENTRY_MASK;
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
        if (res <= (uint32_t)d && !(ANX_DBG(a.dbg) & 2)) {  // src/distance.rs:173-178
          ld = res;
          score = score_tail(S, T, lq, lc, ld, qm, em, q, e, a, ent_freq, ent_var_off, qmaxfreq, qsurv, qexpand, lcs, pre, suf, samecase, keep);
          kq = q; ke = e; kscore = score;
        }
      }
    }
    if (a.store_pairs) {
      A.p_score[p] = score;
      A.p_meta[p] = ld | (samecase << 7) | (lcs << 8) | (pre << 16) | (suf << 24);
    }
  }
  surv_append(so, region, keep, kq, ke, kscore);
  }
}
__global__ void k_score_pairs(SlotList in, PairArgs A, ScoreArgs a, SurvOut so) {
  extern __shared__ uint32_t lds32[];
  score_pairs_body(in, A, a, so, lds32, blockIdx.x % SCAN_REGIONS, blockIdx.x / SCAN_REGIONS, gridDim.x / SCAN_REGIONS);
}

// The small call (small_path.hpp): the three slot-list kernels as ONE launch, one block of 256 threads per pair-list region.  The lists of
// a region are appended to by that region's blocks only, and k_filter_wide's output (list8 / listg entries of the block's own region) is
// consumed by the same block after a barrier -- so no second launch is needed between them (a small call is launch-bound).
struct SmallListArgs {
  SlotList lw, l8, lg;
  int do_wide, do_fast8, fastD;
};
template <int D>
__global__ __launch_bounds__(256) void k_small_lists(SmallListArgs L, FilterArgs f, PairArgs A, ScoreArgs a, SurvOut so) {
  extern __shared__ uint32_t lds32[];
  __shared__ uint32_t s_str[D > 0 ? 256 * 17 : 1];
  const uint32_t region = blockIdx.x;
  if (L.do_wide) {
    filter_wide_body(L.lw, f, A, a, L.fastD, L.l8, L.lg, region, 0u, 1u);
    __threadfence();     // (the entries this block appended to list8 / listg, and their counters, before anybody of the block reads them)
    __syncthreads();
  }
  if (D > 0 && L.do_fast8) score_fast8_body<(D > 0 ? D : 1)>(L.l8, A, a, so, s_str, region, 0u, 1u);
  score_pairs_body(L.lg, A, a, so, lds32, region, 0u, 1u);
}
