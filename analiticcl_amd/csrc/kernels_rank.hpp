// kernels_rank.hpp -- K3b+K4: survivor compaction, ranking with crop / cutoff, result packing (k_compact, k_rank, k_pack_rows, k_export_topk)
// Part of the single translation unit engine.hip (included inside namespace anx); gfx950 only.
#pragma once

// K3b: gather the survivors (score >= threshold) into per-query segments of result rows.  With variant lists a
// survivor contributes one row per VariantOf reference (expand_variants, src/lib.rs:1677-1727: score * variant
// score, min(reference frequency, own frequency), via = itself) and itself unless it is TRANSPARENT.
// Order inside a query is arbitrary; ranking uses a total order whose last key is c_ord (= reference order).
struct CompactArgs {
  int have_freq, any_variants;
};
__global__ __launch_bounds__(256) void k_compact(const SurvRec* __restrict__ surv, const uint32_t* __restrict__ sctr,
                                                 uint32_t region_cap, CompactArgs a,
                                                 uint32_t* __restrict__ qcur, const EntRec* __restrict__ ent_rec,
                                                 const uint32_t* __restrict__ ent_var_off,
                                                 const uint32_t* __restrict__ var_target,
                                                 const uint32_t* __restrict__ var_target_freq,
                                                 const double* __restrict__ var_score, SurvRow* __restrict__ c_rows) {
  const uint32_t region = blockIdx.x % SCAN_REGIONS, i = (blockIdx.x / SCAN_REGIONS) * 256 + threadIdx.x;  // 1-D grid, region fastest
  const bool live = i < sctr[region * RC_STRIDE];
  if (!live) return;
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
  uint32_t pos = atomicAdd(&qcur[q], nrows);  // qcur starts as a copy of soff: one random access instead of two
  for (uint32_t j = v0; j < v1; ++j, ++pos) {  // references first, then the item itself (src/lib.rs:1689-1717)
    const uint32_t tf = var_target_freq[j];
    // min(target frequency, own freq_score)
    c_rows[pos] = SurvRow{s * var_score[j], ord | (unsigned long long)(j - v0), var_target[j],
                          a.have_freq ? (tf < f ? tf : f) : (tf < 1u ? tf : 1u), er.vocab, 0u};
  }
  if (self) c_rows[pos] = SurvRow{s, ord | (unsigned long long)(v1 - v0), er.vocab, f, 0xFFFFFFFFu, 0u};
}

// The same without variant lists (every survivor is exactly one row), for blocks of COMPACT_B survivors of one region.
// Returning atomics on scattered addresses sustain only ~25 G/s in total and VALU / LDS are idle here, so the block first
// groups its survivors by query in an LDS hash table (neighbouring survivors come from the same scan tiles: ~10x fewer
// distinct queries than survivors): LDS atomics hand out the rank inside the group, ONE lane per group bumps the query's
// cursor in global memory, everybody adds its rank.  9.75 M global atomics -> ~1.7 M on config 2.
constexpr uint32_t COMPACT_B = 1024, COMPACT_H = 2048;  // hash slots: twice the block size (open addressing, linear probing)
constexpr uint32_t COMPACT_P = 16;                      // blocks per region: each walks its region's survivors in strides
// The grid does not depend on the number of survivors (the host does not know it yet: no read-back between scoring and
// ranking); `row_cap` = rows c_rows can hold: when the batch has more (soff[nq]), nothing is written and the host, which
// sees the total after the run, grows the buffers and repeats compaction and ranking.
__global__ __launch_bounds__(COMPACT_B) void k_compact_grouped(const SurvRec* __restrict__ surv, const uint32_t* __restrict__ sctr,
                                                               uint32_t region_cap, int have_freq, uint32_t* __restrict__ qcur,
                                                               const EntRec* __restrict__ ent_rec, SurvRow* __restrict__ c_rows,
                                                               const uint32_t* __restrict__ total_rows, uint32_t row_cap, uint32_t* __restrict__ overflow) {
  __shared__ uint32_t h_key[COMPACT_H], h_cnt[COMPACT_H], h_base[COMPACT_H];
  if (*total_rows > row_cap) return;
  const uint32_t region = blockIdx.x % SCAN_REGIONS;
  uint32_t fill = sctr[region * RC_STRIDE];
  if (fill > region_cap) {  // survivors were dropped: the per-query counts no longer match the records; ranking is skipped and
    fill = region_cap;      // the host repeats the run with a larger survivor list
    if (threadIdx.x == 0) *overflow = 1u;
  }
  for (uint32_t i0 = (blockIdx.x / SCAN_REGIONS) * COMPACT_B; i0 < fill; i0 += (gridDim.x / SCAN_REGIONS) * COMPACT_B) {  // block-uniform (COMPACT_P blocks per region; the small path launches fewer)
    for (uint32_t h = threadIdx.x; h < COMPACT_H; h += COMPACT_B) { h_key[h] = 0xFFFFFFFFu; h_cnt[h] = 0; }
    __syncthreads();
    const uint32_t i = i0 + threadIdx.x;
    const bool live = i < fill;
    SurvRec sr{0u, 0u, 0.0};
    uint32_t slot = 0, rank = 0;
    bool owner = false;
    if (live) sr = surv[(size_t)region * region_cap + i];
    // (requested here: the gather travels while the block groups its survivors and waits for the cursors' returning atomics)
    EntRec er{};
    if (live) er = ent_rec[sr.e];
    {
      // The survivors of a query sit next to each other (k_filter_score appends them in pair-list order, a scan tile's queries
      // one after the other): a RUN of equal queries among neighbouring lanes is inserted by its first lane alone, with the run's
      // length -- until round 4 every lane did its own compare-and-swap and increment on the run's ONE table word (10 same-address
      // LDS atomics in a row on BASELINE configs[1]: 56 % of the kernel's LDS cycles were conflicts).
      const uint32_t lane = threadIdx.x & 63u;
      const uint32_t qprev = (uint32_t)__shfl_up((int)sr.q, 1);
      const bool head = live && (lane == 0u || qprev != sr.q);
      const unsigned long long hm = __ballot(head), lm = __ballot(live);
      const unsigned long long below = hm & ((2ull << lane) - 1ull);                     // heads at or below this lane
      const uint32_t start = live ? 63u - (uint32_t)__clzll((long long)below) : 0u;    // (a live lane has a head at or below it)
      const unsigned long long above = hm & ~((2ull << lane) - 1ull);                    // the next run's head
      const uint32_t nlive = (uint32_t)__popcll(lm);
      const uint32_t end = above ? (uint32_t)__ffsll((long long)above) - 1u : nlive;    // (the live lanes are a prefix of the wave)
      uint32_t base_rank = 0;
      if (head) {
        slot = (sr.q * 2654435761u) >> 21;  // 11 bits
        for (;;) {
          const uint32_t prev = atomicCAS(&h_key[slot], 0xFFFFFFFFu, sr.q);
          if (prev == 0xFFFFFFFFu) { owner = true; break; }
          if (prev == sr.q) break;
          slot = (slot + 1u) & (COMPACT_H - 1u);
        }
        base_rank = atomicAdd(&h_cnt[slot], end - lane);
      }
      slot = (uint32_t)__shfl((int)slot, (int)start);
      rank = (uint32_t)__shfl((int)base_rank, (int)start) + (lane - start);
    }
    __syncthreads();
    if (owner) h_base[slot] = atomicAdd(&qcur[sr.q], h_cnt[slot]);  // qcur starts as a copy of soff
    __syncthreads();
    if (live) {
      c_rows[h_base[slot] + rank] = SurvRow{sr.score, (unsigned long long)er.order << 20, er.vocab, have_freq ? er.freq : 1u, 0xFFFFFFFFu, 0u};
    }
    __syncthreads();  // the table is cleared for the next chunk
  }
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
// key <= best / cutoff_threshold (and key < best) lies behind the first such row and is cut; with freq_weight == 0 the crop rule
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
  // Only with freq_weight == 0: then the crop rule compares like with like and can never cut before the cutoff point.  With a
  // frequency weight its tie branch compares dist_score with the (weighted) score of row max_matches (src/lib.rs:1558-1566),
  // which may well be a row the cutoff would drop, and cuts EARLIER than the cutoff point -- the crop has to see the whole list.
  const bool prune = a.cutoff_threshold >= 1.0 && !expanded && !score_weighted && n <= 0xFFFFu;
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
  // (wave-uniform) every row of the list is some lane's `mine`: the ranked row then comes out of the registers of the lane that loaded
  // it (shuffles inside the group) instead of a second gather of c_rows -- one dependent memory round trip less per wave.  The loop
  // below therefore runs with every lane (trip count nloop / G, lanes without a row masked by `have`).
  const bool one_row_per_lane = nloop <= (uint32_t)G;
  n = kept;
  if (valid && n == 0 && gl == 0) r_count[q] = 0;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  // ---- rank by counting -------------------------------------------------------------------------------
  const bool full = score_weighted || a.max_matches == 0 || expanded;
  const uint32_t M = full ? n : (uint32_t)min((uint64_t)n, a.max_matches + 1);
  for (uint32_t base = 0; base < nloop; base += G) {
    const uint32_t i = base + (uint32_t)gl;
    const bool have = i < n;
    const double ki = have ? s_key[i] : 0.0;
    const uint32_t fi = have ? s_freq[i] : 0u;
    const unsigned long long oi = have ? s_ord[i] : 0ull;
    uint32_t rank = have ? 0u : 0xFFFFFFFFu;
    if (have) {
    // every comparison is evaluated and the flags combined with bit operations: short-circuit && / || compiled to nested
    // exec-masked branches, four per candidate; the wave-uniform choice of the order is taken outside the loop
    if (sort_weighted) {
      for (uint32_t j = 0; j < n; ++j) {
        const double kj = s_key[j];
        const unsigned long long oj = s_ord[j];
        rank += (uint32_t)((kj > ki) | ((kj == ki) & (oj < oi)));
      }
    } else {
      for (uint32_t j = 0; j < n; ++j) {
        const double kj = s_key[j];
        const uint32_t fj = s_freq[j];
        const unsigned long long oj = s_ord[j];
        rank += (uint32_t)((kj > ki) | ((kj == ki) & ((fj > fi) | ((fj == fi) & (oj < oi)))));
      }
    }
    }
    uint32_t r_vocab = 0, r_via = 0;
    double r_score = 0.0;
    if (one_row_per_lane) {
      const int from = gshift + (have ? (int)s_src[i] : gl);   // the lane whose `mine` is this slot's source row
      const unsigned long long sbits = (unsigned long long)__double_as_longlong(mine.score);
      r_vocab = (uint32_t)__shfl((int)mine.vocab, from);
      r_via = (uint32_t)__shfl((int)mine.via, from);
      const uint32_t s_lo = (uint32_t)__shfl((int)(uint32_t)sbits, from), s_hi = (uint32_t)__shfl((int)(uint32_t)(sbits >> 32), from);
      r_score = __longlong_as_double((long long)((unsigned long long)s_lo | (unsigned long long)s_hi << 32));
    } else if (have && rank < M) {
      const SurvRow r = c_rows[seg0 + s_src[i]];
      r_vocab = r.vocab; r_via = r.via; r_score = r.score;
    }
    if (have && rank < M) {
      // (without frequency information max_freq is 1 or 0: x / 1.0 == x, no f64 division)
      const double ff = (a.have_freq && max_freq > 0.0) ? (double)fi / max_freq : (double)fi;
      r_rows[seg0 + rank] = DevRow{r_vocab, a.any_variants ? r_via : 0xFFFFFFFFu, r_score, ff};
      if (rank < (uint32_t)G) { s_sdist[rank] = r_score; s_sfreq[rank] = ff; }
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
    // the best row's score over the cutoff: with `prune` it is `thr` from above, bit for bit (the sort key of a row is its
    // result_score there, computed by the same expression from the same operands, and rank 0 holds the largest key)
    double cutthr = thr;
    if (docut && !prune) cutthr = result_score(s_sdist[0], s_sfreq[0], fw) / a.cutoff_threshold;
    const unsigned long long cut = (__ballot(have && docut && gl >= 1 && (uint32_t)gl < len && si <= cutthr) >> gshift) & gmask;
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
// SIMPLE: no variant lists and freq_weight == 0 (the common model): the expansion / dedup and the weighted-score branches fold away
template <bool SIMPLE>
__global__ __launch_bounds__(256) void k_rank(uint32_t nq, const uint32_t* __restrict__ soff,
                                              const SurvRow* __restrict__ c_rows,
                                              const uint32_t* __restrict__ qmaxfreq,
                                              const uint32_t* __restrict__ qexpand, RankArgs aa,
                                              double* __restrict__ t_key, DevRow* __restrict__ r_rows,
                                              uint32_t* __restrict__ r_count, uint32_t row_cap, const uint32_t* __restrict__ overflow) {
  RankArgs a = aa;
  if (SIMPLE) { a.any_variants = 0; a.freq_weight = 0.0f; }
  __shared__ __attribute__((aligned(16))) uint8_t s_raw[4 * RANK_WAVE_BYTES];
  // more candidate rows than c_rows / r_rows hold, or survivor records dropped (k_compact_grouped): the host grows the buffers
  // and repeats the run; nothing of this one is used
  if (soff[nq] > row_cap || *overflow) return;
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
  {  // the queries with <= 16 candidate rows: side by side, 16 lanes each (the others are skipped here)
    const int grp = lane >> 4, gl = lane & 15;
    // per group: 16 keys (8 B), 16 order keys (8 B), 16 freqs (4 B), 16 source rows (2 B), 16 + 16 ranked heads (8 B) = 608 B
    uint8_t* gb = wl + grp * 608;
    const uint32_t ng = (uint32_t)__shfl((int)my_n, grp);
    rank_query<16, 16>(qbase + grp, qbase + grp < nq && ng <= 16u, gl, grp * 16, 16u, reinterpret_cast<double*>(gb),
                       reinterpret_cast<unsigned long long*>(gb + 128), reinterpret_cast<uint32_t*>(gb + 256),
                       reinterpret_cast<uint16_t*>(gb + 320), reinterpret_cast<double*>(gb + 352), reinterpret_cast<double*>(gb + 480),
                       (uint32_t)__shfl((int)my_seg0, grp), ng, (uint32_t)__shfl((int)my_maxf, grp),
                       (uint32_t)__shfl((int)my_qex, grp), c_rows, a, t_key, r_rows, r_count);
  }
  if (nmax > 16) {  // wave-uniform
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    // lists of 17..32 rows (8 % of the queries of config 2): two side by side, 32 lanes each
    uint32_t mask32 = 0;  // wave-uniform
    for (int k = 0; k < RANK_QPW; ++k) {
      const uint32_t nk = (uint32_t)__builtin_amdgcn_readlane((int)my_n, k);
      if (qbase + k < nq && nk > 16u && nk <= 32u) mask32 |= 1u << k;
    }
    while (mask32) {
      const int k0 = __ffs((int)mask32) - 1;
      mask32 &= mask32 - 1u;
      const int k1 = mask32 ? __ffs((int)mask32) - 1 : -1;
      if (k1 >= 0) mask32 &= mask32 - 1u;
      const int half = lane >> 5, gl = lane & 31, kk = half ? k1 : k0;
      const bool valid = kk >= 0;
      const int src = valid ? kk : 0;
      // per half: 32 keys, order keys (8 B each), freqs (4 B), source rows (2 B), 32 + 32 ranked heads (8 B) = 1216 B
      uint8_t* gb = wl + half * 1216;
      rank_query<32, 32>(qbase + (uint32_t)src, valid, gl, half * 32, 32u, reinterpret_cast<double*>(gb),
                         reinterpret_cast<unsigned long long*>(gb + 256), reinterpret_cast<uint32_t*>(gb + 512),
                         reinterpret_cast<uint16_t*>(gb + 640), reinterpret_cast<double*>(gb + 704), reinterpret_cast<double*>(gb + 960),
                         (uint32_t)__shfl((int)my_seg0, src), (uint32_t)__shfl((int)my_n, src), (uint32_t)__shfl((int)my_maxf, src),
                         (uint32_t)__shfl((int)my_qex, src), c_rows, a, t_key, r_rows, r_count);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
    // the longer lists one after the other, 64 lanes each
    for (int k = 0; k < RANK_QPW; ++k) {
      if (qbase + k >= nq) break;  // wave-uniform
      const uint32_t nk = (uint32_t)__builtin_amdgcn_readlane((int)my_n, k);
      if (nk <= 32u) continue;     // wave-uniform: done above
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
// download path: result rows in the CALLER's input order, ready to be copied into the anx_result array
__global__ __launch_bounds__(256) void k_fetch_counts(uint32_t nq, const uint32_t* __restrict__ r_count,
                                                      const uint32_t* __restrict__ q_orig, uint32_t* __restrict__ cnt_orig) {
  const uint32_t s = blockIdx.x * 256 + threadIdx.x;
  if (s < nq) cnt_orig[q_orig[s]] = r_count[s];
}
__global__ __launch_bounds__(256) void k_fetch_rows(uint32_t nq, const uint32_t* __restrict__ soff, const uint32_t* __restrict__ r_count,
                                                    const DevRow* __restrict__ r_rows, const uint32_t* __restrict__ q_orig,
                                                    const uint32_t* __restrict__ off_orig, anx_result* __restrict__ out) {
  const uint32_t s = blockIdx.x * 256 + threadIdx.x;
  if (s >= nq) return;
  const uint32_t n = r_count[s], src = soff[s], dst = off_orig[q_orig[s]];
  for (uint32_t i = 0; i < n; ++i) {
    const DevRow d = r_rows[src + i];
    anx_result r;
    r.vocab_id = d.vocab_id;
    r.dist_score = d.dist_score;
    r.freq_score = d.freq_score;
    r.via = d.via == 0xFFFFFFFFu ? ANX_NO_VIA : (uint64_t)d.via;
    out[dst + i] = r;
  }
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


// compact export: rows of query s (sorted order) -> records at off_orig[q_orig[s]] .. (input order, no padding)
__global__ __launch_bounds__(256) void k_export_rows(uint32_t nq, const uint32_t* __restrict__ soff, const uint32_t* __restrict__ r_count,
                                                     const DevRow* __restrict__ r_rows, const uint32_t* __restrict__ q_orig,
                                                     const uint32_t* __restrict__ off_orig, anx_topk_record* __restrict__ out) {
  const uint32_t s = blockIdx.x * 256 + threadIdx.x;
  if (s >= nq) return;
  const uint32_t n = r_count[s], src = soff[s], dst = off_orig[q_orig[s]];
  for (uint32_t i = 0; i < n; ++i) {
    const DevRow d = r_rows[src + i];
    anx_topk_record r;
    r.vocab_id = d.vocab_id;
    r.freq_score = (float)d.freq_score;
    r.dist_score = d.dist_score;
    out[dst + i] = r;
  }
}

// ------------------------------------------------------------------------------------------------
// The small call (engine.hip small_find): at most SMALL_MAX queries in INPUT order (no sort: query s = input s), one
// scan tile per query.  Two single-block kernels replace the prefix-sum launches, the cursor copy, the order scatter and
// the downloads of the batch path: the whole call is eleven launches and ONE host wait.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t SMALL_MAX = 4096;     // inputs per small call
constexpr uint32_t SMALL_T = 1024;       // threads of the single-block kernels (4 queries per thread)
__device__ inline uint32_t small_block_exscan(uint32_t v, uint32_t* s_w, uint32_t* total) {  // exclusive scan over SMALL_T threads
  const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t u = (uint32_t)__shfl_up((int)inc, o);
    if ((int)lane >= o) inc += u;
  }
  if (lane == 63u) s_w[wid] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (uint32_t i = 0; i < SMALL_T / 64u; ++i) {
    if (i < wid) base += s_w[i];
    tot += s_w[i];
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}
// soff = exclusive scan of qsurv (n + 1 entries), qcur = soff (the cursors of k_compact_grouped)
__global__ __launch_bounds__(SMALL_T) void k_small_offsets(const uint32_t* __restrict__ qsurv, uint32_t n, uint32_t* __restrict__ soff, uint32_t* __restrict__ qcur) {
  __shared__ uint32_t s_w[SMALL_T / 64];
  uint32_t v[4], s = 0;
#pragma unroll
  for (uint32_t i = 0; i < 4u; ++i) {
    const uint32_t q = threadIdx.x * 4u + i;
    v[i] = q < n ? qsurv[q] : 0u;
    s += v[i];
  }
  uint32_t tot;
  uint32_t ex = small_block_exscan(s, s_w, &tot);
#pragma unroll
  for (uint32_t i = 0; i < 4u; ++i) {
    const uint32_t q = threadIdx.x * 4u + i;
    if (q < n) { soff[q] = ex; qcur[q] = ex; }
    ex += v[i];
  }
  if (threadIdx.x == 0) soff[n] = tot;
}
// The ranked rows straight into the caller's (pinned, device-visible) result block, in input order: off[n + 1] (u64) and the
// anx_result rows; behind them the counters the host checks the run's capacities with.  ctl[0] = rows written (0xFFFFFFFF: they
// did not fit row_cap), ctl[1 ..]: largest pair-list / survivor / slot-list fill, candidate rows, the overflow flag.
struct SmallCtl { uint32_t rows, maxfill, surv_fill, list_fill, total_surv, overflow, pad0, pad1; };
__global__ __launch_bounds__(SMALL_T) void k_small_fetch(uint32_t n, const uint32_t* __restrict__ soff, const uint32_t* __restrict__ r_count, const DevRow* __restrict__ r_rows,
                                                         const uint32_t* __restrict__ rctr, const uint32_t* __restrict__ sctr, const uint32_t* __restrict__ lctr,
                                                         const uint32_t* __restrict__ counters, unsigned long long* __restrict__ off, anx_result* __restrict__ out, uint32_t row_cap,
                                                         uint32_t crow_cap, SmallCtl* __restrict__ ctl) {
  __shared__ uint32_t s_w[SMALL_T / 64];
  __shared__ uint32_t s_max[3];
  if (threadIdx.x < 3u) s_max[threadIdx.x] = 0u;
  // k_compact_grouped / k_rank did nothing when the candidate rows did not fit or survivor records were dropped (r_count is stale then)
  const bool ranked = soff[n] <= crow_cap && counters[CTR_OVERFLOW] == 0u;
  uint32_t v[4], s = 0;
#pragma unroll
  for (uint32_t i = 0; i < 4u; ++i) {
    const uint32_t q = threadIdx.x * 4u + i;
    v[i] = (ranked && q < n) ? r_count[q] : 0u;
    s += v[i];
  }
  __shared__ uint32_t s_off[SMALL_MAX + 1];
  uint32_t tot;
  uint32_t ex = small_block_exscan(s, s_w, &tot);  // (its barriers also publish s_max = 0)
  const bool fits = tot <= row_cap;
#pragma unroll
  for (uint32_t i = 0; i < 4u; ++i) {
    const uint32_t q = threadIdx.x * 4u + i;
    if (q < n) { if (blockIdx.x == 0) off[q] = ex; s_off[q] = ex; }
    ex += v[i];
  }
  if (threadIdx.x == 0) s_off[n] = tot;
  __syncthreads();
  // The rows go to HOST memory (the caller's pinned block, over PCIe): consecutive lanes write consecutive rows, so that a wave's store
  // is 2 KB of contiguous bytes (per-query loops wrote scattered 32-byte pieces: a thousand queries took 0.2 ms to leave the device).
  // Output row r belongs to the query q with off[q] <= r < off[q + 1]: a binary search of the offsets in LDS.
  if (fits)
    for (uint32_t r = blockIdx.x * SMALL_T + threadIdx.x; r < tot; r += gridDim.x * SMALL_T) {  // (every block has the offsets; the rows are shared out)
      uint32_t lo = 0, hi = n;  // the last q with s_off[q] <= r
      while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_off[mid] <= r) lo = mid; else hi = mid;
      }
      const DevRow d = r_rows[soff[lo] + (r - s_off[lo])];
      anx_result o;
      o.vocab_id = d.vocab_id;
      o.dist_score = d.dist_score;
      o.freq_score = d.freq_score;
      o.via = d.via == 0xFFFFFFFFu ? ANX_NO_VIA : (uint64_t)d.via;
      out[r] = o;
    }
  if (blockIdx.x != 0) return;
  if (threadIdx.x < SCAN_REGIONS) {
    atomicMax(&s_max[0], rctr[threadIdx.x * RC_STRIDE + RC_RAW]);
    atomicMax(&s_max[1], sctr[threadIdx.x * RC_STRIDE]);
    atomicMax(&s_max[2], max(max(lctr[threadIdx.x * RC_STRIDE], lctr[(SCAN_REGIONS + threadIdx.x) * RC_STRIDE]), lctr[(2u * SCAN_REGIONS + threadIdx.x) * RC_STRIDE]));
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    off[n] = tot;
    SmallCtl c;
    c.rows = (fits && ranked) ? tot : 0xFFFFFFFFu;
    c.maxfill = s_max[0]; c.surv_fill = s_max[1]; c.list_fill = s_max[2];
    c.total_surv = soff[n]; c.overflow = counters[CTR_OVERFLOW]; c.pad0 = c.pad1 = 0u;
    *ctl = c;
  }
}
