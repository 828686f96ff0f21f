// kernels_scan.hpp -- K1: signature-pruned anagram scan (k_scan_bits / k_scan_sad)
// Part of the single translation unit engine.hip (included inside namespace anx); gfx950 only.
#pragma once

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
enum { RC_RAW = 0, RC_VALID = 1, RC_FUSED = 2 /* pairs the fused band-match filter tested */, RC_TESTS = 4 /* u64 per scan kind at 4 + 2*kind */,
       RC_ADJ = 14 /* rows of adjacency lists streamed by the region's tiles */, RC_ADJ_FIRST = 15 /* ... by the first tile of every (length, signature) group */ };

struct WaveOut {
  uint32_t base, left;  // unused part of the current chunk of the pair list (wave-uniform)
  uint32_t emitted;     // pairs appended by this wave (wave-uniform)
  uint32_t nbase;       // chunk reserved by the last wave_reserve when the appended run spills over
  uint32_t split;       // run indices < split go to [base..), the rest to [nbase..)
  uint32_t rbase, rend; // this wave's region of the pair list: slots [rbase, rend)
  uint32_t* ctr;        // the region's counter block
  uint32_t chunk;       // slots reserved per global atomic
};
// Wave-wide exclusive prefix sum of ntot + chunk reservation.  Returns this lane's first index g in the
// wave's appended run; wave_slot(g) maps run indices to pair-list slots.  A run that does not fit in the
// rest of the current chunk fills it up and continues in a freshly reserved chunk (ONE atomic).
// Inclusive prefix sum over the 64 lanes with DPP adds (row_shr 1/2/3, row_shr 4/8 with bank masks, row_bcast 15/31):
// 7 VALU ops instead of 6 ds_bpermute round trips.
__device__ inline uint32_t wave_inclusive_scan(uint32_t x) {
  uint32_t v = x;
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x113, 0xF, 0xF, false);  // row_shr:3  -> sums of 4 inside each bank group
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xE, false);  // row_shr:4, banks 1-3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xC, false);  // row_shr:8, banks 2-3 -> row (16-lane) prefix
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ inline uint32_t wave_reserve(WaveOut& w, uint32_t ntot, uint32_t lane, uint32_t* total_out) {
  const uint32_t incl = wave_inclusive_scan(ntot);
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  *total_out = total;
  w.split = w.left;
  if (total > w.left) {
    const uint32_t rest = total - w.left;
    const uint32_t need = rest > w.chunk ? rest : w.chunk;
    uint32_t b = 0;
    if (lane == 0) b = atomicAdd(&w.ctr[RC_RAW], need);
    w.nbase = w.rbase + __shfl(b, 0);
  }
  return incl - ntot;
}
// the same for a run whose total is known (<= 64 pairs, lane ranks by ballot): sets split / nbase for wave_slot
__device__ inline void wave_reserve_total(WaveOut& w, uint32_t total, uint32_t lane) {
  w.split = w.left;
  if (total > w.left) {
    const uint32_t rest = total - w.left;
    const uint32_t need = rest > w.chunk ? rest : w.chunk;
    uint32_t b = 0;
    if (lane == 0) b = atomicAdd(&w.ctr[RC_RAW], need);
    w.nbase = w.rbase + (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
  }
}
__device__ inline uint32_t wave_slot(const WaveOut& w, uint32_t g) {
  return g < w.split ? w.base + g : w.nbase + (g - w.split);
}
__device__ inline void wave_commit(WaveOut& w, uint32_t total) {
  if (total > w.left) {
    const uint32_t rest = total - w.left;
    const uint32_t need = rest > w.chunk ? rest : w.chunk;
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
  const uint4* scan_rec;    // [E + 1] per ENTRY {planes 1 and 2 of its class, len, class}: ONE 16-byte gather per record (bit-plane kernel)
  const uint2* scan_rec34;  // [E + 1] planes 3 and 4: gathered only by tiles that hold a query with a symbol three or four times
  uint32_t pad_rec;         // = E: a never-matching padding record (planes 0, len 255)
  uint32_t cstride;
  uint32_t pad_class;   // a never-matching padding class (counts 0xFF, len 255; count-vector kernel)
  const uint8_t* cls_len;
  const uint32_t* cls_off;
  const uint4* sig;         // signature table: {groups 0-3, groups 4-7 packed as bytes, first class of the run, classes}
  const uint4* sig_e;       // the same with the run as scan records: {.., .., first entry of the run, entries} (bit-plane kernel)
  const uint4* sighash;     // open-addressing table {sig lo, sig hi, first class of the run, classes}; count 0 = empty
  const uint4* sighash_e;   // the same slots with entry runs (bit-plane kernel)
  uint32_t hash_mask;
  const unsigned long long* ball;  // signature offsets (8 x int8) of the L1 balls, Tile::ball0 / balln index it
  const uint32_t* adj_hdr;  // signature adjacency lists (adjacency.h; tiles with Tile::adj): [list][8] {first row, cumulative rows of the 7 length sections}
  const uint2* adj_planes;  // [row][64] {plane 1, plane 2} of the records
  const uint32_t* adj_ids;  // [row][64] their entry ids (padding: pad_rec)
  const uint32_t* sig_cbeg;
  uint2* raw;
  uint32_t region_cap;  // pair-list slots per region
  uint32_t chunk;       // pair-list slots a wave reserves per global atomic (SCAN_CHUNK; ANX_SCAN_CHUNK)
  uint32_t chunk_fused; // ... in tiles that run the fused prefilter
  uint32_t* rctr;       // [SCAN_REGIONS][RC_STRIDE]
  const uint32_t* qexact;  // per query: class id of its exact anagram class (0xFFFFFFFF = none); stop mode only
  int want_exact;
  int drop_len;             // do not materialise pairs with |len_q - len_c| > d: damerau_levenshtein returns None for them at its
                            // first test (src/distance.rs:109-130); they are only counted as scored pairs
  const uint4* q_rec;       // [Q][2] {first 16 symbols of the query} {meta, ...}   (fused prefilter)
  const uint4* e_rec;       // [E][2] {first 16 symbols of the entry} {meta, row offset, freq, -}
  int fuse;                 // apply the SWAR band-match bound (kernels_swar.hpp) where the pair is born: pairs it rejects are
                            // counted as scored pairs and never written; survivors carry RAW_PREFILTERED
  uint32_t* qpairs;         // per query: scored pairs of THIS run, counted where they are produced (materialised or only counted);
                            // nullptr in normal runs (anx_batch_pair_counts: the per-query check of the production pair list)
  int dbg;  // ANX_SCAN_DBG (timing experiments only; results are wrong when set): 1 one query per pass, 2 skip process(), 4 no run staging,
            // 8 no hit expansion, 16 pairs dropped before the band filter, 32 filtered pairs not written
};

__device__ inline int32_t bcnt_acc(uint32_t x, int32_t acc) {  // acc + popcount(x) in one v_bcnt_u32_b32
  int32_t r;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
  return r;
}
__device__ inline int32_t bcnt_acc_s(uint32_t x, int32_t acc) {  // the same with a wave-uniform addend (an SGPR operand: no VGPR per threshold)
  int32_t r;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "s"(acc));
  return r;
}

// T >= 1: thermometer bit planes.  common(q,c) = sum_t popc(Q_t & C_t) is exact when every symbol of the query
//   occurs at most T times (min(a,b) only needs a's planes; class planes saturate at NBITPLANES).
//   L1 = len_q + len_c - 2 common, so  hit <=> common >= max(1, ceil((len_q + len_c - k) / 2))   (>= 1: the
//   classes share a symbol, src/iterators.rs:177).  2 ops per plane: v_and_b32 + accumulating v_bcnt_u32_b32.
// T == 0: general path (any alphabet size / multiplicity): packed u8 count vectors, NP x v_sad_u8;
//   hit <=> L1 <= k and L1 < len_q + len_c.
// ADJ: the tile's signature has an adjacency list (adjacency.h): the records of its ball are streamed, rows of 64 with one length
//   each, instead of probed / staged / gathered -- no ball walk, no run staging, coalesced 8 + 4 bytes per record, and the
//   thresholds are wave-uniform per row (scalar registers).
template <bool BITS, int NP, bool GEN, bool ADJ = false>
__device__ inline void scan_tile(const ScanArgs& AA, const Tile& t, uint32_t item, uint32_t* __restrict__ stage, uint32_t* __restrict__ hits,
                                 uint32_t* __restrict__ qlds, uint4* __restrict__ qsym, uint32_t* __restrict__ pbuf) {
  // GEN = false: the production instance (no StopAtExactMatch, no per-query pair counts, pairs that fail the DL's length test are
  // only counted): those branches and their arguments stay out of the kernel
  ScanArgs A = AA;
  if (!GEN) { A.want_exact = 0; A.qpairs = nullptr; A.drop_len = 1; A.qexact = nullptr; }
  constexpr int CPL = BITS ? 4 : (NP <= 8 ? 4 : NP <= 16 ? 2 : 1);  // classes per lane
  constexpr int W = BITS ? NBITPLANES : NP;                          // dwords held per class
  constexpr int QSTRIDE = BITS ? NBITPLANES : NP;
  constexpr uint32_t CHUNK = 64u * CPL;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t* __restrict__ cls_words = BITS ? A.cls_bits : A.cls_planes;
  const uint8_t* __restrict__ cls_len = A.cls_len;
  const uint32_t* __restrict__ cls_off = A.cls_off;
  uint2* __restrict__ raw = A.raw;
  const uint32_t cstride = A.cstride;
  const uint32_t region = item % SCAN_REGIONS;
  WaveOut wo{0, 0, 0, 0, 0, region * A.region_cap, (region + 1) * A.region_cap, A.rctr + region * RC_STRIDE, A.chunk};
  uint32_t ns = 0;  // staged class ids (wave-uniform)
  uint32_t nchunks = 0;
  uint32_t arend = 0;               // ADJ: end of the tile's rows (absolute row number, wave-uniform)
  int32_t alc[4] = {0, 0, 0, 0};    // ADJ: record length of each of the chunk's rows (wave-uniform)
  // ADJ: the records of a chunk (4 rows of 64, one record per lane and row) are requested one chunk AHEAD: the stream comes from HBM
  // (~2 us), and a wave that waited for its own loads at the head of every chunk left the SIMD to 4 other waves doing the same
  uint2 apl[4] = {};                // planes 1 / 2 of the lane's record in each row of the chunk being processed
  uint32_t aid[4] = {};             // their entry ids
  auto adj_load = [&](uint32_t row, uint2 (&pl)[4], uint32_t (&id)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      pl[j] = make_uint2(0u, 0u);   // rows beyond the tile's range read as padding (planes 0: no common symbol with anything)
      id[j] = A.pad_rec;
      if (row + (uint32_t)j < arend) {  // wave-uniform
        const size_t p = (size_t)(row + (uint32_t)j) * 64u + lane;
        pl[j] = A.adj_planes[p];
        id[j] = A.adj_ids[p];
      }
    }
  };
  {  // the tile's query words -> LDS: the comparison loop reads them back as broadcasts into VGPRs
    const uint32_t* __restrict__ src = (BITS ? A.q_bits : A.q_cv) + (size_t)t.q0 * QSTRIDE;
    for (uint32_t i = lane; i < t.nq * QSTRIDE; i += 64) qlds[i] = src[i];
  }
  // Fused prefilter (bit-plane tiles of queries <= 16 symbols with d <= 3): the first 16 symbols of the tile's queries sit in LDS,
  // the expansion below makes the pairs DENSE in an LDS buffer and tests 64 of them at a time against the band-match bound.
  const bool fuse = BITS && A.fuse && t.lq <= 16u && t.d <= 3u;  // wave-uniform
  if (fuse) wo.chunk = A.chunk_fused;  // a third of the pairs survive the filter: smaller reservations waste fewer slots
  if (BITS && fuse && lane < t.nq) qsym[lane] = rec32(A.q_rec, t.q0 + lane)[0];
  constexpr uint32_t PBUF = SCAN_PBUF;
  uint32_t nfused = 0;          // pairs the fused filter tested (wave-uniform; statistics)
  uint32_t npb = 0, phead = 0;  // pairs waiting in the ring pbuf (wave-uniform count and read position), each (entry | query-in-tile << 26)

  uint32_t nhits = 0;  // entries in the hit list (wave-uniform)
  uint32_t counted_only = 0;  // per lane: pairs dropped by the length test (they still count as scored pairs)
  // Writes pairs of the dense buffer to the pair list, 64 at a time (fin: also the remainder).  With `fuse` each lane first
  // gathers its entry's symbols and runs the band-match bound against its query's symbols from LDS: 2/3 of the pairs of the
  // bench workload end here (counted as scored pairs, like the ones the DL's length test drops) and never reach HBM.
  auto emit_dense = [&](bool fin) {
    while (npb >= 64u || (fin && npb)) {
      const uint32_t cnt = npb < 64u ? npb : 64u;
      const bool act = lane < cnt;
      const uint32_t pr = pbuf[(phead + lane) & (PBUF - 1u)];
      const uint32_t e = act ? (pr & 0x3FFFFFFu) : 0u, ql = act ? (pr >> 26) : 0u;
      bool keep = act;
      uint32_t flag = 0u;
      if (ANX_DBG(A.dbg) & 16) keep = false;  // timing: the pairs are dropped unfiltered
      if (BITS && fuse && !(ANX_DBG(A.dbg) & 16)) {
        nfused += cnt;
        const uint4 C = rec32(A.e_rec, e)[0];
        const int lc = (int)(rec32(A.e_rec, e)[1].x & 0xFFu), lq = (int)t.lq, d = (int)t.d;
        const uint4 Q = qsym[ql];
        const bool filt = act && lc <= 16;      // |lq - lc| <= d holds: length-incompatible records were only counted
        const int ml = lq > lc ? lq : lc;
        constexpr uint32_t M = 0x7F7F7F7Fu;     // alphabets of the bit-plane kernel have <= 32 symbols: every code < 0x7E (B7 form)
        bool rej;
        if (__any(filt && ml > 12)) {
          const uint32_t q4[4] = {Q.x & M, Q.y & M, Q.z & M, Q.w & M}, c6[6] = {M, C.x & M, C.y & M, C.z & M, C.w & M, M};
          rej = band_bound_rejects<4, true, true>(q4, c6, filt, d, lq, lc);
        } else if (__any(filt && ml > 8)) {
          const uint32_t q3[3] = {Q.x & M, Q.y & M, Q.z & M}, c5[5] = {M, C.x & M, C.y & M, C.z & M, M};
          rej = band_bound_rejects<3, true, true>(q3, c5, filt, d, lq, lc);
        } else {
          const uint32_t q2[2] = {Q.x & M, Q.y & M}, c4[4] = {M, C.x & M, C.y & M, M};
          rej = band_bound_rejects<2, true, true>(q2, c4, filt, d, lq, lc);
        }
        if (rej) { keep = false; ++counted_only; }
        if (filt) flag = RAW_PREFILTERED;
      }
      if (ANX_DBG(A.dbg) & 32) keep = false;  // timing: filtered, nothing written
      const unsigned long long km = __ballot(keep);
      const uint32_t total = (uint32_t)__popcll(km);
      if (total) {  // wave-uniform
        wave_reserve_total(wo, total, lane);
        if (keep) {
          const uint32_t q = t.q0 + ql;
          uint32_t exact = 0u;
          if (A.want_exact) exact = A.qexact[q] == A.scan_rec[e].w ? 0x80000000u : 0u;  // StopAtExactMatch, src/lib.rs:1164-1173
          const uint32_t g = __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
          const uint32_t pos = wave_slot(wo, g);
          if (pos < wo.rend) raw[pos] = make_uint2(q, e | exact | flag);
        }
        wave_commit(wo, total);
      }
      phead = (phead + cnt) & (PBUF - 1u);  // a ring: nothing moves
      npb -= cnt;
    }
  };
  // expands the hit list into (query, entry) pairs, one list entry per lane and round
  auto flush = [&]() {
    for (uint32_t r0 = 0; r0 < nhits; r0 += 64) {
      const uint32_t idx = r0 + lane;
      uint32_t c = 0, m = 0, e0 = 0, ne = 0, qb = 0;   // c: class id (stop mode / SAD path); pairs = queries of m x entries [e0, e0 + ne)
      if (idx < nhits) {
        bool count_only;
        if (BITS) {
          // one scan record per lexicon ENTRY (the planes of its class): a hit is a (query, entry) pair already, there is no
          // entries-per-class loop and no class gather here.  The list holds the record's position in the staged chunk (the
          // stage keeps the chunk's ids until process() returns), bit 8 = pass, bit 9 = "fails the DL's length test"
          const uint32_t hx = reinterpret_cast<const uint16_t*>(hits + SCAN_HITS)[idx];
          m = hits[idx];
          qb = ((hx >> 8) & 1u) << 5;
          count_only = (hx >> 9) & 1u;
          e0 = stage[hx & 0xFFu];
          ne = 1;
          c = A.qpairs ? A.scan_rec[e0].w : 0u;  // class of the entry (pair-count runs only)
        } else {
          c = hits[2 * idx];
          m = hits[2 * idx + 1];
          qb = ((c >> 27) & 1u) << 5;
          c &= (1u << 27) - 1u;
          const uint32_t lc = cls_len[c];
          e0 = cls_off[c];
          ne = cls_off[c + 1] - e0;
          const uint32_t diff = lc > t.lq ? lc - t.lq : t.lq - lc;
          count_only = A.drop_len && diff > t.d;
        }
        if (A.qpairs) {  // per-query pair counts of this very run (StopAtExactMatch: only the exact class counts when there is one)
          for (uint32_t mm = m; mm; mm &= mm - 1u) {
            const uint32_t q = t.q0 + qb + (uint32_t)__ffs((int)mm) - 1u;
            if (!A.want_exact || A.qexact[q] == 0xFFFFFFFFu || A.qexact[q] == c) atomicAdd(&A.qpairs[q], ne);
          }
        }
        if (count_only) {  // every pair of this class fails the length test of the DL: count, do not emit
          counted_only += (uint32_t)__popc(m) * ne;
          m = 0;
        }
      }
      if (BITS) {
        // bit-plane tiles: one pair per lane and trip into the dense LDS buffer (ballot ranks); whenever 64 pairs wait they are
        // filtered and written -- the trips run as long as the fullest mask of the round, the filter rounds are always dense
        while (true) {
          const bool has = m != 0u;
          const unsigned long long bm = __ballot(has);
          if (!bm) break;  // wave-uniform
          if (has) {
            const uint32_t bit = (uint32_t)__ffs((int)m) - 1u;
            m &= m - 1u;
            pbuf[(phead + npb + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u))) & (PBUF - 1u)] = e0 | ((qb + bit) << 26);
          }
          npb += (uint32_t)__popcll(bm);
          if (npb >= 64u) emit_dense(false);
        }
        continue;
      }
      uint32_t total;
      uint32_t g = wave_reserve(wo, (uint32_t)__popc(m) * ne, lane, &total);
      if (total <= wo.split && wo.base + total <= wo.rend) {
        // wave-uniform common case: the run fits the rest of the current chunk, so slot = base + run index, no
        // spill select and no bounds test per pair
        uint2* __restrict__ dst = raw + wo.base;
        while (m) {
          const uint32_t bit = (uint32_t)__ffs((int)m) - 1u;
          m &= m - 1u;
          const uint32_t q = t.q0 + qb + bit;
          const uint32_t exact = (A.want_exact && A.qexact[q] == c) ? 0x80000000u : 0u;
          for (uint32_t i = 0; i < ne; ++i, ++g) dst[g] = make_uint2(q, (e0 + i) | exact);
        }
      } else {
        while (m) {
          const uint32_t bit = (uint32_t)__ffs((int)m) - 1u;
          m &= m - 1u;
          const uint32_t q = t.q0 + qb + bit;
          // the exact anagram class (StopAtExactMatch, src/lib.rs:1164-1173)
          const uint32_t exact = (A.want_exact && A.qexact[q] == c) ? 0x80000000u : 0u;
          for (uint32_t i = 0; i < ne; ++i, ++g) {
            const uint32_t pos = wave_slot(wo, g);
            if (pos < wo.rend) raw[pos] = make_uint2(q, (e0 + i) | exact);
          }
        }
      }
      wave_commit(wo, total);
    }
    nhits = 0;
  };

  // compares the first CHUNK staged classes (padded with the never-matching class) with every query of the tile, 32
  // queries per pass; bit (npass-1-qi) of hm[j] = query qi of the pass hits class j of this lane.  The query loop is
  // branch-free: 2 ops per plane + ONE v_alignbit_b32 per test (it shifts the sign bit of acc = "miss" into the mask).
  auto process = [&]() {
    ++nchunks;
    if (ANX_DBG(A.dbg) & 2) return;
    uint32_t cid[CPL], cw[CPL][W];
    int32_t thr[CPL];   // ADJ: wave-uniform (scalar registers)
    const bool need34 = BITS && ((t.kend >> 8) & 0xFFu) < t.nq;  // some query of the tile is of kind 3 or 4
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const uint32_t idx = (uint32_t)j * 64u + lane;
      int32_t lc;
      if (ADJ) {
        // row j of the chunk: 64 consecutive records of the list, one per lane (adj_load: coalesced 8-byte and 4-byte loads)
        const uint2 pl = apl[j];
        const uint32_t id = aid[j];
        stage[idx] = id;  // the hit list holds positions in the chunk (flush)
        uint2 hi = make_uint2(0u, 0u);
        if (need34) hi = A.scan_rec34[id];
        cw[j][0] = pl.x; cw[j][1] = pl.y;
        if (W > 2) { cw[j][2] = hi.x; cw[j][W - 1] = hi.y; }
        lc = alc[j];
        const int32_t diff = lc > (int32_t)t.lq ? lc - (int32_t)t.lq : (int32_t)t.lq - lc;
        cid[j] = (A.drop_len && diff > (int32_t)t.d) ? id | (1u << 28) : id;
      } else {
      cid[j] = idx < ns ? stage[idx] : (BITS ? A.pad_rec : A.pad_class);
      if (BITS) {  // one 32-B scan record per entry {4 planes of its class} {len, class, -, -} instead of T + 3 gathers
        // 16 B per record {plane 1, plane 2, len, class}: the chunk set-up is bound by the texture addresser (8 -> 4 gathers per lane
        // and chunk); planes 3 / 4 only matter to queries with a symbol three / four times (5 % of the tests), so only tiles that
        // hold such a query fetch them (wave-uniform: the tile's kind boundaries)
        const uint4 mt = A.scan_rec[cid[j]];
        uint2 hi = make_uint2(0u, 0u);
        if (need34) hi = A.scan_rec34[cid[j]];
        const uint32_t plw[4] = {mt.x, mt.y, hi.x, hi.y};
#pragma unroll
        for (int p = 0; p < W; ++p) cw[j][p] = plw[p];
        lc = (int32_t)mt.z;
        const int32_t diff = lc > (int32_t)t.lq ? lc - (int32_t)t.lq : (int32_t)t.lq - lc;
        if (A.drop_len && diff > (int32_t)t.d) cid[j] |= 1u << 28;  // its pairs fail the DL's length test: counted, not emitted
      } else {
#pragma unroll
        for (int p = 0; p < W; ++p) cw[j][p] = cls_words[(size_t)p * cstride + cid[j]];
        lc = (int32_t)cls_len[cid[j]];
      }
      }
      if (BITS) {
        const int32_t need = ((int32_t)t.lq - (int32_t)t.k + lc + 1) >> 1;  // ceil((lq + lc - k) / 2)
        thr[j] = -(need < 1 ? 1 : need);
      } else {
        const int32_t share = (int32_t)t.lq + lc - 1;  // L1 < lq + lc: shares a symbol (src/iterators.rs:177, src/lib.rs:1205)
        thr[j] = share < (int32_t)t.k ? share : (int32_t)t.k;
      }
    }
    for (uint32_t qb = 0; qb < t.nq; qb += 32) {
      const uint32_t npass = (ANX_DBG(A.dbg) & 1) ? 1u : (t.nq - qb < 32u ? t.nq - qb : 32u);
      uint32_t hm[CPL];
#pragma unroll
      for (int j = 0; j < CPL; ++j) hm[j] = 0xFFFFFFFFu;  // miss bits
      // one query against the lane's CPL classes: shifts one bit into every hm[j].  TW = planes compared: the query's
      // kind (bit-plane tiles hold queries of every kind 1..NBITPLANES, sorted by kind) or NP count-vector dwords.
      auto test_query = [&](auto tw, uint32_t qi) {
        constexpr int TW = decltype(tw)::value;
        uint32_t qreg[TW];
#pragma unroll
        for (int p = 0; p < TW; ++p) qreg[p] = qlds[(qb + qi) * QSTRIDE + p];
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          int32_t acc;
          if (BITS) {
            // common - threshold: negative = miss
            acc = ADJ ? bcnt_acc_s(qreg[0] & cw[j][0], thr[j]) : bcnt_acc(qreg[0] & cw[j][0], thr[j]);
#pragma unroll
            for (int p = 1; p < TW; ++p) acc = bcnt_acc(qreg[p] & cw[j][p], acc);
          } else {
            uint32_t sad = 0;
#pragma unroll
            for (int p = 0; p < TW; ++p) sad = __builtin_amdgcn_sad_u8(qreg[p], cw[j][p], sad);
            acc = thr[j] - (int32_t)sad;  // threshold - L1: negative = miss
          }
          hm[j] = __builtin_amdgcn_alignbit(hm[j], (uint32_t)acc, 31);  // (hm << 1) | sign(acc)
        }
      };
      auto run_queries = [&](auto tw, uint32_t lo, uint32_t hi) {  // queries [lo, hi) of the pass, in order
        uint32_t qi = lo;
        for (; qi + 2 <= hi; qi += 2) {  // two queries per trip: half the loop overhead, both LDS reads in flight
          test_query(tw, qi);
          test_query(tw, qi + 1);
        }
        if (qi < hi) test_query(tw, qi);
      };
      if (BITS) {
        // the tile's queries are sorted by kind; ke[t] = end of the kind-t queries.  The four sub-ranges of the pass run
        // in order, so bit positions keep following the query order.
        const uint32_t pend = qb + npass;
        auto clampr = [&](uint32_t x) { return (x < qb ? qb : (x > pend ? pend : x)) - qb; };
        const uint32_t b1 = clampr(t.kend & 0xFFu), b2 = clampr((t.kend >> 8) & 0xFFu), b3 = clampr((t.kend >> 16) & 0xFFu);
        run_queries(std::integral_constant<int, 1>{}, 0u, b1);
        run_queries(std::integral_constant<int, 2>{}, b1, b2);
        run_queries(std::integral_constant<int, 3>{}, b2, b3);
        run_queries(std::integral_constant<int, BITS ? 4 : 1>{}, b3, npass);
      } else {
        run_queries(std::integral_constant<int, BITS ? 1 : NP>{}, 0u, npass);
      }
      // expand the hits of this pass into (query, entry) pairs
      const uint32_t valid = npass >= 32u ? 0xFFFFFFFFu : ((1u << npass) - 1u);
      uint32_t any = 0;
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        hm[j] = ~hm[j] & valid;
        any |= hm[j];
      }
      if (__ballot(any != 0) == 0ull || (ANX_DBG(A.dbg) & 8)) continue;  // wave-uniform (dbg 8: timing without the expansion)
      // The non-empty (class, hit mask) pairs are appended to the wave's LDS hit list; flush() expands the list one
      // entry per lane whenever it gets full (and at the end of the tile), so the expansion rounds run with full
      // waves and the loop runs max-over-entries popcount times.  Mask bit b = query qb + b of the tile.
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const bool nz = hm[j] != 0;
        const unsigned long long bm = __ballot(nz);
        const uint32_t cnt = (uint32_t)__popcll(bm);
        if (cnt) {  // wave-uniform
          if (nz) {
            const uint32_t pos = nhits + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
            const uint32_t hmask = __brev(hm[j]) >> (32u - npass);  // shift-in order -> bit b = query b of the pass
            if (BITS) {
              hits[pos] = hmask;
              reinterpret_cast<uint16_t*>(hits + SCAN_HITS)[pos] = (uint16_t)(((uint32_t)j * 64u + lane) | ((qb >> 5) << 8) | (((cid[j] >> 28) & 1u) << 9));
            } else {
              hits[2 * pos] = cid[j] | ((qb >> 5) << 27);  // class id, pass
              hits[2 * pos + 1] = hmask;
            }
          }
          nhits += cnt;
        }
      }
      if (!BITS && nhits > SCAN_HITS - CHUNK) flush();  // a pass adds at most CHUNK entries
    }
    // bit-plane tiles expand after the passes, when the records' planes and thresholds are dead (the fused filter needs the
    // registers): the two passes of a chunk add at most 2 * CHUNK = SCAN_HITS entries
    if (BITS) flush();
  };

  // The tile's signature window [s0, s1) is aligned to whole 64-signature blocks (no bounds test is needed: the other
  // signatures of the first / last block belong to other charcounts, so their L1 distance to the tile's signature is at
  // least the length difference > k, and the table is padded with never-matching entries).
  // One 16-byte record per signature: the run (first class, count) comes with the signature, so a matching step does not
  // wait for a second, dependent load.
  // Stages the class / record runs (cb, n) of the lanes with ok -- all runs of a probe / walk step at once, 64 ids per trip.
  // The ids of the step form a stream (runs back to back in lane order) that continues the stage at position ns; the stream is
  // cut into blocks of 64 positions aligned with the stage, so every trip fills whole 64-id rows (only the first and the last
  // block of a step are partial).  Lane -> run: a bit per run END in an LDS mask (ds_or, one instruction for all runs), and
  // r = runs that ended before my position = mbcnt of the block's 64-bit mask; id = sdelta[r] + position with
  // sdelta[r] = first id of run r - its first position.  ~14 wave instructions per 64 ids whatever the run lengths; the scalar
  // loop this replaces took ~15 per RUN (16 ids on average: ~260 runs per tile on BASELINE configs[1], 0.28 of the kernel's
  // 1.63 ms and most of its 0.55 SALU instructions per VALU instruction).
  uint32_t* __restrict__ sdelta = stage + CHUNK;                                  // [64] per run of the step, in lane order
  uint32_t* __restrict__ smask = stage + CHUNK + 64;  // [SCAN_MASKW / 32] run ends of the window (one type for the store, the atomic and the load)
  auto stage_runs = [&](bool ok, uint32_t cb, uint32_t n) {
    ok = ok && n != 0u;  // an empty run has no end of its own
    const unsigned long long okm = __ballot(ok);
    if (!okm || (ANX_DBG(A.dbg) & 4)) return;
    const uint32_t cnt = ok ? n : 0u;
    const uint32_t incl = wave_inclusive_scan(cnt);
    const uint32_t carry = ns & 63u;                 // ids of the stage's last, partial row
    const uint32_t total = carry + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);  // stream positions [carry, total)
    uint32_t wbase = ns - carry;                     // stage row the next block goes to (multiple of 64, < CHUNK)
    const uint32_t endp = carry + incl - 1u;         // stream position of the run's last id (ok lanes, n >= 1)
    if (ok) sdelta[__builtin_amdgcn_mbcnt_hi((uint32_t)(okm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)okm, 0u))] = cb + cnt - 1u - endp;
    uint32_t rbase = 0;                              // runs that ended before the current block
    for (uint32_t w0 = 0; w0 < total; w0 += SCAN_MASKW) {  // windows of SCAN_MASKW stream positions (one, as a rule)
      if (lane < SCAN_MASKW / 32u) smask[lane] = 0u;
      if (ok && endp - w0 < SCAN_MASKW) atomicOr(smask + ((endp - w0) >> 5), 1u << (endp & 31u));
      const uint32_t wend = total < w0 + SCAN_MASKW ? total : w0 + SCAN_MASKW;
      for (uint32_t p0 = w0; p0 < wend; p0 += 64) {
        // wave-uniform addresses: broadcasts
        const uint32_t mlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)smask[((p0 - w0) >> 6) * 2u]), mhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)smask[((p0 - w0) >> 6) * 2u + 1u]);
        const uint32_t r = rbase + __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
        rbase += (uint32_t)__popc(mlo) + (uint32_t)__popc(mhi);
        const uint32_t p = p0 + lane;
        const uint32_t id = sdelta[r & 63u] + p;
        if (p >= carry && p < total) stage[wbase + lane] = id;
        if (p0 + 64u <= total) {  // a whole row
          wbase += 64u;
          ns = wbase;
          if (wbase == CHUNK) {
            process();
            ns = 0;
            wbase = 0;
          }
        } else {
          ns = wbase + (total - p0);
        }
      }
    }
  };
  const uint4* __restrict__ sigtab = BITS ? A.sig_e : A.sig;  // runs of scan records (entries) / of classes
  if (ADJ) {
    // the list of the tile's signature: rows [row0 + rbeg, row0 + rend) hold the records of the lengths lq - k .. lq + k (section i of the
    // list = length lq - 3 + i; a list holds the ball of radius 3, a tile with a smaller k tests a few records more than it has to)
    const cptr_u32 hp = (cptr_u32)(A.adj_hdr + (size_t)(t.adj - 1u) * 8u);
    const uint32_t row0 = hp[0], c0 = hp[1], c1 = hp[2], c2 = hp[3], c3 = hp[4], c4 = hp[5], c5 = hp[6], c6 = hp[7];
    uint32_t rbeg = t.k >= 3u ? 0u : t.k == 2u ? c0 : t.k == 1u ? c1 : c2;
    uint32_t rend = t.k >= 3u ? c6 : t.k == 2u ? c5 : t.k == 1u ? c4 : c3;
    if (t.flags & 2u) { rbeg = t.s0; rend = t.s1; }  // the small call splits a query's list over several waves (k_small_tiles): this tile's share of the rows
    arend = row0 + rend;
    if (lane == 0 && rend > rbeg) {  // statistics: rows streamed (bench.py: access bytes / bytes that have to be read at least once)
      atomicAdd(&wo.ctr[RC_ADJ], rend - rbeg);
      if (t.flags & 1u) atomicAdd(&wo.ctr[RC_ADJ_FIRST], rend - rbeg);
    }
    uint2 npl[4];
    uint32_t nid[4];
    if (rbeg < rend) adj_load(row0 + rbeg, npl, nid);
    for (uint32_t r = rbeg; r < rend; r += (uint32_t)CPL) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t rr = r + (uint32_t)j;
        alc[j] = (int32_t)t.lq - 3 + (int32_t)((rr >= c0) + (rr >= c1) + (rr >= c2) + (rr >= c3) + (rr >= c4) + (rr >= c5));
        apl[j] = npl[j];
        aid[j] = nid[j];
      }
      if (r + (uint32_t)CPL < rend) adj_load(row0 + r + (uint32_t)CPL, npl, nid);  // the next chunk: in flight under this one's tests
      process();
    }
  } else if (t.balln) {
    // Signatures within L1 distance k of the tile's: enumerated, not searched.  Lane i adds offset i of the ball (sum |d_g| <= k)
    // to the tile's signature byte-wise and looks the result up in the hash table of the lexicon's signatures -- 377 probes for
    // 6 groups and k = 3 whatever the size of the lexicon, against a walk over every signature of the +-k charcount window.
    const unsigned long long H = 0x8080808080808080ull;
    const unsigned long long sq = (unsigned long long)t.sig_lo | (unsigned long long)t.sig_hi << 32;  // every byte <= SIG_BYTE_MAX
    const uint4* __restrict__ htab = BITS ? A.sighash_e : A.sighash;
    for (uint32_t base = 0; base < t.balln; base += 64) {
      const uint32_t i = base + lane;
      bool found = false;
      uint32_t first = 0, count = 0;
      if (i < t.balln) {
        const unsigned long long dl = A.ball[t.ball0 + i];
        const unsigned long long r = ((sq & ~H) + (dl & ~H)) ^ ((sq ^ dl) & H);  // byte-wise sum, no carry between bytes
        if (!(r & (r << 1) & H)) {  // a byte >= 0xC0 is a negative group sum: no such signature
          const uint32_t lo = (uint32_t)r, hi = (uint32_t)(r >> 32);
          uint32_t h = sig_hash(lo, hi) & A.hash_mask;
          for (int p = 0; p < 17; ++p) {  // the table is built with every key within 16 slots of its home
            const uint4 e = htab[h];
            if (!e.w) break;
            if (e.x == lo && e.y == hi) { found = true; first = e.z; count = e.w; break; }
            h = (h + 1u) & A.hash_mask;
          }
        }
      }
      stage_runs(found, first, count);
    }
  } else {
    // The flat walk: every signature of the +-k charcount window [s0, s1) (aligned to 64-signature blocks: the other
    // signatures of the edge blocks belong to other charcounts, so their L1 distance to the tile's signature is at least
    // the length difference > k; the table is padded with never-matching entries), 64 per step, one 16-byte record per
    // signature loaded one step ahead.  A two-level walk over block bounding boxes was measured and removed in round 2: an
    // L1 ball of radius k in 6-8 small-integer dimensions touches 78-87 % of the blocks (tools/kdsim.py).
    const uint4* __restrict__ sigp = sigtab + t.s0 + lane;
    uint4 sg_next = t.s0 < t.s1 ? *sigp : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);
    for (uint32_t sb = t.s0; sb < t.s1; sb += 64) {
      const uint4 sg = sg_next;  // loaded one step ahead: the step's test does not wait for its own load
      sigp += 64;
      if (sb + 64 < t.s1) sg_next = *sigp;
      const bool ok = __builtin_amdgcn_sad_u8(sg.x, t.sig_lo, __builtin_amdgcn_sad_u8(sg.y, t.sig_hi, 0u)) <= t.k;
      stage_runs(ok, sg.z, sg.w);
    }
  }
  if (!ADJ && ns) process();
  flush();
  if (BITS) emit_dense(true);
  if (A.drop_len) {  // wave sum of the counted-only pairs -> RC_VALID (n_pairs = every DL invocation of the reference)
    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(counted_only), 63);
    wo.emitted += tot;
  }
  wave_close(wo, lane, raw);
  if (BITS && lane == 0 && nfused) atomicAdd(&wo.ctr[RC_FUSED], nfused);
  if (lane == 0 && nchunks) {  // class tests by planes compared (statistics)
    if (BITS) {
      const uint32_t e[5] = {0u, t.kend & 0xFFu, (t.kend >> 8) & 0xFFu, (t.kend >> 16) & 0xFFu, t.nq};
      for (int k = 1; k <= NBITPLANES; ++k)
        if (e[k] > e[k - 1])
          atomicAdd(reinterpret_cast<unsigned long long*>(wo.ctr + RC_TESTS + 2 * k), (unsigned long long)nchunks * CHUNK * (e[k] - e[k - 1]));
    } else {
      atomicAdd(reinterpret_cast<unsigned long long*>(wo.ctr + RC_TESTS), (unsigned long long)nchunks * CHUNK * t.nq);
    }
  }
}

// Every wave takes one tile; tiles are ordered by decreasing cost.  The bit-plane tiles (wave-uniform switch over
// T) and the count-vector tiles run as two launches so that the rarely used wide SAD body does not set the register
// budget (= occupancy) of the common one.
constexpr uint32_t SCAN_STAGE = 64 * 4 + 64 + SCAN_MASKW / 32;  // the chunk's ids, the step's run deltas, the window's run-end masks
template <int NP, bool BITS, bool GEN, bool ADJ = false>
__device__ inline void scan_wave(const ScanArgs& A) {
  constexpr int QWORDS = SCAN_TQ * (BITS ? NBITPLANES : NP);
  __shared__ uint32_t s_qlds[4][QWORDS];
  __shared__ __attribute__((aligned(16))) uint32_t s_stage[4][ADJ ? 64 * 4 : SCAN_STAGE];  // (tiles that stream a list only keep the chunk's ids)
  // per wave, entries awaiting expansion: bit-plane tiles hit mask u32[SCAN_HITS] + (position in the chunk | pass | flag) u16[SCAN_HITS];
  // count-vector tiles (class | pass << 27, hit mask) pairs
  __shared__ uint32_t s_hits[4][BITS ? SCAN_HITS + SCAN_HITS / 2 : 2 * SCAN_HITS];
  __shared__ uint4 s_qsym[BITS ? 4 : 1][BITS ? SCAN_TQ : 1];     // first 16 symbols of the tile's queries (fused prefilter)
  __shared__ uint32_t s_pbuf[BITS ? 4 : 1][BITS ? SCAN_PBUF : 1];      // dense (entry | query << 26) pairs awaiting the filter / the write
  const uint32_t wid = threadIdx.x >> 6;
  // (round 5, measured and dropped: persistent waves that stride over the tiles -- 256 x 6 blocks, each wave taking every 6144th tile
  // of the cost-sorted list -- 1.12 -> 1.42 ms: the hardware's dispatch of one wave per tile in cost order balances the waves, a
  // static stride does not)
  const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + wid));
  if (item >= A.ntiles) return;
  const cptr_u32 tp = (cptr_u32)(A.tiles + item);
  Tile t;
  t.q0 = tp[0]; t.nq = tp[1]; t.s0 = tp[2]; t.s1 = tp[3]; t.k = tp[4]; t.lq = tp[5]; t.sig_lo = tp[6]; t.sig_hi = tp[7]; t.kind = tp[8]; t.d = tp[9]; t.kend = tp[10]; t.ball0 = tp[11]; t.balln = tp[12]; t.adj = tp[13]; t.flags = tp[14];
  // the launches are split by the encoders' tile order: [tiles that stream an adjacency list | other bit-plane tiles | count-vector tiles]
  scan_tile<BITS, NP, GEN, ADJ>(A, t, item, s_stage[wid], s_hits[wid], s_qlds[wid], s_qsym[BITS ? wid : 0], s_pbuf[BITS ? wid : 0]);
}
// <= 80 VGPRs = 6 waves per SIMD for the bit-plane kernel (measured: unconstrained 85 VGPRs -> 2.33 ms, 80 -> 2.20 ms,
// 64 with spills -> 2.60 ms)
// one instance for every alphabet (the bit-plane body does not depend on the count-vector width); GEN: see scan_tile
template <bool GEN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_scan_bits(ScanArgs A) { scan_wave<8, true, GEN>(A); }
// the tiles whose signature has an adjacency list (adjacency.h): a kernel of their own, so that neither path carries the other's registers
template <bool GEN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_scan_adj(ScanArgs A) { scan_wave<8, true, GEN, true>(A); }
template <int NP>
__global__ __launch_bounds__(256) void k_scan_sad(ScanArgs A) { scan_wave<NP, false, true>(A); }


// The small call's scan: ONE launch over all tile slots of the call (engine.hip small_find: a tile per query and part, unused slots
// have nq = 0), each wave taking the body its tile needs.  The three bodies share one kernel here -- occupancy does not matter for a
// few thousand waves, the launches do (a small call is launch-bound).  Tiles hold ONE query.
template <int NP>
__global__ __launch_bounds__(256) void k_scan_small(ScanArgs A) {
  constexpr int QW = NP > NBITPLANES ? NP : NBITPLANES;
  __shared__ uint32_t s_qlds[4][QW];
  __shared__ __attribute__((aligned(16))) uint32_t s_stage[4][SCAN_STAGE];
  __shared__ uint32_t s_hits[4][2 * SCAN_HITS];
  __shared__ uint4 s_qsym[4][1];
  __shared__ uint32_t s_pbuf[4][SCAN_PBUF];
  const uint32_t wid = threadIdx.x >> 6;
  const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + wid));
  if (item >= A.ntiles) return;
  const cptr_u32 tp = (cptr_u32)(A.tiles + item);
  Tile t;
  t.q0 = tp[0]; t.nq = tp[1]; t.s0 = tp[2]; t.s1 = tp[3]; t.k = tp[4]; t.lq = tp[5]; t.sig_lo = tp[6]; t.sig_hi = tp[7]; t.kind = tp[8]; t.d = tp[9]; t.kend = tp[10]; t.ball0 = tp[11]; t.balln = tp[12]; t.adj = tp[13]; t.flags = tp[14];
  if (t.nq == 0u) return;  // an unused slot
  if (t.kind == 0u) scan_tile<false, NP, true>(A, t, item, s_stage[wid], s_hits[wid], s_qlds[wid], s_qsym[wid], s_pbuf[wid]);
  else if (t.adj) scan_tile<true, 8, false, true>(A, t, item, s_stage[wid], s_hits[wid], s_qlds[wid], s_qsym[wid], s_pbuf[wid]);
  else scan_tile<true, 8, false>(A, t, item, s_stage[wid], s_hits[wid], s_qlds[wid], s_qsym[wid], s_pbuf[wid]);
}
