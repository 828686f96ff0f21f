// kernels_swar.hpp -- the register SWAR band-match bound (K2) shared by the scan's fused expansion and the scoring kernels
// Part of the single translation unit engine.hip (included inside namespace anx); gfx950 only.
#pragma once

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

// B7: every byte of q and c is below 0x80 (alphabets of <= 124 classes, rows masked with 0x7F7F7F7F by the caller: the paddings
// become 0x7E / 0x7F and still equal nothing).  Then x = q ^ c has no bit 7 and x + 0x7F7F7F7F sets bit 7 of exactly the
// non-zero bytes without a carry between bytes: one v_add instead of v_and + v_add + v_or3 (13 of the 24 issue cycles a
// word-shift costs are the zero test; only bit 7 of every byte of nz / nmA / nmB is ever looked at).
// UD: the edit-distance bound d is wave-uniform (the scan's fused filter: one tile, one d): no per-lane disable mask.
template <int DELTA, int NW, bool B7, bool UD = false>
__device__ inline void filter_shift(const uint32_t (&q)[NW], const uint32_t (&c)[NW + 2], uint32_t off, uint32_t (&nmA)[NW],
                                    uint32_t (&nmB)[NW]) {
  // off: 0 for lanes that use this shift (d >= |DELTA|), all ones for the others (nothing matches at this shift)
  uint32_t nz[NW + 2];
  nz[0] = 0xFFFFFFFFu;
  nz[NW + 1] = 0xFFFFFFFFu;
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    uint32_t cs;  // bytes C[4k + DELTA ..]
    if (DELTA == 0) cs = c[k + 1];
    else if (DELTA > 0) cs = __builtin_amdgcn_alignbyte(c[k + 2], c[k + 1], DELTA);
    else cs = __builtin_amdgcn_alignbyte(c[k + 1], c[k], 4 + DELTA);
    const uint32_t x = q[k] ^ cs;
    if (UD) nz[k + 1] = B7 ? (x + 0x7F7F7F7Fu) : (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x);
    else nz[k + 1] = B7 ? ((x + 0x7F7F7F7Fu) | off) : (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | off);  // bit7 set where q[i] != c[i + DELTA]
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
// band-match bound: a symbol with no equal symbol of the other string within +-d positions costs at least one edit.
// Rows are padded beyond their length with bytes that equal nothing (query 0xFE, candidate 0xFF), so the 4*NW - len
// padding positions always count as unmatched and are subtracted instead of masked.
template <int NW, bool B7 = false, bool UD = false>
__device__ inline bool band_bound_rejects(const uint32_t (&q)[NW], const uint32_t (&c)[NW + 2], bool filt, int d, int lq, int lc) {
  uint32_t nmA[NW], nmB[NW];
#pragma unroll
  for (int k = 0; k < NW; ++k) { nmA[k] = 0xFFFFFFFFu; nmB[k] = 0xFFFFFFFFu; }
  filter_shift<0, NW, B7, true>(q, c, 0u, nmA, nmB);  // (every lane uses shift 0: no disable mask)
  if (UD) {  // d is wave-uniform: scalar branches, no disable masks
    if (d >= 1) { filter_shift<1, NW, B7, true>(q, c, 0u, nmA, nmB); filter_shift<-1, NW, B7, true>(q, c, 0u, nmA, nmB); }
    if (d >= 2) { filter_shift<2, NW, B7, true>(q, c, 0u, nmA, nmB); filter_shift<-2, NW, B7, true>(q, c, 0u, nmA, nmB); }
    if (d >= 3) { filter_shift<3, NW, B7, true>(q, c, 0u, nmA, nmB); filter_shift<-3, NW, B7, true>(q, c, 0u, nmA, nmB); }
  } else {
    if (__any(filt && d >= 1)) { const uint32_t off = d >= 1 ? 0u : 0xFFFFFFFFu; filter_shift<1, NW, B7>(q, c, off, nmA, nmB); filter_shift<-1, NW, B7>(q, c, off, nmA, nmB); }
    if (__any(filt && d >= 2)) { const uint32_t off = d >= 2 ? 0u : 0xFFFFFFFFu; filter_shift<2, NW, B7>(q, c, off, nmA, nmB); filter_shift<-2, NW, B7>(q, c, off, nmA, nmB); }
    if (__any(filt && d >= 3)) { const uint32_t off = d >= 3 ? 0u : 0xFFFFFFFFu; filter_shift<3, NW, B7>(q, c, off, nmA, nmB); filter_shift<-3, NW, B7>(q, c, off, nmA, nmB); }
  }
  int unA = lq - 4 * NW, unB = lc - 4 * NW;
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    unA += __popc(nmA[k] & 0x80808080u);
    unB += __popc(nmB[k] & 0x80808080u);
  }
  return filt && (unA > d || unB > d);
}

// 32-byte record i of a record array (i < 2^27): a 32-bit byte offset lets the load use the SGPR base + VGPR offset form
__device__ inline const uint4* rec32(const uint4* base, uint32_t i) {
  return reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + (i << 5));
}

// ------------------------------------------------------------------------------------------------
// Round 6: mismatch masks from SYMBOL PLANES (kernels_common.hpp symbol_planes16; alphabets of <= 61 classes): M_k bit i = (s[i] != t[i + k])
// = OR over the six planes of S_b ^ (T_b >> k); k_filter_score's DL and tail run on them (kernels_score.hpp PlaneMasks).
// (Measured and dropped: the scan's band-match bound on the same masks -- positions of q without a partner = AND_k M_k, of c the masks
// moved back by k, two popcounts -- with the candidate's planes and length as ONE 16-byte gather: identical verdicts, but k_scan_adj
// 0.916 -> 0.95 ms: seven diagonals x 13 instructions + the unpacking are no fewer issue cycles than the byte-wise bound over the 2-4
// words a tile's strings need.)
// ------------------------------------------------------------------------------------------------
struct PlaneRows {
  uint32_t S[6], T[6];
  uint32_t inv;            // bits lq .. 31: positions beyond the query match nothing
  // qp = q_rec[q][1], cp = e_planes[e]: {meta, planes 0|1, 2|3, 4|5}; lq = 0 for a lane without a pair
  __device__ __forceinline__ void load(const uint4& qp, const uint4& cp, int lq) {
    S[0] = qp.y & 0xFFFFu; S[1] = qp.y >> 16; S[2] = qp.z & 0xFFFFu; S[3] = qp.z >> 16; S[4] = qp.w & 0xFFFFu; S[5] = qp.w >> 16;
    T[0] = cp.y & 0xFFFFu; T[1] = cp.y >> 16; T[2] = cp.z & 0xFFFFu; T[3] = cp.z >> 16; T[4] = cp.w & 0xFFFFu; T[5] = cp.w >> 16;
    inv = 0xFFFFFFFFu << lq;   // (lq <= 16)
  }
  template <int K>
  __device__ __forceinline__ uint32_t far() const {   // mismatch mask of diagonal K, |K| <= 15
    uint32_t acc = inv;
#pragma unroll
    for (int b = 0; b < 6; ++b) acc |= S[b] ^ (K >= 0 ? T[b] >> (K >= 0 ? K : 0) : T[b] << (K < 0 ? -K : 0));
    return acc;
  }
};
