// lattice.hip -- search mode's lattice decoding ON THE DEVICE (gfx950 / CDNA4): k best paths + bigram-LM rerank.
//
// Replaces, for all stretches of a find_all_matches batch at once, most_likely_sequence
// (/root/reference/src/lib.rs:2088-2495: lattice with one state per boundary, rustfst shortest_path(nshortest = max_seq), rerank
// :2318-2425) and lm_score_tokens (:2580-2674) -- the host decoder in search.cpp is the same algorithm and stays as the A/B
// reference (ANX_LATTICE=host) and for what the device leaves to it (context rules; lattices beyond the limits below).
// One WAVE decodes one stretch:
//   k-best  : states in topological order; the K best paths into a state are the K smallest of {best[src][r] + arc} under
//             (cost, arc, r): every lane holds the head of one incoming arc's (sorted) candidate list, a wave-wide minimum over
//             (f32 cost bits << 32 | arc) pops the next path -- the K-way merge search.cpp runs with a binary heap, same order.
//             A virtual end state behind the final states merges their lists (the host's "ends").
//   LM      : the final paths mark the lattice nodes they run through; states in order, the marked nodes of a state side by side:
//             a node's (f32 log-probability sum, token count, last token) = its parent's, extended by the tokens of its symbol --
//             the same additions in the same order as the host's per-path sum.  Bigram terms come from a device hash table whose
//             VALUES are the host's logf results (no device logf: bit-identical terms).
//   select  : perplexity / cost normalisation with portable_log (same bits as the host), weighted mean, first maximum wins;
//             the winner's symbols are walked back and written out.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <ctime>
#include <functional>
#include <mutex>
#include <type_traits>
#include <string>
#include <vector>

#include <memory>
#include <rocprim/device/device_scan.hpp>

#include "engine_internal.h"
#include "portable_log.hpp"

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

struct DeviceLm {  // per replica: the bigram terms and the token lists of the vocabulary
  int device = 0;
  unsigned long long* bg_key = nullptr;  // open addressing: (a << 32 | b) + 1, 0 = empty
  float* bg_val = nullptr;               // the term lm_score_tokens adds for that bigram (host logf)
  uint32_t bg_mask = 0;
  uint32_t* ngram_off = nullptr;         // [V + 1]
  uint32_t* ngram_ids = nullptr;
  uint32_t nvocab = 0;
  size_t built_vocab = 0, built_bigrams = 0;
};

struct LNode {   // one of the K best paths into a state: 12 B, written for every node
  float cost;
  uint32_t par;  // source state << 16 | rank there; 0xFFFFFFFF = the start node
  uint32_t sym;  // local symbol id, 0xFFFFFFFF = epsilon
};
struct LmNode {  // the LM side of a node, only written for the nodes that lie on a final path (the marks): 12 B
  float lp;      // f32 sum of the bigram terms of the path prefix
  uint32_t n;    // tokens summed
  int32_t prev;  // last token (-1 = out of vocabulary)
};

struct LmSym {   // what a symbol adds to a path's LM sum, as far as it does not depend on the path (k_lattice_lm): 32 B
  int32_t first, last;   // its first / last token (-1 = out of vocabulary)
  uint32_t ntok;         // tokens (n-gram parts, then the boundary text behind it); above LMSYM_TERMS + 1: terms[] is not used
  float terms[5];        // the bigram terms BETWEEN its tokens, in order: term(tok[i - 1], tok[i]), i = 1 .. ntok - 1
};
constexpr uint32_t LMSYM_TERMS = 5;

struct LatArgs {
  const LatStretch* st;
  uint32_t first, count;     // stretches index[first .. first + count) of the batch run in this launch
  const uint32_t* index;     // stretch ids in launch order (narrow ones paired by size for the two-per-wave kernel, then the wide ones)
  const uint32_t* in_off;    // per (stretch, state) CSR into arcs; a stretch owns nstates + 2 entries (virtual end state included)
  const LatArc* arcs;
  const LatSym* syms;
  const uint32_t* btok_off;  // per (stretch, boundary) CSR into btok; a stretch owns nb + 1 entries
  const int32_t* btok;
  LNode* nodes;              // node pool of this launch
  LmNode* lm;                // [same index]: LM sums of the marked nodes
  uint8_t* marks;            // a byte per node ((K + 3) & ~3 per state): the node lies on one of the final paths
  uint16_t* cnts;            // nodes per state, [node0 / K + state]: from k_lattice to k_lattice_lm
  LmSym* lmsym;              // [symbol]: filled and used by k_lattice_lm
  uint32_t K;
  uint32_t ring_max;         // cost lists the LDS ring of this launch holds (LatStretch::ring above it: host fallback)
  uint32_t cnt_cap;          // states (virtual end state included) of the launch's longest stretch: size of the per-state counts in LDS
  int use_lm;
  float lm_weight, variantmodel_weight, contextrules_weight;
  const unsigned long long* bg_key; const float* bg_val; uint32_t bg_mask;
  const uint32_t* ngram_off; const uint32_t* ngram_ids; uint32_t nvocab;
  uint32_t* out_n;           // per stretch: symbols of the chosen path, 0xFFFFFFFF = not decoded here (host fallback)
  uint32_t* out_syms;        // [LatStretch::out0 ..]
};

constexpr uint32_t LAT_MAX_STATES = 1024;   // per stretch, virtual end state included
constexpr float LAT_SMOOTH = -13.815510557964274f;  // src/search.rs:4

__device__ inline unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o; o >>= 1) {
    const unsigned long long other = ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)v, o);
    v = other < v ? other : v;
  }
  return v;
}
__device__ inline float lat_term(const LatArgs& a, int32_t x, int32_t y) {  // one bigram term of lm_score_tokens (src/lib.rs:2632-2674)
  if (x < 0 || y < 0) return LAT_SMOOTH;
  const unsigned long long key = (((unsigned long long)(uint32_t)x << 32) | (uint32_t)y) + 1ull;
  uint32_t h = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 32) & a.bg_mask;
  for (;;) {
    const unsigned long long k = a.bg_key[h];
    if (k == key) return a.bg_val[h];
    if (k == 0ull) return LAT_SMOOTH;
    h = (h + 1u) & a.bg_mask;
  }
}

// minimum of non-negative floats (as their bit patterns) over a GROUP of G lanes (G = 64: the wave; G = 32: each half on its own),
// result in every lane of the group: a rotate-and-min butterfly inside the rows of 16 lanes (DPP row_ror: every lane of a row ends up
// with the row's minimum), then gfx950's row swaps -- v_permlane16_swap of the value with itself leaves {row 0, row 0, row 2, row 2}
// in one register and {row 1, row 1, row 3, row 3} in the other, v_permlane32_swap the two halves -- and a min of the two registers.
// No trip through SGPRs (v_readlane + v_mov + select), which the pop loop's dependent chain used to wait for.
template <uint32_t G>
__device__ inline uint32_t group_min_u32(uint32_t v) {
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x121, 0xF, 0xF, false));  // row_ror:1
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x122, 0xF, 0xF, false));  // row_ror:2
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x124, 0xF, 0xF, false));  // row_ror:4
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x128, 0xF, 0xF, false));  // row_ror:8
  const auto r16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  v = min((uint32_t)r16[0], (uint32_t)r16[1]);
  if (G == 64) {
    const auto r32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    v = min((uint32_t)r32[0], (uint32_t)r32[1]);
  }
  return v;
}

template <uint32_t G>
__device__ inline uint32_t group_sum_u32(uint32_t v) {  // the same butterfly with adds: after the four rotations every lane of a row holds the row's sum
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xF, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xF, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);
  const auto r16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  v = (uint32_t)r16[0] + (uint32_t)r16[1];
  if (G == 64) {
    const auto r32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    v = (uint32_t)r32[0] + (uint32_t)r32[1];
  }
  return v;
}

// G lanes per stretch: G = 64 one stretch per wave (states with up to 128 incoming arcs, two candidate heads per lane); G = 32 two
// stretches per wave, side by side in its halves (up to 64 incoming arcs per state: the common case -- ~30 on the bench's text): the
// wave-wide minimum, the ballot and the bookkeeping of a pop then serve two lattices.  The host pairs stretches of similar size.
template <uint32_t G>
__global__ __launch_bounds__(64) void k_lattice(LatArgs a) {
  constexpr uint32_t NG = 64u / G;
  extern __shared__ float s_ring_all[];  // [NG][ring_max][K]: the costs of the K best paths of the last `ring` states (what a merge reads)
  const uint32_t lane = threadIdx.x, grp = lane / G, gl = lane % G;
  const uint32_t slot = blockIdx.x * NG + grp;
  const bool have = slot < a.count;
  const uint32_t si = a.index[a.first + (have ? slot : 0u)];
  const LatStretch S = a.st[si];
  const uint32_t K = a.K;
  float* __restrict__ s_ring = s_ring_all + (size_t)grp * a.ring_max * K;
  // the (incoming arc, rank at its source) of the nodes of the state being merged: its K nodes leave for HBM together, every lane
  // writing whole nodes (source state and symbol come from the arc again), instead of one 24-byte store by ONE lane per pop (two
  // memory instructions per path: ~0.7 G single-lane requests per 12.5 MB of text, which -- not the merge's instructions -- bounded
  // the kernel).  One word per node: LDS is what bounds the waves in flight.
  uint32_t* __restrict__ s_par = reinterpret_cast<uint32_t*>(s_ring_all + (size_t)NG * a.ring_max * K) + (size_t)grp * K;
  // nodes per state (sized by the launch's longest stretch: a fixed 1024 entries per group were 4 of the kernel's 14 KB of LDS, and
  // LDS is what bounds the waves in flight of this latency-bound kernel)
  uint16_t* __restrict__ s_cnt = reinterpret_cast<uint16_t*>(s_ring_all + (size_t)NG * (a.ring_max + 1u) * K) + (size_t)grp * a.cnt_cap;
  const uint32_t ns = S.nstates + 1u;  // with the virtual end state
  const uint32_t* __restrict__ ioff = a.in_off + S.in_off0;
  bool alive = have && !(ns > LAT_MAX_STATES || ns > a.cnt_cap || K > 0xFFFFu || S.ring == 0u || S.ring > a.ring_max);
  if (alive)
    for (uint32_t d = 1; d < ns; ++d)
      if (ioff[d + 1] - ioff[d] > 2u * G) { alive = false; break; }  // group-uniform
  if (have && !alive && gl == 0) a.out_n[si] = 0xFFFFFFFFu;  // not decoded here: the host decoder takes it
  LNode* __restrict__ nodes = a.nodes + (size_t)(S.node0);
  LmNode* __restrict__ lmn = a.lm + (size_t)(S.node0);
  const uint32_t ring = S.ring ? S.ring : 1u;
  uint32_t nsmax = alive ? ns : 0u;  // the wave's states loop runs as long as its longest stretch
#pragma unroll
  for (int o = 32; o >= (int)G && o < 64; o >>= 1) nsmax = max(nsmax, (uint32_t)__shfl_xor((int)nsmax, o));
  // ---- k best paths into every state ----------------------------------------------------------------------------------------
  if (alive && gl == 0) {
    nodes[0] = LNode{0.0f, 0xFFFFFFFFu, 0xFFFFFFFFu};  // the start node
    if (a.use_lm) lmn[0] = LmNode{0.0f, 0u, 0};        // <bos> (token 0), nothing summed yet
    s_cnt[0] = 1;
    s_ring[0] = 0.0f;
  }
  __syncthreads();
  for (uint32_t d = 1; d < nsmax; ++d) {
    const bool act = alive && d < ns;
    const uint32_t a0 = act ? ioff[d] : 0u, indeg = act ? ioff[d + 1] - a0 : 0u;
    float* __restrict__ mine = s_ring + (size_t)(d % ring) * K;  // this state's costs (sources are at most ring - 1 states back)
    // two candidate heads per lane: arcs gl and gl + G of the state's incoming list (ordered by source state, arc number).
    // hx: the NEXT cost of the head's list, fetched when the head moves up (the winner's list[r + 1] + arc cost is the same sum
    // either way).  ho: the list's place in the group's ring (an LDS index, not a pointer: ds_read instead of flat loads).
    // hc: the head's cost as its bit pattern, 0xFFFFFFFF = no head (exhausted list, no arc): larger than every cost.
    uint32_t hc[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
    float ac[2] = {0.0f, 0.0f}, hx[2] = {0.0f, 0.0f};
    uint32_t hr[2] = {0u, 0u}, hn[2] = {0u, 0u}, ho[2] = {0u, 0u};
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const uint32_t ai = gl + G * (uint32_t)w;
      if (ai < indeg) {
        const LatArc arc = a.arcs[S.arc0 + a0 + ai];
        ac[w] = arc.cost;
        hn[w] = s_cnt[arc.src];
        ho[w] = (arc.src % ring) * K;
        // (hx of a one-element list is the word behind it -- the next list, or the (arc, rank) array behind the rings: inside the
        // wave's LDS either way, and never used)
        if (hn[w]) { hc[w] = __float_as_uint(s_ring[ho[w]] + arc.cost); hx[w] = s_ring[ho[w] + 1u]; }
      }
    }
    // Pops of this state: K, or every candidate when the lists hold fewer -- known up front (group-uniform), so the loop counts in a
    // scalar register and the pop number is the place in the state's list.
    const uint32_t count = min(K, group_sum_u32<G>(hn[0] + hn[1]));
    uint32_t itmax_v = count;
#pragma unroll
    for (int o = 32; o >= (int)G && o < 64; o >>= 1) itmax_v = max(itmax_v, (uint32_t)__shfl_xor((int)itmax_v, o));
    const uint32_t itmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)itmax_v);
    const bool two = __any(indeg > G);  // wave-uniform: some group of the wave uses its second heads
    // A pop: the smallest cost (bit pattern of a non-negative float) over the group's heads, then the smallest arc among the heads
    // that have it -- arcs 0 .. G-1 (first heads, by lane) precede arcs G .. 2G-1 (second heads); the winning lane notes (arc, rank)
    // and the cost, and moves its head up.  TWO = false (no state of the wave has more than G incoming arcs: the common case)
    // compiles the loop without the second heads.
    // The kernel is bound by the instructions of this loop (a wave-wide vector instruction occupies its SIMD for 4 cycles; the 3-4
    // waves of a SIMD keep it busy): round 5 took it from ~33 vector instructions per pop to ~17 --
    //   * the winner is found in SCALAR registers: the compare writes a lane mask, the lowest set bit of each group's part of it
    //     (x & -x) IS the winner's lane, and that mask becomes EXEC (no per-lane "is a lane below me equal" arithmetic, no ballot
    //     round trip through a vector register);
    //   * the minimum is clamped to 0xFFFFFFFE: an exhausted head (0xFFFFFFFF) never equals it, so there is no "anything left" test;
    //   * the winner's update is branch-free (the new head cost is selected, the read of the cost after it runs one word past a
    //     list's end at worst), and the pop number is the loop counter.
    const uint32_t arcw[2] = {gl << 16, (gl + G) << 16};
    auto move_up = [&](int w, uint32_t it, uint32_t best) {
      const uint32_t rank = hr[w];
      // the sum first ("memory": the read below may not move above it): the read then lands in hx's own register -- scheduled ahead of
      // the sum it went to a temporary, and the copy into hx waited for it in the same pop
      float nx;
      asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(nx) : "v"(hx[w]), "v"(ac[w]) : "memory");
      hx[w] = s_ring[ho[w] + rank + 2u];
      s_par[it] = arcw[w] | rank;
      mine[it] = __uint_as_float(best);
      hr[w] = rank + 1u;
      hc[w] = rank + 1u < hn[w] ? __float_as_uint(nx) : 0xFFFFFFFFu;
    };
    auto lowbit = [](uint32_t x) { return x & (0u - x); };
    auto pops = [&](auto two_c) {
      constexpr bool TWO = decltype(two_c)::value;
      auto pop = [&](uint32_t it) {
        const uint32_t best = min(group_min_u32<G>(TWO ? min(hc[0], hc[1]) : hc[0]), 0xFFFFFFFEu);
        const unsigned long long m0 = __builtin_amdgcn_uicmp(hc[0], best, 32 /* == */);
        unsigned long long w0, w1 = 0ull;
        if (G == 64) {
          w0 = m0 & (0ull - m0);
          if (TWO) { const unsigned long long m1 = __builtin_amdgcn_uicmp(hc[1], best, 32); w1 = m0 ? 0ull : (m1 & (0ull - m1)); }
        } else {
          const uint32_t l0 = (uint32_t)m0, h0 = (uint32_t)(m0 >> 32);
          w0 = (unsigned long long)lowbit(l0) | ((unsigned long long)lowbit(h0) << 32);
          if (TWO) {
            const unsigned long long m1 = __builtin_amdgcn_uicmp(hc[1], best, 32);
            const uint32_t l1 = l0 ? 0u : (uint32_t)m1, h1 = h0 ? 0u : (uint32_t)(m1 >> 32);  // a group's second heads only when none of its first heads has it
            w1 = (unsigned long long)lowbit(l1) | ((unsigned long long)lowbit(h1) << 32);
          }
        }
        if (__builtin_amdgcn_inverse_ballot_w64(w0)) move_up(0, it, best);
        if (TWO && __builtin_amdgcn_inverse_ballot_w64(w1)) move_up(1, it, best);
      };
      uint32_t it = 0;
      for (; it + 4u <= itmax; it += 4u) { pop(it); pop(it + 1u); pop(it + 2u); pop(it + 3u); }  // (four per trip: one taken branch and one pair of LDS address updates per four pops)
      for (; it < itmax; ++it) pop(it);
    };
    if (two) pops(std::true_type{}); else pops(std::false_type{});
    if (act && gl == 0) s_cnt[d] = (uint16_t)count;
    __syncthreads();  // the state's costs (and its count) are read by the states behind it
    if (act)
      for (uint32_t r = gl; r < count; r += G) {
        const uint32_t pk = s_par[r];
        const LatArc arc = a.arcs[S.arc0 + a0 + (pk >> 16)];
        nodes[(size_t)d * K + r] = LNode{mine[r], (arc.src << 16) | (pk & 0xFFFFu), arc.sym};
      }
  }
  // the nodes per state and "decoded so far" for the second kernel (the language model and the choice of the path need no LDS: with
  // the cost rings out of the way twice as many waves are in flight for that latency-bound part, and this kernel's waves make room
  // for the next ones as soon as their merges are done)
  if (alive) {
    uint16_t* __restrict__ c = a.cnts + (size_t)(S.node0 / K);
    for (uint32_t d = gl; d < ns; d += G) c[d] = s_cnt[d];
    if (gl == 0) a.out_n[si] = 0u;
  }
}

// The second half of a stretch's decoding: LM sums of the nodes on the final paths, rerank, the chosen path's symbols.  Same launch
// geometry as k_lattice (G lanes per stretch), no LDS.
template <uint32_t G>
__global__ __launch_bounds__(64) void k_lattice_lm(LatArgs a) {
  constexpr uint32_t NG = 64u / G;
  const uint32_t lane = threadIdx.x, grp = lane / G, gl = lane % G;
  const uint32_t slot = blockIdx.x * NG + grp;
  const bool have = slot < a.count;
  const uint32_t si = a.index[a.first + (have ? slot : 0u)];
  const LatStretch S = a.st[si];
  const uint32_t K = a.K;
  const uint32_t ns = S.nstates + 1u;  // with the virtual end state
  bool alive = have && a.out_n[si] != 0xFFFFFFFFu;   // (k_lattice leaves 0xFFFFFFFF for what it hands to the host decoder)
  LNode* __restrict__ nodes = a.nodes + (size_t)(S.node0);
  LmNode* __restrict__ lmn = a.lm + (size_t)(S.node0);
  const uint32_t MK = (K + 3u) & ~3u;  // mark bytes per state
  uint8_t* __restrict__ marks = a.marks + (size_t)(S.node0 / K) * MK;
  const uint16_t* __restrict__ s_cnt = a.cnts + (size_t)(S.node0 / K);
  uint32_t nsmax = alive ? ns : 0u;  // the wave's states loops run as long as its longest stretch
#pragma unroll
  for (int o = 32; o >= (int)G && o < 64; o >>= 1) nsmax = max(nsmax, (uint32_t)__shfl_xor((int)nsmax, o));
  const uint32_t end = ns - 1u;
  uint32_t npaths = alive ? s_cnt[end] : 0u;
  if (alive && npaths == 0) { if (gl == 0) a.out_n[si] = 0xFFFFFFFFu; alive = false; }  // no complete path (cannot happen: the epsilon chain): host
  // ---- LM: (log-probability sum, tokens, last token) of every node on a final path -------------------------------------------
  if (a.use_lm) {
    // The nodes on the final paths, state by state from the end: a marked node marks its parent (a byte per node in HBM, plain
    // stores of 1 -- every writer stores the same; the parent lies in an earlier state, so a state's marks are complete when its
    // turn comes).  One pass of ~nstates steps with every lane at work, instead of one walk per final path (250 walks of ~20
    // dependent loads, 8 of them per lane).
    if (alive) {
      uint32_t* __restrict__ mw = reinterpret_cast<uint32_t*>(marks);
      for (uint32_t w = gl; w < ns * (MK / 4u); w += G) mw[w] = 0u;
    }
    __syncthreads();
    if (alive)
      for (uint32_t i = gl; i < npaths; i += G) marks[(size_t)end * MK + i] = 1;
    __syncthreads();
    for (uint32_t dd = 1; dd < nsmax; ++dd) {  // (the wave's loop runs as long as its longest stretch)
      if (alive && dd < ns) {
        const uint32_t d = ns - dd;  // end .. 1
        const uint32_t cnt = s_cnt[d];
        for (uint32_t r = gl; r < cnt; r += G)
          if (marks[(size_t)d * MK + r]) {
            const uint32_t par = nodes[(size_t)d * K + r].par;
            if (par != 0xFFFFFFFFu) marks[(size_t)(par >> 16) * MK + (par & 0xFFFFu)] = 1;
          }
      }
      __syncthreads();
    }
    const uint32_t* __restrict__ boff = a.btok_off + S.btok_off0;
    // Per SYMBOL, once (a lane per arc of the stretch): its tokens' first / last / count and the bigram terms between them -- none of
    // that depends on the path.  A node then adds term(parent's last token, first) and the stored terms, in the same order as the
    // token walk below (the same float additions), with one hash look-up instead of one per token: the walk's chain of dependent
    // loads (symbol -> n-gram offsets -> ids -> key -> value, per token) is what this kernel's time was.
    if (alive) {
      const uint32_t* __restrict__ ioff = a.in_off + S.in_off0;
      const uint32_t narcs = ioff[ns];
      for (uint32_t ai = gl; ai < narcs; ai += G) {
        const uint32_t sym = a.arcs[S.arc0 + ai].sym;
        if (sym == 0xFFFFFFFFu) continue;
        const LatSym sy = a.syms[S.sym0 + sym];
        LmSym rec;
        rec.first = -1; rec.last = -1; rec.ntok = 0u;
#pragma unroll
        for (uint32_t i = 0; i < LMSYM_TERMS; ++i) rec.terms[i] = 0.0f;
        auto push = [&](int32_t t) {
          if (rec.ntok == 0u) rec.first = t;
          else {
            const float tm = rec.ntok <= LMSYM_TERMS ? lat_term(a, rec.last, t) : 0.0f;
#pragma unroll
            for (uint32_t i = 0; i < LMSYM_TERMS; ++i)   // (no run-time index into a register array)
              if (rec.ntok == i + 1u) rec.terms[i] = tm;
          }
          rec.last = t;
          ++rec.ntok;
        };
        if (sy.vocab_id == 0u) push(-1);
        else if (sy.vocab_id < a.nvocab)
          for (uint32_t k = a.ngram_off[sy.vocab_id]; k < a.ngram_off[sy.vocab_id + 1]; ++k) push((int32_t)a.ngram_ids[k]);
        for (uint32_t k = boff[sy.boundary]; k < boff[sy.boundary + 1]; ++k) push(a.btok[S.btok0 + k]);
        a.lmsym[S.sym0 + sym] = rec;
      }
    }
    __syncthreads();
    for (uint32_t d = 1; d < nsmax; ++d) {
      const uint32_t cnt = (alive && d < ns) ? s_cnt[d] : 0u;
      for (uint32_t r = gl; r < cnt; r += G) {
        if (!marks[(size_t)d * MK + r]) continue;
        const LNode nd = nodes[(size_t)d * K + r];
        const LmNode pa = lmn[(size_t)(nd.par >> 16) * K + (nd.par & 0xFFFFu)];
        float lp = pa.lp;
        uint32_t n = pa.n;
        int32_t prev = pa.prev;
        const LmSym rec = nd.sym != 0xFFFFFFFFu ? a.lmsym[S.sym0 + nd.sym] : LmSym{-1, -1, 0u, {0.0f, 0.0f, 0.0f, 0.0f, 0.0f}};
        if (nd.sym != 0xFFFFFFFFu && rec.ntok <= LMSYM_TERMS + 1u) {
          if (rec.ntok) {
            lp += lat_term(a, prev, rec.first);
#pragma unroll
            for (uint32_t i = 0; i < LMSYM_TERMS; ++i)
              if (i + 1u < rec.ntok) lp += rec.terms[i];
            n += rec.ntok;
            prev = rec.last;
          }
        } else if (nd.sym != 0xFFFFFFFFu) {  // the tokens of the symbol: its n-gram parts, then the boundary text behind it (src/lib.rs:2580-2629)
          const LatSym sy = a.syms[S.sym0 + nd.sym];
          if (sy.vocab_id == 0u) { lp += lat_term(a, prev, -1); ++n; prev = -1; }
          else if (sy.vocab_id < a.nvocab)
            for (uint32_t k = a.ngram_off[sy.vocab_id]; k < a.ngram_off[sy.vocab_id + 1]; ++k) {
              const int32_t t = (int32_t)a.ngram_ids[k];
              lp += lat_term(a, prev, t); ++n; prev = t;
            }
          for (uint32_t k = boff[sy.boundary]; k < boff[sy.boundary + 1]; ++k) {
            const int32_t t = a.btok[S.btok0 + k];
            lp += lat_term(a, prev, t); ++n; prev = t;
          }
        }
        lmn[(size_t)d * K + r] = LmNode{lp, n, prev};
      }
      __syncthreads();
    }
  }
  // ---- rerank (src/lib.rs:2318-2425): no context rules here (the host decodes models that have them) -----------------------------
  double best_ppl = 999999.0;
  float best_cost = S.best_cost_init;
  for (uint32_t i = gl; i < npaths; i += G) {
    const LNode& nd = nodes[(size_t)end * K + i];
    if (a.use_lm) {
      const LmNode lm = lmn[(size_t)end * K + i];
      const float logprob = lm.lp + lat_term(a, lm.prev, 1);  // <eos>
      const double ppl = -1.0 / (double)(lm.n + 1u) * (double)logprob;
      if (ppl < best_ppl) best_ppl = ppl;
    }
    if (nd.cost < best_cost) best_cost = nd.cost;
  }
#pragma unroll
  for (int o = (int)G / 2; o; o >>= 1) {  // within the group
    const double op = __shfl_xor(best_ppl, o);
    const float oc = __shfl_xor(best_cost, o);
    best_ppl = op < best_ppl ? op : best_ppl;
    best_cost = oc < best_cost ? oc : best_cost;
  }
  const bool shortcut = !a.use_lm;  // (!have_lm || lm_weight == 0) && no rules
  double my_score = 0.0;
  uint32_t my_i = 0xFFFFFFFFu;
  for (uint32_t i = gl; i < npaths; i += G) {
    const LNode& nd = nodes[(size_t)end * K + i];
    double norm_lm = 0.0;
    if (a.use_lm) {
      const LmNode lm = lmn[(size_t)end * K + i];
      const float logprob = lm.lp + lat_term(a, lm.prev, 1);
      const double ppl = -1.0 / (double)(lm.n + 1u) * (double)logprob;
      norm_lm = portable_log(best_ppl / ppl);
    }
    const double norm_var = portable_log((double)best_cost / (double)nd.cost);
    const double norm_ctx = portable_log(1.0 / 1.0);
    double score;
    if (shortcut) score = norm_var;
    else
      score = ((double)a.lm_weight * norm_lm + (double)a.variantmodel_weight * norm_var + (double)a.contextrules_weight * norm_ctx) /
              ((double)a.lm_weight + (double)a.variantmodel_weight + (double)a.contextrules_weight);
    if (my_i == 0xFFFFFFFFu || score > my_score) { my_score = score; my_i = i; }  // first maximum of this lane's paths (ascending i)
  }
  // first maximum over the group: larger score wins, equal scores: the smaller path index.  (A NaN score never wins a comparison,
  // on the host neither: path 0 stands unless a later score is greater.)
#pragma unroll
  for (int o = (int)G / 2; o; o >>= 1) {
    const double os = __shfl_xor(my_score, o);
    const uint32_t oi = (uint32_t)__shfl_xor((int)my_i, o);
    const bool take = oi != 0xFFFFFFFFu && (my_i == 0xFFFFFFFFu || os > my_score || (os == my_score && oi < my_i) || (my_score != my_score && oi < my_i && !(os != os)));
    if (take) { my_score = os; my_i = oi; }
  }
  if (alive && gl == 0) {  // the winner's symbols, walked back over the back-pointers, written in path order
    uint32_t st_ = end, r = my_i, cnt = 0;
    for (;;) {
      const LNode& nd = nodes[(size_t)st_ * K + r];
      if (nd.par == 0xFFFFFFFFu) break;
      if (nd.sym != 0xFFFFFFFFu) ++cnt;
      st_ = nd.par >> 16; r = nd.par & 0xFFFFu;
    }
    a.out_n[si] = cnt;
    uint32_t w = cnt;
    st_ = end; r = my_i;
    for (;;) {
      const LNode& nd = nodes[(size_t)st_ * K + r];
      if (nd.par == 0xFFFFFFFFu) break;
      if (nd.sym != 0xFFFFFFFFu) a.out_syms[S.out0 + --w] = nd.sym;
      st_ = nd.par >> 16; r = nd.par & 0xFFFFu;
    }
  }
}

// ---- host driver -----------------------------------------------------------------------------------------------------------
namespace {
std::mutex g_lm_mu;
template <typename T>
int lt_upload(T** dst, const void* src, size_t count, std::string& err) {
  if (*dst) { pool_free(*dst); *dst = nullptr; }
  HIP_TRY(pool_malloc(reinterpret_cast<void**>(dst), std::max<size_t>(count * sizeof(T), 16)));
  if (count) HIP_TRY(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
  return ANX_OK;
}
}  // namespace

void lm_free(DeviceLm* d) {
  if (!d) return;
  (void)hipSetDevice(d->device);
  for (void* p : {(void*)d->bg_key, (void*)d->bg_val, (void*)d->ngram_off, (void*)d->ngram_ids})
    if (p) pool_free(p);
  delete d;
}

static int lm_ensure(const HostModel& m, const DeviceLexicon* dl, std::string& err) {
  std::lock_guard<std::mutex> g(g_lm_mu);
  DeviceLm*& d = dl->dlm;
  if (!d) { d = new DeviceLm(); d->device = dl->device; }
  if (d->bg_key && d->built_vocab == m.decoder.size() && d->built_bigrams == m.bigrams.size()) return ANX_OK;
  // the bigram terms exactly as search.cpp's term() computes them (src/lib.rs:2632-2674), on the host
  uint32_t cap = 64;
  while (cap < 2 * m.bigrams.size() + 16) cap <<= 1;
  std::vector<unsigned long long> keys(cap, 0ull);
  std::vector<float> vals(cap, 0.0f);
  for (const auto& kv : m.bigrams) {
    const uint64_t aa = kv.first >> 32;
    auto pit = m.unigrams.find(aa);
    const uint32_t prior = pit == m.unigrams.end() ? 1u : pit->second;
    const float v = prior < kv.second ? logf((float)kv.second) : logf((float)kv.second / (float)prior);
    const unsigned long long key = kv.first + 1ull;
    uint32_t h = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 32) & (cap - 1);
    while (keys[h]) h = (h + 1) & (cap - 1);
    keys[h] = key;
    vals[h] = v;
  }
  int rc;
  std::vector<uint32_t> off = m.ngram_off, ids = m.ngram_ids;
  if (off.empty()) off.assign(1, 0u);
  if ((rc = lt_upload(&d->bg_key, keys.data(), keys.size(), err)) || (rc = lt_upload(&d->bg_val, vals.data(), vals.size(), err)) ||
      (rc = lt_upload(&d->ngram_off, off.data(), off.size(), err)) || (rc = lt_upload(&d->ngram_ids, ids.data(), ids.size(), err)))
    return rc;
  d->bg_mask = cap - 1;
  d->nvocab = (uint32_t)(off.size() - 1);
  d->built_vocab = m.decoder.size();
  d->built_bigrams = m.bigrams.size();
  return ANX_OK;
}

// The launches of a set of n lattices whose arrays are (or are about to be) on the device: launch order, node pools, kernels.
// hst: the stretches (indices relative to the device arrays; node0 is assigned here), maxdeg[j]: (an upper bound of) the most
// incoming arcs of a state of stretch j.  The planned stretch array is uploaded to d_st, then after_stretch_upload (may be empty)
// enqueues whatever still has to happen to the arrays on `st` before the kernels run.  Blocks allocated here go to `owned`.
static int lattice_launch(const HostModel& m, const DeviceLexicon* dl, std::vector<LatStretch>& hst, const std::vector<uint32_t>& maxdeg, LatStretch* d_st,
                          const uint32_t* d_inoff, const LatArc* d_arcs, const LatSym* d_syms, size_t nsyms_cap, const uint32_t* d_boff, const int32_t* d_btok, uint32_t* d_outn,
                          uint32_t* d_outs, const anx_search_params& p, hipStream_t st, const std::function<int()>& after_stretch_upload, std::vector<void*>& owned,
                          std::string& err) {
  const size_t n = hst.size();
  const DeviceLm* lm = dl->dlm;
  const uint32_t K = std::max<uint32_t>(1u, p.max_seq);
  LNode* d_nodes = nullptr;
  int rc;
  auto dalloc_ = [&](void** p_, size_t bytes) -> int { HIP_TRY(pool_malloc(p_, std::max<size_t>(bytes, 16))); owned.push_back(*p_); return ANX_OK; };
  // node pool: (nstates + 1) * K nodes per stretch; launches of as many stretches as fit the budget
  const size_t budget_nodes = ((size_t)6 << 30) / sizeof(LNode);
  // Launch order: the stretches none of whose states has more than 64 incoming arcs (and whose cost ring fits half the LDS budget)
  // first, by decreasing number of states -- k_lattice<32> decodes them two per wave, neighbours of this order side by side --,
  // then the others (k_lattice<64>: one per wave, up to 128 incoming arcs).
  const uint32_t ring_cap64 = (uint32_t)std::max<size_t>(3, ((size_t)48 << 10) / ((size_t)K * sizeof(float))) - 1u;  // one list of K words beside the rings
  const uint32_t ring_cap32 = (uint32_t)std::max<size_t>(3, ((size_t)24 << 10) / ((size_t)K * sizeof(float))) - 1u;
  std::vector<uint32_t> order;
  order.reserve(n);
  std::vector<uint32_t> wide;
  for (size_t j = 0; j < n; ++j) {
    if (maxdeg[j] <= 64u && hst[j].ring <= ring_cap32) order.push_back((uint32_t)j);
    else wide.push_back((uint32_t)j);
  }
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return hst[x].nstates > hst[y].nstates; });
  const size_t n_narrow = order.size();
  order.insert(order.end(), wide.begin(), wide.end());
  struct Launch { uint32_t first, count, lanes; };
  std::vector<Launch> launches;
  size_t max_pool = 0;
  for (size_t i = 0; i < n;) {
    const size_t stop = i < n_narrow ? n_narrow : n;  // a launch holds stretches of one kind
    size_t used = 0, j = i;
    while (j < stop) {
      const size_t need = (size_t)(hst[order[j]].nstates + 1) * K;
      if (j > i && used + need > budget_nodes) break;
      hst[order[j]].node0 = used;
      used += need;
      ++j;
    }
    launches.push_back(Launch{(uint32_t)i, (uint32_t)(j - i), i < n_narrow ? 32u : 64u});
    max_pool = std::max(max_pool, used);
    i = j;
  }
  uint32_t* d_index = nullptr;
  if ((rc = dalloc_((void**)&d_index, n * 4))) return rc;
  HIP_TRY(hipMemcpyAsync(d_index, order.data(), n * 4, hipMemcpyHostToDevice, st));
  if ((rc = dalloc_((void**)&d_nodes, max_pool * sizeof(LNode)))) return rc;
  LmNode* d_lm = nullptr;
  uint8_t* d_marks = nullptr;
  const bool lm_on = m.have_lm && p.lm_weight > 0.0f;
  if (lm_on && ((rc = dalloc_((void**)&d_lm, max_pool * sizeof(LmNode))) || (rc = dalloc_((void**)&d_marks, max_pool / K * ((K + 3u) & ~3u))))) return rc;
  uint16_t* d_cnts = nullptr;
  if ((rc = dalloc_((void**)&d_cnts, (max_pool / K + 1) * sizeof(uint16_t)))) return rc;
  LmSym* d_lmsym = nullptr;
  if (lm_on && (rc = dalloc_((void**)&d_lmsym, (nsyms_cap + 1) * sizeof(LmSym)))) return rc;
  HIP_TRY(hipMemcpyAsync(d_st, hst.data(), n * sizeof(LatStretch), hipMemcpyHostToDevice, st));
  if (after_stretch_upload) { const int rcu = after_stretch_upload(); if (rcu) return rcu; }  // the caller's uploads / kernels that complete the lattice arrays
  LatArgs a;
  a.st = d_st; a.in_off = d_inoff; a.arcs = d_arcs; a.syms = d_syms; a.btok_off = d_boff; a.btok = d_btok; a.nodes = d_nodes; a.lm = d_lm; a.marks = d_marks; a.cnts = d_cnts; a.lmsym = d_lmsym; a.K = K;
  a.use_lm = (m.have_lm && p.lm_weight > 0.0f) ? 1 : 0;
  a.lm_weight = p.lm_weight; a.variantmodel_weight = p.variantmodel_weight; a.contextrules_weight = p.contextrules_weight;
  a.bg_key = lm->bg_key; a.bg_val = lm->bg_val; a.bg_mask = lm->bg_mask; a.ngram_off = lm->ngram_off; a.ngram_ids = lm->ngram_ids; a.nvocab = lm->nvocab;
  a.out_n = d_outn; a.out_syms = d_outs;
  a.index = d_index;
  // LDS ring of cost lists: as many states as the widest arc of the launch's stretches spans (+ 1), capped by 48 KB per wave
  for (const Launch& l : launches) {
    uint32_t ring_need = 2;
    for (uint32_t i = 0; i < l.count; ++i) ring_need = std::max(ring_need, hst[order[l.first + i]].ring);
    a.ring_max = std::min(ring_need, l.lanes == 32u ? ring_cap32 : ring_cap64);
    uint32_t cnt_cap = 2;
    for (uint32_t i = 0; i < l.count; ++i) cnt_cap = std::max(cnt_cap, std::min(hst[order[l.first + i]].nstates + 1u, LAT_MAX_STATES));
    a.cnt_cap = (cnt_cap + 1u) & ~1u;
    // cost rings + the merged state's (arc, rank) list + the per-state node counts
    const size_t lds = (size_t)(64u / l.lanes) * (((size_t)a.ring_max + 1u) * K * sizeof(float) + (size_t)a.cnt_cap * sizeof(uint16_t));
    a.first = l.first; a.count = l.count;
    const int kt = ktimer_begin("k_lattice", st);
    if (l.lanes == 32u) hipLaunchKernelGGL(k_lattice<32>, dim3((l.count + 1u) / 2u), dim3(64), lds, st, a);
    else hipLaunchKernelGGL(k_lattice<64>, dim3(l.count), dim3(64), lds, st, a);
    ktimer_end(kt, st);
    const int kt2 = ktimer_begin("k_lattice_lm", st);
    if (l.lanes == 32u) hipLaunchKernelGGL(k_lattice_lm<32>, dim3((l.count + 1u) / 2u), dim3(64), 0, st, a);
    else hipLaunchKernelGGL(k_lattice_lm<64>, dim3(l.count), dim3(64), 0, st, a);
    ktimer_end(kt2, st);
  }
  HIP_TRY(hipGetLastError());
  return ANX_OK;
}

// Decodes the stretches of `in` on the replica `dl`.  out_n[i] = symbols of the chosen path of stretch i (0xFFFFFFFF: not decoded,
// the caller's host decoder takes it), out_syms[st[i].out0 ..] = their local symbol ids in path order.
int lattice_decode(const HostModel& m, const DeviceLexicon* dl, const LatView& whole, size_t first, size_t count, const anx_search_params& p,
                   uint32_t* out_n, uint32_t* out_syms, std::string& err) {
  if (!dl) { err = "model is not resident on a device"; return ANX_ENODEVICE; }
  HIP_TRY(hipSetDevice(dl->device));
  if (!count) return ANX_OK;
  // the sub-range [first, first + count) of the call's lattices as a view of its own: the arrays of consecutive stretches are
  // consecutive, so only their part is uploaded (a replica of a multi-device model decodes its share)
  const LatStretch& s0 = whole.st[first];
  const bool last = first + count == whole.nst;
  const LatStretch* s1 = last ? nullptr : &whole.st[first + count];
  LatView in;
  in.st = whole.st + first; in.nst = count;
  in.in_off = whole.in_off + s0.in_off0; in.nin = (last ? whole.nin : s1->in_off0) - s0.in_off0;
  in.arcs = whole.arcs + s0.arc0; in.narcs = (last ? whole.narcs : s1->arc0) - s0.arc0;
  in.syms = whole.syms + s0.sym0; in.nsyms = (last ? whole.nsyms : s1->sym0) - s0.sym0;
  in.btok_off = whole.btok_off + s0.btok_off0; in.nboff = (last ? whole.nboff : s1->btok_off0) - s0.btok_off0;
  in.btok = whole.btok + s0.btok0; in.nbtok = (last ? whole.nbtok : s1->btok0) - s0.btok0;
  in.out_total = (last ? whole.out_total : s1->out0) - s0.out0;
  out_n += first;
  out_syms += s0.out0;
  const size_t n = count;
  int rc = lm_ensure(m, dl, err);
  if (rc) return rc;
  const uint32_t K = std::max<uint32_t>(1u, p.max_seq);
  if (K > 4096u) {  // node pools of (states x K) and the LDS cost ring are sized for the reference's default of 250: the host decoder takes these
    for (size_t i = 0; i < count; ++i) out_n[i] = 0xFFFFFFFFu;
    return ANX_OK;
  }
  hipStream_t st = encoder_stream_acquire(dl->device);
  struct Rel { hipStream_t s; int dev; ~Rel() { (void)hipStreamSynchronize(s); encoder_stream_release(dev, s); } } rel{st, dl->device};
  LatStretch* d_st = nullptr; uint32_t* d_inoff = nullptr; LatArc* d_arcs = nullptr; LatSym* d_syms = nullptr; uint32_t* d_boff = nullptr;
  int32_t* d_btok = nullptr; uint32_t *d_outn = nullptr, *d_outs = nullptr;
  std::vector<void*> owned;
  auto dalloc_ = [&](void** p_, size_t bytes) -> int { HIP_TRY(pool_malloc(p_, std::max<size_t>(bytes, 16))); owned.push_back(*p_); return ANX_OK; };
  struct Free { std::vector<void*>& v; hipStream_t s; ~Free() { (void)hipStreamSynchronize(s); for (void* q : v) pool_free(q); } } fr{owned, st};
  const size_t nout = in.out_total;
  if ((rc = dalloc_((void**)&d_st, n * sizeof(LatStretch))) || (rc = dalloc_((void**)&d_inoff, in.nin * 4)) ||
      (rc = dalloc_((void**)&d_arcs, in.narcs * sizeof(LatArc))) || (rc = dalloc_((void**)&d_syms, in.nsyms * sizeof(LatSym))) ||
      (rc = dalloc_((void**)&d_boff, in.nboff * 4)) || (rc = dalloc_((void**)&d_btok, in.nbtok * 4)) ||
      (rc = dalloc_((void**)&d_outn, n * 4)) || (rc = dalloc_((void**)&d_outs, nout * 4)))
    return rc;
  std::vector<LatStretch> hst(in.st, in.st + n);
  std::vector<uint32_t> maxdeg(n, 0u);
  for (size_t j = 0; j < n; ++j) {  // indices relative to the uploaded parts
    LatStretch& S = hst[j];
    S.in_off0 -= s0.in_off0; S.arc0 -= s0.arc0; S.sym0 -= s0.sym0; S.btok_off0 -= s0.btok_off0; S.btok0 -= s0.btok0; S.out0 -= s0.out0;
    const uint32_t* io = in.in_off + S.in_off0;
    for (uint32_t d = 1; d <= S.nstates; ++d) maxdeg[j] = std::max(maxdeg[j], io[d + 1] - io[d]);
  }
  auto uploads = [&]() -> int {
    HIP_TRY(hipMemcpyAsync(d_inoff, in.in_off, in.nin * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_arcs, in.arcs, in.narcs * sizeof(LatArc), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_syms, in.syms, in.nsyms * sizeof(LatSym), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_boff, in.btok_off, in.nboff * 4, hipMemcpyHostToDevice, st));
    if (in.nbtok) HIP_TRY(hipMemcpyAsync(d_btok, in.btok, in.nbtok * 4, hipMemcpyHostToDevice, st));
    return ANX_OK;
  };
  if ((rc = lattice_launch(m, dl, hst, maxdeg, d_st, d_inoff, d_arcs, d_syms, in.nsyms, d_boff, d_btok, d_outn, d_outs, p, st, uploads, owned, err))) return rc;
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out_n, d_outn, n * 4, hipMemcpyDeviceToHost, st));
  if (nout) HIP_TRY(hipMemcpyAsync(out_syms, d_outs, nout * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return ANX_OK;
}


// ------------------------------------------------------------------------------------------------------------------------------------
// Search mode in ONE device pass (round 5).  Until round 4 a part of a find_all_matches call downloaded every ranked row of every
// segment, built row views and the lattice input on the host threads (search.cpp build_lattice) and uploaded it again.  Here the
// lattices are BUILT on the device from the device-resident rows of the part's two batches (unigram segments; segments of the higher
// orders): the host only describes the structure that does not depend on the results -- which segment connects which boundaries
// (states), in which order the arcs of a state are listed -- and the device fills in what does: redundant_match
// (/root/reference/src/search.rs:317-336: a higher-order segment whose unigrams all have an exact match is not looked up), one arc and
// one output symbol per variant with cost = n + (1 - score) (/root/reference/src/lib.rs:2104-2276), out-of-vocabulary arcs for
// unigrams without variants, the fail-safe epsilon arcs and the arcs into the virtual end state.  k_lattice decodes as before; only
// the matches ON the chosen paths come back, with their rows (a third of all ranked rows on BASELINE configs[4]).
//   arcs of a lattice are listed per destination state in (source state, insertion) order: the host lays the "arc groups" (a
//   segment's variants | the epsilon arc of a state | an arc into the end state) out in exactly that order, the device counts the
//   arcs of every group and an exclusive scan over the groups IS the arc layout; symbols are numbered in segment order (a second scan).
// ------------------------------------------------------------------------------------------------------------------------------------
struct OpArgs {
  uint32_t nmatch, ngroup, nin, nst;
  const uint32_t *m_q, *m_u0, *m_u1, *m_pack, *m_lat, *g_ref, *e_g0, *e_lat, *st_m0, *st_e0;
  // the two batches: ranked rows per sorted query, input index -> sorted query
  const uint32_t *inv_u, *inv_h, *soff_u, *soff_h, *cnt_u, *cnt_h;
  const DevRow *rows_u, *rows_h;
  uint32_t nin_u, nin_h;      // inputs of the batches
  float freq_weight;
  uint32_t *mc, *gc;          // symbols per match, arcs per group
  uint8_t* mflag;             // bit 0: looked up (not redundant)
  uint32_t *sym_off, *arc_off;
  LatStretch* st;
  uint32_t* in_off;
  LatArc* arcs;
  LatSym* syms;
  uint2* refs;                // per symbol: (match index within its stretch, variant index | 0xFFFFFFFF)
  uint32_t narc_cap, nsym_cap;
  uint32_t* overflow;
  // emit
  const uint32_t *out_n, *out_syms;
  uint32_t *e_match, *e_sel, *e_cnt, *e_row0;
  anx_result* e_rows;
  uint32_t out_total;
};
__global__ __launch_bounds__(256) void k_op_inv(uint32_t nq, const uint32_t* __restrict__ q_orig, uint32_t* __restrict__ inv) {
  const uint32_t s = blockIdx.x * 256 + threadIdx.x;
  if (s < nq) inv[q_orig[s]] = s;
}
__device__ inline uint32_t op_rows_of(const OpArgs& a, uint32_t order, uint32_t q, uint32_t* sorted) {
  *sorted = 0xFFFFFFFFu;
  if (q == 0xFFFFFFFFu) return 0u;
  const bool uni = order == 1u;
  if (q >= (uni ? a.nin_u : a.nin_h)) return 0u;
  const uint32_t sq = (uni ? a.inv_u : a.inv_h)[q];
  if (sq == 0xFFFFFFFFu) return 0u;  // not encodable (empty / too long): no variants
  *sorted = sq;
  return (uni ? a.cnt_u : a.cnt_h)[sq];
}
// symbols (= arcs) a segment contributes: its variants, or one out-of-vocabulary symbol for a unigram without variants; nothing for a
// redundant higher-order segment (redundant_match, src/search.rs:317-336, decided from the unigrams' device-resident rows) and for a
// segment that ends at no boundary of its stretch
__global__ __launch_bounds__(256) void k_op_counts(OpArgs a) {
  const uint32_t m = blockIdx.x * 256 + threadIdx.x;
  if (m >= a.nmatch) return;
  const uint32_t pk = a.m_pack[m], order = (pk >> 24) & 0x7Fu;
  bool looked = true;
  if (order > 1u) {  // redundant <=> every unigram inside has variants and its best one is an exact match (dist_score >= 1.0)
    bool red = true;
    for (uint32_t u = a.m_u0[m]; u < a.m_u1[m] && red; ++u) {
      uint32_t sq;
      const uint32_t n = op_rows_of(a, 1u, a.m_q[u], &sq);
      if (n == 0u || a.rows_u[a.soff_u[sq]].dist_score < 1.0) red = false;
    }
    looked = !red;
  }
  uint32_t sq;
  const uint32_t rows = looked ? op_rows_of(a, order, a.m_q[m], &sq) : 0u;
  uint32_t c = rows ? rows : ((looked && order == 1u) ? 1u : 0u);
  if ((pk >> 31) || a.m_lat[m] == 0xFFFFFFFFu) c = 0u;  // no destination state / a stretch without a lattice
  a.mc[m] = c;
  a.mflag[m] = looked ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_op_gcount(OpArgs a) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= a.ngroup) return;
  const uint32_t r = a.g_ref[g];
  a.gc[g] = (r >> 30) == 0u ? a.mc[r & 0x3FFFFFFFu] : 1u;
}
__device__ inline double op_vr_score(const DevRow& r, float fw) {  // src/types.rs:335-341 (search.cpp vr_score)
  if (fw == 0.0f) return r.dist_score;
  return (r.dist_score + ((double)fw * r.freq_score)) / (1.0 + (double)fw);
}
__global__ __launch_bounds__(256) void k_op_fill(OpArgs a) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= a.ngroup) return;
  const uint32_t r = a.g_ref[g], kind = r >> 30, v = r & 0x3FFFFFFFu;
  const uint32_t ao = a.arc_off[g];
  if (kind != 0u) {  // fail-safe epsilon transition (src/lib.rs: cost 100) / arc into the virtual end state (cost 0)
    if (ao < a.narc_cap) a.arcs[ao] = LatArc{kind == 1u ? 100.0f : 0.0f, v, 0xFFFFFFFFu};
    else atomicOr(a.overflow, 1u);
    return;
  }
  const uint32_t m = v, c = a.mc[m];
  if (!c) return;
  const uint32_t pk = a.m_pack[m], src = pk & 0xFFFu, dst = (pk >> 12) & 0xFFFu, order = (pk >> 24) & 0x7Fu, lat = a.m_lat[m];
  const uint32_t so = a.sym_off[m], s_base = a.sym_off[a.st_m0[lat]], mloc = m - a.st_m0[lat];
  if (ao + c > a.narc_cap || so + c > a.nsym_cap) { atomicOr(a.overflow, 1u); return; }
  uint32_t sq;
  const uint32_t rows = op_rows_of(a, order, a.m_q[m], &sq);
  if (!rows) {  // out of vocabulary: one symbol without a vocabulary id (unigrams only: k_op_counts)
    a.arcs[ao] = LatArc{(float)order + 1.0f, src, so - s_base};
    a.syms[so] = LatSym{0u, dst - 1u};
    a.refs[so] = make_uint2(mloc, 0xFFFFFFFFu);
    return;
  }
  const bool uni = order == 1u;
  const DevRow* rr = (uni ? a.rows_u : a.rows_h) + (uni ? a.soff_u : a.soff_h)[sq];
  for (uint32_t vi = 0; vi < c; ++vi) {
    const DevRow d = rr[vi];
    const float cost = (float)order + (1.0f - (float)op_vr_score(d, a.freq_weight));
    a.arcs[ao + vi] = LatArc{cost, src, so + vi - s_base};
    a.syms[so + vi] = LatSym{d.vocab_id, dst - 1u};
    a.refs[so + vi] = make_uint2(mloc, vi);
  }
}
__global__ __launch_bounds__(256) void k_op_inoff(OpArgs a) {
  const uint32_t e = blockIdx.x * 256 + threadIdx.x;
  if (e >= a.nin) return;
  const uint32_t lat = a.e_lat[e];
  a.in_off[e] = a.arc_off[a.e_g0[e]] - a.arc_off[a.e_g0[a.st_e0[lat]]];
}
__global__ __launch_bounds__(256) void k_op_stfix(OpArgs a) {
  const uint32_t li = blockIdx.x * 256 + threadIdx.x;
  if (li >= a.nst) return;
  a.st[li].arc0 = a.arc_off[a.e_g0[a.st_e0[li]]];
  a.st[li].sym0 = a.sym_off[a.st_m0[li]];
}
// the chosen symbols of a lattice -> (match, variant) and the match's row count
__global__ __launch_bounds__(256) void k_op_emit_count(OpArgs a) {
  const uint32_t li = blockIdx.x * 256 + threadIdx.x;
  if (li >= a.nst) return;
  const LatStretch S = a.st[li];
  const uint32_t n = a.out_n[li];
  const uint32_t m0 = a.st_m0[li];
  for (uint32_t j = 0; j < S.nstates; ++j) {  // the lattice's out slots (a path has at most one symbol per state it enters)
    uint32_t cnt = 0, ml = 0xFFFFFFFFu, sel = 0xFFFFFFFFu;
    if (n != 0xFFFFFFFFu && j < n) {
      const uint2 rf = a.refs[S.sym0 + a.out_syms[S.out0 + j]];
      ml = rf.x; sel = rf.y;
      const uint32_t m = m0 + rf.x;
      uint32_t sq;
      cnt = op_rows_of(a, (a.m_pack[m] >> 24) & 0x7Fu, a.m_q[m], &sq);
    }
    a.e_match[S.out0 + j] = ml;
    a.e_sel[S.out0 + j] = sel;
    a.e_cnt[S.out0 + j] = cnt;
  }
}
__global__ __launch_bounds__(256) void k_op_emit_rows(OpArgs a) {
  const uint32_t li = blockIdx.x * 256 + threadIdx.x;
  if (li >= a.nst) return;
  const LatStretch S = a.st[li];
  const uint32_t n = a.out_n[li];
  if (n == 0xFFFFFFFFu) return;
  const uint32_t m0 = a.st_m0[li];
  for (uint32_t j = 0; j < n && j < S.nstates; ++j) {
    const uint32_t m = m0 + a.e_match[S.out0 + j], order = (a.m_pack[m] >> 24) & 0x7Fu;
    uint32_t sq;
    const uint32_t cnt = op_rows_of(a, order, a.m_q[m], &sq);
    if (!cnt) continue;
    const bool uni = order == 1u;
    const DevRow* rr = (uni ? a.rows_u : a.rows_h) + (uni ? a.soff_u : a.soff_h)[sq];
    anx_result* out = a.e_rows + a.e_row0[S.out0 + j];
    for (uint32_t i = 0; i < cnt; ++i) {
      const DevRow d = rr[i];
      anx_result r;
      r.vocab_id = d.vocab_id;
      r.dist_score = d.dist_score;
      r.freq_score = d.freq_score;
      r.via = d.via == 0xFFFFFFFFu ? ANX_NO_VIA : (uint64_t)d.via;
      out[i] = r;
    }
  }
}
// higher-order queries whose segment is redundant: their bit planes are cleared before the batch runs, so the scan finds nothing for
// them (a clean text has few misspelt words: two thirds of its higher-order segments would be looked up for nothing)
__global__ __launch_bounds__(256) void k_op_skip(OpArgs a, uint32_t* __restrict__ q_bits_h) {
  const uint32_t m = blockIdx.x * 256 + threadIdx.x;
  if (m >= a.nmatch) return;
  const uint32_t order = (a.m_pack[m] >> 24) & 0x7Fu;
  if (order <= 1u) return;
  for (uint32_t u = a.m_u0[m]; u < a.m_u1[m]; ++u) {
    uint32_t sq;
    const uint32_t n = op_rows_of(a, 1u, a.m_q[u], &sq);
    if (n == 0u || a.rows_u[a.soff_u[sq]].dist_score < 1.0) return;  // not redundant
  }
  const uint32_t q = a.m_q[m];
  if (q == 0xFFFFFFFFu || q >= a.nin_h) return;
  const uint32_t sq = a.inv_h[q];
  if (sq == 0xFFFFFFFFu) return;
#pragma unroll
  for (int p = 0; p < NBITPLANES; ++p) q_bits_h[(size_t)sq * NBITPLANES + p] = 0u;
}

namespace {
template <typename T>
int op_scan(const T* in, T* out, size_t n, hipStream_t st, std::vector<void*>& owned, std::string& err) {
  size_t bytes = 0;
  HIP_TRY(rocprim::exclusive_scan(nullptr, bytes, in, out, T(0), n, rocprim::plus<T>(), st));
  void* tmp = nullptr;
  HIP_TRY(pool_malloc(&tmp, bytes + 16));
  owned.push_back(tmp);
  HIP_TRY(rocprim::exclusive_scan(tmp, bytes, in, out, T(0), n, rocprim::plus<T>(), st));
  return ANX_OK;
}
}  // namespace

struct OnePassState {  // between search_onepass_prepare and search_onepass_finish
  hipStream_t st = nullptr;
  int device = 0;
  std::vector<void*> owned;
  OpArgs a{};
  uint32_t* d_inv_h = nullptr;
  ~OnePassState() {
    if (st) (void)hipStreamSynchronize(st);
    for (void* q : owned) pool_free(q);
    if (st) encoder_stream_release(device, st);
  }
};
void search_onepass_free(OnePassState* s) { delete s; }

// Phase 1 (after the unigram batch bu has run, before the higher-order batch bh runs): uploads the part's tables, maps the unigram
// batch, and clears the bit planes of the redundant higher-order queries of bh (already encoded).  bh may be nullptr (max_ngram 1).
int search_onepass_prepare(const DeviceLexicon* dl, const Batch* bu, Batch* bh, const OnePassIn& in, const anx_search_params& p, OnePassState** out, std::string& err) {
  *out = nullptr;
  if (!dl || !bu || !bu->ran) { err = "one-pass search: the unigram batch has not run"; return ANX_EINVAL; }
  HIP_TRY(hipSetDevice(dl->device));
  std::unique_ptr<OnePassState> S(new OnePassState());
  S->device = dl->device;
  S->st = encoder_stream_acquire(dl->device);
  hipStream_t st = S->st;
  OpArgs& a = S->a;
  int rc;
  auto dalloc_ = [&](void** p_, size_t bytes) -> int { HIP_TRY(pool_malloc(p_, std::max<size_t>(bytes, 16))); S->owned.push_back(*p_); return ANX_OK; };
  auto up = [&](const uint32_t** dst, const uint32_t* src, size_t n) -> int {
    void* q = nullptr;
    int r = dalloc_(&q, n * 4);
    if (r) return r;
    if (n) HIP_TRY(hipMemcpyAsync(q, src, n * 4, hipMemcpyHostToDevice, st));
    *dst = static_cast<const uint32_t*>(q);
    return ANX_OK;
  };
  a.nmatch = (uint32_t)in.nmatch; a.ngroup = (uint32_t)in.ngroup; a.nin = (uint32_t)in.nin; a.nst = (uint32_t)in.nst;
  if ((rc = up(&a.m_q, in.m_q, in.nmatch)) || (rc = up(&a.m_u0, in.m_u0, in.nmatch)) || (rc = up(&a.m_u1, in.m_u1, in.nmatch)) ||
      (rc = up(&a.m_pack, in.m_pack, in.nmatch)) || (rc = up(&a.m_lat, in.m_lat, in.nmatch)) || (rc = up(&a.g_ref, in.g_ref, in.ngroup)) ||
      (rc = up(&a.e_g0, in.e_g0, in.nin)) || (rc = up(&a.e_lat, in.e_lat, in.nin)) || (rc = up(&a.st_m0, in.st_m0, in.nst)) || (rc = up(&a.st_e0, in.st_e0, in.nst)))
    return rc;
  a.freq_weight = p.base.freq_weight;
  // input index -> sorted query of both batches
  uint32_t *inv_u = nullptr, *inv_h = nullptr;
  a.nin_u = (uint32_t)bu->n_input;
  a.nin_h = bh ? (uint32_t)bh->n_input : 0u;
  if ((rc = dalloc_((void**)&inv_u, (size_t)a.nin_u * 4)) || (rc = dalloc_((void**)&inv_h, (size_t)a.nin_h * 4))) return rc;
  HIP_TRY(hipMemsetAsync(inv_u, 0xFF, std::max<size_t>((size_t)a.nin_u * 4, 4), st));
  HIP_TRY(hipMemsetAsync(inv_h, 0xFF, std::max<size_t>((size_t)a.nin_h * 4, 4), st));
  if (bu->nq) hipLaunchKernelGGL(k_op_inv, dim3(((uint32_t)bu->nq + 255) / 256), dim3(256), 0, st, (uint32_t)bu->nq, bu->q_orig, inv_u);
  if (bh && bh->nq) hipLaunchKernelGGL(k_op_inv, dim3(((uint32_t)bh->nq + 255) / 256), dim3(256), 0, st, (uint32_t)bh->nq, bh->q_orig, inv_h);
  a.inv_u = inv_u; a.inv_h = inv_h;
  S->d_inv_h = inv_h;
  a.soff_u = bu->soff; a.cnt_u = bu->r_count; a.rows_u = bu->r_rows;
  if (bh && bh->nq && a.nmatch) {  // the redundant higher-order queries find nothing
    hipLaunchKernelGGL(k_op_skip, dim3((a.nmatch + 255) / 256), dim3(256), 0, st, a, bh->q_bits);
    HIP_TRY(hipStreamSynchronize(st));  // the batch runs on another stream
  }
  HIP_TRY(hipGetLastError());
  *out = S.release();
  return ANX_OK;
}

// Phase 2 (both batches have run): builds the lattices, decodes them and brings the matches of the chosen paths back.
int search_onepass_finish(const HostModel& m, const DeviceLexicon* dl, OnePassState* S, const Batch* bu, const Batch* bh, OnePassIn& in, const anx_search_params& p,
                          OnePassOut& out, std::string& err) {
  HIP_TRY(hipSetDevice(dl->device));
  hipStream_t st = S->st;
  OpArgs& a = S->a;
  int rc = lm_ensure(m, dl, err);
  if (rc) return rc;
  const bool timing = switches().search_timing != 0;  // (the laps synchronise: timing runs only)
  auto tnow = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
  double t_prev = timing ? tnow() : 0.0;
  auto lap = [&](const char* what) { if (timing) { (void)hipStreamSynchronize(st); const double t = tnow(); fprintf(stderr, "[anx search]     one pass / device: %-22s %8.2f ms\n", what, (t - t_prev) * 1e3); t_prev = t; } };
  auto dalloc_ = [&](void** p_, size_t bytes) -> int { HIP_TRY(pool_malloc(p_, std::max<size_t>(bytes, 16))); S->owned.push_back(*p_); return ANX_OK; };
  if (bh) {
    if (!bh->ran) { err = "one-pass search: the higher-order batch has not run"; return ANX_EINVAL; }
    a.soff_h = bh->soff; a.cnt_h = bh->r_count; a.rows_h = bh->r_rows;
  } else { a.soff_h = a.soff_u; a.cnt_h = a.cnt_u; a.rows_h = a.rows_u; }
  const size_t nres = bu->n_results + (bh ? bh->n_results : 0);
  size_t n_uni = 0;
  for (size_t i = 0; i < in.nmatch; ++i) n_uni += ((in.m_pack[i] >> 24) & 0x7Fu) == 1u;
  const size_t nsym_cap = nres + n_uni + 16, narc_cap = nsym_cap + (in.ngroup - in.nmatch) + 16;
  if (narc_cap >= ((size_t)1 << 32)) { err = "more than 2^32 lattice arcs in one part"; return ANX_ELIMIT; }
  a.narc_cap = (uint32_t)narc_cap; a.nsym_cap = (uint32_t)nsym_cap;
  uint32_t *d_outn = nullptr, *d_outs = nullptr, *d_boff = nullptr;
  int32_t* d_btok = nullptr;
  if ((rc = dalloc_((void**)&a.mc, ((size_t)a.nmatch + 1) * 4)) || (rc = dalloc_((void**)&a.gc, ((size_t)a.ngroup + 1) * 4)) || (rc = dalloc_((void**)&a.mflag, (size_t)a.nmatch + 4)) ||
      (rc = dalloc_((void**)&a.sym_off, ((size_t)a.nmatch + 1) * 4)) || (rc = dalloc_((void**)&a.arc_off, ((size_t)a.ngroup + 1) * 4)) ||
      (rc = dalloc_((void**)&a.st, (size_t)a.nst * sizeof(LatStretch))) || (rc = dalloc_((void**)&a.in_off, (size_t)a.nin * 4)) ||
      (rc = dalloc_((void**)&a.arcs, narc_cap * sizeof(LatArc))) || (rc = dalloc_((void**)&a.syms, nsym_cap * sizeof(LatSym))) ||
      (rc = dalloc_((void**)&a.refs, nsym_cap * sizeof(uint2))) || (rc = dalloc_((void**)&a.overflow, 16)) ||
      (rc = dalloc_((void**)&d_outn, (size_t)a.nst * 4)) || (rc = dalloc_((void**)&d_outs, in.out_total * 4)) ||
      (rc = dalloc_((void**)&d_boff, in.nboff * 4)) || (rc = dalloc_((void**)&d_btok, in.nbtok * 4)))
    return rc;
  HIP_TRY(hipMemsetAsync(a.overflow, 0, 16, st));
  HIP_TRY(hipMemsetAsync(a.mc + a.nmatch, 0, 4, st));
  HIP_TRY(hipMemsetAsync(a.gc + a.ngroup, 0, 4, st));
  if (in.nboff) HIP_TRY(hipMemcpyAsync(d_boff, in.btok_off, in.nboff * 4, hipMemcpyHostToDevice, st));
  if (in.nbtok) HIP_TRY(hipMemcpyAsync(d_btok, in.btok, in.nbtok * 4, hipMemcpyHostToDevice, st));
  if (a.nmatch) hipLaunchKernelGGL(k_op_counts, dim3((a.nmatch + 255) / 256), dim3(256), 0, st, a);
  if (a.ngroup) hipLaunchKernelGGL(k_op_gcount, dim3((a.ngroup + 255) / 256), dim3(256), 0, st, a);
  if ((rc = op_scan(a.mc, a.sym_off, (size_t)a.nmatch + 1, st, S->owned, err)) || (rc = op_scan(a.gc, a.arc_off, (size_t)a.ngroup + 1, st, S->owned, err))) return rc;
  if (a.ngroup) hipLaunchKernelGGL(k_op_fill, dim3((a.ngroup + 255) / 256), dim3(256), 0, st, a);
  if (a.nin) hipLaunchKernelGGL(k_op_inoff, dim3((a.nin + 255) / 256), dim3(256), 0, st, a);
  lap("build (counts, scans, fill)");
  a.out_n = d_outn; a.out_syms = d_outs; a.out_total = (uint32_t)in.out_total;
  std::vector<LatStretch> hst(in.st, in.st + in.nst);
  std::vector<uint32_t> maxdeg(in.maxdeg, in.maxdeg + in.nst);
  const uint32_t K = std::max<uint32_t>(1u, p.max_seq);
  out.handed_back = false;
  if (a.nst) {
    if (K > 4096u) { out.handed_back = true; return ANX_OK; }
    HIP_TRY(hipMemsetAsync(d_outn, 0, (size_t)a.nst * 4, st));
    auto fix = [&]() -> int {
      hipLaunchKernelGGL(k_op_stfix, dim3((a.nst + 255) / 256), dim3(256), 0, st, a);
      return ANX_OK;
    };
    if ((rc = lattice_launch(m, dl, hst, maxdeg, a.st, a.in_off, a.arcs, a.syms, nsym_cap, d_boff, d_btok, d_outn, d_outs, p, st, fix, S->owned, err))) return rc;
  }
  lap("k_lattice");
  // the matches on the chosen paths and their rows
  if ((rc = dalloc_((void**)&a.e_match, (in.out_total + 1) * 4)) || (rc = dalloc_((void**)&a.e_sel, (in.out_total + 1) * 4)) ||
      (rc = dalloc_((void**)&a.e_cnt, (in.out_total + 1) * 4)) || (rc = dalloc_((void**)&a.e_row0, (in.out_total + 1) * 4)) ||
      (rc = dalloc_((void**)&a.e_rows, std::max<size_t>(nres, 1) * sizeof(anx_result))))
    return rc;
  HIP_TRY(hipMemsetAsync(a.e_cnt, 0, (in.out_total + 1) * 4, st));
  if (a.nst) hipLaunchKernelGGL(k_op_emit_count, dim3((a.nst + 255) / 256), dim3(256), 0, st, a);
  if ((rc = op_scan(a.e_cnt, a.e_row0, in.out_total + 1, st, S->owned, err))) return rc;
  if (a.nst) hipLaunchKernelGGL(k_op_emit_rows, dim3((a.nst + 255) / 256), dim3(256), 0, st, a);
  // download: per lattice the symbols of its path; per out slot (match, variant, first row); then the rows themselves
  const size_t o_n = 0, o_m = o_n + ((size_t)a.nst * 4 + 63) / 64 * 64, o_s = o_m + ((in.out_total + 1) * 4 + 63) / 64 * 64, o_r = o_s + ((in.out_total + 1) * 4 + 63) / 64 * 64,
               o_end = o_r + ((in.out_total + 1) * 4 + 63) / 64 * 64 + 64;
  char* blk = static_cast<char*>(host_result_alloc(o_end));
  if (!blk) { err = "out of memory"; return ANX_EINVAL; }
  out.block = blk;
  out.out_n = reinterpret_cast<uint32_t*>(blk + o_n); out.e_match = reinterpret_cast<uint32_t*>(blk + o_m);
  out.e_sel = reinterpret_cast<uint32_t*>(blk + o_s); out.e_row0 = reinterpret_cast<uint32_t*>(blk + o_r);
  uint32_t* h_over = reinterpret_cast<uint32_t*>(blk + o_end - 64);
  if (a.nst) HIP_TRY(hipMemcpyAsync(out.out_n, d_outn, (size_t)a.nst * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(out.e_match, a.e_match, (in.out_total + 1) * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(out.e_sel, a.e_sel, (in.out_total + 1) * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(out.e_row0, a.e_row0, (in.out_total + 1) * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(h_over, a.overflow, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  lap("emit + small downloads");
  if (*h_over) { err = "internal error: a lattice outgrew the bounds of its part"; return ANX_ELIMIT; }
  for (uint32_t li = 0; li < a.nst; ++li)
    if (out.out_n[li] == 0xFFFFFFFFu) { out.handed_back = true; return ANX_OK; }  // a lattice beyond the kernel's limits: the caller takes the classic path
  const size_t total_rows = out.e_row0[in.out_total];
  if (total_rows > nres) { err = "internal error: more emitted rows than ranked rows"; return ANX_ELIMIT; }
  out.n_rows = total_rows;
  out.rows = static_cast<anx_result*>(host_result_alloc(std::max<size_t>(total_rows, 1) * sizeof(anx_result)));
  if (!out.rows) { err = "out of memory"; return ANX_EINVAL; }
  // the rows are on their way when this returns: the caller builds its matches from the small arrays meanwhile and calls
  // search_onepass_rows_wait before it reads a row
  if (total_rows) HIP_TRY(hipMemcpyAsync(out.rows, a.e_rows, total_rows * sizeof(anx_result), hipMemcpyDeviceToHost, st));
  if (timing) { HIP_TRY(hipStreamSynchronize(st)); lap("rows download"); }
  return ANX_OK;
}
int search_onepass_rows_wait(OnePassState* S, std::string& err) {
  HIP_TRY(hipStreamSynchronize(S->st));
  return ANX_OK;
}

}  // namespace anx
