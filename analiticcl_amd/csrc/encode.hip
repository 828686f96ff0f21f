// encode.hip -- device-side query encoder of the anx engine (gfx950 / CDNA4).
//
// Replaces, for a whole batch at once, what the reference does per input at the top of find_variants()
// (/root/reference/src/lib.rs:972-1012): normalize_to_alphabet / the count vector behind anahash (src/anahash.rs:16-80),
// the threshold clamps (src/lib.rs:982-1012), plus what the scan kernels want on top of it: thermometer planes, group
// signature, the (scan kernel, length, signature, kind) order, tiles in longest-processing-time order, and (StopAtExactMatch,
// src/lib.rs:1164-1173) the exact anagram class of every query.  The host only hands over the input bytes and their offsets.
//   k_enc_strings : one lane per input string: greedy first-match alphabet walk (class order, member order; multi-character
//                   members consume their characters), codes, count vector, planes, signature, clamps, first-char case
//   sort          : one packed key (scan kernel, length, signature, kind) -- rocPRIM device radix sort (plumbing)
//   k_enc_gather  : sorted position -> query records, rows, planes, count vectors, exact class
//   k_tile_*      : segment heads -> tiles of <= SCAN_TQ queries -> LPT order (one more radix sort)
// The threaded host encoder in engine.hip (ANX_ENCODE=host) produces the same arrays and is kept as the A/B reference.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

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

struct EncArgs {
  const uint8_t* blob;      // input bytes; string i = [off[i], off[i+1] - 1), one separator byte behind it
  const uint32_t* off;      // [n + 1]
  uint32_t n;
  DevAlphabet al;
  int A;                    // alphabet classes; symbol slots 0..A (slot A = unknown), norm code of unknown = A + 1
  int NP;                   // count-vector dwords
  int bits_ok;              // nsym <= 32: thermometer planes usable
  anx_threshold kth, dth;
  uint8_t* codes;           // norm codes of string i at codes[code_off(off[i], i) ..]: dword-aligned, so that they move as dwords
  uint32_t* meta;           // [n] len | k<<8 | d<<16 | first_is_lower<<24 ; 0 = not encodable
  uint32_t* bits;           // [n][NBITPLANES]
  unsigned long long* sig;  // [n]
  uint32_t* kind;           // [n]
  uint32_t* cv;             // [n][NP]; zero-initialised when !bits_ok; with planes only rows of kind 0 are written
  unsigned long long* key;  // [n] sort key: (scan kernel, length) << (3 + 5 ngroups) | signature, 5 bits per group, << 3 | kind; the bit above = not
                            // encodable (sorts last).  The order has to bring equal (scan kernel, length, signature) together, kinds
                            // ascending, and keeps neighbouring signatures together (tiles of equal cost run side by side: their
                            // table probes share cache lines -- a hashed signature cost 0.1 ms per step in the scan).  Group counts
                            // saturate at 31: signatures that differ only beyond that interleave and split into more tiles, nothing else.
  int ngroups;              // signature groups in use (1..8)
  uint32_t* blk;            // [blocks][3] per block: max len, max d, encodable inputs (k_enc_totals -> ctr[0..2])
  int dbg;                  // ANX_ENC_DBG (debug builds, timing only, results WRONG): 1 no count-vector writes, 2 no code stores, 4 no walk, 8 no record stores
  int zero_cv;              // !bits_ok: every lane clears its own count-vector row first (the small path: no memset launch before the kernel)
  uint32_t stage_off[17];   // k_enc_strings<true> (<= 4096 strings): off[256 b] for every block b and off[n]: the block's byte range without a read of the (host) offsets
};

// Where the codes of string i start: the bytes of string i and its separator are >= symbols + 1, and rounding every start up to a
// dword plus one dword per string keeps the regions disjoint (start(i+1) - start(i) is a multiple of 4 that is >= symbols + 2).
// The buffer holds blob bytes + 4 n + 16.
__device__ inline uint32_t code_off(uint32_t off_i, uint32_t i) { return ((off_i + 3u) & ~3u) + 4u * i; }
// 16 bytes of the blob from any byte position: aligned dwords + v_alignbyte; only dwords that hold one of the `left` bytes are read
__device__ inline void load_window(const uint8_t* __restrict__ blob, uint32_t pos, uint32_t left, uint32_t (&r)[4]) {
  const uint32_t sh = pos & 3u;
  const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(blob + (pos - sh));
  uint32_t dw[5];
#pragma unroll
  for (uint32_t x = 0; x < 5u; ++x) dw[x] = 4u * x < sh + left ? src[x] : 0u;
#pragma unroll
  for (uint32_t x = 0; x < 4u; ++x) r[x] = __builtin_amdgcn_alignbyte(dw[x + 1], dw[x], sh);
}
// dword w of the count vector (4 symbol slots, a byte each) of a string whose counts are all <= 4, from its thermometer planes
__device__ inline uint32_t cv_word_of_planes(const uint4 pl, uint32_t w) {
  uint32_t word = 0;
  if (w < 8u) {
#pragma unroll
    for (uint32_t y = 0; y < 4u; ++y) {
      const uint32_t sl = 4u * w + y;
      word |= (((pl.x >> sl) & 1u) + ((pl.y >> sl) & 1u) + ((pl.z >> sl) & 1u) + ((pl.w >> sl) & 1u)) << (8u * y);
    }
  }
  return word;
}
__device__ inline int dev_u8len(uint32_t c) { return c < 0x80u ? 1 : (c >> 5) == 0x6u ? 2 : (c >> 4) == 0xEu ? 3 : (c >> 3) == 0x1Eu ? 4 : 1; }

__device__ inline int dev_clamp_threshold(const anx_threshold& t, int len, int absolute_max) {  // host_model.cpp clamp_threshold
  if (t.kind == ANX_RATIO || t.kind == ANX_RATIO_WITH_LIMIT) {
    const float v = floorf((float)len * t.ratio);
    const int x = v < 0.0f ? 0 : v > 255.0f ? 255 : (int)v;
    const int lim = t.kind == ANX_RATIO ? absolute_max : (int)t.value;
    return x < lim ? x : lim;
  }
  const int half = len / 2 < 255 ? len / 2 : 255;
  return (int)t.value < half ? (int)t.value : half;
}

// STAGE (the small call): the blob is pinned HOST memory; the block first copies the bytes of its 256 strings into LDS with coalesced
// 16-byte loads (one burst over PCIe instead of a dword read per lane and window) and the lanes walk them there.  Strings of at most
// ENC_STAGE_BYTES bytes each.
constexpr uint32_t ENC_STAGE_BYTES = 64;
template <bool STAGE>
__global__ __launch_bounds__(256) void k_enc_strings(EncArgs a) {
  // the per-byte tables of the walk in LDS: one lane walks one string, every step is a chain of dependent loads -- from LDS they
  // cost ~64 cycles instead of a trip to L1 / L2
  __shared__ int16_t s_fast[256];
  __shared__ uint8_t s_group[176];   // signature group per symbol slot (<= 168 slots)
  for (uint32_t x = threadIdx.x; x < 256u; x += 256u) s_fast[x] = a.al.fast[x];
  for (uint32_t x = threadIdx.x; x < (uint32_t)a.NP * 4u && x < 176u; x += 256u) s_group[x] = a.al.sym_group[x];
  __syncthreads();
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  uint32_t len = 0, d = 0, ok = 0;
  __shared__ uint4 s_stage[STAGE ? (256 * (ENC_STAGE_BYTES + 1) + 48) / 16 : 1];
  uint32_t stage0 = 0;
  // (STAGE: the lane's own offsets are requested first, so that they travel over PCIe together with the block's bytes, not after them)
  const uint32_t off_i = i < a.n ? a.off[i] : 0u, off_i1 = i < a.n ? a.off[i + 1] : 1u;
  if (STAGE) {
    const uint32_t b0 = a.stage_off[blockIdx.x] & ~15u, b1 = a.stage_off[blockIdx.x + 1u] + 16u;   // (the window of the last string may read 16 bytes past it)
    for (uint32_t x = threadIdx.x; x < (b1 - b0 + 15u) / 16u; x += 256u) s_stage[x] = reinterpret_cast<const uint4*>(a.blob + b0)[x];
    stage0 = b0;
    __syncthreads();
  }
  if (i < a.n) {
    const uint32_t begin = off_i, end = off_i1 - 1u;
    const uint8_t* __restrict__ s = STAGE ? reinterpret_cast<const uint8_t*>(s_stage) - stage0 : a.blob;
    uint32_t n = 0, skip = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, over = 0;
    unsigned long long sig = 0;
    uint8_t* cvb = reinterpret_cast<uint8_t*>(a.cv) + (size_t)i * (size_t)a.NP * 4u;
    if (!a.bits_ok && a.zero_cv)
      for (uint32_t w = 0; w < (uint32_t)a.NP; ++w) a.cv[(size_t)i * (size_t)a.NP + w] = 0u;
    uint32_t* __restrict__ cw = reinterpret_cast<uint32_t*>(a.codes + code_off(begin, i));
    uint32_t cword = 0;
    bool too_long = false;
    // The walk reads the string through a 16-byte register window (one burst of loads per 16 bytes instead of a dependent byte load
    // per character) and writes the codes four at a time.
    uint32_t r[4], wend = begin, first4 = 0;
    for (uint32_t pos = begin; pos < end && !(ANX_DBG(a.dbg) & 4);) {
      if (pos >= wend) {
        load_window(s, pos, end - pos, r);
        if (pos == begin) first4 = r[0];
        wend = pos + 16u;
      }
      const uint32_t b = r[0] & 0xFFu;
      const uint32_t here = pos;
      const uint32_t l = (uint32_t)dev_u8len(b);
      pos += l;
      {  // the window moves on by l bytes
        const unsigned long long lo = ((unsigned long long)r[1] << 32 | r[0]) >> (8u * l), mi = ((unsigned long long)r[2] << 32 | r[1]) >> (8u * l),
                                 hi = ((unsigned long long)r[3] << 32 | r[2]) >> (8u * l);
        r[0] = (uint32_t)lo; r[1] = (uint32_t)mi; r[2] = (uint32_t)hi; r[3] = l == 4u ? 0u : r[3] >> (8u * l);
      }
      if (skip) { --skip; continue; }
      int hit = s_fast[b];
      if (hit == -2) {  // members starting with this byte, in (class, member) file order: the first that matches wins
        hit = -1;
        for (uint32_t c = a.al.coff[b]; c < a.al.coff[b + 1]; ++c) {
          const uint4 cd = a.al.cand[c];  // {class, characters, bytes, offset into the byte pool}
          if (here + cd.z > end) continue;
          bool eq = true;
          for (uint32_t x = 0; x < cd.z; ++x) eq = eq && s[here + x] == a.al.bytes[cd.w + x];
          if (eq) { hit = (int)cd.x; skip = cd.y - 1u; break; }
        }
      }
      if (n >= (uint32_t)kMaxSymbols) { too_long = true; break; }
      const uint32_t slot = hit >= 0 ? (uint32_t)hit : (uint32_t)a.A;               // src/anahash.rs:42
      cword |= (uint32_t)(hit >= 0 ? hit : a.A + 1) << (8u * (n & 3u));             // src/anahash.rs:76
      if ((n & 3u) == 3u) { if (!(ANX_DBG(a.dbg) & 2)) cw[n >> 2] = cword; cword = 0; }
      ++n;
      if (a.bits_ok) {  // bit-sliced saturating counter: plane t has the bit iff count > t
        const uint32_t bit = 1u << slot;
        over |= p4 & bit;
        p4 |= p3 & bit; p3 |= p2 & bit; p2 |= p1 & bit; p1 |= bit;
      } else {
        cvb[slot] = (uint8_t)(cvb[slot] + 1u);
      }
      sig += 1ull << (8u * s_group[slot < 176u ? slot : 175u]);
    }
    if ((n & 3u) && !too_long && !(ANX_DBG(a.dbg) & 2)) cw[n >> 2] = cword;  // the last, partial dword of codes
    // The count vector: with the thermometer planes the walk above does not touch it -- a read-modify-write of a global byte per
    // symbol -- and only the rare string with a symbol five times or more (kind 0: the count-vector scan) gets one at all, from a
    // second walk over the codes just written; k_enc_gather rebuilds the others' from the planes where it needs them.
    if (a.bits_ok && !too_long && over) {
      for (uint32_t w = 0; w < (uint32_t)a.NP; ++w) a.cv[(size_t)i * (size_t)a.NP + w] = 0u;
      for (uint32_t x = 0; x < n; ++x) {
        const uint32_t code = (x >> 2) == (n >> 2) ? (cword >> (8u * (x & 3u))) & 0xFFu : (cw[x >> 2] >> (8u * (x & 3u))) & 0xFFu;
        const uint32_t slot = code == (uint32_t)a.A + 1u ? (uint32_t)a.A : code;
        cvb[slot] = (uint8_t)(cvb[slot] + 1u);
      }
    }
    uint32_t meta = 0, kind = 0;
    unsigned long long key = 1ull << (12 + 5 * a.ngroups);
    if (!too_long && n > 0) {
      int fl;
      {  // char::is_lowercase on the first character of the ORIGINAL string (src/lib.rs:1367-1377)
        const uint32_t avail = end - begin;  // first4: the first four bytes of the string (zero behind its end)
        int l = dev_u8len(first4 & 0xFFu);
        if ((uint32_t)l > avail) l = 1;
        uint32_t cp = first4 & 0xFFu;
        const uint32_t c1 = (first4 >> 8) & 0x3Fu, c2 = (first4 >> 16) & 0x3Fu, c3 = (first4 >> 24) & 0x3Fu;
        if (l == 2) cp = ((cp & 0x1Fu) << 6) | c1;
        else if (l == 3) cp = ((cp & 0x0Fu) << 12) | (c1 << 6) | c2;
        else if (l == 4) cp = ((cp & 0x07u) << 18) | (c1 << 12) | (c2 << 6) | c3;
        int lo = 0, hi = (int)a.al.nlower - 1;
        fl = 0;
        if (cp < 0x80u) { fl = cp >= 'a' && cp <= 'z'; hi = -1; }  // the Lowercase property below U+0080 is a-z: no table walk
        while (lo <= hi) {
          const int mid = (lo + hi) >> 1;
          const uint2 r = a.al.lower[mid];
          if (cp < r.x) hi = mid - 1;
          else if (cp > r.y) lo = mid + 1;
          else { fl = 1; break; }
        }
      }
      const int k = dev_clamp_threshold(a.kth, (int)n, kMaxAnagramDistance);
      const int dd = dev_clamp_threshold(a.dth, (int)n, kMaxEditDistance);
      meta = n | ((uint32_t)k << 8) | ((uint32_t)dd << 16) | ((uint32_t)fl << 24);
      kind = (a.bits_ok && !over) ? (p4 ? 4u : p3 ? 3u : p2 ? 2u : 1u) : 0u;
      unsigned long long sc = 0;  // group ngroups - 1 is the most significant byte in use: numeric order of the signature
      for (int g = a.ngroups - 1; g >= 0; --g) sc = sc << 5 | min((uint32_t)(sig >> (8 * g)) & 0xFFu, 31u);
      key = ((unsigned long long)((kind ? 256u : 0u) + n) << (5 * a.ngroups) | sc) << 3 | kind;
      len = n; d = (uint32_t)dd; ok = 1;
    }
    if (ANX_DBG(a.dbg) & 8) return;
    a.meta[i] = meta;
    a.kind[i] = kind;
    a.key[i] = key;
    a.sig[i] = sig;
    reinterpret_cast<uint4*>(a.bits)[i] = make_uint4(p1, p2, p3, p4);
  }
  // wave maxima / counts -> block -> one row per block, reduced by k_enc_totals.  No atomics: the blocks of all eight XCDs would
  // meet on one word, and same-address atomics across XCDs sustain only ~15 M/s -- three per wave (47 k per million strings) were
  // 0.2 ms of this kernel, one per block still most of what was left.
#pragma unroll
  for (int o = 32; o; o >>= 1) {
    len = max(len, (uint32_t)__shfl_xor((int)len, o));
    d = max(d, (uint32_t)__shfl_xor((int)d, o));
  }
  const uint32_t cnt = (uint32_t)__popcll(__ballot(ok));
  __shared__ uint32_t s_red[3][4];
  if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = len; s_red[1][threadIdx.x >> 6] = d; s_red[2][threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a.blk[3 * blockIdx.x + 0] = max(max(s_red[0][0], s_red[0][1]), max(s_red[0][2], s_red[0][3]));
    a.blk[3 * blockIdx.x + 1] = max(max(s_red[1][0], s_red[1][1]), max(s_red[1][2], s_red[1][3]));
    a.blk[3 * blockIdx.x + 2] = s_red[2][0] + s_red[2][1] + s_red[2][2] + s_red[2][3];
  }
}
// one block: rows of k_enc_strings -> ctr[0] max len, ctr[1] max d, ctr[2] encodable inputs
__global__ __launch_bounds__(256) void k_enc_totals(const uint32_t* __restrict__ blk, uint32_t nblk, uint32_t* __restrict__ ctr) {
  uint32_t len = 0, d = 0, cnt = 0;
  for (uint32_t x = threadIdx.x; x < nblk; x += 256u) { len = max(len, blk[3 * x]); d = max(d, blk[3 * x + 1]); cnt += blk[3 * x + 2]; }
#pragma unroll
  for (int o = 32; o; o >>= 1) {
    len = max(len, (uint32_t)__shfl_xor((int)len, o));
    d = max(d, (uint32_t)__shfl_xor((int)d, o));
    cnt += (uint32_t)__shfl_xor((int)cnt, o);
  }
  __shared__ uint32_t s_red[3][4];
  if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = len; s_red[1][threadIdx.x >> 6] = d; s_red[2][threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    ctr[0] = max(max(s_red[0][0], s_red[0][1]), max(s_red[0][2], s_red[0][3]));
    ctr[1] = max(max(s_red[1][0], s_red[1][1]), max(s_red[1][2], s_red[1][3]));
    ctr[2] = s_red[2][0] + s_red[2][1] + s_red[2][2] + s_red[2][3];
  }
}

__global__ void k_iota(uint32_t* p, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = i;
}
template <typename T>
__global__ void k_gather(const T* __restrict__ src, const uint32_t* __restrict__ perm, T* __restrict__ dst, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[perm[i]];
}

// ---- offsets of NUL-separated strings (anx_batch_encode_packed hands over the buffer as it is) ----------------------------
// The host used to walk the buffer with one memchr per string: 3 ms per million short strings, more than the whole encoder
// on the device.  Two passes of 4096 bytes per block: count the zero bytes, (exclusive scan of the block counts), write
// off[r + 1] = position after the r-th zero byte for r < n.  Zero test per byte, exact (no borrow between bytes):
// a byte of ~(((x & 0x7F..) + 0x7F..) | x | 0x7F..) is 0x80 iff the byte of x is 0.
constexpr uint32_t NUL_BLK = 4096;
__device__ inline uint32_t zero_byte_flags(uint32_t x) {
  return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
}
// the 4 words of thread t of block b (bytes beyond len read as non-zero; the buffer is allocated in 16-byte multiples)
__device__ inline void nul_flags(const uint8_t* __restrict__ blob, uint32_t len, uint32_t base, uint32_t (&f)[4]) {
  f[0] = f[1] = f[2] = f[3] = 0;
  if (base >= len) return;
  const uint4 v = *reinterpret_cast<const uint4*>(blob + base);
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t pos = base + 4u * (uint32_t)i;
    if (pos >= len) break;
    uint32_t x = w[i];
    if (len - pos < 4u) x |= 0xFFFFFFFFu << (8u * (len - pos));
    f[i] = zero_byte_flags(x);
  }
}
__global__ __launch_bounds__(256) void k_nul_count(const uint8_t* __restrict__ blob, uint32_t len, uint32_t* __restrict__ counts,
                                                   uint32_t* __restrict__ total) {
  __shared__ uint32_t s_w[4];
  uint32_t f[4];
  nul_flags(blob, len, blockIdx.x * NUL_BLK + threadIdx.x * 16u, f);
  uint32_t c = (uint32_t)(__popc(f[0]) + __popc(f[1]) + __popc(f[2]) + __popc(f[3]));
#pragma unroll
  for (int o = 32; o; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    counts[blockIdx.x] = t;
    if (t) atomicAdd(total, t);
  }
}
__global__ __launch_bounds__(256) void k_nul_emit(const uint8_t* __restrict__ blob, uint32_t len, const uint32_t* __restrict__ block_first,
                                                  uint32_t n, uint32_t* __restrict__ off) {
  __shared__ uint32_t s_w[4];
  const uint32_t base = blockIdx.x * NUL_BLK + threadIdx.x * 16u, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t f[4];
  nul_flags(blob, len, base, f);
  const uint32_t c = (uint32_t)(__popc(f[0]) + __popc(f[1]) + __popc(f[2]) + __popc(f[3]));
  uint32_t incl = c;  // inclusive prefix over the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
    if ((int)lane >= o) incl += up;
  }
  if (lane == 63) s_w[wid] = incl;
  __syncthreads();
  uint32_t r = block_first[blockIdx.x] + incl - c;  // rank of this thread's first zero byte
  for (uint32_t w = 0; w < wid; ++w) r += s_w[w];
  if (blockIdx.x == 0 && threadIdx.x == 0) off[0] = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    for (uint32_t m = f[i]; m; m &= m - 1u, ++r)
      if (r < n) off[r + 1] = base + 4u * (uint32_t)i + ((uint32_t)__ffs((int)m) >> 3);  // flag bit 8j+7 -> byte j, + 1
}

struct GatherArgs {
  uint32_t nq, qw;
  int NP, want_exact;
  const uint32_t* perm;     // sorted position -> input index
  const uint32_t* off;
  const uint8_t* codes;
  const uint32_t *meta, *bits, *kind, *cv;
  const unsigned long long* sig;
  // outputs (sorted order)
  uint4* q_rec; uint4* q_rows; uint32_t *q_bits, *q_cv, *q_meta, *q_orig, *qexact, *s_kind;
  unsigned long long* s_sig;
  // exact-class lookup (StopAtExactMatch)
  const uint4* sigtab; const uint32_t* siglen_begin; const uint32_t* cls_planes; uint32_t cstride;
};
__device__ inline void enc_gather_body(const GatherArgs& g, uint32_t s) {   // sorted position s (s < g.nq)
  const uint32_t i = g.perm[s];
  const uint32_t meta = g.meta[i], len = meta & 0xFFu;
  const unsigned long long sig = g.sig[i];
  g.q_meta[s] = meta;
  g.q_orig[s] = i;
  g.s_sig[s] = sig;
  static_assert(NBITPLANES == 4, "planes move as one uint4");
  const uint4 planes = reinterpret_cast<const uint4*>(g.bits)[i];
  const uint32_t kind = g.kind[i];
  reinterpret_cast<uint4*>(g.q_bits)[s] = planes;
  g.s_kind[s] = kind;
  if (kind == 0)  // only the count-vector scan reads q_cv (kind 0 sorts first: its rows are the head of the array)
    for (int p = 0; p < g.NP; ++p) g.q_cv[(size_t)s * g.NP + p] = g.cv[(size_t)i * g.NP + p];
  // the codes of a string start at a dword (code_off): only dwords that hold a symbol are fetched, bytes from len on read 0xFE
  const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(g.codes + code_off(g.off[i], i));
  for (uint32_t w = 0; w < g.qw; ++w) {  // rows padded with bytes that equal nothing (0xFE; kernels_score.hpp)
    const uint32_t left = len > w * 16u ? len - w * 16u : 0u;  // symbols of the string from this row on
    uint32_t v[4];
#pragma unroll
    for (uint32_t x = 0; x < 4u; ++x) {
      const uint32_t word = 4u * x < left ? src[w * 4u + x] : 0u;
      const uint32_t valid = left > 4u * x ? left - 4u * x : 0u;
      const uint32_t mask = valid >= 4u ? 0xFFFFFFFFu : (1u << (8u * valid)) - 1u;
      v[x] = (word & mask) | (0xFEFEFEFEu & ~mask);
    }
    const uint4 row = make_uint4(v[0], v[1], v[2], v[3]);
    g.q_rows[(size_t)s * g.qw + w] = row;
    if (w == 0) {
      uint32_t pw[3] = {0u, 0u, 0u};
      {  // symbol planes of the first 16 symbols (kernels_common.hpp symbol_planes16; bytes from len on are padding)
        uint32_t p[6] = {0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
        for (uint32_t i = 0; i < 16u; ++i) {
          const uint32_t c = ((v[i >> 2] >> (8u * (i & 3u))) & 0xFFu) + 1u;
          const uint32_t on = i < len ? 1u : 0u;
#pragma unroll
          for (uint32_t b = 0; b < 6u; ++b) p[b] |= (((c >> b) & on)) << i;
        }
        pw[0] = p[0] | p[1] << 16; pw[1] = p[2] | p[3] << 16; pw[2] = p[4] | p[5] << 16;
      }
      g.q_rec[2 * (size_t)s] = row;
      g.q_rec[2 * (size_t)s + 1] = make_uint4(meta, pw[0], pw[1], pw[2]);
    }
  }
  uint32_t xc = 0xFFFFFFFFu;
  if (g.want_exact) {  // the exact anagram class: the run of this signature in its charcount range, then the count vectors
    int lo = (int)g.siglen_begin[len], hi = (int)g.siglen_begin[len + 1] - 1;
    while (lo <= hi) {
      const int mid = (lo + hi) >> 1;
      const uint4 r = g.sigtab[mid];  // {sig lo, sig hi, first class, classes}
      const unsigned long long v = (unsigned long long)r.x | (unsigned long long)r.y << 32;
      if (sig < v) hi = mid - 1;
      else if (sig > v) lo = mid + 1;
      else {
        for (uint32_t c = r.z; c < r.z + r.w && xc == 0xFFFFFFFFu; ++c) {
          bool eq = true;
          for (int p = 0; p < g.NP; ++p)
            eq = eq && g.cls_planes[(size_t)p * g.cstride + c] == (kind ? cv_word_of_planes(planes, (uint32_t)p) : g.cv[(size_t)i * g.NP + p]);
          if (eq) xc = c;
        }
        break;
      }
    }
  }
  g.qexact[s] = xc;
}
__global__ __launch_bounds__(256) void k_enc_gather(GatherArgs g) {
  const uint32_t s = blockIdx.x * 256 + threadIdx.x;
  if (s < g.nq) enc_gather_body(g, s);
}

// ---- tiles ------------------------------------------------------------------------------------------------------------
struct TileArgs {
  uint32_t nq, tq;
  const uint32_t *q_meta, *s_kind;
  const unsigned long long* s_sig;
  const uint32_t* siglen_begin;
  uint32_t* head;       // [nq] s if s starts a (scan kernel, length, signature) segment else 0  -> (max-scan) segment start
  uint32_t* tcount;     // [nq + 1] tiles started at s -> (exclusive scan) first tile index
  Tile* tiles;          // unsorted
  uint32_t* tkey;       // LPT sort key
  uint32_t* ctr;        // [3] = SAD tiles
  const uint32_t* ball_tab;  // ball_off[13] ++ ball_n[13] (DeviceLexicon::ball_tab)
  int probe;            // tiles may probe the signature hash table (ANX_SCAN_WALK=flat: never)
  const uint4* adj_hash;  // signature adjacency lists (adjacency.h): {sig lo, sig hi, header index + 1, rows}; adj_mask 0 = none
  uint32_t adj_mask;
};
// header index + 1 of the signature's adjacency list (0 = none), *rows = its length
__device__ inline uint32_t adj_lookup(const uint4* tab, uint32_t mask, uint32_t lo, uint32_t hi, uint32_t* rows) {
  if (!mask) return 0u;
  uint32_t h = sig_hash(lo, hi) & mask;
  for (int p = 0; p < 17; ++p) {  // every key within 16 slots of its home
    const uint4 e = tab[h];
    if (!e.z) return 0u;
    if (e.x == lo && e.y == hi) { *rows = e.w; return e.z; }
    h = (h + 1u) & mask;
  }
  return 0u;
}
__device__ inline bool same_segment(const TileArgs& t, uint32_t a, uint32_t b) {
  return (t.s_kind[a] == 0) == (t.s_kind[b] == 0) && (t.q_meta[a] & 0xFFu) == (t.q_meta[b] & 0xFFu) && t.s_sig[a] == t.s_sig[b];
}
__device__ inline void tile_window(const TileArgs& t, uint32_t s, uint32_t& s0, uint32_t& s1, uint32_t& step, uint32_t& nparts,
                                   uint32_t& ball0, uint32_t& balln) {
  const uint32_t meta = t.q_meta[s], lq = meta & 0xFFu, k = (meta >> 8) & 0xFFu;
  const int lo = max(1, (int)lq - (int)k), hi = min(kMaxSymbols, (int)lq + (int)k);
  s0 = t.siglen_begin[lo] & ~63u;
  s1 = (t.siglen_begin[hi + 1] + 63u) & ~63u;
  ball0 = balln = 0;  // hash-table probes with the L1 ball of signature offsets instead of the window walk (engine.hip)
  if (t.probe && k <= 12u && tile_probes(t.ball_tab[13 + k], s1 - s0, t.s_sig[s])) { ball0 = t.ball_tab[k]; balln = t.ball_tab[13 + k]; }
  const uint32_t nsplit = (t.s_kind[s] == 0 && !balln) ? 8u : 1u;  // count-vector tiles: the window split over 8 waves (engine.hip)
  step = (((s1 - s0) + nsplit - 1u) / nsplit + 63u) & ~63u;
  nparts = 0;
  for (uint32_t part = 0; part < nsplit; ++part) {
    const uint32_t a0 = s0 + part * step, a1 = min(s1, a0 + step);
    if (a0 >= a1 && part) break;
    ++nparts;
  }
}
__global__ __launch_bounds__(256) void k_tile_heads(TileArgs t) {
  const uint32_t s = blockIdx.x * 256 + threadIdx.x;
  if (s >= t.nq) return;
  t.head[s] = (s > 0 && !same_segment(t, s, s - 1)) ? s : 0u;
}
__global__ __launch_bounds__(256) void k_tile_count(TileArgs t) {
  const uint32_t s = blockIdx.x * 256 + threadIdx.x;
  if (s > t.nq) return;
  uint32_t c = 0;
  if (s < t.nq && (s - t.head[s]) % t.tq == 0) {  // head[] holds the segment start after the max-scan
    uint32_t s0, s1, step, nparts, ball0, balln;
    tile_window(t, s, s0, s1, step, nparts, ball0, balln);
    c = nparts;
  }
  t.tcount[s] = c;
}
__global__ __launch_bounds__(256) void k_tile_emit(TileArgs t) {
  const uint32_t s = blockIdx.x * 256 + threadIdx.x;
  if (s >= t.nq || (s - t.head[s]) % t.tq != 0) return;
  // queries of the tile = up to tq positions of the segment from s on; head[] (segment start per position) is monotone and the
  // kinds inside a segment ascend, so the tile's end and its three kind boundaries are four binary searches of <= 6 steps -- the
  // walk over the successors they replace was up to 64 x 3 dependent loads for the slowest lane of a wave (0.11 -> 0.02 ms)
  const uint32_t hs = t.head[s];
  uint32_t lo = s + 1u, hi = min(s + t.tq, t.nq);  // first position in [lo, hi] of another segment (hi: none)
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (t.head[mid] != hs) hi = mid; else lo = mid + 1u;
  }
  const uint32_t tn = lo - s;
  uint32_t ke[3];
#pragma unroll
  for (uint32_t v = 1; v <= 3u; ++v) {  // ke[v - 1] = queries of kind 1..v = first position of a kind > v
    uint32_t a = s, b = s + tn;
    while (a < b) {
      const uint32_t mid = (a + b) >> 1;
      if (t.s_kind[mid] > v) b = mid; else a = mid + 1u;
    }
    ke[v - 1] = a - s;
  }
  const bool sad = t.s_kind[s] == 0;
  const uint32_t meta = t.q_meta[s], lq = meta & 0xFFu, k = (meta >> 8) & 0xFFu, d = (meta >> 16) & 0xFFu;
  uint32_t s0, s1, step, nparts, ball0, balln;
  tile_window(t, s, s0, s1, step, nparts, ball0, balln);
  const unsigned long long sig = t.s_sig[s];
  const uint32_t base = t.tcount[s];
  uint32_t adj_rows = 0;
  const uint32_t adj = (!sad && k <= (uint32_t)kAdjRadius) ? adj_lookup(t.adj_hash, t.adj_mask, (uint32_t)sig, (uint32_t)(sig >> 32), &adj_rows) : 0u;
  for (uint32_t part = 0; part < nparts; ++part) {
    const uint32_t a0 = s0 + part * step, a1 = min(s1, a0 + step);
    Tile tl;
    tl.q0 = s; tl.nq = tn; tl.s0 = a0; tl.s1 = a1; tl.k = k; tl.lq = lq; tl.sig_lo = (uint32_t)sig; tl.sig_hi = (uint32_t)(sig >> 32);
    tl.kind = sad ? 0u : 1u; tl.d = d; tl.kend = sad ? 0u : (ke[0] | ke[1] << 8 | ke[2] << 16);
    tl.ball0 = ball0; tl.balln = balln; tl.adj = adj; tl.flags = (s == hs && part == 0u) ? 1u : 0u;
    t.tiles[base + part] = tl;
    // longest-processing-time first, bit-plane tiles before the count-vector ones: ascending key, stable
    // (a tile that streams an adjacency list: rows x (set-up + a test per query); the others: the signatures of their window)
    const unsigned long long cost = adj ? (unsigned long long)adj_rows * (8u + tn) + 16u : (unsigned long long)tn * (a1 - a0 + 64u);
    // order: tiles with an adjacency list (k_scan_adj) | other bit-plane tiles | count-vector tiles
    t.tkey[base + part] = (sad ? 0x80000000u : adj ? 0u : 0x40000000u) | (0x3FFFFFFFu - (uint32_t)(cost < 0x3FFFFFFFull ? cost : 0x3FFFFFFFull));
  }
  if (sad) atomicAdd(&t.ctr[3], nparts);
}

// ctr[5] = number of leading tiles with an adjacency list (the tiles are sorted: those come first)
__global__ __launch_bounds__(256) void k_tile_adj_count(const Tile* tiles, uint32_t n, uint32_t* ctr) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n && tiles[i].adj && (i + 1 == n || !tiles[i + 1].adj)) ctr[5] = i + 1u;
}

// ---- the small call (engine.hip small_find): queries stay in INPUT order, one tile per query -----------------------------------------
// Tile slot of (query s, part) = part * n + s: the bit-plane tiles (part 0) of consecutive queries take consecutive pair-list regions;
// a count-vector tile that walks its window is split over up to 8 slots (as in k_tile_emit); unused slots get nq = 0 and are skipped
// by k_scan_small.  The kernel also clears what the run accumulates into (counters, per-query sums): no memset launches.
__global__ __launch_bounds__(256) void k_small_tiles(GatherArgs g, TileArgs t, SmallZero z, uint32_t slots, const uint32_t* __restrict__ adj_hdr) {
  const uint32_t s = blockIdx.x * 256 + threadIdx.x, nthreads = gridDim.x * 256u;
#pragma unroll
  for (int a = 0; a < 8; ++a)
    for (uint32_t i = s; i < z.n[a]; i += nthreads) z.p[a][i] = 0u;
  if (s >= t.nq) return;
  enc_gather_body(g, s);  // the query's arrays (k_enc_gather's work, input order: g.perm is the identity) -- one launch less
  const uint32_t meta = t.q_meta[s], lq = meta & 0xFFu, k = (meta >> 8) & 0xFFu, d = (meta >> 16) & 0xFFu;
  Tile nul{};
  nul.kind = 1u;
  uint32_t nparts = 0;
  if (meta) {
    const uint32_t kind = t.s_kind[s];
    const bool sad = kind == 0;
    uint32_t s0, s1, step, ball0, balln;
    tile_window(t, s, s0, s1, step, nparts, ball0, balln);
    const unsigned long long sig = t.s_sig[s];
    uint32_t adj_rows = 0;
    const uint32_t adj = (!sad && k <= (uint32_t)kAdjRadius) ? adj_lookup(t.adj_hash, t.adj_mask, (uint32_t)sig, (uint32_t)(sig >> 32), &adj_rows) : 0u;
    // A tile that streams an adjacency list is ONE wave waiting for its own loads chunk after chunk (~2 us per chunk of 4 rows, 100 us
    // for a list of 200 rows): in a batch sixty thousand other waves fill the gaps, in a small call nothing does.  The rows of the list
    // are therefore shared out over up to `slots` tiles of at least 8 rows (two chunks) each.
    uint32_t rbeg = 0, rend = 0;
    if (adj) {
      const uint32_t* hp = adj_hdr + (size_t)(adj - 1u) * 8u;   // {first row, cumulative rows of the 7 length sections} -- as scan_tile reads it
      rbeg = k >= 3u ? 0u : k == 2u ? hp[1] : k == 1u ? hp[2] : hp[3];
      rend = k >= 3u ? hp[7] : k == 2u ? hp[6] : k == 1u ? hp[5] : hp[4];
      const uint32_t rows = rend > rbeg ? rend - rbeg : 0u;
      nparts = min(slots, max(1u, rows / 8u));
      step = ((rows + nparts - 1u) / nparts + 3u) & ~3u;      // whole chunks
    }
    for (uint32_t part = 0; part < nparts; ++part) {
      const uint32_t a0 = s0 + part * step, a1 = min(s1, a0 + step);
      Tile tl;
      tl.q0 = s; tl.nq = 1u; tl.s0 = a0; tl.s1 = a1; tl.k = k; tl.lq = lq; tl.sig_lo = (uint32_t)sig; tl.sig_hi = (uint32_t)(sig >> 32);
      tl.kind = sad ? 0u : 1u; tl.d = d;
      tl.kend = sad ? 0u : ((kind <= 1u ? 1u : 0u) | (kind <= 2u ? 1u : 0u) << 8 | (kind <= 3u ? 1u : 0u) << 16);  // ends of the kind-1 / -2 / -3 queries
      tl.ball0 = ball0; tl.balln = balln; tl.adj = adj; tl.flags = part == 0u ? 1u : 0u;
      if (adj) {
        tl.s0 = min(rend, rbeg + part * step);
        tl.s1 = min(rend, tl.s0 + step);
        tl.flags |= 2u;
        if (tl.s0 >= tl.s1 && part) tl.nq = 0u;  // (nothing left for this part)
      }
      t.tiles[(size_t)part * t.nq + s] = tl;
    }
  }
  (void)nul;
  for (uint32_t part = nparts; part < slots; ++part) t.tiles[(size_t)part * t.nq + s].nq = 0u;  // an unused slot: k_scan_small only looks at nq
}

int small_encode_launch(const HostModel& m, const DeviceLexicon* dl, const SmallEnc& e, const uint8_t* blob, const uint32_t* off, uint32_t n, uint32_t qw,
                        const anx_params& p, const SmallZero& z, uint32_t slots, bool stage_lds, const uint32_t* host_off, hipStream_t st, std::string& err) {
  const int NP = dl->nplanes;
  EncArgs ea;
  ea.blob = blob; ea.off = off; ea.n = n; ea.al = dl->alpha; ea.A = m.alphabet.size(); ea.NP = NP;
  ea.bits_ok = (dl->nsym <= 32 && !switches().scan_sad) ? 1 : 0;
  ea.ngroups = 1;
  for (uint8_t g : m.lex.sym_group) ea.ngroups = std::max(ea.ngroups, (int)g + 1);
  ea.kth = p.max_anagram_distance; ea.dth = p.max_edit_distance;
  ea.codes = e.codes; ea.meta = e.meta; ea.bits = e.bits; ea.sig = e.sig; ea.kind = e.kind; ea.cv = e.cv; ea.key = e.key; ea.blk = e.blk;
  ea.dbg = 0;
  ea.zero_cv = 1;
  for (uint32_t bk = 0; bk <= 16u; ++bk) ea.stage_off[bk] = host_off ? host_off[std::min<uint32_t>(bk * 256u, n)] : 0u;
  const dim3 gn((n + 255) / 256);
  if (stage_lds && host_off && n <= 4096u) hipLaunchKernelGGL(k_enc_strings<true>, gn, dim3(256), 0, st, ea);
  else hipLaunchKernelGGL(k_enc_strings<false>, gn, dim3(256), 0, st, ea);
  GatherArgs ga;
  ga.nq = n; ga.qw = qw; ga.NP = NP; ga.want_exact = 0;
  ga.perm = e.perm; ga.off = off; ga.codes = e.codes; ga.meta = e.meta; ga.bits = e.bits; ga.kind = e.kind; ga.cv = e.cv; ga.sig = e.sig;
  ga.q_rec = e.q_rec; ga.q_rows = e.q_rows; ga.q_bits = e.q_bits; ga.q_cv = e.q_cv; ga.q_meta = e.q_meta; ga.q_orig = e.q_orig;
  ga.qexact = e.qexact; ga.s_kind = e.s_kind; ga.s_sig = e.s_sig;
  ga.sigtab = dl->sig; ga.siglen_begin = dl->alpha.siglen_begin; ga.cls_planes = dl->cls_planes; ga.cstride = dl->cstride;
  TileArgs ta;
  ta.tq = 1; ta.nq = n; ta.q_meta = e.q_meta; ta.s_kind = e.s_kind; ta.s_sig = e.s_sig; ta.siglen_begin = dl->alpha.siglen_begin; ta.ctr = nullptr;
  ta.ball_tab = dl->ball_tab; ta.probe = probe_enabled() ? 1 : 0;
  ta.adj_hash = dl->adj_hash; ta.adj_mask = switches().scan_adj ? dl->adj_mask : 0u;
  ta.head = nullptr; ta.tcount = nullptr; ta.tiles = e.tiles; ta.tkey = nullptr;
  hipLaunchKernelGGL(k_small_tiles, gn, dim3(256), 0, st, ga, ta, z, slots, dl->adj_hdr);
  HIP_TRY(hipGetLastError());
  return ANX_OK;
}
int small_iota(uint32_t* perm, uint32_t n, hipStream_t st) {
  hipLaunchKernelGGL(k_iota, dim3((n + 255) / 256), dim3(256), 0, st, perm, n);
  return hipGetLastError() == hipSuccess ? ANX_OK : ANX_ENODEVICE;
}

// ---- host driver --------------------------------------------------------------------------------------------------------
namespace {
struct Scratch {  // pool blocks released together, and the private stream of the call
  std::vector<void*> blocks;
  int device;
  hipStream_t st;
  explicit Scratch(int dev) : device(dev), st(encoder_stream_acquire(dev)) {}
  template <typename T>
  int get(T** p, size_t count, std::string& err) {
    void* q = nullptr;
    HIP_TRY(pool_malloc(&q, std::max<size_t>(count * sizeof(T), 16)));
    blocks.push_back(q);
    *p = static_cast<T*>(q);
    return ANX_OK;
  }
  // the blocks go back to the pool, where another batch may take them at once: nothing enqueued by this call (on its own
  // stream) may still be using them -- the normal path has synchronised already, an error path has not
  ~Scratch() { (void)hipStreamSynchronize(st); for (void* q : blocks) pool_free(q); encoder_stream_release(device, st); }
};
template <typename K>
int sort_pairs(const K* kin, K* kout, const uint32_t* vin, uint32_t* vout, size_t n, unsigned b0, unsigned b1, Scratch& sc, hipStream_t st,
               std::string& err) {
  // (rocPRIM sorts up to 1 M items with a merge sort -- block sort + 10 merge passes of two launches, 0.2 ms for a million keys --
  // and larger inputs with Onesweep, one pass per 8 key bits; forcing Onesweep at 1 M 47-bit keys measured the same 0.22 ms)
  size_t bytes = 0;
  HIP_TRY(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, b0, b1, st));
  char* tmp = nullptr;
  int rc = sc.get(&tmp, bytes + 16, err);
  if (rc) return rc;
  HIP_TRY(rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, n, b0, b1, st));
  return ANX_OK;
}
template <typename T>
int balloc(T** dst, size_t count, std::string& err) {
  HIP_TRY(pool_malloc(reinterpret_cast<void**>(dst), std::max<size_t>(count * sizeof(T), 16)));
  return ANX_OK;
}
}  // namespace

// Fills the query and tile arrays of `b` (device) from the packed inputs.  Returns ANX_OK or an error code.
// off == nullptr: the n strings are the first n NUL-terminated spans of blob[0, blob_len); their offsets are found on the device.
int batch_encode_device(const HostModel& m, const DeviceLexicon* dl, Batch* b, const char* blob, size_t blob_bytes, const uint32_t* off,
                        size_t n, const anx_params& p, std::string& err, bool blob_on_device, bool after_stream, void* src_stream) {
  const bool timing = switches().encode_timing != 0;
  double t_prev = 0.0;
  auto tnow = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
  if (timing) t_prev = tnow();
  Scratch sc(dl->device);
  hipStream_t st = sc.st;  // private and non-blocking: encodes of other host threads and batches in flight do not serialise with this one
  auto lap = [&](const char* what) { if (timing) { (void)hipStreamSynchronize(st); const double t = tnow(); fprintf(stderr, "[anx encode/device] %-24s %8.2f ms\n", what, (t - t_prev) * 1e3); t_prev = t; } };
  const uint32_t n32 = (uint32_t)n;
  const size_t blob_len = !n ? 0 : off ? off[n] : blob_bytes;
  const int NP = dl->nplanes;
  int rc;
  if (blob_len + 4 * (size_t)n + 16 >= ((size_t)1 << 32)) { err = "inputs exceed 4 GB per batch (bytes + 4 per string): split the batch"; return ANX_ELIMIT; }
  if (n == 0) {  // nothing to encode: empty query arrays (the launches below do not take an empty grid)
    b->nq = 0; b->dmax = 0; b->qw = 1; b->ntiles = 0; b->n_sad_tiles = 0; b->n_adj_tiles = 0;
    if ((rc = balloc(&b->q_rec, 0, err)) || (rc = balloc(&b->qexact, 0, err)) || (rc = balloc(&b->q_cv, 0, err)) || (rc = balloc(&b->q_bits, 0, err)) ||
        (rc = balloc(&b->q_rows, 0, err)) || (rc = balloc(&b->q_meta, 0, err)) || (rc = balloc(&b->q_orig, 0, err)) || (rc = balloc(&b->d_tiles, 0, err)))
      return rc;
    return ANX_OK;
  }
  uint8_t *d_blob = nullptr, *d_codes = nullptr;
  uint32_t *d_off = nullptr, *d_meta = nullptr, *d_bits = nullptr, *d_kind = nullptr, *d_cv = nullptr, *d_ctr = nullptr, *d_blk = nullptr;
  unsigned long long* d_key = nullptr;
  unsigned long long* d_sig = nullptr;
  if (b->keep_text) {  // the inputs stay with the batch: the confusable weighting on the device reads them (conf.hip)
    if ((rc = balloc(&b->d_text, blob_len + 16, err)) || (rc = balloc(&b->d_textoff, n + 1, err))) return rc;
    d_blob = b->d_text;
    d_off = b->d_textoff;
    b->text_bytes = blob_len;
  } else if ((rc = sc.get(&d_blob, blob_len + 16, err)) || (rc = sc.get(&d_off, n + 1, err))) return rc;
  if ((rc = sc.get(&d_codes, blob_len + 4 * (size_t)n + 16, err)) ||
      (rc = sc.get(&d_meta, n, err)) || (rc = sc.get(&d_bits, n * NBITPLANES, err)) || (rc = sc.get(&d_kind, n, err)) ||
      (rc = sc.get(&d_cv, n * (size_t)NP, err)) || (rc = sc.get(&d_key, n, err)) || (rc = sc.get(&d_sig, n, err)) || (rc = sc.get(&d_ctr, 8, err)) ||
      (rc = sc.get(&d_blk, 3 * ((n + 255) / 256), err)))
    return rc;
  if (after_stream) {  // the caller's stream produced the buffer: this stream reads it behind that work (anx_batch_encode_packed_device_on)
    hipEvent_t ev = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e1 = hipEventRecord(ev, reinterpret_cast<hipStream_t>(src_stream));
    if (e1 == hipSuccess) e1 = hipStreamWaitEvent(st, ev, 0);
    (void)hipEventDestroy(ev);  // (released by the runtime once the wait has been satisfied)
    HIP_TRY(e1);
  }
  HIP_TRY(hipMemcpyAsync(d_blob, blob, blob_len, blob_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
  const int bits_ok = (dl->nsym <= 32 && !switches().scan_sad) ? 1 : 0;
  if (!bits_ok) HIP_TRY(hipMemsetAsync(d_cv, 0, n * (size_t)NP * 4, st));
  HIP_TRY(hipMemsetAsync(d_ctr, 0, 8 * sizeof(uint32_t), st));
  if (off) {
    HIP_TRY(hipMemcpyAsync(d_off, off, (n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  } else {
    const uint32_t nblk = (uint32_t)((blob_len + NUL_BLK - 1) / NUL_BLK);
    uint32_t* d_cnt = nullptr;
    if ((rc = sc.get(&d_cnt, nblk, err))) return rc;
    hipLaunchKernelGGL(k_nul_count, dim3(nblk), dim3(256), 0, st, d_blob, (uint32_t)blob_len, d_cnt, d_ctr + 4);
    size_t bytes = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, bytes, d_cnt, d_cnt, 0u, (size_t)nblk, rocprim::plus<uint32_t>(), st));
    char* tmp = nullptr;
    if ((rc = sc.get(&tmp, bytes + 16, err))) return rc;
    HIP_TRY(rocprim::exclusive_scan(tmp, bytes, d_cnt, d_cnt, 0u, (size_t)nblk, rocprim::plus<uint32_t>(), st));
    hipLaunchKernelGGL(k_nul_emit, dim3(nblk), dim3(256), 0, st, d_blob, (uint32_t)blob_len, d_cnt, n32, d_off);
    uint32_t have = 0;  // the encoder below must not run over offsets that were never written
    HIP_TRY(hipMemcpyAsync(&have, d_ctr + 4, sizeof have, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (have < n) { err = "packed inputs hold fewer strings than announced"; return ANX_EINVAL; }
  }
  EncArgs ea;
  ea.blob = d_blob; ea.off = d_off; ea.n = n32; ea.al = dl->alpha; ea.A = m.alphabet.size(); ea.NP = NP;
  ea.bits_ok = bits_ok;
  ea.ngroups = 1;
  for (uint8_t g : m.lex.sym_group) ea.ngroups = std::max(ea.ngroups, (int)g + 1);
  ea.kth = p.max_anagram_distance; ea.dth = p.max_edit_distance;
  ea.codes = d_codes; ea.meta = d_meta; ea.bits = d_bits; ea.sig = d_sig; ea.kind = d_kind; ea.cv = d_cv; ea.key = d_key; ea.blk = d_blk;
  ea.dbg = 0;
  ea.zero_cv = 0;
#ifdef ANX_DEBUG_SWITCHES
  { const char* e = getenv("ANX_ENC_DBG"); ea.dbg = e ? atoi(e) : 0; }
#endif
  const dim3 gn((n32 + 255) / 256);
  lap("alloc + H2D");
  hipLaunchKernelGGL(k_enc_strings<false>, gn, dim3(256), 0, st, ea);
  hipLaunchKernelGGL(k_enc_totals, dim3(1), dim3(256), 0, st, d_blk, gn.x, d_ctr);
  lap("k_enc_strings");
  // ---- (scan kernel, length, signature, kind) order: one sort of a packed key (until round 3: three stable passes over kind, the
  // 64-bit signature and the length with two gathers between them, 0.53 ms per million strings) ----
  uint32_t *perm_a = nullptr, *perm_b = nullptr;
  unsigned long long* k64_b = nullptr;
  if ((rc = sc.get(&perm_a, n, err)) || (rc = sc.get(&perm_b, n, err)) || (rc = sc.get(&k64_b, n, err))) return rc;
  hipLaunchKernelGGL(k_iota, gn, dim3(256), 0, st, perm_a, n32);
  if ((rc = sort_pairs(d_key, k64_b, perm_a, perm_b, n, 0, 13 + 5 * ea.ngroups, sc, st, err))) return rc;  // 9 + 5 ngroups + 3 bits, and "not encodable"
  const uint32_t* perm = perm_b;
  uint32_t h_ctr[8];
  HIP_TRY(hipMemcpyAsync(h_ctr, d_ctr, sizeof h_ctr, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  lap("sort");
  const uint32_t nq = h_ctr[2];
  b->nq = nq;
  b->dmax = h_ctr[1];
  b->qw = (std::max<uint32_t>(h_ctr[0], 1u) + 15u) / 16u;
  // ---- query arrays in sorted order -------------------------------------------------------------------------------------
  uint32_t* s_kind = nullptr;
  unsigned long long* s_sig = nullptr;
  if ((rc = sc.get(&s_kind, nq, err)) || (rc = sc.get(&s_sig, nq, err))) return rc;
  if ((rc = balloc(&b->q_rec, 2 * (size_t)nq, err)) || (rc = balloc(&b->qexact, nq, err)) || (rc = balloc(&b->q_cv, (size_t)nq * NP, err)) ||
      (rc = balloc(&b->q_bits, (size_t)nq * NBITPLANES, err)) || (rc = balloc(&b->q_rows, (size_t)nq * b->qw, err)) ||
      (rc = balloc(&b->q_meta, nq, err)) || (rc = balloc(&b->q_orig, nq, err)))
    return rc;
  b->ntiles = 0;
  b->n_sad_tiles = 0;
  b->n_adj_tiles = 0;
  if (nq == 0) {
    if ((rc = balloc(&b->d_tiles, 0, err))) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    return ANX_OK;
  }
  GatherArgs ga;
  ga.nq = nq; ga.qw = b->qw; ga.NP = NP; ga.want_exact = p.stop_at_exact_match ? 1 : 0;
  ga.perm = perm; ga.off = d_off; ga.codes = d_codes; ga.meta = d_meta; ga.bits = d_bits; ga.kind = d_kind; ga.cv = d_cv; ga.sig = d_sig;
  ga.q_rec = b->q_rec; ga.q_rows = b->q_rows; ga.q_bits = b->q_bits; ga.q_cv = b->q_cv; ga.q_meta = b->q_meta; ga.q_orig = b->q_orig;
  ga.qexact = b->qexact; ga.s_kind = s_kind; ga.s_sig = s_sig;
  ga.sigtab = dl->sig; ga.siglen_begin = dl->alpha.siglen_begin; ga.cls_planes = dl->cls_planes; ga.cstride = dl->cstride;
  const dim3 gq((nq + 255) / 256);
  hipLaunchKernelGGL(k_enc_gather, gq, dim3(256), 0, st, ga);
  lap("k_enc_gather");
  // ---- tiles --------------------------------------------------------------------------------------------------------------
  TileArgs ta;
  ta.tq = switches().scan_tq ? (uint32_t)switches().scan_tq : default_scan_tq(ea.ngroups);
  ta.nq = nq; ta.q_meta = b->q_meta; ta.s_kind = s_kind; ta.s_sig = s_sig; ta.siglen_begin = dl->alpha.siglen_begin; ta.ctr = d_ctr;
  ta.ball_tab = dl->ball_tab; ta.probe = probe_enabled() ? 1 : 0;
  ta.adj_hash = dl->adj_hash; ta.adj_mask = switches().scan_adj ? dl->adj_mask : 0u;
  uint32_t *d_head = nullptr, *d_tcount = nullptr;
  if ((rc = sc.get(&d_head, nq, err)) || (rc = sc.get(&d_tcount, (size_t)nq + 1, err))) return rc;
  ta.head = d_head; ta.tcount = d_tcount; ta.tiles = nullptr; ta.tkey = nullptr;
  hipLaunchKernelGGL(k_tile_heads, gq, dim3(256), 0, st, ta);
  {
    size_t bytes = 0;
    HIP_TRY(rocprim::inclusive_scan(nullptr, bytes, d_head, d_head, (size_t)nq, rocprim::maximum<uint32_t>(), st));
    char* tmp = nullptr;
    if ((rc = sc.get(&tmp, bytes + 16, err))) return rc;
    HIP_TRY(rocprim::inclusive_scan(tmp, bytes, d_head, d_head, (size_t)nq, rocprim::maximum<uint32_t>(), st));
  }
  hipLaunchKernelGGL(k_tile_count, dim3((nq + 1 + 255) / 256), dim3(256), 0, st, ta);
  {
    size_t bytes = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, bytes, d_tcount, d_tcount, 0u, (size_t)nq + 1, rocprim::plus<uint32_t>(), st));
    char* tmp = nullptr;
    if ((rc = sc.get(&tmp, bytes + 16, err))) return rc;
    HIP_TRY(rocprim::exclusive_scan(tmp, bytes, d_tcount, d_tcount, 0u, (size_t)nq + 1, rocprim::plus<uint32_t>(), st));
  }
  uint32_t ntiles = 0;
  HIP_TRY(hipMemcpyAsync(&ntiles, d_tcount + nq, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  Tile *t_unsorted = nullptr;
  uint32_t *tkey_a = nullptr, *tkey_b = nullptr, *tperm_a = nullptr, *tperm_b = nullptr;
  if ((rc = sc.get(&t_unsorted, ntiles, err)) || (rc = sc.get(&tkey_a, ntiles, err)) || (rc = sc.get(&tkey_b, ntiles, err)) ||
      (rc = sc.get(&tperm_a, ntiles, err)) || (rc = sc.get(&tperm_b, ntiles, err)) || (rc = balloc(&b->d_tiles, ntiles, err)))
    return rc;
  ta.tiles = t_unsorted; ta.tkey = tkey_a;
  hipLaunchKernelGGL(k_tile_emit, gq, dim3(256), 0, st, ta);
  if (ntiles) {
    const dim3 gt((ntiles + 255) / 256);
    hipLaunchKernelGGL(k_iota, gt, dim3(256), 0, st, tperm_a, ntiles);
    if ((rc = sort_pairs(tkey_a, tkey_b, tperm_a, tperm_b, ntiles, 0, 32, sc, st, err))) return rc;
    hipLaunchKernelGGL(k_gather<Tile>, gt, dim3(256), 0, st, t_unsorted, tperm_b, b->d_tiles, ntiles);
    hipLaunchKernelGGL(k_tile_adj_count, gt, dim3(256), 0, st, b->d_tiles, ntiles, d_ctr);
  }
  HIP_TRY(hipMemcpyAsync(h_ctr, d_ctr, sizeof h_ctr, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  b->ntiles = ntiles;
  b->n_sad_tiles = h_ctr[3];
  b->n_adj_tiles = h_ctr[5];
  lap("tiles");
  return ANX_OK;
}

}  // namespace anx
