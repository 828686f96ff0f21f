// adjacency.hip -- the signature adjacency lists (adjacency.h) BUILT ON THE DEVICE, from the lexicon tables the replica already holds.
//
// The host builder (adjacency.cpp) takes seconds for a large lexicon (2.5 M lists, 12.8 GB for the 1 M-entry lexicon of BASELINE
// configs[3]: 3.2 s of 16 threads + the upload of 12.8 GB from pageable memory); the device has everything it needs -- the
// signature table with its entry runs, the signature hash table, the scan records, the L1 balls of signature offsets -- and builds
// the same lists in milliseconds:
//   keys   : every lexicon signature + every offset of the ball of radius `closure`, sorted, duplicates dropped (rocPRIM)
//   count  : one wave per key walks the ball of radius kAdjRadius through the signature hash table (the scan's own probe walk) and
//            sums the entries of the runs it finds per length section -> rows per section, distance of the key from the lexicon
//   layout : the host picks the lists that fit the budget ((distance, rows) ascending; all of them as a rule) and lays them out
//   fill   : one wave per list again: the runs of a probe step are ranked per section (ballot + prefix sum) and copied
//   table  : signature -> list, open addressing (every key within 16 slots of its home, as the probes expect)
// Same lists as the host builder up to the order of the records inside a section (nothing depends on it: the scan tests every
// record of a section against every query of the tile); tests/test_gpu_adjacency.py compares the two.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

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

namespace {

// sig + offset (8 x int8): false when a group sum leaves [0, 255] or the length leaves [1, kMaxSymbols]; *dlen = sum of the offset
__device__ inline bool adjb_apply(unsigned long long sig, unsigned long long off, unsigned long long* out, int* dlen) {
  unsigned long long r = 0;
  int len = 0, dl = 0;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int o = (int)(signed char)(off >> (8 * g));
    const int v = (int)((sig >> (8 * g)) & 0xFFu) + o;
    if (v < 0 || v > 255) return false;
    len += v;
    dl += o;
    r |= (unsigned long long)v << (8 * g);
  }
  if (len < 1 || len > kMaxSymbols) return false;
  *out = r;
  *dlen = dl;
  return true;
}
// the entry run of a lexicon signature (first entry, entries), entries == 0: no such signature
__device__ inline uint2 adjb_probe(const uint4* __restrict__ htab, uint32_t mask, unsigned long long v) {
  const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  uint32_t h = sig_hash(lo, hi) & mask;
  for (int p = 0; p < 17; ++p) {
    const uint4 e = htab[h];
    if (!e.w) break;
    if (e.x == lo && e.y == hi) return make_uint2(e.z, e.w);
    h = (h + 1u) & mask;
  }
  return make_uint2(0u, 0u);
}

__global__ __launch_bounds__(256) void k_adjb_keys(const uint4* __restrict__ sig_e, uint32_t nsigs, const unsigned long long* __restrict__ near, uint32_t nnear,
                                                   unsigned long long* __restrict__ keys) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)nsigs * nnear) return;
  const uint32_t i = (uint32_t)(idx / nnear), j = (uint32_t)(idx % nnear);
  const uint4 s = sig_e[i];
  unsigned long long u = 0;
  int dl;
  keys[idx] = adjb_apply((unsigned long long)s.x | (unsigned long long)s.y << 32, near[j], &u, &dl) ? u : ~0ull;
}

struct AdjbArgs {
  const unsigned long long* keys;   // the signatures that get a list, ascending
  uint32_t nk;
  const uint4* htab;                // DeviceLexicon::sighash_e
  uint32_t hmask;
  const unsigned long long* ball;   // offsets of radius kAdjRadius
  uint32_t nball;
  const unsigned long long* ball1;  // offsets of radius 1 (the distance of a key from the lexicon)
  uint32_t nball1;
  uint32_t* cum;                    // [nk][8]: {first row, cumulative rows of the 7 sections} -- AdjHdr
  uint32_t* nrec;                   // [nk] records without padding
  uint8_t* tier;                    // [nk] distance from the lexicon (0, 1, 2 = further)
  const uint4* scan_rec;            // planes of every entry
  uint2* planes;
  uint32_t* ids;
  uint32_t pad_id;
  uint4* table;                     // signature -> list
  uint32_t tmask;
  uint32_t* maxprobe;
};

// rows per length section of every key's list, and how far the key is from the lexicon
__global__ __launch_bounds__(256) void k_adjb_count(AdjbArgs a) {
  __shared__ uint32_t s_cnt[4][8];
  const uint32_t wid = threadIdx.x >> 6, lane = threadIdx.x & 63, k = blockIdx.x * 4 + wid;
  if (lane < 8) s_cnt[wid][lane] = 0;
  __syncthreads();
  if (k < a.nk) {
    const unsigned long long key = a.keys[k];
    for (uint32_t b0 = 0; b0 < a.nball; b0 += 64) {
      const uint32_t j = b0 + lane;
      unsigned long long v;
      int dl;
      if (j < a.nball && adjb_apply(key, a.ball[j], &v, &dl)) {
        const uint2 run = adjb_probe(a.htab, a.hmask, v);
        if (run.y) atomicAdd(&s_cnt[wid][dl + kAdjRadius], run.y);
      }
    }
    bool near1 = false, self = false;
    for (uint32_t b0 = 0; b0 < a.nball1; b0 += 64) {
      const uint32_t j = b0 + lane;
      unsigned long long v;
      int dl;
      if (j < a.nball1 && adjb_apply(key, a.ball1[j], &v, &dl) && adjb_probe(a.htab, a.hmask, v).y) { near1 = true; if (v == key) self = true; }
    }
    const bool any1 = __any(near1), anyself = __any(self);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      uint32_t rows = 0, recs = 0;
      a.cum[(size_t)k * 8] = 0;
      for (int s = 0; s < kAdjSections; ++s) {
        const uint32_t c = s_cnt[wid][s];
        rows += (c + kAdjRow - 1) / kAdjRow;
        recs += c;
        a.cum[(size_t)k * 8 + 1 + s] = rows;
      }
      a.nrec[k] = recs;
      a.tier[k] = anyself ? 0 : any1 ? 1 : 2;
    }
  }
}

__device__ inline uint32_t adjb_wave_incl(uint32_t x, uint32_t lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)x, o);
    if ((int)lane >= o) x += up;
  }
  return x;
}
// the records of every kept list: cum[k][0] = its first row (0xFFFFFFFF: dropped), sections padded to whole rows
__global__ __launch_bounds__(256) void k_adjb_fill(AdjbArgs a) {
  __shared__ uint32_t s_pos[4][8];
  const uint32_t wid = threadIdx.x >> 6, lane = threadIdx.x & 63, k = blockIdx.x * 4 + wid;
  if (k >= a.nk) return;
  const uint32_t row0 = a.cum[(size_t)k * 8];
  if (row0 == 0xFFFFFFFFu) return;  // wave-uniform
  if (lane < (uint32_t)kAdjSections) s_pos[wid][lane] = 0;
  __builtin_amdgcn_wave_barrier();
  const unsigned long long key = a.keys[k];
  const uint32_t* __restrict__ cum = a.cum + (size_t)k * 8 + 1;
  for (uint32_t b0 = 0; b0 < a.nball; b0 += 64) {
    const uint32_t j = b0 + lane;
    unsigned long long v;
    int dl = 0;
    uint2 run = make_uint2(0u, 0u);
    if (j < a.nball && adjb_apply(key, a.ball[j], &v, &dl)) run = adjb_probe(a.htab, a.hmask, v);
    const uint32_t sec = (uint32_t)(dl + kAdjRadius);
    unsigned long long pos = 0;
    for (uint32_t s = 0; s < (uint32_t)kAdjSections; ++s) {  // the step's runs of section s, ranked in lane order
      const bool mine = run.y != 0 && sec == s;
      if (!__any(mine)) continue;  // wave-uniform
      const uint32_t incl = adjb_wave_incl(mine ? run.y : 0u, lane);
      const uint32_t total = (uint32_t)__shfl((int)incl, 63);
      const uint32_t base = s_pos[wid][s];
      if (mine) pos = ((unsigned long long)row0 + (s ? cum[s - 1] : 0u)) * kAdjRow + base + (incl - run.y);
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) s_pos[wid][s] = base + total;
      __builtin_amdgcn_wave_barrier();
    }
    for (uint32_t i = 0; i < run.y; ++i) {
      const uint4 r = a.scan_rec[run.x + i];
      a.planes[pos + i] = make_uint2(r.x, r.y);
      a.ids[pos + i] = run.x + i;
    }
  }
  __builtin_amdgcn_wave_barrier();
  for (uint32_t s = 0; s < (uint32_t)kAdjSections; ++s) {  // padding: shares no symbol with anything
    const unsigned long long beg = ((unsigned long long)row0 + (s ? cum[s - 1] : 0u)) * kAdjRow + s_pos[wid][s], end = ((unsigned long long)row0 + cum[s]) * kAdjRow;
    for (unsigned long long p = beg + lane; p < end; p += 64) { a.planes[p] = make_uint2(0u, 0u); a.ids[p] = a.pad_id; }
  }
}
// table signature -> list: slot {sig lo, sig hi, header index + 1, rows}; hidx[k] = header index of key k (0xFFFFFFFF: dropped)
__global__ __launch_bounds__(256) void k_adjb_table(AdjbArgs a, const uint32_t* __restrict__ hidx) {
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= a.nk || hidx[k] == 0xFFFFFFFFu) return;
  const unsigned long long key = a.keys[k];
  const uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
  uint32_t h = sig_hash(lo, hi) & a.tmask, probes = 0;
  uint32_t* words = reinterpret_cast<uint32_t*>(a.table);
  while (atomicCAS(&words[(size_t)h * 4 + 2], 0u, hidx[k] + 1u) != 0u) { h = (h + 1u) & a.tmask; ++probes; }
  words[(size_t)h * 4] = lo;
  words[(size_t)h * 4 + 1] = hi;
  words[(size_t)h * 4 + 3] = a.cum[(size_t)k * 8 + kAdjSections];
  atomicMax(a.maxprobe, probes);
}
__global__ __launch_bounds__(256) void k_adjb_hdr(AdjbArgs a, const uint32_t* __restrict__ hidx, uint32_t* __restrict__ hdr) {
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= a.nk || hidx[k] == 0xFFFFFFFFu) return;
#pragma unroll
  for (int w = 0; w < 8; ++w) hdr[(size_t)hidx[k] * 8 + w] = a.cum[(size_t)k * 8 + w];
}

}  // namespace

// Builds the lists on the replica `d` (its lexicon tables are uploaded) and fills the adj_* members; stats: what the host builder
// reports (AdjIndex: counts, rows, len_records / class_records / class_nsig for the length split).  img: the host image (entries
// per signature for the statistics).
int adjacency_build_device(DeviceLexicon* d, const LexiconImage& img, int closure, size_t budget_bytes, AdjIndex& stats, std::string& err) {
  const auto t0 = std::chrono::steady_clock::now();
  stats.nsig_lexicon = img.nsigs;
  stats.nsig_closure = stats.nsig_kept = 0;
  stats.records = stats.rows = stats.rows_wanted = 0;
  if (img.nsym > 32 || img.nsigs == 0 || !d->sighash_e) return ANX_OK;
  closure = std::max(0, std::min(closure, kAdjMaxClosure));
  if (!d->ball_n[kAdjRadius] || !d->ball_n[closure] || !d->ball_n[1]) return ANX_OK;  // (a ball too large to enumerate: no lists)
  HIP_TRY(hipSetDevice(d->device));
  hipStream_t st = nullptr;  // the NULL stream: model set-up, nothing else runs
  std::vector<void*> owned;
  struct Free { std::vector<void*>& v; ~Free() { (void)hipDeviceSynchronize(); for (void* q : v) pool_free(q); } } fr{owned};
  auto dal = [&](void** p_, size_t bytes) -> int { HIP_TRY(pool_malloc(p_, std::max<size_t>(bytes, 16))); owned.push_back(*p_); return ANX_OK; };
  int rc;
  const uint32_t nnear = d->ball_n[closure];
  const size_t nraw = (size_t)img.nsigs * nnear;
  unsigned long long *raw = nullptr, *sorted = nullptr, *ukeys = nullptr;
  uint32_t* d_nk = nullptr;
  if ((rc = dal((void**)&raw, nraw * 8)) || (rc = dal((void**)&sorted, nraw * 8)) || (rc = dal((void**)&ukeys, nraw * 8)) || (rc = dal((void**)&d_nk, 16))) return rc;
  hipLaunchKernelGGL(k_adjb_keys, dim3((unsigned)((nraw + 255) / 256)), dim3(256), 0, st, d->sig_e, img.nsigs, d->ball + d->ball_off[closure], nnear, raw);
  {
    size_t bytes = 0;
    HIP_TRY(rocprim::radix_sort_keys(nullptr, bytes, raw, sorted, nraw, 0, 64, st));
    void* tmp = nullptr;
    if ((rc = dal(&tmp, bytes + 16))) return rc;
    HIP_TRY(rocprim::radix_sort_keys(tmp, bytes, raw, sorted, nraw, 0, 64, st));
    bytes = 0;
    HIP_TRY(rocprim::unique(nullptr, bytes, sorted, ukeys, d_nk, nraw, rocprim::equal_to<unsigned long long>(), st));
    void* tmp2 = nullptr;
    if ((rc = dal(&tmp2, bytes + 16))) return rc;
    HIP_TRY(rocprim::unique(tmp2, bytes, sorted, ukeys, d_nk, nraw, rocprim::equal_to<unsigned long long>(), st));
  }
  uint32_t nk = 0;
  HIP_TRY(hipMemcpy(&nk, d_nk, 4, hipMemcpyDeviceToHost));
  {  // the invalid key (all ones) sorts last
    unsigned long long last = 0;
    if (nk) HIP_TRY(hipMemcpy(&last, ukeys + (nk - 1), 8, hipMemcpyDeviceToHost));
    if (nk && last == ~0ull) --nk;
  }
  stats.nsig_closure = nk;
  if (!nk) return ANX_OK;
  AdjbArgs a{};
  a.keys = ukeys; a.nk = nk; a.htab = d->sighash_e; a.hmask = d->hash_mask;
  a.ball = d->ball + d->ball_off[kAdjRadius]; a.nball = d->ball_n[kAdjRadius];
  a.ball1 = d->ball + d->ball_off[1]; a.nball1 = d->ball_n[1];
  a.scan_rec = d->scan_rec; a.pad_id = img.nentries;
  if ((rc = dal((void**)&a.cum, (size_t)nk * 8 * 4)) || (rc = dal((void**)&a.nrec, (size_t)nk * 4)) || (rc = dal((void**)&a.tier, (size_t)nk + 16))) return rc;
  hipLaunchKernelGGL(k_adjb_count, dim3((nk + 3) / 4), dim3(256), 0, st, a);
  // ---- layout on the host: which lists fit the budget, their first rows, the statistics of the length split -------------------------
  std::vector<unsigned long long> hkeys(nk);
  std::vector<uint32_t> hcum((size_t)nk * 8), hnrec(nk), hidx(nk, 0xFFFFFFFFu);
  std::vector<uint8_t> htier(nk);
  HIP_TRY(hipMemcpy(hkeys.data(), ukeys, (size_t)nk * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(hcum.data(), a.cum, (size_t)nk * 32, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(hnrec.data(), a.nrec, (size_t)nk * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(htier.data(), a.tier, nk, hipMemcpyDeviceToHost));
  const uint64_t row_bytes = (uint64_t)kAdjRow * (sizeof(AdjPlanes) + sizeof(uint32_t));
  uint64_t want = 0;
  for (uint32_t k = 0; k < nk; ++k) want += hcum[(size_t)k * 8 + kAdjSections];
  stats.rows_wanted = want;
  std::vector<uint8_t> keep(nk, 1);
  {  // the lists are an accelerator, not a requirement: never more than 60 % of what the device has free right now (the build's own
     // transient buffers above are allocated already, so they are counted), whatever ANX_ADJ_MB says -- a busier or smaller device,
     // or many replicas on one device, get fewer lists instead of a failed model load
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget_bytes = std::min<size_t>(budget_bytes, free_b / 10 * 6);
    else (void)hipGetLastError();
  }
  if (want * row_bytes > budget_bytes) {
    std::vector<uint32_t> ord(nk);
    for (uint32_t k = 0; k < nk; ++k) ord[k] = k;
    std::sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) {
      if (htier[x] != htier[y]) return htier[x] < htier[y];
      const uint32_t rx = hcum[(size_t)x * 8 + kAdjSections], ry = hcum[(size_t)y * 8 + kAdjSections];
      return rx != ry ? rx < ry : x < y;
    });
    uint64_t used = 0;
    for (uint32_t k : ord) {
      const uint64_t need = (uint64_t)hcum[(size_t)k * 8 + kAdjSections] * row_bytes;
      if (used + need <= budget_bytes) used += need; else keep[k] = 0;
    }
  }
  uint64_t rows = 0;
  uint32_t nkept = 0;
  for (uint32_t k = 0; k < nk; ++k) {
    const uint32_t r = hcum[(size_t)k * 8 + kAdjSections];
    if (keep[k] && rows + r < 0xFFFFFFFFull) {
      hcum[(size_t)k * 8] = (uint32_t)rows;
      rows += r;
      stats.records += hnrec[k];
      hidx[k] = nkept++;
    } else hcum[(size_t)k * 8] = 0xFFFFFFFFu;
  }
  stats.rows = rows;
  stats.nsig_kept = nkept;
  {  // records per query of every length / class, signatures per class (capi.cpp LengthCost), from the lexicon's own signatures
    double num[256] = {}, den[256] = {};
    std::vector<double> cnum(64 * 1024, 0.0), cden(64 * 1024, 0.0);
    stats.class_nsig.assign(64 * 1024, 0.0f);
    for (uint32_t k = 0; k < nk; ++k) {
      if (htier[k] != 0) continue;
      const unsigned long long key = hkeys[k];
      int len = 0;
      for (int g = 0; g < 8; ++g) len += (int)((key >> (8 * g)) & 0xFFu);
      // the signature's run in the image: signatures are stored by (length, signature)
      const uint32_t lo_i = img.siglen_begin[std::min(len, kMaxSymbols)], hi_i = img.siglen_begin[std::min(len, kMaxSymbols) + 1];
      uint32_t l = lo_i, h = hi_i;
      while (l < h) {
        const uint32_t mid = (l + h) / 2;
        const unsigned long long v = (unsigned long long)img.sig_lo[mid] | (unsigned long long)img.sig_hi[mid] << 32;
        if (v < key) l = mid + 1; else h = mid;
      }
      if (l >= hi_i || ((unsigned long long)img.sig_lo[l] | (unsigned long long)img.sig_hi[l] << 32) != key) continue;
      const double e = (double)(img.cls_off[std::min(img.sig_cbeg[l + 1], img.nclasses)] - img.cls_off[std::min(img.sig_cbeg[l], img.nclasses)]);
      num[std::min(len, 255)] += e * (double)hnrec[k];
      den[std::min(len, 255)] += e;
      if (len < 64) {
        const size_t c = (size_t)len * 1024 + std::min<size_t>(key & 0xFFu, 31) * 32 + std::min<size_t>((key >> 8) & 0xFFu, 31);
        cnum[c] += e * (double)hnrec[k];
        cden[c] += e;
        stats.class_nsig[c] += 1.0f;
      }
    }
    for (int L = 0; L < 256; ++L) stats.len_records[L] = den[L] > 0.0 ? num[L] / den[L] : 0.0;
    stats.class_records.assign(64 * 1024, 0.0f);
    for (size_t c = 0; c < cnum.size(); ++c) stats.class_records[c] = cden[c] > 0.0 ? (float)(cnum[c] / cden[c]) : 0.0f;
  }
  // ---- the lists, their headers and the table: the replica's own (pool) memory -----------------------------------------------------
  auto own = [&](void** p_, size_t bytes) -> int { HIP_TRY(pool_malloc(p_, std::max<size_t>(bytes, 16))); d->bytes += std::max<size_t>(bytes, 16); return ANX_OK; };
  uint32_t* d_hidx = nullptr;
  if ((rc = dal((void**)&d_hidx, (size_t)nk * 4)) || (rc = dal((void**)&a.maxprobe, 16))) return rc;
  HIP_TRY(hipMemcpy(a.cum, hcum.data(), (size_t)nk * 32, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_hidx, hidx.data(), (size_t)nk * 4, hipMemcpyHostToDevice));
  if ((rc = own((void**)&d->adj_planes, rows * kAdjRow * sizeof(uint2))) || (rc = own((void**)&d->adj_ids, rows * kAdjRow * 4)) ||
      (rc = own((void**)&d->adj_hdr, (size_t)std::max<uint32_t>(nkept, 1) * 32)))
    return rc;
  if (switches().adj_fail) { err = "injected failure (ANX_ADJ_FAIL)"; return ANX_ENODEVICE; }
  a.planes = d->adj_planes; a.ids = d->adj_ids;
  hipLaunchKernelGGL(k_adjb_fill, dim3((nk + 3) / 4), dim3(256), 0, st, a);
  hipLaunchKernelGGL(k_adjb_hdr, dim3((nk + 255) / 256), dim3(256), 0, st, a, d_hidx, d->adj_hdr);
  uint32_t hsize = 64;
  while (hsize < 2 * nkept) hsize <<= 1;
  for (;; hsize <<= 1) {
    void* tab = nullptr;
    HIP_TRY(pool_malloc(&tab, (size_t)hsize * 16));
    HIP_TRY(hipMemsetAsync(tab, 0, (size_t)hsize * 16, st));
    HIP_TRY(hipMemsetAsync(a.maxprobe, 0, 4, st));
    a.table = static_cast<uint4*>(tab); a.tmask = hsize - 1;
    hipLaunchKernelGGL(k_adjb_table, dim3((nk + 255) / 256), dim3(256), 0, st, a, d_hidx);
    uint32_t mp = 0;
    HIP_TRY(hipMemcpy(&mp, a.maxprobe, 4, hipMemcpyDeviceToHost));
    if (mp <= 16) { d->adj_hash = a.table; d->bytes += (size_t)hsize * 16; break; }  // every key within 16 slots of its home
    pool_free(tab);
    if (hsize >= (1u << 30)) { err = "adjacency table: probe sequences do not fit"; return ANX_ELIMIT; }
  }
  d->adj_mask = hsize - 1;
  d->adj_nhdr = nkept;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipGetLastError());
  stats.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return ANX_OK;
}

// host copies of the table and the headers (the threaded host encoder, ANX_ENCODE=host, looks the tiles' lists up itself)
int adjacency_host_copies(const DeviceLexicon* d, std::string& err) {
  if (!d->adj_mask || !d->adj_hash_host.empty()) return ANX_OK;
  HIP_TRY(hipSetDevice(d->device));
  d->adj_hash_host.resize((size_t)d->adj_mask + 1);
  d->adj_hdr_host.resize(std::max<uint32_t>(d->adj_nhdr, 1));
  HIP_TRY(hipMemcpy(d->adj_hash_host.data(), d->adj_hash, d->adj_hash_host.size() * sizeof(AdjSlot), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(d->adj_hdr_host.data(), d->adj_hdr, (size_t)std::max<uint32_t>(d->adj_nhdr, 1) * sizeof(AdjHdr), hipMemcpyDeviceToHost));
  return ANX_OK;
}

// test hook: the lists of the given signatures as the replica holds them (whoever built them), in the form of anx_debug_adjacency
int adjacency_debug_lists(const DeviceLexicon* d, const uint64_t* sigs, size_t n, uint32_t* out_cum, uint32_t** out_ids, std::string& err) {
  int rc = adjacency_host_copies(d, err);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(d->device));
  std::vector<uint32_t> h1(n, 0u);
  size_t total = 0;
  for (size_t i = 0; i < n; ++i) {
    if (d->adj_hash_host.empty()) continue;
    const uint32_t lo = (uint32_t)sigs[i], hi = (uint32_t)(sigs[i] >> 32);
    uint32_t h = sig_hash(lo, hi) & d->adj_mask;
    for (int p = 0; p < 17; ++p) {
      const AdjSlot& s = d->adj_hash_host[h];
      if (!s.hdr1) break;
      if (s.lo == lo && s.hi == hi) { h1[i] = s.hdr1; break; }
      h = (h + 1u) & d->adj_mask;
    }
    if (h1[i]) total += (size_t)d->adj_hdr_host[h1[i] - 1].cum[kAdjSections - 1] * kAdjRow;
  }
  uint32_t* ids = static_cast<uint32_t*>(malloc(std::max<size_t>(total, 1) * sizeof(uint32_t)));
  if (!ids) { err = "out of memory"; return ANX_EINVAL; }
  size_t pos = 0;
  for (size_t i = 0; i < n; ++i) {
    uint32_t* c = out_cum + i * (kAdjSections + 1);
    if (!h1[i]) { for (int s = 0; s <= kAdjSections; ++s) c[s] = 0xFFFFFFFFu; continue; }
    const AdjHdr& h = d->adj_hdr_host[h1[i] - 1];
    c[0] = (uint32_t)(pos / kAdjRow);
    for (int s = 0; s < kAdjSections; ++s) c[s + 1] = h.cum[s];
    const size_t cnt = (size_t)h.cum[kAdjSections - 1] * kAdjRow;
    if (cnt && hipMemcpy(ids + pos, d->adj_ids + (size_t)h.row0 * kAdjRow, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) {
      free(ids);
      err = "hipMemcpy failed";
      return ANX_ENODEVICE;
    }
    pos += cnt;
  }
  *out_ids = ids;
  return ANX_OK;
}

}  // namespace anx
