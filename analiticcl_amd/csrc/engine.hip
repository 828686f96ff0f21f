// engine.hip -- the HIP (gfx950 / CDNA4) variant-query pipeline of the anx engine.
//
// Replaces, for a whole batch of queries at once, the reference's
//   find_nearest_anahashes  (/root/reference/src/lib.rs:1143-1308)  -> k_scan_bits / k_scan_sad       (kernels_scan.hpp)
//   gather_instances        (src/lib.rs:1311-1402, src/distance.rs)  -> k_filter_score, k_score_fast8,
//                                                                      k_score_pairs                  (kernels_score.hpp)
//   score_and_rank          (src/lib.rs:1405-1653, src/types.rs:334-365) -> the same + k_compact, k_rank (kernels_rank.hpp)
// This file: device memory pool, lexicon upload, batch encoding / tiling, the launch sequence, result download.
// One translation unit (the kernel headers are included below).  Integer work only: no MFMA.  Wave = 64 lanes
// everywhere.  See DESIGN.md for layout and rooflines.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <numeric>
#include <thread>
#include <type_traits>

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
#include "kernels_swar.hpp"
#include "kernels_scan.hpp"
#include "kernels_prefix.hpp"
#include "kernels_score.hpp"
#include "kernels_rank.hpp"

// ------------------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------------------
// layout of the pinned block a run reads back into (Batch::h_read), in uint32 words
enum { HR_RCTR = 0, HR_SCTR = SCAN_REGIONS * RC_STRIDE, HR_LCTR = 2 * SCAN_REGIONS * RC_STRIDE, HR_CTR = 5 * SCAN_REGIONS * RC_STRIDE,
       HR_TOTAL_SURV = HR_CTR + CTR_N, HR_TOTAL_RESULTS = HR_TOTAL_SURV + 1, HR_CONF = HR_TOTAL_RESULTS + 1 /* 2 words */, HR_N = HR_CONF + 2 };
constexpr size_t HR_COLD_OFF = (HR_N * sizeof(uint32_t) + 63) & ~(size_t)63;  // byte offset of the FsCold staging copy in Batch::h_read
static void shells_destroy(int device);
static void small_ctxs_destroy(int device);

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
struct SmallCtx;  // small_path.hpp
namespace {
struct DevPool {
  std::mutex mu;
  std::multimap<size_t, void*> free_blocks;        // size -> block
  std::unordered_map<void*, size_t> size_of;       // every live or cached block of this pool
  size_t cached = 0;
  std::unordered_map<void*, uint64_t> freed_at;   // cached blocks: when they came back (a full cache lets its oldest blocks go)
  uint64_t clock = 0;
  int lexicons = 0;
  std::vector<hipStream_t> idle_streams;           // non-blocking streams of the device-side encoder, handed out per call
  hipStream_t run_streams[2] = {nullptr, nullptr}; // the library's own streams for asynchronous runs (batch_run_async)
  unsigned run_counter = 0;
  std::vector<BatchShell> shells;                  // events + pinned read-back blocks of freed batches (batch_free), handed to the next batch
  std::vector<SmallCtx*> small_idle;               // contexts of the small call (small_path.hpp), checked out per call
  std::vector<std::pair<void*, size_t>> doomed;    // blocks evicted from the cache, not yet handed to hipFree (pool_free: hipFree waits for the device)
  size_t doomed_bytes = 0;
};
// bytes of freed blocks kept per device (MI355X: 288 GB HBM; a 1 M-query batch holds 3-6 GB of scratch).  ANX_POOL_CACHE_MB
// overrides the default; anx_device_pool_trim() hands the cache back to the driver at any time.
static size_t pool_cache_limit() {
  static const size_t lim = []() {
    const char* e = getenv("ANX_POOL_CACHE_MB");
    const long long v = e ? atoll(e) : -1;
    return v >= 0 ? (size_t)v << 20 : (size_t)96 << 30;   // (32 GB until round 6: six un-hinted first runs of a million queries, 7 GB each, outgrew it)
  }();
  return lim;
}
DevPool& pool_of(int device) {
  static DevPool pools[64];
  return pools[device >= 0 && device < 64 ? device : 0];
}
}  // namespace
// A private non-blocking stream for one encoder call (encode.hip): on the legacy NULL stream every encode would serialise with
// the in-flight batches of other host threads that run on blocking streams.  Streams are kept per device and reused.
// A thread may bring its own encoder stream (anx_pipeline's encode thread: created together with the pipeline's run streams, so that
// the runtime's least-used-hardware-queue rule puts the three on different queues -- see anx_pipeline_new).
static thread_local hipStream_t t_encoder_stream = nullptr;
// A non-blocking stream on the current device.  high = the device's highest stream priority: the encoder of batch i + 1 (and search
// mode's lattice-build kernels) are chains of ~45 short dependent launches that run beside the scan / scoring kernels of batch i --
// tens of thousands of waves queued on streams of normal priority; at equal priority every launch of the chain waits its turn behind
// them (round 5: the encoder "under" a run took the run's length), at high priority the dispatcher serves its few workgroups first.
// ANX_ENC_PRIORITY=0: normal priority (A/B).
void* make_stream(bool high) {
  hipStream_t s = nullptr;
  int least = 0, greatest = 0;
  if (high && switches().enc_priority && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest < least) {
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest) == hipSuccess) return s;
    (void)hipGetLastError();
  }
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return s;
}
void encoder_stream_set_override(void* s) { t_encoder_stream = reinterpret_cast<hipStream_t>(s); }
hipStream_t encoder_stream_acquire(int device) {
  if (t_encoder_stream) return t_encoder_stream;
  DevPool& pl = pool_of(device);
  {
    std::lock_guard<std::mutex> g(pl.mu);
    if (!pl.idle_streams.empty()) { hipStream_t s = pl.idle_streams.back(); pl.idle_streams.pop_back(); return s; }
  }
  return static_cast<hipStream_t>(make_stream(true));  // nullptr = the NULL stream: still correct
}
void encoder_stream_release(int device, hipStream_t s) {
  if (!s || s == t_encoder_stream) return;
  DevPool& pl = pool_of(device);
  std::lock_guard<std::mutex> g(pl.mu);
  pl.idle_streams.push_back(s);
}
// The calling thread's current HIP device is not ours to change: entry points that run on a CALLER's thread (model upload / free,
// stream create / destroy, pool trim) switch to the replica's device and put the caller's back when they return.
struct DeviceGuard {
  int prev = -1;
  DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); } }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
// One stream for everything a host thread sends to the device during a stretch of work (a part of a search-mode call: its batches'
// encodes and runs, its lattices): taken from the encoder's pool, installed as the thread's encoder stream, handed back at the end.
void* thread_stream_begin(int device) {
  if (t_encoder_stream) return nullptr;  // the thread brought its own
  DeviceGuard guard;  // (the pool may have to create the stream: on the replica's device, not on the thread's current one)
  if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  hipStream_t s = encoder_stream_acquire(device);
  t_encoder_stream = s;
  return s;
}
void thread_stream_end(int device, void* s) {
  if (!s) return;
  t_encoder_stream = nullptr;
  encoder_stream_release(device, reinterpret_cast<hipStream_t>(s));
}
// streams of the replicas of a multi-device model (capi.cpp owns them; HIP stays behind this file)
void* stream_create(int device, std::string& err, bool high_priority) {
  DeviceGuard guard;
  void* s = nullptr;
  if (hipSetDevice(device) != hipSuccess || !(s = make_stream(high_priority))) {
    err = std::string("hipStreamCreate: ") + hipGetErrorString(hipGetLastError());
    return nullptr;
  }
  return s;
}
void stream_destroy(int device, void* s) {
  DeviceGuard guard;
  if (!s) return;
  (void)hipSetDevice(device);
  (void)hipStreamDestroy(reinterpret_cast<hipStream_t>(s));
}

// Blocks the cache let go (pool_free) wait here for a call that can afford hipFree -- it waits for the whole device, and pool_free runs
// on the hot path: a pipeline of fresh batches whose (un-hinted, 7 GB) first runs had filled the cache freed a block per batch and stalled
// its fetch thread behind the runs in flight: 273 instead of 355 M queries/s in bench.py's end-to-end section (round 6).
static void pool_reap(DevPool& pl) {
  std::vector<std::pair<void*, size_t>> drop;
  { std::lock_guard<std::mutex> g(pl.mu); drop.swap(pl.doomed); pl.doomed_bytes = 0; }
  for (auto& d : drop) (void)hipFree(d.first);
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
      pl.freed_at.erase(*p);
      return hipSuccess;
    }
  }
  pool_reap(pl);  // (a cache miss pays a hipMalloc anyway: the evicted blocks go to the driver here)
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) {  // out of memory: give the cache back and retry once
    std::vector<void*> drop;
    {
      std::lock_guard<std::mutex> g(pl.mu);
      for (auto& kv : pl.free_blocks) { drop.push_back(kv.second); pl.size_of.erase(kv.second); }
      pl.free_blocks.clear();
      pl.freed_at.clear();
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
  std::vector<void*> evicted;
  bool kept = false;
  {
    std::lock_guard<std::mutex> g(pl.mu);
    auto it = pl.size_of.find(p);
    if (it != pl.size_of.end() && it->second <= pool_cache_limit()) {
      // A full cache makes room by letting its OLDEST blocks go.  (Until the end of round 5 the block that came back was handed to
      // hipFree instead -- a call that waits for the device: after a workload that had filled the cache -- bench.py's 1 M-entry
      // lexicon -- every call of the next one paid its frees and mallocs in full: 18-22 ms per part of a search-mode call.)
      while (pl.cached + it->second > pool_cache_limit() && !pl.free_blocks.empty()) {
        auto old = pl.free_blocks.begin();
        for (auto jt = pl.free_blocks.begin(); jt != pl.free_blocks.end(); ++jt)
          if (pl.freed_at[jt->second] < pl.freed_at[old->second]) old = jt;
        pl.cached -= old->first;
        pl.doomed.emplace_back(old->second, old->first);   // freed by the next cache miss / trim (pool_reap), not here
        pl.doomed_bytes += old->first;
        pl.freed_at.erase(old->second);
        pl.size_of.erase(old->second);
        pl.free_blocks.erase(old);
      }
      if (pl.doomed_bytes > pool_cache_limit() / 2) { evicted.reserve(pl.doomed.size()); for (auto& d : pl.doomed) evicted.push_back(d.first); pl.doomed.clear(); pl.doomed_bytes = 0; }  // (a bound all the same)
      pl.free_blocks.emplace(it->second, p);
      pl.freed_at[p] = ++pl.clock;
      pl.cached += it->second;
      kept = true;
    } else if (it != pl.size_of.end()) {
      pl.size_of.erase(it);
    }
  }
  for (void* q : evicted) (void)hipFree(q);
  if (!kept) (void)hipFree(p);
}
static void pool_trim(int device) {
  DevPool& pl = pool_of(device);
  small_ctxs_destroy(device);  // (first: their blocks go back to the pool that is emptied below)
  pool_reap(pl);
  std::vector<void*> drop;
  {
    std::lock_guard<std::mutex> g(pl.mu);
    for (auto& kv : pl.free_blocks) { drop.push_back(kv.second); pl.size_of.erase(kv.second); }
    pl.free_blocks.clear();
    pl.freed_at.clear();
    pl.cached = 0;
  }
  for (void* d : drop) (void)hipFree(d);
  shells_destroy(device);
}
// Pinned host buffers for the downloaded results.  A fresh malloc'd buffer of 141 MB (1 M queries of config 2) is pageable and
// untouched: the D2H copy is staged and page-faults its way through it (~20 ms); a pinned buffer takes the rows at PCIe speed.
// Pinning costs more than the copy, so freed buffers are kept (anx_results_free returns them here) up to ANX_PINNED_CACHE_MB
// (default 2048: a pipeline of 6-8 batches of a million queries in flight, 75-140 MB of rows each, outgrew 1024 and pinned afresh
// for every batch at half the rate); anx_device_pool_trim releases them.  Without a HIP device the buffers are plain malloc blocks.
namespace {
struct HostCache {
  std::mutex mu;
  std::multimap<size_t, void*> free_blocks;
  std::unordered_map<void*, std::pair<size_t, bool>> live;  // every block handed out or cached: (bytes, pinned)
  std::unordered_map<void*, uint64_t> freed_at;             // cached blocks: when they came back (the oldest leave first)
  uint64_t clock = 0;
  size_t cached = 0;
  std::vector<void*> doomed;   // evicted pinned blocks awaiting hipHostFree (which waits for the device: done on a cache miss, not on the hot path)
  size_t doomed_bytes = 0;
};
HostCache& host_cache() { static HostCache c; return c; }
size_t host_cache_limit() {
  static const size_t lim = []() { const char* e = getenv("ANX_PINNED_CACHE_MB"); const long long v = e ? atoll(e) : -1; return v >= 0 ? (size_t)v << 20 : (size_t)2 << 30; }();
  return lim;
}
}  // namespace
static std::atomic<uint64_t> g_host_hits{0}, g_host_misses{0}, g_host_miss_bytes{0};
void host_result_cache_stats(uint64_t* hits, uint64_t* misses, uint64_t* miss_bytes) { *hits = g_host_hits.load(); *misses = g_host_misses.load(); *miss_bytes = g_host_miss_bytes.load(); }
void* host_result_alloc(size_t bytes) {
  HostCache& hc = host_cache();
  bytes = (std::max<size_t>(bytes, 64) + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);  // whole MB: a few distinct sizes
  {
    std::lock_guard<std::mutex> g(hc.mu);
    auto it = hc.free_blocks.lower_bound(bytes);
    if (it != hc.free_blocks.end() && it->first <= 2 * bytes) {
      void* p = it->second;
      hc.cached -= it->first;
      hc.free_blocks.erase(it);
      hc.freed_at.erase(p);
      ++g_host_hits;
      return p;
    }
  }
  ++g_host_misses;
  g_host_miss_bytes += bytes;
  {  // a miss pins a fresh block anyway: the evicted ones are released here
    std::vector<void*> drop;
    { std::lock_guard<std::mutex> g(hc.mu); drop.swap(hc.doomed); hc.doomed_bytes = 0; }
    for (void* q : drop) (void)hipHostFree(q);
  }
  void* p = nullptr;
  bool pinned = hipHostMalloc(&p, bytes, hipHostMallocPortable) == hipSuccess && p;  // portable: the replicas of a multi-device model download into one buffer
  if (!pinned) {
    (void)hipGetLastError();
    p = malloc(bytes);
    if (!p) return nullptr;
  }
  std::lock_guard<std::mutex> g(hc.mu);
  hc.live[p] = std::make_pair(bytes, pinned);
  return p;
}
bool host_result_is_pinned(void* p) {
  HostCache& hc = host_cache();
  std::lock_guard<std::mutex> g(hc.mu);
  auto it = hc.live.find(p);
  return it != hc.live.end() && it->second.second;
}
void host_result_free(void* p) {
  if (!p) return;
  HostCache& hc = host_cache();
  size_t bytes = 0;
  bool pinned = false, known = false;
  std::vector<void*> evicted;
  {
    std::lock_guard<std::mutex> g(hc.mu);
    auto it = hc.live.find(p);
    if (it != hc.live.end()) {
      known = true;
      bytes = it->second.first;
      pinned = it->second.second;
      if (pinned && bytes <= host_cache_limit()) {
        // A full cache makes room by letting its OLDEST blocks go.  (Until the end of round 5 the block that came back was dropped
        // instead: a process that had filled the cache with the buffers of one workload -- bench.py after its 1 M-entry lexicon --
        // then pinned and unpinned the buffers of the next one on every call: search mode 225-240 instead of 250-280 MB/s.)
        while (hc.cached + bytes > host_cache_limit() && !hc.free_blocks.empty()) {
          auto old = hc.free_blocks.begin();
          for (auto jt = hc.free_blocks.begin(); jt != hc.free_blocks.end(); ++jt)
            if (hc.freed_at[jt->second] < hc.freed_at[old->second]) old = jt;
          hc.cached -= old->first;
          hc.doomed.push_back(old->second);
          hc.doomed_bytes += old->first;
          hc.freed_at.erase(old->second);
          hc.live.erase(old->second);
          hc.free_blocks.erase(old);
        }
        if (hc.doomed_bytes > host_cache_limit()) { evicted.swap(hc.doomed); hc.doomed_bytes = 0; }  // (a bound all the same)
        hc.free_blocks.emplace(bytes, p);
        hc.freed_at[p] = ++hc.clock;
        hc.cached += bytes;
        known = false;  // (cached: nothing to release below)
        p = nullptr;
      } else {
        hc.live.erase(it);
      }
    }
  }
  for (void* q : evicted) (void)hipHostFree(q);
  if (!p) return;
  if (known && pinned) (void)hipHostFree(p);
  else free(p);  // a malloc block (also: rows assembled by anx_find_variants_batch from several device batches)
}
static void host_cache_trim() {
  HostCache& hc = host_cache();
  std::vector<void*> drop;
  {
    std::lock_guard<std::mutex> g(hc.mu);
    for (auto& kv : hc.free_blocks) { drop.push_back(kv.second); hc.live.erase(kv.second); }
    for (void* q : hc.doomed) drop.push_back(q);
    hc.doomed.clear();
    hc.doomed_bytes = 0;
    hc.free_blocks.clear();
    hc.freed_at.clear();
    hc.cached = 0;
  }
  for (void* p : drop) (void)hipHostFree(p);
}
void device_pool_trim(int device) {
  int cur = -1;
  const bool have_cur = hipGetDevice(&cur) == hipSuccess;
  if (hipSetDevice(device) == hipSuccess) pool_trim(device);
  else (void)hipGetLastError();
  if (have_cur && cur != device) (void)hipSetDevice(cur);  // the caller's current device is not ours to change
  host_cache_trim();
}

// Host staging array WITHOUT value-initialisation: the threaded fill loops write every element, so the pages are first
// touched (and faulted in) by the worker threads instead of being zero-filled by the calling thread (a million queries
// stage ~150 MB; the single-threaded zero fill of std::vector cost a quarter of the encode time).
template <typename T>
struct HostBuf {
  T* p = nullptr;
  size_t n = 0;
  explicit HostBuf(size_t count) : p(static_cast<T*>(malloc(std::max<size_t>(count, 1) * sizeof(T)))), n(count) {}
  ~HostBuf() { free(p); }
  HostBuf(const HostBuf&) = delete;
  HostBuf& operator=(const HostBuf&) = delete;
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
  T* data() { return p; }
  size_t size() const { return n; }
};

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

DeviceLexicon* lexicon_upload(const LexiconImage& img, const EncodeTables& et, const AdjIndex* adj, int device, std::string& err, int dev_closure, size_t dev_budget,
                              AdjIndex* dev_stats) {
  DeviceGuard guard;
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
  std::vector<uint4> erec(2 * img.ent_vocab.size());
  for (size_t e = 0; e < img.ent_vocab.size(); ++e) {
    memcpy(&erec[2 * e], &img.rows[(size_t)img.ent_rowoff[e] * 16], 16);
    erec[2 * e + 1] = make_uint4(img.ent_meta[e], img.ent_rowoff[e], img.ent_freq[e], 0u);
  }
  std::vector<uint4> eplanes(img.ent_vocab.size());   // {meta, symbol planes of the first 16 symbols}
  for (size_t e = 0; e < img.ent_vocab.size(); ++e) {
    uint32_t pw[3];
    symbol_planes16(&img.rows[(size_t)img.ent_rowoff[e] * 16], img.ent_meta[e] & 0xFFu, pw);
    eplanes[e] = make_uint4(img.ent_meta[e], pw[0], pw[1], pw[2]);
  }
  // scan records of the bit-plane kernel: one per ENTRY (class-major, the order of the entry ids) carrying the planes of its
  // class, so that a scan hit is a (query, entry) pair -- classes with several entries (8 % of eng.aspell) are tested once per
  // entry, and the hit expansion has no entries-per-class loop (it ran as long as the largest class among 64 hits)
  std::vector<uint4> srec((size_t)img.nentries + 1, make_uint4(0u, 0u, 0u, 0u));   // {plane 1, plane 2, len, class}
  std::vector<uint2> srec34((size_t)img.nentries + 1, make_uint2(0u, 0u));         // {plane 3, plane 4}
  for (uint32_t c = 0; c < img.nclasses; ++c)
    for (uint32_t e = img.cls_off[c]; e < img.cls_off[c + 1]; ++e) {
      srec[e] = make_uint4(img.cls_bits[c], img.cls_bits[(size_t)img.cstride + c], img.cls_len[c], c);
      srec34[e] = make_uint2(img.cls_bits[2 * (size_t)img.cstride + c], img.cls_bits[3 * (size_t)img.cstride + c]);
    }
  srec[img.nentries] = make_uint4(0u, 0u, 255u, 0xFFFFFFFFu);  // the padding record
  std::vector<uint4> sig2(img.sig_lo.size());  // {signature, first class of the run, classes in the run}
  for (size_t i = 0; i < sig2.size(); ++i)
    sig2[i] = make_uint4(img.sig_lo[i], img.sig_hi[i], img.sig_cbeg[i], i + 1 < img.sig_cbeg.size() ? img.sig_cbeg[i + 1] - img.sig_cbeg[i] : 0u);
  std::vector<uint32_t> off = img.cls_off;
  if (off.empty()) off.push_back(0);
  // Signature hash table + L1 balls of signature offsets: instead of walking the whole +-k charcount window of the signature
  // table (thousands of steps on a large lexicon) a scan tile enumerates the offsets d with sum |d_g| <= k, adds each to its own
  // signature and probes the table: 377 probes for 6 groups and k = 3, whatever the size of the lexicon.
  uint32_t hmask = 64;
  while (hmask < 4 * (uint32_t)img.nsigs) hmask <<= 1;
  std::vector<uint4> shash, shash_e;  // {sig lo, sig hi, first class / first entry of the run, classes / entries}; count 0 = empty slot
  for (;; hmask <<= 1) {  // every key within 16 probes of its home slot
    shash.assign(hmask, make_uint4(0u, 0u, 0u, 0u));
    shash_e.assign(hmask, make_uint4(0u, 0u, 0u, 0u));
    bool ok = true;
    for (uint32_t i = 0; i < img.nsigs && ok; ++i) {
      uint32_t h = sig_hash(img.sig_lo[i], img.sig_hi[i]) & (hmask - 1);
      int probes = 0;
      while (shash[h].w && probes < 16) { h = (h + 1) & (hmask - 1); ++probes; }
      if (probes == 16) { ok = false; break; }
      const uint32_t c0 = img.sig_cbeg[i], c1 = img.sig_cbeg[i + 1];  // a signature's run holds at least one class, a class one entry
      shash[h] = make_uint4(img.sig_lo[i], img.sig_hi[i], c0, c1 - c0);
      shash_e[h] = make_uint4(img.sig_lo[i], img.sig_hi[i], img.cls_off[c0], img.cls_off[c1] - img.cls_off[c0]);
    }
    if (ok) break;
  }
  d->hash_mask = hmask - 1;
  std::vector<unsigned long long> ball;
  {
    int ng = 1;
    for (uint8_t g : img.sym_group) ng = std::max(ng, (int)g + 1);
    for (int k = 0; k <= 12; ++k) {
      std::vector<unsigned long long> cur;
      int8_t dv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      bool too_big = false;
      std::function<void(int, int)> rec = [&](int g, int left) {
        if (too_big) return;
        if (g == ng) {
          unsigned long long v = 0;
          for (int i = 0; i < 8; ++i) v |= (unsigned long long)(uint8_t)dv[i] << (8 * i);
          cur.push_back(v);
          if (cur.size() > BALL_MAX) too_big = true;
          return;
        }
        for (int x = -left; x <= left; ++x) { dv[g] = (int8_t)x; rec(g + 1, left - (x < 0 ? -x : x)); }
        dv[g] = 0;
      };
      rec(0, k);
      d->ball_off[k] = (uint32_t)ball.size();
      d->ball_n[k] = too_big ? 0u : (uint32_t)cur.size();
      if (!too_big) ball.insert(ball.end(), cur.begin(), cur.end());
    }
  }
  std::vector<uint32_t> btab(26);
  for (int k = 0; k <= 12; ++k) { btab[k] = d->ball_off[k]; btab[13 + k] = d->ball_n[k]; }
  std::vector<uint4> sig2e(sig2.size());  // {signature, first entry of the run, entries in the run}
  for (size_t i = 0; i < sig2.size(); ++i) {
    const uint32_t c0 = std::min(img.sig_cbeg[i], img.nclasses), c1 = std::min(i + 1 < img.sig_cbeg.size() ? img.sig_cbeg[i + 1] : img.nclasses, img.nclasses);
    sig2e[i] = make_uint4(img.sig_lo[i], img.sig_hi[i], off[c0], off[c1] - off[c0]);
  }
  if ((rc = upload(&d->cls_planes, img.cls_planes.data(), img.cls_planes.size(), err, &d->bytes)) ||
      (rc = upload(&d->cls_bits, img.cls_bits.data(), img.cls_bits.size(), err, &d->bytes)) ||
      (rc = upload(&d->cls_len, img.cls_len.data(), img.cls_len.size(), err, &d->bytes)) ||
      (rc = upload(&d->cls_off, off.data(), off.size(), err, &d->bytes)) ||
      (rc = upload(&d->scan_rec, srec.data(), srec.size(), err, &d->bytes)) ||
      (rc = upload(&d->scan_rec34, srec34.data(), srec34.size(), err, &d->bytes)) ||
      (rc = upload(&d->sig_e, sig2e.data(), sig2e.size(), err, &d->bytes)) ||
      (rc = upload(&d->sighash, shash.data(), shash.size(), err, &d->bytes)) ||
      (rc = upload(&d->sighash_e, shash_e.data(), shash_e.size(), err, &d->bytes)) ||
      (rc = upload(&d->ball, ball.data(), ball.size(), err, &d->bytes)) ||
      (rc = upload(&d->ball_tab, btab.data(), btab.size(), err, &d->bytes)) ||
      (rc = upload(&d->sig, sig2.data(), sig2.size(), err, &d->bytes)) ||
      (rc = upload(&d->sig_cbeg, img.sig_cbeg.data(), img.sig_cbeg.size(), err, &d->bytes)) ||
      (rc = upload(&d->ent_vocab, img.ent_vocab.data(), img.ent_vocab.size(), err, &d->bytes)) ||
      (rc = upload(&d->ent_freq, img.ent_freq.data(), img.ent_freq.size(), err, &d->bytes)) ||
      (rc = upload(&d->ent_meta, img.ent_meta.data(), img.ent_meta.size(), err, &d->bytes)) ||
      (rc = upload(&d->ent_rowoff, img.ent_rowoff.data(), img.ent_rowoff.size(), err, &d->bytes)) ||
      (rc = upload(&d->ent_order, img.ent_order.data(), img.ent_order.size(), err, &d->bytes)) ||
      (rc = upload(&d->ent_rec, rec.data(), rec.size(), err, &d->bytes)) ||
      (rc = upload(&d->e_rec, erec.data(), erec.size(), err, &d->bytes)) ||
      (rc = upload(&d->e_planes, eplanes.data(), eplanes.size(), err, &d->bytes)) ||
      (rc = upload(&d->ent_var_off, img.ent_var_off.data(), img.ent_var_off.size(), err, &d->bytes)) ||
      (rc = upload(&d->var_target, img.var_target.data(), img.var_target.size(), err, &d->bytes)) ||
      (rc = upload(&d->var_target_freq, img.var_target_freq.data(), img.var_target_freq.size(), err, &d->bytes)) ||
      (rc = upload(&d->var_score, img.var_score.data(), img.var_score.size(), err, &d->bytes)) ||
      (rc = upload(reinterpret_cast<uint8_t**>(&d->rows), img.rows.data(), img.rows.size(), err, &d->bytes)) ||
      // tables of the device-side query encoder (encode.hip): flattened alphabet, symbol groups, lowercase ranges
      (rc = upload(&d->alpha.fast, et.fast, 256, err, &d->bytes)) || (rc = upload(&d->alpha.coff, et.coff, 257, err, &d->bytes)) ||
      (rc = upload(reinterpret_cast<uint32_t**>(&d->alpha.cand), et.cand.data(), et.cand.size(), err, &d->bytes)) ||
      (rc = upload(&d->alpha.bytes, et.bytes.data(), et.bytes.size(), err, &d->bytes)) ||
      (rc = upload(&d->alpha.sym_group, img.sym_group.data(), img.sym_group.size(), err, &d->bytes)) ||
      (rc = upload(reinterpret_cast<uint32_t**>(&d->alpha.lower), et.lower.data(), et.lower.size(), err, &d->bytes)) ||
      (rc = upload(&d->alpha.siglen_begin, img.siglen_begin, kMaxSymbols + 2, err, &d->bytes))) {
    lexicon_free(d);
    return nullptr;
  }
  d->alpha.nlower = (uint32_t)(et.lower.size() / 2);
  {  // x / L for x, L <= 32 as the host's IEEE division computes it (ScoreArgs::quot; until round 5 uploaded again with every batch)
    std::vector<double> quot(33 * 33, 0.0);
    for (int x = 0; x <= 32; ++x)
      for (int L = 1; L <= 32; ++L) {
        volatile double num = (double)x, den = (double)L;  // a real division at run time, as the reference does
        quot[(size_t)x * 33 + L] = num / den;
      }
    if ((rc = upload(&d->quot, quot.data(), quot.size(), err, &d->bytes))) { lexicon_free(d); return nullptr; }
  }
  if (adj && !adj->hash.empty() && img.nsym <= 32) {  // signature adjacency lists (adjacency.h): streamed by the bit-plane scan
    static_assert(sizeof(AdjSlot) == sizeof(uint4) && sizeof(AdjHdr) == 32 && sizeof(AdjPlanes) == sizeof(uint2), "adjacency records");
    if ((rc = upload(reinterpret_cast<AdjSlot**>(&d->adj_hash), adj->hash.data(), adj->hash.size(), err, &d->bytes)) ||
        (rc = upload(reinterpret_cast<AdjHdr**>(&d->adj_hdr), adj->hdr.data(), adj->hdr.size(), err, &d->bytes)) ||
        (rc = upload(reinterpret_cast<AdjPlanes**>(&d->adj_planes), adj->planes, adj->rows * kAdjRow, err, &d->bytes)) ||
        (rc = upload(&d->adj_ids, adj->ids, adj->rows * kAdjRow, err, &d->bytes))) {
      lexicon_free(d);
      return nullptr;
    }
    d->adj_mask = adj->hash_mask;
    d->adj_nhdr = (uint32_t)adj->hdr.size();
    d->adj_hash_host = adj->hash;
    d->adj_hdr_host = adj->hdr;
  } else if (!adj && dev_closure >= 0 && img.nsym <= 32) {  // built here, on the device, from the tables just uploaded (adjacency.hip)
    AdjIndex local;
    if ((rc = adjacency_build_device(d, img, dev_closure, dev_budget, dev_stats ? *dev_stats : local, err))) {
      // the lists only accelerate the scan (k_scan_bits probes the ball of a tile without a list): a build that failed -- out of
      // memory on a busy device, as a rule -- leaves a working replica without lists
      fprintf(stderr, "[anx] device %d: signature adjacency lists not built (%s): the scan probes every tile's ball itself\n", device, err.c_str());
      (void)hipGetLastError();
      for (void** p : {(void**)&d->adj_hash, (void**)&d->adj_hdr, (void**)&d->adj_planes, (void**)&d->adj_ids})
        if (*p) { pool_free(*p); *p = nullptr; }
      d->adj_mask = 0;
      d->adj_nhdr = 0;
      if (dev_stats) { dev_stats->nsig_kept = 0; dev_stats->rows = 0; dev_stats->records = 0; }
      err.clear();
    }
  }
  return d;
}

void lexicon_free(DeviceLexicon* d) {
  if (!d) return;
  DeviceGuard guard;
  (void)hipSetDevice(d->device);
  for (void* p : {(void*)d->cls_planes, (void*)d->cls_bits, (void*)d->cls_len, (void*)d->cls_off, (void*)d->scan_rec, (void*)d->scan_rec34, (void*)d->sig_e, (void*)d->sighash, (void*)d->sighash_e, (void*)d->ball, (void*)d->ball_tab, (void*)d->sig, (void*)d->sig_cbeg, (void*)d->ent_vocab,
                  (void*)d->ent_freq, (void*)d->ent_meta, (void*)d->ent_rowoff, (void*)d->ent_order, (void*)d->ent_rec, (void*)d->e_rec, (void*)d->e_planes, (void*)d->ent_var_off,
                  (void*)d->var_target, (void*)d->var_target_freq, (void*)d->var_score, (void*)d->rows, (void*)d->alpha.fast, (void*)d->alpha.coff,
                  (void*)d->alpha.cand, (void*)d->alpha.bytes, (void*)d->alpha.sym_group, (void*)d->alpha.lower, (void*)d->alpha.siglen_begin,
                  (void*)d->adj_hash, (void*)d->adj_hdr, (void*)d->adj_planes, (void*)d->adj_ids, (void*)d->quot})
    if (p) pool_free(p);
  conf_free(d->dconf);
  lm_free(d->dlm);
  bool last;
  { DevPool& pl = pool_of(d->device); std::lock_guard<std::mutex> g(pl.mu); last = --pl.lexicons <= 0; }
  if (last) pool_trim(d->device);  // the last model of this device: hand the cached blocks back to the driver
  delete d;
}

static int scan_mode() { return switches().scan_sad; }  // ANX_SCAN=sad forces the general count-vector kernel (A/B testing)

// The threaded HOST encoder (ANX_ENCODE=host): the A/B reference of the device-side encoder in encode.hip, same arrays.
static int encode_host(const HostModel& m, const DeviceLexicon* dl, Batch* b, const char* const* utf8, size_t n, const anx_params& p,
                       std::string& err) {
  const bool timing = switches().encode_timing != 0;
  auto tnow = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_prev = tnow();
  auto lap = [&](const char* what) { if (timing) { const double t = tnow(); fprintf(stderr, "[anx encode] %-28s %8.2f ms\n", what, (t - t_prev) * 1e3); t_prev = t; } };
  b->status.assign(n, 0);
  const int NP = dl->nplanes;
  const bool bits_ok = dl->nsym <= 32 && !scan_mode();
  // ---- host encoding, threaded: normalisation (src/anahash.rs:50-80), count vector, threshold clamps -------
  struct Enc {
    uint32_t meta;   // len | k<<8 | d<<16 | first_is_lower<<24 ; 0 = not encodable
    uint32_t key;    // (bit-plane kinds ? 256 : 0) + len : bucket for the counting sort
    uint32_t kind;   // 0 = count-vector (SAD) scan, 1..NBITPLANES = planes the bit-plane scan compares for this query
    uint32_t off;    // offset of the norm string in its thread's arena
    uint16_t thread;
    uint64_t sig;    // per-group symbol counts (LexiconImage::sym_group)
  };
  HostBuf<Enc> enc(n);
  unsigned nthreads = std::max(1u, std::min(32u, usable_hw_threads()));
  if (n < 4096) nthreads = 1;
  std::vector<std::vector<uint8_t>> arena(nthreads);
  const int A = m.alphabet.size();
  const size_t cvbytes = (size_t)NP * 4;
  HostBuf<uint8_t> cv_all(n * cvbytes);
  auto encode_range = [&](unsigned tid, size_t lo, size_t hi) {
    std::vector<uint8_t>& ar = arena[tid];
    ar.reserve((hi - lo) * 12);
    int16_t codes[kMaxSymbols];
    for (size_t i = lo; i < hi; ++i) {
      Enc& e = enc[i];
      e.meta = 0; e.key = 0; e.kind = 0; e.off = 0; e.thread = (uint16_t)tid; e.sig = 0;
      memset(&cv_all[i * cvbytes], 0, cvbytes);
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
      e.kind = kind;
      e.key = (kind ? 256u : 0u) + (uint32_t)len;
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
  // (scan kernel, length)-bucketed order, stable (counting sort): a bucket shares k, d and the class window
  constexpr uint32_t NKEYS = 2 * 256;
  std::vector<size_t> kstart(NKEYS + 1, 0);
  size_t maxlen = 1;
  // per-thread histograms over contiguous input ranges -> per-thread cursors: a parallel, stable counting sort
  std::vector<std::vector<size_t>> hist(nthreads, std::vector<size_t>(NKEYS, 0));
  std::vector<size_t> t_maxlen(nthreads, 1);
  std::vector<uint32_t> t_dmax(nthreads, 0);
  auto run_threads = [&](const std::function<void(unsigned, size_t, size_t)>& f, size_t count) {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthreads; ++t) {
      const size_t lo = count * t / nthreads, hi = count * (t + 1) / nthreads;
      if (nthreads == 1) f(0, lo, hi);
      else th.emplace_back(f, t, lo, hi);
    }
    for (auto& x : th) x.join();
  };
  run_threads([&](unsigned tid, size_t lo, size_t hi) {
    std::vector<size_t>& h = hist[tid];
    size_t ml = 1;      // thread-local: neighbouring elements of t_maxlen / t_dmax share a cache line
    uint32_t dm = 0;
    for (size_t i = lo; i < hi; ++i)
      if (enc[i].meta) {
        h[enc[i].key]++;
        ml = std::max<size_t>(ml, enc[i].meta & 0xFF);
        dm = std::max<uint32_t>(dm, (enc[i].meta >> 16) & 0xFF);
      }
    t_maxlen[tid] = ml;
    t_dmax[tid] = dm;
  }, n);
  for (unsigned t = 0; t < nthreads; ++t) {
    maxlen = std::max(maxlen, t_maxlen[t]);
    b->dmax = std::max(b->dmax, t_dmax[t]);
  }
  {
    size_t run = 0;
    for (uint32_t kx = 0; kx < NKEYS; ++kx) {
      kstart[kx] = run;
      for (unsigned t = 0; t < nthreads; ++t) {  // thread t's items of this key follow those of the threads before it
        const size_t c = hist[t][kx];
        hist[t][kx] = run;
        run += c;
      }
    }
    kstart[NKEYS] = run;
  }
  const size_t nq = kstart[NKEYS];
  b->nq = nq;
  b->qw = (uint32_t)((maxlen + 15) / 16);
  HostBuf<uint32_t> h_cv(nq * (size_t)NP), h_bits(nq * (size_t)NBITPLANES), h_meta(nq), h_orig(nq), h_kind(nq);
  HostBuf<uint64_t> h_sig(nq);
  HostBuf<uint8_t> h_rows(nq * (size_t)b->qw * 16);
  HostBuf<uint4> h_qrec(2 * nq);
  b->order.resize(nq);
  run_threads([&](unsigned tid, size_t lo, size_t hi) {
    std::vector<size_t>& cursor = hist[tid];
    for (size_t i = lo; i < hi; ++i)
      if (enc[i].meta) b->order[cursor[enc[i].key]++] = (uint32_t)i;
  }, n);
  lap("counting sort");
  {  // inside a (scan kernel, length) bucket: by (signature, kind), stable; buckets are independent -> threads take them
     // round-robin.  LSD radix sort (8-bit digits, digits on which the whole bucket agrees are skipped; the kind is the
     // least significant digit) over (signature, kind, input index) records: contiguous keys instead of a comparison sort
     // through enc[] (23 -> 8 ms per million queries)
    struct SigIdx { uint64_t sig; uint32_t idx; uint32_t kind; };
    auto sort_buckets = [&](unsigned tid) {
      std::vector<SigIdx> a, t2;
      for (uint32_t kx = tid; kx < NKEYS; kx += nthreads) {
        const size_t b0 = kstart[kx], cnt = kstart[kx + 1] - b0;
        if (cnt < 2) continue;
        a.resize(cnt);
        t2.resize(cnt);
        uint64_t all_or = 0, all_and = ~0ull;
        uint32_t kind_or = 0, kind_and = ~0u;
        for (size_t i = 0; i < cnt; ++i) {
          const uint32_t x = b->order[b0 + i];
          a[i] = SigIdx{enc[x].sig, x, enc[x].kind};
          all_or |= a[i].sig;
          all_and &= a[i].sig;
          kind_or |= a[i].kind;
          kind_and &= a[i].kind;
        }
        const uint64_t varying = all_or ^ all_and;  // bits on which some keys differ
        SigIdx *src = a.data(), *dst = t2.data();
        if (kind_or != kind_and) {  // least significant digit: the kind (0..NBITPLANES)
          size_t cnts[NBITPLANES + 2] = {0};
          for (size_t i = 0; i < cnt; ++i) cnts[src[i].kind + 1]++;
          for (int v = 0; v <= NBITPLANES; ++v) cnts[v + 1] += cnts[v];
          for (size_t i = 0; i < cnt; ++i) dst[cnts[src[i].kind]++] = src[i];
          std::swap(src, dst);
        }
        for (int byte = 0; byte < 8; ++byte) {
          if (!((varying >> (8 * byte)) & 0xFF)) continue;
          size_t cnts[257] = {0};
          for (size_t i = 0; i < cnt; ++i) cnts[((src[i].sig >> (8 * byte)) & 0xFF) + 1]++;
          for (int v = 0; v < 256; ++v) cnts[v + 1] += cnts[v];
          for (size_t i = 0; i < cnt; ++i) dst[cnts[(src[i].sig >> (8 * byte)) & 0xFF]++] = src[i];
          std::swap(src, dst);
        }
        for (size_t i = 0; i < cnt; ++i) b->order[b0 + i] = src[i].idx;
      }
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
      if (s + 8 < hi) {  // the sorted order gathers enc / cv_all / arena at random: prefetch a few queries ahead
        const size_t ip = b->order[s + 8];
        __builtin_prefetch(&enc[ip]);
        __builtin_prefetch(&cv_all[ip * cvbytes]);
      }
      if (s + 4 < hi) { const Enc& ep = enc[b->order[s + 4]]; __builtin_prefetch(&arena[ep.thread][ep.off]); }
      const size_t i = b->order[s];
      const Enc& e = enc[i];
      const uint8_t* cv = &cv_all[i * cvbytes];
      memcpy(&h_cv[s * (size_t)NP], cv, cvbytes);
      for (uint32_t tp = 0; tp < (uint32_t)NBITPLANES; ++tp) h_bits[s * NBITPLANES + tp] = 0;
      for (size_t sym = 0; sym < cvbytes && sym < 32; ++sym) {
        const uint32_t c = cv[sym];  // thermometer code: plane tp has the bit iff count > tp
        for (uint32_t tp = 0; tp < c && tp < (uint32_t)NBITPLANES; ++tp) h_bits[s * NBITPLANES + tp] |= 1u << sym;
      }
      uint8_t* row = &h_rows[s * (size_t)b->qw * 16];
      memset(row, 0xFE, (size_t)b->qw * 16);  // padding that equals nothing (kernels_score.hpp)
      memcpy(row, &arena[e.thread][e.off], e.meta & 0xFF);
      memcpy(&h_qrec[2 * s], row, 16);  // 32-B query record: first 16 symbols + meta
      {
        uint32_t pw[3];
        symbol_planes16(reinterpret_cast<const uint8_t*>(row), e.meta & 0xFFu, pw);
        h_qrec[2 * s + 1] = make_uint4(e.meta, pw[0], pw[1], pw[2]);
      }
      h_meta[s] = e.meta;
      h_orig[s] = (uint32_t)i;
      h_kind[s] = e.kind;
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
  // tiles: <= SCAN_TQ queries of one scan kernel, length and signature (sorted by kind inside); one wave of k_scan each
  for (size_t i = 0; i < nq;) {
    size_t j = i;
    const bool sad = h_kind[i] == 0;
    while (j < nq && (h_kind[j] == 0) == sad && (h_meta[j] & 0xFF) == (h_meta[i] & 0xFF) && h_sig[j] == h_sig[i]) ++j;
    const uint32_t lq = h_meta[i] & 0xFF, k = (h_meta[i] >> 8) & 0xFF;
    const int lo = std::max<int>(1, (int)lq - (int)k), hi = std::min<int>(kMaxSymbols, (int)lq + (int)k);
    // aligned to whole 64-signature blocks (the neighbours inside the edge blocks belong to charcounts outside the window)
    const uint32_t s0 = m.lex.siglen_begin[lo] & ~63u, s1 = (m.lex.siglen_begin[hi + 1] + 63u) & ~63u;
    uint32_t ngroups_ = 1;
    for (uint8_t g_ : m.lex.sym_group) ngroups_ = std::max<uint32_t>(ngroups_, (uint32_t)g_ + 1u);
    const uint32_t tq = switches().scan_tq ? (uint32_t)switches().scan_tq : default_scan_tq((int)ngroups_);
    // The count-vector (SAD) tiles are rare (queries with a symbol more than NBITPLANES times) and run as a launch of
    // their own: a handful of waves whose time is the latency of ONE wave walking the whole signature window.  Their
    // windows are therefore split over several waves (disjoint signature ranges = disjoint classes: same pairs).
    // probe the signature hash table with the L1 ball of offsets when that is cheaper than walking the window (engine: lexicon_upload)
    uint32_t ball0 = 0, balln = 0;
    if (probe_enabled() && k <= 12 && tile_probes(dl->ball_n[k], s1 - s0, h_sig[i])) { ball0 = dl->ball_off[k]; balln = dl->ball_n[k]; }
    const uint32_t nsplit = (sad && !balln) ? 8u : 1u;
    const uint32_t step = (((s1 - s0) + nsplit - 1) / nsplit + 63u) & ~63u;
    uint32_t adj = 0;  // the signature's adjacency list (adjacency.h), as the device encoder's adj_lookup finds it
    if (!sad && k <= (uint32_t)kAdjRadius && switches().scan_adj && dl->adj_mask && dl->adj_hash_host.empty()) {
      const int rca = adjacency_host_copies(dl, err);  // (lists built on the device: the table comes down once)
      if (rca) return rca;
    }
    if (!sad && k <= (uint32_t)kAdjRadius && switches().scan_adj && !dl->adj_hash_host.empty()) {
      uint32_t h = sig_hash((uint32_t)h_sig[i], (uint32_t)(h_sig[i] >> 32)) & dl->adj_mask;
      for (int pr = 0; pr < 17; ++pr) {
        const AdjSlot& e = dl->adj_hash_host[h];
        if (!e.hdr1) break;
        if (e.lo == (uint32_t)h_sig[i] && e.hi == (uint32_t)(h_sig[i] >> 32)) { adj = e.hdr1; break; }
        h = (h + 1u) & dl->adj_mask;
      }
    }
    for (size_t s = i; s < j; s += tq) {
      const uint32_t tn = (uint32_t)std::min<size_t>(tq, j - s);
      uint32_t ke[3] = {0, 0, 0};  // end of the kind-1 / kind-2 / kind-3 queries inside the tile
      for (uint32_t x = 0; x < tn; ++x)
        for (uint32_t kd = h_kind[s + x]; kd >= 1 && kd <= 3; ++kd) ke[kd - 1]++;
      const uint32_t kend = sad ? 0u : (ke[0] | ke[1] << 8 | ke[2] << 16);
      for (uint32_t part = 0; part < nsplit; ++part) {
        const uint32_t a0 = s0 + part * step, a1 = std::min(s1, a0 + step);
        if (a0 >= a1 && part) break;
        b->tiles.push_back(Tile{(uint32_t)s, tn, a0, a1, k, lq, (uint32_t)h_sig[i], (uint32_t)(h_sig[i] >> 32), sad ? 0u : 1u,
                                (h_meta[i] >> 16) & 0xFFu, kend, ball0, balln, adj, (s == i && part == 0u) ? 1u : 0u});
      }
    }
    i = j;
  }
  // longest-processing-time-first: cost ~ queries (the compatible classes per query vary little inside a length)
  // bit-plane tiles first, the SAD tiles (kind 0) after them: two launches
  auto tile_cost = [&](const Tile& x) {
    return x.adj ? (uint64_t)dl->adj_hdr_host[x.adj - 1].cum[kAdjSections - 1] * (8u + x.nq) + 16u : (uint64_t)x.nq * (x.s1 - x.s0 + 64);
  };
  std::stable_sort(b->tiles.begin(), b->tiles.end(), [&](const Tile& x, const Tile& y) {
    if ((x.kind == 0) != (y.kind == 0)) return y.kind == 0;
    if ((x.adj == 0) != (y.adj == 0)) return y.adj == 0;   // the tiles of k_scan_adj first
    return tile_cost(x) > tile_cost(y);
  });
  b->n_adj_tiles = 0;
  for (const Tile& t : b->tiles) { b->n_sad_tiles += t.kind == 0; b->n_adj_tiles += t.adj != 0; }
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
  if ((rc = upload(&b->q_rec, h_qrec.data(), h_qrec.size(), err, nullptr)) ||
      (rc = upload(&b->qexact, h_xcls.data(), nq, err, nullptr)) ||
      (rc = upload(&b->q_cv, h_cv.data(), h_cv.size(), err, nullptr)) ||
      (rc = upload(&b->q_bits, h_bits.data(), h_bits.size(), err, nullptr)) ||
      (rc = upload(reinterpret_cast<uint8_t**>(&b->q_rows), h_rows.data(), h_rows.size(), err, nullptr)) ||
      (rc = upload(&b->q_meta, h_meta.data(), nq, err, nullptr)) || (rc = upload(&b->q_orig, h_orig.data(), nq, err, nullptr)) ||
      (rc = upload(&b->d_tiles, b->tiles.data(), b->tiles.size(), err, nullptr)))
    return rc;
  b->ntiles = (uint32_t)b->tiles.size();
  std::vector<Tile>().swap(b->tiles);
  lap("uploads");
  return ANX_OK;
}

// events + pinned read-back block of a batch: from the device's pool of shells (freed batches), created when the pool is empty
static int shell_acquire(Batch* b, std::string& err) {
  DevPool& pl = pool_of(b->device);
  BatchShell sh;
  bool have = false;
  {
    std::lock_guard<std::mutex> g(pl.mu);
    if (!pl.shells.empty()) { sh = pl.shells.back(); pl.shells.pop_back(); have = true; }
  }
  if (!have) {
    for (auto& e : sh.ev) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipEventCreate(&sh.ev_fs0));
    HIP_TRY(hipEventCreate(&sh.ev_fs1));
    HIP_TRY(hipEventCreate(&sh.ev_scan0));
    HIP_TRY(hipEventCreate(&sh.ev_done));
    HIP_TRY(hipEventCreateWithFlags(&sh.ev_in, hipEventDisableTiming));
    // pinned: the counters the run reads back, and behind them the staging copy of FsCold (a pageable source would make the host
    // wait, in stream order behind the scan just enqueued, until the copy has been staged)
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&sh.h_read), HR_COLD_OFF + sizeof(FsCold), hipHostMallocDefault));
  }
  for (int i = 0; i < 6; ++i) b->ev[i] = sh.ev[i];
  b->ev_fs0 = sh.ev_fs0; b->ev_fs1 = sh.ev_fs1; b->ev_scan0 = sh.ev_scan0; b->ev_done = sh.ev_done; b->ev_in = sh.ev_in; b->h_read = sh.h_read;
  return ANX_OK;
}
static void shell_release(Batch* b) {
  if (!b->ev_done) return;  // never acquired (an encode that failed early)
  BatchShell sh;
  for (int i = 0; i < 6; ++i) { sh.ev[i] = b->ev[i]; b->ev[i] = nullptr; }
  sh.ev_fs0 = b->ev_fs0; sh.ev_fs1 = b->ev_fs1; sh.ev_scan0 = b->ev_scan0; sh.ev_done = b->ev_done; sh.ev_in = b->ev_in; sh.h_read = b->h_read;
  b->ev_fs0 = b->ev_fs1 = b->ev_scan0 = b->ev_done = b->ev_in = nullptr; b->h_read = nullptr;
  DevPool& pl = pool_of(b->device);
  std::lock_guard<std::mutex> g(pl.mu);
  pl.shells.push_back(sh);
}
static void shells_destroy(int device) {  // (the current device is `device`)
  DevPool& pl = pool_of(device);
  std::vector<BatchShell> drop;
  { std::lock_guard<std::mutex> g(pl.mu); drop.swap(pl.shells); }
  for (BatchShell& sh : drop) {
    for (auto& e : sh.ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : {sh.ev_fs0, sh.ev_fs1, sh.ev_scan0, sh.ev_done, sh.ev_in}) if (e) (void)hipEventDestroy(e);
    if (sh.h_read) (void)hipHostFree(sh.h_read);
  }
}

// what every batch needs besides its query arrays: counters, per-query accumulators, events
static int encode_tail(Batch* b, std::string& err) {
  const size_t nq = b->nq;
  int rc;
  const size_t nblk = (nq + SCAN_TILE - 1) / SCAN_TILE + 2;
  if ((rc = dalloc(&b->counters, CTR_N, err)) || (rc = dalloc(&b->rctr, SCAN_REGIONS * RC_STRIDE, err)) || (rc = dalloc(&b->sctr, SCAN_REGIONS * RC_STRIDE, err)) || (rc = dalloc(&b->lctr, 3 * SCAN_REGIONS * RC_STRIDE, err)) || (rc = dalloc(&b->qexpand, nq, err)) || (rc = dalloc(&b->qsurv, nq, err)) ||
      (rc = dalloc(&b->soff, nq + 1, err)) || (rc = dalloc(&b->qcur, nq, err)) || (rc = dalloc(&b->qmaxfreq, nq, err)) ||
      (rc = dalloc(&b->scan_tmp, nblk, err)) || (rc = dalloc(&b->r_count, nq, err)) || (rc = dalloc(&b->r_off, nq + 1, err)))
    return rc;
  return shell_acquire(b, err);
}

Batch* batch_encode_spans(const HostModel& m, const DeviceLexicon* dl, const char* blob, size_t blob_bytes, const uint32_t* off, size_t n,
                          const anx_params& p, std::string& err, int* code, bool keep_text, bool blob_on_device, bool after_stream, void* src_stream) {
  *code = ANX_OK;
  if (blob_on_device && (off || switches().encode_host)) { err = "inputs in device memory need the device-side encoder and no host offsets"; *code = ANX_EINVAL; return nullptr; }
  if (!dl) { err = "model is not resident on a device (no HIP device / anx_model_to_device not called)"; *code = ANX_ENODEVICE; return nullptr; }
  if (hipSetDevice(dl->device) != hipSuccess) { err = "hipSetDevice failed"; *code = ANX_ENODEVICE; return nullptr; }
  if (n >= (1u << 27)) { err = "more than 2^27 inputs per batch"; *code = ANX_ELIMIT; return nullptr; }
  Batch* b = new Batch();
  b->device = dl->device;
  b->params = p;
  b->n_input = n;
  b->keep_text = keep_text;
  int rc;
  if (switches().encode_host) {  // ANX_ENCODE=host: the threaded host encoder (A/B reference)
    std::vector<uint32_t> hoff;
    if (!off) {
      if (!packed_offsets(blob, blob_bytes, n, hoff)) { err = "packed inputs hold fewer strings than announced"; *code = ANX_EINVAL; batch_free(b); return nullptr; }
      off = hoff.data();
    }
    std::vector<const char*> ptrs(n);
    for (size_t i = 0; i < n; ++i) ptrs[i] = blob + off[i];  // every span is followed by a NUL byte
    rc = encode_host(m, dl, b, ptrs.data(), n, p, err);
    if (!rc && keep_text) {  // the host encoder does not upload the inputs: the device-side confusable weighting reads them
      const size_t bytes = n ? (size_t)off[n] : 0;
      std::vector<uint32_t> o(off, off + n + 1);
      if (!n) o.assign(1, 0u);
      if ((rc = upload(&b->d_text, blob, bytes, err, nullptr)) == 0) rc = upload(&b->d_textoff, o.data(), o.size(), err, nullptr);
      b->text_bytes = bytes;
    }
  } else {
    rc = batch_encode_device(m, dl, b, blob, blob_bytes, off, n, p, err, blob_on_device, after_stream, src_stream);
  }
  if (!rc) rc = encode_tail(b, err);
  if (rc) { *code = rc; batch_free(b); return nullptr; }
  return b;
}

// n NUL-terminated strings -> one buffer + offsets (threaded), then batch_encode_spans
Batch* batch_encode(const HostModel& m, const DeviceLexicon* dl, const char* const* utf8, size_t n, const anx_params& p,
                    std::string& err, int* code, bool keep_text) {
  std::vector<uint32_t> off(n + 1, 0);
  unsigned nthreads = std::max(1u, std::min(16u, usable_hw_threads()));
  if (n < 16384) nthreads = 1;
  std::vector<size_t> part(nthreads + 1, 0);
  auto run_threads = [&](const std::function<void(unsigned, size_t, size_t)>& f) {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthreads; ++t) {
      const size_t lo = n * t / nthreads, hi = n * (t + 1) / nthreads;
      if (nthreads == 1) f(0, lo, hi);
      else th.emplace_back(f, t, lo, hi);
    }
    for (auto& x : th) x.join();
  };
  std::vector<uint32_t> lens(n);
  std::atomic<bool> too_big{false};
  run_threads([&](unsigned t, size_t lo, size_t hi) {
    size_t sum = 0;
    for (size_t i = lo; i < hi; ++i) {
      const size_t l = utf8[i] ? strlen(utf8[i]) : 0;  // a NULL input encodes like an empty one: no results
      if (l >= (1u << 28)) too_big.store(true, std::memory_order_relaxed);
      lens[i] = (uint32_t)l;
      sum += l + 1;
    }
    part[t + 1] = sum;
  });
  for (unsigned t = 0; t < nthreads; ++t) part[t + 1] += part[t];
  if (too_big.load() || part[nthreads] >= ((size_t)1 << 32)) { err = "inputs exceed 4 GB per batch: split the batch"; *code = ANX_ELIMIT; return nullptr; }
  HostBuf<char> blob(part[nthreads] + 1);
  if (!blob.data()) { err = "out of memory"; *code = ANX_ELIMIT; return nullptr; }
  run_threads([&](unsigned t, size_t lo, size_t hi) {
    size_t pos = part[t];
    for (size_t i = lo; i < hi; ++i) {
      off[i] = (uint32_t)pos;
      if (lens[i]) memcpy(&blob[pos], utf8[i], lens[i]);
      blob[pos + lens[i]] = '\0';
      pos += (size_t)lens[i] + 1;
    }
  });
  off[n] = (uint32_t)part[nthreads];
  return batch_encode_spans(m, dl, blob.data(), blob.size(), off.data(), n, p, err, code, keep_text);
}

static int exclusive_scan(const uint32_t* in, uint32_t n, uint32_t* out, uint32_t* tmp, hipStream_t st, uint32_t* maxout = nullptr) {
  const uint32_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nb == 0) return hipMemsetAsync(out, 0, sizeof(uint32_t), st) == hipSuccess ? 0 : -1;
  hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(SCAN_THREADS), 0, st, in, n, out, tmp, maxout);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_THREADS), 0, st, tmp, nb);
  hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(SCAN_THREADS), 0, st, out, n, tmp, nb);
  return 0;
}

// ---- kernel timer (engine_internal.h) ---------------------------------------------------------------------------------
namespace {
struct KTimerRec { std::string name; hipEvent_t e0 = nullptr, e1 = nullptr; int device = 0; bool closed = false; };
struct KTimer {
  std::mutex mu;
  std::atomic<bool> on{false};
  std::vector<KTimerRec> pending;
  uint32_t generation = 0;  // of `pending`: part of the handles ktimer_begin hands out
  std::map<std::string, std::pair<double, uint64_t>> acc;
};
KTimer& ktimer() { static KTimer* t = new KTimer; return *t; }
void ktimer_resolve_locked(KTimer& t) {
  for (KTimerRec& r : t.pending) {
    if (r.closed && hipEventSynchronize(r.e1) == hipSuccess) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { auto& a = t.acc[r.name]; a.first += ms; a.second += 1; }
    }
    if (r.e0) (void)hipEventDestroy(r.e0);
    if (r.e1) (void)hipEventDestroy(r.e1);
  }
  t.pending.clear();
  ++t.generation;  // handles of the records just dropped (also unclosed ones of a launch in progress on another thread) are void now
}
}  // namespace
int ktimer_begin(const char* name, hipStream_t st) {
  KTimer& t = ktimer();
  if (!t.on.load(std::memory_order_relaxed)) return -1;
  KTimerRec r;
  r.name = name;
  if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess || hipEventRecord(r.e0, st) != hipSuccess) {
    if (r.e0) (void)hipEventDestroy(r.e0);
    if (r.e1) (void)hipEventDestroy(r.e1);
    return -1;
  }
  std::lock_guard<std::mutex> lk(t.mu);
  if (t.pending.size() >= 0xFFFFu) return -1;
  t.pending.push_back(r);
  return (int)(((t.generation & 0x7FFFu) << 16) | (uint32_t)(t.pending.size() - 1));  // generation + index: stable across kernel_timer_read
}
void ktimer_end(int handle, hipStream_t st) {
  if (handle < 0) return;
  KTimer& t = ktimer();
  std::lock_guard<std::mutex> lk(t.mu);
  const uint32_t idx = (uint32_t)handle & 0xFFFFu;
  if ((((uint32_t)handle >> 16) & 0x7FFFu) != (t.generation & 0x7FFFu) || idx >= t.pending.size()) return;  // the totals were read in between: the record is gone
  KTimerRec& r = t.pending[idx];
  if (hipEventRecord(r.e1, st) == hipSuccess) r.closed = true;
}
void kernel_timer_enable(bool on) {
  KTimer& t = ktimer();
  std::lock_guard<std::mutex> lk(t.mu);
  ktimer_resolve_locked(t);
  t.acc.clear();
  t.on.store(on);
}
bool kernel_timer_read(const char* name, double* total_ms, uint64_t* launches) {
  KTimer& t = ktimer();
  std::lock_guard<std::mutex> lk(t.mu);
  ktimer_resolve_locked(t);
  auto it = t.acc.find(name ? name : "");
  if (it == t.acc.end()) { if (total_ms) *total_ms = 0.0; if (launches) *launches = 0; return false; }
  if (total_ms) *total_ms = it->second.first;
  if (launches) *launches = it->second.second;
  return true;
}

template <int NP>
static void launch_scan(ScanArgs A, uint32_t nadj, uint32_t nbits, uint32_t nsad, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  // tiles: [bit-plane tiles with an adjacency list | other bit-plane tiles | SAD kind]
  (void)hipEventRecord(e0, st);  // e0 .. e1 = k_scan_adj + k_scan_bits (anx_batch_stats.ms_scan_kernel)
  const bool gen = A.want_exact || A.qpairs || !A.drop_len;
  if (nadj) {
    A.ntiles = nadj;
    if (gen) hipLaunchKernelGGL((k_scan_adj<true>), dim3((nadj + 3) / 4), dim3(256), 0, st, A);
    else hipLaunchKernelGGL((k_scan_adj<false>), dim3((nadj + 3) / 4), dim3(256), 0, st, A);
    A.tiles += nadj;
    nbits -= nadj;
  }
  if (nbits) {
    A.ntiles = nbits;
    // the general instance for StopAtExactMatch, per-query pair counts and runs that keep every pair; production runs take the lean one
    if (A.want_exact || A.qpairs || !A.drop_len) hipLaunchKernelGGL((k_scan_bits<true>), dim3((nbits + 3) / 4), dim3(256), 0, st, A);
    else hipLaunchKernelGGL((k_scan_bits<false>), dim3((nbits + 3) / 4), dim3(256), 0, st, A);
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
  if ((rc = dalloc(&b->raw, cap, err))) return rc;
  b->raw_cap = cap;
  b->region_shift = shift;
  return ANX_OK;
}
// per-slot outputs of the scoring kernels (12 B per slot): only the debug view of every pair needs them
static int ensure_pair_outputs(Batch* b, std::string& err) {
  if (b->p_meta) return ANX_OK;  // ensure_raw drops them whenever the pair list is regrown
  int rc;
  if ((rc = dalloc(&b->p_score, b->raw_cap, err)) || (rc = dalloc(&b->p_meta, b->raw_cap, err))) return rc;
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

// ---- one run of the pipeline = batch_launch (everything enqueued on the stream, NO host round trip in between) + batch_finish
// (wait for the read-back, check the capacities the launch assumed, statistics).  The launch sizes its grids and buffers from the
// previous run of the batch (first run: estimates); every kernel bounds-checks its appends, the fills come back with the one
// read-back at the end, and a run whose assumptions did not hold is repeated with the measured sizes.

// ANX_CAP_DIV=n (tests): the first-run capacity ESTIMATES are divided by n, so that the overflow -> regrow -> repeat path runs
static size_t cap_div() { return (size_t)switches().cap_div; }

// k_rank<true> for models without variant lists at freq_weight == 0, k_rank<false> otherwise (ra = the RankArgs of the launch)
#define ANX_RANK_LAUNCH(...) do { if (!ra.any_variants && ra.freq_weight == 0.0f) hipLaunchKernelGGL(k_rank<true>, __VA_ARGS__); else hipLaunchKernelGGL(k_rank<false>, __VA_ARGS__); } while (0)
// ---- sizes carried from batch to batch (RunHints, kernels_common.hpp) ----------------------------------------------------------------
static bool same_threshold(const anx_threshold& a, const anx_threshold& b) { return a.kind == b.kind && a.value == b.value && a.ratio == b.ratio; }
static bool hints_match(const RunHints& h, const Batch* b) {
  return h.valid && same_threshold(h.kth, b->params.max_anagram_distance) && same_threshold(h.dth, b->params.max_edit_distance) &&
         h.score_threshold == b->params.score_threshold && h.stop == (b->params.stop_at_exact_match ? 1 : 0);
}
// first launch of a batch: fills per region / rows per query of the last finished batch of the same parameters, scaled to this batch's
// queries, + 25 %.  Only ever SHRINKS what the worst-case estimates would take.
static void hints_apply(const DeviceLexicon* dl, Batch* b) {
  if (b->runs_finished || b->hinted || b->prev_maxfill || !switches().hints || switches().cap_div != 1 || b->nq < 4096) return;
  RunHints& h = dl->hints;
  std::lock_guard<std::mutex> g(h.mu);
  if (!hints_match(h, b)) return;
  const double nq = (double)b->nq, m = 1.25;
  const double worst_fill = (nq * 140.0 + (double)b->ntiles * SCAN_CHUNK) / SCAN_REGIONS + 4096.0;
  const double fill = h.maxfill * nq * m + 8192.0;
  if (fill >= worst_fill) return;
  b->prev_maxfill = (uint32_t)fill;
  b->prev_surv_fill = (uint32_t)(h.surv_fill * nq * m + 1024.0);
  b->prev_list_fill = h.list_fill > 0 ? (uint32_t)(h.list_fill * nq * m + 1024.0) : 0u;
  b->hint_rows = (size_t)std::min(nq * 16.0 + 1024.0, h.total_surv * nq * m + 4096.0);
  b->hinted = true;
}
// a first run that stood: what it measured, per query (a running maximum that decays: alternating kinds of batches -- search mode's
// unigram and higher-order batches -- keep the larger one's sizes)
static void hints_record(const DeviceLexicon* dl, const Batch* b, uint32_t maxfill, uint32_t surv_fill, uint32_t list_fill, uint32_t total_surv) {
  if (b->nq < 4096 || !switches().hints) return;
  RunHints& h = dl->hints;
  std::lock_guard<std::mutex> g(h.mu);
  const double nq = (double)b->nq, decay = hints_match(h, b) ? 0.9 : 0.0;
  h.maxfill = std::max(maxfill / nq, h.maxfill * decay);
  h.surv_fill = std::max(surv_fill / nq, h.surv_fill * decay);
  h.list_fill = std::max(list_fill / nq, h.list_fill * decay);
  h.total_surv = std::max(total_surv / nq, h.total_surv * decay);
  h.kth = b->params.max_anagram_distance; h.dth = b->params.max_edit_distance; h.score_threshold = b->params.score_threshold;
  h.stop = b->params.stop_at_exact_match ? 1 : 0;
  h.nq = nq;
  h.valid = true;
}

static int batch_launch(const HostModel& m, const DeviceLexicon* dl, Batch* b, void* stream, std::string& err) {
  if (!dl) { err = "model is not resident on a device"; return ANX_ENODEVICE; }
  if (b->nq >= (1u << 27) || dl->nentries >= (1u << 26)) { err = "more than 2^27 queries per batch or 2^26 lexicon entries (32-bit record offsets, packed pair records)"; return ANX_ELIMIT; }
  HIP_TRY(hipSetDevice(dl->device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const uint32_t nq = (uint32_t)b->nq;
  if (b->launched) {  // a run enqueued with anx_batch_run_async and never waited for: its events and read-back buffer are reused
    HIP_TRY(hipEventSynchronize(b->ev_done));
    b->launched = false;
  }
  b->last_stream = stream;
  b->ran = false;
  b->ran_keep_all = b->keep_all_pairs;  // the run that also stores the per-slot outputs the debug fetch of every pair reads
  b->n_pairs = b->n_results = b->n_surv = 0;
  b->n_raw = 0;
  if (nq == 0) { b->ran = true; return ANX_OK; }
  const int stop = b->params.stop_at_exact_match ? 1 : 0;
  int rc;
  hints_apply(dl, b);
  if (b->raw_cap == 0 && b->hinted && (rc = ensure_raw(b, (size_t)b->prev_maxfill + (b->prev_maxfill >> 3) + 4096, err))) return rc;
  if (b->raw_cap == 0 && (rc = ensure_raw(b, (nq * (size_t)140 + (size_t)b->ntiles * SCAN_CHUNK) / SCAN_REGIONS / cap_div() + 4096, err)))
    return rc;
  const uint32_t region_cap = 1u << b->region_shift;
  // slots per region the scoring grid covers: the previous fill + 1/8 (first run: the whole region; blocks beyond a region's
  // fill return at once)
  const uint32_t fill_cap = b->prev_maxfill ? (uint32_t)std::min<size_t>(region_cap, (size_t)b->prev_maxfill + (b->prev_maxfill >> 3) + FS_BLK) : region_cap;
  b->fill_cap_launched = fill_cap;
  HIP_TRY(hipEventRecord(b->ev[0], st));
  // ---- scan ------------------------------------------------------------------------------------------
  HIP_TRY(hipMemsetAsync(b->counters, 0, CTR_N * sizeof(uint32_t), st));
  HIP_TRY(hipMemsetAsync(b->rctr, 0, SCAN_REGIONS * RC_STRIDE * sizeof(uint32_t), st));
  if (b->ntiles == 0) { HIP_TRY(hipEventRecord(b->ev_scan0, st)); HIP_TRY(hipEventRecord(b->ev[5], st)); }
  if (b->ntiles) {
    ScanArgs A;
    A.tiles = b->d_tiles; A.ntiles = b->ntiles; A.q_bits = b->q_bits; A.q_cv = b->q_cv;
    A.cls_bits = dl->cls_bits; A.cls_planes = dl->cls_planes; A.scan_rec = dl->scan_rec; A.scan_rec34 = dl->scan_rec34; A.pad_rec = dl->nentries; A.cstride = dl->cstride; A.pad_class = dl->nclasses;
    A.cls_len = dl->cls_len; A.cls_off = dl->cls_off; A.sig = dl->sig; A.sig_e = dl->sig_e; A.sig_cbeg = dl->sig_cbeg; A.sighash = dl->sighash; A.sighash_e = dl->sighash_e; A.hash_mask = dl->hash_mask; A.ball = dl->ball;
    A.adj_hdr = dl->adj_hdr; A.adj_planes = dl->adj_planes; A.adj_ids = dl->adj_ids;
    A.chunk = SCAN_CHUNK;
    A.chunk_fused = switches().scan_chunk_fused ? (uint32_t)switches().scan_chunk_fused : SCAN_CHUNK_FUSED;
#ifdef ANX_DEBUG_SWITCHES
    { const char* e = getenv("ANX_SCAN_CHUNK"); const int v = e ? atoi(e) : 0; if (v >= 32 && v <= 1024) A.chunk = (uint32_t)v; }
#endif
    A.raw = b->raw; A.region_cap = region_cap; A.rctr = b->rctr; A.qexact = b->qexact; A.want_exact = stop;
    A.drop_len = (!stop && !b->keep_all_pairs) ? 1 : 0;
    A.q_rec = b->q_rec; A.e_rec = dl->e_rec;
    // the band-match bound where the pair is born (not in the run that materialises every pair for the debug view, and not with
    // StopAtExactMatch, whose dropped pairs are counted by the scoring kernel from the materialised list)
    A.fuse = (switches().fuse_prefilter && switches().prefilter && A.drop_len) ? 1 : 0;
    A.qpairs = nullptr;
    if (b->count_pairs) {
      if (!b->qpairs && (rc = dalloc(&b->qpairs, nq, err))) return rc;
      HIP_TRY(hipMemsetAsync(b->qpairs, 0, nq * sizeof(uint32_t), st));
      A.qpairs = b->qpairs;
    }
    A.dbg = 0;
#ifdef ANX_DEBUG_SWITCHES
    { const char* e = getenv("ANX_SCAN_DBG"); A.dbg = e ? atoi(e) : 0; }  // read per run: tools/scan_probe.py switches it between runs
#endif
    const uint32_t nsad = b->n_sad_tiles, nbits = A.ntiles - nsad, nadj = std::min(b->n_adj_tiles, nbits);
    switch (dl->nplanes) {
      case 8: launch_scan<8>(A, nadj, nbits, nsad, st, b->ev_scan0, b->ev[5]); break;
      case 16: launch_scan<16>(A, nadj, nbits, nsad, st, b->ev_scan0, b->ev[5]); break;
      case 24: launch_scan<24>(A, nadj, nbits, nsad, st, b->ev_scan0, b->ev[5]); break;
      case 32: launch_scan<32>(A, nadj, nbits, nsad, st, b->ev_scan0, b->ev[5]); break;
      default: launch_scan<42>(A, nadj, nbits, nsad, st, b->ev_scan0, b->ev[5]); break;
    }
  }
  HIP_TRY(hipMemcpyAsync(b->h_read + HR_RCTR, b->rctr, SCAN_REGIONS * RC_STRIDE * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipEventRecord(b->ev[1], st));
  b->n_raw = (uint32_t)(SCAN_REGIONS << b->region_shift);  // slot space (regions are sparse)
  // ---- score -----------------------------------------------------------------------------------------
  HIP_TRY(hipMemsetAsync(b->qsurv, 0, nq * sizeof(uint32_t), st));
  HIP_TRY(hipMemsetAsync(b->qmaxfreq, 0, nq * sizeof(uint32_t), st));
  if (dl->any_variants) HIP_TRY(hipMemsetAsync(b->qexpand, 0, nq * sizeof(uint32_t), st));
  ScoreArgs sa;
  sa.dbg = 0;
#ifdef ANX_DEBUG_SWITCHES
  { const char* e = getenv("ANX_SCORE_DBG"); sa.dbg = e ? atoi(e) : 0; }
#endif
  sa.quot = dl->quot;
  sa.store_pairs = b->keep_all_pairs ? 1 : 0;
  if (sa.store_pairs && (rc = ensure_pair_outputs(b, err))) return rc;
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
  {
    const int enable_filter = switches().prefilter, enable_fast = switches().score_fast;
    const int fastD = (enable_fast && d >= 1 && d <= 3) ? (int)d : 0;
    // survivor records: region r of the survivor list takes the survivors of region r of the pair list.  Sized from the
    // previous run (first run: half the slots the grid covers -- ~10 % of the slots survive on config 2); an overflow is
    // detected by batch_finish and the run repeated with the measured size
    {
      const size_t need = b->prev_surv_fill ? (size_t)b->prev_surv_fill + (b->prev_surv_fill >> 3) + 256 : (size_t)fill_cap / 2 / cap_div() + 256;
      if (need > b->surv_region_cap) {
        if (b->surv) pool_free(b->surv);
        b->surv = nullptr;
        b->surv_region_cap = 0;
        if ((rc = dalloc(&b->surv, need * SCAN_REGIONS, err))) return rc;
        b->surv_region_cap = need;
      }
    }
    so.list = b->surv;
    so.region_cap = (uint32_t)b->surv_region_cap;
    // slot lists for the pairs the fused kernel cannot score inline: strings of 17..32 symbols (8-word kernel) and everything
    // else (longer strings, d > 3)
    const bool need_lists = !fastD || have_long_q || dl->max_len > 16;
    if (need_lists) {
      // without the inline DL (d > 3) every length-compatible pair goes to the general list; else the prefilter passes ~1/3
      const size_t need = b->prev_list_fill ? (size_t)b->prev_list_fill + (b->prev_list_fill >> 3) + 256 : (fastD ? (size_t)fill_cap / 2 : (size_t)fill_cap) / cap_div() + 256;
      if (need > b->list_cap) {
        for (void* p : {(void*)b->list8, (void*)b->listg, (void*)b->listw})
          if (p) pool_free(p);
        b->list8 = b->listg = b->listw = nullptr;
        b->list_cap = 0;
        if ((rc = dalloc(&b->list8, need * SCAN_REGIONS, err)) || (rc = dalloc(&b->listg, need * SCAN_REGIONS, err)) ||
            (rc = dalloc(&b->listw, need * SCAN_REGIONS, err))) return rc;
        b->list_cap = need;
      }
    }
    HIP_TRY(hipMemsetAsync(b->sctr, 0, SCAN_REGIONS * RC_STRIDE * sizeof(uint32_t), st));
    HIP_TRY(hipMemsetAsync(b->lctr, 0, 3 * SCAN_REGIONS * RC_STRIDE * sizeof(uint32_t), st));
    const SlotList l8{b->list8, b->lctr, (uint32_t)b->list_cap}, lg{b->listg, b->lctr + SCAN_REGIONS * RC_STRIDE, (uint32_t)b->list_cap},
                   lw{b->listw, b->lctr + 2 * SCAN_REGIONS * RC_STRIDE, (uint32_t)b->list_cap};
    const PairArgs pa{b->raw, b->q_meta, b->q_rows, b->q_rec, dl->e_rec, dl->ent_meta, dl->ent_rowoff, dl->rows, dl->ent_freq, dl->ent_var_off,
                      b->p_score, b->p_meta, b->qmaxfreq, b->qsurv, b->qexpand, dl->e_planes};
    FilterArgs fa;
    fa.region_shift = b->region_shift; fa.rctr = b->rctr; fa.qexact = b->qexact; fa.stop = stop; fa.enable = enable_filter;
    // pairs with a string of 17..32 symbols go to the 8-word register DL also when no QUERY is that long (the few 17..19-symbol candidates
    // of a short-query batch): until round 6 those went to the general LDS kernel, whose one round cost 0.05 ms (scoring stage of BASELINE
    // configs[1] 0.555 -> 0.53 ms; the three slot-list kernels as ONE launch, k_small_lists, on top of that: 0.527, not kept)
    fa.use_nw8 = (have_long_q || fastD > 0) ? 1 : 0; fa.counters = b->counters; fa.stat_ctr = b->sctr; fa.fill_cap = fill_cap; fa.blk = FS_BLK;
    const dim3 fgrid(((fill_cap + FS_BLK - 1) / FS_BLK) * SCAN_REGIONS);
    {
      FsCold* cold = reinterpret_cast<FsCold*>(reinterpret_cast<char*>(b->h_read) + HR_COLD_OFF);  // pinned: a truly asynchronous copy
      *cold = FsCold{sa, so, l8, lg, lw};
      if (!b->d_cold && (rc = dalloc(reinterpret_cast<char**>(&b->d_cold), sizeof(FsCold), err))) return rc;
      HIP_TRY(hipMemcpyAsync(b->d_cold, cold, sizeof(FsCold), hipMemcpyHostToDevice, st));
    }
    HIP_TRY(hipEventRecord(b->ev_fs0, st));  // ev_fs0 .. ev_fs1 = k_filter_score alone (anx_batch_stats.ms_filter_score_kernel)
    // the 8-word prefilter of the wide pairs (a string of 17..32 symbols) runs in k_filter_wide: its state inline costs the fused
    // kernel 105 instead of 70 VGPRs.  Round 6: for batches with long queries as well (BASELINE configs[2]: 9.68 -> 9.52 ms per pass,
    // configs[3]'s share: 6.73 -> 6.42 ms per 1 M queries)
    const bool split_wide = switches().fs_split != 0;
    // the one-add zero test of the prefilter needs every symbol code (classes, unknown = A + 1) below the masked paddings 0x7E / 0x7F
    const int enable_b7 = switches().fs_b7;
    const bool b7 = enable_b7 && m.alphabet.size() + 1 < 0x7E;
    // symbol planes instead of byte rows for the inline DL and its tail (codes + 1 in six bits)
    const bool planes = b7 && switches().fs_planes && (int)m.alphabet.size() <= kSymbolPlanesMaxA;
#define ANX_FS_LAUNCH(DD, WW, BB) hipLaunchKernelGGL((k_filter_score<DD, WW, BB>), fgrid, dim3(256), 0, st, fa, pa, static_cast<const FsCold*>(b->d_cold))
#define ANX_FS_PICK(WW, BB)                      \
  do {                                           \
    if (fastD == 1) ANX_FS_LAUNCH(1, WW, BB);    \
    else if (fastD == 2) ANX_FS_LAUNCH(2, WW, BB); \
    else if (fastD == 3) ANX_FS_LAUNCH(3, WW, BB); \
    else ANX_FS_LAUNCH(0, WW, BB);               \
  } while (0)
    if (split_wide) { if (planes) ANX_FS_PICK(false, 2); else if (b7) ANX_FS_PICK(false, 1); else ANX_FS_PICK(false, 0); }
    else { if (b7) ANX_FS_PICK(true, 1); else ANX_FS_PICK(true, 0); }   // (ANX_FS_SPLIT=0, A/B: byte rows)
#undef ANX_FS_PICK
#undef ANX_FS_LAUNCH
    HIP_TRY(hipEventRecord(b->ev_fs1, st));
    if (need_lists) {  // the list fills are only known on the device: fixed grids walk the lists in strides
      const dim3 lgrid(LIST_P * SCAN_REGIONS);
      if (split_wide && enable_filter) hipLaunchKernelGGL(k_filter_wide, lgrid, dim3(256), 0, st, lw, fa, pa, sa, fastD, l8, lg);
      if (fastD) {
        if (fastD == 1) hipLaunchKernelGGL(k_score_fast8<1>, lgrid, dim3(256), 0, st, l8, pa, sa, so);
        else if (fastD == 2) hipLaunchKernelGGL(k_score_fast8<2>, lgrid, dim3(256), 0, st, l8, pa, sa, so);
        else hipLaunchKernelGGL(k_score_fast8<3>, lgrid, dim3(256), 0, st, l8, pa, sa, so);
      }
      hipLaunchKernelGGL(k_score_pairs, dim3(LIST_P * SCAN_REGIONS), dim3(threads), threads * sa.stride, st,
                         lg, pa, sa, so);
    }
  }
  HIP_TRY(hipEventRecord(b->ev[2], st));
  // ---- compact survivors + rank -----------------------------------------------------------------------
  RankArgs ra;
  // confusables weighted on the device, late mode: k_rank crops without the cutoff, conf.hip re-ranks and cuts off afterwards
  ra.cutoff_threshold = b->conf_mode == 1 ? 0.0 : b->params.cutoff_threshold;
  ra.max_matches = b->params.max_matches;
  ra.freq_weight = b->params.freq_weight;
  ra.have_freq = m.have_freq ? 1 : 0;
  ra.any_variants = dl->any_variants;
  exclusive_scan(b->qsurv, nq, b->soff, b->scan_tmp, st);
  HIP_TRY(hipMemcpyAsync(b->qcur, b->soff, nq * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));  // cursors of the compaction
  if (dl->any_variants) {
    // variant lists: a survivor expands to several rows; the row buffer and the grid of k_compact come from the survivor
    // counts, so this (rare) configuration keeps one host round trip in the middle of the run
    uint32_t total_surv = 0;
    HIP_TRY(hipMemcpyAsync(&total_surv, b->soff + nq, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(b->h_read + HR_SCTR, b->sctr, SCAN_REGIONS * RC_STRIDE * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    uint32_t surv_fill = 0;
    for (uint32_t r = 0; r < SCAN_REGIONS; ++r) surv_fill = std::max(surv_fill, b->h_read[HR_SCTR + r * RC_STRIDE]);
    if ((size_t)surv_fill > b->surv_region_cap) {  // survivor records were dropped: nothing of this run is used (batch_finish repeats it)
      surv_fill = 0;
      HIP_TRY(hipMemsetAsync(b->counters + CTR_OVERFLOW, 0xFF, sizeof(uint32_t), st));
    }
    if ((rc = ensure_surv(b, (size_t)total_surv + (total_surv >> 2) + 1024, err))) return rc;
    if (surv_fill) {
      CompactArgs ca{m.have_freq ? 1 : 0, dl->any_variants};
      hipLaunchKernelGGL(k_compact, dim3(((surv_fill + 255) / 256) * SCAN_REGIONS), dim3(256), 0, st, b->surv, b->sctr,
                         (uint32_t)b->surv_region_cap, ca, b->qcur, dl->ent_rec, dl->ent_var_off, dl->var_target,
                         dl->var_target_freq, dl->var_score, b->c_rows);
    }
    HIP_TRY(hipEventRecord(b->ev[3], st));
    if (b->conf_mode == 2 && (rc = conf_launch(m, dl, b, st, true, (uint32_t)std::min<size_t>(b->surv_cap, 0xFFFFFFFFu), err))) return rc;
    ANX_RANK_LAUNCH( dim3((nq + 4 * RANK_QPW - 1) / (4 * RANK_QPW)), dim3(256), 0, st, nq, b->soff, b->c_rows, b->qmaxfreq,
                       b->qexpand, ra, b->t_key, b->r_rows, b->r_count, 0xFFFFFFFFu, b->counters + CTR_OVERFLOW);
    if (b->conf_mode == 1 && (rc = conf_launch(m, dl, b, st, false, (uint32_t)std::min<size_t>(b->surv_cap, 0xFFFFFFFFu), err))) return rc;
  } else {
    // No host round trip between scoring and ranking: the row buffers keep the size of the previous run (first run:
    // an estimate), the kernels check the total on the device, and batch_finish repeats the run if it did not fit.
    if (b->surv_cap == 0 && (rc = ensure_surv(b, b->hint_rows ? b->hint_rows : (size_t)nq * 16 / cap_div() + 1024, err))) return rc;
    const uint32_t row_cap = (uint32_t)std::min<size_t>(b->surv_cap, 0xFFFFFFFFu);
    hipLaunchKernelGGL(k_compact_grouped, dim3(COMPACT_P * SCAN_REGIONS), dim3(COMPACT_B), 0, st, b->surv, b->sctr,
                       (uint32_t)b->surv_region_cap, m.have_freq ? 1 : 0, b->qcur, dl->ent_rec, b->c_rows, b->soff + nq, row_cap,
                       b->counters + CTR_OVERFLOW);
    HIP_TRY(hipEventRecord(b->ev[3], st));
    if (b->conf_mode == 2 && (rc = conf_launch(m, dl, b, st, true, row_cap, err))) return rc;
    ANX_RANK_LAUNCH( dim3((nq + 4 * RANK_QPW - 1) / (4 * RANK_QPW)), dim3(256), 0, st, nq, b->soff, b->c_rows, b->qmaxfreq,
                       b->qexpand, ra, b->t_key, b->r_rows, b->r_count, row_cap, b->counters + CTR_OVERFLOW);
    if (b->conf_mode == 1 && (rc = conf_launch(m, dl, b, st, false, row_cap, err))) return rc;
  }
  exclusive_scan(b->r_count, nq, b->r_off, b->scan_tmp, st, b->counters + CTR_MAXROWS);
  HIP_TRY(hipEventRecord(b->ev[4], st));
  // ---- the read-back of the run --------------------------------------------------------------------------
  HIP_TRY(hipMemcpyAsync(b->h_read + HR_SCTR, b->sctr, SCAN_REGIONS * RC_STRIDE * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(b->h_read + HR_LCTR, b->lctr, 3 * SCAN_REGIONS * RC_STRIDE * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(b->h_read + HR_CTR, b->counters, CTR_N * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(b->h_read + HR_TOTAL_SURV, b->soff + nq, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(b->h_read + HR_TOTAL_RESULTS, b->r_off + nq, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  b->h_read[HR_CONF] = b->h_read[HR_CONF + 1] = 0;
  if (b->conf_mode && b->cf_ctr) HIP_TRY(hipMemcpyAsync(b->h_read + HR_CONF, b->cf_ctr, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipEventRecord(b->ev_done, st));
  HIP_TRY(hipGetLastError());
  b->launched = true;
  return ANX_OK;
}

// Waits for the launched run.  Returns ANX_OK with *retry = false when the run stands; *retry = true when a capacity the launch
// assumed was exceeded (the buffers have been regrown: launch again).
static int batch_finish(const HostModel& m, const DeviceLexicon* dl, Batch* b, bool* retry, std::string& err) {
  (void)m;
  *retry = false;
  if (!b->launched) return ANX_OK;  // empty batch
  HIP_TRY(hipSetDevice(dl->device));
  HIP_TRY(hipEventSynchronize(b->ev_done));
  b->launched = false;
  const uint32_t nq = (uint32_t)b->nq;
  const uint32_t* h = b->h_read;
  uint32_t maxfill = 0, surv_fill = 0, list_fill = 0;
  uint64_t n_valid = 0, n_slots = 0, nsel = 0, n_fused = 0, n_adj_rows = 0, n_adj_rows_first = 0;
  b->n_class_tests = 0;
  for (int i = 0; i <= NBITPLANES; ++i) b->n_tests_kind[i] = 0;
  for (uint32_t r = 0; r < SCAN_REGIONS; ++r) {
    const uint32_t* c = h + HR_RCTR + r * RC_STRIDE;
    maxfill = std::max(maxfill, c[RC_RAW]);
    b->region_fill[r] = c[RC_RAW];
    n_slots += c[RC_RAW];
    n_valid += c[RC_VALID];
    n_fused += c[RC_FUSED];
    n_adj_rows += c[RC_ADJ];
    n_adj_rows_first += c[RC_ADJ_FIRST];
    for (int i = 0; i <= NBITPLANES; ++i) {
      uint64_t v;
      memcpy(&v, c + RC_TESTS + 2 * i, sizeof v);
      b->n_tests_kind[i] += v;
      b->n_class_tests += v;
    }
    surv_fill = std::max(surv_fill, h[HR_SCTR + r * RC_STRIDE]);
    nsel += h[HR_SCTR + r * RC_STRIDE + 1];
    for (int l = 0; l < 3; ++l) list_fill = std::max(list_fill, h[HR_LCTR + (l * SCAN_REGIONS + r) * RC_STRIDE]);
  }
  const uint32_t total_surv = h[HR_TOTAL_SURV], total_results = h[HR_TOTAL_RESULTS];
  b->conf_fallback = b->conf_mode != 0 && (h[HR_CONF + 1] != 0 || b->conf_skipped);
  b->stats.n_conf_scripts = b->conf_mode ? h[HR_CONF] : 0;
  b->stats.n_adj_tiles = std::min(b->n_adj_tiles, b->ntiles - b->n_sad_tiles);
  // ---- did the run fit what the launch assumed? ------------------------------------------------------------
  int rc;
  bool again = false;
  if (maxfill > (1u << b->region_shift)) {  // pair list
    if ((rc = ensure_raw(b, (size_t)maxfill + (maxfill >> 3) + 4096, err))) return rc;
    again = true;
  }
  if (maxfill > b->fill_cap_launched) again = true;  // slots the scoring grid did not cover
  if ((size_t)surv_fill > b->surv_region_cap) again = true;
  if (list_fill && (size_t)list_fill > b->list_cap) again = true;
  if (!dl->any_variants && (size_t)total_surv > b->surv_cap) {
    if ((rc = ensure_surv(b, (size_t)total_surv + (total_surv >> 2) + 1024, err))) return rc;
    again = true;
  }
  b->prev_maxfill = std::max(b->prev_maxfill, maxfill);
  b->prev_surv_fill = std::max(b->prev_surv_fill, surv_fill);
  b->prev_list_fill = std::max(b->prev_list_fill, list_fill);
  if (again) { *retry = true; b->runs_finished++; return ANX_OK; }
  if (b->runs_finished++ == 0) hints_record(dl, b, maxfill, surv_fill, list_fill, total_surv);
  if (maxfill == 0) b->n_raw = 0;
  b->n_pairs = n_valid - h[HR_CTR + CTR_SKIPPED];
  b->n_surv = total_surv;
  b->n_sel = nsel;
  b->n_results = total_results;
  b->max_rows = h[HR_CTR + CTR_MAXROWS];
  b->ran = true;
  anx_batch_stats& s = b->stats;
  s.n_queries = nq;
  s.n_pairs = b->n_pairs;
  s.n_class_tests = b->n_class_tests;
  s.n_results = total_results;
  s.n_scan_blocks = b->ntiles;
  for (int i = 0; i <= NBITPLANES; ++i) s.n_tests_kind[i] = b->n_tests_kind[i];
  s.n_pair_slots = n_slots;
  s.n_survivors = total_surv;
  s.n_selected = b->n_sel;
  s.n_prefiltered_in_scan = n_fused;
  s.n_adj_records = n_adj_rows * kAdjRow;
  s.n_adj_records_first = n_adj_rows_first * kAdjRow;
  (void)hipEventElapsedTime(&s.ms_scan, b->ev[0], b->ev[1]);
  (void)hipEventElapsedTime(&s.ms_score, b->ev[1], b->ev[2]);
  (void)hipEventElapsedTime(&s.ms_group, b->ev[2], b->ev[3]);
  (void)hipEventElapsedTime(&s.ms_rank, b->ev[3], b->ev[4]);
  (void)hipEventElapsedTime(&s.ms_total, b->ev[0], b->ev[4]);
  (void)hipEventElapsedTime(&s.ms_scan_kernel, b->ev_scan0, b->ev[5]);
  (void)hipEventElapsedTime(&s.ms_filter_score_kernel, b->ev_fs0, b->ev_fs1);
  return ANX_OK;
}

// asynchronous form: enqueue the run on `stream` and return; batch_wait completes it (repeating it synchronously in the rare
// case a capacity estimate did not hold).  Several batches in flight on different streams overlap the latency-bound tail of
// one run (compaction, ranking) with the scan of the next.
// own_streams (single-replica models, ANX_RUN_OVERLAP != 0): the run goes to one of two streams of the library's own, alternately,
// ordered behind what the caller's stream holds at this point (an event) -- so that consecutive asynchronous runs of DIFFERENT
// batches overlap (the scan of one under the scoring tail, compaction and ranking of the other) although the caller uses one
// stream: 3.11 -> 2.7-2.8 ms per step on BASELINE configs[1], what two caller streams gave before.  Everything that reads a batch's
// results requires a finished run (batch_wait has synchronised with it), so the caller's stream needs no event back.
int batch_run_async(const HostModel& m, const DeviceLexicon* dl, Batch* b, void* stream, bool own_streams, std::string& err) {
  void* use = stream;
  if (own_streams && dl && switches().run_overlap) {
    HIP_TRY(hipSetDevice(dl->device));
    DevPool& pl = pool_of(dl->device);
    hipStream_t s = nullptr;
    {
      std::lock_guard<std::mutex> g(pl.mu);
      hipStream_t& slot = pl.run_streams[pl.run_counter++ & 1u];
      if (!slot && hipStreamCreateWithFlags(&slot, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); slot = nullptr; }
      s = slot;
    }
    if (s) {
      HIP_TRY(hipEventRecord(b->ev_in, reinterpret_cast<hipStream_t>(stream)));
      HIP_TRY(hipStreamWaitEvent(s, b->ev_in, 0));
      use = s;
    }
  }
  const int rc = batch_launch(m, dl, b, use, err);
  // what follows a finished run (fetches, exports, a repeated launch) goes to the CALLER's stream: on the library's shared run stream a
  // fetch would queue behind the whole run of another batch in flight there
  if (rc == ANX_OK && use != stream) b->last_stream = stream;
  return rc;
}
int batch_wait(const HostModel& m, const DeviceLexicon* dl, Batch* b, std::string& err) {
  for (int attempt = 0;; ++attempt) {
    bool retry = false;
    int rc = batch_finish(m, dl, b, &retry, err);
    if (rc || !retry) return rc;
    if (attempt == 3) { err = "the run did not fit its buffers after three regrows"; return ANX_ENODEVICE; }
    if ((rc = batch_launch(m, dl, b, b->last_stream, err))) return rc;
  }
}
int batch_run(const HostModel& m, const DeviceLexicon* dl, Batch* b, void* stream, std::string& err) {
  const int rc = batch_launch(m, dl, b, stream, err);
  return rc ? rc : batch_wait(m, dl, b, err);
}

void batch_set_last_stream(Batch* b, void* stream) { if (b && !b->launched) b->last_stream = stream; }
size_t batch_n_results(const Batch* b) { return b->ran ? (size_t)b->n_results : 0; }
size_t batch_n_input(const Batch* b) { return b->n_input; }

// Downloads the ranked rows of the batch into caller-provided storage: out[0 .. n_results) and off[0 .. n_input] = base + the
// CSR offsets (a shard of a multi-device batch writes its slice of the whole call's arrays).
int batch_fetch_into(const Batch* b, anx_result* out, size_t* off, size_t base, std::string& err) {
  if (!b->ran) { err = "batch has not been run"; return ANX_EINVAL; }
  HIP_TRY(hipSetDevice(b->device));
  const size_t n = b->n_input;
  int rc = ANX_OK;
  if (b->nq && b->n_results) {
    // the device lays the rows out in the caller's input order (counts scattered to the original indices, exclusive
    // scan, row copy), so the host only copies: offsets (u32) and the finished anx_result array.  Runs on the stream of
    // the batch's last run; every exit frees the temporaries.
    uint32_t *d_cnt = nullptr, *d_off = nullptr, *d_tmp = nullptr;
    anx_result* d_out = nullptr;
    hipStream_t st = reinterpret_cast<hipStream_t>(b->last_stream);
    auto body = [&]() -> int {
      const uint32_t n32 = (uint32_t)n, nq32 = (uint32_t)b->nq;
      const size_t nblk = (n + SCAN_TILE - 1) / SCAN_TILE + 2;
      HIP_TRY(pool_malloc(reinterpret_cast<void**>(&d_cnt), n * sizeof(uint32_t)));
      HIP_TRY(pool_malloc(reinterpret_cast<void**>(&d_off), (n + 1) * sizeof(uint32_t)));
      HIP_TRY(pool_malloc(reinterpret_cast<void**>(&d_tmp), nblk * sizeof(uint32_t)));
      HIP_TRY(pool_malloc(reinterpret_cast<void**>(&d_out), b->n_results * sizeof(anx_result)));
      HIP_TRY(hipMemsetAsync(d_cnt, 0, n * sizeof(uint32_t), st));
      hipLaunchKernelGGL(k_fetch_counts, dim3((nq32 + 255) / 256), dim3(256), 0, st, nq32, b->r_count, b->q_orig, d_cnt);
      exclusive_scan(d_cnt, n32, d_off, d_tmp, st);
      hipLaunchKernelGGL(k_fetch_rows, dim3((nq32 + 255) / 256), dim3(256), 0, st, nq32, b->soff, b->r_count, b->r_rows, b->q_orig, d_off, d_out);
      std::vector<uint32_t> h_off(n + 1);
      HIP_TRY(hipMemcpyAsync(h_off.data(), d_off, (n + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(out, d_out, b->n_results * sizeof(anx_result), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      for (size_t i = 0; i <= n; ++i) off[i] = base + h_off[i];
      return ANX_OK;
    };
    rc = body();
    if (rc) (void)hipStreamSynchronize(st);  // nothing of this call may still be in flight when its blocks return to the pool
    for (void* p : {(void*)d_cnt, (void*)d_off, (void*)d_tmp, (void*)d_out}) pool_free(p);
  } else {
    for (size_t i = 0; i <= n; ++i) off[i] = base;
  }
  return rc;
}

// The same rows as 16-byte anx_topk_record {vocab u32, freq f32, dist f64} with u32 offsets: half the bytes over PCIe (config 2:
// 70 MB instead of 141 MB per million queries).  out[0 .. n_results), off[0 .. n_input] = base + CSR offsets.
int batch_fetch_compact_into(const Batch* b, anx_topk_record* out, uint32_t* off, uint32_t base, std::string& err) {
  if (!b->ran) { err = "batch has not been run"; return ANX_EINVAL; }
  HIP_TRY(hipSetDevice(b->device));
  const size_t n = b->n_input;
  int rc = ANX_OK;
  if (b->nq && b->n_results) {
    uint32_t *d_cnt = nullptr, *d_off = nullptr, *d_tmp = nullptr;
    anx_topk_record* d_out = nullptr;
    hipStream_t st = reinterpret_cast<hipStream_t>(b->last_stream);
    auto body = [&]() -> int {
      const uint32_t n32 = (uint32_t)n, nq32 = (uint32_t)b->nq;
      const size_t nblk = (n + SCAN_TILE - 1) / SCAN_TILE + 2;
      HIP_TRY(pool_malloc(reinterpret_cast<void**>(&d_cnt), n * sizeof(uint32_t)));
      HIP_TRY(pool_malloc(reinterpret_cast<void**>(&d_off), (n + 1) * sizeof(uint32_t)));
      HIP_TRY(pool_malloc(reinterpret_cast<void**>(&d_tmp), nblk * sizeof(uint32_t)));
      HIP_TRY(pool_malloc(reinterpret_cast<void**>(&d_out), b->n_results * sizeof(anx_topk_record)));
      HIP_TRY(hipMemsetAsync(d_cnt, 0, n * sizeof(uint32_t), st));
      hipLaunchKernelGGL(k_fetch_counts, dim3((nq32 + 255) / 256), dim3(256), 0, st, nq32, b->r_count, b->q_orig, d_cnt);
      exclusive_scan(d_cnt, n32, d_off, d_tmp, st);
      hipLaunchKernelGGL(k_export_rows, dim3((nq32 + 255) / 256), dim3(256), 0, st, nq32, b->soff, b->r_count, b->r_rows, b->q_orig, d_off, d_out);
      HIP_TRY(hipMemcpyAsync(off, d_off, (n + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(out, d_out, b->n_results * sizeof(anx_topk_record), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      if (base)
        for (size_t i = 0; i <= n; ++i) off[i] += base;
      return ANX_OK;
    };
    rc = body();
    if (rc) (void)hipStreamSynchronize(st);
    for (void* p : {(void*)d_cnt, (void*)d_off, (void*)d_tmp, (void*)d_out}) pool_free(p);
  } else {
    for (size_t i = 0; i <= n; ++i) off[i] = base;
  }
  return rc;
}

int batch_fetch(const HostModel& m, const DeviceLexicon* dl, const Batch* b, anx_result** rows, size_t** offs,
                std::string& err) {
  (void)m;
  (void)dl;
  if (!b->ran) { err = "batch has not been run"; return ANX_EINVAL; }
  HIP_TRY(hipSetDevice(b->device));
  size_t* off = static_cast<size_t*>(malloc((b->n_input + 1) * sizeof(size_t)));
  anx_result* out = static_cast<anx_result*>(host_result_alloc(std::max<size_t>(1, b->n_results) * sizeof(anx_result)));
  if (!off || !out) { free(off); host_result_free(out); err = "out of memory"; return ANX_EINVAL; }
  const int rc = batch_fetch_into(b, out, off, 0, err);
  if (rc) { free(off); host_result_free(out); return rc; }
  *rows = out;
  *offs = off;
  return ANX_OK;
}

// sorted position -> input index on the host (the device-side encoder only leaves it in q_orig): downloaded on first use
static int ensure_order(const Batch* cb, std::string& err) {
  Batch* b = const_cast<Batch*>(cb);
  if (b->order.size() == b->nq) return ANX_OK;
  b->order.resize(b->nq);
  if (b->nq) HIP_TRY(hipMemcpy(b->order.data(), b->q_orig, b->nq * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return ANX_OK;
}

int batch_fetch_pairs(const HostModel& m, const DeviceLexicon* dl, const Batch* cb, anx_pair** out, size_t* n,
                      std::string& err) {
  if (!cb->ran) { err = "batch has not been run"; return ANX_EINVAL; }
  HIP_TRY(hipSetDevice(cb->device));
  if (!cb->ran_keep_all) {
    // the production run only counts the pairs that fail the DL's length test; this debug view lists every pair, so the
    // batch is run once more with all of them materialised (same results, same statistics)
    Batch* mb = const_cast<Batch*>(cb);
    mb->keep_all_pairs = true;
    const int rc = batch_run(m, dl, mb, nullptr, err);
    if (rc) return rc;
  }
  const Batch* b = cb;
  { const int rc = ensure_order(b, err); if (rc) return rc; }
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
      r.vocab_id = ev[pr[i].y & RAW_ENTRY_MASK];
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
  if (stride < b->max_rows) {  // rows beyond the stride would be dropped silently (freq_weight != 0, variant lists or max_matches = 0
                               // can return more than max_matches + 1 rows): refuse
    err = "export stride " + std::to_string(stride) + " is smaller than the longest result list (" + std::to_string(b->max_rows) + " rows): use a larger stride or anx_batch_export_compact";
    return ANX_ELIMIT;
  }
  HIP_TRY(hipSetDevice(b->device));
  b->async_stream = stream; b->async_pending = true;
  const uint64_t total = (uint64_t)b->nq * stride;
  if (total)
    hipLaunchKernelGGL(k_export_topk, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), (uint32_t)b->nq, stride, b->soff, b->r_count,
                       b->r_rows, b->q_orig, static_cast<anx_topk_record*>(dst));
  HIP_TRY(hipGetLastError());
  return ANX_OK;
}

// Compact form of the same records: uint32 offsets[n+1] (input order; padded to a 16-byte multiple), then the
// n_results records back to back.  The size is known on the host (n_results is read back by the run), so the
// multi-GPU gather moves the used bytes only (config 2: 4.4 results per query, 74 MB per million instead of 176 MB).
int batch_export_compact(const DeviceLexicon* dl, const Batch* b, void* dst, size_t capacity, void* stream,
                         size_t* used, std::string& err) {
  (void)dl;
  if (!b->ran) { err = "batch has not been run"; return ANX_EINVAL; }
  if (!dst || !used) { err = "bad export arguments"; return ANX_EINVAL; }
  const size_t n = b->n_input;
  const size_t off_bytes = ((n + 1) * sizeof(uint32_t) + 15) & ~(size_t)15;
  *used = off_bytes + (size_t)b->n_results * sizeof(anx_topk_record);
  if (capacity < *used) { err = "export buffer too small: " + std::to_string(*used) + " bytes needed"; return ANX_ELIMIT; }
  HIP_TRY(hipSetDevice(b->device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  b->async_stream = stream; b->async_pending = true;
  uint32_t* d_off = static_cast<uint32_t*>(dst);
  anx_topk_record* d_rows = reinterpret_cast<anx_topk_record*>(static_cast<char*>(dst) + off_bytes);
  if (n == 0 || b->nq == 0) { HIP_TRY(hipMemsetAsync(d_off, 0, off_bytes, st)); return ANX_OK; }
  if (!b->x_cnt) {
    const size_t nblk = (n + SCAN_TILE - 1) / SCAN_TILE + 2;
    HIP_TRY(pool_malloc(reinterpret_cast<void**>(&b->x_cnt), n * sizeof(uint32_t)));
    HIP_TRY(pool_malloc(reinterpret_cast<void**>(&b->x_tmp), nblk * sizeof(uint32_t)));
  }
  const uint32_t n32 = (uint32_t)n, nq32 = (uint32_t)b->nq;
  HIP_TRY(hipMemsetAsync(b->x_cnt, 0, n * sizeof(uint32_t), st));
  hipLaunchKernelGGL(k_fetch_counts, dim3((nq32 + 255) / 256), dim3(256), 0, st, nq32, b->r_count, b->q_orig, b->x_cnt);
  exclusive_scan(b->x_cnt, n32, d_off, b->x_tmp, st);
  hipLaunchKernelGGL(k_export_rows, dim3((nq32 + 255) / 256), dim3(256), 0, st, nq32, b->soff, b->r_count, b->r_rows, b->q_orig, d_off, d_rows);
  HIP_TRY(hipGetLastError());
  return ANX_OK;
}

size_t batch_compact_bytes(const Batch* b) {
  return (((b->n_input + 1) * sizeof(uint32_t) + 15) & ~(size_t)15) + (b->ran ? (size_t)b->n_results : 0) * sizeof(anx_topk_record);
}
int batch_gather_compact(const DeviceLexicon* dl, const Batch* b, int dst_device, void* dst, size_t capacity, void* stream, std::string& err) {
  if (!b->ran) { err = "batch has not been run"; return ANX_EINVAL; }
  const size_t need = batch_compact_bytes(b);
  if (capacity < need) { err = "gather buffer too small: " + std::to_string(need) + " bytes needed for this shard"; return ANX_ELIMIT; }
  size_t used = 0;
  if (b->device == dst_device) {  // already where the rows are wanted
    const int rc = batch_export_compact(dl, b, dst, capacity, stream, &used, err);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
    return ANX_OK;
  }
  HIP_TRY(hipSetDevice(b->device));
  {  // direct access between the two devices where the topology has it (xGMI); without it the peer copy is staged by the runtime
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, b->device, dst_device) == hipSuccess && can) {
      const hipError_t e = hipDeviceEnablePeerAccess(dst_device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
      else (void)hipGetLastError();
    } else (void)hipGetLastError();
  }
  void* tmp = nullptr;
  HIP_TRY(pool_malloc(&tmp, std::max<size_t>(need, 16)));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int rc = batch_export_compact(dl, b, tmp, need, stream, &used, err);
  if (rc == ANX_OK) {
    const hipError_t e = hipMemcpyPeerAsync(dst, dst_device, tmp, b->device, used, st);
    if (e != hipSuccess) { err = std::string("hipMemcpyPeerAsync: ") + hipGetErrorString(e); rc = ANX_ENODEVICE; }
  }
  (void)hipStreamSynchronize(st);
  pool_free(tmp);
  return rc;
}

// Scored pairs per input query, counted by the scan of a PRODUCTION run (pairs that fail the DL's length test are only
// counted there, never materialised): the batch is run once more with the per-query counters switched on.
int batch_pair_counts(const HostModel& m, const DeviceLexicon* dl, Batch* b, uint32_t** out, std::string& err) {
  if (!b->ran) { err = "batch has not been run"; return ANX_EINVAL; }
  HIP_TRY(hipSetDevice(b->device));
  const bool keep = b->keep_all_pairs;
  b->keep_all_pairs = false;
  b->count_pairs = true;
  const int rc = batch_run(m, dl, b, b->last_stream, err);
  b->count_pairs = false;
  b->keep_all_pairs = keep;
  if (rc) return rc;
  { const int rc2 = ensure_order(b, err); if (rc2) return rc2; }
  uint32_t* res = static_cast<uint32_t*>(calloc(std::max<size_t>(1, b->n_input), sizeof(uint32_t)));
  if (!res) { err = "out of memory"; return ANX_EINVAL; }
  if (b->nq) {
    std::vector<uint32_t> h(b->nq);
    if (hipMemcpy(h.data(), b->qpairs, b->nq * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) { free(res); err = "hipMemcpy failed"; return ANX_ENODEVICE; }
    for (size_t s = 0; s < b->nq; ++s) res[b->order[s]] = h[s];
  }
  *out = res;
  return ANX_OK;
}

void batch_set_run_mode(Batch* b, const anx_params& p, int conf_mode) {
  b->params = p;
  b->conf_mode = conf_mode;
}
bool batch_conf_fallback(const Batch* b) { return b->conf_fallback; }
int batch_download_text(const Batch* b, std::string& text, std::vector<uint32_t>& off, std::string& err) {
  if (b->n_input == 0) {  // a shard that received no inputs never allocated its text: it contributes nothing
    text.clear();
    off.assign(1, 0u);
    return ANX_OK;
  }
  if (!b->d_text || !b->d_textoff) { err = "the batch does not hold its inputs"; return ANX_EINVAL; }
  HIP_TRY(hipSetDevice(b->device));
  off.assign(b->n_input + 1, 0u);
  HIP_TRY(hipMemcpy(off.data(), b->d_textoff, (b->n_input + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
  text.assign(b->n_input ? (size_t)off[b->n_input] : 0, '\0');
  if (!text.empty()) HIP_TRY(hipMemcpy(&text[0], b->d_text, text.size(), hipMemcpyDeviceToHost));
  return ANX_OK;
}

void batch_stats(const Batch* b, anx_batch_stats* s) { *s = b->stats; }

void batch_free(Batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  // the export kernels are asynchronous on the caller's stream and read this batch's buffers: they must have finished
  // before the blocks go back to the pool, where another batch / thread / stream may take them at once
  if (b->async_pending) (void)hipStreamSynchronize(reinterpret_cast<hipStream_t>(b->async_stream));
  if (b->launched) (void)hipEventSynchronize(b->ev_done);  // a run enqueued with batch_run_async and never waited for
  for (void* p : {(void*)b->q_cv, (void*)b->q_bits, (void*)b->q_rows, (void*)b->q_rec, (void*)b->q_meta, (void*)b->q_orig, (void*)b->d_tiles, (void*)b->rctr, (void*)b->sctr, (void*)b->surv,
                  (void*)b->counters, (void*)b->qexact, (void*)b->qsurv, (void*)b->soff, (void*)b->qcur,
                  (void*)b->qmaxfreq, (void*)b->d_cold, (void*)b->qpairs, (void*)b->x_cnt, (void*)b->x_tmp, (void*)b->scan_tmp, (void*)b->raw, (void*)b->p_score, (void*)b->p_meta, (void*)b->list8, (void*)b->listg, (void*)b->listw, (void*)b->lctr,
                  (void*)b->c_rows, (void*)b->qexpand, (void*)b->r_rows, (void*)b->t_key, (void*)b->r_count, (void*)b->r_off,
                  (void*)b->d_text, (void*)b->d_textoff, (void*)b->cf_weight, (void*)b->cf_need, (void*)b->cf_ctr, b->cf_work, (void*)b->cf_sort, b->cf_sort_tmp})
    if (p) pool_free(p);
  shell_release(b);  // events + pinned read-back block: to the device's pool (hipHostFree / hipEventDestroy here would wait for the device)
  delete b;
}

#include "small_path.hpp"

// ---- test hook: the band-match bound alone (anx_debug_band_bound; tests/test_gpu_switches.py) ------------------------------------
// One lane per pair, rows as the kernels see them (first 16 symbols, query padded with 0xFE, candidate with 0xFF); the three forms
// in use: 0 the scan's fused filter (B7 masks, wave-uniform d, words by the wave's longest string), 1 k_filter_score's (B7, per-lane
// d), 2 the general zero test (alphabets above 124 classes).  out[i] = 1: the bound says damerau_levenshtein(q, c) > d.
__global__ __launch_bounds__(64) void k_debug_band_bound(const uint4* __restrict__ q, const uint4* __restrict__ c, const uint8_t* __restrict__ lqs,
                                                         const uint8_t* __restrict__ lcs, uint32_t n, int d, int form, uint8_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * 64u + threadIdx.x;
  const bool act = i < n;
  const uint4 Q = act ? q[i] : make_uint4(0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu);
  const uint4 C = act ? c[i] : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
  const int lq = act ? lqs[i] : 1, lc = act ? lcs[i] : 1;
  const int ml = lq > lc ? lq : lc;
  bool rej;
  if (form == 2) {
    constexpr uint32_t M = 0xFFFFFFFFu;
    if (__any(act && ml > 12)) { const uint32_t q4[4] = {Q.x, Q.y, Q.z, Q.w}, c6[6] = {M, C.x, C.y, C.z, C.w, M}; rej = band_bound_rejects<4, false>(q4, c6, act, d, lq, lc); }
    else if (__any(act && ml > 8)) { const uint32_t q3[3] = {Q.x, Q.y, Q.z}, c5[5] = {M, C.x, C.y, C.z, M}; rej = band_bound_rejects<3, false>(q3, c5, act, d, lq, lc); }
    else { const uint32_t q2[2] = {Q.x, Q.y}, c4[4] = {M, C.x, C.y, M}; rej = band_bound_rejects<2, false>(q2, c4, act, d, lq, lc); }
  } else {
    constexpr uint32_t M = 0x7F7F7F7Fu;
    const uint32_t q4[4] = {Q.x & M, Q.y & M, Q.z & M, Q.w & M}, c6[6] = {M, C.x & M, C.y & M, C.z & M, C.w & M, M};
    const uint32_t q3[3] = {q4[0], q4[1], q4[2]}, c5[5] = {M, c6[1], c6[2], c6[3], M};
    const uint32_t q2[2] = {q4[0], q4[1]}, c4[4] = {M, c6[1], c6[2], M};
    if (form == 0) {
      if (__any(act && ml > 12)) rej = band_bound_rejects<4, true, true>(q4, c6, act, d, lq, lc);
      else if (__any(act && ml > 8)) rej = band_bound_rejects<3, true, true>(q3, c5, act, d, lq, lc);
      else rej = band_bound_rejects<2, true, true>(q2, c4, act, d, lq, lc);
    } else {
      if (__any(act && ml > 12)) rej = band_bound_rejects<4, true>(q4, c6, act, d, lq, lc);
      else if (__any(act && ml > 8)) rej = band_bound_rejects<3, true>(q3, c5, act, d, lq, lc);
      else rej = band_bound_rejects<2, true>(q2, c4, act, d, lq, lc);
    }
  }
  if (act) out[i] = rej ? 1 : 0;
}
int debug_band_bound(int device, const uint8_t* q_rows, const uint8_t* c_rows, const uint8_t* lq, const uint8_t* lc, size_t n, int d, int form,
                     uint8_t* out, std::string& err) {
  if (n == 0) return ANX_OK;
  if (n > (1u << 30) || d < 0 || d > 3 || form < 0 || form > 2) { err = "band bound: bad arguments"; return ANX_EINVAL; }
  HIP_TRY(hipSetDevice(device));
  uint4 *dq = nullptr, *dc = nullptr;
  uint8_t *dlq = nullptr, *dlc = nullptr, *dout = nullptr;
  int rc = ANX_OK;
  auto cleanup = [&]() { for (void* p : {(void*)dq, (void*)dc, (void*)dlq, (void*)dlc, (void*)dout}) if (p) (void)hipFree(p); };
  auto fail_hip = [&](hipError_t e) { err = std::string("HIP: ") + hipGetErrorString(e); cleanup(); return ANX_ENODEVICE; };
  hipError_t e;
  if ((e = hipMalloc(reinterpret_cast<void**>(&dq), n * 16)) != hipSuccess || (e = hipMalloc(reinterpret_cast<void**>(&dc), n * 16)) != hipSuccess ||
      (e = hipMalloc(reinterpret_cast<void**>(&dlq), n)) != hipSuccess || (e = hipMalloc(reinterpret_cast<void**>(&dlc), n)) != hipSuccess ||
      (e = hipMalloc(reinterpret_cast<void**>(&dout), n)) != hipSuccess)
    return fail_hip(e);
  if ((e = hipMemcpy(dq, q_rows, n * 16, hipMemcpyHostToDevice)) != hipSuccess || (e = hipMemcpy(dc, c_rows, n * 16, hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(dlq, lq, n, hipMemcpyHostToDevice)) != hipSuccess || (e = hipMemcpy(dlc, lc, n, hipMemcpyHostToDevice)) != hipSuccess)
    return fail_hip(e);
  hipLaunchKernelGGL(k_debug_band_bound, dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, 0, dq, dc, dlq, dlc, (uint32_t)n, d, form, dout);
  if ((e = hipDeviceSynchronize()) != hipSuccess || (e = hipMemcpy(out, dout, n, hipMemcpyDeviceToHost)) != hipSuccess) return fail_hip(e);
  cleanup();
  return rc;
}

}  // namespace anx
