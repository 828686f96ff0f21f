// small_path.hpp -- the small call: find_variants for a handful of inputs at the reference's own granularity
// Part of the single translation unit engine.hip (included inside namespace anx, after the batch pipeline); gfx950 only.
//
// The reference's callers hand find_variants ONE string (/root/reference/src/lib.rs:972) and fan out in batches of 1 000
// (src/bin/analiticcl.rs:416,445-448; bindings/python/src/lib.rs:704-749).  The batch pipeline above is built for a million queries:
// ~45 launches and 5 host waits in its encoder (radix sorts, tile sort), ~35 commands and a dozen pool allocations per run, events,
// a pageable download -- 0.3 ms for one query, 0.65 ms for a thousand (round 6, tools/fresh_batch.py small).  A call of at most
// SMALL_MAX inputs of at most SMALL_MAX_BYTES bytes takes this path instead:
//   * a context (stream, pinned staging, every device buffer at its fixed capacity) is checked out of a per-device pool: no allocation,
//     no event, no memset command per call;
//   * the inputs are packed into pinned memory the encoder kernels read over PCIe; queries stay in INPUT order (no sort, no
//     permutation), one scan tile per query (k_small_tiles), capacities are fixed and every append is bounds-checked as always;
//   * nine launches -- k_enc_strings, k_small_tiles (with k_enc_gather's work) | k_scan_small (all three scan bodies in one launch) |
//     k_filter_score, k_small_lists (k_filter_wide + k_score_fast8 + k_score_pairs, a block per region) | k_small_offsets,
//     k_compact_grouped, k_rank | k_small_fetch (rows and offsets straight into pinned host memory, with the run's fills) -- and ONE host wait
//     (eleven in the first version: 66 us for one input);
//   * a run whose fills exceeded a fixed capacity (a handful of very short queries can) is discarded and the call takes the batch path.
// Same kernels, same arithmetic as the batch path: the results are identical (tests/test_gpu_small.py: against the batch path and the oracle).
#pragma once

constexpr uint32_t SMALL_MAX_BYTES = 64;     // longest input (bytes) the small path takes: query rows of <= 4 words
constexpr uint32_t SMALL_SHIFT = 15;         // pair-list slots per region (2 M slots in all)
constexpr uint32_t SMALL_SURV_CAP = 16384;   // survivor records / slot-list entries per region
constexpr uint32_t SMALL_LIST_BLOCKS = 2;    // blocks per region of the slot-list kernels (LIST_P of the batch path: 64)
constexpr uint32_t SMALL_FS_BLK = 512;       // pair-list slots per block of k_filter_score (FS_BLK = 4096 in the batch path)
constexpr uint32_t SMALL_FETCH_BLOCKS = 8;   // blocks of k_small_fetch: each computes the offsets, copies a share of the rows
constexpr uint32_t SMALL_ROWS_PER_Q = 16;    // candidate rows per query the row buffers hold on average

struct SmallCtx {
  int device = 0;
  hipStream_t st = nullptr;
  char* h_in = nullptr;        // pinned: [blob | offsets u32[SMALL_MAX + 1]]
  char* h_out = nullptr;       // pinned: [SmallCtl | offsets u64[SMALL_MAX + 1]]
  SmallEnc enc;
  uint32_t *counters = nullptr, *rctr = nullptr, *sctr = nullptr, *lctr = nullptr, *qsurv = nullptr, *soff = nullptr, *qcur = nullptr, *qmaxfreq = nullptr, *qexpand = nullptr,
           *r_count = nullptr;
  uint2* raw = nullptr;
  SurvRec* surv = nullptr;
  uint32_t *list8 = nullptr, *listg = nullptr, *listw = nullptr;
  SurvRow* c_rows = nullptr;
  DevRow* r_rows = nullptr;
  double* t_key = nullptr;
  FsCold* d_cold = nullptr;
  FsCold h_cold_last;          // what d_cold holds (uploaded again only when it changes: cold_key)
  double cold_key[12] = {};
  bool cold_valid = false;
  int nplanes = 0;             // count-vector dwords the buffers were sized for
  std::vector<void*> blocks;   // device allocations (pool)
  size_t row_cap = 0;
};
constexpr size_t SMALL_IN_BLOB = (size_t)SMALL_MAX * (SMALL_MAX_BYTES + 1) + 64;

static void small_ctx_destroy(SmallCtx* c) {  // (the current device is the context's)
  if (!c) return;
  if (c->st) (void)hipStreamSynchronize(c->st);
  for (void* p : c->blocks) pool_free(p);
  if (c->h_in) (void)hipHostFree(c->h_in);
  if (c->h_out) (void)hipHostFree(c->h_out);
  if (c->st) (void)hipStreamDestroy(c->st);
  delete c;
}
static SmallCtx* small_ctx_create(const DeviceLexicon* dl, std::string& err) {
  std::unique_ptr<SmallCtx, void (*)(SmallCtx*)> c(new SmallCtx(), small_ctx_destroy);
  c->device = dl->device;
  c->nplanes = 42;  // the widest count vector: a context serves every model of the device
  c->st = static_cast<hipStream_t>(make_stream(true));
  if (!c->st) { err = "small path: no stream"; return nullptr; }
  auto pinned = [&](char** p, size_t bytes) { return hipHostMalloc(reinterpret_cast<void**>(p), bytes, hipHostMallocDefault) == hipSuccess; };
  if (!pinned(&c->h_in, SMALL_IN_BLOB + (SMALL_MAX + 1) * sizeof(uint32_t)) || !pinned(&c->h_out, 64 + (SMALL_MAX + 1) * sizeof(unsigned long long))) {
    (void)hipGetLastError();
    err = "small path: pinned staging";
    return nullptr;
  }
  bool ok = true;
  auto dev = [&](auto** p, size_t count) {
    void* q = nullptr;
    if (!ok || pool_malloc(&q, std::max<size_t>(count * sizeof(**p), 16)) != hipSuccess) { ok = false; return; }
    c->blocks.push_back(q);
    *p = static_cast<std::remove_reference_t<decltype(*p)>>(q);
  };
  const size_t N = SMALL_MAX, NP = (size_t)c->nplanes, QW = (SMALL_MAX_BYTES + 15) / 16;
  SmallEnc& e = c->enc;
  dev(&e.codes, SMALL_IN_BLOB + 4 * N + 16); dev(&e.meta, N); dev(&e.bits, N * NBITPLANES); dev(&e.kind, N); dev(&e.cv, N * NP); dev(&e.blk, 3 * (N / 256 + 1));
  dev(&e.perm, N); dev(&e.key, N); dev(&e.sig, N);
  dev(&e.q_rec, 2 * N); dev(&e.q_rows, N * QW); dev(&e.q_bits, N * NBITPLANES); dev(&e.q_cv, N * NP); dev(&e.q_meta, N); dev(&e.q_orig, N); dev(&e.qexact, N);
  dev(&e.s_kind, N); dev(&e.s_sig, N); dev(&e.tiles, 8 * N);
  dev(&c->counters, CTR_N); dev(&c->rctr, SCAN_REGIONS * RC_STRIDE); dev(&c->sctr, SCAN_REGIONS * RC_STRIDE); dev(&c->lctr, 3 * SCAN_REGIONS * RC_STRIDE);
  dev(&c->qsurv, N); dev(&c->soff, N + 1); dev(&c->qcur, N); dev(&c->qmaxfreq, N); dev(&c->qexpand, N); dev(&c->r_count, N);
  dev(&c->raw, (size_t)SCAN_REGIONS << SMALL_SHIFT);
  dev(&c->surv, (size_t)SCAN_REGIONS * SMALL_SURV_CAP);
  dev(&c->list8, (size_t)SCAN_REGIONS * SMALL_SURV_CAP); dev(&c->listg, (size_t)SCAN_REGIONS * SMALL_SURV_CAP); dev(&c->listw, (size_t)SCAN_REGIONS * SMALL_SURV_CAP);
  c->row_cap = N * SMALL_ROWS_PER_Q + 1024;
  dev(&c->c_rows, c->row_cap); dev(&c->r_rows, c->row_cap); dev(&c->t_key, c->row_cap);
  dev(&c->d_cold, 1);
  if (!ok) { (void)hipGetLastError(); err = "small path: device buffers"; return nullptr; }
  if (small_iota(e.perm, SMALL_MAX, c->st) != ANX_OK || hipStreamSynchronize(c->st) != hipSuccess) { err = "small path: set-up kernel"; return nullptr; }
  return c.release();
}
static SmallCtx* small_ctx_acquire(const DeviceLexicon* dl, std::string& err) {
  DevPool& pl = pool_of(dl->device);
  {
    std::lock_guard<std::mutex> g(pl.mu);
    if (!pl.small_idle.empty()) { SmallCtx* c = pl.small_idle.back(); pl.small_idle.pop_back(); return c; }
  }
  return small_ctx_create(dl, err);
}
static void small_ctx_release(SmallCtx* c) {
  DevPool& pl = pool_of(c->device);
  std::lock_guard<std::mutex> g(pl.mu);
  pl.small_idle.push_back(c);
}
static void small_ctxs_destroy(int device) {  // (the current device is `device`)
  DevPool& pl = pool_of(device);
  std::vector<SmallCtx*> drop;
  { std::lock_guard<std::mutex> g(pl.mu); drop.swap(pl.small_idle); }
  for (SmallCtx* c : drop) small_ctx_destroy(c);
}

static std::atomic<uint64_t> g_small_taken{0}, g_small_overflow{0};
void small_stats(uint64_t* out) { out[0] = g_small_taken.load(); out[1] = g_small_overflow.load(); }

// 0: done (*out_rows: a block of the pinned result cache, *out_offs: malloc'd); 1: not taken (the caller uses the batch path);
// negative: an error of the device
int small_find(const HostModel& m, const DeviceLexicon* dl, const char* const* utf8, size_t n, const anx_params& p, anx_result** out_rows, size_t** out_offs,
               std::string& err) {
  if (!dl || n == 0 || n > SMALL_MAX || !switches().small_path || dl->any_variants || p.stop_at_exact_match || dl->nplanes > 42) return 1;
  uint32_t lens[SMALL_MAX];
  uint32_t maxbytes = 0;
  for (size_t i = 0; i < n; ++i) {
    const size_t l = utf8[i] ? strlen(utf8[i]) : 0;
    if (l > SMALL_MAX_BYTES) return 1;
    lens[i] = (uint32_t)l;
    maxbytes = std::max(maxbytes, (uint32_t)l);
  }
  if (hipSetDevice(dl->device) != hipSuccess) { err = "hipSetDevice failed"; return ANX_ENODEVICE; }
  SmallCtx* c = small_ctx_acquire(dl, err);
  if (!c) { (void)hipGetLastError(); return 1; }  // (no context: the batch path still works)
  struct Release { SmallCtx* c; ~Release() { small_ctx_release(c); } } rel{c};
  hipStream_t st = c->st;
  // ---- inputs -> pinned staging -----------------------------------------------------------------------------------------------------
  uint32_t* h_off = reinterpret_cast<uint32_t*>(c->h_in + SMALL_IN_BLOB);
  {
    size_t pos = 0;
    for (size_t i = 0; i < n; ++i) {
      h_off[i] = (uint32_t)pos;
      if (lens[i]) memcpy(c->h_in + pos, utf8[i], lens[i]);
      c->h_in[pos + lens[i]] = '\0';
      pos += (size_t)lens[i] + 1;
    }
    h_off[n] = (uint32_t)pos;
    memset(c->h_in + pos, 0, 16);  // (the encoder's 16-byte window may read past the last string)
  }
  const uint32_t n32 = (uint32_t)n;
  const uint32_t qw = std::max<uint32_t>(1u, (maxbytes + 15u) / 16u);                      // symbols <= bytes
  const uint32_t d = (uint32_t)clamp_threshold(p.max_edit_distance, (int)maxbytes, kMaxEditDistance);  // >= every query's clamped d (monotone in the length)
  // ---- the caller's rows: a block of the pinned result cache the last kernel writes into -----------------------------------------
  const size_t row_cap = std::min<size_t>(c->row_cap, n * (size_t)SMALL_ROWS_PER_Q + 64);
  anx_result* rows = static_cast<anx_result*>(host_result_alloc(row_cap * sizeof(anx_result)));
  if (!rows) return 1;
  if (!host_result_is_pinned(rows)) { host_result_free(rows); return 1; }  // (pinning failed: the kernel could not write into it)
  SmallCtl* h_ctl = reinterpret_cast<SmallCtl*>(c->h_out);
  unsigned long long* h_off64 = reinterpret_cast<unsigned long long*>(c->h_out + 64);
  h_ctl->rows = 0xFFFFFFFEu;
  // ---- encode + tiles (+ the counters cleared) ----------------------------------------------------------------------------------------
  SmallZero z{};
  {
    uint32_t* zp[8] = {c->counters, c->rctr, c->sctr, c->lctr, c->qsurv, c->qmaxfreq, c->qexpand, nullptr};
    const uint32_t zn[8] = {CTR_N, SCAN_REGIONS * RC_STRIDE, SCAN_REGIONS * RC_STRIDE, 3 * SCAN_REGIONS * RC_STRIDE, n32, n32, n32, 0u};
    for (int i = 0; i < 8; ++i) { z.p[i] = zp[i]; z.n[i] = zn[i]; }
  }
  // the encoder kernels read the pinned staging buffer themselves (k_enc_strings<true>: a coalesced burst per block into LDS): a copy
  // command ahead of the first kernel cost 15-18 us of the call (round 6 traces)
  const uint8_t* in_blob = reinterpret_cast<const uint8_t*>(c->h_in);
  const uint32_t* in_off = h_off;
  // tile slots per query: 8 at 4096 inputs, up to 32 for the smallest calls (the rows of a query's adjacency list are shared out over them)
  // (a list of R rows is cut into min(slots, R / 8) parts: 2-3 for the typical list; every unused slot is still a wave that starts and
  // returns: 1 000 inputs 164 -> 156 us with 8 instead of 32 slots per query)
  const uint32_t slots = n32 <= 128u ? 32u : n32 <= 512u ? 16u : 8u;
  int rc = small_encode_launch(m, dl, c->enc, in_blob, in_off, n32, qw, p, z, slots, true, h_off, st, err);
  if (rc) { host_result_free(rows); return rc; }
  // ---- scan -----------------------------------------------------------------------------------------------------------------------------
  const uint32_t region_cap = 1u << SMALL_SHIFT;
  {
    ScanArgs A;
    A.tiles = c->enc.tiles; A.ntiles = slots * n32; A.q_bits = c->enc.q_bits; A.q_cv = c->enc.q_cv;
    A.cls_bits = dl->cls_bits; A.cls_planes = dl->cls_planes; A.scan_rec = dl->scan_rec; A.scan_rec34 = dl->scan_rec34; A.pad_rec = dl->nentries; A.cstride = dl->cstride; A.pad_class = dl->nclasses;
    A.cls_len = dl->cls_len; A.cls_off = dl->cls_off; A.sig = dl->sig; A.sig_e = dl->sig_e; A.sig_cbeg = dl->sig_cbeg; A.sighash = dl->sighash; A.sighash_e = dl->sighash_e; A.hash_mask = dl->hash_mask; A.ball = dl->ball;
    A.adj_hdr = dl->adj_hdr; A.adj_planes = dl->adj_planes; A.adj_ids = dl->adj_ids;
    A.chunk = 64; A.chunk_fused = 32;  // (SCAN_CHUNK / SCAN_CHUNK_FUSED of the batch path: 256 / 128 -- a wave here holds a share of ONE query's pairs)
    A.raw = c->raw; A.region_cap = region_cap; A.rctr = c->rctr; A.qexact = c->enc.qexact; A.want_exact = 0; A.drop_len = 1;
    A.q_rec = c->enc.q_rec; A.e_rec = dl->e_rec;

    A.fuse = (switches().fuse_prefilter && switches().prefilter) ? 1 : 0;
    A.qpairs = nullptr; A.dbg = 0;
    const dim3 grid((A.ntiles + 3) / 4);
    switch (dl->nplanes) {
      case 8: hipLaunchKernelGGL(k_scan_small<8>, grid, dim3(256), 0, st, A); break;
      case 16: hipLaunchKernelGGL(k_scan_small<16>, grid, dim3(256), 0, st, A); break;
      case 24: hipLaunchKernelGGL(k_scan_small<24>, grid, dim3(256), 0, st, A); break;
      case 32: hipLaunchKernelGGL(k_scan_small<32>, grid, dim3(256), 0, st, A); break;
      default: hipLaunchKernelGGL(k_scan_small<42>, grid, dim3(256), 0, st, A); break;
    }
  }
  // ---- score (the launch logic of batch_launch, fixed capacities) -----------------------------------------------------------------------
  ScoreArgs sa;
  sa.dbg = 0;
  sa.quot = dl->quot;
  sa.store_pairs = 0;
  sa.w_ld = m.weights.ld; sa.w_lcs = m.weights.lcs; sa.w_prefix = m.weights.prefix; sa.w_suffix = m.weights.suffix; sa.w_case = m.weights.casew;
  sa.w_sum = m.weights.ld + m.weights.lcs + m.weights.prefix + m.weights.suffix + m.weights.casew;
  sa.score_threshold = p.score_threshold;
  sa.have_freq = m.have_freq ? 1 : 0;
  sa.any_variants = 0;
  sa.lqp = qw * 16;
  sa.lcp = (dl->max_len + 15) / 16 * 16;
  uint32_t stride = sa.lqp + sa.lcp + (d + 2) * (2 * d + 3);
  stride = (stride + 3) / 4;
  if ((stride & 1) == 0) stride++;
  sa.stride = stride * 4;
  sa.qw = qw;
  uint32_t threads = 256;
  while (threads > 64 && (size_t)threads * sa.stride > 64 * 1024) threads >>= 1;
  if ((size_t)threads * sa.stride > 64 * 1024) { host_result_free(rows); (void)hipStreamSynchronize(st); return 1; }
  const SurvOut so{c->surv, c->sctr, SMALL_SURV_CAP};
  const bool have_long_q = qw > 1;
  const int enable_filter = switches().prefilter, enable_fast = switches().score_fast;
  const int fastD = (enable_fast && d >= 1 && d <= 3) ? (int)d : 0;
  const SlotList l8{c->list8, c->lctr, SMALL_SURV_CAP}, lg{c->listg, c->lctr + SCAN_REGIONS * RC_STRIDE, SMALL_SURV_CAP}, lw{c->listw, c->lctr + 2 * SCAN_REGIONS * RC_STRIDE, SMALL_SURV_CAP};
  const PairArgs pa{c->raw, c->enc.q_meta, c->enc.q_rows, c->enc.q_rec, dl->e_rec, dl->ent_meta, dl->ent_rowoff, dl->rows, dl->ent_freq, dl->ent_var_off,
                    nullptr, nullptr, c->qmaxfreq, c->qsurv, c->qexpand, dl->e_planes};
  // slots per region the scoring grid covers: the whole region from a few hundred inputs on, less for the smallest calls (a region filled
  // beyond it hands the call to the batch path, like every other capacity)
  uint32_t fs_cap = 2048;
  while (fs_cap < region_cap && fs_cap < n32 * 12u + 2048u) fs_cap *= 2u;
  FilterArgs fa;
  fa.region_shift = SMALL_SHIFT; fa.rctr = c->rctr; fa.qexact = c->enc.qexact; fa.stop = 0; fa.enable = enable_filter;
  // pairs with a string of 17..32 symbols go to the 8-word register DL also when no QUERY is that long (the batch path leaves a short-query
  // batch's few long candidates to the general LDS kernel: there its extra launch costs more than it saves; here both run inside k_small_lists and
  // one round of the general kernel is 25 us of a 170 us call)
  const bool use8 = fastD > 0 && threads == 256;
  fa.use_nw8 = (have_long_q || use8) ? 1 : 0; fa.counters = c->counters; fa.stat_ctr = c->sctr; fa.fill_cap = fs_cap; fa.blk = SMALL_FS_BLK;
  {
    // k_filter_score's rarely used arguments live in device memory (FsCold): uploaded again only when they change (another model,
    // other weights / thresholds / row width) -- compared field by field (struct padding is not)
    const double key[12] = {sa.w_ld, sa.w_lcs, sa.w_prefix, sa.w_suffix, sa.w_case, sa.w_sum, sa.score_threshold, (double)sa.have_freq, (double)sa.lqp, (double)sa.lcp, (double)sa.stride,
                            (double)sa.qw + 1e3 * (double)(reinterpret_cast<uintptr_t>(sa.quot) & 0xFFFFFFFFu)};
    if (!c->cold_valid || memcmp(key, c->cold_key, sizeof key) != 0) {
      memcpy(c->cold_key, key, sizeof key);
      c->h_cold_last = FsCold{sa, so, l8, lg, lw};
      if (hipMemcpyAsync(c->d_cold, &c->h_cold_last, sizeof(FsCold), hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        host_result_free(rows);
        err = "small path: argument upload";
        return ANX_ENODEVICE;
      }
      c->cold_valid = true;
    }
  }
  const dim3 fgrid(((fs_cap + SMALL_FS_BLK - 1) / SMALL_FS_BLK) * SCAN_REGIONS);
  const bool split_wide = switches().fs_split != 0;
  const bool b7 = switches().fs_b7 && m.alphabet.size() + 1 < 0x7E;
  const bool planes = b7 && switches().fs_planes && (int)m.alphabet.size() <= kSymbolPlanesMaxA;
#define ANX_FS_LAUNCH(DD, WW, BB) hipLaunchKernelGGL((k_filter_score<DD, WW, BB>), fgrid, dim3(256), 0, st, fa, pa, static_cast<const FsCold*>(c->d_cold))
#define ANX_FS_PICK(WW, BB)                        \
  do {                                             \
    if (fastD == 1) ANX_FS_LAUNCH(1, WW, BB);      \
    else if (fastD == 2) ANX_FS_LAUNCH(2, WW, BB); \
    else if (fastD == 3) ANX_FS_LAUNCH(3, WW, BB); \
    else ANX_FS_LAUNCH(0, WW, BB);                 \
  } while (0)
  if (split_wide) { if (planes) ANX_FS_PICK(false, 2); else if (b7) ANX_FS_PICK(false, 1); else ANX_FS_PICK(false, 0); }
  else { if (b7) ANX_FS_PICK(true, 1); else ANX_FS_PICK(true, 0); }
#undef ANX_FS_PICK
#undef ANX_FS_LAUNCH
  if (threads == 256) {  // the slot-list kernels as one launch, a block per region (k_small_lists)
    SmallListArgs L{lw, l8, lg, (split_wide && enable_filter) ? 1 : 0, (fastD && (have_long_q || use8)) ? 1 : 0, fastD};
    const dim3 lgrid(SCAN_REGIONS);
    const size_t dyn = (size_t)threads * sa.stride;
    if (fastD == 1) hipLaunchKernelGGL(k_small_lists<1>, lgrid, dim3(256), dyn, st, L, fa, pa, sa, so);
    else if (fastD == 2) hipLaunchKernelGGL(k_small_lists<2>, lgrid, dim3(256), dyn, st, L, fa, pa, sa, so);
    else if (fastD == 3) hipLaunchKernelGGL(k_small_lists<3>, lgrid, dim3(256), dyn, st, L, fa, pa, sa, so);
    else hipLaunchKernelGGL(k_small_lists<0>, lgrid, dim3(256), dyn, st, L, fa, pa, sa, so);
  } else {
    const dim3 lgrid(SMALL_LIST_BLOCKS * SCAN_REGIONS);
    if (split_wide && enable_filter) hipLaunchKernelGGL(k_filter_wide, lgrid, dim3(256), 0, st, lw, fa, pa, sa, fastD, l8, lg);
    if (fastD && have_long_q) {
      if (fastD == 1) hipLaunchKernelGGL(k_score_fast8<1>, lgrid, dim3(256), 0, st, l8, pa, sa, so);
      else if (fastD == 2) hipLaunchKernelGGL(k_score_fast8<2>, lgrid, dim3(256), 0, st, l8, pa, sa, so);
      else hipLaunchKernelGGL(k_score_fast8<3>, lgrid, dim3(256), 0, st, l8, pa, sa, so);
    }
    hipLaunchKernelGGL(k_score_pairs, lgrid, dim3(threads), threads * sa.stride, st, lg, pa, sa, so);
  }
  // ---- compact + rank + the rows into the caller's block ---------------------------------------------------------------------------------
  RankArgs ra;
  ra.cutoff_threshold = p.cutoff_threshold;
  ra.max_matches = p.max_matches;
  ra.freq_weight = p.freq_weight;
  ra.have_freq = m.have_freq ? 1 : 0;
  ra.any_variants = 0;
  const uint32_t crow_cap = (uint32_t)c->row_cap;
  hipLaunchKernelGGL(k_small_offsets, dim3(1), dim3(SMALL_T), 0, st, c->qsurv, n32, c->soff, c->qcur);
  hipLaunchKernelGGL(k_compact_grouped, dim3(SCAN_REGIONS), dim3(COMPACT_B), 0, st, c->surv, c->sctr, SMALL_SURV_CAP, m.have_freq ? 1 : 0, c->qcur, dl->ent_rec, c->c_rows,
                     c->soff + n32, crow_cap, c->counters + CTR_OVERFLOW);
  ANX_RANK_LAUNCH(dim3((n32 + 4 * RANK_QPW - 1) / (4 * RANK_QPW)), dim3(256), 0, st, n32, c->soff, c->c_rows, c->qmaxfreq, c->qexpand, ra, c->t_key, c->r_rows, c->r_count, crow_cap,
                  c->counters + CTR_OVERFLOW);
  hipLaunchKernelGGL(k_small_fetch, dim3(n32 > 256u ? SMALL_FETCH_BLOCKS : 1u), dim3(SMALL_T), 0, st, n32, c->soff, c->r_count, c->r_rows, c->rctr, c->sctr, c->lctr, c->counters, h_off64, rows, (uint32_t)row_cap, crow_cap, h_ctl);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
    host_result_free(rows);
    err = std::string("small path: ") + hipGetErrorString(hipGetLastError());
    return ANX_ENODEVICE;
  }
  // ---- did the run fit the fixed capacities? -----------------------------------------------------------------------------------------------
  const SmallCtl ctl = *h_ctl;
  if (ctl.rows > row_cap || ctl.maxfill > fs_cap || ctl.surv_fill > SMALL_SURV_CAP || ctl.list_fill > SMALL_SURV_CAP || ctl.total_surv > crow_cap || ctl.overflow) {
    host_result_free(rows);
    g_small_overflow.fetch_add(1, std::memory_order_relaxed);
    return 1;  // the batch path sizes its buffers from what it measures
  }
  size_t* offs = static_cast<size_t*>(malloc((n + 1) * sizeof(size_t)));
  if (!offs) { host_result_free(rows); err = "out of memory"; return ANX_EINVAL; }
  static_assert(sizeof(size_t) == sizeof(unsigned long long), "offsets are copied as they are");
  memcpy(offs, h_off64, (n + 1) * sizeof(size_t));
  *out_rows = rows;
  *out_offs = offs;
  g_small_taken.fetch_add(1, std::memory_order_relaxed);
  return ANX_OK;
}
