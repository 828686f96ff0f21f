// capi.cpp -- the extern "C" boundary declared in include/anx.h.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <charconv>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>

#include "engine.h"
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include "host_model.h"

// One replica of the lexicon = one device-resident copy + the stream and the host thread that drive it.  A model built with
// anx_model_to_devices holds several (normally one per GPU of the node): the reference's fan-out over inputs
// (src/bin/analiticcl.rs:445-448, src/lib.rs:1883) becomes contiguous input ranges, one per replica, run concurrently.
namespace {
struct Worker {  // a persistent host thread per replica: every HIP call of a shard is made from its replica's thread
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::function<void()>> q;
  bool stop = false;
  Worker() {
    th = std::thread([this]() {
      for (;;) {
        std::function<void()> f;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [this]() { return stop || !q.empty(); });
          if (q.empty()) return;  // stop requested and nothing left
          f = std::move(q.front());
          q.pop_front();
        }
        f();
      }
    });
  }
  ~Worker() {
    { std::lock_guard<std::mutex> g(mu); stop = true; }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
  void post(std::function<void()> f) {
    { std::lock_guard<std::mutex> g(mu); q.push_back(std::move(f)); }
    cv.notify_one();
  }
};
struct Replica {
  anx::DeviceLexicon* dev = nullptr;
  int device = -1;
  void* stream = nullptr;              // non-blocking stream of the replica (multi-replica batches run on it)
  std::unique_ptr<Worker> worker;      // only with more than one replica
};
struct Shard {  // the part of a batch one replica holds: inputs [lo, lo + n) of the call, or -- length-partitioned split -- the inputs idx[0 .. n)
  int replica = 0;
  anx::Batch* b = nullptr;
  size_t lo = 0, n = 0;
  std::vector<uint32_t> idx;  // ascending original indices of the shard's inputs; empty = the consecutive range
  std::vector<uint32_t> lhist;  // length-partitioned split: inputs per byte-length class, and what the split predicted they cost
  double predicted = 0.0;
  size_t input(size_t i) const { return idx.empty() ? lo + i : idx[i]; }
};
}  // namespace

// What one input of a byte length costs the device, relative to the other lengths: the weights of the length-partitioned split.
// prior x learned correction: after every run of a length-partitioned batch the correction of the lengths a shard held moves
// towards (its measured device time / its predicted cost), so the shares of the following calls even out whatever the lexicon,
// the alphabet and the parameters are (LengthCost::learn; the corrections start over when the distance parameters change).
struct LengthCost {
  static constexpr uint32_t LMAX = 256;  // byte lengths >= 255 share the last class
  std::mutex mu;
  double scale[LMAX];
  // prior from the signature adjacency lists (anx_model_to_devices): records the scan tests per query of every length, averaged over
  // the lexicon's entries of that length; have_records = false: the window-size prior below
  double records[LMAX] = {};
  std::vector<float> class_nsig;      // lexicon signatures per split class (tile fill of a batch's class)
  std::vector<float> class_records;   // by split class (length < 64: length * 1024 + group sum 0 * 32 + group sum 1); 0 = no entry there
  bool have_records = false;
  unsigned updates = 0;   // learning steps since the corrections started over
  anx_threshold k_of, d_of;
  bool init = false;
};
struct anx_model {
  anx::HostModel host;
  anx::DeviceLexicon* dev = nullptr;   // == replicas[0].dev
  std::vector<Replica> replicas;
  mutable LengthCost len_cost;
};
struct anx_batch {
  const anx_model* model = nullptr;
  std::vector<Shard> shards;      // one per replica that got inputs (never empty): consecutive input ranges in input order, or the
                                  // length-partitioned split (Shard::idx; split_by_length below)
  size_t n_input = 0;
  // confusables loaded: the device ranks without the cutoff (late) or without crop and cutoff (early); the host
  // rescoring in anx_batch_fetch needs the caller's parameters and the input texts
  bool rescore = false;
  // ... unless the weighting runs on the device (conf.hip; default): the rows come back final.  A run that met a row the device
  // could not weight (string beyond its fixed working memory) falls back to the host path for the whole batch.
  bool dev_conf = false;
  mutable bool learned = false;   // the length split's cost model has taken this batch's shard times (learn_from_batch)
  anx_params params;
  std::string in_text;            // the inputs, each followed by a NUL byte (one copy of the caller's buffer, not a string each)
  std::vector<uint32_t> in_off;   // n + 1 offsets into in_text
};

static thread_local std::string g_err;
static thread_local int g_code = 0;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  g_code = code;
  return code;
}

const anx::HostModel& anx_host_of(const anx_model* m) { return m->host; }
const anx::DeviceLexicon* anx_replica_of(const anx_model* m, size_t i) { return i < m->replicas.size() ? m->replicas[i].dev : nullptr; }
int anx_replica_device(const anx_model* m, size_t i) { return i < m->replicas.size() ? m->replicas[i].device : -1; }
int anx_fail(int code, const std::string& msg) { return fail(code, msg); }
// the device batch of a batch that lives on ONE replica (search.cpp's one-pass path reads its device-resident rows); nullptr otherwise
anx::Batch* anx_batch_single(const anx_batch* b) { return (b && b->shards.size() == 1 && !b->rescore) ? b->shards[0].b : nullptr; }


// Confusable rescoring of the ranked lists (src/lib.rs:1505-1508 early, :1591-1595 late) followed by the steps the device
// left out: [early: crop, :1536-1589] and the cutoff (:1598-1622).  rows are compacted in place, offs rewritten.
static double vr_score(const anx_result& r, float fw) {  // src/types.rs:335-341
  if (fw == 0.0f) return r.dist_score;
  return (r.dist_score + ((double)fw * r.freq_score)) / (1.0 + (double)fw);
}
static void rescore_with_confusables(const anx::HostModel& m, const std::string& in_text, const std::vector<uint32_t>& in_off,
                                     const anx_params& p, anx_result* rows, size_t* offs) {
  const float fw = p.freq_weight;
  const bool early = m.confusables_before_pruning;
  const size_t n = in_off.empty() ? 0 : in_off.size() - 1;
  std::vector<size_t> newlen(n, 0);
  // early: the reference weights its candidates in gather order and sorts once (src/lib.rs:1505-1535), so rows that tie after
  // the weighting keep that order.  The device handed the rows over ranked by the unweighted score: put them back first.
  const std::vector<uint32_t>* gorder = early ? &m.vocab_gather_order() : nullptr;
  // every input is independent: edit scripts, re-ranking and cut-off on host threads, in place inside the input's row range
  auto work = [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) {
      anx_result* v = rows + offs[i];
      const size_t cnt = offs[i + 1] - offs[i];
      if (gorder)
        std::stable_sort(v, v + cnt, [&](const anx_result& a, const anx_result& b) {
          const uint64_t ia = a.via != ANX_NO_VIA ? a.via : a.vocab_id, ib = b.via != ANX_NO_VIA ? b.via : b.vocab_id;
          return (ia < gorder->size() ? (*gorder)[ia] : 0xFFFFFFFFu) < (ib < gorder->size() ? (*gorder)[ib] : 0xFFFFFFFFu);
        });
      // the weight belongs to the matched item: the variant itself for rows reached through a variant list
      uint64_t ids[64];
      double wts[64];
      for (size_t k0 = 0; k0 < cnt; k0 += 64) {
        const size_t kn = std::min<size_t>(64, cnt - k0);
        for (size_t k = 0; k < kn; ++k) ids[k] = v[k0 + k].via != ANX_NO_VIA ? v[k0 + k].via : v[k0 + k].vocab_id;
        m.confusable_weights(in_text.data() + in_off[i], (size_t)(in_off[i + 1] - in_off[i]) - 1, ids, kn, wts);
        for (size_t k = 0; k < kn; ++k) v[k0 + k].dist_score *= wts[k];
      }
      std::stable_sort(v, v + cnt, [&](const anx_result& a, const anx_result& b) {  // rank_cmp, src/types.rs:344-365
        if (fw > 0.0f) return vr_score(a, fw) > vr_score(b, fw);
        if (a.dist_score != b.dist_score) return a.dist_score > b.dist_score;
        return a.freq_score > b.freq_score;
      });
      size_t len = cnt;
      if (early && p.max_matches > 0 && len > p.max_matches) {  // crop with the tie rule
        const size_t mm = (size_t)p.max_matches;
        const double last = vr_score(v[mm - 1], fw), cropped = vr_score(v[mm], fw);
        if (cropped < last) len = mm;
        else {
          size_t early_cut = 0, late_cut = 0;
          for (size_t k = 0; k < cnt; ++k) {
            if (v[k].dist_score == cropped && early_cut == 0) early_cut = k;
            if (v[k].dist_score < cropped) { late_cut = k; break; }
          }
          if (early_cut > 0) len = early_cut + 1;
          else if (late_cut > 0) len = late_cut + 1;
        }
      }
      if (p.cutoff_threshold >= 1.0) {
        bool have = false;
        double best = 0.0;
        for (size_t k = 0; k < len; ++k) {
          const double s = vr_score(v[k], fw);
          if (have) {
            if (s <= best / p.cutoff_threshold) { len = k; break; }
          } else { best = s; have = true; }
        }
      }
      newlen[i] = len;
    }
  };
  unsigned nthreads = std::max(1u, std::min(64u, anx::usable_hw_threads()));
  if (n < 256) nthreads = 1;
  if (nthreads == 1) work(0, n);
  else {
    std::vector<std::thread> th;
    std::atomic<size_t> next{0};
    const size_t chunk = 256;
    for (unsigned t = 0; t < nthreads; ++t)
      th.emplace_back([&]() {
        for (;;) {
          const size_t lo = next.fetch_add(chunk);
          if (lo >= n) break;
          work(lo, std::min(n, lo + chunk));
        }
      });
    for (auto& x : th) x.join();
  }
  size_t w = 0;  // compact the shortened lists
  for (size_t i = 0; i < n; ++i) {
    const size_t b0 = offs[i];
    offs[i] = w;
    if (w != b0) memmove(rows + w, rows + b0, newlen[i] * sizeof(anx_result));
    w += newlen[i];
  }
  offs[n] = w;
}

extern "C" {

void anx_results_free(anx_result* rows, size_t* offsets);

const char* anx_last_error(void) { return g_err.c_str(); }
int anx_last_error_code(void) { return g_code; }
int anx_abi_version(void) { return ANX_ABI_VERSION; }

void anx_default_weights(anx_weights* w) {  // src/types.rs:57-67
  w->ld = 0.5;
  w->lcs = w->prefix = w->suffix = w->casew = 0.125;
}
void anx_default_params(anx_params* p) {  // src/types.rs:170-192
  p->max_anagram_distance = anx_threshold{ANX_ABSOLUTE, 3, 0.0f};
  p->max_edit_distance = anx_threshold{ANX_ABSOLUTE, 3, 0.0f};
  p->max_matches = 20;
  p->score_threshold = 0.25;
  p->cutoff_threshold = 2.0;
  p->stop_at_exact_match = 0;
  p->freq_weight = 0.0f;
}
void anx_default_vocab_params(anx_vocab_params* p) {  // src/vocab.rs:121-131
  p->text_column = 0;
  p->freq_column = 1;
  p->freq_handling = ANX_FREQ_MAX;
  p->vocab_type = ANX_VOCAB_INDEXED;
}

anx_model* anx_model_new_with_alphabet(const char* tsv, const anx_weights* weights, int debug) {
  if (!tsv) { fail(ANX_EINVAL, "alphabet is NULL"); return nullptr; }
  anx_model* m = new anx_model();
  std::string err;
  if (!anx::parse_alphabet(tsv, m->host.alphabet, err)) {
    fail(ANX_ELIMIT, err);
    delete m;
    return nullptr;
  }
  if (weights) m->host.weights = *weights;
  m->host.debug = debug;
  return m;
}
anx_model* anx_model_new(const char* path, const anx_weights* weights, int debug) {
  if (!path) { fail(ANX_EINVAL, "alphabet path is NULL"); return nullptr; }
  std::ifstream f(path, std::ios::binary);
  if (!f) { fail(ANX_EIO, std::string("Error loading alphabet file ") + path); return nullptr; }
  std::ostringstream ss;
  ss << f.rdbuf();
  return anx_model_new_with_alphabet(ss.str().c_str(), weights, debug);
}
static void drop_replicas(anx_model* m) {
  for (Replica& r : m->replicas) {
    r.worker.reset();  // joins the replica's thread (its queue is empty: every call waits for its shards)
    anx::stream_destroy(r.device, r.stream);
    anx::lexicon_free(r.dev);
  }
  m->replicas.clear();
  m->dev = nullptr;
}
void anx_model_free(anx_model* m) {
  if (!m) return;
  drop_replicas(m);
  delete m;
}
int anx_model_read_vocabulary(anx_model* m, const char* path, const anx_vocab_params* p) {
  if (!m || !path) return fail(ANX_EINVAL, "NULL argument");
  anx_vocab_params vp;
  anx_default_vocab_params(&vp);
  if (p) vp = *p;
  std::string err;
  int rc = m->host.read_vocabulary(path, vp, err);
  return rc ? fail(rc, err) : ANX_OK;
}
uint64_t anx_model_add_to_vocabulary(anx_model* m, const char* utf8, int has_frequency, uint32_t frequency,
                                     const anx_vocab_params* p) {
  if (!m || !utf8) { fail(ANX_EINVAL, "NULL argument"); return UINT64_MAX; }
  anx_vocab_params vp;
  anx_default_vocab_params(&vp);
  if (p) vp = *p;
  return m->host.add_to_vocabulary(utf8, has_frequency != 0, frequency, vp, (uint8_t)m->host.lexicons.size());
}
int anx_model_add_variant(anx_model* m, uint64_t ref_id, const char* variant, double score, int has_frequency,
                          uint32_t frequency, const anx_vocab_params* p) {
  if (!m || !variant) return fail(ANX_EINVAL, "NULL argument");
  anx_vocab_params vp;
  anx_default_vocab_params(&vp);
  if (p) vp = *p;
  int rc = m->host.add_variant(ref_id, variant, score, has_frequency != 0, frequency, vp, (uint8_t)m->host.lexicons.size());
  return rc < 0 ? fail(rc, "invalid reference id") : rc;
}
int anx_model_read_variants(anx_model* m, const char* path, const anx_vocab_params* p, int transparent) {
  if (!m || !path) return fail(ANX_EINVAL, "NULL argument");
  anx_vocab_params vp;
  anx_default_vocab_params(&vp);
  if (p) vp = *p;
  std::string err;
  int rc = m->host.read_variants(path, vp, transparent != 0, err);
  return rc ? fail(rc, err) : ANX_OK;
}
int anx_model_to_devices(anx_model* m, const int* devices, int n) {
  if (!m || (!devices && n > 0)) return fail(ANX_EINVAL, "NULL argument");
  if (n < 1 || n > 64) return fail(ANX_EINVAL, "1..64 replicas");
  if (!m->host.built) return fail(ANX_ENOTBUILT, "Model has not been built yet! Call build() first");
  drop_replicas(m);
  std::string err;
  anx::EncodeTables et;
  anx::build_encode_tables(m->host.alphabet, et);
  // the signature adjacency lists (adjacency.h): built once per call, uploaded to every replica, not kept on the host
  // Default: every replica builds them ON ITS DEVICE from the lexicon tables it has just received (adjacency.hip: milliseconds);
  // ANX_ADJ_BUILD=host: the host threads build them once (adjacency.cpp, seconds for a large lexicon) and every replica gets a copy.
  std::unique_ptr<anx::AdjIndex> adj, stats;
  const bool want_adj = anx::switches().scan_adj && m->host.lex.nsym <= 32;
  const bool on_device = want_adj && !anx::switches().adj_build_host;
  if (want_adj && !on_device) {
    adj.reset(new anx::AdjIndex());
    anx::build_adjacency(m->host.lex, anx::switches().adj_closure, (size_t)anx::switches().adj_budget_mb << 20, anx::usable_hw_threads(), *adj);
  }
  auto publish = [&](const anx::AdjIndex& a, bool have) {
    LengthCost& lc = m->len_cost;
    std::lock_guard<std::mutex> lk(lc.mu);
    for (uint32_t L = 0; L < LengthCost::LMAX; ++L) lc.records[L] = a.len_records[L];
    lc.class_records = a.class_records;
    lc.class_nsig = a.class_nsig;
    lc.have_records = have;
    lc.init = false;  // the corrections were learned against another prior
    if (getenv("ANX_ADJ_TIMING"))
      fprintf(stderr, "[anx adjacency] %s: %u lexicon signatures, %u in the closure, %u lists, %llu records in %llu rows (%.1f MB), %.1f ms\n", on_device ? "device" : "host",
              a.nsig_lexicon, a.nsig_closure, a.nsig_kept, (unsigned long long)a.records, (unsigned long long)a.rows, a.rows * 64.0 * 12.0 / 1e6, a.build_ms);
  };
  if (adj) publish(*adj, !adj->hash.empty());
  for (int i = 0; i < n; ++i) {
    Replica r;
    r.device = devices[i];
    if (on_device && i == 0) stats.reset(new anx::AdjIndex());
    r.dev = anx::lexicon_upload(m->host.lex, et, adj.get(), devices[i], err, on_device ? anx::switches().adj_closure : -1, (size_t)anx::switches().adj_budget_mb << 20,
                                (on_device && i == 0) ? stats.get() : nullptr);
    if (r.dev && on_device && i == 0) publish(*stats, stats->nsig_kept != 0);
    if (r.dev && n > 1 && !(r.stream = anx::stream_create(devices[i], err))) { anx::lexicon_free(r.dev); r.dev = nullptr; }
    if (!r.dev) { drop_replicas(m); return fail(ANX_ENODEVICE, err); }
    if (n > 1) r.worker.reset(new Worker());
    m->replicas.push_back(std::move(r));
  }
  m->dev = m->replicas[0].dev;
  return ANX_OK;
}
int anx_model_to_device(anx_model* m, int device) { return anx_model_to_devices(m, &device, 1); }
int anx_model_num_replicas(const anx_model* m) { return m ? (int)m->replicas.size() : 0; }
int anx_model_replica_device(const anx_model* m, int i) { return (m && i >= 0 && (size_t)i < m->replicas.size()) ? m->replicas[(size_t)i].device : -1; }
int anx_debug_signature(const anx_model* m, const char* utf8, uint64_t* out_sig) {
  if (!m || !utf8 || !out_sig) return fail(ANX_EINVAL, "NULL argument");
  if (!m->host.built) return fail(ANX_ENOTBUILT, "Model has not been built yet!");
  std::vector<uint8_t> norm, cv;
  if (!m->host.encode(utf8, norm, cv)) return fail(ANX_ELIMIT, "more than 255 symbols");
  cv.resize((size_t)m->host.lex.nplanes * 4, 0);
  *out_sig = anx::signature_of(cv.data(), cv.size(), m->host.lex.sym_group);
  return ANX_OK;
}
int anx_debug_entries(const anx_model* m, uint32_t** out_vocab_ids, size_t* n) {
  if (!m || !out_vocab_ids || !n) return fail(ANX_EINVAL, "NULL argument");
  if (!m->host.built) return fail(ANX_ENOTBUILT, "Model has not been built yet!");
  const std::vector<uint32_t>& v = m->host.lex.ent_vocab;
  uint32_t* o = static_cast<uint32_t*>(malloc(std::max<size_t>(v.size(), 1) * sizeof(uint32_t)));
  if (!o) return fail(ANX_ELIMIT, "out of memory");
  if (!v.empty()) memcpy(o, v.data(), v.size() * sizeof(uint32_t));
  *out_vocab_ids = o;
  *n = v.size();
  return ANX_OK;
}
int anx_debug_adjacency(const anx_model* m, int closure, uint64_t budget_bytes, const uint64_t* sigs, size_t n, uint32_t* out_cum, uint32_t** out_ids,
                        uint64_t* out_stats) {
  if (!m || (!sigs && n) || (!out_cum && n) || !out_ids) return fail(ANX_EINVAL, "NULL argument");
  if (!m->host.built) return fail(ANX_ENOTBUILT, "Model has not been built yet!");
  anx::AdjIndex adj;
  anx::build_adjacency(m->host.lex, closure, (size_t)budget_bytes, anx::usable_hw_threads(), adj);
  if (out_stats) { out_stats[0] = adj.nsig_lexicon; out_stats[1] = adj.nsig_closure; out_stats[2] = adj.nsig_kept; out_stats[3] = adj.records; out_stats[4] = adj.rows; out_stats[5] = (uint64_t)adj.build_ms; out_stats[6] = adj.rows_wanted; }
  size_t total = 0;
  std::vector<uint32_t> h1(n);
  for (size_t i = 0; i < n; ++i) {
    h1[i] = adj.find((uint32_t)sigs[i], (uint32_t)(sigs[i] >> 32));
    if (h1[i]) total += (size_t)adj.hdr[h1[i] - 1].cum[anx::kAdjSections - 1] * anx::kAdjRow;
  }
  uint32_t* ids = static_cast<uint32_t*>(malloc(std::max<size_t>(total, 1) * sizeof(uint32_t)));
  if (!ids) return fail(ANX_ELIMIT, "out of memory");
  size_t pos = 0;
  for (size_t i = 0; i < n; ++i) {
    uint32_t* c = out_cum + i * (anx::kAdjSections + 1);
    if (!h1[i]) { for (int s = 0; s <= anx::kAdjSections; ++s) c[s] = 0xFFFFFFFFu; continue; }
    const anx::AdjHdr& h = adj.hdr[h1[i] - 1];
    c[0] = (uint32_t)(pos / anx::kAdjRow);  // first row of this list in *out_ids
    for (int s = 0; s < anx::kAdjSections; ++s) c[s + 1] = h.cum[s];
    const size_t cnt = (size_t)h.cum[anx::kAdjSections - 1] * anx::kAdjRow;
    memcpy(ids + pos, adj.ids + (size_t)h.row0 * anx::kAdjRow, cnt * sizeof(uint32_t));
    pos += cnt;
  }
  *out_ids = ids;
  return ANX_OK;
}
int anx_debug_adjacency_device(const anx_model* m, const uint64_t* sigs, size_t n, uint32_t* out_cum, uint32_t** out_ids) {
  if (!m || (!sigs && n) || (!out_cum && n) || !out_ids) return fail(ANX_EINVAL, "NULL argument");
  if (m->replicas.empty()) return fail(ANX_ENODEVICE, "model is not resident on a device");
  std::string err;
  const int rc = anx::adjacency_debug_lists(m->replicas[0].dev, sigs, n, out_cum, out_ids, err);
  return rc ? fail(rc, err) : ANX_OK;
}
int anx_debug_set_switch(const char* name, const char* value) {
  return anx::set_switch(name, value) ? ANX_OK : fail(ANX_EINVAL, "unknown switch");
}
void anx_debug_kernel_timer(int enable) { anx::kernel_timer_enable(enable != 0); }
int anx_debug_small_stats(uint64_t* out) { if (!out) return fail(ANX_EINVAL, "NULL argument"); anx::small_stats(out); return ANX_OK; }
int anx_debug_kernel_time(const char* name, double* total_ms, uint64_t* launches) {
  return anx::kernel_timer_read(name, total_ms, launches) ? ANX_OK : fail(ANX_EINVAL, "no launch of that kernel was timed");
}
int anx_debug_band_bound(int device, const uint8_t* q_rows, const uint8_t* c_rows, const uint8_t* lq, const uint8_t* lc, size_t n, int d, int form,
                         uint8_t* out_reject) {
  if ((!q_rows || !c_rows || !lq || !lc || !out_reject) && n) return fail(ANX_EINVAL, "NULL argument");
  std::string err;
  const int rc = anx::debug_band_bound(device, q_rows, c_rows, lq, lc, n, d, form, out_reject, err);
  return rc ? fail(rc, err) : ANX_OK;
}
int anx_model_build(anx_model* m, int device) {
  if (!m) return fail(ANX_EINVAL, "NULL model");
  std::string err;
  int rc = m->host.build_index(err);
  if (rc) return fail(rc, err);
  drop_replicas(m);
  if (device < 0) return ANX_OK;
  return anx_model_to_device(m, device);
}
int anx_model_save_index(const anx_model* m, const char* path) {
  if (!m || !path) return fail(ANX_EINVAL, "NULL argument");
  std::string err;
  const int rc = m->host.save_index(path, err);
  return rc ? fail(rc, err) : ANX_OK;
}
int anx_model_set_index_tag(anx_model* m, const char* tag) {
  if (!m || !tag) return fail(ANX_EINVAL, "NULL argument");
  m->host.index_tag = tag;
  return ANX_OK;
}
char* anx_index_read_tag(const char* path) {
  if (!path) { fail(ANX_EINVAL, "NULL argument"); return nullptr; }
  std::string tag, err;
  const int rc = anx::index_read_tag(path, &tag, err);
  if (rc) { fail(rc, err); return nullptr; }
  char* out = static_cast<char*>(malloc(tag.size() + 1));
  if (!out) { fail(ANX_EINVAL, "out of memory"); return nullptr; }
  memcpy(out, tag.c_str(), tag.size() + 1);
  return out;
}
int anx_model_load_index(anx_model* m, const char* path, int device) {
  if (!m || !path) return fail(ANX_EINVAL, "NULL argument");
  std::string err;
  const int rc = m->host.load_index(path, err);
  if (rc) return fail(rc, err);
  drop_replicas(m);
  if (device < 0) return ANX_OK;
  return anx_model_to_device(m, device);
}
uint64_t anx_model_num_lexicons(const anx_model* m) { return m ? m->host.lexicons.size() : 0; }
const char* anx_model_lexicon_name(const anx_model* m, uint64_t i) {
  return (m && i < m->host.lexicons.size()) ? m->host.lexicons[i].c_str() : nullptr;
}
int anx_model_has(const anx_model* m, const char* utf8) { return (m && utf8 && m->host.has(utf8)) ? 1 : 0; }
uint64_t anx_model_vocab_size(const anx_model* m) { return m ? m->host.decoder.size() : 0; }
const char* anx_model_vocab_text(const anx_model* m, uint64_t id) {
  return (m && id < m->host.decoder.size()) ? m->host.decoder[id].text.c_str() : nullptr;
}
uint32_t anx_model_vocab_frequency(const anx_model* m, uint64_t id) {
  return (m && id < m->host.decoder.size()) ? m->host.decoder[id].frequency : 0;
}
uint32_t anx_model_vocab_lexindex(const anx_model* m, uint64_t id) {
  return (m && id < m->host.decoder.size()) ? m->host.decoder[id].lexindex : 0;
}
uint64_t anx_model_num_instances(const anx_model* m) { return m ? m->host.lex.nentries : 0; }
uint64_t anx_model_num_classes(const anx_model* m) { return m ? m->host.lex.nclasses : 0; }
uint64_t anx_model_bucket_size(const anx_model* m, int c) {
  if (!m || !m->host.built || c < 0 || c > anx::kMaxSymbols) return 0;
  return m->host.lex.bucket_begin[c + 1] - m->host.lex.bucket_begin[c];
}
int anx_model_alphabet_size(const anx_model* m) { return m ? m->host.alphabet_size() : 0; }
int anx_model_normalize(const anx_model* m, const char* utf8, uint8_t* out, int cap) {
  if (!m || !utf8 || !out) return fail(ANX_EINVAL, "NULL argument");
  std::vector<uint8_t> norm, cv;
  if (!m->host.encode(utf8, norm, cv)) return fail(ANX_ELIMIT, "input longer than 255 symbols");
  if ((int)norm.size() > cap) return fail(ANX_EINVAL, "buffer too small");
  memcpy(out, norm.data(), norm.size());
  return (int)norm.size();
}
int anx_model_anahash(const anx_model* m, const char* utf8, char* out, int cap) {
  if (!m || !utf8 || !out) return fail(ANX_EINVAL, "NULL argument");
  anx::BigVal v;
  if (!m->host.anahash(utf8, v)) return fail(ANX_ELIMIT, "input longer than 255 symbols");
  const std::string s = v.to_decimal();
  if ((int)s.size() + 1 > cap) return fail(ANX_EINVAL, "buffer too small");
  memcpy(out, s.c_str(), s.size() + 1);
  return (int)s.size();
}

// Rust's `{}` for f64: shortest digits that round-trip, positional notation, "1" for 1.0, NaN / inf / -inf
static void append_rust_f64(std::string& o, double x) {
  if (x != x) { o += "NaN"; return; }
  if (x == __builtin_inf()) { o += "inf"; return; }
  if (x == -__builtin_inf()) { o += "-inf"; return; }
  char buf[400];
  const auto r = std::to_chars(buf, buf + sizeof buf, x, std::chars_format::fixed);
  o.append(buf, r.ptr);
}
static void append_json_escaped(std::string& o, const char* s) {  // only '"' is escaped (src/bin/analiticcl.rs:94)
  for (; *s; ++s) {
    if (*s == '"') o += '\\';
    o += *s;
  }
}
namespace {
struct OutputFormatter {  // shared by the query and search writers
  const anx::HostModel& h;
  double fw;
  bool lexmatch;
  std::string o;
  const char* text_of(uint64_t id) const { return id < h.decoder.size() ? h.decoder[id].text.c_str() : ""; }
  void lexnames(uint64_t id, const char* sep, bool quoted) {
    const uint32_t lexindex = id < h.decoder.size() ? h.decoder[id].lexindex : 0u;
    bool first = true;
    for (size_t i = 0; i < h.lexicons.size() && i < 32; ++i)
      if (lexindex & (1u << i)) {
        if (!first) o += sep;
        first = false;
        if (quoted) { o += '"'; append_json_escaped(o, h.lexicons[i].c_str()); o += '"'; }
        else o += h.lexicons[i];
      }
  }
  double score(const anx_result& v) const { return fw == 0.0 ? v.dist_score : (v.dist_score + fw * v.freq_score) / (1.0 + fw); }
  void tsv_variant(const anx_result& v) {  // output_result_as_tsv, bin:60-76
    o += '\t';
    o += text_of(v.vocab_id);
    o += '\t';
    append_rust_f64(o, score(v));
    o += '\t';
    if (lexmatch) { o += "\t\""; lexnames(v.vocab_id, ";", false); o += '"'; }
  }
  void json_variant(const anx_result& v, bool first) {  // output_result_as_json, bin:150-187
    if (!first) o += ",\n";
    o += "        { \"text\": \"";
    append_json_escaped(o, text_of(v.vocab_id));
    o += "\", \"score\": ";
    append_rust_f64(o, score(v));
    o += ", \"dist_score\": ";
    append_rust_f64(o, v.dist_score);
    o += ", \"freq_score\": ";
    append_rust_f64(o, v.freq_score);
    if (v.via != ANX_NO_VIA) {
      o += ", \"via\": \"";
      append_json_escaped(o, text_of(v.via));
      o += '"';
    }
    if (lexmatch) { o += ", \"lexicons\": [ "; lexnames(v.vocab_id, ", ", true); o += " ]"; }
    o += " }";
  }
  int finish(char** out, size_t* out_len) {
    char* buf = static_cast<char*>(malloc(o.size() + 1));
    if (!buf) return fail(ANX_EINVAL, "out of memory");
    memcpy(buf, o.data(), o.size());
    buf[o.size()] = 0;
    *out = buf;
    *out_len = o.size();
    return ANX_OK;
  }
};
}  // namespace

int anx_format_query_output(const anx_model* m, const char* const* inputs, size_t n, const anx_result* rows,
                            const size_t* offs, float freq_weight, int json, int output_lexmatch,
                            uint64_t first_seqnr, char** out, size_t* out_len) {
  if (!m || (!inputs && n) || !offs || (!rows && n && offs[n]) || !out || !out_len) return fail(ANX_EINVAL, "NULL argument");
  OutputFormatter f{m->host, (double)freq_weight, output_lexmatch != 0, {}};
  f.o.reserve(n * 96);
  for (size_t i = 0; i < n; ++i) {
    const char* inp = inputs[i] ? inputs[i] : "";
    if (!json) {  // output_matches_as_tsv, bin:21-76
      f.o += inp;
      for (size_t r = offs[i]; r < offs[i + 1]; ++r) f.tsv_variant(rows[r]);
      f.o += '\n';
      continue;
    }
    // output_matches_as_json, bin:78-187
    f.o += first_seqnr + i > 1 ? "    ," : "    ";
    f.o += "{ \"input\": \"";
    append_json_escaped(f.o, inp);
    f.o += "\", \"variants\": [ \n";
    for (size_t r = offs[i]; r < offs[i + 1]; ++r) f.json_variant(rows[r], r == offs[i]);
    f.o += "\n    ] }\n";
  }
  return f.finish(out, out_len);
}

int anx_format_search_output(const anx_model* m, const char* const* texts, size_t n, const anx_match* matches,
                             const size_t* offs, const anx_result* rows, const anx_match_tag* tags, float freq_weight,
                             int json, int output_lexmatch, uint64_t first_seqnr, char** out, size_t* out_len) {
  if (!m || (!texts && n) || !offs || (!matches && n && offs[n]) || !out || !out_len) return fail(ANX_EINVAL, "NULL argument");
  OutputFormatter f{m->host, (double)freq_weight, output_lexmatch != 0, {}};
  uint64_t seqnr = first_seqnr;
  for (size_t t = 0; t < n; ++t) {
    const char* text = texts[t] ? texts[t] : "";
    const size_t len = strlen(text);
    for (size_t j = offs[t]; j < offs[t + 1]; ++j, ++seqnr) {
      const anx_match& mt = matches[j];
      if (mt.begin > mt.end || mt.end > len) return fail(ANX_EINVAL, "match offsets outside the text (byte offsets are required)");
      const std::string inp(text + mt.begin, text + mt.end);
      const size_t vb = mt.var_begin, ve = mt.var_end;
      const bool has_sel = mt.selected >= 0 && vb + (size_t)mt.selected < ve;
      if (!json) {  // bin:21-58: input, begin:end, the selected variant first
        f.o += inp;
        f.o += '\t';
        f.o += std::to_string(mt.begin);
        f.o += ':';
        f.o += std::to_string(mt.end);
        if (has_sel) f.tsv_variant(rows[vb + (size_t)mt.selected]);
        for (size_t r = vb; r < ve; ++r)
          if (!has_sel || r != vb + (size_t)mt.selected) f.tsv_variant(rows[r]);
        f.o += '\n';
        continue;
      }
      f.o += seqnr > 1 ? "    ," : "    ";
      f.o += "{ \"input\": \"";
      append_json_escaped(f.o, inp.c_str());
      f.o += "\", \"begin\": " + std::to_string(mt.begin) + ", \"end\": " + std::to_string(mt.end);
      if (tags && mt.tag_end > mt.tag_begin) {  // bin:99-121
        f.o += ", \"tag\": [";
        for (uint32_t k = mt.tag_begin; k < mt.tag_end; ++k) {
          if (k > mt.tag_begin) f.o += ',';
          f.o += '"';
          f.o += tags[k].tag < m->host.tags.size() ? m->host.tags[tags[k].tag] : std::string();
          f.o += '"';
        }
        f.o += "], \"seqnr\": [ ";
        for (uint32_t k = mt.tag_begin; k < mt.tag_end; ++k) {
          if (k > mt.tag_begin) f.o += ',';
          f.o += std::to_string((unsigned)tags[k].seqnr);
        }
        f.o += ']';
      }
      f.o += ", \"variants\": [ \n";
      bool first = true;
      if (has_sel) { f.json_variant(rows[vb + (size_t)mt.selected], true); first = false; }
      for (size_t r = vb; r < ve; ++r)
        if (!has_sel || r != vb + (size_t)mt.selected) { f.json_variant(rows[r], first); first = false; }
      f.o += "\n    ] }\n";
    }
  }
  return f.finish(out, out_len);
}
void anx_string_free(char* s) { free(s); }

int anx_model_add_to_confusables(anx_model* m, const char* editscript, double weight) {
  if (!m || !editscript) return fail(ANX_EINVAL, "NULL argument");
  std::string err;
  const int rc = m->host.add_to_confusables(editscript, weight, err);
  return rc ? fail(rc, err) : ANX_OK;
}
int anx_model_read_confusablelist(anx_model* m, const char* path) {
  if (!m || !path) return fail(ANX_EINVAL, "NULL argument");
  std::string err;
  const int rc = m->host.read_confusablelist(path, err);
  return rc ? fail(rc, err) : ANX_OK;
}
int anx_model_add_contextrule(anx_model* m, const char* pattern, float score, const char* const* tags, size_t n_tags,
                              const char* const* tagoffsets, size_t n_tagoffsets) {
  if (!m || !pattern || (!tags && n_tags) || (!tagoffsets && n_tagoffsets)) return fail(ANX_EINVAL, "NULL argument");
  std::vector<std::string> t, o;
  for (size_t i = 0; i < n_tags; ++i) t.emplace_back(tags[i] ? tags[i] : "");
  for (size_t i = 0; i < n_tagoffsets; ++i) o.emplace_back(tagoffsets[i] ? tagoffsets[i] : "");
  std::string err;
  const int rc = m->host.add_contextrule(pattern, score, t, o, err);
  return rc == ANX_OK ? ANX_OK : fail(rc, err);
}
int anx_model_read_contextrules(anx_model* m, const char* path) {
  if (!m || !path) return fail(ANX_EINVAL, "NULL argument");
  std::string err;
  const int rc = m->host.read_contextrules(path, err);
  return rc == ANX_OK ? ANX_OK : fail(rc, err);
}
size_t anx_model_num_tags(const anx_model* m) { return m ? m->host.tags.size() : 0; }
const char* anx_model_tag_name(const anx_model* m, size_t index) {
  return m && index < m->host.tags.size() ? m->host.tags[index].c_str() : nullptr;
}
void anx_model_set_confusables_before_pruning(anx_model* m) {
  if (m) m->host.confusables_before_pruning = true;
}
int anx_model_confusable_weight(const anx_model* m, const char* input, uint64_t vocab_id, double* out) {
  if (!m || !input || !out) return fail(ANX_EINVAL, "NULL argument");
  if (vocab_id >= m->host.decoder.size()) return fail(ANX_EINVAL, "no such vocabulary item");
  *out = m->host.confusable_weight(input, vocab_id);
  return ANX_OK;
}
int anx_edit_script(const char* source, const char* target, char* out, int cap) {
  if (!source || !target || !out) return fail(ANX_EINVAL, "NULL argument");
  const std::string s = anx::edit_script_string(source, target);
  if ((int)s.size() + 1 > cap) return fail(ANX_EINVAL, "buffer too small");
  memcpy(out, s.c_str(), s.size() + 1);
  return (int)s.size();
}

// ---- shards ------------------------------------------------------------------------------------------------------
// Runs fn(shard index) for every shard of a batch: inline for one shard, else on the replicas' threads, all at once.  Returns the
// first failure (code + message of the lowest failing shard).
namespace {
struct ShardErr { int code = ANX_OK; std::string msg; };
int on_replicas(const anx_model* m, const std::vector<int>& replicas, const std::function<int(size_t, std::string&)>& fn) {
  const size_t ns = replicas.size();
  std::vector<ShardErr> res(ns);
  if (ns == 1 && !m->replicas[(size_t)replicas[0]].worker) {
    res[0].code = fn(0, res[0].msg);
  } else {
    std::mutex mu;
    std::condition_variable cv;
    size_t pending = ns;
    for (size_t i = 0; i < ns; ++i)
      m->replicas[(size_t)replicas[i]].worker->post([&, i]() {
        res[i].code = fn(i, res[i].msg);
        std::lock_guard<std::mutex> g(mu);
        if (--pending == 0) cv.notify_one();
      });
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&]() { return pending == 0; });
  }
  for (size_t i = 0; i < ns; ++i)
    if (res[i].code != ANX_OK) return fail(res[i].code, ns > 1 ? "replica " + std::to_string(replicas[i]) + ": " + res[i].msg : res[i].msg);
  return ANX_OK;
}
int on_shards(const anx_batch* b, const std::function<int(size_t, std::string&)>& fn) {
  std::vector<int> reps(b->shards.size());
  for (size_t i = 0; i < reps.size(); ++i) reps[i] = b->shards[i].replica;
  return on_replicas(b->model, reps, fn);
}
// replicas a call of n inputs is spread over: all of them when every one gets at least ANX_SHARD_MIN inputs
size_t shards_for(const anx_model* m, size_t n) {
  const size_t R = m->replicas.size();
  if (R <= 1) return 1;
  const size_t smin = (size_t)std::max<long>(1, anx::switches().shard_min);
  return std::max<size_t>(1, std::min(R, n / smin));
}
void free_shards(anx_batch* h) {
  for (Shard& s : h->shards) anx::batch_free(s.b);
  h->shards.clear();
}
// the stream a shard runs on: the caller's for a single-replica model, the replica's own otherwise
void* shard_stream(const anx_batch* b, const Shard& s, void* caller) {
  const Replica& r = b->model->replicas[(size_t)s.replica];
  return r.stream ? r.stream : caller;
}
int check_batch(const anx_model* m, const anx_batch* b, void* stream) {
  if (!m || !b || b->model != m) return fail(ANX_EINVAL, "batch does not belong to this model");
  if (stream && m->replicas.size() > 1) return fail(ANX_EINVAL, "a multi-device model runs every replica on its own stream: pass NULL");
  return ANX_OK;
}
// zero bytes of [p, p + len): 8 bytes per step (exact SWAR test)
size_t count_nul(const char* p, size_t len) {
  size_t c = 0, i = 0;
  for (; i + 8 <= len; i += 8) {
    uint64_t v;
    memcpy(&v, p + i, 8);
    const uint64_t t = ~(((v & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | v | 0x7F7F7F7F7F7F7F7Full);  // 0x80 in every zero byte
    c += (size_t)__builtin_popcountll(t);
  }
  for (; i < len; ++i) c += p[i] == 0;
  return c;
}

// body(lo, hi, t) over [0, n) on up to 16 host threads (one for small n)
void parallel_ranges(size_t n, const std::function<void(size_t, size_t, unsigned)>& body, unsigned* used = nullptr) {
  unsigned T = n < (1u << 16) ? 1u : std::max(1u, std::min(16u, anx::usable_hw_threads()));
  if (used) *used = T;
  if (T == 1) { body(0, n, 0); return; }
  std::vector<std::thread> th;
  for (unsigned t = 0; t < T; ++t) th.emplace_back(body, n * t / T, n * (t + 1) / T, t);
  for (auto& x : th) x.join();
}
// offsets of the first n NUL-terminated spans of blob (n + 1 values), found by several threads: every thread takes a byte range that
// starts at a string start, counts its strings, then writes their offsets behind the strings of the ranges before it
bool packed_offsets_mt(const char* blob, size_t len, size_t n, std::vector<uint32_t>& off) {
  const unsigned T = len < (1u << 20) ? 1u : std::max(1u, std::min(16u, anx::usable_hw_threads()));
  if (T == 1) return anx::packed_offsets(blob, len, n, off);
  std::vector<size_t> pos(T + 1, len), cnt(T, 0);
  pos[0] = 0;
  for (unsigned t = 1; t < T; ++t) {
    size_t p = len * t / T;
    if (p > 0 && blob[p - 1] != '\0') {
      const void* z = memchr(blob + p, 0, len - p);
      p = z ? (size_t)(static_cast<const char*>(z) - blob) + 1 : len;
    }
    pos[t] = std::max(p, pos[t - 1]);
  }
  {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t) th.emplace_back([&, t]() { cnt[t] = count_nul(blob + pos[t], pos[t + 1] - pos[t]); });
    for (auto& x : th) x.join();
  }
  std::vector<size_t> first(T + 1, 0);
  for (unsigned t = 0; t < T; ++t) first[t + 1] = first[t] + cnt[t];
  if (first[T] < n) return false;
  off.assign(n + 1, 0u);
  {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t)
      th.emplace_back([&, t]() {
        size_t k = first[t];
        const char* p = blob + pos[t];
        const char* end = blob + pos[t + 1];
        while (p < end && k < n) {
          off[k++] = (uint32_t)(p - blob);
          p = static_cast<const char*>(memchr(p, 0, (size_t)(end - p))) + 1;
        }
        if (k == n && first[t] < n && first[t + 1] >= n) off[n] = (uint32_t)(p - blob);  // the thread that holds the n-th string closes the list
      });
    for (auto& x : th) x.join();
  }
  return true;
}
// Length-partitioned split (BASELINE configs[3]: "10M queries len 4-32 length-bucketed, query-sharded over 8 GPUs").  The device
// groups a batch's queries by (length, signature) into scan tiles of <= 64; consecutive input ranges give every replica 1 / S of
// EVERY group -- 4.9 queries per tile at 1.25 M queries per GPU against the 1 M-entry lexicon, where the whole job on one device
// has 13 -- and a third of each GPU's time goes into per-tile overheads the split itself creates.  The reference fans out over
// independent inputs in arbitrary order (src/bin/analiticcl.rs:416-448), so any partition is legal: here the inputs are ordered
// by CLASS = (byte length, how many of the bytes belong to signature groups 0 and 1) -- a function of (length, signature) for
// strings of one-byte alphabet members, so every (length, signature) group lies inside one class; other strings only land in a
// less fitting share -- and cut into S cost-balanced consecutive pieces of that order: a replica owns whole classes, only a class a
// cut runs through is split (by input order).  A few hundred classes instead of ~30 lengths: the heavy lengths (one length can be
// more than a share) are cut between signature ranges, not through their groups.  Cost of an input = LengthCost of its length.
// Results go back to input order when they are fetched (Shard::idx).
constexpr uint32_t SPLIT_SUBLEN = 64;                      // lengths below this are subdivided by the two group counts (5 bits each)
constexpr uint32_t SPLIT_NCLS = SPLIT_SUBLEN * 1024 + 256;  // ... the others are one class per length (255 = that and longer)
inline uint32_t split_class(uint32_t len, uint32_t g0, uint32_t g1) {
  return len < SPLIT_SUBLEN ? len * 1024u + std::min(g0, 31u) * 32u + std::min(g1, 31u) : SPLIT_SUBLEN * 1024u + std::min(len, 255u);
}
inline uint32_t split_class_len(uint32_t cls) { return cls < SPLIT_SUBLEN * 1024u ? cls >> 10 : cls - SPLIT_SUBLEN * 1024u; }
// byte -> signature group of the alphabet class whose one-byte member it is (0xFF: none)
void split_byte_groups(const anx_model* m, uint8_t (&tab)[256]) {
  memset(tab, 0xFF, sizeof tab);
  const anx::Alphabet& a = m->host.alphabet;
  const std::vector<uint8_t>& sg = m->host.lex.sym_group;
  for (size_t c = 0; c < a.classes.size() && c < sg.size(); ++c)
    for (const anx::AlphabetMember& mem : a.classes[c])
      if (mem.bytes.size() == 1 && tab[(uint8_t)mem.bytes[0]] == 0xFF) tab[(uint8_t)mem.bytes[0]] = sg[c];
}
// the class of every input; get(i) = (pointer, byte length)
void split_classes(const anx_model* m, size_t n, const std::function<void(size_t, const char**, size_t*)>& get, std::vector<uint32_t>& cls) {
  uint8_t tab[256];
  split_byte_groups(m, tab);
  cls.resize(n);
  parallel_ranges(n, [&](size_t lo, size_t hi, unsigned) {
    for (size_t i = lo; i < hi; ++i) {
      const char* s;
      size_t len;
      get(i, &s, &len);
      uint32_t g0 = 0, g1 = 0;
      if (len < SPLIT_SUBLEN)
        for (size_t j = 0; j < len; ++j) { const uint8_t g = tab[(uint8_t)s[j]]; g0 += g == 0; g1 += g == 1; }
      cls[i] = split_class((uint32_t)std::min<size_t>(len, 0xFFFFu), g0, g1);
    }
  });
}
// the prior of the length split (split_by_length): record tests' worth of per-query work that does not depend on the records, and what
// a record test goes on to cost as a function of x = k / length
static inline double split_fixed_cost(uint32_t L) { return 6500.0 * (1.0 + 0.012 * (double)(L > 6u ? L - 6u : 0u)); }
static inline double split_record_factor(double x) { const double x2 = x * x; return 0.2 + 38.0 * x2 * x2; }

void split_by_length(const anx_model* m, const uint32_t* cls, size_t n, const anx_params& p, std::vector<Shard>& shards) {
  const size_t S = shards.size();
  constexpr uint32_t LMAX = LengthCost::LMAX;
  unsigned T = 1;
  std::vector<std::vector<uint32_t>> hist;
  parallel_ranges(n, [&](size_t, size_t, unsigned) {}, &T);
  hist.assign(T, std::vector<uint32_t>(SPLIT_NCLS, 0));
  parallel_ranges(n, [&](size_t lo, size_t hi, unsigned t) {
    std::vector<uint32_t>& h = hist[t];
    for (size_t i = lo; i < hi; ++i) ++h[cls[i]];
  });
  // cost per input of every length.  Prior: a constant + the lexicon's classes within the anagram-distance window (what the scan
  // walks through), times a factor that grows with k / length -- a short string has far more lexicon entries within its distance
  // bounds than a long one (measured on the 1 M-entry lexicon of configs[3]: 31 ns per query of 4 symbols, 15 at 8, 6 at 12-16).
  // The learned correction (LengthCost) takes it from there.
  const anx::LexiconImage& lex = m->host.lex;
  double w[LMAX];
  std::vector<double> wc(SPLIT_NCLS, 0.0);  // per class: the length's weight, refined by the class's own records where the lexicon has entries there
  {
    LengthCost& lc = m->len_cost;
    std::lock_guard<std::mutex> lk(lc.mu);
    auto same = [](const anx_threshold& a, const anx_threshold& b) { return a.kind == b.kind && a.value == b.value && a.ratio == b.ratio; };
    if (!lc.init || !same(lc.k_of, p.max_anagram_distance) || !same(lc.d_of, p.max_edit_distance)) {
      for (double& x : lc.scale) x = 1.0;
      lc.updates = 0;
      lc.k_of = p.max_anagram_distance;
      lc.d_of = p.max_edit_distance;
      lc.init = true;
    }
    for (uint32_t L = 0; L < LMAX; ++L) {
      const int k = anx::clamp_threshold(p.max_anagram_distance, (int)L, anx::kMaxAnagramDistance);
      double window = 0.0;
      for (int c = std::max(1, (int)L - k); c <= std::min(anx::kMaxSymbols, (int)L + k); ++c) window += (double)(lex.bucket_begin[c + 1] - lex.bucket_begin[c]);
      const double rel = L ? (double)k / (double)L : 0.0;
      if (lc.have_records && k <= anx::kAdjRadius) {
        // round 5: the adjacency lists say how many records a query of this length meets (lengths without lexicon entries take
        // the nearest length that has some) plus the per-query work that does not depend on them (encoder, tile
        // set-up, ranking: ~1 ns against 0.3 ps per record test on BASELINE configs[1] / [3]).  First-call balance of the configs[3]
        // job: 0.57 with the window prior, see DESIGN.md section 6
        double r = lc.records[L];
        for (uint32_t o = 1; r == 0.0 && o < LMAX; ++o) {
          if (L >= o && lc.records[L - o] > 0.0) r = lc.records[L - o];
          else if (L + o < LMAX && lc.records[L + o] > 0.0) r = lc.records[L + o];
        }
        // what a record test goes on to cost depends on the share of the records that are hits and survive the band filter and the
        // DL: large for short strings (k / L large: most of a short string's neighbourhood is within reach), small for long ones.
        // Fitted on the 10 M-query configs[3] job (ns per query by length: 21.7 at 6 symbols, 10.5 at 8, 5 at 10, 3 at 12, 2.3 from 16 on:
        // 6500 record tests ~ the per-query work that does not depend on the records -- encoder, tile set-up at 5 queries per tile, ranking)
        // (refitted at the end of round 5 on the corrections the learner converges to on that job -- 1.13 at 6 symbols, 0.78 at 10-12,
        // 1.19 from 13 on: the share of the record tests that go on to cost something falls faster with k / L than a cube, and the
        // per-query work grows with the length: rows, the DL's band, the wide path above 16 symbols)
        const double x = std::min(rel, 0.5);
        w[L] = (split_fixed_cost(L) + r * split_record_factor(x)) * lc.scale[L];
      } else {
        w[L] = (1024.0 + window / 16.0) * (1.0 + 30.0 * rel * rel * rel) * lc.scale[L];
      }
    }
    for (uint32_t c = 0; c < SPLIT_NCLS; ++c) {
      const uint32_t L = std::min(split_class_len(c), LMAX - 1);
      wc[c] = w[L];
      if (lc.have_records && c < lc.class_records.size() && lc.class_records[c] > 0.0f && lc.records[L] > 0.0) {
        const int k = anx::clamp_threshold(p.max_anagram_distance, (int)L, anx::kMaxAnagramDistance);
        if (k <= anx::kAdjRadius) {
          // the class's own records, and how full its scan tiles get in THIS batch: m inputs over the signatures its queries fall on
          // (about three closure signatures per lexicon signature) -- a tile of few queries pays its set-up (rows x 8 query-tests'
          // worth, 48 queries share it in a full tile) and its launch (~3.5 ns of device time, 11 600 record tests' worth) alone
          uint32_t m_c = 0;
          for (unsigned t = 0; t < T; ++t) m_c += hist[t][c];
          const double sigs = 3.0 * std::max(1.0, (double)lc.class_nsig[c]);
          const double fill = std::min(48.0, std::max(1.0, (double)m_c / sigs));
          const double x = std::min(L ? (double)k / (double)L : 0.0, 0.5), f = split_record_factor(x), g = split_fixed_cost(L) / 6500.0;
          const double mine = (4300.0 + 11600.0 / fill) * g + (double)lc.class_records[c] * f * (1.0 + 8.0 / fill) / (1.0 + 8.0 / 48.0);
          wc[c] = w[L] * mine / (split_fixed_cost(L) + lc.records[L] * f);
        }
      }
    }
  }
  if (getenv("ANX_SPLIT_DEBUG")) {
    fprintf(stderr, "[anx split] have_records %d updates %u:", (int)m->len_cost.have_records, m->len_cost.updates);
    for (uint32_t L = 1; L < 36; ++L) fprintf(stderr, " %u:%.0f", L, w[L]);
    fprintf(stderr, "\n");
  }
  // cumulative cost at the first input of every class; rank of every thread's first input inside its class
  std::vector<double> base(SPLIT_NCLS + 1, 0.0);
  std::vector<std::vector<uint32_t>>& start = hist;  // turned into the running ranks in place
  for (uint32_t c = 0; c < SPLIT_NCLS; ++c) {
    uint32_t run = 0;
    for (unsigned t = 0; t < T; ++t) { const uint32_t h = hist[t][c]; start[t][c] = run; run += h; }
    base[c + 1] = base[c] + wc[c] * (double)run;
  }
  const double total = base[SPLIT_NCLS] > 0.0 ? base[SPLIT_NCLS] : 1.0;
  std::vector<uint8_t> gid(n);
  std::vector<std::vector<uint32_t>> per(T, std::vector<uint32_t>(S, 0));
  parallel_ranges(n, [&](size_t lo, size_t hi, unsigned t) {
    std::vector<uint32_t>& st = start[t];
    for (size_t i = lo; i < hi; ++i) {
      const uint32_t c = cls[i];
      const double cum = base[c] + wc[c] * (double)st[c]++;
      const size_t g = std::min(S - 1, (size_t)(cum * (double)S / total));
      gid[i] = (uint8_t)g;
      ++per[t][g];
    }
  });
  std::vector<std::vector<size_t>> at(T, std::vector<size_t>(S, 0));
  for (size_t g = 0; g < S; ++g) {
    size_t run = 0;
    for (unsigned t = 0; t < T; ++t) { at[t][g] = run; run += per[t][g]; }
    shards[g].idx.assign(run, 0u);
    shards[g].n = run;
    shards[g].lo = 0;
  }
  parallel_ranges(n, [&](size_t lo, size_t hi, unsigned t) {
    std::vector<size_t>& a = at[t];
    for (size_t i = lo; i < hi; ++i) shards[gid[i]].idx[a[gid[i]]++] = (uint32_t)i;  // a thread's inputs are consecutive: ascending per shard
  });
  for (Shard& sh : shards) {
    sh.lhist.assign(LMAX, 0u);
    for (uint32_t i : sh.idx) ++sh.lhist[std::min<uint32_t>(split_class_len(cls[i]), LMAX - 1)];
    sh.predicted = 0.0;
    for (uint32_t i : sh.idx) sh.predicted += wc[cls[i]];
  }
  shards.erase(std::remove_if(shards.begin(), shards.end(), [](const Shard& s) { return s.n == 0; }), shards.end());
}
// after a run: the shards' device times against what the split predicted -> the per-length corrections of the next split
void learn_length_costs(const anx_model* m, const std::vector<Shard>& shards, const std::vector<double>& t) {
  const size_t S = shards.size();
  if (S < 2) return;
  double tsum = 0.0, psum = 0.0;
  for (size_t g = 0; g < S; ++g) {
    if (shards[g].lhist.empty() || shards[g].predicted <= 0.0 || !(t[g] > 0.02)) return;  // not a length split / too short to say anything
    tsum += t[g];
    psum += shards[g].predicted;
  }
  LengthCost& lc = m->len_cost;
  std::lock_guard<std::mutex> lk(lc.mu);
  for (uint32_t L = 0; L < LengthCost::LMAX; ++L) {
    double num = 0.0, den = 0.0;
    for (size_t g = 0; g < S; ++g) {
      const double c = (double)shards[g].lhist[L];
      num += c * (t[g] / shards[g].predicted) / (tsum / psum);
      den += c;
    }
    // the first two updates take the measured ratio as it is (the prior may be far off for this lexicon / these parameters); later
    // ones are damped (square root): a noisy run moves the weights half way, and the next split is measured afresh
    if (den > 0.0) lc.scale[L] = std::min(64.0, std::max(1.0 / 64.0, lc.scale[L] * (lc.updates < 2 ? num / den : std::sqrt(num / den))));
  }
  ++lc.updates;
}
void learn_from_batch(const anx_batch* b) {
  // once per split: Shard::predicted is what the split assumed when it was made -- a batch that is run again and again (a resident
  // benchmark batch) would otherwise apply the same ratio every time and drive the weights to their clamps
  if (b->learned) return;
  b->learned = true;
  std::vector<double> t(b->shards.size(), 0.0);
  for (size_t g = 0; g < b->shards.size(); ++g) {
    anx_batch_stats st;
    anx::batch_stats(b->shards[g].b, &st);
    t[g] = (double)st.ms_total;
  }
  learn_length_costs(b->model, b->shards, t);
}
bool use_length_split(const anx_model* m, size_t S, bool rescore) {
  return S > 1 && S <= 255 && !rescore && anx::switches().shard_by_length && m->host.built;
}
}  // namespace

// device parameters of a batch: with confusables the cutoff (and in early mode the crop) follows the host-side rescoring
static anx_params device_params(const anx_model* m, const anx_params* p, bool* rescore) {
  anx_params dp = *p;
  *rescore = !m->host.confusables.empty();
  if (*rescore) {
    dp.cutoff_threshold = 0.0;                               // the cutoff follows the late rescoring (src/lib.rs:1591-1622)
    if (m->host.confusables_before_pruning) dp.max_matches = 0;  // early: crop after rescoring as well (src/lib.rs:1505-1589)
  }
  return dp;
}
static int check_encode_args(const anx_model* m, const void* inputs, size_t n, const anx_params* p) {
  if (!m || (!inputs && n) || !p) return fail(ANX_EINVAL, "NULL argument");
  if (!m->host.built) return fail(ANX_ENOTBUILT, "Model has not been built yet! Call build() before find_variants()");
  return ANX_OK;
}
static int check_resident(const anx_model* m) {  // there is no CPU fallback
  if (m->replicas.empty()) return fail(ANX_ENODEVICE, "model is not resident on a device (no HIP device / anx_model_to_device not called)");
  return ANX_OK;
}
int anx_debug_length_split(const anx_model* m, const char* const* utf8, size_t n, const anx_params* p, int n_shards, uint8_t* out_shard,
                           const double* learn_ms) {
  if (!m || (!utf8 && n) || !p || (!out_shard && n) || n_shards < 1 || n_shards > 255) return fail(ANX_EINVAL, "bad argument");
  if (!m->host.built) return fail(ANX_ENOTBUILT, "Model has not been built yet!");
  std::vector<Shard> shards((size_t)n_shards);
  for (int g = 0; g < n_shards; ++g) shards[(size_t)g].replica = g;
  std::vector<uint32_t> cls;
  split_classes(m, n, [&](size_t i, const char** sp, size_t* len) { *sp = utf8[i] ? utf8[i] : ""; *len = strlen(*sp); }, cls);
  split_by_length(m, cls.data(), n, *p, shards);
  for (const Shard& s : shards)
    for (uint32_t i : s.idx) out_shard[i] = (uint8_t)s.replica;
  if (learn_ms && shards.size() == (size_t)n_shards) learn_length_costs(m, shards, std::vector<double>(learn_ms, learn_ms + n_shards));
  return ANX_OK;
}
anx_batch* anx_batch_encode(const anx_model* m, const char* const* utf8, size_t n, const anx_params* p) {
  if (check_encode_args(m, utf8, n, p) || check_resident(m)) return nullptr;
  bool rescore;
  anx_params dp = device_params(m, p, &rescore);
  const bool dev_conf = rescore && !anx::switches().confusables_host;
  if (dev_conf) { dp = *p; rescore = false; }
  anx_batch* h = new anx_batch();
  h->model = m;
  h->n_input = n;
  h->rescore = rescore;
  h->dev_conf = dev_conf;
  h->params = *p;
  const size_t S = shards_for(m, n);
  h->shards.resize(S);
  for (size_t g = 0; g < S; ++g) { h->shards[g].replica = (int)g; h->shards[g].lo = n * g / S; h->shards[g].n = n * (g + 1) / S - n * g / S; }
  if (use_length_split(m, S, rescore)) {
    std::vector<uint32_t> cls;
    split_classes(m, n, [&](size_t i, const char** sp, size_t* len) { *sp = utf8[i] ? utf8[i] : ""; *len = strlen(*sp); }, cls);
    split_by_length(m, cls.data(), n, *p, h->shards);
  }
  const int rc = on_shards(h, [&](size_t g, std::string& err) {
    Shard& s = h->shards[g];
    int code = ANX_OK;
    if (s.idx.empty()) {
      s.b = anx::batch_encode(m->host, m->replicas[(size_t)s.replica].dev, utf8 + s.lo, s.n, dp, err, &code, dev_conf);
    } else {
      std::vector<const char*> ptrs(s.n);
      for (size_t i = 0; i < s.n; ++i) ptrs[i] = utf8[s.idx[i]];
      s.b = anx::batch_encode(m->host, m->replicas[(size_t)s.replica].dev, ptrs.data(), s.n, dp, err, &code, dev_conf);
    }
    if (s.b && dev_conf) anx::batch_set_run_mode(s.b, dp, m->host.confusables_before_pruning ? 2 : 1);
    return s.b ? ANX_OK : (code ? code : ANX_ENODEVICE);
  });
  if (rc) { free_shards(h); delete h; return nullptr; }
  if (rescore) {
    h->in_off.reserve(n + 1);
    h->in_off.push_back(0);
    for (size_t i = 0; i < n; ++i) {
      if (utf8[i]) h->in_text.append(utf8[i]);
      h->in_text.push_back('\0');
      h->in_off.push_back((uint32_t)h->in_text.size());
    }
  }
  return h;
}
static anx_batch* encode_packed_device(const anx_model* m, const void* device_blob, size_t blob_len, size_t n, const anx_params* p, bool ordered, void* stream);
anx_batch* anx_batch_encode_packed_device(const anx_model* m, const void* device_blob, size_t blob_len, size_t n, const anx_params* p) {
  return encode_packed_device(m, device_blob, blob_len, n, p, false, nullptr);
}
anx_batch* anx_batch_encode_packed_device_on(const anx_model* m, const void* device_blob, size_t blob_len, size_t n, const anx_params* p, void* stream) {
  return encode_packed_device(m, device_blob, blob_len, n, p, true, stream);
}
static anx_batch* encode_packed_device(const anx_model* m, const void* device_blob, size_t blob_len, size_t n, const anx_params* p, bool ordered, void* stream) {
  if (!m || (!device_blob && n) || !p) { fail(ANX_EINVAL, "NULL argument"); return nullptr; }
  if (!m->host.built) { fail(ANX_ENOTBUILT, "Model has not been built yet! Call build() first"); return nullptr; }
  if (blob_len >= ((size_t)1 << 32)) { fail(ANX_ELIMIT, "inputs exceed 4 GB per batch: split the batch"); return nullptr; }
  if (check_resident(m)) return nullptr;
  if (m->replicas.size() != 1) { fail(ANX_EINVAL, "inputs in device memory: the model must be on exactly one device (the one that holds them)"); return nullptr; }
  bool rescore;
  anx_params dp = device_params(m, p, &rescore);
  const bool dev_conf = rescore && !anx::switches().confusables_host;
  if (dev_conf) { dp = *p; rescore = false; }
  if (rescore) { fail(ANX_EINVAL, "inputs in device memory cannot be rescored on the host (ANX_CONFUSABLES=host)"); return nullptr; }
  anx_batch* h = new anx_batch();
  h->model = m;
  h->n_input = n;
  h->dev_conf = dev_conf;
  h->params = *p;
  h->shards.resize(1);
  Shard& s = h->shards[0];
  s.replica = 0; s.lo = 0; s.n = n;
  std::string err;
  int code = ANX_OK;
  s.b = anx::batch_encode_spans(m->host, m->replicas[0].dev, static_cast<const char*>(device_blob), blob_len, nullptr, n, dp, err, &code, dev_conf, true, ordered, stream);
  if (!s.b) { fail(code ? code : ANX_ENODEVICE, err); delete h; return nullptr; }
  if (dev_conf) anx::batch_set_run_mode(s.b, dp, m->host.confusables_before_pruning ? 2 : 1);
  return h;
}
anx_batch* anx_batch_encode_packed(const anx_model* m, const char* blob, size_t blob_len, size_t n, const anx_params* p) {
  if (check_encode_args(m, blob, n, p)) return nullptr;
  if (n && (blob_len == 0 || blob[blob_len - 1] != '\0')) { fail(ANX_EINVAL, "packed inputs must end with a NUL byte"); return nullptr; }
  if (blob_len >= ((size_t)1 << 32)) { fail(ANX_ELIMIT, "inputs exceed 4 GB per batch: split the batch"); return nullptr; }
  // the buffer goes to the device as it is; the device-side encoder finds the strings' offsets there (the host only needs
  // them when confusables are loaded: rescoring reads the input strings)
  bool rescore;
  anx_params dp = device_params(m, p, &rescore);
  const bool dev_conf = rescore && !anx::switches().confusables_host;
  if (dev_conf) { dp = *p; rescore = false; }  // weighted on the device: the host needs neither the offsets nor a copy of the inputs
  std::vector<uint32_t> off;
  if (rescore && !anx::packed_offsets(blob, blob_len, n, off)) { fail(ANX_EINVAL, "packed inputs hold fewer strings than announced"); return nullptr; }
  if (check_resident(m)) return nullptr;
  anx_batch* h = new anx_batch();
  h->model = m;
  h->n_input = n;
  h->rescore = rescore;
  h->dev_conf = dev_conf;
  h->params = *p;
  const size_t S = shards_for(m, n);
  h->shards.resize(S);
  for (size_t g = 0; g < S; ++g) h->shards[g].replica = (int)g;
  if (use_length_split(m, S, rescore)) {
    // length-partitioned split: the host finds the strings (threaded), orders them by length and hands every replica its own
    // list; each replica's thread packs and uploads its strings (the char** path).  Costs the host one pass over the buffer that
    // the byte-balanced split below does not need -- and buys full scan tiles on every device.
    std::vector<uint32_t> poff;
    if (!packed_offsets_mt(blob, blob_len, n, poff)) { fail(ANX_EINVAL, "packed inputs hold fewer strings than announced"); delete h; return nullptr; }
    std::vector<uint32_t> cls;
    split_classes(m, n, [&](size_t i, const char** sp, size_t* len) { *sp = blob + poff[i]; *len = poff[i + 1] - poff[i] - 1u; }, cls);
    split_by_length(m, cls.data(), n, *p, h->shards);
    const int rcl = on_shards(h, [&](size_t g, std::string& err) {
      Shard& s = h->shards[g];
      int code = ANX_OK;
      std::vector<const char*> ptrs(s.n);
      for (size_t i = 0; i < s.n; ++i) ptrs[i] = blob + poff[s.idx[i]];
      s.b = anx::batch_encode(m->host, m->replicas[(size_t)s.replica].dev, ptrs.data(), s.n, dp, err, &code, dev_conf);
      if (s.b && dev_conf) anx::batch_set_run_mode(s.b, dp, m->host.confusables_before_pruning ? 2 : 1);
      return s.b ? ANX_OK : (code ? code : ANX_ENODEVICE);
    });
    if (rcl) { free_shards(h); delete h; return nullptr; }
    return h;
  }
  // Byte-balanced split: shard g takes the strings that START in bytes [pos[g], pos[g + 1]), where pos[g] is the first string
  // start at or behind blob_len * g / S.  Each replica's thread counts the strings of its own slice (phase 1); the prefix sums give
  // every shard its first input index, and the call's n cuts the tail ("the first n NUL-terminated spans").
  std::vector<size_t> pos(S + 1, blob_len), cnt(S, 0);
  pos[0] = 0;
  if (S > 1) {
    for (size_t g = 1; g < S; ++g) {
      size_t t = blob_len * g / S;
      if (t > 0 && blob[t - 1] != '\0') {
        const void* z = memchr(blob + t, 0, blob_len - t);
        t = z ? (size_t)(static_cast<const char*>(z) - blob) + 1 : blob_len;
      }
      pos[g] = std::max(t, pos[g - 1]);
    }
    int rc1 = on_shards(h, [&](size_t g, std::string&) { cnt[g] = count_nul(blob + pos[g], pos[g + 1] - pos[g]); return ANX_OK; });
    (void)rc1;
    size_t run = 0;
    for (size_t g = 0; g < S; ++g) {
      h->shards[g].lo = std::min(run, n);
      h->shards[g].n = std::min(cnt[g], n - h->shards[g].lo);
      run += cnt[g];
    }
    if (run < n) { fail(ANX_EINVAL, "packed inputs hold fewer strings than announced"); delete h; return nullptr; }
  } else {
    h->shards[0].lo = 0;
    h->shards[0].n = n;
  }
  const int rc = on_shards(h, [&](size_t g, std::string& err) {
    Shard& s = h->shards[g];
    int code = ANX_OK;
    const uint32_t* o = rescore ? off.data() + s.lo : nullptr;  // offsets relative to the whole buffer: rebased below
    std::vector<uint32_t> rel;
    const char* base = blob + pos[g];
    size_t bytes = pos[g + 1] - pos[g];
    if (o) {
      rel.resize(s.n + 1);
      for (size_t i = 0; i <= s.n; ++i) rel[i] = o[i] - o[0];
      base = blob + o[0];
      bytes = rel[s.n];
      o = rel.data();
    }
    s.b = anx::batch_encode_spans(m->host, m->replicas[(size_t)s.replica].dev, base, bytes, o, s.n, dp, err, &code, dev_conf);
    if (s.b && dev_conf) anx::batch_set_run_mode(s.b, dp, m->host.confusables_before_pruning ? 2 : 1);
    return s.b ? ANX_OK : (code ? code : ANX_ENODEVICE);
  });
  if (rc) { free_shards(h); delete h; return nullptr; }
  if (rescore) {
    h->in_text.assign(blob, off[n]);  // the first n strings with their NUL bytes
    h->in_off = std::move(off);
  }
  return h;
}
// anx_pipeline's launching thread: its runs go to the pipeline's own two normal-priority streams (alternating per job), not to the
// library's pair of run streams (anx_pipeline_new)
static thread_local bool t_runs_on_caller_stream = false;
int anx_batch_run_async(const anx_model* m, anx_batch* b, void* stream) {
  if (int rc = check_batch(m, b, stream)) return rc;
  return on_shards(b, [&](size_t g, std::string& err) {
    const Shard& s = b->shards[g];
    return anx::batch_run_async(m->host, m->replicas[(size_t)s.replica].dev, s.b, shard_stream(b, s, stream), m->replicas.size() == 1 && !t_runs_on_caller_stream, err);
  });
}
// A device-weighted run met a row the device could not weight (a string beyond the fixed working memory of conf.hip): the whole
// batch takes the host path instead -- the inputs come back from the device, every shard is re-run with the host-mode parameters
// (no cutoff; early mode: no crop) and anx_batch_fetch rescoring takes over.  Rare by construction (strings of > 64 code points).
static int conf_fallback_to_host(const anx_model* m, anx_batch* b, void* stream) {
  bool any = false;
  for (const Shard& s : b->shards) any = any || anx::batch_conf_fallback(s.b);
  if (!b->dev_conf || !any) return ANX_OK;
  bool rescore;
  const anx_params dp = device_params(m, &b->params, &rescore);
  std::vector<std::string> text(b->shards.size());
  std::vector<std::vector<uint32_t>> off(b->shards.size());
  const int rc = on_shards(b, [&](size_t g, std::string& err) {
    const Shard& s = b->shards[g];
    if (int r = anx::batch_download_text(s.b, text[g], off[g], err)) return r;
    anx::batch_set_run_mode(s.b, dp, 0);
    return anx::batch_run(m->host, m->replicas[(size_t)s.replica].dev, s.b, shard_stream(b, s, stream), err);
  });
  if (rc) return rc;
  // the inputs in the call's order (a length-partitioned shard holds a scattered subset)
  b->in_off.assign(b->n_input + 1, 0u);
  for (size_t g = 0; g < b->shards.size(); ++g) {
    const Shard& s = b->shards[g];
    for (size_t i = 0; i + 1 < off[g].size(); ++i) b->in_off[s.input(i) + 1] = off[g][i + 1] - off[g][i];
  }
  for (size_t i = 0; i < b->n_input; ++i) b->in_off[i + 1] += b->in_off[i];
  b->in_text.assign(b->in_off[b->n_input], '\0');
  for (size_t g = 0; g < b->shards.size(); ++g) {
    const Shard& s = b->shards[g];
    for (size_t i = 0; i + 1 < off[g].size(); ++i)
      memcpy(&b->in_text[b->in_off[s.input(i)]], text[g].data() + off[g][i], off[g][i + 1] - off[g][i]);
  }
  b->dev_conf = false;
  b->rescore = true;
  return ANX_OK;
}
int anx_batch_wait(const anx_model* m, anx_batch* b) {
  if (int rc = check_batch(m, b, nullptr)) return rc;
  const int rc = on_shards(b, [&](size_t g, std::string& err) {
    const Shard& s = b->shards[g];
    return anx::batch_wait(m->host, m->replicas[(size_t)s.replica].dev, s.b, err);
  });
  if (!rc) learn_from_batch(b);
  return rc ? rc : conf_fallback_to_host(m, b, nullptr);
}
int anx_batch_run(const anx_model* m, anx_batch* b, void* stream) {
  if (int rc = check_batch(m, b, stream)) return rc;
  const int rc = on_shards(b, [&](size_t g, std::string& err) {
    const Shard& s = b->shards[g];
    return anx::batch_run(m->host, m->replicas[(size_t)s.replica].dev, s.b, shard_stream(b, s, stream), err);
  });
  if (!rc) learn_from_batch(b);
  return rc ? rc : conf_fallback_to_host(m, b, stream);
}
}  // extern "C"
// Rows of a batch whose shards hold scattered inputs (length-partitioned split), back in the call's input order: every shard
// downloads into a buffer of its own (pinned cache), the per-input counts give the global offsets, and the shards' threads copy
// their rows to where they belong.  Row = anx_result / anx_topk_record, Off = size_t / uint32_t.
template <typename Row, typename Off, typename FetchFn>
static int scatter_fetch(const anx_batch* b, Row* out, Off* off, const FetchFn& fetch_into) {
  const size_t n = b->n_input, S = b->shards.size();
  std::vector<Row*> rows(S, nullptr);
  std::vector<std::vector<Off>> loff(S);
  auto release = [&]() { for (Row* r : rows) anx::host_result_free(r); };
  int rc = on_shards(b, [&](size_t g, std::string& err) {
    const Shard& s = b->shards[g];
    rows[g] = static_cast<Row*>(anx::host_result_alloc(std::max<size_t>(1, anx::batch_n_results(s.b)) * sizeof(Row)));
    if (!rows[g]) { err = "out of memory"; return (int)ANX_EINVAL; }
    loff[g].assign(s.n + 1, 0);
    return fetch_into(s.b, rows[g], loff[g].data(), err);
  });
  if (rc) { release(); return rc; }
  for (size_t i = 0; i <= n; ++i) off[i] = 0;
  (void)on_shards(b, [&](size_t g, std::string&) {
    const Shard& s = b->shards[g];
    for (size_t i = 0; i < s.n; ++i) off[s.input(i) + 1] = loff[g][i + 1] - loff[g][i];
    return (int)ANX_OK;
  });
  for (size_t i = 0; i < n; ++i) off[i + 1] += off[i];
  (void)on_shards(b, [&](size_t g, std::string&) {
    const Shard& s = b->shards[g];
    for (size_t i = 0; i < s.n; ++i) {
      const size_t c = (size_t)(loff[g][i + 1] - loff[g][i]);
      if (c) memcpy(out + off[s.input(i)], rows[g] + loff[g][i], c * sizeof(Row));
    }
    return (int)ANX_OK;
  });
  release();
  return ANX_OK;
}
static bool scattered(const anx_batch* b) {
  for (const Shard& s : b->shards) if (!s.idx.empty()) return true;
  return false;
}
extern "C" {
int anx_batch_fetch(const anx_batch* b, anx_result** rows, size_t** offs) {
  if (!b || !rows || !offs) return fail(ANX_EINVAL, "NULL argument");
  // one row array for the whole call; every shard downloads straight into its slice (the sizes are known since the run)
  const size_t n = b->n_input, S = b->shards.size();
  std::vector<size_t> base(S + 1, 0);
  for (size_t g = 0; g < S; ++g) base[g + 1] = base[g] + anx::batch_n_results(b->shards[g].b);
  size_t* off = static_cast<size_t*>(malloc((n + 1) * sizeof(size_t)));
  anx_result* out = static_cast<anx_result*>(anx::host_result_alloc(std::max<size_t>(1, base[S]) * sizeof(anx_result)));
  if (!off || !out) { free(off); anx::host_result_free(out); return fail(ANX_EINVAL, "out of memory"); }
  if (scattered(b)) {
    const int rcs = scatter_fetch<anx_result, size_t>(b, out, off, [](const anx::Batch* sb, anx_result* r, size_t* o, std::string& err) {
      return anx::batch_fetch_into(sb, r, o, 0, err);
    });
    if (rcs) { free(off); anx::host_result_free(out); return rcs; }
    if (b->rescore) rescore_with_confusables(b->model->host, b->in_text, b->in_off, b->params, out, off);
    *rows = out;
    *offs = off;
    return ANX_OK;
  }
  const int rc = on_shards(b, [&](size_t g, std::string& err) {
    const Shard& s = b->shards[g];
    std::vector<size_t> tmp;  // a shard writes n + 1 offsets; its last one is the next shard's first (same value): keep the slices disjoint
    size_t* dst = off + s.lo;
    if (g + 1 < S) { tmp.resize(s.n + 1); dst = tmp.data(); }
    const int r = anx::batch_fetch_into(s.b, out + base[g], dst, base[g], err);
    if (r == ANX_OK && g + 1 < S && s.n) memcpy(off + s.lo, tmp.data(), s.n * sizeof(size_t));
    return r;
  });
  if (rc) { free(off); anx::host_result_free(out); return rc; }
  off[n] = base[S];
  if (b->rescore) rescore_with_confusables(b->model->host, b->in_text, b->in_off, b->params, out, off);
  *rows = out;
  *offs = off;
  return ANX_OK;
}
int anx_batch_fetch_compact(const anx_batch* b, anx_topk_record** rows, uint32_t** offs) {
  if (!b || !rows || !offs) return fail(ANX_EINVAL, "NULL argument");
  if (b->rescore) return fail(ANX_EINVAL, "confusables are loaded: results are rescored on the host, use anx_batch_fetch");
  if (b->model->host.lex.any_variants) return fail(ANX_EINVAL, "variant lists are loaded: compact records carry no `via`, use anx_batch_fetch");
  const size_t n = b->n_input, S = b->shards.size();
  std::vector<size_t> base(S + 1, 0);
  for (size_t g = 0; g < S; ++g) base[g + 1] = base[g] + anx::batch_n_results(b->shards[g].b);
  if (base[S] >= ((size_t)1 << 32)) return fail(ANX_ELIMIT, "more than 2^32 result rows: use anx_batch_fetch");
  // offsets (pinned as well: they are a D2H target) and rows in ONE pinned block of the result cache: [rows | offsets]
  const size_t row_bytes = (std::max<size_t>(1, base[S]) * sizeof(anx_topk_record) + 63) & ~(size_t)63;
  char* blk = static_cast<char*>(anx::host_result_alloc(row_bytes + (n + 2) * sizeof(uint32_t)));
  if (!blk) return fail(ANX_EINVAL, "out of memory");
  anx_topk_record* out = reinterpret_cast<anx_topk_record*>(blk);
  uint32_t* off = reinterpret_cast<uint32_t*>(blk + row_bytes);
  if (scattered(b)) {
    const int rcs = scatter_fetch<anx_topk_record, uint32_t>(b, out, off, [](const anx::Batch* sb, anx_topk_record* r, uint32_t* o, std::string& err) {
      return anx::batch_fetch_compact_into(sb, r, o, 0u, err);
    });
    if (rcs) { anx::host_result_free(blk); return rcs; }
    *rows = out;
    *offs = off;
    return ANX_OK;
  }
  const int rc = on_shards(b, [&](size_t g, std::string& err) {
    const Shard& s = b->shards[g];
    std::vector<uint32_t> tmp;  // the last offset of a shard is the first of the next: keep the slices disjoint
    uint32_t* dst = off + s.lo;
    if (g + 1 < S) { tmp.resize(s.n + 1); dst = tmp.data(); }
    const int r = anx::batch_fetch_compact_into(s.b, out + base[g], dst, (uint32_t)base[g], err);
    if (r == ANX_OK && g + 1 < S && s.n) memcpy(off + s.lo, tmp.data(), s.n * sizeof(uint32_t));
    return r;
  });
  if (rc) { anx::host_result_free(blk); return rc; }
  off[n] = (uint32_t)base[S];
  *rows = out;
  *offs = off;
  return ANX_OK;
}
void anx_compact_free(anx_topk_record* rows, uint32_t* offsets) {
  (void)offsets;  // one block: the offsets live behind the rows
  anx::host_result_free(rows);
}
void anx_compact_to_results(const anx_topk_record* rows, size_t n_rows, anx_result* out) {
  if (!rows || !out) return;
  auto work = [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) out[i] = anx_result{rows[i].vocab_id, rows[i].dist_score, (double)rows[i].freq_score, ANX_NO_VIA};
  };
  unsigned nthreads = n_rows < (1u << 16) ? 1u : std::max(1u, std::min(16u, anx::usable_hw_threads()));
  if (nthreads == 1) { work(0, n_rows); return; }
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nthreads; ++t) th.emplace_back(work, n_rows * t / nthreads, n_rows * (t + 1) / nthreads);
  for (auto& x : th) x.join();
}
int anx_batch_fetch_pairs(const anx_batch* b, anx_pair** out, size_t* n) {
  if (!b || !out || !n) return fail(ANX_EINVAL, "NULL argument");
  const size_t S = b->shards.size();
  std::vector<anx_pair*> part(S, nullptr);
  std::vector<size_t> cnt(S, 0);
  const int rc = on_shards(b, [&](size_t g, std::string& err) {
    const Shard& s = b->shards[g];
    return anx::batch_fetch_pairs(b->model->host, b->model->replicas[(size_t)s.replica].dev, s.b, &part[g], &cnt[g], err);
  });
  if (rc) { for (anx_pair* p : part) free(p); return rc; }
  if (S == 1 && !scattered(b)) { *out = part[0]; *n = cnt[0]; return ANX_OK; }
  size_t total = 0;
  for (size_t c : cnt) total += c;
  anx_pair* all = static_cast<anx_pair*>(malloc(std::max<size_t>(1, total) * sizeof(anx_pair)));
  if (!all) { for (anx_pair* p : part) free(p); return fail(ANX_EINVAL, "out of memory"); }
  size_t w = 0;
  for (size_t g = 0; g < S; ++g) {
    for (size_t i = 0; i < cnt[g]; ++i) { all[w] = part[g][i]; all[w].query = (uint32_t)b->shards[g].input(part[g][i].query); ++w; }
    free(part[g]);
  }
  *out = all;
  *n = total;
  return ANX_OK;
}
void anx_pairs_free(anx_pair* p) { free(p); }
int anx_batch_pair_counts(anx_batch* b, uint32_t** out) {
  if (!b || !out) return fail(ANX_EINVAL, "NULL argument");
  const size_t S = b->shards.size();
  std::vector<uint32_t*> part(S, nullptr);
  const int rc = on_shards(b, [&](size_t g, std::string& err) {
    const Shard& s = b->shards[g];
    return anx::batch_pair_counts(b->model->host, b->model->replicas[(size_t)s.replica].dev, s.b, &part[g], err);
  });
  if (rc) { for (uint32_t* p : part) free(p); return rc; }
  if (S == 1 && !scattered(b)) { *out = part[0]; return ANX_OK; }
  uint32_t* all = static_cast<uint32_t*>(calloc(std::max<size_t>(1, b->n_input), sizeof(uint32_t)));
  if (!all) { for (uint32_t* p : part) free(p); return fail(ANX_EINVAL, "out of memory"); }
  for (size_t g = 0; g < S; ++g) {
    const Shard& sh = b->shards[g];
    if (sh.idx.empty()) { if (sh.n) memcpy(all + sh.lo, part[g], sh.n * sizeof(uint32_t)); }
    else for (size_t i = 0; i < sh.n; ++i) all[sh.idx[i]] = part[g][i];
    free(part[g]);
  }
  *out = all;
  return ANX_OK;
}
void anx_counts_free(uint32_t* p) { free(p); }
static int check_export(const anx_batch* b) {
  if (!b) return fail(ANX_EINVAL, "NULL batch");
  if (b->rescore) return fail(ANX_EINVAL, "confusables are loaded: results are rescored on the host, use anx_batch_fetch");
  if (b->shards.size() != 1) return fail(ANX_EINVAL, "the batch is spread over several replicas: export one shard at a time (anx_batch_shard_*)");
  return ANX_OK;
}
int anx_batch_export_topk(const anx_batch* b, void* dst, uint32_t stride, void* stream) {
  if (int rc = check_export(b)) return rc;
  std::string err;
  int rc = anx::batch_export_topk(b->model->replicas[(size_t)b->shards[0].replica].dev, b->shards[0].b, dst, stride, stream, err);
  return rc ? fail(rc, err) : ANX_OK;
}
int anx_batch_export_compact(const anx_batch* b, void* dst, size_t capacity, void* stream, size_t* used) {
  if (!used) return fail(ANX_EINVAL, "NULL argument");
  if (int rc = check_export(b)) return rc;
  std::string err;
  int rc = anx::batch_export_compact(b->model->replicas[(size_t)b->shards[0].replica].dev, b->shards[0].b, dst, capacity, stream, used, err);
  return rc ? fail(rc, err) : ANX_OK;
}
int anx_batch_gather_compact(const anx_batch* b, int dst_device, void* device_dst, size_t capacity, size_t* shard_offsets, size_t* used) {
  if (!b || !device_dst || !used) return fail(ANX_EINVAL, "NULL argument");
  if (b->rescore) return fail(ANX_EINVAL, "host-rescored confusable batches have no device-side export: anx_batch_fetch");
  const size_t S = b->shards.size();
  std::vector<size_t> off(S + 1, 0);
  for (size_t g = 0; g < S; ++g) {
    if (!b->shards[g].b) return fail(ANX_EINVAL, "batch has not been run");
    off[g + 1] = off[g] + ((anx::batch_compact_bytes(b->shards[g].b) + 255) & ~(size_t)255);
  }
  *used = off[S];
  if (shard_offsets) for (size_t g = 0; g <= S; ++g) shard_offsets[g] = off[g];
  if (capacity < off[S]) return fail(ANX_ELIMIT, "gather buffer too small: " + std::to_string(off[S]) + " bytes needed");
  // every shard from its replica's own thread and stream: the exports run side by side, the copies use the links of their own devices
  char* dst = static_cast<char*>(device_dst);
  return on_shards(b, [&](size_t g, std::string& err) {
    const Shard& s = b->shards[g];
    return anx::batch_gather_compact(b->model->replicas[(size_t)s.replica].dev, s.b, dst_device, dst + off[g], off[g + 1] - off[g], shard_stream(b, s, nullptr), err);
  });
}
int anx_batch_num_shards(const anx_batch* b) { return b ? (int)b->shards.size() : 0; }
int anx_batch_shard_info(const anx_batch* b, int shard, int* device, size_t* first_input, size_t* n_inputs) {
  if (!b || shard < 0 || (size_t)shard >= b->shards.size()) return fail(ANX_EINVAL, "no such shard");
  const Shard& s = b->shards[(size_t)shard];
  if (device) *device = b->model->replicas[(size_t)s.replica].device;
  if (first_input) *first_input = s.idx.empty() ? s.lo : s.idx[0];
  if (n_inputs) *n_inputs = s.n;
  return ANX_OK;
}
int anx_batch_shard_inputs(const anx_batch* b, int shard, const uint32_t** indices) {
  if (!b || !indices || shard < 0 || (size_t)shard >= b->shards.size()) return fail(ANX_EINVAL, "no such shard");
  const Shard& s = b->shards[(size_t)shard];
  *indices = s.idx.empty() ? nullptr : s.idx.data();
  return ANX_OK;
}
int anx_batch_get_stats(const anx_batch* b, anx_batch_stats* out, size_t struct_size) {
  if (!b || !out) return fail(ANX_EINVAL, "NULL argument");
  // the struct as ABI version 2 introduced it ended with n_conf_scripts; anything shorter (or absurdly long) is not a struct size
  if (struct_size < offsetof(anx_batch_stats, n_conf_scripts) + sizeof(uint64_t) || struct_size > 4096) return fail(ANX_EINVAL, "struct_size is not the size of an anx_batch_stats");
  anx_batch_stats full;
  anx_batch_stats* s = &full;
  anx::batch_stats(b->shards[0].b, s);
  for (size_t g = 1; g < b->shards.size(); ++g) {  // counts add up, times are those of the slowest replica
    anx_batch_stats t;
    anx::batch_stats(b->shards[g].b, &t);
    s->n_queries += t.n_queries; s->n_pairs += t.n_pairs; s->n_class_tests += t.n_class_tests; s->n_results += t.n_results;
    s->n_scan_blocks += t.n_scan_blocks; s->n_pair_slots += t.n_pair_slots; s->n_survivors += t.n_survivors; s->n_selected += t.n_selected; s->n_prefiltered_in_scan += t.n_prefiltered_in_scan; s->n_conf_scripts += t.n_conf_scripts; s->n_adj_tiles += t.n_adj_tiles; s->n_adj_records += t.n_adj_records; s->n_adj_records_first += t.n_adj_records_first;
    for (int i = 0; i < 5; ++i) s->n_tests_kind[i] += t.n_tests_kind[i];
    s->ms_scan = std::max(s->ms_scan, t.ms_scan); s->ms_group = std::max(s->ms_group, t.ms_group); s->ms_score = std::max(s->ms_score, t.ms_score);
    s->ms_rank = std::max(s->ms_rank, t.ms_rank); s->ms_total = std::max(s->ms_total, t.ms_total);
    s->ms_scan_kernel = std::max(s->ms_scan_kernel, t.ms_scan_kernel); s->ms_filter_score_kernel = std::max(s->ms_filter_score_kernel, t.ms_filter_score_kernel);
  }
  memcpy(out, s, std::min(struct_size, sizeof full));  // a caller compiled against an older (shorter) struct stays in bounds
  return ANX_OK;
}
void anx_device_pool_trim(int device) { anx::device_pool_trim(device); }

void anx_batch_free(anx_batch* b) {
  if (!b) return;
  if (b->shards.size() > 1) (void)on_shards(b, [&](size_t g, std::string&) { anx::batch_free(b->shards[g].b); b->shards[g].b = nullptr; return ANX_OK; });
  free_shards(b);
  delete b;
}

int anx_find_variants_batch(const anx_model* m, const char* const* utf8, size_t n, const anx_params* p,
                            anx_result** out_rows, size_t** out_offsets) {
  if (!out_rows || !out_offsets) return fail(ANX_EINVAL, "NULL output argument");
  // One device batch holds at most 2^31 pair-list slots (~10 M queries of BASELINE config 2): larger calls are run as
  // consecutive rounds of ANX_MAX_BATCH inputs per replica and their CSR results concatenated.
  const size_t per_round = (size_t)anx::switches().max_batch * std::max<size_t>(1, m ? m->replicas.size() : 1);
  // the small call (engine small_path.hpp): the reference's own granularity -- one string per call, 1 000 per batch
  // (src/lib.rs:972, src/bin/analiticcl.rs:416) -- in eleven launches and one host wait instead of the batch pipeline
  if (m && utf8 && p && n >= 1 && n <= 4096 && m->host.built && m->replicas.size() == 1 && m->replicas[0].dev && anx::switches().small_path) {
    bool rescore = false;
    const anx_params dp = device_params(m, p, &rescore);
    if (!rescore) {   // (confusables: weighted by the batch path)
      std::string err;
      const int rc = anx::small_find(m->host, m->replicas[0].dev, utf8, n, dp, out_rows, out_offsets, err);
      if (rc == ANX_OK) return ANX_OK;
      if (rc < 0) return fail(rc, err);
    }
  }
  if (n <= per_round) {
    anx_batch* b = anx_batch_encode(m, utf8, n, p);
    if (!b) return g_code ? g_code : ANX_EINVAL;
    int rc = anx_batch_run(m, b, nullptr);
    if (rc == ANX_OK) rc = anx_batch_fetch(b, out_rows, out_offsets);
    anx_batch_free(b);
    return rc;
  }
  size_t* offs = static_cast<size_t*>(calloc(n + 1, sizeof(size_t)));
  anx_result* rows = nullptr;
  size_t total = 0, cap = 0;
  if (!offs) return fail(ANX_EINVAL, "out of memory");
  for (size_t lo = 0; lo < n; lo += per_round) {
    const size_t cnt = std::min(per_round, n - lo);
    anx_result* r = nullptr;
    size_t* o = nullptr;
    const int rc = anx_find_variants_batch(m, utf8 + lo, cnt, p, &r, &o);
    if (rc != ANX_OK) { free(rows); free(offs); return rc; }
    const size_t add = o[cnt];
    if (total + add > cap) {
      cap = std::max(total + add, cap * 2);
      anx_result* grown = static_cast<anx_result*>(realloc(rows, std::max<size_t>(1, cap) * sizeof(anx_result)));
      if (!grown) { free(rows); free(offs); anx_results_free(r, o); return fail(ANX_EINVAL, "out of memory"); }
      rows = grown;
    }
    if (add) memcpy(rows + total, r, add * sizeof(anx_result));
    for (size_t i = 0; i < cnt; ++i) offs[lo + i] = total + o[i];
    total += add;
    anx_results_free(r, o);
  }
  offs[n] = total;
  if (!rows) rows = static_cast<anx_result*>(malloc(sizeof(anx_result)));
  *out_rows = rows;
  *out_offsets = offs;
  return ANX_OK;
}
// ---- pipeline: encode(i + 2) / run(i + 1) / fetch(i) in flight for ONE caller thread -------------------------------------------------
// Three library threads, one per stage, and two alternating run streams (single-replica models): while batch i is downloaded
// (copy engine), batch i + 1 runs and batch i + 2 is uploaded and encoded.  The stages are the public entry points
// (anx_batch_encode_packed -> anx_batch_run -> anx_batch_fetch_compact); results come back in submission order.
struct PipeJob {
  const char* blob = nullptr;
  size_t blob_len = 0, n = 0;
  anx_params p;
  anx_batch* b = nullptr;
  int rc = ANX_OK;
  std::string err;
  anx_topk_record* rows = nullptr;
  uint32_t* offs = nullptr;
  int stage = 0;  // 0 submitted, 1 encoded, 2 run, 3 fetched (done)
  uint64_t seq = 0;
};
struct anx_pipeline {
  const anx_model* m = nullptr;
  size_t depth = 6;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::shared_ptr<PipeJob>> jobs;  // in submission order; the front is what anx_pipeline_next returns
  bool stop = false;
  uint64_t next_seq = 0;
  void* streams[2] = {nullptr, nullptr};
  void* enc_stream = nullptr;  // the encode thread's own stream (single-replica models; highest stream priority)
  void* fetch_stream = nullptr;  // the downloads' stream (highest stream priority as well)
  int running = 0;             // runs launched and not yet waited for (at most 2: the scan of one under the tail of the other)
  std::thread th[3];
};
static void pipeline_stage(anx_pipeline* pl, int stage) {
  if (stage == 0 && pl->enc_stream) anx::encoder_stream_set_override(pl->enc_stream);
  if (stage == 0 && pl->streams[0]) t_runs_on_caller_stream = true;
  for (;;) {
    std::shared_ptr<PipeJob> job;
    {
      std::unique_lock<std::mutex> lk(pl->mu);
      pl->cv.wait(lk, [&]() {
        if (pl->stop) return true;
        for (auto& j : pl->jobs) if (j->stage == stage) { job = j; return true; }  // the oldest job waiting for this stage
        return false;
      });
      if (!job) return;  // stop
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (job->rc == ANX_OK) {
      if (stage == 0) {
        job->b = anx_batch_encode_packed(pl->m, job->blob, job->blob_len, job->n, &job->p);
        if (!job->b) { job->rc = g_code ? g_code : ANX_EINVAL; job->err = g_err; }
        else {
          // The run is ENQUEUED by this thread, as soon as the batch is encoded and fewer than two runs are in flight; the run thread
          // only waits (round 5: the run thread launched job i + 1 after it had waited for job i -- the device idled for the length
          // of a launch sequence and a wake-up between any two runs).
          {
            std::unique_lock<std::mutex> lk(pl->mu);
            pl->cv.wait(lk, [&]() { return pl->stop || pl->running < 2; });
            ++pl->running;
          }
          job->rc = anx_batch_run_async(pl->m, job->b, pl->streams[job->seq & 1]);
          if (job->rc) {
            job->err = g_err;
            std::lock_guard<std::mutex> lk(pl->mu);
            --pl->running;
          }
        }
      } else if (stage == 1) {
        job->rc = anx_batch_wait(pl->m, job->b);
        if (job->rc) job->err = g_err;
        { std::lock_guard<std::mutex> lk(pl->mu); --pl->running; }
      } else {
        if (pl->fetch_stream)  // the run has been waited for: its rows are downloaded on the pipeline's download stream
          for (Shard& s_ : job->b->shards) anx::batch_set_last_stream(s_.b, pl->fetch_stream);
        job->rc = anx_batch_fetch_compact(job->b, &job->rows, &job->offs);
        if (job->rc) job->err = g_err;
      }
    }
    if (anx::switches().encode_timing)
      fprintf(stderr, "[anx pipeline] job %llu stage %d: %.2f ms\n", (unsigned long long)job->seq, stage,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (stage == 2 || job->rc != ANX_OK) {  // the batch's device buffers go back to the pool as soon as its rows are on the host
      if (job->b) anx_batch_free(job->b);
      job->b = nullptr;
    }
    {
      std::lock_guard<std::mutex> lk(pl->mu);
      job->stage = job->rc == ANX_OK ? stage + 1 : 3;
    }
    pl->cv.notify_all();
  }
}
anx_pipeline* anx_pipeline_new(const anx_model* m, int depth) {
  if (!m) { fail(ANX_EINVAL, "NULL model"); return nullptr; }
  if (check_resident(m)) return nullptr;
  anx_pipeline* pl = new anx_pipeline();
  pl->m = m;
  pl->depth = depth > 0 ? (size_t)depth : 6;
  if (m->replicas.size() == 1) {  // a multi-replica model runs every shard on its replica's own stream
    std::string err;
    pl->enc_stream = anx::stream_create(m->replicas[0].device, err, true);  // (nullptr: the encoder's pooled streams)
    // Streams of a pipeline: the encoder's and the downloads' at the highest stream priority, two of normal priority that carry the
    // runs, alternating per job (the scan of one job under the tail of the other).  The runtime maps streams onto a handful of hardware
    // queues per priority level (GPU_MAX_HW_QUEUES, 4 by default): with the runs on the library's pair of run streams AND two normal-
    // priority download streams, a 70 MB download could share a hardware queue with a run stream and hold up that run's kernels --
    // depending on which streams the process had created before (bench.py's end-to-end section: 272 M queries/s inside the full
    // run, 355 M in a fresh process).  High-priority streams never share a queue with normal ones, and a download on a stream of its
    // own is never enqueued behind the run after next (runs AND downloads on the two run streams: 234 M in a fresh process).
    pl->fetch_stream = anx::stream_create(m->replicas[0].device, err, true);
    for (void*& st : pl->streams)
      if (!(st = anx::stream_create(m->replicas[0].device, err))) {
        for (void* x : pl->streams) if (x) anx::stream_destroy(m->replicas[0].device, x);
        if (pl->enc_stream) anx::stream_destroy(m->replicas[0].device, pl->enc_stream);
        if (pl->fetch_stream) anx::stream_destroy(m->replicas[0].device, pl->fetch_stream);
        delete pl;
        fail(ANX_ENODEVICE, err);
        return nullptr;
      }
  }
  for (int s = 0; s < 3; ++s) pl->th[s] = std::thread(pipeline_stage, pl, s);
  return pl;
}
int anx_pipeline_submit_packed(anx_pipeline* pl, const char* blob, size_t blob_len, size_t n, const anx_params* p) {
  if (!pl || (!blob && n) || !p) return fail(ANX_EINVAL, "NULL argument");
  auto job = std::make_shared<PipeJob>();
  job->blob = blob; job->blob_len = blob_len; job->n = n; job->p = *p;
  {
    std::unique_lock<std::mutex> lk(pl->mu);
    // a job leaves the queue in anx_pipeline_next only (finished jobs count until their results were taken): waiting here would
    // block the one caller thread the pipeline is made for, for good
    if (pl->jobs.size() >= pl->depth) return fail(ANX_ELIMIT, "pipeline full: take a result with anx_pipeline_next before submitting another job");
    job->seq = pl->next_seq++;
    pl->jobs.push_back(job);
  }
  pl->cv.notify_all();
  return ANX_OK;
}
int anx_pipeline_pending(const anx_pipeline* pl) {
  if (!pl) return 0;
  std::lock_guard<std::mutex> lk(const_cast<anx_pipeline*>(pl)->mu);
  return (int)pl->jobs.size();
}
int anx_pipeline_next(anx_pipeline* pl, anx_topk_record** rows, uint32_t** offs, size_t* n) {
  if (!pl || !rows || !offs) return fail(ANX_EINVAL, "NULL argument");
  std::shared_ptr<PipeJob> job;
  {
    std::unique_lock<std::mutex> lk(pl->mu);
    if (pl->jobs.empty()) return fail(ANX_EINVAL, "no job in flight");
    job = pl->jobs.front();
    pl->cv.wait(lk, [&]() { return job->stage == 3; });
    pl->jobs.pop_front();
  }
  pl->cv.notify_all();
  if (job->rc != ANX_OK) return fail(job->rc, job->err);
  *rows = job->rows;
  *offs = job->offs;
  if (n) *n = job->n;
  return ANX_OK;
}
void anx_pipeline_free(anx_pipeline* pl) {
  if (!pl) return;
  for (;;) {  // the jobs in flight finish; their results are dropped
    std::shared_ptr<PipeJob> job;
    {
      std::unique_lock<std::mutex> lk(pl->mu);
      if (pl->jobs.empty()) break;
      job = pl->jobs.front();
      pl->cv.wait(lk, [&]() { return job->stage == 3; });
      pl->jobs.pop_front();
    }
    if (job->rc == ANX_OK) anx_compact_free(job->rows, job->offs);
  }
  {
    std::lock_guard<std::mutex> lk(pl->mu);
    pl->stop = true;
  }
  pl->cv.notify_all();
  for (std::thread& t : pl->th) t.join();
  for (void* st : pl->streams) if (st) anx::stream_destroy(pl->m->replicas[0].device, st);
  if (pl->fetch_stream) anx::stream_destroy(pl->m->replicas[0].device, pl->fetch_stream);
  if (pl->enc_stream) anx::stream_destroy(pl->m->replicas[0].device, pl->enc_stream);
  delete pl;
}

void anx_results_free(anx_result* rows, size_t* offsets) {
  anx::host_result_free(rows);  // a cached pinned buffer of batch_fetch, or a malloc block
  free(offsets);
}

}  // extern "C"
