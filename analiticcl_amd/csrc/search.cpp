// search.cpp -- search mode: the caller of the hot path.
// Host-side restatement of VariantModel::find_all_matches (/root/reference/src/lib.rs:1790-1957), the text
// segmentation of src/search.rs:190-336 and the sequence decoding of src/lib.rs:2088-2495 (context rules: contextrules.cpp),
// re-organised so that ALL segments of one n-gram order -- over every hard-boundary stretch of every input text --
// go to the device as ONE anx_find_variants_batch call (the reference calls find_variants once per segment from a
// rayon par_iter, src/lib.rs:1883-1899).  Order n depends on the unigram results through redundant_match, so the
// orders are processed one after another.
#include <pthread.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <map>
#include <mutex>
#include <new>
#include <iterator>
#include <thread>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <limits>
#include <vector>

#include "engine.h"
#include "host_model.h"
#include "portable_log.hpp"

extern "C" {
int anx_find_variants_batch(const anx_model*, const char* const*, size_t, const anx_params*, anx_result**, size_t**);
void anx_results_free(anx_result*, size_t*);
int anx_last_error_code(void);
}
const anx::HostModel& anx_host_of(const anx_model* m);  // capi.cpp
const anx::DeviceLexicon* anx_replica_of(const anx_model* m, size_t i);
int anx_replica_device(const anx_model* m, size_t i);  // capi.cpp
int anx_fail(int code, const std::string& msg);         // capi.cpp
anx::Batch* anx_batch_single(const anx_batch* b);       // capi.cpp

namespace {

using anx::HostModel;
std::atomic<uint64_t> g_lat_ns[6];  // ANX_SEARCH_TIMING: time inside most_likely_sequence by part, all threads
#define g_lat_timing (anx::switches().search_timing != 0)
inline uint64_t lat_now() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct LatLap { uint64_t t; LatLap() : t(g_lat_timing ? lat_now() : 0) {} void lap(int i) { if (g_lat_timing) { const uint64_t n = lat_now(); g_lat_ns[i] += n - t; t = n; } } };

// The host threads of search mode: one pool for the process (usable cores - 1 threads; the caller of a loop works too).  A call
// runs a dozen parallel loops per part, several parts at a time: starting and joining 16 threads per loop cost ~0.5 ms each,
// a sixth of a part's time.  The threads start on first use and are JOINED by anx_shutdown() and when the library is unloaded
// (a destructor function: dlclose / exit with no call in flight); the next call starts a fresh pool.  A fork()ed child does not
// inherit threads: its atfork handler forgets the parent's pool object (leaked: its mutex may be held by a thread that does not
// exist in the child), so the child's first call builds its own.
class HostPool;
std::atomic<HostPool*> g_pool{nullptr};
// The loops of a call's parts queue up with the part's number: a pool thread takes the waiting task of the EARLIEST part (first come,
// first served inside one part), so that the first part -- whose host phase the device waits for -- gets the threads first.  A task
// carries its number to the thread that runs it (loops started from inside a loop body).
thread_local int tl_pool_prio = 0;
std::mutex g_pool_mu;
class HostPool {
 public:
  static HostPool& get() {
    HostPool* p = g_pool.load(std::memory_order_acquire);
    if (p) return *p;
    std::lock_guard<std::mutex> g(g_pool_mu);
    p = g_pool.load(std::memory_order_relaxed);
    if (!p) {
      static std::once_flag fork_once;
      std::call_once(fork_once, []() { pthread_atfork(nullptr, nullptr, []() { g_pool.store(nullptr); new (&g_pool_mu) std::mutex; }); });
      p = new HostPool(std::max(1u, std::min(64u, anx::usable_hw_threads())));
      g_pool.store(p, std::memory_order_release);
    }
    return *p;
  }
  // joins the pool's threads and deletes it (no loop may be running); the next get() starts a new one
  static void shutdown() {
    std::lock_guard<std::mutex> g(g_pool_mu);
    HostPool* p = g_pool.exchange(nullptr);
    if (!p) return;
    {
      std::lock_guard<std::mutex> g2(p->mu_);
      p->stop_ = true;
    }
    p->cv_.notify_all();
    for (std::thread& t : p->threads_) t.join();
    delete p;
  }
  unsigned width() const { return nthreads_ + 1; }
  // f() on a pool thread, not waited for (on the caller when there is no pool thread); shutdown() runs what is still queued
  // (a queue of its own, served by the pool's threads when no loop has work for them: a caller waiting for its loop in run() takes
  // loop tasks only -- it must not spend milliseconds of its critical path on somebody's frees)
  void post(std::function<void()> f) {
    if (!nthreads_) { f(); return; }
    std::function<void()> overdue;  // loops that never leave the pool idle must not let posted work pile up: the poster then takes the oldest
    {
      std::lock_guard<std::mutex> g(mu_);
      bg_.push_back(std::move(f));
      if (bg_.size() > 16) { overdue = std::move(bg_.front()); bg_.pop_front(); }
    }
    cv_.notify_one();
    if (overdue) overdue();
  }
  // body() on up to `helpers` pool threads and on the caller; returns when every started body has returned.  A caller that waits
  // takes queued tasks itself, so loops started from inside a pool thread cannot starve each other.
  void run(unsigned helpers, const std::function<void()>& body) {
    helpers = std::min(helpers, nthreads_);
    struct State { std::atomic<unsigned> remaining; std::mutex m; std::condition_variable cv; };
    auto st = std::make_shared<State>();
    st->remaining.store(helpers);
    if (helpers) {
      {
        const int prio = tl_pool_prio;
        std::lock_guard<std::mutex> g(mu_);
        auto at = q_.end();
        while (at != q_.begin() && std::prev(at)->prio > prio) --at;   // behind the waiting tasks of this and of earlier parts
        for (unsigned i = 0; i < helpers; ++i)
          at = std::next(q_.insert(at, Task{prio, [st, &body, prio]() {
            const int saved = tl_pool_prio;
            tl_pool_prio = prio;
            body();
            tl_pool_prio = saved;
            if (st->remaining.fetch_sub(1) == 1) { std::lock_guard<std::mutex> g2(st->m); st->cv.notify_all(); }
          }}));
      }
      cv_.notify_all();
    }
    body();
    while (st->remaining.load() != 0) {
      std::function<void()> f;
      {
        std::lock_guard<std::mutex> g(mu_);
        if (!q_.empty()) { f = std::move(q_.front().f); q_.pop_front(); }
      }
      if (f) { f(); continue; }
      std::unique_lock<std::mutex> l(st->m);
      st->cv.wait_for(l, std::chrono::microseconds(200), [&]() { return st->remaining.load() == 0; });
    }
  }

 private:
  explicit HostPool(unsigned hw) : nthreads_(hw > 1 ? hw - 1 : 0) {
    for (unsigned i = 0; i < nthreads_; ++i)
      threads_.emplace_back([this]() {
        for (;;) {
          std::function<void()> f;
          {
            std::unique_lock<std::mutex> l(mu_);
            cv_.wait(l, [&]() { return stop_ || !q_.empty() || !bg_.empty(); });
            if (!q_.empty()) { f = std::move(q_.front().f); q_.pop_front(); }
            else if (!bg_.empty()) { f = std::move(bg_.front()); bg_.pop_front(); }
            else return;  // stop_ and nothing left to do
          }
          f();
        }
      });
  }
  unsigned nthreads_;
  bool stop_ = false;
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_;
  struct Task { int prio; std::function<void()> f; };
  std::deque<Task> q_;                    // loop tasks, ordered by (part number, arrival)
  std::deque<std::function<void()>> bg_;  // post(): run when no loop task waits
};

struct RowView {  // the ranked variants of a segment: a range of one n-gram order's result array (kept until the end)
  const anx_result* p = nullptr;
  size_t n = 0;
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  const anx_result& operator[](size_t i) const { return p[i]; }
  const anx_result* begin() const { return p; }
  const anx_result* end() const { return p + n; }
};

struct Span {  // Match without variants (boundaries, segments); byte offsets into the text
  Span(size_t b, size_t e) : begin(b), end(e) {}
  size_t begin, end;
  uint32_t n = 0;
  int64_t var_slot = -1;  // index into the per-order result table (-1: no lookup done => variants = None)
  RowView variants;
  bool has_variants = false;  // Some(vec) vs None
  int selected = -1;
  // Match.tag / Match.seqnr (src/search.rs:60-66), set by context rules: entries [tag0, tag0 + ntags) of the tag pool of stretch
  // tag_stretch (a Span stays trivially copyable: four million of them are moved around per 12 MB of text)
  uint32_t tag0 = 0, ntags = 0, tag_stretch = 0;
};

// find_boundaries (src/search.rs:190-233)
void find_boundaries(const char* text, size_t len, std::vector<Span>& out) {
  bool open = false;
  size_t b = 0;
  for (size_t i = 0; i < len;) {
    int l;
    bool alpha;
    const unsigned char c0 = (unsigned char)text[i];
    if (c0 < 0x80) {  // ASCII: the letters are the alphabetic ones (the table's ASCII part), one byte each
      l = 1;
      alpha = (unsigned)((c0 | 32u) - 'a') < 26u;
    } else {
      const uint32_t cp = anx::utf8_decode_at(text + i, len - i, &l);
      alpha = anx::is_alphabetic_cp(cp);
    }
    if (open) {
      if (alpha) { out.push_back(Span{b, i}); open = false; }
    } else if (!alpha) { b = i; open = true; }
    i += (size_t)l;
  }
  if (open) out.push_back(Span{b, len});
  else out.push_back(Span{len, len});
}
enum Strength { WEAK, NORMAL, HARD };
// classify_boundaries (src/search.rs:238-258): byte length > 1 => Hard
Strength classify(const char* text, const std::vector<Span>& bs, size_t i) {
  if (i == bs.size() - 1) return HARD;
  const size_t l = bs[i].end - bs[i].begin;
  if (l > 1) return HARD;
  if (l == 1 && (text[bs[i].begin] == '\'' || text[bs[i].begin] == '-' || text[bs[i].begin] == '_')) return WEAK;
  return NORMAL;
}
// Match::internal_boundaries (src/search.rs:103-120), with its single-boundary behaviour
size_t count_internal(const Span& m, const Span* bs, size_t nb) {
  long begin = -1;
  size_t end = 0;
  for (size_t i = 0; i < nb; ++i)
    if (bs[i].begin > m.begin && bs[i].end < m.end) {
      if (begin < 0) begin = (long)i;
      else end = i + 1;
    }
  if (begin < 0 || (size_t)begin >= end) return 0;
  return end - (size_t)begin;
}
// find_match_ngrams (src/search.rs:262-313)
// states: optionally, per segment pushed, the lattice states it connects as source | destination << 12 (state i + 1 = behind boundary
// i of the stretch, state 0 = its start: what build_lattice finds by comparing offsets) -- 0xFFFFFFFF for the rare tail segment,
// whose states the caller looks up
void find_match_ngrams(const char* text, const Span* bs, size_t nb, uint32_t order, size_t begin, size_t end,
                       std::vector<Span>& out, std::vector<uint32_t>* states = nullptr) {
  size_t i = 0;
  while (i + order - 1 < nb) {
    const Span& boundary = bs[i + order - 1];
    if (boundary.begin > end) break;
    if (boundary.begin > begin && !(boundary.begin - begin == 1 && text[begin] == ' ')) {
      Span s{begin, boundary.begin};
      s.n = order;
      out.push_back(s);
      // the segment starts behind boundary i - 1 (or at the stretch's start) and ends at boundary i + order - 1; a zero-length
      // boundary (only the one find_boundaries appends at the end of a text) could tie with those: then the caller looks it up
      if (states) states->push_back((bs[nb - 1].begin == bs[nb - 1].end) ? 0xFFFFFFFFu : ((uint32_t)i | ((uint32_t)(i + order) << 12)));
    }
    begin = bs[i].end;
    ++i;
  }
  if (begin < end && !(end - begin == 1 && text[begin] == ' ')) {
    Span s{begin, end};
    s.n = order;
    if (count_internal(s, bs, nb) == order) { out.push_back(s); if (states) states->push_back(0xFFFFFFFFu); }
  }
}
// redundant_match (src/search.rs:317-336)
bool redundant_match(const Span& cand, const std::vector<Span>& matches) {
  for (const Span& r : matches) {
    if (r.n != 1) break;
    if (r.begin >= cand.begin && r.end <= cand.end) {
      if (!r.has_variants) return false;
      if (r.variants.empty() || r.variants[0].dist_score < 1.0) return false;
    }
  }
  return true;
}
double vr_score(const anx_result& r, float fw) {  // src/types.rs:335-341
  if (fw == 0.0f) return r.dist_score;
  return (r.dist_score + ((double)fw * r.freq_score)) / (1.0 + (double)fw);
}

struct OutSym {  // OutputSymbol, src/search.rs:133-150
  uint64_t vocab_id;
  size_t match_index;
  int variant_index;
  size_t boundary_index;
};

// most_likely_sequence (src/lib.rs:2088-2495).  The reference decodes with rustfst's
// shortest_path(nshortest = max_seq) over a lattice whose states are the boundaries; this is the exact k-best over the
// same DAG.  The order among equal-cost paths is rustfst-internal in the reference and is not pinned.
typedef std::vector<std::pair<uint16_t, uint8_t>> TagPool;
void most_likely_sequence(const HostModel& m, const char* text, std::vector<Span>& matches, const Span* bs, size_t nb,
                          size_t end_offset, const anx_search_params& p, std::vector<Span>& out, TagPool& tagpool, uint32_t stretch_index) {
  LatLap lat;
  struct Arc { float cost; size_t dst; long sym; };
  const size_t nstates = nb + 1;
  // per-thread buffers that keep their capacity from one stretch to the next: the candidate lists of a state grow to thousands
  // of nodes, and growing them afresh for each of the ~90 k stretches of a 12 MB text was a third of the decoding time
  static thread_local std::vector<std::vector<Arc>> arcs;
  if (arcs.size() < nstates) arcs.resize(nstates);
  for (size_t i = 0; i < nstates; ++i) arcs[i].clear();
  static thread_local std::vector<OutSym> symbols;
  static thread_local std::vector<size_t> finals;
  symbols.assign(1, OutSym{0, 0, -1, 0});
  finals.clear();
  for (size_t i = 0; i < nb; ++i)
    if (bs[i].begin == end_offset || bs[i].end == end_offset) finals.push_back(i + 1);
  for (size_t mi = 0; mi < matches.size(); ++mi) {
    const Span& mt = matches[mi];
    long prevb = -1, nextb = -1;
    for (size_t i = 0; i < nb; ++i) {
      if (mt.begin == bs[i].end) prevb = (long)i;
      else if (mt.end == bs[i].begin) nextb = (long)i;
    }
    if (nextb < 0) continue;  // "next boundary must exist"
    const long n = prevb >= 0 ? nextb - prevb : nextb + 1;
    const size_t src = prevb >= 0 ? (size_t)prevb + 1 : 0, dst = (size_t)nextb + 1;
    if (mt.has_variants && !mt.variants.empty()) {
      for (size_t vi = 0; vi < mt.variants.size(); ++vi) {
        symbols.push_back(OutSym{mt.variants[vi].vocab_id, mi, (int)vi, (size_t)nextb});
        const float cost = (float)n + (1.0f - (float)vr_score(mt.variants[vi], p.base.freq_weight));
        arcs[src].push_back(Arc{cost, dst, (long)symbols.size() - 1});
      }
    } else if (n == 1) {
      symbols.push_back(OutSym{0, mi, -1, (size_t)nextb});
      arcs[src].push_back(Arc{(float)n + 1.0f, dst, (long)symbols.size() - 1});
    }
  }
  for (size_t i = 0; i < nb; ++i) arcs[i].push_back(Arc{100.0f, i + 1, -1});  // failsafe epsilon transitions
  if (symbols.size() == 1 || finals.empty()) { out.insert(out.end(), matches.begin(), matches.end()); return; }
  lat.lap(0);
  // k-best paths into every state, kept as back-pointers (source state, rank there, symbol).  Arcs only run forward, so the
  // states are in topological order by index and best[s] is final before any state behind it is built.
  // The K best of a state = the K smallest of { best[src][r].cost + arc.cost } under (cost, source state, arc number, r) -- the
  // order a stable sort by cost gives the candidates when they are listed arc by arc.  Per incoming arc the candidates are
  // already sorted (best[src] is, and adding a constant is monotone in f32), so the K smallest come out of a K-way merge: a
  // binary heap with one entry per incoming arc, keyed (cost, arc), K pops.  (Until round 2 every candidate below a running
  // bound was materialised and the list cut with nth_element: 7.5 of the decoder's 9.8 s on a 12 MB text.)
  struct Node { float cost; uint32_t ps, pr; long sym; uint32_t seq; };
  const size_t K = std::max<uint32_t>(1, p.max_seq);
  static thread_local std::vector<std::vector<Node>> best;
  if (best.size() < nstates) best.resize(nstates);
  for (size_t i = 0; i < nstates; ++i) best[i].clear();
  best[0].push_back(Node{0.0f, UINT32_MAX, 0, -1, 0});  // the start node
  auto cmp = [](const Node& a, const Node& b) { return a.cost < b.cost || (a.cost == b.cost && a.seq < b.seq); };
  auto cut_to_k = [K, &cmp](std::vector<Node>& v) {
    if (v.size() > K) {
      std::nth_element(v.begin(), v.begin() + (long)K, v.end(), cmp);
      v.resize(K);
    }
  };
  struct In { float cost; uint32_t src; long sym; };
  static thread_local std::vector<std::vector<In>> in;  // incoming arcs per state, ordered by (source state, arc number)
  if (in.size() < nstates) in.resize(nstates);
  for (size_t i = 0; i < nstates; ++i) in[i].clear();
  for (size_t s = 0; s < nstates; ++s)
    for (const Arc& a : arcs[s]) in[a.dst].push_back(In{a.cost, (uint32_t)s, a.sym});
  struct Head { float cost; uint32_t arc, r; };
  static thread_local std::vector<Head> heap;
  auto before = [](const Head& a, const Head& b) { return a.cost < b.cost || (a.cost == b.cost && a.arc < b.arc); };
  for (size_t d = 1; d < nstates; ++d) {
    const std::vector<In>& inc = in[d];
    heap.clear();
    for (size_t ai = 0; ai < inc.size(); ++ai)
      if (!best[inc[ai].src].empty()) heap.push_back(Head{best[inc[ai].src][0].cost + inc[ai].cost, (uint32_t)ai, 0u});
    size_t hn = heap.size();
    auto sift_down = [&](size_t i) {
      const Head x = heap[i];
      for (;;) {
        size_t c = 2 * i + 1;
        if (c >= hn) break;
        if (c + 1 < hn && before(heap[c + 1], heap[c])) ++c;
        if (!before(heap[c], x)) break;
        heap[i] = heap[c];
        i = c;
      }
      heap[i] = x;
    };
    for (size_t i = hn / 2; i-- > 0;) sift_down(i);
    std::vector<Node>& dv = best[d];
    while (hn && dv.size() < K) {
      const Head h = heap[0];
      const In& a = inc[h.arc];
      dv.push_back(Node{h.cost, a.src, h.r, a.sym, 0u});
      const std::vector<Node>& sv = best[a.src];
      if (h.r + 1 < sv.size()) heap[0] = Head{sv[h.r + 1].cost + a.cost, h.arc, h.r + 1};
      else heap[0] = heap[--hn];
      if (hn) sift_down(0);
    }
  }
  lat.lap(1);
  static thread_local std::vector<Node> ends;
  ends.clear();
  for (size_t f : finals) ends.insert(ends.end(), best[f].begin(), best[f].end());
  for (size_t i = 0; i < ends.size(); ++i) ends[i].seq = (uint32_t)i;
  cut_to_k(ends);
  std::sort(ends.begin(), ends.end(), cmp);
  // the symbols of final path i, walked back over the back-pointers: only wanted for the path that wins (and, with context
  // rules, for every path -- the rules look at whole sequences); costs and LM sums live on the lattice nodes
  const size_t npaths = ends.size();
  auto path_syms = [&](size_t i, std::vector<long>& syms) {
    syms.clear();
    for (Node cur = ends[i]; cur.ps != UINT32_MAX; cur = best[cur.ps][cur.pr])
      if (cur.sym >= 0) syms.push_back(cur.sym);
    std::reverse(syms.begin(), syms.end());
  };
  static thread_local std::vector<long> syms_tmp;
  lat.lap(2);
  // rerank (src/lib.rs:2318-2425)
  const bool use_lm = m.have_lm && p.lm_weight > 0.0f;
  const bool use_rules = !m.context_rules.empty();
  double best_ppl = 999999.0, best_ctx = 0.0;
  float best_cost = (float)(nb - 1) * 2.0f;
  static thread_local std::vector<double> ppls, ctx;
  ppls.assign(npaths, 0.0);
  ctx.assign(npaths, 1.0);
  std::vector<std::vector<std::vector<anx::PatternMatchResult>>> ctx_results(use_rules ? npaths : 0);
  std::vector<std::pair<uint64_t, uint32_t>> idseq;
  // LM scoring of up to max_seq paths of one lattice: the paths share almost all of their bigrams, so the tokens of every
  // symbol (its n-gram parts + the boundary text behind it, src/lib.rs:2580-2629) are looked up once, and every bigram term
  // (src/lib.rs:2632-2674) once per lattice; the f32 sum runs over the same terms in the same order as lm_score_tokens.
  static thread_local std::vector<uint32_t> tok_off, btok_off;
  static thread_local std::vector<int64_t> tok, btok;
  struct Term { int64_t a, b; float v; uint32_t gen; };
  static thread_local std::vector<Term> memo(1024, Term{0, 0, 0.0f, 0u});  // slots of earlier lattices are stale by their generation
  static thread_local uint32_t memo_gen = 0;
  // LM sum of the path prefix that ends in a lattice node, stamped with the lattice's generation (memo_gen): the per-node
  // table is never cleared, only grown (clearing 250 nodes x 15 states per lattice was most of the LM time)
  struct LmState { float lp; uint32_t n; int64_t prev; uint32_t gen; };
  static thread_local std::vector<std::vector<LmState>> lmst;
  size_t memo_used = 0;
  auto term = [&](int64_t a, int64_t b) -> float {
    const float SMOOTH = -13.815510557964274f;  // src/search.rs:4
    if (a < 0 || b < 0) return SMOOTH;
    size_t h = ((uint64_t)a * 0x9E3779B97F4A7C15ull ^ (uint64_t)b * 0xC2B2AE3D27D4EB4Full) >> 54;  // 1024 slots
    for (;; h = (h + 1) & 1023) {
      Term& t = memo[h];
      const bool used = t.gen == memo_gen;
      if (used && t.a == a && t.b == b) return t.v;
      if (!used) {
        auto pit = m.unigrams.find((uint64_t)a);
        const uint32_t priorcount = pit == m.unigrams.end() ? 1u : pit->second;
        auto jit = m.bigrams.find(((uint64_t)a << 32) | ((uint64_t)b & 0xFFFFFFFFull));
        float v = SMOOTH;
        if (jit != m.bigrams.end()) v = priorcount < jit->second ? logf((float)jit->second) : logf((float)jit->second / (float)priorcount);
        if (memo_used < 768) { t = Term{a, b, v, memo_gen}; ++memo_used; }  // a full table stops caching, never loops
        return v;
      }
    }
  };
  if (use_lm) {
    if (++memo_gen == 0) {  // wrapped: nothing stale may look current
      for (Term& t : memo) t.gen = 0;
      for (auto& v : lmst) for (LmState& x : v) x.gen = 0;
      memo_gen = 1;
    }
    // the tokens of the boundary text behind a symbol only depend on the boundary: once per boundary, not per symbol
    btok.clear();
    btok_off.assign(nb + 1, 0);
    for (size_t bi = 0; bi < nb; ++bi) {
      btok_off[bi] = (uint32_t)btok.size();
      const Span& nbs = bs[bi];
      if (!(nbs.end - nbs.begin == 1 && text[nbs.begin] == ' ') && nbs.end > nbs.begin) {
        const std::string bt = anx::trim_whitespace(std::string(text + nbs.begin, nbs.end - nbs.begin));
        if (!bt.empty()) {
          auto it = m.encoder.find(bt);
          if (it != m.encoder.end()) for (uint32_t k = m.ngram_off[it->second]; k < m.ngram_off[it->second + 1]; ++k) btok.push_back((int64_t)m.ngram_ids[k]);
          else btok.push_back(-1);
        }
      }
    }
    btok_off[nb] = (uint32_t)btok.size();
    tok.clear();
    tok_off.assign(symbols.size() + 1, 0);
    for (size_t sy = 1; sy < symbols.size(); ++sy) {
      tok_off[sy] = (uint32_t)tok.size();
      const OutSym& o = symbols[sy];
      if (o.vocab_id == 0) tok.push_back(-1);
      else for (uint32_t k = m.ngram_off[o.vocab_id]; k < m.ngram_off[o.vocab_id + 1]; ++k) tok.push_back((int64_t)m.ngram_ids[k]);
      for (uint32_t k = btok_off[o.boundary_index]; k < btok_off[o.boundary_index + 1]; ++k) tok.push_back(btok[k]);
    }
    tok_off[symbols.size()] = (uint32_t)tok.size();
  }
  static thread_local std::vector<std::pair<uint32_t, uint32_t>> chain;
  if (use_lm) {
    if (lmst.size() < nstates) lmst.resize(nstates);
    for (size_t st = 0; st < nstates; ++st)
      if (lmst[st].size() < best[st].size()) lmst[st].resize(best[st].size(), LmState{0.0f, 0u, 0, 0u});
  }
  for (size_t i = 0; i < npaths; ++i) {
    if (use_lm) {
      // the running f32 sum of a path prefix belongs to the lattice node it ends in (every node has ONE parent), so it is
      // computed once per node on the final paths, not once per path: the same additions in the same order
      auto extend = [&](const LmState& from, long sy) {
        LmState r = from;
        if (sy >= 0)
          for (uint32_t k = tok_off[(size_t)sy]; k < tok_off[(size_t)sy + 1]; ++k) { r.lp += term(r.prev, tok[k]); ++r.n; r.prev = tok[k]; }
        r.gen = memo_gen;
        return r;
      };
      // walk up to the first node whose prefix is known, then back down
      chain.clear();
      uint32_t cs = ends[i].ps, cr = ends[i].pr;
      while (cs != UINT32_MAX && lmst[cs][cr].gen != memo_gen) { chain.emplace_back(cs, cr); const Node& nd = best[cs][cr]; cs = nd.ps; cr = nd.pr; }
      LmState cur = cs == UINT32_MAX ? LmState{0.0f, 0u, 0, memo_gen} : lmst[cs][cr];  // the start node: BOS, nothing summed yet
      for (size_t c = chain.size(); c-- > 0;) {
        const Node& nd = best[chain[c].first][chain[c].second];
        cur = nd.ps == UINT32_MAX ? LmState{0.0f, 0u, 0, memo_gen} : extend(cur, nd.sym);
        lmst[chain[c].first][chain[c].second] = cur;
      }
      cur = extend(cur, ends[i].sym);
      const float logprob = cur.lp + term(cur.prev, 1);  // EOS
      const size_t n = (size_t)cur.n + 1;
      ppls[i] = -1.0 / (double)n * (double)logprob;
      if (ppls[i] < best_ppl) best_ppl = ppls[i];
    }
    if (use_rules) {  // src/lib.rs:2345-2363, 2505-2518
      idseq.clear();
      path_syms(i, syms_tmp);
      for (long sy : syms_tmp) {
        const uint64_t id = symbols[(size_t)sy].vocab_id;
        idseq.emplace_back(id, id != 0 && id < m.decoder.size() ? m.decoder[id].lexindex : 0u);
      }
      ctx[i] = m.test_context_rules(idseq, ctx_results[i]);
    }
    if (ends[i].cost < best_cost) best_cost = ends[i].cost;
    if (ctx[i] > best_ctx) best_ctx = ctx[i];
  }
  lat.lap(3);
  const bool shortcut = (!m.have_lm || p.lm_weight == 0.0f) && (!use_rules || p.contextrules_weight == 0.0f);
  double best_score = -99999999.0;
  long best_i = -1;
  for (size_t i = 0; i < npaths; ++i) {
    const double norm_lm = use_lm ? anx::portable_log(best_ppl / ppls[i]) : 0.0;  // (portable_log: the device decoder returns the same bits)
    const double norm_var = anx::portable_log((double)best_cost / (double)ends[i].cost);
    const double norm_ctx = anx::portable_log(ctx[i] / best_ctx);
    double score;
    if (shortcut) score = norm_var;
    else
      score = ((double)p.lm_weight * norm_lm + (double)p.variantmodel_weight * norm_var + (double)p.contextrules_weight * norm_ctx) /
              ((double)p.lm_weight + (double)p.variantmodel_weight + (double)p.contextrules_weight);
    if (score > best_score || best_i < 0) { best_score = score; best_i = (long)i; }
  }
  path_syms((size_t)best_i, syms_tmp);
  const std::vector<long>& best_syms = syms_tmp;
  for (size_t j = 0; j < best_syms.size(); ++j) {
    const OutSym& o = symbols[(size_t)best_syms[j]];
    Span r = matches[o.match_index];
    r.selected = o.variant_index;
    if (use_rules) {
      r.tag0 = (uint32_t)tagpool.size();
      r.tag_stretch = stretch_index;
      for (const anx::PatternMatchResult& pm : ctx_results[(size_t)best_i][j])
        if (pm.tag >= 0) tagpool.emplace_back((uint16_t)pm.tag, pm.seqnr);
      r.ntags = (uint32_t)tagpool.size() - r.tag0;
    }
    out.push_back(r);
  }
  lat.lap(4);
}

// The lattice of one stretch in the flat form lattice.hip decodes (same arcs, same order as most_likely_sequence builds them), written
// straight into the call's arrays (pinned host memory) at the cursors of `K`: a chunk of stretches owns a region of every array, sized
// by upper bounds, so the chunks build side by side and nothing is copied afterwards (the regions' unused tails stay as gaps: a
// stretch addresses its own parts by absolute offsets).  The symbols' (match, variant) go to `refs` at the symbols' positions.
// Returns false when the stretch needs no decoding (no symbols / no final state: the matches pass through, src/lib.rs:2277-2290).
struct SymRef { uint32_t match_index; int32_t variant_index; };
struct LatSink {
  uint32_t* in_off; anx::LatArc* arcs; anx::LatSym* syms; SymRef* refs; uint32_t* btok_off; int32_t* btok;
  size_t in_pos, arc_pos, sym_pos, boff_pos, btok_pos, out_pos;      // cursors (absolute indices)
  size_t in_end, arc_end, sym_end, boff_end, btok_end;               // the region's ends
  bool overflow = false;                                             // an upper bound did not hold (a bug: the call fails)
};
bool build_lattice(const HostModel& m, const char* text, const std::vector<Span>& matches, const Span* bs, size_t nb, size_t end_offset,
                   const anx_search_params& p, bool use_lm, LatSink& K, anx::LatStretch& S) {
  const size_t nstates = nb + 1;
  struct Arc { float cost; uint32_t dst; uint32_t sym; };
  static thread_local std::vector<std::vector<Arc>> arcs;
  if (arcs.size() < nstates) arcs.resize(nstates);
  for (size_t i = 0; i < nstates; ++i) arcs[i].clear();
  static thread_local std::vector<anx::LatSym> syms;
  static thread_local std::vector<SymRef> refs;
  static thread_local std::vector<uint32_t> finals;
  syms.clear(); refs.clear(); finals.clear();
  for (size_t i = 0; i < nb; ++i)
    if (bs[i].begin == end_offset || bs[i].end == end_offset) finals.push_back((uint32_t)i + 1);
  for (size_t mi = 0; mi < matches.size(); ++mi) {
    const Span& mt = matches[mi];
    long prevb = -1, nextb = -1;
    for (size_t i = 0; i < nb; ++i) {
      if (mt.begin == bs[i].end) prevb = (long)i;
      else if (mt.end == bs[i].begin) nextb = (long)i;
    }
    if (nextb < 0) continue;
    const long n = prevb >= 0 ? nextb - prevb : nextb + 1;
    const size_t src = prevb >= 0 ? (size_t)prevb + 1 : 0;
    const uint32_t dst = (uint32_t)nextb + 1;
    if (mt.has_variants && !mt.variants.empty()) {
      for (size_t vi = 0; vi < mt.variants.size(); ++vi) {
        const float cost = (float)n + (1.0f - (float)vr_score(mt.variants[vi], p.base.freq_weight));
        arcs[src].push_back(Arc{cost, dst, (uint32_t)syms.size()});
        syms.push_back(anx::LatSym{(uint32_t)mt.variants[vi].vocab_id, (uint32_t)nextb});
        refs.push_back(SymRef{(uint32_t)mi, (int32_t)vi});
      }
    } else if (n == 1) {
      arcs[src].push_back(Arc{(float)n + 1.0f, dst, (uint32_t)syms.size()});
      syms.push_back(anx::LatSym{0u, (uint32_t)nextb});
      refs.push_back(SymRef{(uint32_t)mi, -1});
    }
  }
  if (syms.empty() || finals.empty()) return false;
  for (size_t i = 0; i < nb; ++i) arcs[i].push_back(Arc{100.0f, (uint32_t)i + 1, 0xFFFFFFFFu});  // failsafe epsilon transitions
  // incoming arcs per state in (source state, arc number) order, then the virtual end state behind the finals
  static thread_local std::vector<uint32_t> indeg;
  indeg.assign(nstates + 2, 0u);
  for (size_t sidx = 0; sidx < nstates; ++sidx)
    for (const Arc& a : arcs[sidx]) ++indeg[a.dst + 1];
  indeg[nstates + 1] = (uint32_t)finals.size();
  for (size_t d = 1; d <= nstates + 1; ++d) indeg[d] += indeg[d - 1];
  const size_t narcs = indeg[nstates + 1];
  if (K.in_pos + nstates + 2 > K.in_end || K.arc_pos + narcs > K.arc_end || K.sym_pos + syms.size() > K.sym_end ||
      K.boff_pos + nb + 1 > K.boff_end) { K.overflow = true; return false; }
  S.nstates = (uint32_t)nstates;
  S.in_off0 = (uint32_t)K.in_pos;
  S.arc0 = (uint32_t)K.arc_pos;
  S.sym0 = (uint32_t)K.sym_pos;
  S.btok_off0 = (uint32_t)K.boff_pos;
  S.btok0 = (uint32_t)K.btok_pos;
  S.out0 = (uint32_t)K.out_pos;
  S.best_cost_init = (float)(nb - 1) * 2.0f;
  S.node0 = 0;
  for (size_t d = 0; d <= nstates + 1; ++d) K.in_off[K.in_pos + d] = indeg[d];  // [d] = first incoming arc of state d; nstates + 2 entries
  K.in_pos += nstates + 2;
  anx::LatArc* out_arcs = K.arcs + K.arc_pos;
  static thread_local std::vector<uint32_t> cur;
  cur.assign(indeg.begin(), indeg.end());
  uint32_t span = 1;
  for (size_t sidx = 0; sidx < nstates; ++sidx)
    for (const Arc& a : arcs[sidx]) {
      out_arcs[cur[a.dst]++] = anx::LatArc{a.cost, (uint32_t)sidx, a.sym};
      span = std::max(span, a.dst - (uint32_t)sidx);
    }
  for (uint32_t f : finals) {
    out_arcs[cur[nstates]++] = anx::LatArc{0.0f, f, 0xFFFFFFFFu};
    span = std::max(span, (uint32_t)nstates - f);
  }
  K.arc_pos += narcs;
  S.ring = span + 1;
  memcpy(K.syms + K.sym_pos, syms.data(), syms.size() * sizeof(anx::LatSym));
  memcpy(K.refs + K.sym_pos, refs.data(), refs.size() * sizeof(SymRef));
  K.sym_pos += syms.size();
  // LM tokens of the boundary text behind a symbol (src/lib.rs:2606-2629): once per boundary
  const size_t btok0 = K.btok_pos;
  for (size_t bi = 0; bi < nb; ++bi) {
    K.btok_off[K.boff_pos++] = (uint32_t)(K.btok_pos - btok0);
    if (!use_lm) continue;
    const Span& nbs = bs[bi];
    if (!(nbs.end - nbs.begin == 1 && text[nbs.begin] == ' ') && nbs.end > nbs.begin) {
      const std::string bt = anx::trim_whitespace(std::string(text + nbs.begin, nbs.end - nbs.begin));
      if (!bt.empty()) {
        auto it = m.encoder.find(bt);
        const size_t nt = it != m.encoder.end() ? m.ngram_off[it->second + 1] - m.ngram_off[it->second] : 1;
        if (K.btok_pos + nt > K.btok_end) { K.overflow = true; return false; }
        if (it != m.encoder.end()) for (uint32_t k = m.ngram_off[it->second]; k < m.ngram_off[it->second + 1]; ++k) K.btok[K.btok_pos++] = (int32_t)m.ngram_ids[k];
        else K.btok[K.btok_pos++] = -1;
      }
    }
  }
  K.btok_off[K.boff_pos++] = (uint32_t)(K.btok_pos - btok0);
  K.out_pos += nstates;  // a path has at most one symbol per state it enters
  return true;
}

struct Stretch {  // one hard-boundary "batch" of the reference (src/lib.rs:1821-1940)
  size_t text_index, begin, end, b0, b1;  // boundaries [b0, b1)
  std::vector<Span> matches;
};

}  // namespace

static void search_out_cache_trim();   // (the cache of output blocks, below)
extern "C" {
// joins the library's host threads (the pool of search mode) and releases the cached output blocks; see include/anx.h
void anx_shutdown(void) { HostPool::shutdown(); search_out_cache_trim(); }
}
// the library is unloaded (dlclose) or the process exits through exit(): no thread of ours may be left running code that is about to be unmapped
__attribute__((destructor)) static void anx_on_unload() { HostPool::shutdown(); }

extern "C" {

void anx_default_search_params(anx_search_params* p) {  // src/types.rs:170-192
  anx_default_params(&p->base);
  p->max_ngram = 3;
  p->max_seq = 250;
  p->lm_weight = 1.0f;
  p->variantmodel_weight = 3.0f;
  p->contextrules_weight = 1.0f;
  p->unicodeoffsets = 0;
}

const char* anx_last_error(void);
void anx_matches_free(anx_match* matches, size_t* offsets, anx_result* rows, anx_match_tag* tags);
// work(lo, hi) over [0, count) in chunks handed to the host threads
static void pool_for(size_t count, size_t chunk, size_t serial_below, const std::function<void(size_t, size_t)>& work) {
  HostPool& pool = HostPool::get();
  const unsigned hw = pool.width();
  if (count < serial_below || hw == 1) { work(0, count); return; }
  std::atomic<size_t> next{0};
  const size_t nchunks = (count + chunk - 1) / chunk;
  pool.run((unsigned)std::min<size_t>(hw - 1, nchunks - 1), [&]() {
    for (;;) {
      const size_t lo = next.fetch_add(chunk);
      if (lo >= count) break;
      work(lo, std::min(count, lo + chunk));
    }
  });
}
// The call's big output arrays are fresh memory every time (the caller frees them): with 4 KB pages writing 250 MB of them means
// 60 k page faults inside the output loops.  Where transparent huge pages are available on request (the usual `madvise` setting)
// the 2 MB-aligned inside of such an array asks for them.
static void advise_huge(void* p, size_t bytes) {
#ifdef MADV_HUGEPAGE
  const uintptr_t H = (uintptr_t)2 << 20;
  const uintptr_t a = ((uintptr_t)p + H - 1) & ~(H - 1), e = ((uintptr_t)p + bytes) & ~(H - 1);
  if (bytes >= 4 * H && e > a) (void)madvise(reinterpret_cast<void*>(a), e - a, MADV_HUGEPAGE);
#else
  (void)p; (void)bytes;
#endif
}
// Round 6: the big output arrays of a call (matches, variant rows: 190 MB for 12.5 MB of text) come from a small cache of blocks that
// anx_matches_free hands back, instead of fresh memory every time.  Fresh memory means page faults and zeroing inside the output loops
// (with transparent huge pages: 2 MB pieces that the kernel has to find contiguous), and in a long-lived process that had churned through
// a hundred GB of host memory two calls of a series of five took 80 instead of 41-50 ms (bench.py's search line, no cgroup throttling;
// a fresh process: 37-46 ms).  Blocks below 1 MB are plain malloc blocks; the cache keeps at most ANX_SEARCH_OUT_CACHE_MB (default 1024).
extern "C++" {
namespace {
struct OutCache {
  std::mutex mu;
  std::unordered_map<void*, size_t> live;   // blocks handed out (capacity in bytes)
  std::multimap<size_t, void*> idle;        // blocks that came back
  size_t idle_bytes = 0;
};
OutCache& out_cache() { static OutCache* c = new OutCache(); return *c; }   // (never destroyed: anx_matches_free may run at exit)
size_t out_cache_limit() {
  static const size_t lim = []() { const char* e = getenv("ANX_SEARCH_OUT_CACHE_MB"); return (size_t)(e ? strtoull(e, nullptr, 10) : 1024ull) << 20; }();
  return lim;
}
void* out_alloc(size_t bytes) {
  bytes = std::max<size_t>(bytes, 1);
  if (bytes < ((size_t)1 << 20) || out_cache_limit() == 0) return malloc(bytes);
  OutCache& c = out_cache();
  {
    std::lock_guard<std::mutex> g(c.mu);
    auto it = c.idle.lower_bound(bytes);
    if (it != c.idle.end() && it->first <= 2 * bytes) {
      void* p = it->second;
      c.idle_bytes -= it->first;
      c.live[p] = it->first;
      c.idle.erase(it);
      return p;
    }
  }
  void* p = malloc(bytes);
  if (!p) return nullptr;
  advise_huge(p, bytes);
  std::lock_guard<std::mutex> g(c.mu);
  c.live[p] = bytes;
  return p;
}
// true: p was a cached block (kept or released); false: not one of ours, the caller frees it
bool out_release(void* p) {
  if (!p) return true;
  OutCache& c = out_cache();
  size_t cap = 0;
  {
    std::lock_guard<std::mutex> g(c.mu);
    auto it = c.live.find(p);
    if (it == c.live.end()) return false;
    cap = it->second;
    c.live.erase(it);
    if (c.idle_bytes + cap <= out_cache_limit()) {
      c.idle.emplace(cap, p);
      c.idle_bytes += cap;
      return true;
    }
  }
  free(p);
  return true;
}
void out_cache_trim() {
  OutCache& c = out_cache();
  std::vector<void*> drop;
  {
    std::lock_guard<std::mutex> g(c.mu);
    for (auto& kv : c.idle) drop.push_back(kv.second);
    c.idle.clear();
    c.idle_bytes = 0;
  }
  for (void* p : drop) free(p);
}
}  // namespace
static void search_out_cache_trim() { out_cache_trim(); }
}  // extern "C++"

// What the pipeline leaves for a part of a call: the matches of every text (their variants are views of the kept result arrays of
// the n-gram orders), the tags, and the sizes of the part's share of the output arrays.
struct OrderRows { anx_result* rows; size_t* offs; };
struct PartOut {
  std::vector<std::vector<Span>> per_text;
  std::vector<TagPool> tagpools;
  std::vector<OrderRows> kept;
  std::vector<size_t> text_rows, text_tags;  // per text: variant rows / tags of its matches (summed where per_text is built, by the pool)
  size_t total = 0, total_rows = 0, total_tags = 0;
  int rc = ANX_OK;
  std::string err;
  // progress, for the call's thread that writes the output while later parts are still on the device (anx_find_all_matches_batch):
  // an upper bound of the part's matches once its boundaries are known (a path has at most one match per token), and "done"
  std::atomic<size_t> ub_matches{(size_t)-1};
  std::atomic<int> done{0};
  std::mutex* sig_mu = nullptr;
  std::condition_variable* sig_cv = nullptr;
  void signal() { if (sig_mu) { { std::lock_guard<std::mutex> g(*sig_mu); } sig_cv->notify_all(); } }
  void free_kept() { for (OrderRows& o : kept) anx_results_free(o.rows, o.offs); kept.clear(); }
  ~PartOut() { free_kept(); }
};
static int find_all_part(const anx_model* model, const char* const* texts, size_t n, const anx_search_params* sp, PartOut& po);
// The part's matches, rows and tags into the call's arrays: om / orows / otags at the part's bases, oo[t + 1] for its texts (oo points
// at the part's first text; oo[0] is the previous part's business).
static void write_part(const PartOut& po, const char* const* texts, size_t n, const anx_search_params* sp, anx_match* om, size_t* oo,
                       anx_result* orows, anx_match_tag* otags, size_t m_base, size_t r_base, size_t t_base) {
  const std::vector<std::vector<Span>>& per_text = po.per_text;
  // first match / row / tag of every text, then the texts are written side by side
  std::vector<size_t> m0(n + 1, m_base), row0(n + 1, r_base), tag0(n + 1, t_base);
  for (size_t t = 0; t < n; ++t) {  // (a pass over the 1.3 M spans of 12 MB of text here, by one thread, was a third of the output phase)
    m0[t + 1] = m0[t] + per_text[t].size();
    row0[t + 1] = row0[t] + po.text_rows[t];
    tag0[t + 1] = tag0[t] + po.text_tags[t];
    oo[t + 1] = m0[t + 1];
  }
  pool_for(n, 8, 64, [&](size_t lo, size_t hi) {
    for (size_t t = lo; t < hi; ++t) {
      size_t w = m0[t], rw = row0[t], tw = tag0[t];
      std::vector<size_t> cpmap;  // byte offset -> code point index (remap_offsets_to_unicodepoints, src/search.rs:527-546)
      if (sp->unicodeoffsets && texts[t]) {
        const size_t len = strlen(texts[t]);
        cpmap.assign(len + 1, 0);
        size_t cp = 0;
        for (size_t i = 0; i < len;) {
          int l;
          anx::utf8_decode_at(texts[t] + i, len - i, &l);
          for (int k = 0; k < l && i + (size_t)k < len; ++k) cpmap[i + (size_t)k] = cp;
          i += (size_t)l;
          ++cp;
        }
        cpmap[len] = cp;
      }
      for (const Span& s : per_text[t]) {
        anx_match& o = om[w++];
        o.begin = cpmap.empty() ? s.begin : cpmap[s.begin];
        o.end = cpmap.empty() ? s.end : cpmap[s.end];
        o.n = s.n;
        o.selected = (s.has_variants && !s.variants.empty()) ? s.selected : -1;
        o.var_begin = rw;
        for (const anx_result& r : s.variants) orows[rw++] = r;
        o.var_end = rw;
        o.tag_begin = otags ? (uint32_t)tw : 0u;  // (no tag array asked for: empty ranges)
        if (otags)
          for (uint32_t k = 0; k < s.ntags; ++k) { const auto& tg = po.tagpools[s.tag_stretch][s.tag0 + k]; otags[tw++] = anx_match_tag{tg.first, tg.second, 0}; }
        o.tag_end = otags ? (uint32_t)tw : 0u;
      }
    }
  });
}

// Texts are independent of each other (src/lib.rs:1790-1957 works text by text); a call only batches them for the device.  A large
// call is cut into contiguous parts of ~4 MB of text (ANX_SEARCH_PART_BYTES) and ANX_SEARCH_PARTS (4) of them are in flight at a
// time, each the whole pipeline in a thread of its own: the phases of the pipeline alternate between the host threads and the
// device, so one part's device batches and lattices run under the others' host phases (a single pass leaves the device idle
// for two thirds of the call and the host threads for the rest).
// When all parts are done their sizes are known: the call's arrays are allocated once and every part writes its matches, rows
// and tags at its base (write_part).
// diagnosis (anx_debug_search_stats): multi-part calls, and how many of them wrote their output while later parts were still at work /
// had to fall back to writing it at the end although early output was eligible
static std::atomic<uint64_t> g_search_multi_calls{0}, g_search_early_kept{0}, g_search_early_dropped{0};
int anx_debug_search_stats(uint64_t* out) { if (!out) return ANX_EINVAL; out[0] = g_search_multi_calls.load(); out[1] = g_search_early_kept.load(); out[2] = g_search_early_dropped.load(); out[3] = 0; return ANX_OK; }
int anx_find_all_matches_batch(const anx_model* model, const char* const* texts, size_t n, const anx_search_params* sp,
                               anx_match** out_matches, size_t** out_offsets, anx_result** out_rows, size_t* out_n_rows,
                               anx_match_tag** out_tags) {
  if (out_tags) *out_tags = nullptr;
  if (!model || (!texts && n) || !sp || !out_matches || !out_offsets || !out_rows || !out_n_rows)
    return anx_fail(ANX_EINVAL, "NULL argument");
  const size_t workers = std::min((size_t)std::max(1, anx::switches().search_parts), n);
  std::vector<size_t> len_upto;
  size_t parts = 1;
  if (workers > 1) {  // worth it from a few megabytes on (ANX_SEARCH_PARTS_MIN)
    len_upto.assign(n + 1, 0);
    for (size_t t = 0; t < n; ++t) len_upto[t + 1] = len_upto[t] + (texts[t] ? strlen(texts[t]) : 0);
    if (len_upto[n] >= (size_t)anx::switches().search_parts_min) {
      // parts of ~ANX_SEARCH_PART_BYTES (a single pass over tens of megabytes is slower per byte: its arrays outgrow the caches and
      // the allocator), a whole number of rounds of the workers
      const size_t per_round = workers * (size_t)std::max(1l, anx::switches().search_part_bytes);
      parts = std::min(n, workers * ((len_upto[n] + per_round - 1) / per_round));
    }
  }
  parts = std::max<size_t>(parts, 1);
  std::vector<size_t> cut(parts + 1, n);
  cut[0] = 0;
  {
    // part r ends where the text reaches its share: even shares, except that the first part may be smaller (ANX_SEARCH_FIRST_PCT)
    const double first = parts > 1 ? (double)anx::switches().search_first_pct / 100.0 : 1.0;
    const double total_w = first + (double)(parts - 1);
    for (size_t r = 1, t = 0; r < parts; ++r) {
      const double share = (first + (double)(r - 1)) / total_w;
      while (t < n && (double)len_upto[t] < (double)len_upto[n] * share) ++t;
      cut[r] = t;
    }
  }
  std::vector<std::unique_ptr<PartOut>> P(parts);
  for (auto& p : P) p.reset(new PartOut());
  struct Early { bool on = false; anx_match* om = nullptr; size_t* oo = nullptr; anx_result* orows = nullptr; size_t m_cap = 0, r_cap = 0, m_total = 0, r_total = 0, parts_written = 0; double write_s = 0.0; } early;
  if (parts == 1) {
    P[0]->rc = find_all_part(model, texts, n, sp, *P[0]);
    if (P[0]->rc != ANX_OK) return P[0]->rc;  // (the message is this thread's)
  } else {
    std::atomic<size_t> next_part{0};
    std::atomic<bool> failed{false};
    std::mutex sig_mu;
    std::condition_variable sig_cv;
    for (auto& p : P) { p->sig_mu = &sig_mu; p->sig_cv = &sig_cv; }
    std::vector<std::thread> th;
    for (size_t w = 0; w < workers; ++w)
      th.emplace_back([&]() {
        for (;;) {
          const size_t r = next_part.fetch_add(1);
          if (r >= parts) break;
          PartOut& p = *P[r];
          if (!failed.load()) {
            tl_pool_prio = anx::switches().search_prio ? (int)r + 1 : 0;   // (the call's own thread, which writes the output, keeps 0)
            const auto tp = std::chrono::steady_clock::now();
            p.rc = find_all_part(model, texts + cut[r], cut[r + 1] - cut[r], sp, p);
            if (anx::switches().search_timing) fprintf(stderr, "[anx search] part %zu, all of it          %8.2f ms\n", r, std::chrono::duration<double>(std::chrono::steady_clock::now() - tp).count() * 1e3);
            if (p.rc != ANX_OK) { p.err = anx_last_error(); failed.store(true); }  // (the message is the worker thread's)
          } else p.rc = ANX_EINVAL;
          {  // a part that failed before its boundaries: nobody waits for its bound.  Only if the part never published one: a finished
             // part keeps its real bound (calls with more parts than workers sum the bounds after the first round of parts is done)
            size_t unset = (size_t)-1;
            p.ub_matches.compare_exchange_strong(unset, 0, std::memory_order_acq_rel);
          }
          p.done.store(1, std::memory_order_release);
          p.signal();
        }
      });
    // The output while the later parts are still at work (round 5: the middle of a call is bound by the device, the host threads wait;
    // written at the end, the 256 MB of a 12.5 MB call were 6-7 ms with the device idle).  The arrays are sized by upper bounds
    // -- matches: one per token, known after every part's boundaries; rows: max_matches + 2 per match -- and cut back to what was
    // written at the end.  Not with unlimited lists (max_matches = 0), tags, or when a bound does not hold: those write at the end.
    early.on = anx::switches().search_early_output && sp->base.max_matches > 0 && !out_tags;
    if (early.on) {
      size_t ub = 0;
      {
        std::unique_lock<std::mutex> l(sig_mu);
        sig_cv.wait(l, [&]() { for (auto& p : P) if (p->ub_matches.load(std::memory_order_acquire) == (size_t)-1) return false; return true; });
      }
      for (auto& p : P) ub += p->ub_matches.load(std::memory_order_acquire);
      early.m_cap = std::max<size_t>(1, ub);
      early.r_cap = std::max<size_t>(1, ub * ((size_t)sp->base.max_matches + 2));
      early.om = static_cast<anx_match*>(out_alloc(early.m_cap * sizeof(anx_match)));
      early.oo = static_cast<size_t*>(calloc(n + 1, sizeof(size_t)));
      early.orows = static_cast<anx_result*>(out_alloc(early.r_cap * sizeof(anx_result)));
      if (!early.om || !early.oo || !early.orows) { if (!out_release(early.om)) free(early.om); free(early.oo); if (!out_release(early.orows)) free(early.orows); early = Early(); }
    }
    if (early.on) {
      size_t mb = 0, rb = 0;
      for (size_t r = 0; r < parts && early.on; ++r) {
        {
          std::unique_lock<std::mutex> l(sig_mu);
          sig_cv.wait(l, [&]() { return P[r]->done.load(std::memory_order_acquire) != 0; });
        }
        if (P[r]->rc != ANX_OK) break;  // (reported below)
        if (mb + P[r]->total > early.m_cap || rb + P[r]->total_rows > early.r_cap) { early.on = false; break; }  // a bound did not hold: at the end, exactly
        const auto tw0 = std::chrono::steady_clock::now();
        write_part(*P[r], texts + cut[r], cut[r + 1] - cut[r], sp, early.om, early.oo + cut[r], early.orows, nullptr, mb, rb, 0);
        early.write_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count();
        mb += P[r]->total;
        rb += P[r]->total_rows;
        early.parts_written = r + 1;
      }
      early.m_total = mb;
      early.r_total = rb;
    }
    for (auto& x : th) x.join();
    g_search_multi_calls.fetch_add(1, std::memory_order_relaxed);
    if (early.on && early.om && early.parts_written == parts) g_search_early_kept.fetch_add(1, std::memory_order_relaxed);
    else if (anx::switches().search_early_output && sp->base.max_matches > 0 && !out_tags) g_search_early_dropped.fetch_add(1, std::memory_order_relaxed);
    if (early.om && (!early.on || early.parts_written != parts)) { if (!out_release(early.om)) free(early.om); free(early.oo); if (!out_release(early.orows)) free(early.orows); early = Early(); }
    for (auto& p : P)
      if (p->rc != ANX_OK && !p->err.empty()) return anx_fail(p->rc, p->err);
    for (auto& p : P)
      if (p->rc != ANX_OK) return anx_fail(p->rc, "a part of the call failed");
  }
  // ONE set of arrays for the call, written by every part at its base (the parts used to build arrays of their own that were
  // copied here: twice the fresh pages, and 38 ms of copying per 12.5 MB of text)
  const bool timing = anx::switches().search_timing != 0;
  const auto t_out = std::chrono::steady_clock::now();
  if (early.on && early.om && early.parts_written == parts) {  // everything is written (the arrays keep their capacity: cached blocks)
    anx_match* om2 = early.om;
    anx_result* or2 = early.orows;
    for (size_t r = 0; r < parts; ++r) P[r]->free_kept();
    auto garbage = std::make_shared<std::vector<std::unique_ptr<PartOut>>>(std::move(P));
    HostPool::get().post([garbage]() { garbage->clear(); });
    if (timing) fprintf(stderr, "[anx search] output (%zu parts), written as the parts finished: %.2f ms of writing, %.2f ms behind the last part\n", parts, early.write_s * 1e3,
                        std::chrono::duration<double>(std::chrono::steady_clock::now() - t_out).count() * 1e3);
    *out_matches = om2 ? om2 : early.om;
    *out_offsets = early.oo;
    *out_rows = or2 ? or2 : early.orows;
    *out_n_rows = early.r_total;
    return ANX_OK;
  }
  std::vector<size_t> m0(parts + 1, 0), r0(parts + 1, 0), t0(parts + 1, 0);
  for (size_t r = 0; r < parts; ++r) {
    m0[r + 1] = m0[r] + P[r]->total;
    r0[r + 1] = r0[r] + P[r]->total_rows;
    t0[r + 1] = t0[r] + P[r]->total_tags;
  }
  anx_match* om = static_cast<anx_match*>(out_alloc(std::max<size_t>(1, m0[parts]) * sizeof(anx_match)));
  size_t* oo = static_cast<size_t*>(calloc(n + 1, sizeof(size_t)));
  anx_result* orows = static_cast<anx_result*>(out_alloc(std::max<size_t>(1, r0[parts]) * sizeof(anx_result)));
  anx_match_tag* otags = out_tags ? static_cast<anx_match_tag*>(malloc(std::max<size_t>(1, t0[parts]) * sizeof(anx_match_tag))) : nullptr;
  if (!om || !oo || !orows || (out_tags && !otags)) { if (!out_release(om)) free(om); free(oo); if (!out_release(orows)) free(orows); free(otags); return anx_fail(ANX_EINVAL, "out of memory"); }
  double t_write = 0.0, t_reset = 0.0;
  {
    const auto ta = std::chrono::steady_clock::now();
    for (size_t r = 0; r < parts; ++r) write_part(*P[r], texts + cut[r], cut[r + 1] - cut[r], sp, om, oo + cut[r], orows, otags, m0[r], r0[r], t0[r]);
    const auto tb = std::chrono::steady_clock::now();
    // the kept result arrays go back to the pinned cache now (the next call's fetches find them there); the parts' own data (the
    // spans of every text: 4-5 ms of frees, slower still from several threads at once) is released by a pool thread after the return
    for (size_t r = 0; r < parts; ++r) P[r]->free_kept();
    auto garbage = std::make_shared<std::vector<std::unique_ptr<PartOut>>>(std::move(P));
    HostPool::get().post([garbage]() { garbage->clear(); });
    t_write = std::chrono::duration<double>(tb - ta).count();
    t_reset = std::chrono::duration<double>(std::chrono::steady_clock::now() - tb).count();
  }
  if (timing) { uint64_t h, mi, mb; anx::host_result_cache_stats(&h, &mi, &mb); fprintf(stderr, "[anx search]   pinned result cache since load: %llu hits, %llu misses (%.0f MB pinned afresh)\n", (unsigned long long)h, (unsigned long long)mi, (double)mb / 1048576.0); }
  if (timing) fprintf(stderr, "[anx search]   output: writing %.2f ms, releasing the parts %.2f ms\n", t_write * 1e3, t_reset * 1e3);
  if (timing) fprintf(stderr, "[anx search] output (%zu parts)            %8.2f ms\n", parts, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_out).count() * 1e3);
  *out_matches = om;
  *out_offsets = oo;
  *out_rows = orows;
  *out_n_rows = r0[parts];
  if (out_tags) *out_tags = otags;
  return ANX_OK;
}


// ---- the one-device-pass form of a part (round 5; lattice.hip search_onepass_*) ------------------------------------------------------
// The host generates the segments of ALL n-gram orders at once (whether a higher-order segment is looked up no longer waits for
// the unigrams' results on the host: redundant_match, src/search.rs:317-336, is decided on the device from the unigrams' rows),
// describes the lattice structure that does not depend on results (states, arc order) and lets the device build and decode the
// lattices; only the matches of the chosen paths come back, with their rows.  Fills decoded[] / done[] like the classic path and
// leaves the rows in po.kept.  *fallback: the part needs the classic path (a lattice beyond the device decoder's limits, a stretch
// with matches but no lattice): nothing was changed.
static int onepass_part(const anx_model* model, const HostModel& m, const char* const* texts, const anx_search_params* sp, std::vector<std::vector<Span>>& bounds,
                        std::vector<Stretch>& stretches, std::vector<std::vector<Span>>& decoded, std::vector<uint8_t>& done, PartOut& po, void* stream, bool* fallback,
                        const std::function<void(const char*)>& lap) {
  *fallback = false;
  const size_t NS = stretches.size();
  const bool use_lm = m.have_lm && sp->lm_weight > 0.0f;
  const uint32_t max_ngram = sp->max_ngram;
  const size_t SC = 64, nsc = (NS + SC - 1) / SC;
  struct MRec { uint32_t pack, u0, u1; };   // u0 / u1 local to the stretch
  struct Chunk {
    std::vector<MRec> rec;                // per match of the chunk's stretches, in order
    std::vector<int32_t> btok;            // LM tokens of the boundaries
    std::vector<uint32_t> btok_off;       // per (stretch, boundary) + 1 per stretch, relative to the stretch's first token
    bool odd = false;                     // a stretch with matches but no lattice
  };
  std::vector<Chunk> chunks(nsc);
  std::vector<uint32_t> st_nu(NS + 1, 0), st_nm(NS + 1, 0);       // unigram matches / matches per stretch (prefix sums below)
  std::vector<size_t> st_bu(NS + 1, 0), st_bh(NS + 1, 0);         // arena bytes per stretch
  std::vector<uint8_t> has_lat(NS, 0);
  std::vector<uint32_t> st_nfin(NS, 0), st_ntok(NS, 0);           // final states / LM tokens per stretch
  auto state_of_end = [](const Span* bs, size_t nb, size_t off) -> long {  // last boundary whose END is off (build_lattice: prevb)
    size_t lo = 0, hi = nb;
    while (lo < hi) { const size_t mid = (lo + hi) / 2; if (bs[mid].end < off) lo = mid + 1; else hi = mid; }
    long r = -1;
    for (size_t i = lo; i < nb && bs[i].end == off; ++i) r = (long)i;
    return r;
  };
  auto state_of_begin = [](const Span* bs, size_t nb, size_t off, size_t mbegin) -> long {  // last boundary whose BEGIN is off and whose end is not the match's begin (nextb)
    size_t lo = 0, hi = nb;
    while (lo < hi) { const size_t mid = (lo + hi) / 2; if (bs[mid].begin < off) lo = mid + 1; else hi = mid; }
    long r = -1;
    for (size_t i = lo; i < nb && bs[i].begin == off; ++i) if (bs[i].end != mbegin) r = (long)i;
    return r;
  };
  pool_for(nsc, 4, 8, [&](size_t clo, size_t chi) {
    std::vector<uint32_t> sd;  // states of the segments as find_match_ngrams knows them
    for (size_t c = clo; c < chi; ++c) {
      Chunk& C = chunks[c];
      const size_t s_lo = c * SC, s_hi = std::min(NS, s_lo + SC);
      for (size_t si = s_lo; si < s_hi; ++si) {
        Stretch& st = stretches[si];
        const char* text = texts[st.text_index];
        const Span* bs = bounds[st.text_index].data() + st.b0;
        const size_t nb = st.b1 - st.b0;
        std::vector<Span>& mv = st.matches;
        mv.clear();
        mv.reserve((nb + 1) * max_ngram);
        sd.clear();
        for (uint32_t order = 1; order <= max_ngram; ++order) {
          find_match_ngrams(text, bs, nb, order, st.begin, st.end, mv, &sd);
          if (order == 1) st_nu[si + 1] = (uint32_t)mv.size();
        }
        st_nm[si + 1] = (uint32_t)mv.size();
        const uint32_t nu = st_nu[si + 1];
        bool any_sym = false;
        uint32_t nfin = 0;
        for (size_t i = 0; i < nb; ++i) nfin += (bs[i].begin == st.end || bs[i].end == st.end) ? 1u : 0u;
        st_nfin[si] = nfin;
        const bool any_final = nfin != 0;
        uint32_t ulo = 0;
        for (size_t mi = 0; mi < mv.size(); ++mi) {
          const Span& mt = mv[mi];
          MRec r{0u, 0u, 0u};
          bool has_dst = true;
          if (sd[mi] != 0xFFFFFFFFu) r.pack = sd[mi] | (mt.n << 24);
          else {  // build_lattice's own search over the boundaries' offsets
            const long prevb = state_of_end(bs, nb, mt.begin), nextb = state_of_begin(bs, nb, mt.end, mt.begin);
            const uint32_t src = prevb >= 0 ? (uint32_t)prevb + 1u : 0u;
            has_dst = nextb >= 0;
            if (!has_dst) r.pack = src | (mt.n << 24) | 0x80000000u;
            else r.pack = src | (((uint32_t)nextb + 1u) << 12) | (mt.n << 24);
          }
          if (mt.n == 1) { any_sym = any_sym || has_dst; st_bu[si + 1] += mt.end - mt.begin + 1; }
          else {
            if (mi == nu || mv[mi - 1].n != mt.n) ulo = 0;  // a new order: its segments ascend again
            while (ulo < nu && mv[ulo].begin < mt.begin) ++ulo;
            uint32_t uhi = ulo;
            while (uhi < nu && mv[uhi].end <= mt.end) ++uhi;
            r.u0 = ulo; r.u1 = uhi;
            st_bh[si + 1] += mt.end - mt.begin + 1;
          }
          C.rec.push_back(r);
        }
        if (nb + 1 > 0xFFEu) C.odd = true;  // states beyond the packed 12 bits (the device decoder's limit is lower still)
        has_lat[si] = any_sym && any_final;
        if (!has_lat[si] && !mv.empty()) C.odd = true;
        // LM tokens of the boundary texts (src/lib.rs:2606-2629), as build_lattice lays them out
        if (has_lat[si]) {
          const size_t t0 = C.btok.size();
          for (size_t bi = 0; bi < nb; ++bi) {
            C.btok_off.push_back((uint32_t)(C.btok.size() - t0));
            if (!use_lm) continue;
            const Span& nbs = bs[bi];
            if (!(nbs.end - nbs.begin == 1 && text[nbs.begin] == ' ') && nbs.end > nbs.begin) {
              const std::string bt = anx::trim_whitespace(std::string(text + nbs.begin, nbs.end - nbs.begin));
              if (!bt.empty()) {
                auto it = m.encoder.find(bt);
                if (it != m.encoder.end()) for (uint32_t k = m.ngram_off[it->second]; k < m.ngram_off[it->second + 1]; ++k) C.btok.push_back((int32_t)m.ngram_ids[k]);
                else C.btok.push_back(-1);
              }
            }
          }
          C.btok_off.push_back((uint32_t)(C.btok.size() - t0));
          st_ntok[si] = (uint32_t)(C.btok.size() - t0);
        }
      }
    }
  });
  for (const Chunk& C : chunks) if (C.odd) { *fallback = true; return ANX_OK; }
  // prefix sums: matches, unigram / higher-order queries and arena bytes per stretch
  std::vector<uint32_t> m0(NS + 1, 0), qu0(NS + 1, 0), qh0(NS + 1, 0);
  for (size_t si = 0; si < NS; ++si) {
    m0[si + 1] = m0[si] + st_nm[si + 1];
    qu0[si + 1] = qu0[si] + st_nu[si + 1];
    qh0[si + 1] = qh0[si] + (st_nm[si + 1] - st_nu[si + 1]);
    st_bu[si + 1] += st_bu[si];
    st_bh[si + 1] += st_bh[si];
  }
  const size_t M = m0[NS], NU = qu0[NS], NH = qh0[NS];
  if (M >= (1u << 30) || st_bu[NS] >= ((size_t)1 << 32) || st_bh[NS] >= ((size_t)1 << 32) || NU > ((size_t)4 << 20) || NH > ((size_t)4 << 20)) { *fallback = true; return ANX_OK; }
  // lattices: the stretches that have one, in stretch order; in_off entries, out slots, LM token offsets
  std::vector<uint32_t> lat_of_st(NS, 0xFFFFFFFFu), st_of_lat;
  for (size_t si = 0; si < NS; ++si) if (has_lat[si]) { lat_of_st[si] = (uint32_t)st_of_lat.size(); st_of_lat.push_back((uint32_t)si); }
  const size_t NL = st_of_lat.size();
  std::vector<anx::LatStretch> lst(NL);
  std::vector<uint32_t> st_m0(NL), st_e0(NL), maxdeg(NL, 0), g0_of_lat(NL + 1, 0);
  size_t nin = 0, nout = 0, nboff = 0, nbtok = 0;
  {
    std::vector<size_t> ch_btok0(nsc + 1, 0), ch_boff0(nsc + 1, 0);
    for (size_t c = 0; c < nsc; ++c) { ch_btok0[c + 1] = ch_btok0[c] + chunks[c].btok.size(); ch_boff0[c + 1] = ch_boff0[c] + chunks[c].btok_off.size(); }
    nbtok = ch_btok0[nsc]; nboff = ch_boff0[nsc];
    size_t li = 0;
    for (size_t c = 0; c < nsc; ++c) {
      size_t boff = ch_boff0[c];
      size_t tok_before = 0;  // tokens of the chunk's earlier lattices
      for (size_t si = c * SC; si < std::min(NS, (c + 1) * SC); ++si) {
        if (!has_lat[si]) continue;
        const size_t nb = stretches[si].b1 - stretches[si].b0;
        anx::LatStretch& S = lst[li];
        S.nstates = (uint32_t)nb + 1u;
        S.in_off0 = (uint32_t)nin;
        S.arc0 = 0; S.sym0 = 0;
        S.btok_off0 = (uint32_t)boff;
        S.btok0 = (uint32_t)(ch_btok0[c] + tok_before);
        tok_before += st_ntok[si];
        S.out0 = (uint32_t)nout;
        S.best_cost_init = (float)(nb - 1) * 2.0f;
        S.ring = 2;
        S.node0 = 0;
        st_m0[li] = m0[si];
        st_e0[li] = (uint32_t)nin;
        nin += nb + 3;
        nout += nb + 1;
        boff += nb + 1;
        ++li;
      }
    }
  }
  if (nin >= ((size_t)1 << 32) || nout >= ((size_t)1 << 32)) { *fallback = true; return ANX_OK; }
  // per-match tables, arenas, arc groups (parallel over the chunks)
  // (the tables and the arenas live in ONE pinned block of the result cache: they go to the device at PCIe speed; from pageable
  // vectors the 35 MB per part took 6-10 ms)
  // groups per lattice: one per match with a destination + one epsilon per boundary + the finals
  std::vector<uint32_t> g_cnt(NL + 1, 0);
  for (size_t li = 0; li < NL; ++li) {
    const size_t si = st_of_lat[li];
    const size_t nb = stretches[si].b1 - stretches[si].b0;
    g_cnt[li + 1] = g_cnt[li] + st_nm[si + 1] + (uint32_t)nb + st_nfin[si];  // (matches without a destination get a group too: zero arcs)
  }
  const size_t G = g_cnt[NL];
  auto al64 = [](size_t x) { return (x + 63) & ~(size_t)63; };
  const size_t o_q = 0, o_u0 = o_q + al64(M * 4), o_u1 = o_u0 + al64(M * 4), o_pk = o_u1 + al64(M * 4), o_lt = o_pk + al64(M * 4), o_g = o_lt + al64(M * 4),
               o_e0 = o_g + al64(G * 4), o_el = o_e0 + al64(nin * 4), o_bo = o_el + al64(nin * 4), o_bt = o_bo + al64(std::max<size_t>(nboff, 1) * 4),
               o_au = o_bt + al64(std::max<size_t>(nbtok, 1) * 4), o_ah = o_au + al64(st_bu[NS] + 16), o_end = o_ah + al64(st_bh[NS] + 16);
  char* tblk = static_cast<char*>(anx::host_result_alloc(o_end));
  if (!tblk) return anx_fail(ANX_EINVAL, "out of memory");
  struct TblFree { char* p; ~TblFree() { anx::host_result_free(p); } } tbl_free{tblk};
  uint32_t *t_q = reinterpret_cast<uint32_t*>(tblk + o_q), *t_u0 = reinterpret_cast<uint32_t*>(tblk + o_u0), *t_u1 = reinterpret_cast<uint32_t*>(tblk + o_u1),
           *t_pack = reinterpret_cast<uint32_t*>(tblk + o_pk), *t_lat = reinterpret_cast<uint32_t*>(tblk + o_lt), *g_ref = reinterpret_cast<uint32_t*>(tblk + o_g),
           *e_g0 = reinterpret_cast<uint32_t*>(tblk + o_e0), *e_lat = reinterpret_cast<uint32_t*>(tblk + o_el), *btok_off = reinterpret_cast<uint32_t*>(tblk + o_bo);
  int32_t* btok = reinterpret_cast<int32_t*>(tblk + o_bt);
  char *arena_u = tblk + o_au, *arena_h = tblk + o_ah;
  const uint32_t rows_bound = sp->base.max_matches ? sp->base.max_matches + 1u : 200u;
  pool_for(nsc, 4, 8, [&](size_t clo, size_t chi) {
    std::vector<uint32_t> cnt, pos, ord;
    for (size_t c = clo; c < chi; ++c) {
      const Chunk& C = chunks[c];
      size_t rpos = 0;
      for (size_t si = c * SC; si < std::min(NS, (c + 1) * SC); ++si) {
        const Stretch& st = stretches[si];
        const char* text = texts[st.text_index];
        const std::vector<Span>& mv = st.matches;
        const uint32_t nu = st_nu[si + 1], li = lat_of_st[si];
        char* wu = arena_u + st_bu[si];
        char* wh = arena_h + st_bh[si];
        uint32_t iu = qu0[si], ih = qh0[si];
        for (size_t mi = 0; mi < mv.size(); ++mi, ++rpos) {
          const size_t gm = (size_t)m0[si] + mi;
          const MRec& r = C.rec[rpos];
          const size_t l = mv[mi].end - mv[mi].begin;
          char*& w = mi < nu ? wu : wh;
          memcpy(w, text + mv[mi].begin, l);
          w[l] = '\0';
          w += l + 1;
          t_q[gm] = mi < nu ? iu++ : ih++;
          t_u0[gm] = m0[si] + r.u0; t_u1[gm] = m0[si] + r.u1;
          t_pack[gm] = r.pack;
          t_lat[gm] = li;
        }
        if (li == 0xFFFFFFFFu) continue;
        // arc groups in (destination state, source state, match) order, the epsilon arc of a state behind its matches, then the finals
        const Span* bs = bounds[st.text_index].data() + st.b0;
        const size_t nb = st.b1 - st.b0, nstates = nb + 1;
        cnt.assign(nstates + 2, 0u);
        const size_t rbase = rpos - mv.size();
        uint32_t nodst = 0;
        for (size_t mi = 0; mi < mv.size(); ++mi) {
          const uint32_t pk = C.rec[rbase + mi].pack;
          if (pk >> 31) ++nodst; else ++cnt[((pk >> 12) & 0xFFFu) + 1];
        }
        for (size_t d = 1; d <= nstates + 1; ++d) cnt[d] += cnt[d - 1];
        ord.assign(mv.size(), 0u);
        pos.assign(cnt.begin(), cnt.end());
        for (size_t mi = 0; mi < mv.size(); ++mi) {
          const uint32_t pk = C.rec[rbase + mi].pack;
          if (!(pk >> 31)) ord[pos[(pk >> 12) & 0xFFFu]++] = (uint32_t)mi;
        }
        uint32_t g = g_cnt[li], e = st_e0[li], span = 1, deg_max = 0;
        // the matches that end at no boundary come first: zero arcs, any place will do
        for (size_t mi = 0; mi < mv.size(); ++mi) if (C.rec[rbase + mi].pack >> 31) g_ref[g++] = (uint32_t)(m0[si] + mi);
        e_g0[e] = g; e_lat[e] = li; ++e;                       // state 0: no incoming arcs
        for (size_t d = 1; d < nstates; ++d) {
          e_g0[e] = g; e_lat[e] = li; ++e;
          const uint32_t a0 = cnt[d], a1 = cnt[d + 1];
          // by source state ascending, then by match index (a handful of entries)
          std::sort(ord.begin() + a0, ord.begin() + a1, [&](uint32_t x, uint32_t y) {
            const uint32_t sx = C.rec[rbase + x].pack & 0xFFFu, sy = C.rec[rbase + y].pack & 0xFFFu;
            return sx != sy ? sx < sy : x < y;
          });
          uint32_t deg = 1;
          for (uint32_t k = a0; k < a1; ++k) {
            g_ref[g++] = (uint32_t)(m0[si] + ord[k]);
            span = std::max(span, (uint32_t)d - (C.rec[rbase + ord[k]].pack & 0xFFFu));
            deg += rows_bound;
          }
          g_ref[g++] = (1u << 30) | (uint32_t)(d - 1);          // the fail-safe epsilon arc d - 1 -> d
          deg_max = std::max(deg_max, deg);
        }
        e_g0[e] = g; e_lat[e] = li; ++e;                       // the virtual end state: the finals
        uint32_t nfin = 0;
        for (size_t i = 0; i < nb; ++i)
          if (bs[i].begin == st.end || bs[i].end == st.end) { g_ref[g++] = (2u << 30) | (uint32_t)(i + 1); span = std::max(span, (uint32_t)nstates - (uint32_t)(i + 1)); ++nfin; }
        deg_max = std::max(deg_max, nfin);
        e_g0[e] = g; e_lat[e] = li; ++e;                       // end of the lattice's arcs
        (void)nodst;
        lst[li].ring = span + 1;
        maxdeg[li] = deg_max;
      }
    }
  });
  {
    size_t tb = 0, ob = 0;
    for (const Chunk& C : chunks) {
      if (!C.btok.empty()) memcpy(btok + tb, C.btok.data(), C.btok.size() * sizeof(int32_t));
      if (!C.btok_off.empty()) memcpy(btok_off + ob, C.btok_off.data(), C.btok_off.size() * sizeof(uint32_t));
      tb += C.btok.size(); ob += C.btok_off.size();
    }
  }
  lap("one pass: segments + tables");
  // ---- the device: both batches, the redundant higher-order queries cleared in between ------------------------------------------
  struct BatchFree { anx_batch* b; ~BatchFree() { if (b) anx_batch_free(b); } };
  BatchFree bu{nullptr}, bh{nullptr};
  auto failed = [&]() { const int code = anx_last_error_code(); return code ? code : ANX_ENODEVICE; };
  if (!NU) { *fallback = true; return ANX_OK; }  // no unigram at all: nothing to decode (the classic path passes the stretches through)
  bu.b = anx_batch_encode_packed(model, arena_u, st_bu[NS], NU, &sp->base);
  if (!bu.b) return failed();
  anx::Batch* eu = anx_batch_single(bu.b);
  if (!eu) { *fallback = true; return ANX_OK; }
  // the unigrams run while the higher orders are encoded (the encoder waits for the device several times: that time is the run's)
  int rc = anx_batch_run_async(model, bu.b, stream);
  if (rc != ANX_OK) return rc;
  if (NH) { bh.b = anx_batch_encode_packed(model, arena_h, st_bh[NS], NH, &sp->base); if (!bh.b) { (void)anx_batch_wait(model, bu.b); return failed(); } }
  anx::Batch* eh = bh.b ? anx_batch_single(bh.b) : nullptr;
  rc = anx_batch_wait(model, bu.b);
  if (rc != ANX_OK) return rc;
  if (bh.b && !eh) { *fallback = true; return ANX_OK; }
  anx::OnePassIn in;
  in.nmatch = M; in.ngroup = G; in.nin = nin; in.nst = NL;
  in.m_q = t_q; in.m_u0 = t_u0; in.m_u1 = t_u1; in.m_pack = t_pack; in.m_lat = t_lat; in.g_ref = g_ref;
  in.e_g0 = e_g0; in.e_lat = e_lat; in.st_m0 = st_m0.data(); in.st_e0 = st_e0.data(); in.st = lst.data(); in.maxdeg = maxdeg.data();
  in.btok_off = btok_off; in.nboff = nboff; in.btok = btok; in.nbtok = nbtok; in.out_total = nout;
  const anx::DeviceLexicon* dl = anx_replica_of(model, 0);
  std::string err;
  anx::OnePassState* stp = nullptr;
  rc = anx::search_onepass_prepare(dl, eu, eh, in, *sp, &stp, err);
  if (rc != ANX_OK) return anx_fail(rc, err);
  struct StFree { anx::OnePassState* s; ~StFree() { anx::search_onepass_free(s); } } st_free{stp};
  if (bh.b && (rc = anx_batch_run(model, bh.b, stream)) != ANX_OK) return rc;
  lap("one pass: device batches");
  anx::OnePassOut out;
  rc = anx::search_onepass_finish(m, dl, stp, eu, eh, in, *sp, out, err);
  struct OutFree { anx::OnePassOut& o; bool keep_rows = false; ~OutFree() { anx::host_result_free(o.block); if (!keep_rows) anx::host_result_free(o.rows); } } out_free{out};
  if (rc != ANX_OK) return anx_fail(rc, err);
  if (out.handed_back) { *fallback = true; return ANX_OK; }
  lap("one pass: lattices on the device");
  // the matches of the chosen paths; their variants are views of the part's row array
  po.kept.push_back(OrderRows{out.rows, nullptr});
  out_free.keep_rows = true;
  pool_for(NL, 256, 512, [&](size_t lo, size_t hi) {
    for (size_t li = lo; li < hi; ++li) {
      const size_t si = st_of_lat[li];
      const anx::LatStretch& S = lst[li];
      std::vector<Span>& o = decoded[si];
      const uint32_t n = out.out_n[li];
      o.reserve(n);
      for (uint32_t j = 0; j < n; ++j) {
        const size_t slot = (size_t)S.out0 + j;
        Span r = stretches[si].matches[out.e_match[slot]];
        r.has_variants = true;
        r.variants = RowView{out.rows + out.e_row0[slot], (size_t)(out.e_row0[slot + 1] - out.e_row0[slot])};
        r.selected = out.e_sel[slot] == 0xFFFFFFFFu ? -1 : (int)out.e_sel[slot];
        o.push_back(r);
      }
      done[si] = 1;
    }
  });
  for (size_t si = 0; si < NS; ++si) if (!has_lat[si]) done[si] = 1;  // (no matches: checked above)
  rc = anx::search_onepass_rows_wait(stp, err);  // the rows came down under the loop above (the views only point at them)
  if (rc != ANX_OK) return anx_fail(rc, err);
  lap("one pass: output");
  return ANX_OK;
}

static int find_all_part(const anx_model* model, const char* const* texts, size_t n, const anx_search_params* sp, PartOut& po) {
  const bool timing = anx::switches().search_timing != 0;
  auto tnow = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_prev = tnow();
  auto lap = [&](const char* what) { if (timing) { const double t = tnow(); fprintf(stderr, "[anx search] %-28s %8.2f ms\n", what, (t - t_prev) * 1e3); t_prev = t; } };
  const HostModel& m = anx_host_of(model);
  if (!m.built || m.lex.nclasses == 0)  // src/lib.rs:1801-1805 (the reference eprintln!s and returns no matches)
    return anx_fail(ANX_ENOTBUILT, "Model has not been built yet! Call build() before find_all_matches()");
  // One stream for everything this part sends to the device (the encodes and runs of its batches, its lattices): with four parts in
  // flight that is four streams for the runtime's four hardware queues, instead of four encoder streams plus the NULL stream for
  // the runs, mapped however the threads' first calls happened to fall.  (Multi-replica models: every shard runs on its replica's.)
  struct PartStream { int dev; void* s; ~PartStream() { anx::thread_stream_end(dev, s); } };
  const bool one_replica = anx_replica_of(model, 0) && !anx_replica_of(model, 1);
  PartStream part_stream{anx_replica_device(model, 0), one_replica ? anx::thread_stream_begin(anx_replica_device(model, 0)) : nullptr};
  std::vector<std::vector<Span>> bounds(n);
  std::vector<Stretch> stretches;
  auto parallel_for = [&](size_t count, size_t chunk, size_t serial_below, const std::function<void(size_t, size_t)>& work) { pool_for(count, chunk, serial_below, work); };
  {
    std::vector<std::vector<Stretch>> per_text_stretches(n);
    parallel_for(n, 8, 64, [&](size_t lo, size_t hi) {
      for (size_t t = lo; t < hi; ++t) {
        const char* text = texts[t];
        const size_t len = text ? strlen(text) : 0;
        if (!len) continue;
        find_boundaries(text, len, bounds[t]);
        size_t begin = 0, begin_index = 0;
        for (size_t i = 0; i < bounds[t].size(); ++i)
          if (classify(text, bounds[t], i) == HARD && bounds[t][i].begin != begin) {
            per_text_stretches[t].push_back(Stretch{t, begin, bounds[t][i].begin, begin_index, i + 1, {}});
            begin = bounds[t][i].end;
            begin_index = i + 1;
          }
      }
    });
    size_t ns = 0;
    for (auto& v : per_text_stretches) ns += v.size();
    stretches.reserve(ns);
    for (auto& v : per_text_stretches)
      for (Stretch& x : v) stretches.push_back(std::move(x));
  }
  {
    size_t ub = n;
    for (size_t t = 0; t < n; ++t) ub += bounds[t].size();
    po.ub_matches.store(ub, std::memory_order_release);
    po.signal();
  }
  lap("boundaries");
  std::vector<std::vector<Span>> decoded(stretches.size());
  std::vector<uint8_t> done(stretches.size(), 0);
  bool onepass_done = false;
  {
    // round 5: the whole part in one device pass (onepass_part) where the device decodes the lattices anyway: one replica, no context
    // rules (the host decoder applies them), rows final on the device (no host-side confusable rescoring), max_seq within the
    // decoder's node pools
    const bool lattice_needed = sp->max_ngram > 1 || m.have_lm;
    const bool eligible = anx::switches().search_onepass && one_replica && lattice_needed && m.context_rules.empty() && !anx::switches().lattice_host &&
                          !(anx::switches().confusables_host && !m.confusables.empty()) && sp->max_seq <= 4096u && sp->max_ngram >= 1 && sp->max_ngram <= 100u &&
                          !stretches.empty();
    if (eligible) {
      bool fb = false;
      const int rc1 = onepass_part(model, m, texts, sp, bounds, stretches, decoded, done, po, part_stream.s, &fb, lap);
      if (rc1 != ANX_OK) { po.free_kept(); return rc1; }
      onepass_done = !fb;
      if (fb) {  // the classic path starts from clean stretches
        for (Stretch& st : stretches) st.matches.clear();
        for (auto& d : decoded) d.clear();
        std::fill(done.begin(), done.end(), 0);
      }
    }
  }
  // the result arrays of the device batches stay alive until the output has been written
  std::vector<OrderRows>& kept = po.kept;
  auto free_kept = [&]() { po.free_kept(); };
  std::unique_ptr<char[]> arena_buf;  // the segments of an order, packed (kept over the orders)
  size_t arena_cap = 0;
  double seg_part[5] = {0, 0, 0, 0, 0};  // timing: n-grams, arena, device batch, row views, append
  double seg_t = tnow();
  auto seg_lap = [&](int i) { if (timing) { const double t = tnow(); seg_part[i] += t - seg_t; seg_t = t; } };
  // Two device batches per part, not one per n-gram order: the unigrams, then the segments of every higher order together -- whether
  // a higher-order segment is looked up (redundant_match, src/search.rs:317-336) only depends on the unigrams' results.
  // The segments of an order, chunk by chunk: 64 consecutive stretches share one segment vector (a vector per stretch and order
  // meant a quarter of a million small blocks allocated by one thread and freed by another: the allocator's arenas became the
  // bottleneck of this phase).
  const size_t SC = 64, nsc = (stretches.size() + SC - 1) / SC;
  struct ChunkSegs { std::vector<Span> segs; std::vector<uint8_t> look; std::vector<uint32_t> first; };  // first[j]: first segment of the chunk's j-th stretch
  auto parallel_chunks = [&](const std::function<void(size_t)>& work) {
    parallel_for(nsc, 4, 8, [&](size_t lo, size_t hi) { for (size_t c = lo; c < hi; ++c) work(c); });
  };
  for (uint32_t o_lo = 1; o_lo <= sp->max_ngram && !onepass_done; o_lo = (o_lo == 1 ? 2 : sp->max_ngram + 1)) {
    const uint32_t o_hi = o_lo == 1 ? 1 : sp->max_ngram, no = o_hi - o_lo + 1;
    seg_lap(4);
    std::vector<std::vector<ChunkSegs>> cs(no, std::vector<ChunkSegs>(nsc));
    // per order and stretch: first segment / first arena byte of its looked-up segments within the batch (counts first, prefix sums
    // below; the batch holds the group's orders one after the other)
    std::vector<std::vector<size_t>> seg0(no, std::vector<size_t>(stretches.size() + 1, 0)), byte0(no, std::vector<size_t>(stretches.size() + 1, 0));
    parallel_chunks([&](size_t c) {
      const size_t s_lo = c * SC, s_hi = std::min(stretches.size(), s_lo + SC);
      size_t nbound = 0;
      for (size_t si = s_lo; si < s_hi; ++si) nbound += stretches[si].b1 - stretches[si].b0 + 1;
      for (uint32_t oi = 0; oi < no; ++oi) {
        const uint32_t order = o_lo + oi;
        ChunkSegs& C = cs[oi][c];
        C.segs.reserve(nbound);  // at most one segment per boundary
        C.first.reserve(s_hi - s_lo + 1);
        for (size_t si = s_lo; si < s_hi; ++si) {
          Stretch& st = stretches[si];
          C.first.push_back((uint32_t)C.segs.size());
          find_match_ngrams(texts[st.text_index], bounds[st.text_index].data() + st.b0, st.b1 - st.b0, order, st.begin, st.end, C.segs);
        }
        C.first.push_back((uint32_t)C.segs.size());
        C.look.resize(C.segs.size());
        for (size_t si = s_lo; si < s_hi; ++si) {
          const Stretch& st = stretches[si];
          size_t ns = 0, nb = 0;
          for (uint32_t k = C.first[si - s_lo]; k < C.first[si - s_lo + 1]; ++k) {
            C.look[k] = (order == 1 || !redundant_match(C.segs[k], st.matches)) ? 1 : 0;
            if (C.look[k]) { ++ns; nb += C.segs[k].end - C.segs[k].begin + 1; }
          }
          seg0[oi][si + 1] = ns;
          byte0[oi][si + 1] = nb;
        }
      }
    });
    size_t nseg = 0, bytes = 0;
    for (uint32_t oi = 0; oi < no; ++oi) {
      seg0[oi][0] = nseg;
      byte0[oi][0] = bytes;
      for (size_t si = 0; si < stretches.size(); ++si) { seg0[oi][si + 1] += seg0[oi][si]; byte0[oi][si + 1] += byte0[oi][si]; }
      nseg = seg0[oi][stretches.size()];
      bytes = byte0[oi][stretches.size()];
    }
    seg_lap(0);
    if (nseg) {
      // all segments of the group in one NUL-separated arena, every chunk writing its own parts
      if (arena_cap < bytes) { arena_buf.reset(new char[bytes]); arena_cap = bytes; }  // (not zero-filled: every byte is written below)
      struct { char* p; size_t n; char* data() const { return p; } size_t size() const { return n; } } arena{arena_buf.get(), bytes};
      parallel_chunks([&](size_t c) {
        const size_t s_lo = c * SC, s_hi = std::min(stretches.size(), s_lo + SC);
        for (uint32_t oi = 0; oi < no; ++oi) {
          const ChunkSegs& C = cs[oi][c];
          for (size_t si = s_lo; si < s_hi; ++si) {
            const char* text = texts[stretches[si].text_index];
            char* w = arena.data() + byte0[oi][si];
            for (uint32_t k = C.first[si - s_lo]; k < C.first[si - s_lo + 1]; ++k)
              if (C.look[k]) {
                const size_t l = C.segs[k].end - C.segs[k].begin;
                memcpy(w, text + C.segs[k].begin, l);
                w[l] = '\0';
                w += l + 1;
              }
          }
        }
      });
      seg_lap(1);
      anx_result* rows = nullptr;
      size_t* offs = nullptr;
      int rc;
      if (nseg <= ((size_t)4 << 20) && arena.size() < ((size_t)1 << 32)) {
        // the arena IS the packed form of the batch (every segment followed by a NUL byte): it goes to the device as it is
        anx_batch* bt = anx_batch_encode_packed(model, arena.data(), arena.size(), nseg, &sp->base);
        rc = bt ? anx_batch_run(model, bt, part_stream.s) : ANX_EINVAL;
        if (bt && rc == ANX_OK) rc = anx_batch_fetch(bt, &rows, &offs);
        if (bt) anx_batch_free(bt);
        if (!bt) {  // code and message of the failed encode stay in anx_last_error_code() / anx_last_error()
          free_kept();
          const int code = anx_last_error_code();
          return code ? code : ANX_ENODEVICE;
        }
      } else {  // more segments than one device batch holds: the pointer form splits them
        std::vector<const char*> ptrs(nseg);
        for (uint32_t oi = 0; oi < no; ++oi) {
          size_t i = seg0[oi][0];
          for (size_t si = 0; si < stretches.size(); ++si) {
            const ChunkSegs& C = cs[oi][si / SC];
            const char* w = arena.data() + byte0[oi][si];
            for (uint32_t k = C.first[si % SC]; k < C.first[si % SC + 1]; ++k)
              if (C.look[k]) { ptrs[i++] = w; w += C.segs[k].end - C.segs[k].begin + 1; }
          }
        }
        rc = anx_find_variants_batch(model, ptrs.data(), nseg, &sp->base, &rows, &offs);
      }
      if (rc != ANX_OK) { free_kept(); return rc; }
      kept.push_back(OrderRows{rows, offs});
      seg_lap(2);
    }
    // row views + the group's segments behind the stretch's matches, order by order
    const OrderRows* orows_cur = nseg ? &kept.back() : nullptr;
    parallel_chunks([&](size_t c) {
      const size_t s_lo = c * SC, s_hi = std::min(stretches.size(), s_lo + SC);
      for (size_t si = s_lo; si < s_hi; ++si) {
        std::vector<Span>& mv = stretches[si].matches;  // one allocation per stretch for all orders
        for (uint32_t oi = 0; oi < no; ++oi) {
          ChunkSegs& C = cs[oi][c];
          size_t i = seg0[oi][si];
          const uint32_t k0 = C.first[si - s_lo], k1 = C.first[si - s_lo + 1];
          for (uint32_t k = k0; k < k1; ++k)
            if (C.look[k]) {
              Span& sg = C.segs[k];
              sg.has_variants = true;
              sg.variants = RowView{orows_cur->rows + orows_cur->offs[i], orows_cur->offs[i + 1] - orows_cur->offs[i]};
              ++i;
            }
          if (o_lo == 1) mv.reserve((size_t)(k1 - k0) * sp->max_ngram + 1);
          mv.insert(mv.end(), C.segs.begin() + k0, C.segs.begin() + k1);
        }
      }
    });
    {  // the group's segment vectors were allocated by the pool's threads: a pool thread frees them (see the end of this function)
      auto g = std::make_shared<std::vector<std::vector<ChunkSegs>>>(std::move(cs));
      HostPool::get().post([g]() mutable { g.reset(); });
    }
    seg_lap(3);
  }
  seg_lap(4);
  lap("segments + device batches");
  if (timing) {
    static const char* names[5] = {"n-grams", "arena", "device batch", "row views", "append"};
    for (int i = 0; i < 5; ++i) fprintf(stderr, "[anx search]   segments part %-15s %8.2f ms\n", names[i], seg_part[i] * 1e3);
  }
  // consolidate per stretch: the lattices are independent.  Default: all of them in one go on the device (lattice.hip: one wave per
  // stretch); models with context rules, ANX_LATTICE=host, and the lattices the device hands back are decoded by the host threads
  // (the reference: rayon over the segments and a sequential loop over the stretches, src/lib.rs:1821-1940).
  std::vector<TagPool>& tagpools = po.tagpools;
  tagpools.assign(m.context_rules.empty() ? 0 : stretches.size(), TagPool());
  TagPool no_tags;
  const bool need_lattice = sp->max_ngram > 1 || m.have_lm || !m.context_rules.empty();  // src/lib.rs:1912
  const anx::DeviceLexicon* lat_dev = anx_replica_of(model, 0);
  const bool on_device = !onepass_done && need_lattice && m.context_rules.empty() && !anx::switches().lattice_host && lat_dev && !stretches.empty();
  if (on_device) {
    const bool use_lm = m.have_lm && sp->lm_weight > 0.0f;
    // Chunks of stretches build their lattices side by side, straight into the call's arrays: ONE pinned block (the result cache of
    // engine.hip) [stretches | in_off | arcs | syms | btok_off | btok | out_n | out_syms], in which every chunk owns a region of each
    // array sized by upper bounds (symbols <= one per variant or match; arcs <= symbols + an epsilon per boundary + the finals; LM
    // tokens of a boundary <= its bytes + 1).  Until round 4 the chunks built vectors of their own that were copied into the block.
    const size_t CH = 256, nch = (stretches.size() + CH - 1) / CH;
    struct Base { size_t in, arc, sym, boff, btok, out; };
    std::vector<Base> base(nch + 1, Base{0, 0, 0, 0, 0, 0});
    parallel_for(nch, 4, 8, [&](size_t lo, size_t hi) {
      for (size_t c = lo; c < hi; ++c) {
        Base B{0, 0, 0, 0, 0, 0};
        for (size_t si = c * CH; si < std::min(stretches.size(), (c + 1) * CH); ++si) {
          const Stretch& st = stretches[si];
          const size_t nb = st.b1 - st.b0;
          size_t nsym = 0;
          for (const Span& mt : st.matches) nsym += mt.variants.empty() ? 1 : mt.variants.size();
          B.sym += nsym;
          B.arc += nsym + 2 * nb + 1;
          B.in += nb + 3;
          B.boff += nb + 1;
          B.out += nb + 1;
          if (use_lm) {
            const Span* bs = bounds[st.text_index].data() + st.b0;
            for (size_t bi = 0; bi < nb; ++bi) B.btok += bs[bi].end - bs[bi].begin + 1;
          }
        }
        base[c + 1] = B;  // (sizes; the prefix sums follow)
      }
    });
    for (size_t c = 0; c < nch; ++c) {
      const Base& A = base[c];
      Base& B = base[c + 1];
      B = Base{A.in + B.in, A.arc + B.arc, A.sym + B.sym, A.boff + B.boff, A.btok + B.btok, A.out + B.out};
    }
    const Base T = base[nch];
    if (T.arc >= ((size_t)1 << 32) || T.sym >= ((size_t)1 << 32) || T.in >= ((size_t)1 << 32) || T.out >= ((size_t)1 << 32) || T.btok >= ((size_t)1 << 32)) {
      free_kept();
      return anx_fail(ANX_ELIMIT, "more than 2^32 lattice arcs in one call: split the texts");
    }
    const size_t max_st = stretches.size();
    auto al = [](size_t x) { return (x + 63) & ~(size_t)63; };
    const size_t o_st = 0, o_in = o_st + al(max_st * sizeof(anx::LatStretch)), o_arc = o_in + al(T.in * 4), o_sym = o_arc + al(T.arc * sizeof(anx::LatArc)),
                 o_boff = o_sym + al(T.sym * sizeof(anx::LatSym)), o_btok = o_boff + al(T.boff * 4), o_outn = o_btok + al(std::max<size_t>(1, T.btok) * 4),
                 o_outs = o_outn + al(max_st * 4), o_end = o_outs + al(std::max<size_t>(1, T.out) * 4);
    char* blk = static_cast<char*>(anx::host_result_alloc(o_end));
    if (!blk) { free_kept(); return anx_fail(ANX_EINVAL, "out of memory"); }
    struct BlkFree { char* p; ~BlkFree() { anx::host_result_free(p); } } blk_free{blk};
    anx::LatStretch* g_st = reinterpret_cast<anx::LatStretch*>(blk + o_st);
    uint32_t* g_in = reinterpret_cast<uint32_t*>(blk + o_in);
    anx::LatArc* g_arc = reinterpret_cast<anx::LatArc*>(blk + o_arc);
    anx::LatSym* g_sym = reinterpret_cast<anx::LatSym*>(blk + o_sym);
    uint32_t* g_boff = reinterpret_cast<uint32_t*>(blk + o_boff);
    int32_t* g_btok = reinterpret_cast<int32_t*>(blk + o_btok);
    uint32_t* out_n = reinterpret_cast<uint32_t*>(blk + o_outn);
    uint32_t* out_syms = reinterpret_cast<uint32_t*>(blk + o_outs);
    // (plain arrays, not zero-filled vectors: 20 MB of zeroes per part otherwise)
    std::unique_ptr<SymRef[]> osym(new SymRef[std::max<size_t>(1, T.sym)]);
    std::unique_ptr<uint32_t[]> lat_of(new uint32_t[std::max<size_t>(1, max_st)]);  // lattice -> stretch
    std::vector<std::vector<anx::LatStretch>> part_st(nch);  // the chunk's lattices, in stretch order
    std::vector<std::vector<uint32_t>> part_idx(nch);        // ... and their stretches
    std::atomic<bool> overflow{false};
    parallel_for(nch, 1, 2, [&](size_t lo, size_t hi) {
      for (size_t c = lo; c < hi; ++c) {
        const size_t s0 = c * CH, s1 = std::min(stretches.size(), (c + 1) * CH);
        const Base &B = base[c], &E = base[c + 1];
        LatSink K{g_in, g_arc, g_sym, osym.get(), g_boff, g_btok, B.in, B.arc, B.sym, B.boff, B.btok, B.out, E.in, E.arc, E.sym, E.boff, E.btok};
        part_st[c].reserve(s1 - s0);
        part_idx[c].reserve(s1 - s0);
        for (size_t si = s0; si < s1; ++si) {
          Stretch& st = stretches[si];
          anx::LatStretch S;
          if (build_lattice(m, texts[st.text_index], st.matches, bounds[st.text_index].data() + st.b0, st.b1 - st.b0, st.end, *sp, use_lm, K, S)) {
            part_st[c].push_back(S);
            part_idx[c].push_back((uint32_t)si);
          } else { decoded[si].insert(decoded[si].end(), st.matches.begin(), st.matches.end()); done[si] = 1; }  // src/lib.rs:2277-2290
        }
        if (K.overflow) overflow.store(true);
      }
    });
    if (overflow.load()) { free_kept(); return anx_fail(ANX_ELIMIT, "internal error: a lattice outgrew the bounds of its chunk"); }
    size_t nlat = 0;  // the lattices, dense and in stretch order (the arrays they point into keep their gaps)
    for (size_t c = 0; c < nch; ++c) {
      if (!part_st[c].empty()) memcpy(g_st + nlat, part_st[c].data(), part_st[c].size() * sizeof(anx::LatStretch));
      for (size_t i = 0; i < part_idx[c].size(); ++i) lat_of[nlat + i] = part_idx[c][i];
      nlat += part_st[c].size();
    }
    struct { size_t st, in, arc, sym, boff, btok, out; } const TT{nlat, T.in, T.arc, T.sym, T.boff, T.btok, T.out};
    lap("lattice input");
    const anx::LatView L{g_st, TT.st, g_in, TT.in, g_arc, TT.arc, g_sym, TT.sym, g_boff, TT.boff, g_btok, TT.btok, TT.out};
    {  // every replica of the model decodes a contiguous share of the lattices (balanced by lattice nodes), each from a thread of its own
      size_t nrep = 0;
      while (anx_replica_of(model, nrep)) ++nrep;
      if (TT.st < (size_t)std::max<long>(1, anx::switches().shard_min / 2) * nrep) nrep = 1;  // (ANX_SHARD_MIN: 4096 lattices per replica by default)
      std::vector<size_t> cut(nrep + 1, TT.st);
      cut[0] = 0;
      if (nrep > 1) {
        size_t total_nodes = 0, run = 0, r = 1;
        for (size_t i = 0; i < TT.st; ++i) total_nodes += g_st[i].nstates + 1;
        for (size_t i = 0; i < TT.st && r < nrep; ++i) {
          run += g_st[i].nstates + 1;
          if (run * nrep >= total_nodes * r) cut[r++] = i + 1;
        }
      }
      std::vector<std::string> errs(nrep);
      std::vector<int> rcs(nrep, ANX_OK);
      auto job = [&](size_t r) { rcs[r] = anx::lattice_decode(m, anx_replica_of(model, r), L, cut[r], cut[r + 1] - cut[r], *sp, out_n, out_syms, errs[r]); };
      if (nrep == 1) job(0);
      else {
        std::vector<std::thread> th;
        for (size_t r = 0; r < nrep; ++r) th.emplace_back(job, r);
        for (auto& x : th) x.join();
      }
      for (size_t r = 0; r < nrep; ++r)
        if (rcs[r] != ANX_OK) { free_kept(); return anx_fail(rcs[r], errs[r]); }
    }
    lap("lattice on the device");
    parallel_for(TT.st, 256, 512, [&](size_t lo, size_t hi) {
      for (size_t li = lo; li < hi; ++li) {
        if (out_n[li] == 0xFFFFFFFFu) continue;  // handed back: the host decoder below
        const size_t si = lat_of[li];
        const anx::LatStretch& S = g_st[li];
        std::vector<Span>& out = decoded[si];
        out.reserve(out_n[li]);
        for (uint32_t j = 0; j < out_n[li]; ++j) {
          const SymRef& o = osym[S.sym0 + out_syms[S.out0 + j]];
          Span r = stretches[si].matches[o.match_index];
          r.selected = o.variant_index;
          out.push_back(std::move(r));
        }
        done[si] = 1;
      }
    });
  }
  {
    auto work = [&](size_t lo, size_t hi) {
      for (size_t si = lo; si < hi; ++si) {
        if (done[si]) continue;
        Stretch& st = stretches[si];
        if (need_lattice)
          most_likely_sequence(m, texts[st.text_index], st.matches, bounds[st.text_index].data() + st.b0, st.b1 - st.b0, st.end,
                               *sp, decoded[si], tagpools.empty() ? no_tags : tagpools[si], (uint32_t)si);
        else
          for (Span& s : st.matches) { s.selected = 0; decoded[si].push_back(std::move(s)); }
      }
    };
    parallel_for(stretches.size(), 16, 64, work);  // interleaved chunks: neighbouring stretches have similar cost
  }
  std::vector<std::vector<Span>>& per_text = po.per_text;
  per_text.assign(n, std::vector<Span>());
  po.text_rows.assign(n, 0);
  po.text_tags.assign(n, 0);
  {  // the stretches of a text are consecutive: every text gathers its own
    std::vector<size_t> first(n + 1, stretches.size());
    for (size_t si = stretches.size(); si-- > 0;) first[stretches[si].text_index] = si;
    for (size_t t = n; t-- > 0;) if (first[t] == stretches.size()) first[t] = first[t + 1];
    parallel_for(n, 8, 64, [&](size_t lo, size_t hi) {
      for (size_t t = lo; t < hi; ++t) {
        size_t total_t = 0;
        for (size_t si = first[t]; si < first[t + 1] && stretches[si].text_index == t; ++si) total_t += decoded[si].size();
        std::vector<Span>& dst = per_text[t];
        dst.reserve(total_t);
        size_t nr = 0, nt = 0;
        for (size_t si = first[t]; si < first[t + 1] && stretches[si].text_index == t; ++si)
          for (Span& s_ : decoded[si]) { nr += s_.variants.size(); nt += s_.ntags; dst.push_back(std::move(s_)); }
        po.text_rows[t] = nr;
        po.text_tags[t] = nt;
      }
    });
  }
  lap("lattice + LM");
  if (timing) {
    static const char* names[5] = {"arcs", "k-best", "paths", "LM + rules", "select + output"};
    for (int i = 0; i < 5; ++i) fprintf(stderr, "[anx search]   lattice part %-16s %8.2f ms (summed over threads)\n", names[i], (double)g_lat_ns[i].exchange(0) * 1e-6);
  }
  for (size_t t = 0; t < n; ++t) { po.total += per_text[t].size(); po.total_rows += po.text_rows[t]; po.total_tags += po.text_tags[t]; }
  {  // ~50 k vectors allocated by the pool's threads: freeing them here costs this part's thread 5-10 ms; a pool thread does it instead
    struct Garbage { std::vector<std::vector<Span>> bounds, decoded; std::vector<Stretch> stretches; };
    auto g = std::make_shared<Garbage>();
    g->bounds = std::move(bounds);
    g->decoded = std::move(decoded);
    g->stretches = std::move(stretches);
    HostPool::get().post([g]() mutable { g.reset(); });
  }
  return ANX_OK;
}

void anx_matches_free(anx_match* matches, size_t* offsets, anx_result* rows, anx_match_tag* tags) {
  if (!out_release(matches)) free(matches);   // (the two big arrays go back to the cache of output blocks)
  free(offsets);
  if (!out_release(rows)) free(rows);
  free(tags);
}

}  // extern "C"
