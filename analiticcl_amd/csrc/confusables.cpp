// confusables.cpp -- confusable weighting of ranked results, host side.
// Follows src/confusables.rs:5-128 (pattern syntax, found_in), src/lib.rs:409-458 (loaders), :1505-1508 / :1591-1595
// (early / late rescoring) and :1733-1756 (compute_confusable_weight).  The edit script and the pattern matcher themselves live
// in confusables_core.hpp, ONE body of code compiled for the host (here) and for the device (conf.hip): this file owns the
// pattern parser, the flattened pattern tables both sides read, and the host driver (growing buffers: the host never overflows).
#include <algorithm>
#include <cstring>
#include <fstream>

#include "host_model.h"

namespace anx {
namespace {

typedef std::u32string U;

U to_u32(const char* s, size_t n) {
  U out;
  out.reserve(n);
  for (size_t i = 0; i < n;) {
    int l;
    out.push_back(utf8_decode_at(s + i, n - i, &l));
    i += (size_t)l;
  }
  return out;
}
U to_u32(const std::string& s) { return to_u32(s.data(), s.size()); }
typedef cdiff::Core<1> HC;  // the shared core over a contiguous lane memory (stride 1)

// The working memory of one edit script on the host: per thread, ONE array of words [input | candidate | arena | diffs | diagonal
// arrays | frames] (confusables_core.hpp), grown (doubled) and the script recomputed when the core reports an overflow, so a host
// call always succeeds.  The result lives in the buffer until the thread's next call.
struct HostCtx {
  std::vector<uint32_t> mem;
  cdiff::Ctx c;
  cdiff::View in, cand;
  void size_for(size_t la, size_t lb, unsigned scale) {
    const size_t n = la + lb + 8;
    const size_t arena = n * 16 * scale, nd = n * 2 * scale + 16, nv = (n + 4) * 4 * scale, nf = n * scale + 8;
    mem.resize(la + lb + arena + nd * cdiff::DIFF_WORDS + nv + nf * cdiff::FRAME_WORDS);
    c.mem = mem.data();
    in = cdiff::mk(0, (uint32_t)la);
    cand = cdiff::mk((uint32_t)la, (uint32_t)lb);
    c.arena_off = (uint32_t)(la + lb); c.arena_cap = (uint32_t)arena; c.arena_used = 0;
    c.d_off = c.arena_off + c.arena_cap; c.d_cap = (uint32_t)nd; c.nd = 0;
    c.v_off = c.d_off + c.d_cap * cdiff::DIFF_WORDS; c.v_cap = (uint32_t)nv;
    c.f_off = c.v_off + c.v_cap; c.frame_cap = (uint32_t)nf;
    uint32_t na = 0;
    c.alpha = alphabetic_ranges(&na);
    c.nalpha = na;
    c.overflow = false;
  }
  // the two strings into the lane memory (after size_for)
  void load(const cdiff::cp_t* a, const cdiff::cp_t* b) {
    for (uint32_t i = 0; i < in.n; ++i) mem[in.p + i] = a[i];
    for (uint32_t i = 0; i < cand.n; ++i) mem[cand.p + i] = b[i];
  }
};
HostCtx& host_ctx() { static thread_local HostCtx h; return h; }

// edit script of (a -> b) in the calling thread's context
HostCtx& edit_script(const U& a, const U& b) {
  HostCtx& h = host_ctx();
  for (unsigned scale = 1;; scale *= 2) {
    h.size_for(a.size(), b.size(), scale);
    h.load(reinterpret_cast<const cdiff::cp_t*>(a.data()), reinterpret_cast<const cdiff::cp_t*>(b.data()));
    if (HC::edit_script(h.c, h.in, h.cand)) return h;
  }
}

std::string to_utf8(const cdiff::Ctx& cx, const cdiff::View& s) {
  std::string out;
  for (uint32_t i = 0; i < s.n; ++i) {
    const uint32_t c = HC::at(cx, s, i);
    if (c < 0x80) out.push_back((char)c);
    else if (c < 0x800) { out.push_back((char)(0xC0 | (c >> 6))); out.push_back((char)(0x80 | (c & 0x3F))); }
    else if (c < 0x10000) { out.push_back((char)(0xE0 | (c >> 12))); out.push_back((char)(0x80 | ((c >> 6) & 0x3F))); out.push_back((char)(0x80 | (c & 0x3F))); }
    else { out.push_back((char)(0xF0 | (c >> 18))); out.push_back((char)(0x80 | ((c >> 12) & 0x3F))); out.push_back((char)(0x80 | ((c >> 6) & 0x3F))); out.push_back((char)(0x80 | (c & 0x3F))); }
  }
  return out;
}

}  // namespace

std::string edit_script_string(const std::string& source, const std::string& target) {
  std::string out;
  const U a = to_u32(source), b = to_u32(target);
  const HostCtx& h = edit_script(a, b);
  for (uint32_t i = 0; i < h.c.nd; ++i) {
    const cdiff::Diff d = HC::diff_at(h.c, i);
    out.push_back((char)d.op);
    out.push_back('[');
    out += to_utf8(h.c, d.text);
    out.push_back(']');
  }
  return out;
}

// the flattened pattern tables host and device read (confusables_core.hpp Patterns)
void HostModel::rebuild_conf_tables() {
  ConfTables& t = conf_tables;
  t = ConfTables();
  for (const Confusable& c : confusables) {
    cdiff::FlatConf fc;
    fc.weight = c.weight;
    fc.op_begin = (uint32_t)t.ops.size();
    fc.nops = (uint32_t)c.ops.size();
    fc.strictbegin = c.strictbegin ? 1u : 0u;
    fc.strictend = c.strictend ? 1u : 0u;
    for (size_t k = 0; k < c.ops.size(); ++k) {
      cdiff::FlatOp fo;
      fo.op = (uint32_t)(unsigned char)c.ops[k];
      fo.simple = c.screen[k].simple ? 1u : 0u;
      fo.bits[0] = c.screen[k].bits[0];
      fo.bits[1] = c.screen[k].bits[1];
      fo.opt_begin = (uint32_t)t.opts.size();
      fo.nopt = (uint32_t)c.options[k].size();
      for (const U& o : c.options[k]) {
        t.opts.push_back(cdiff::FlatOpt{(uint32_t)t.pool.size(), (uint32_t)o.size()});
        for (char32_t ch : o) t.pool.push_back((uint32_t)ch);
      }
      t.ops.push_back(fo);
    }
    t.conf.push_back(fc);
  }
  if (t.pool.empty()) t.pool.push_back(0);
}
cdiff::Patterns HostModel::conf_patterns() const {
  cdiff::Patterns P;
  P.conf = conf_tables.conf.data();
  P.nconf = (uint32_t)conf_tables.conf.size();
  P.ops = conf_tables.ops.data();
  P.opts = conf_tables.opts.data();
  P.pool = conf_tables.pool.data();
  return P;
}

int HostModel::add_to_confusables(const std::string& script, double weight, std::string& err) {  // src/lib.rs:446-458
  if (script.empty()) { err = "empty confusable pattern"; return ANX_EINVAL; }
  Confusable c;
  c.weight = weight;
  c.strictbegin = script.front() == '^';
  c.strictend = script.back() == '$';
  const std::string body = script.substr(c.strictbegin ? 1 : 0, script.size() - (c.strictbegin ? 1 : 0) - (c.strictend ? 1 : 0));
  size_t begin = 0;
  for (size_t i = 0; i < body.size(); ++i)
    if (body[i] == ']') {
      const std::string ins = body.substr(begin, i + 1 - begin);
      if (ins.size() <= 3 || ins[1] != '[' || (ins[0] != '=' && ins[0] != '+' && ins[0] != '-')) {
        err = "invalid edit instruction '" + ins + "'";
        return ANX_EINVAL;
      }
      c.ops.push_back(ins[0]);
      std::vector<U> opts;
      const std::string inner = ins.substr(2, ins.size() - 3);
      size_t pos = 0;
      for (;;) {
        const size_t e = inner.find('|', pos);
        opts.push_back(to_u32(inner.substr(pos, e == std::string::npos ? std::string::npos : e - pos)));
        if (e == std::string::npos) break;
        pos = e + 1;
      }
      Confusable::Screen sc;
      sc.simple = true;
      for (const U& o : opts) {
        if (o.size() == 1 && o[0] < 128) sc.bits[o[0] >> 6] |= 1ull << (o[0] & 63);
        else sc.simple = false;
      }
      c.screen.push_back(sc);
      c.options.push_back(std::move(opts));
      begin = i + 1;
    }
  if (c.ops.empty()) { err = "confusable pattern without instructions"; return ANX_EINVAL; }
  confusables.push_back(std::move(c));
  rebuild_conf_tables();
  return ANX_OK;
}

int HostModel::read_confusablelist(const std::string& path, std::string& err) {  // src/lib.rs:409-443
  std::ifstream f(path, std::ios::binary);
  if (!f) { err = "cannot open " + path; return ANX_EIO; }
  std::string line;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty()) continue;
    const size_t tab = line.find('\t');
    double w = 1.0;
    if (tab != std::string::npos) {
      const size_t tab2 = line.find('\t', tab + 1);
      const std::string field = line.substr(tab + 1, tab2 == std::string::npos ? std::string::npos : tab2 - tab - 1);
      char* endp = nullptr;
      w = strtod(field.c_str(), &endp);
      if (endp == field.c_str() || *endp) { err = "score should be a float: '" + field + "'"; return ANX_EINVAL; }
    }
    const int rc = add_to_confusables(line.substr(0, tab), w, err);
    if (rc) return rc;
  }
  return ANX_OK;
}

struct HostModel::ConfCache {  // the vocabulary as code points + ASCII presence bits (also what the device replica uploads)
  std::vector<uint32_t> pool;      // code points of every item, back to back
  std::vector<uint32_t> off;       // [V + 1]
  std::vector<cdiff::CharSet> cs;  // [V]
};

const HostModel::ConfCache& HostModel::conf_vocab() const {
  const ConfCache* held = conf_cache.load(std::memory_order_acquire);
  if (!held || held->cs.size() != decoder.size()) {  // first use, or the vocabulary grew since (items are only ever appended)
    std::lock_guard<std::mutex> g(conf_cache_mu);
    held = conf_cache.load(std::memory_order_acquire);
    if (!held || held->cs.size() != decoder.size()) {
      auto cc = std::make_shared<ConfCache>();
      cc->off.reserve(decoder.size() + 1);
      cc->cs.resize(decoder.size());
      cc->off.push_back(0);
      for (size_t i = 0; i < decoder.size(); ++i) {
        const U t = to_u32(decoder[i].text);
        for (char32_t ch : t) cc->pool.push_back((uint32_t)ch);
        cc->off.push_back((uint32_t)cc->pool.size());
        cc->cs[i] = cdiff::charset_of_array(reinterpret_cast<const cdiff::cp_t*>(t.data()), (uint32_t)t.size());
      }
      if (cc->pool.empty()) cc->pool.push_back(0);
      conf_cache_owned.push_back(cc);
      conf_cache.store(cc.get(), std::memory_order_release);
      held = cc.get();
    }
  }
  return *held;
}
void HostModel::conf_vocab_arrays(const uint32_t** pool, size_t* npool, const uint32_t** off, const void** cs, size_t* n) const {
  const ConfCache& cc = conf_vocab();
  *pool = cc.pool.data(); *npool = cc.pool.size(); *off = cc.off.data(); *cs = cc.cs.data(); *n = cc.cs.size();
}

void HostModel::confusable_weights(const char* input, size_t len, const uint64_t* ids, size_t n, double* out) const {  // src/lib.rs:1733-1756
  const ConfCache& cc = conf_vocab();
  const U in = to_u32(input, len);
  const cdiff::cp_t* inp = reinterpret_cast<const cdiff::cp_t*>(in.data());
  const cdiff::CharSet ins = cdiff::charset_of_array(inp, (uint32_t)in.size());
  const cdiff::Patterns P = conf_patterns();
  HostCtx& h = host_ctx();
  for (size_t k = 0; k < n; ++k) {
    double weight = 1.0;
    if (ids[k] < cc.cs.size()) {
      const cdiff::cp_t* cand = cc.pool.data() + cc.off[ids[k]];
      const uint32_t ncand = cc.off[ids[k] + 1] - cc.off[ids[k]];
      for (unsigned scale = 1;; scale *= 2) {  // (the screen inside confusable_weight decides whether a script is computed at all)
        h.size_for(in.size(), ncand, scale);
        h.load(inp, cand);
        if (HC::confusable_weight(h.c, P, h.in, ins, h.cand, cc.cs[ids[k]], &weight)) break;
      }
    }
    out[k] = weight;
  }
}

const std::vector<uint32_t>& HostModel::vocab_gather_order() const {
  const uint64_t gen = index_generation.load(std::memory_order_acquire);
  const VocabOrder* cur = vocab_order.load(std::memory_order_acquire);
  if (!cur || cur->generation != gen || cur->order.size() != decoder.size()) {
    std::lock_guard<std::mutex> g(conf_cache_mu);
    cur = vocab_order.load(std::memory_order_acquire);
    if (!cur || cur->generation != gen || cur->order.size() != decoder.size()) {
      std::unique_ptr<VocabOrder> o(new VocabOrder{gen, std::vector<uint32_t>(decoder.size(), 0xFFFFFFFFu)});
      for (size_t e = 0; e < lex.ent_vocab.size() && e < lex.ent_order.size(); ++e)
        if (lex.ent_vocab[e] < o->order.size()) o->order[lex.ent_vocab[e]] = lex.ent_order[e];
      cur = o.get();
      vocab_order_owned.push_back(std::move(o));
      vocab_order.store(cur, std::memory_order_release);
    }
  }
  return cur->order;
}

double HostModel::confusable_weight(const std::string& input, uint64_t candidate) const {
  double w = 1.0;
  confusable_weights(input, &candidate, 1, &w);
  return w;
}

}  // namespace anx
