// host_model.cpp -- see host_model.h.  Host-only code (no HIP here).
#include "host_model.h"
#include "adjacency.h"

#include <sched.h>

#include <thread>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <numeric>
#include <sstream>

#include "unicode_tables.inc"

namespace anx {

// The first 168 primes (reference: PRIMES, src/types.rs:20-30); generated, not transcribed.
static std::vector<uint32_t> make_primes() {
  std::vector<uint32_t> p;
  for (uint32_t n = 2; p.size() < 168; ++n) {
    bool prime = true;
    for (uint32_t d : p) {
      if (d * d > n) break;
      if (n % d == 0) { prime = false; break; }
    }
    if (prime) p.push_back(n);
  }
  return p;
}
static const std::vector<uint32_t>& primes() {
  static const std::vector<uint32_t> p = make_primes();
  return p;
}

// ---- UTF-8 / Unicode ------------------------------------------------------------------------------
static inline int u8len(unsigned char c) {
  return c < 0x80 ? 1 : (c >> 5) == 0x6 ? 2 : (c >> 4) == 0xE ? 3 : (c >> 3) == 0x1E ? 4 : 1;
}
static uint32_t u8decode(const char* s, size_t avail, int* len) {
  const unsigned char* p = reinterpret_cast<const unsigned char*>(s);
  int l = u8len(p[0]);
  if ((size_t)l > avail) l = 1;
  *len = l;
  if (l == 1) return p[0];
  if (l == 2) return ((p[0] & 0x1Fu) << 6) | (p[1] & 0x3Fu);
  if (l == 3) return ((p[0] & 0x0Fu) << 12) | ((p[1] & 0x3Fu) << 6) | (p[2] & 0x3Fu);
  return ((p[0] & 0x07u) << 18) | ((p[1] & 0x3Fu) << 12) | ((p[2] & 0x3Fu) << 6) | (p[3] & 0x3Fu);
}
static bool in_table(const unsigned int (*r)[2], int n, uint32_t cp) {
  int lo = 0, hi = n - 1;
  while (lo <= hi) {
    int mid = (lo + hi) >> 1;
    if (cp < r[mid][0]) hi = mid - 1;
    else if (cp > r[mid][1]) lo = mid + 1;
    else return true;
  }
  return false;
}
bool first_char_is_lowercase(const char* s) {
  if (!s || !*s) return false;
  int l;
  return in_table(anx_uc_lower, anx_uc_lower_n, u8decode(s, strlen(s), &l));
}
bool is_alphabetic_cp(uint32_t cp) { return in_table(anx_uc_alpha, anx_uc_alpha_n, cp); }
const uint32_t (*alphabetic_ranges(uint32_t* n))[2] { *n = (uint32_t)anx_uc_alpha_n; return anx_uc_alpha; }
uint32_t utf8_decode_at(const char* s, size_t avail, int* len) { return u8decode(s, avail, len); }
static std::string trim_ws(const std::string& f);
std::string trim_whitespace(const std::string& s) { return trim_ws(s); }
static std::string trim_ws(const std::string& f) {  // str::trim(): Unicode White_Space
  size_t b = 0, e = f.size();
  while (b < e) {
    int l;
    uint32_t cp = u8decode(f.data() + b, e - b, &l);
    if (!in_table(anx_uc_ws, anx_uc_ws_n, cp)) break;
    b += (size_t)l;
  }
  while (e > b) {
    size_t k = e - 1;
    while (k > b && (static_cast<unsigned char>(f[k]) & 0xC0) == 0x80) --k;
    int l;
    uint32_t cp = u8decode(f.data() + k, e - k, &l);
    if (!in_table(anx_uc_ws, anx_uc_ws_n, cp)) break;
    e = k;
  }
  return f.substr(b, e - b);
}
static int count_chars(const std::string& s) {
  int n = 0;
  for (size_t i = 0; i < s.size(); i += (size_t)u8len((unsigned char)s[i])) ++n;
  return n;
}
// BufRead::lines(): '\n' separated, one trailing '\r' removed, no final empty line
static void split_lines(const std::string& data, std::vector<std::string>& out) {
  size_t pos = 0;
  while (pos < data.size()) {
    size_t e = data.find('\n', pos);
    size_t end = e == std::string::npos ? data.size() : e;
    size_t l = end - pos;
    if (l > 0 && data[pos + l - 1] == '\r') --l;
    out.emplace_back(data, pos, l);
    if (e == std::string::npos) break;
    pos = e + 1;
  }
}
static bool read_file(const char* path, std::string& out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::ostringstream ss;
  ss << f.rdbuf();
  out = ss.str();
  return true;
}

// ---- alphabet -----------------------------------------------------------------------------------------
bool parse_alphabet(const std::string& tsv, Alphabet& out, std::string& err) {
  std::vector<std::string> lines;
  split_lines(tsv, lines);
  for (const std::string& line : lines) {
    if (line.empty()) continue;
    std::vector<AlphabetMember> cls;
    size_t pos = 0;
    for (;;) {
      size_t e = line.find('\t', pos);
      std::string field = line.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
      std::string val;
      if (field == "\\s") val = " ";
      else if (field == "\\t") val = "\t";
      else if (field == "\\n") val = "\n";
      else val = trim_ws(field);
      if (!val.empty()) cls.push_back(AlphabetMember{val, count_chars(val)});
      if (e == std::string::npos) break;
      pos = e + 1;
    }
    out.classes.push_back(std::move(cls));
  }
  if (out.size() > kMaxAlphabet) {
    err = "alphabet has more than 166 classes (the reference's PRIMES table has 168 entries)";
    return false;
  }
  out.index();
  return true;
}

void Alphabet::index() {
  for (auto& v : by_first) v.clear();
  for (int c = 0; c < size(); ++c)
    for (const AlphabetMember& m : classes[c])
      if (!m.bytes.empty()) by_first[(unsigned char)m.bytes[0]].push_back(Cand{(int16_t)c, &m});
  for (int b = 0; b < 256; ++b) {
    const std::vector<Cand>& v = by_first[b];
    fast[b] = v.empty() ? (int16_t)-1 : (v[0].m->bytes.size() == 1 && v[0].m->nchars == 1 ? v[0].cls : (int16_t)-2);
  }
}

// Same result as trying every class and member in file order at each position (src/anahash.rs:25-39): only
// members starting with the byte at `pos` can match, and by_first keeps them in file order.
int Alphabet::scan_into(const char* text, size_t nbytes, int16_t* out, int cap) const {
  int n = 0, skip = 0;
  for (size_t pos = 0; pos < nbytes; pos += (size_t)u8len((unsigned char)text[pos])) {
    if (skip > 0) { --skip; continue; }
    int hit = fast[(unsigned char)text[pos]];
    if (hit == -2) hit = -1;
    else { if (n >= cap) return -1; out[n++] = (int16_t)hit; continue; }
    for (const Cand& cd : by_first[(unsigned char)text[pos]]) {
      const AlphabetMember& m = *cd.m;
      if (pos + m.bytes.size() <= nbytes && memcmp(text + pos, m.bytes.data(), m.bytes.size()) == 0) {
        hit = cd.cls;
        skip = m.nchars - 1;
        break;
      }
    }
    if (n >= cap) return -1;
    out[n++] = (int16_t)hit;
  }
  return n;
}

void build_encode_tables(const Alphabet& a, EncodeTables& out) {
  out.cand.clear();
  out.bytes.clear();
  for (int b = 0; b < 256; ++b) {
    out.fast[b] = a.fast[b];
    out.coff[b] = (uint32_t)(out.cand.size() / 4);
    for (const Alphabet::Cand& cd : a.by_first[b]) {
      out.cand.push_back((uint32_t)cd.cls);
      out.cand.push_back((uint32_t)cd.m->nchars);
      out.cand.push_back((uint32_t)cd.m->bytes.size());
      out.cand.push_back((uint32_t)out.bytes.size());
      out.bytes.insert(out.bytes.end(), cd.m->bytes.begin(), cd.m->bytes.end());
    }
  }
  out.coff[256] = (uint32_t)(out.cand.size() / 4);
  out.lower.clear();
  for (int i = 0; i < anx_uc_lower_n; ++i) { out.lower.push_back(anx_uc_lower[i][0]); out.lower.push_back(anx_uc_lower[i][1]); }
}

bool Alphabet::scan(const char* text, size_t nbytes, std::vector<int16_t>& out) const {
  int16_t buf[kMaxSymbols];
  const int n = scan_into(text, nbytes, buf, kMaxSymbols);
  if (n < 0) { out.clear(); return false; }
  out.assign(buf, buf + n);
  return true;
}

// ---- BigVal ---------------------------------------------------------------------------------------------
void BigVal::mul_small(uint32_t m) {
  uint64_t carry = 0;
  for (uint32_t& x : w) {
    uint64_t t = (uint64_t)x * m + carry;
    x = (uint32_t)t;
    carry = t >> 32;
  }
  if (carry) w.push_back((uint32_t)carry);
}
int BigVal::cmp(const BigVal& o) const {
  if (w.size() != o.w.size()) return w.size() < o.w.size() ? -1 : 1;
  for (size_t i = w.size(); i-- > 0;)
    if (w[i] != o.w[i]) return w[i] < o.w[i] ? -1 : 1;
  return 0;
}
std::string BigVal::to_decimal() const {
  std::vector<uint32_t> t = w;
  std::string digits;
  while (!t.empty()) {
    uint64_t rem = 0;
    for (size_t i = t.size(); i-- > 0;) {
      uint64_t cur = (rem << 32) | t[i];
      t[i] = (uint32_t)(cur / 10);
      rem = cur % 10;
    }
    digits.push_back((char)('0' + rem));
    while (!t.empty() && t.back() == 0) t.pop_back();
  }
  if (digits.empty()) digits = "0";
  std::reverse(digits.begin(), digits.end());
  return digits;
}

// ---- model ------------------------------------------------------------------------------------------------
HostModel::HostModel() {
  anx_default_weights(&weights);
  // init_vocab (src/vocab.rs:145-181): ids 0,1,2 reserved, VocabType::NONE
  for (const char* t : {"<bos>", "<eos>", "<unk>"}) {
    encoder.emplace(t, decoder.size());
    VocabEntry e;
    e.text = t;
    e.frequency = 0;
    e.lexindex = 0;
    e.tokencount = 1;
    e.vocabtype = ANX_VOCAB_NONE;
    decoder.push_back(std::move(e));
  }
}

bool HostModel::encode(const char* text, std::vector<uint8_t>& norm, std::vector<uint8_t>& cv) const {
  std::vector<int16_t> codes;
  if (!alphabet.scan(text, strlen(text), codes)) return false;
  const int A = alphabet.size();
  norm.resize(codes.size());
  cv.assign((size_t)lex.nplanes > 0 ? (size_t)lex.nplanes * 4 : (size_t)((A + 1 + 3) / 4 * 4), 0);
  for (size_t i = 0; i < codes.size(); ++i) {
    norm[i] = (uint8_t)(codes[i] >= 0 ? codes[i] : A + 1);  // src/anahash.rs:76
    cv[(size_t)(codes[i] >= 0 ? codes[i] : A)]++;           // src/anahash.rs:42 (prime index of UNK)
  }
  return true;
}

bool HostModel::anahash(const char* text, BigVal& out) const {
  std::vector<int16_t> codes;
  if (!alphabet.scan(text, strlen(text), codes)) return false;
  out.set_one();
  for (int16_t c : codes) out.mul_small(primes()[(size_t)(c >= 0 ? c : alphabet.size())]);
  return true;
}

uint64_t HostModel::add_to_vocabulary(const char* text, bool has_freq, uint32_t freq, const anx_vocab_params& p,
                                      uint8_t lexicon_index) {
  const uint32_t frequency = has_freq ? freq : 1;
  auto it = encoder.find(text);
  if (it != encoder.end()) {
    VocabEntry& item = decoder[it->second];
    switch (p.freq_handling) {
      case ANX_FREQ_SUM: item.frequency += frequency; break;
      case ANX_FREQ_MAX: if (frequency > item.frequency) item.frequency = frequency; break;
      case ANX_FREQ_MIN: if (frequency < item.frequency) item.frequency = frequency; break;
      default: item.frequency = frequency; break;
    }
    if (it->second <= 2) item.vocabtype = ANX_VOCAB_LM;
    else if ((item.vocabtype & ANX_VOCAB_TRANSPARENT) && !(p.vocab_type & ANX_VOCAB_TRANSPARENT))
      item.vocabtype ^= ANX_VOCAB_TRANSPARENT;
    item.lexindex |= 1u << lexicon_index;
    return it->second;
  }
  VocabEntry e;
  e.text = text;
  std::vector<int16_t> codes;
  if (alphabet.scan(text, strlen(text), codes)) {
    e.norm.resize(codes.size());
    for (size_t i = 0; i < codes.size(); ++i) e.norm[i] = (uint8_t)(codes[i] >= 0 ? codes[i] : alphabet.size() + 1);
  }
  e.frequency = frequency;
  e.tokencount = (uint8_t)(std::count(e.text.begin(), e.text.end(), ' ') + 1);
  e.lexindex = 1u << lexicon_index;
  e.vocabtype = p.vocab_type;
  encoder.emplace(e.text, decoder.size());
  decoder.push_back(std::move(e));
  built = false;
  return decoder.size() - 1;
}

int HostModel::read_vocabulary(const char* path, const anx_vocab_params& p, std::string& err) {
  std::string data;
  if (!read_file(path, data)) { err = std::string("cannot read ") + path; return ANX_EIO; }
  std::vector<std::string> lines;
  split_lines(data, lines);
  const uint8_t lexidx = (uint8_t)lexicons.size();
  std::vector<std::string> fields;
  for (const std::string& line : lines) {
    if (line.empty()) continue;
    fields.clear();
    size_t pos = 0;
    for (;;) {
      size_t e = line.find('\t', pos);
      fields.emplace_back(line, pos, e == std::string::npos ? std::string::npos : e - pos);
      if (e == std::string::npos) break;
      pos = e + 1;
    }
    if (p.text_column >= fields.size()) { err = "Expected text column not found"; return ANX_EINVAL; }
    uint32_t freq = 1;
    if (p.freq_column >= 0) {
      if (p.vocab_type & ANX_VOCAB_INDEXED) have_freq = true;  // src/lib.rs:544-547
      if ((size_t)p.freq_column < fields.size()) {
        const std::string& f = fields[(size_t)p.freq_column];
        char* endp = nullptr;
        unsigned long long v = strtoull(f.c_str(), &endp, 10);
        if (f.empty() || *endp || v > 0xFFFFFFFFull) { err = "frequency should be a valid integer"; return ANX_EINVAL; }
        freq = (uint32_t)v;
      }
    }
    add_to_vocabulary(fields[p.text_column].c_str(), true, freq, p, lexidx);
  }
  lexicons.push_back(path);
  return ANX_OK;
}

int HostModel::add_variant(uint64_t ref_id, const char* variant, double score, bool has_freq, uint32_t freq,
                           const anx_vocab_params& p, uint8_t lexicon_index) {
  if (ref_id >= decoder.size()) return ANX_EINVAL;
  const uint64_t variantid = add_to_vocabulary(variant, has_freq, freq, p, lexicon_index);
  if (variantid == ref_id) return 0;
  built = false;
  {  // link reference to variant: only the first mention counts
    VocabEntry& r = decoder[ref_id];
    bool dup = false;
    for (const VariantRef& x : r.variants) dup |= !x.variant_of && x.id == variantid;
    if (!dup) r.variants.push_back(VariantRef{false, variantid, score});
    r.has_variants = true;
  }
  {  // link variant to reference; the reference compares the stored id with `variantid` (src/lib.rs:502-505)
    VocabEntry& v = decoder[variantid];
    bool dup = false;
    for (const VariantRef& x : v.variants) dup |= x.variant_of && x.id == variantid;
    if (!dup) v.variants.push_back(VariantRef{true, ref_id, score});
    v.has_variants = true;
  }
  return 1;
}

int HostModel::read_variants(const char* path, const anx_vocab_params& p0, bool transparent, std::string& err) {
  std::string data;
  if (!read_file(path, data)) { err = std::string("cannot read ") + path; return ANX_EIO; }
  std::vector<std::string> lines;
  split_lines(data, lines);
  const uint8_t lexidx = (uint8_t)lexicons.size();
  anx_vocab_params p = p0, tp = p0;
  if (transparent) tp.vocab_type |= ANX_VOCAB_TRANSPARENT;
  int has_freq = -1;  // Option<bool>: autodetected on the first line (src/lib.rs:812-830)
  auto parse_u32 = [](const std::string& f, uint32_t& out) {
    if (f.empty() || f[0] == '-' ) return false;
    char* endp = nullptr;
    unsigned long long v = strtoull(f.c_str() + (f[0] == '+' ? 1 : 0), &endp, 10);
    if (*endp || v > 0xFFFFFFFFull) return false;
    out = (uint32_t)v;
    return true;
  };
  std::vector<std::string> fields;
  size_t linenr = 0;
  for (const std::string& line : lines) {
    ++linenr;
    if (line.empty()) continue;
    fields.clear();
    size_t pos = 0;
    for (;;) {
      size_t e = line.find('\t', pos);
      fields.emplace_back(line, pos, e == std::string::npos ? std::string::npos : e - pos);
      if (e == std::string::npos) break;
      pos = e + 1;
    }
    bool havef = false;
    uint32_t freq = 0;
    if (has_freq < 0) {
      if (fields.size() >= 2 && (fields.size() - 2) % 3 == 0) {
        if (parse_u32(fields[1], freq)) { has_freq = 1; havef = true; }
      } else has_freq = 0;
    } else if (has_freq == 1) {
      if (fields.size() < 2 || !parse_u32(fields[1], freq)) {
        err = "Frequency must be an integer (line " + std::to_string(linenr) + ", column 2)";
        return ANX_EINVAL;
      }
      havef = true;
    }
    const uint64_t ref_id = add_to_vocabulary(fields[0].c_str(), havef, freq, p, lexidx);
    const size_t step = has_freq == 1 ? 3 : 2;
    for (size_t i = has_freq == 1 ? 2 : 1; i + step - 1 < fields.size(); i += step) {
      char* endp = nullptr;
      const double score = strtod(fields[i + 1].c_str(), &endp);
      if (fields[i + 1].empty() || *endp) { err = "Variant scores must be a floating point value (line " + std::to_string(linenr) + ")"; return ANX_EINVAL; }
      uint32_t vf = 0;
      if (has_freq == 1 && !parse_u32(fields[i + 2], vf)) { err = "Variant frequency must be an integer (line " + std::to_string(linenr) + ")"; return ANX_EINVAL; }
      add_variant(ref_id, fields[i].c_str(), score, has_freq == 1, vf, transparent ? tp : p, lexidx);
    }
  }
  lexicons.push_back(path);
  return ANX_OK;
}

static int pick_planes(int nsym) {  // kernel variants are instantiated for these widths
  const int need = (nsym + 3) / 4;
  for (int v : {8, 16, 24, 32, 42})
    if (need <= v) return v;
  return -1;
}

namespace {
struct SwitchDef { const char* name; void (*set)(Switches&, const char*); };
int flag01(const char* v, int dflt) { return !v ? dflt : (v[0] == '0' ? 0 : 1); }
const SwitchDef kSwitches[] = {
    {"ANX_ENCODE", [](Switches& s, const char* v) { s.encode_host = v && strcmp(v, "host") == 0; }},
    {"ANX_SCAN", [](Switches& s, const char* v) { s.scan_sad = v && strcmp(v, "sad") == 0; }},
    {"ANX_SCAN_WALK", [](Switches& s, const char* v) { s.scan_walk_flat = v && strcmp(v, "flat") == 0; }},
    {"ANX_SCAN_TQ", [](Switches& s, const char* v) { const int x = v ? atoi(v) : 0; s.scan_tq = x >= 1 && x <= 64 ? x : 0; }},
    {"ANX_SCAN_ADJ", [](Switches& s, const char* v) { s.scan_adj = flag01(v, 1); }},
    {"ANX_ADJ_BUILD", [](Switches& s, const char* v) { s.adj_build_host = v && strcmp(v, "host") == 0; }},
    {"ANX_ADJ_CLOSURE", [](Switches& s, const char* v) { const int x = v ? atoi(v) : 2; s.adj_closure = x >= 0 && x <= kAdjMaxClosure ? x : kAdjMaxClosure; }},
    {"ANX_SCAN_CHUNK_FUSED", [](Switches& s, const char* v) { const int x = v ? atoi(v) : 0; s.scan_chunk_fused = x >= 32 && x <= 1024 ? x : 0; }},
    {"ANX_ADJ_FAIL", [](Switches& s, const char* v) { s.adj_fail = flag01(v, 0); }},
    {"ANX_SMALL", [](Switches& s, const char* v) { s.small_path = flag01(v, 1); }},
    {"ANX_ENC_PRIORITY", [](Switches& s, const char* v) { s.enc_priority = flag01(v, 1); }},
    {"ANX_HINTS", [](Switches& s, const char* v) { s.hints = flag01(v, 1); }},
    {"ANX_ADJ_MB", [](Switches& s, const char* v) { const long x = v ? atol(v) : 0; s.adj_budget_mb = x > 0 ? x : 16384; }},
    {"ANX_SIG_GROUPS", [](Switches& s, const char* v) { const int x = v ? atoi(v) : 0; s.sig_groups = x >= 1 && x <= 8 ? x : 0; }},
    {"ANX_PREFILTER", [](Switches& s, const char* v) { s.prefilter = flag01(v, 1); }},
    {"ANX_SCORE_FAST", [](Switches& s, const char* v) { s.score_fast = flag01(v, 1); }},
    {"ANX_FS_SPLIT", [](Switches& s, const char* v) { s.fs_split = flag01(v, 1); }},
    {"ANX_FS_PLANES", [](Switches& s, const char* v) { s.fs_planes = flag01(v, 1); }},
    {"ANX_FS_B7", [](Switches& s, const char* v) { s.fs_b7 = flag01(v, 1); }},
    {"ANX_SCAN_FUSE", [](Switches& s, const char* v) { s.fuse_prefilter = flag01(v, 1); }},
    {"ANX_CAP_DIV", [](Switches& s, const char* v) { const long x = v ? atol(v) : 0; s.cap_div = x > 1 ? x : 1; }},
    {"ANX_MAX_BATCH", [](Switches& s, const char* v) { const long x = v ? atol(v) : 0; s.max_batch = x > 0 ? x : (4l << 20); }},
    {"ANX_RUN_OVERLAP", [](Switches& s, const char* v) { s.run_overlap = (v && atoi(v) == 0 && v[0] == '0') ? 0 : 1; }},
    {"ANX_SHARD_POLICY", [](Switches& s, const char* v) { s.shard_by_length = (v && strcmp(v, "range") == 0) ? 0 : 1; }},
    {"ANX_SHARD_MIN", [](Switches& s, const char* v) { const long x = v ? atol(v) : 0; s.shard_min = x > 0 ? x : 8192; }},
    {"ANX_CONFUSABLES", [](Switches& s, const char* v) { s.confusables_host = v && strcmp(v, "host") == 0; }},
    {"ANX_LATTICE", [](Switches& s, const char* v) { s.lattice_host = v && strcmp(v, "host") == 0; }},
    {"ANX_SEARCH_ONEPASS", [](Switches& s, const char* v) { s.search_onepass = flag01(v, 1); }},
    {"ANX_ENCODE_TIMING", [](Switches& s, const char* v) { s.encode_timing = v != nullptr && v[0] != 0 && v[0] != '0'; }},
    {"ANX_SEARCH_TIMING", [](Switches& s, const char* v) { s.search_timing = v != nullptr && v[0] != 0 && v[0] != '0'; }},
    {"ANX_SEARCH_PARTS", [](Switches& s, const char* v) { const int x = v ? atoi(v) : 0; s.search_parts = x >= 1 && x <= 8 ? x : 4; }},
    {"ANX_SEARCH_PART_BYTES", [](Switches& s, const char* v) { const long x = v ? atol(v) : 0; s.search_part_bytes = x > 0 ? x : (4l << 20); }},
    {"ANX_SEARCH_PRIO", [](Switches& s, const char* v) { s.search_prio = flag01(v, 1); }},
    {"ANX_SEARCH_EARLY_OUTPUT", [](Switches& s, const char* v) { s.search_early_output = flag01(v, 1); }},
    {"ANX_SEARCH_FIRST_PCT", [](Switches& s, const char* v) { const int x = v ? atoi(v) : 0; s.search_first_pct = x >= 10 && x <= 100 ? x : 50; }},
    {"ANX_SEARCH_PARTS_MIN", [](Switches& s, const char* v) { const long x = v ? atol(v) : 0; s.search_parts_min = x > 0 ? x : (2l << 20); }},
};
}  // namespace
Switches& switches() {
  static Switches sw = []() {
    Switches s;
    for (const SwitchDef& d : kSwitches)
      if (const char* v = getenv(d.name)) d.set(s, v);
    return s;
  }();
  return sw;
}
bool set_switch(const char* name, const char* value) {
  if (!name) return false;
  for (const SwitchDef& d : kSwitches)
    if (strcmp(d.name, name) == 0) { d.set(switches(), value); return true; }
  return false;
}

unsigned usable_hw_threads() {
  static const unsigned cached = []() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n, (unsigned)std::max(1, CPU_COUNT(&set)));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota> <period>" or "max <period>"
      char q[32];
      long period = 0;
      if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
        const long quota = atol(q);
        if (quota > 0) n = std::min<unsigned>(n, (unsigned)std::max(1L, (quota + period - 1) / period));
      }
      fclose(f);
    } else {
      long quota = -1, period = 0;
      if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%ld", &quota) != 1) quota = -1; fclose(g); }
      if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%ld", &period) != 1) period = 0; fclose(g); }
      if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max(1L, (quota + period - 1) / period));
    }
    return n;
  }();
  return cached;
}

uint64_t signature_of(const uint8_t* cv, size_t n, const std::vector<uint8_t>& sym_group) {
  uint8_t g[8] = {};
  for (size_t sl = 0; sl < n; ++sl) g[sym_group[sl]] = (uint8_t)(g[sym_group[sl]] + cv[sl]);
  // bytes 0-3 -> sig_lo, 4-7 -> sig_hi: the packing only has to agree between classes and queries (v_sad_u8 is
  // bytewise) and the sort order only has to make equal signatures adjacent
  uint64_t s = 0;
  for (int i = 0; i < 8; ++i) s |= (uint64_t)g[i] << (8 * i);
  return s;
}

int HostModel::build_index(std::string& err) {
  const int A = alphabet.size();
  lex = LexiconImage();
  lex.nsym = A + 1;
  lex.nplanes = pick_planes(lex.nsym);
  if (lex.nplanes < 0) { err = "alphabet too large"; return ANX_ELIMIT; }
  const size_t cvbytes = (size_t)lex.nplanes * 4;
  class_of_cv.clear();

  struct Tmp {
    std::string cv;
    BigVal value;
    int charcount;
    std::vector<uint32_t> ids;
  };
  std::vector<Tmp> tmp;
  std::unordered_map<std::string, uint32_t> seen;
  std::vector<int16_t> codes;
  for (size_t id = 0; id < decoder.size(); ++id) {
    const VocabEntry& v = decoder[id];
    if (!(v.vocabtype & ANX_VOCAB_INDEXED)) continue;
    if (!alphabet.scan(v.text.data(), v.text.size(), codes)) continue;  // > 255 symbols: not indexable
    std::string cv(cvbytes, '\0');
    for (int16_t c : codes) cv[(size_t)(c >= 0 ? c : A)]++;
    auto it = seen.find(cv);
    if (it == seen.end()) {
      Tmp t;
      t.cv = cv;
      t.value.set_one();
      for (int16_t c : codes) t.value.mul_small(primes()[(size_t)(c >= 0 ? c : A)]);
      t.charcount = (int)codes.size();
      it = seen.emplace(cv, (uint32_t)tmp.size()).first;
      tmp.push_back(std::move(t));
    }
    tmp[it->second].ids.push_back((uint32_t)id);  // ascending id order (src/lib.rs:215-219)
  }
  {  // symbol groups of the signature: slots by decreasing total count, each to the currently lightest group
    std::vector<uint64_t> slot_freq(cvbytes, 0);
    for (const Tmp& t : tmp)
      for (size_t sl = 0; sl < cvbytes; ++sl) slot_freq[sl] += (uint8_t)t.cv[sl];
    std::vector<uint32_t> slots(cvbytes);
    std::iota(slots.begin(), slots.end(), 0u);
    std::stable_sort(slots.begin(), slots.end(), [&](uint32_t a, uint32_t b) { return slot_freq[a] > slot_freq[b]; });
    const int ngroups = switches().sig_groups ? switches().sig_groups : (decoder.size() <= kSigGroupsWideMax ? kSigGroupsWide : kSigGroups);
    uint64_t weight[8] = {};
    lex.sym_group.assign(cvbytes, 0);
    for (uint32_t sl : slots) {
      int g = 0;
      for (int i = 1; i < ngroups; ++i)
        if (weight[i] < weight[g]) g = i;
      lex.sym_group[sl] = (uint8_t)g;
      weight[g] += slot_freq[sl];
    }
  }
  std::vector<uint64_t> sig(tmp.size());
  for (size_t i = 0; i < tmp.size(); ++i) sig[i] = signature_of(reinterpret_cast<const uint8_t*>(tmp[i].cv.data()), cvbytes, lex.sym_group);
  std::vector<uint32_t> order(tmp.size());
  std::iota(order.begin(), order.end(), 0u);
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    if (tmp[a].charcount != tmp[b].charcount) return tmp[a].charcount < tmp[b].charcount;
    if (sig[a] != sig[b]) return sig[a] < sig[b];
    return tmp[a].value.cmp(tmp[b].value) < 0;
  });

  lex.nclasses = (uint32_t)tmp.size();
  lex.cstride = ((lex.nclasses + 2047u) & ~2047u) + 2048u;  // + one workgroup chunk (<= 2048) of never-matching padding
  lex.cls_planes.assign((size_t)lex.nplanes * lex.cstride, 0xFFFFFFFFu);  // padding classes never match
  lex.cls_len.assign(lex.cstride, 255);
  lex.cls_bits.assign((size_t)4 * lex.cstride, 0u);
  lex.cls_off.assign(lex.nclasses + 1, 0);
  lex.cls_value.resize(lex.nclasses);
  for (int c = 0; c <= kMaxSymbols + 1; ++c) lex.bucket_begin[c] = lex.siglen_begin[c] = 0;
  for (uint32_t r = 0; r < lex.nclasses; ++r) {
    Tmp& t = tmp[order[r]];
    if (r == 0 || sig[order[r]] != sig[order[r - 1]] || t.charcount != tmp[order[r - 1]].charcount) {
      lex.sig_lo.push_back((uint32_t)sig[order[r]]);
      lex.sig_hi.push_back((uint32_t)(sig[order[r]] >> 32));
      lex.sig_cbeg.push_back(r);
      lex.siglen_begin[t.charcount + 1]++;
    }
    for (int p = 0; p < lex.nplanes; ++p) {
      uint32_t wv;
      memcpy(&wv, t.cv.data() + 4 * p, 4);
      lex.cls_planes[(size_t)p * lex.cstride + r] = wv;
    }
    lex.cls_len[r] = (uint8_t)t.charcount;
    if (lex.nsym <= 32)
      for (int sidx = 0; sidx < lex.nsym; ++sidx)
        for (int tp = 0; tp < 4; ++tp)
          if ((uint8_t)t.cv[(size_t)sidx] > tp) lex.cls_bits[(size_t)tp * lex.cstride + r] |= 1u << sidx;
    lex.bucket_begin[t.charcount + 1]++;
    class_of_cv.emplace(t.cv, r);
    lex.cls_value[r] = std::move(t.value);
    lex.cls_off[r] = lex.nentries;
    for (uint32_t id : t.ids) {
      const VocabEntry& v = decoder[id];
      const uint32_t len = (uint32_t)v.norm.size();
      lex.ent_vocab.push_back(id);
      lex.ent_freq.push_back(v.frequency);
      lex.ent_meta.push_back(len | (first_char_is_lowercase(v.text.c_str()) ? 0x100u : 0u) |
                             (v.has_variants ? 0x200u : 0u) | ((v.vocabtype & ANX_VOCAB_TRANSPARENT) ? 0x400u : 0u));
      lex.ent_var_off.push_back((uint32_t)lex.var_target.size());
      for (const VariantRef& vr : v.variants)
        if (vr.variant_of) {  // expand_variants only follows VariantOf (src/lib.rs:1690-1711)
          lex.var_target.push_back((uint32_t)vr.id);
          lex.var_target_freq.push_back(decoder[vr.id].frequency);
          lex.var_score.push_back(vr.score);
        }
      if (v.has_variants) lex.any_variants = true;
      lex.ent_rowoff.push_back((uint32_t)(lex.rows.size() / 16));
      const size_t padded = std::max<size_t>(16, (len + 15) / 16 * 16);
      const size_t base = lex.rows.size();
      lex.rows.resize(base + padded, 0xFF);
      if (len) memcpy(&lex.rows[base], v.norm.data(), len);
      lex.nentries++;
    }
  }
  lex.cls_off[lex.nclasses] = lex.nentries;
  lex.ent_var_off.push_back((uint32_t)lex.var_target.size());
  {  // global enumeration order of entries: classes by numeric anagram value, regardless of charcount
    std::vector<uint32_t> byval(lex.nclasses);
    std::iota(byval.begin(), byval.end(), 0u);
    std::sort(byval.begin(), byval.end(), [&](uint32_t a, uint32_t b) { return lex.cls_value[a].cmp(lex.cls_value[b]) < 0; });
    lex.ent_order.assign(lex.nentries, 0);
    uint32_t pos = 0;
    for (uint32_t r : byval)
      for (uint32_t e = lex.cls_off[r]; e < lex.cls_off[r + 1]; ++e) lex.ent_order[e] = pos++;
  }
  for (int c = 0; c <= kMaxSymbols; ++c) lex.bucket_begin[c + 1] += lex.bucket_begin[c];
  for (int c = 0; c <= kMaxSymbols; ++c) lex.siglen_begin[c + 1] += lex.siglen_begin[c];
  lex.nsigs = (uint32_t)lex.sig_lo.size();
  const size_t nsig_pad = ((size_t)lex.nsigs + 63) / 64 * 64 + 64;  // the scan reads whole 64-signature steps
  lex.sig_lo.resize(nsig_pad, 0xFFFFFFFFu);
  lex.sig_hi.resize(nsig_pad, 0xFFFFFFFFu);
  lex.sig_cbeg.resize(nsig_pad + 1, lex.nclasses);
  build_lm();
  index_generation.fetch_add(1, std::memory_order_release);  // caches derived from the image (vocab_gather_order) start over
  built = true;
  return ANX_OK;
}

// into_ngram (src/lib.rs:2688-2729) with encode_token(use_unk = true): unknown parts become UNK (2)
bool HostModel::into_ngram(uint64_t vocab_id, std::vector<uint64_t>& out) const {
  out.clear();
  if (vocab_id >= decoder.size()) return false;
  const std::string& text = decoder[vocab_id].text;
  const size_t tokencount = (size_t)std::count(text.begin(), text.end(), ' ') + 1;
  if (tokencount > 5) return false;  // "Can only deal with n-grams up to order 5"
  size_t pos = 0;
  for (;;) {
    const size_t e = text.find(' ', pos);
    const std::string tok = text.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
    auto it = encoder.find(tok);
    out.push_back(it == encoder.end() ? 2 : it->second);
    if (e == std::string::npos) break;
    pos = e + 1;
  }
  return true;
}
std::string HostModel::ngram_key(const uint64_t* ids, size_t n) {
  return std::string(reinterpret_cast<const char*>(ids), n * sizeof(uint64_t));
}
void HostModel::build_lm() {
  ngrams.clear();
  unigrams.clear();
  bigrams.clear();
  std::vector<uint64_t> ng;
  for (size_t id = 0; id < decoder.size(); ++id)
    if ((decoder[id].vocabtype & ANX_VOCAB_LM) && into_ngram(id, ng)) {
      ngrams[ngram_key(ng.data(), ng.size())] += decoder[id].frequency;
      if (ng.size() == 1) unigrams[ng[0]] += decoder[id].frequency;
      else if (ng.size() == 2) bigrams[(ng[0] << 32) | (ng[1] & 0xFFFFFFFFull)] += decoder[id].frequency;
    }
  have_lm = !ngrams.empty();
  ngram_off.assign(1, 0u);
  ngram_ids.clear();
  if (have_lm) {
    ngram_off.reserve(decoder.size() + 1);
    for (size_t id = 0; id < decoder.size(); ++id) {
      if (into_ngram(id, ng))
        for (uint64_t t : ng) ngram_ids.push_back((uint32_t)t);
      ngram_off.push_back((uint32_t)ngram_ids.size());
    }
  }
}

bool HostModel::has(const char* text) const {
  if (!built) return false;
  std::vector<int16_t> codes;
  if (!alphabet.scan(text, strlen(text), codes)) return false;
  std::string cv((size_t)lex.nplanes * 4, '\0');
  for (int16_t c : codes) cv[(size_t)(c >= 0 ? c : alphabet.size())]++;
  auto it = class_of_cv.find(cv);
  if (it == class_of_cv.end()) return false;
  for (uint32_t e = lex.cls_off[it->second]; e < lex.cls_off[it->second + 1]; ++e)
    if (decoder[lex.ent_vocab[e]].text == text) return true;
  return false;
}

int clamp_threshold(const anx_threshold& t, int len, int absolute_max) {
  if (t.kind == ANX_RATIO || t.kind == ANX_RATIO_WITH_LIMIT) {
    const float v = std::floor((float)len * t.ratio);
    const int x = v < 0.0f ? 0 : v > 255.0f ? 255 : (int)v;  // `as u8` saturates
    const int lim = t.kind == ANX_RATIO ? absolute_max : (int)t.value;
    return std::min(x, lim);
  }
  const int half = std::min(255, (int)std::floor((double)len / 2.0));
  return std::min((int)t.value, half);
}

bool packed_offsets(const char* blob, size_t len, size_t n, std::vector<uint32_t>& off) {
  off.clear();
  off.reserve(n + 1);
  off.push_back(0);
  // 8 bytes per step; a byte of z is 0x80 iff the byte of x is 0 (exact: no carry crosses a byte)
  const unsigned long long L7 = 0x7F7F7F7F7F7F7F7Full;
  size_t i = 0;
  for (; i + 8 <= len && off.size() <= n; i += 8) {
    unsigned long long x;
    memcpy(&x, blob + i, 8);
    for (unsigned long long z = ~(((x & L7) + L7) | x | L7); z && off.size() <= n; z &= z - 1)
      off.push_back((uint32_t)(i + ((size_t)__builtin_ctzll(z) >> 3) + 1));
  }
  for (; i < len && off.size() <= n; ++i)
    if (!blob[i]) off.push_back((uint32_t)(i + 1));
  return off.size() == n + 1;
}

}  // namespace anx
