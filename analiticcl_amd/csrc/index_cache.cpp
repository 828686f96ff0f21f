// index_cache.cpp -- on-disk image of a BUILT model ("packed-lexicon cache avoids
// rebuilding the index per run").  The reference rebuilds its index on every start (VariantModel::build,
// src/lib.rs:192-245: anagram values of all entries, sort of the secondary index); here the result of build_index --
// the vocabulary with the reference's id assignment and the SoA lexicon image the GPU consumes -- is written once
// and read back with plain freads.  The file is bound to the alphabet it was built with (its members are stored and
// compared on load) and to this library's layout version.
#include <algorithm>
#include <cstdio>
#include <cstring>

#include "host_model.h"

namespace anx {
namespace {

constexpr char kMagic[8] = {'A', 'N', 'X', 'I', 'D', 'X', '0', '3'};

struct Writer {
  FILE* f;
  bool ok = true;
  void raw(const void* p, size_t n) { if (ok && n && fwrite(p, 1, n, f) != n) ok = false; }
  template <typename T> void pod(const T& v) { raw(&v, sizeof v); }
  void str(const std::string& s) { pod<uint64_t>(s.size()); raw(s.data(), s.size()); }
  template <typename T> void vec(const std::vector<T>& v) { pod<uint64_t>(v.size()); raw(v.data(), v.size() * sizeof(T)); }
};
struct Reader {
  FILE* f;
  bool ok = true;
  void raw(void* p, size_t n) { if (ok && n && fread(p, 1, n, f) != n) ok = false; }
  template <typename T> void pod(T& v) { raw(&v, sizeof v); }
  void str(std::string& s) {
    uint64_t n = 0;
    pod(n);
    if (!ok || n > ((uint64_t)1 << 32)) { ok = false; return; }
    s.resize(n);
    raw(&s[0], n);
  }
  template <typename T> void vec(std::vector<T>& v) {
    uint64_t n = 0;
    pod(n);
    if (!ok || n > ((uint64_t)1 << 34) / sizeof(T)) { ok = false; return; }
    v.resize(n);
    raw(v.data(), n * sizeof(T));
  }
};

std::string alphabet_fingerprint(const Alphabet& a) {  // the member strings in file order, unambiguous
  std::string s;
  for (const auto& cls : a.classes) {
    for (const auto& m : cls) { s += m.bytes; s.push_back('\x1f'); }
    s.push_back('\x1e');
  }
  return s;
}

}  // namespace

int HostModel::save_index(const std::string& path, std::string& err) const {
  if (!built) { err = "Model has not been built yet! Call build() before save_index()"; return ANX_ENOTBUILT; }
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { err = "cannot create " + path; return ANX_EIO; }
  Writer w{f};
  w.raw(kMagic, sizeof kMagic);
  {
    uint32_t ng = 1;
    for (uint8_t g : lex.sym_group) ng = std::max<uint32_t>(ng, (uint32_t)g + 1u);
    w.pod<uint32_t>(ng);  // signature groups in use (7 or 8 by the size of the lexicon; the per-slot groups follow in the image)
  }
  w.str(alphabet_fingerprint(alphabet));
  w.str(index_tag);  // the caller's description of what the image was built from (anx_model_set_index_tag)
  w.pod<uint8_t>(have_freq ? 1 : 0);
  w.pod<uint64_t>(lexicons.size());
  for (const std::string& l : lexicons) w.str(l);
  w.pod<uint64_t>(decoder.size());
  for (const VocabEntry& v : decoder) {
    w.str(v.text);
    w.vec(v.norm);
    w.pod(v.frequency); w.pod(v.lexindex); w.pod(v.tokencount); w.pod(v.vocabtype);
    w.pod<uint8_t>(v.has_variants ? 1 : 0);
    w.pod<uint64_t>(v.variants.size());
    for (const VariantRef& r : v.variants) { w.pod<uint8_t>(r.variant_of ? 1 : 0); w.pod(r.id); w.pod(r.score); }
  }
  const LexiconImage& x = lex;
  w.pod<int32_t>(x.nsym); w.pod<int32_t>(x.nplanes); w.pod(x.nclasses); w.pod(x.nentries); w.pod(x.cstride);
  w.pod<uint8_t>(x.any_variants ? 1 : 0); w.pod(x.nsigs);
  w.raw(x.bucket_begin, sizeof x.bucket_begin);
  w.raw(x.siglen_begin, sizeof x.siglen_begin);
  w.vec(x.cls_planes); w.vec(x.cls_bits); w.vec(x.cls_len); w.vec(x.cls_off);
  w.vec(x.ent_vocab); w.vec(x.ent_freq); w.vec(x.ent_meta); w.vec(x.ent_var_off); w.vec(x.var_target);
  w.vec(x.var_target_freq); w.vec(x.var_score); w.vec(x.ent_rowoff); w.vec(x.ent_order); w.vec(x.rows);
  w.vec(x.sym_group); w.vec(x.sig_lo); w.vec(x.sig_hi); w.vec(x.sig_cbeg);
  w.raw(kMagic, sizeof kMagic);  // trailer: a truncated file fails the load
  const bool ok = w.ok && fclose(f) == 0;
  if (!ok) { err = "write error on " + path; return ANX_EIO; }
  return ANX_OK;
}

int HostModel::load_index(const std::string& path, std::string& err) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { err = "cannot open " + path; return ANX_EIO; }
  Reader r{f};
  char magic[8];
  r.raw(magic, sizeof magic);
  uint32_t groups = 0;
  r.pod(groups);
  std::string fp;
  r.str(fp);
  if (!r.ok || memcmp(magic, kMagic, sizeof kMagic) != 0 || groups < 1u || groups > 8u) {
    fclose(f);
    err = path + " is not an index image of this library version";
    return ANX_EINVAL;
  }
  if (fp != alphabet_fingerprint(alphabet)) {
    fclose(f);
    err = path + " was built with a different alphabet";
    return ANX_EINVAL;
  }
  r.str(index_tag);
  uint8_t b8 = 0;
  uint64_t n = 0;
  r.pod(b8); have_freq = b8 != 0;
  r.pod(n);
  if (n > (1u << 20)) r.ok = false;
  lexicons.assign(r.ok ? n : 0, std::string());
  for (std::string& l : lexicons) r.str(l);
  r.pod(n);
  if (n > ((uint64_t)1 << 32)) r.ok = false;
  decoder.assign(r.ok ? n : 0, VocabEntry());
  for (VocabEntry& v : decoder) {
    if (!r.ok) break;
    r.str(v.text);
    r.vec(v.norm);
    r.pod(v.frequency); r.pod(v.lexindex); r.pod(v.tokencount); r.pod(v.vocabtype);
    r.pod(b8); v.has_variants = b8 != 0;
    uint64_t nv = 0;
    r.pod(nv);
    if (nv > (1u << 24)) { r.ok = false; break; }
    v.variants.assign(nv, VariantRef());
    for (VariantRef& vr : v.variants) { r.pod(b8); vr.variant_of = b8 != 0; r.pod(vr.id); r.pod(vr.score); }
  }
  lex = LexiconImage();
  LexiconImage& x = lex;
  int32_t i32 = 0;
  r.pod(i32); x.nsym = i32; r.pod(i32); x.nplanes = i32; r.pod(x.nclasses); r.pod(x.nentries); r.pod(x.cstride);
  r.pod(b8); x.any_variants = b8 != 0; r.pod(x.nsigs);
  r.raw(x.bucket_begin, sizeof x.bucket_begin);
  r.raw(x.siglen_begin, sizeof x.siglen_begin);
  r.vec(x.cls_planes); r.vec(x.cls_bits); r.vec(x.cls_len); r.vec(x.cls_off);
  r.vec(x.ent_vocab); r.vec(x.ent_freq); r.vec(x.ent_meta); r.vec(x.ent_var_off); r.vec(x.var_target);
  r.vec(x.var_target_freq); r.vec(x.var_score); r.vec(x.ent_rowoff); r.vec(x.ent_order); r.vec(x.rows);
  r.vec(x.sym_group); r.vec(x.sig_lo); r.vec(x.sig_hi); r.vec(x.sig_cbeg);
  r.raw(magic, sizeof magic);
  // Every array the upload and the kernels index is checked against the counts it must agree with, and every offset array
  // for monotonicity and range: a corrupt or foreign image must not make lexicon_upload or a kernel read out of bounds.
  auto monotone = [](const std::vector<uint32_t>& v, size_t n, uint64_t last) {
    if (v.size() != n) return false;
    for (size_t i = 1; i < v.size(); ++i)
      if (v[i] < v[i - 1]) return false;
    return v.empty() ? last == 0 : v.back() == last;
  };
  auto below = [](const std::vector<uint32_t>& v, uint64_t bound) {
    for (uint32_t a : v)
      if (a >= bound) return false;
    return true;
  };
  bool ok = r.ok && memcmp(magic, kMagic, sizeof kMagic) == 0 && x.nsym == alphabet.size() + 1 && x.nplanes > 0 && x.nplanes <= 64 &&
            x.nplanes * 4 >= x.nsym && x.cstride >= x.nclasses && x.cstride % 2048 == 0 && x.nsigs <= x.sig_lo.size() &&
            x.cls_len.size() == x.cstride && x.cls_planes.size() == (size_t)x.nplanes * x.cstride && x.cls_bits.size() == (size_t)4 * x.cstride &&
            monotone(x.cls_off, (size_t)x.nclasses + 1, x.nentries) &&
            x.ent_vocab.size() == x.nentries && x.ent_freq.size() == x.nentries && x.ent_meta.size() == x.nentries &&
            x.ent_rowoff.size() == x.nentries && x.ent_order.size() == x.nentries &&
            monotone(x.ent_var_off, (size_t)x.nentries + 1, x.var_target.size()) && x.var_target_freq.size() == x.var_target.size() &&
            x.var_score.size() == x.var_target.size() && below(x.var_target, decoder.size()) && below(x.ent_vocab, decoder.size()) &&
            below(x.ent_order, std::max<uint64_t>(x.nentries, 1)) && x.rows.size() % 16 == 0 &&
            x.sym_group.size() == (size_t)x.nplanes * 4 && x.sig_lo.size() % 64 == 0 && x.sig_hi.size() == x.sig_lo.size() &&
            x.sig_cbeg.size() == x.sig_lo.size() + 1;
  if (ok) {
    for (uint8_t g : x.sym_group) ok = ok && g < 8;
    for (size_t i = 0; i < x.sig_cbeg.size(); ++i) ok = ok && x.sig_cbeg[i] <= x.nclasses && (i == 0 || x.sig_cbeg[i] >= x.sig_cbeg[i - 1]);
    for (int c = 0; c <= kMaxSymbols; ++c)
      ok = ok && x.siglen_begin[c] <= x.siglen_begin[c + 1] && x.siglen_begin[c + 1] <= x.nsigs && x.bucket_begin[c] <= x.bucket_begin[c + 1] &&
           x.bucket_begin[c + 1] <= x.nclasses;
    for (uint32_t e = 0; ok && e < x.nentries; ++e) {  // token rows: inside the row pool, padded to 16-byte words
      const size_t len = x.ent_meta[e] & 0xFFu, padded = std::max<size_t>(16, (len + 15) / 16 * 16);
      ok = (size_t)x.ent_rowoff[e] * 16 + padded <= x.rows.size();
    }
  }
  fclose(f);
  if (!ok) {
    decoder.clear();
    lex = LexiconImage();
    built = false;
    err = path + " is truncated or corrupt";
    return ANX_EIO;
  }
  // derived lookups
  encoder.clear();
  encoder.reserve(decoder.size() * 2);
  for (size_t id = 0; id < decoder.size(); ++id) encoder.emplace(decoder[id].text, (uint64_t)id);
  class_of_cv.clear();
  class_of_cv.reserve((size_t)x.nclasses * 2);
  std::string cv((size_t)x.nplanes * 4, '\0');
  for (uint32_t c = 0; c < x.nclasses; ++c) {
    for (int p = 0; p < x.nplanes; ++p) memcpy(&cv[(size_t)p * 4], &x.cls_planes[(size_t)p * x.cstride + c], 4);
    class_of_cv.emplace(cv, c);
  }
  build_lm();
  index_generation.fetch_add(1, std::memory_order_release);
  built = true;
  return ANX_OK;
}

// The tag of an image without loading it (ANX_OK and *tag, or an error code)
int index_read_tag(const std::string& path, std::string* tag, std::string& err) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { err = "cannot open " + path; return ANX_EIO; }
  Reader r{f};
  char magic[8];
  r.raw(magic, sizeof magic);
  uint32_t groups = 0;
  r.pod(groups);
  std::string fp;
  r.str(fp);
  r.str(*tag);
  fclose(f);
  if (!r.ok || memcmp(magic, kMagic, sizeof kMagic) != 0) { err = path + " is not an index image of this library version"; return ANX_EINVAL; }
  return ANX_OK;
}

}  // namespace anx
