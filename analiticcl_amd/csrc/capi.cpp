// capi.cpp -- the extern "C" boundary declared in include/anx.h.
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>

#include "engine.h"
#include "host_model.h"

struct anx_model {
  anx::HostModel host;
  anx::DeviceLexicon* dev = nullptr;
};
struct anx_batch {
  const anx_model* model = nullptr;
  anx::Batch* b = nullptr;
};

static thread_local std::string g_err;
static thread_local int g_code = 0;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  g_code = code;
  return code;
}

const anx::HostModel& anx_host_of(const anx_model* m) { return m->host; }
int anx_fail(int code, const std::string& msg) { return fail(code, msg); }

extern "C" {

const char* anx_last_error(void) { return g_err.c_str(); }
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
void anx_model_free(anx_model* m) {
  if (!m) return;
  anx::lexicon_free(m->dev);
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
int anx_model_to_device(anx_model* m, int device) {
  if (!m) return fail(ANX_EINVAL, "NULL model");
  if (!m->host.built) return fail(ANX_ENOTBUILT, "Model has not been built yet! Call build() first");
  anx::lexicon_free(m->dev);
  m->dev = nullptr;
  std::string err;
  m->dev = anx::lexicon_upload(m->host.lex, device, err);
  return m->dev ? ANX_OK : fail(ANX_ENODEVICE, err);
}
int anx_model_build(anx_model* m, int device) {
  if (!m) return fail(ANX_EINVAL, "NULL model");
  std::string err;
  int rc = m->host.build_index(err);
  if (rc) return fail(rc, err);
  anx::lexicon_free(m->dev);
  m->dev = nullptr;
  if (device < 0) return ANX_OK;
  return anx_model_to_device(m, device);
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

anx_batch* anx_batch_encode(const anx_model* m, const char* const* utf8, size_t n, const anx_params* p) {
  if (!m || (!utf8 && n) || !p) { fail(ANX_EINVAL, "NULL argument"); return nullptr; }
  if (!m->host.built) { fail(ANX_ENOTBUILT, "Model has not been built yet! Call build() before find_variants()"); return nullptr; }
  std::string err;
  int code = ANX_OK;
  anx::Batch* b = anx::batch_encode(m->host, m->dev, utf8, n, *p, err, &code);
  if (!b) { fail(code ? code : ANX_ENODEVICE, err); return nullptr; }
  anx_batch* h = new anx_batch();
  h->model = m;
  h->b = b;
  return h;
}
int anx_batch_run(const anx_model* m, anx_batch* b, void* stream) {
  if (!m || !b || b->model != m) return fail(ANX_EINVAL, "batch does not belong to this model");
  std::string err;
  int rc = anx::batch_run(m->host, m->dev, b->b, stream, err);
  return rc ? fail(rc, err) : ANX_OK;
}
int anx_batch_fetch(const anx_batch* b, anx_result** rows, size_t** offs) {
  if (!b || !rows || !offs) return fail(ANX_EINVAL, "NULL argument");
  std::string err;
  int rc = anx::batch_fetch(b->model->host, b->model->dev, b->b, rows, offs, err);
  return rc ? fail(rc, err) : ANX_OK;
}
int anx_batch_fetch_pairs(const anx_batch* b, anx_pair** out, size_t* n) {
  if (!b || !out || !n) return fail(ANX_EINVAL, "NULL argument");
  std::string err;
  int rc = anx::batch_fetch_pairs(b->model->host, b->model->dev, b->b, out, n, err);
  return rc ? fail(rc, err) : ANX_OK;
}
void anx_pairs_free(anx_pair* p) { free(p); }
int anx_batch_export_topk(const anx_batch* b, void* dst, uint32_t stride, void* stream) {
  if (!b) return fail(ANX_EINVAL, "NULL batch");
  std::string err;
  int rc = anx::batch_export_topk(b->model->dev, b->b, dst, stride, stream, err);
  return rc ? fail(rc, err) : ANX_OK;
}
int anx_batch_get_stats(const anx_batch* b, anx_batch_stats* s) {
  if (!b || !s) return fail(ANX_EINVAL, "NULL argument");
  anx::batch_stats(b->b, s);
  return ANX_OK;
}
void anx_batch_free(anx_batch* b) {
  if (!b) return;
  anx::batch_free(b->b);
  delete b;
}

int anx_find_variants_batch(const anx_model* m, const char* const* utf8, size_t n, const anx_params* p,
                            anx_result** out_rows, size_t** out_offsets) {
  if (!out_rows || !out_offsets) return fail(ANX_EINVAL, "NULL output argument");
  anx_batch* b = anx_batch_encode(m, utf8, n, p);
  if (!b) return g_code ? g_code : ANX_EINVAL;
  int rc = anx_batch_run(m, b, nullptr);
  if (rc == ANX_OK) rc = anx_batch_fetch(b, out_rows, out_offsets);
  anx_batch_free(b);
  return rc;
}
void anx_results_free(anx_result* rows, size_t* offsets) {
  free(rows);
  free(offsets);
}

}  // extern "C"
