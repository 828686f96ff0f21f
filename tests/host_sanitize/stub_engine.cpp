// Device side replaced by stubs so that the HOST code of libanx (model, index, confusables, context rules, index image,
// formatters, C ABI, and the multi-replica sharding of the batch calls) can run under AddressSanitizer / UBSan on a box without
// a GPU.  Test infrastructure.  Default: "no device" (every device call fails).  With ANX_STUB_FAKE=1 in the environment the stub
// pretends to be 4 devices whose "engine" derives a deterministic result list from the bytes of every input alone, so that
// shard == whole can be checked for the host-side split / concatenation logic.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../analiticcl_amd/csrc/engine.h"

namespace anx {
struct DeviceLexicon { int device; };
struct Batch {
  std::vector<std::string> in;
  bool ran = false;
  std::vector<anx_result> rows;
  std::vector<size_t> off;
};
static bool fake() { const char* e = getenv("ANX_STUB_FAKE"); return e && e[0] == '1'; }
static uint64_t fnv(const std::string& s) { uint64_t h = 1469598103934665603ull; for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; } return h; }

int device_count(std::string& err) { if (fake()) return 4; err = "stub: no device"; return 0; }
DeviceLexicon* lexicon_upload(const LexiconImage&, const EncodeTables&, const AdjIndex*, int device, std::string& err, int, size_t, AdjIndex*) {
  if (!fake()) { err = "stub: no device"; return nullptr; }
  if (device < 0 || device >= 4) { err = "stub: invalid device ordinal"; return nullptr; }
  return new DeviceLexicon{device};
}
void lexicon_free(DeviceLexicon* d) { delete d; }
void device_pool_trim(int) {}
void kernel_timer_enable(bool) {}
bool kernel_timer_read(const char*, double* ms, uint64_t* n) { if (ms) *ms = 0.0; if (n) *n = 0; return false; }
int debug_band_bound(int, const uint8_t*, const uint8_t*, const uint8_t*, const uint8_t*, size_t, int, int, uint8_t*, std::string& err) { err = "stub: no device"; return ANX_ENODEVICE; }
void* stream_create(int, std::string&, bool) { return malloc(1); }
void stream_destroy(int, void* s) { free(s); }
void* host_result_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
bool host_result_is_pinned(void*) { return false; }
void batch_set_last_stream(Batch*, void*) {}
void small_stats(uint64_t* out) { out[0] = out[1] = 0; }
// the small call needs the device: the stub never takes it (the batch entry points run on the fake devices)
int small_find(const HostModel&, const DeviceLexicon*, const char* const*, size_t, const anx_params&, anx_result**, size_t**, std::string&) { return 1; }
void encoder_stream_set_override(void*) {}
void* thread_stream_begin(int) { return nullptr; }
void thread_stream_end(int, void*) {}
void host_result_cache_stats(uint64_t* hits, uint64_t* misses, uint64_t* miss_bytes) { *hits = *misses = *miss_bytes = 0; }
void host_result_free(void* p) { free(p); }
void batch_set_run_mode(Batch*, const anx_params&, int) {}
bool batch_conf_fallback(const Batch*) { return false; }
int batch_download_text(const Batch*, std::string&, std::vector<uint32_t>&, std::string& err) { err = "stub"; return ANX_ENODEVICE; }
Batch* batch_encode(const HostModel&, const DeviceLexicon* dl, const char* const* utf8, size_t n, const anx_params&, std::string& err, int* code, bool) {
  if (!dl) { err = "stub: no device"; if (code) *code = ANX_ENODEVICE; return nullptr; }
  Batch* b = new Batch();
  for (size_t i = 0; i < n; ++i) b->in.emplace_back(utf8[i] ? utf8[i] : "");
  return b;
}
Batch* batch_encode_spans(const HostModel&, const DeviceLexicon* dl, const char* blob, size_t bytes, const uint32_t* off, size_t n, const anx_params&, std::string& err, int* code, bool, bool, bool, void*) {
  if (!dl) { err = "stub: no device"; if (code) *code = ANX_ENODEVICE; return nullptr; }
  Batch* b = new Batch();
  if (off) {
    for (size_t i = 0; i < n; ++i) b->in.emplace_back(blob + off[i], blob + off[i + 1] - 1);
  } else {
    size_t p = 0;
    for (size_t i = 0; i < n; ++i) {
      const void* z = p < bytes ? memchr(blob + p, 0, bytes - p) : nullptr;
      if (!z) { err = "packed inputs hold fewer strings than announced"; if (code) *code = ANX_EINVAL; delete b; return nullptr; }
      const size_t e = (size_t)(static_cast<const char*>(z) - blob);
      b->in.emplace_back(blob + p, blob + e);
      p = e + 1;
    }
  }
  return b;
}
int batch_run(const HostModel&, const DeviceLexicon* dl, Batch* b, void*, std::string& err) {
  if (!dl || !b) { err = "stub"; return ANX_ENODEVICE; }
  b->rows.clear();
  b->off.assign(1, 0);
  for (const std::string& s : b->in) {
    const uint64_t h = fnv(s);
    for (size_t j = 0; j < s.size() % 4; ++j) b->rows.push_back(anx_result{h % 1000 + j, 1.0 / (1.0 + (double)j), 1.0, ANX_NO_VIA});
    b->off.push_back(b->rows.size());
  }
  b->ran = true;
  return ANX_OK;
}
int batch_run_async(const HostModel& m, const DeviceLexicon* dl, Batch* b, void* st, bool, std::string& err) { return batch_run(m, dl, b, st, err); }
int batch_wait(const HostModel&, const DeviceLexicon* dl, Batch* b, std::string& err) { if (!dl || !b || !b->ran) { err = "stub"; return ANX_ENODEVICE; } return ANX_OK; }
size_t batch_n_results(const Batch* b) { return b->ran ? b->rows.size() : 0; }
size_t batch_n_input(const Batch* b) { return b->in.size(); }
int batch_fetch_into(const Batch* b, anx_result* rows, size_t* offs, size_t base, std::string& err) {
  if (!b->ran) { err = "batch has not been run"; return ANX_EINVAL; }
  if (!b->rows.empty()) memcpy(rows, b->rows.data(), b->rows.size() * sizeof(anx_result));
  for (size_t i = 0; i < b->off.size(); ++i) offs[i] = base + b->off[i];
  return ANX_OK;
}
int batch_fetch_compact_into(const Batch* b, anx_topk_record* rows, uint32_t* offs, uint32_t base, std::string& err) {
  if (!b->ran) { err = "batch has not been run"; return ANX_EINVAL; }
  for (size_t i = 0; i < b->rows.size(); ++i) rows[i] = anx_topk_record{(uint32_t)b->rows[i].vocab_id, (float)b->rows[i].freq_score, b->rows[i].dist_score};
  for (size_t i = 0; i < b->off.size(); ++i) offs[i] = base + (uint32_t)b->off[i];
  return ANX_OK;
}
int batch_fetch(const HostModel&, const DeviceLexicon*, const Batch* b, anx_result** rows, size_t** offs, std::string& err) {
  if (!b) { err = "stub"; return ANX_ENODEVICE; }
  *rows = static_cast<anx_result*>(malloc((b->rows.size() + 1) * sizeof(anx_result)));
  *offs = static_cast<size_t*>(malloc(b->off.size() * sizeof(size_t)));
  return batch_fetch_into(b, *rows, *offs, 0, err);
}
int batch_fetch_pairs(const HostModel&, const DeviceLexicon*, const Batch* b, anx_pair** out, size_t* n, std::string& err) {
  if (!b || !b->ran) { err = "stub"; return ANX_ENODEVICE; }
  std::vector<anx_pair> v;
  for (size_t i = 0; i < b->in.size(); ++i)
    for (size_t j = 0; j < b->in[i].size(); ++j) v.push_back(anx_pair{(uint32_t)i, (uint32_t)(unsigned char)b->in[i][j], 0, 0, 0, 0, 1, 0, 0.0});
  *out = static_cast<anx_pair*>(malloc((v.size() + 1) * sizeof(anx_pair)));
  if (!v.empty()) memcpy(*out, v.data(), v.size() * sizeof(anx_pair));
  *n = v.size();
  return ANX_OK;
}
int batch_pair_counts(const HostModel&, const DeviceLexicon*, Batch* b, uint32_t** out, std::string& err) {
  if (!b || !b->ran) { err = "stub"; return ANX_ENODEVICE; }
  *out = static_cast<uint32_t*>(calloc(b->in.size() + 1, sizeof(uint32_t)));
  for (size_t i = 0; i < b->in.size(); ++i) (*out)[i] = (uint32_t)b->in[i].size();
  return ANX_OK;
}
int batch_export_topk(const DeviceLexicon*, const Batch*, void*, uint32_t, void*, std::string& err) { err = "stub"; return ANX_ENODEVICE; }
int batch_export_compact(const DeviceLexicon*, const Batch*, void*, size_t, void*, size_t*, std::string& err) { err = "stub"; return ANX_ENODEVICE; }
// the gather behind the C ABI: the fake devices' "device memory" is host memory, the section has the real layout
size_t batch_compact_bytes(const Batch* b) { return (((b->in.size() + 1) * sizeof(uint32_t) + 15) & ~(size_t)15) + (b->ran ? b->rows.size() : 0) * sizeof(anx_topk_record); }
int batch_gather_compact(const DeviceLexicon* dl, const Batch* b, int, void* dst, size_t capacity, void*, std::string& err) {
  if (!dl || !b || !b->ran) { err = "stub"; return ANX_ENODEVICE; }
  if (capacity < batch_compact_bytes(b)) { err = "gather buffer too small"; return ANX_ELIMIT; }
  const size_t off_bytes = ((b->in.size() + 1) * sizeof(uint32_t) + 15) & ~(size_t)15;
  return batch_fetch_compact_into(b, reinterpret_cast<anx_topk_record*>(static_cast<char*>(dst) + off_bytes), static_cast<uint32_t*>(dst), 0, err);
}
// search mode's one-pass path: the fake device hands every part back to the classic path (after the host has built its tables)
struct OnePassState { int unused; };
int search_onepass_prepare(const DeviceLexicon* dl, const Batch*, Batch*, const OnePassIn&, const anx_search_params&, OnePassState** out, std::string& err) {
  if (!dl) { err = "stub"; return ANX_ENODEVICE; }
  *out = new OnePassState{0};
  return ANX_OK;
}
int search_onepass_finish(const HostModel&, const DeviceLexicon*, OnePassState*, const Batch*, const Batch*, OnePassIn&, const anx_search_params&, OnePassOut& out, std::string&) {
  out.handed_back = true;
  return ANX_OK;
}
int search_onepass_rows_wait(OnePassState*, std::string&) { return ANX_OK; }
void search_onepass_free(OnePassState* s) { delete s; }
// the fake device hands every lattice back (out_n = 0xFFFFFFFF): search.cpp's host decoder takes them -- the fallback path
int lattice_decode(const HostModel&, const DeviceLexicon* dl, const LatView&, size_t first, size_t count, const anx_search_params&, uint32_t* out_n, uint32_t*, std::string& err) {
  if (!dl) { err = "stub"; return ANX_ENODEVICE; }
  for (size_t i = first; i < first + count; ++i) out_n[i] = 0xFFFFFFFFu;
  return ANX_OK;
}
int adjacency_debug_lists(const DeviceLexicon*, const uint64_t*, size_t, uint32_t*, uint32_t**, std::string& err) { err = "stub"; return ANX_ENODEVICE; }
void batch_stats(const Batch* b, anx_batch_stats* s) { memset(s, 0, sizeof *s); if (b) { s->n_queries = b->in.size(); s->n_results = b->rows.size(); s->ms_total = 1.0f; } }
void batch_free(Batch* b) { delete b; }
}  // namespace anx
