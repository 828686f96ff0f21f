// Device side replaced by "no device" stubs so that the HOST code of libanx (model, index, confusables, context rules,
// index image, formatters, C ABI) can run under AddressSanitizer / UBSan on a box without a GPU.  Test infrastructure.
#include <cstdlib>
#include "../../analiticcl_amd/csrc/engine.h"

namespace anx {
int device_count(std::string& err) { err = "stub: no device"; return 0; }
DeviceLexicon* lexicon_upload(const LexiconImage&, const EncodeTables&, int, std::string& err) { err = "stub: no device"; return nullptr; }
void lexicon_free(DeviceLexicon*) {}
void device_pool_trim(int) {}
void* host_result_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
void host_result_free(void* p) { free(p); }
Batch* batch_encode(const HostModel&, const DeviceLexicon*, const char* const*, size_t, const anx_params&, std::string& err, int* code) {
  err = "stub: no device"; if (code) *code = ANX_ENODEVICE; return nullptr;
}
Batch* batch_encode_spans(const HostModel&, const DeviceLexicon*, const char*, size_t, const uint32_t*, size_t, const anx_params&, std::string& err, int* code) {
  err = "stub: no device"; if (code) *code = ANX_ENODEVICE; return nullptr;
}
int batch_run(const HostModel&, const DeviceLexicon*, Batch*, void*, std::string& err) { err = "stub"; return ANX_ENODEVICE; }
int batch_run_async(const HostModel&, const DeviceLexicon*, Batch*, void*, std::string& err) { err = "stub"; return ANX_ENODEVICE; }
int batch_wait(const HostModel&, const DeviceLexicon*, Batch*, std::string& err) { err = "stub"; return ANX_ENODEVICE; }
int batch_fetch(const HostModel&, const DeviceLexicon*, const Batch*, anx_result**, size_t**, std::string& err) { err = "stub"; return ANX_ENODEVICE; }
int batch_fetch_pairs(const HostModel&, const DeviceLexicon*, const Batch*, anx_pair**, size_t*, std::string& err) { err = "stub"; return ANX_ENODEVICE; }
int batch_pair_counts(const HostModel&, const DeviceLexicon*, Batch*, uint32_t**, std::string& err) { err = "stub"; return ANX_ENODEVICE; }
int batch_export_topk(const DeviceLexicon*, const Batch*, void*, uint32_t, void*, std::string& err) { err = "stub"; return ANX_ENODEVICE; }
int batch_export_compact(const DeviceLexicon*, const Batch*, void*, size_t, void*, size_t*, std::string& err) { err = "stub"; return ANX_ENODEVICE; }
void batch_stats(const Batch*, anx_batch_stats*) {}
void batch_free(Batch*) {}
}  // namespace anx
