// engine.h -- device side of the anx engine (gfx950 only).  See DESIGN.md for the data layout.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "host_model.h"
#include "adjacency.h"

namespace anx {

struct DeviceLexicon;  // HBM-resident SoA lexicon
struct Batch;          // encoded queries + pipeline buffers + results, HBM-resident

int device_count(std::string& err);
// adj: the signature adjacency lists of the image (adjacency.h; nullptr = none: every scan tile probes its ball itself)
// dev_closure >= 0 (and adj == nullptr): the lists are built ON THE DEVICE (adjacency.hip) for that closure within dev_budget bytes;
// *dev_stats (may be nullptr) receives what the host builder reports (counts and the statistics of the length split)
DeviceLexicon* lexicon_upload(const LexiconImage& img, const EncodeTables& et, const AdjIndex* adj, int device, std::string& err, int dev_closure = -1,
                              size_t dev_budget = 0, AdjIndex* dev_stats = nullptr);
void lexicon_free(DeviceLexicon*);
// test hook: the band-match bound of the scan / scoring kernels on n (query row, candidate row) pairs (engine.hip k_debug_band_bound)
int debug_band_bound(int device, const uint8_t* q_rows, const uint8_t* c_rows, const uint8_t* lq, const uint8_t* lc, size_t n, int d, int form,
                     uint8_t* out, std::string& err);
void kernel_timer_enable(bool on);  // also clears the totals
bool kernel_timer_read(const char* name, double* total_ms, uint64_t* launches);  // waits for the recorded launches
void device_pool_trim(int device);  // hands the cached scratch blocks of the device (and the pinned result buffers) back to the driver
// result rows of batch_fetch live in cached pinned host buffers: release them with host_result_free (falls back to free())
// a non-blocking stream on `device` for a replica of a multi-device model (hipStream_t behind void*)
void* stream_create(int device, std::string& err, bool high_priority = false);  // high_priority: see engine.hip make_stream
void stream_destroy(int device, void* stream);
void* host_result_alloc(size_t bytes);
bool host_result_is_pinned(void* p);  // a block of host_result_alloc that is pinned (device-visible) host memory
void host_result_free(void* p);
void* thread_stream_begin(int device);            // see engine.hip; nullptr = nothing to end
void thread_stream_end(int device, void* stream);
void encoder_stream_set_override(void* stream);  // this thread's encodes (and lattice decodes) use `stream` (nullptr: the pool's streams again)
void host_result_cache_stats(uint64_t* hits, uint64_t* misses, uint64_t* miss_bytes);  // since the library was loaded (diagnosis)

// keep_text: the inputs' bytes stay on the device with the batch (confusable weighting on the device reads them)
Batch* batch_encode(const HostModel& m, const DeviceLexicon* dl, const char* const* utf8, size_t n,
                    const anx_params& p, std::string& err, int* code, bool keep_text = false);
// the same with the inputs in one buffer: input i = blob[off[i] .. off[i+1] - 1), followed by one NUL byte.  off == nullptr: the
// inputs are the first n NUL-terminated spans of blob[0, blob_bytes) and the device finds their offsets itself
// blob_on_device: `blob` is device memory of the replica's device (off must be nullptr then): the encoder copies it device to device
Batch* batch_encode_spans(const HostModel& m, const DeviceLexicon* dl, const char* blob, size_t blob_bytes, const uint32_t* off, size_t n,
                          const anx_params& p, std::string& err, int* code, bool keep_text = false, bool blob_on_device = false,
                          bool after_stream = false, void* src_stream = nullptr);  // after_stream: the encoder's stream first waits for what src_stream holds now
// how the following runs treat confusables: conf_mode 0 = not on the device (the caller rescored / has none), 1 = late, 2 = early
// (src/lib.rs:1591-1595 / :1505-1508); `p` = the parameters those runs use
void batch_set_run_mode(Batch* b, const anx_params& p, int conf_mode);
bool batch_conf_fallback(const Batch* b);  // the last run met a row the device could not weight: repeat it with host-side weighting
// the inputs as the batch holds them on the device (keep_text): bytes and n + 1 offsets
int batch_download_text(const Batch* b, std::string& text, std::vector<uint32_t>& off, std::string& err);

int batch_run(const HostModel& m, const DeviceLexicon* dl, Batch* b, void* stream, std::string& err);
// the same in two halves: enqueue on `stream` and return / wait for it (statistics, results usable afterwards)
int batch_run_async(const HostModel& m, const DeviceLexicon* dl, Batch* b, void* stream, bool own_streams, std::string& err);
int batch_wait(const HostModel& m, const DeviceLexicon* dl, Batch* b, std::string& err);
void batch_set_last_stream(Batch* b, void* stream);  // where the fetches / exports of a FINISHED run are enqueued from now on
int batch_fetch(const HostModel& m, const DeviceLexicon* dl, const Batch* b, anx_result** rows, size_t** offs,
                std::string& err);
// the same into caller-provided storage: rows[0 .. batch_n_results) and offs[0 .. batch_n_input] = base + CSR offsets
// The small call (small_path.hpp): find_variants for at most 4096 inputs of at most 64 bytes in eleven launches and one host wait,
// no allocation.  0: done (*rows: a block of the pinned result cache, *offs: malloc'd, as batch_fetch returns them); 1: not taken
// (too many / too long inputs, variant lists, StopAtExactMatch, a fixed capacity exceeded): use the batch path; negative: device error.
int small_find(const HostModel& m, const DeviceLexicon* dl, const char* const* utf8, size_t n, const anx_params& p, anx_result** rows, size_t** offs, std::string& err);
void small_stats(uint64_t* out);  // out[0] = calls the small path answered, out[1] = calls it handed to the batch path after a capacity overflow
size_t batch_n_results(const Batch* b);
size_t batch_n_input(const Batch* b);
int batch_fetch_into(const Batch* b, anx_result* rows, size_t* offs, size_t base, std::string& err);
int batch_fetch_compact_into(const Batch* b, anx_topk_record* rows, uint32_t* offs, uint32_t base, std::string& err);
int batch_fetch_pairs(const HostModel& m, const DeviceLexicon* dl, const Batch* b, anx_pair** out, size_t* n,
                      std::string& err);
int batch_pair_counts(const HostModel& m, const DeviceLexicon* dl, Batch* b, uint32_t** out, std::string& err);
int batch_export_topk(const DeviceLexicon* dl, const Batch* b, void* dst, uint32_t stride, void* stream,
                      std::string& err);
int batch_export_compact(const DeviceLexicon* dl, const Batch* b, void* dst, size_t capacity, void* stream,
                         size_t* used, std::string& err);
// the top-k gather behind the C ABI: the compact export of a batch (as batch_export_compact) placed at dst on device dst_device --
// exported in place when the batch lives there, else exported on its own device and copied over (hipMemcpyPeerAsync: xGMI where the
// devices see each other, staged through the host otherwise); returns when the bytes are at their destination
size_t batch_compact_bytes(const Batch* b);
int batch_gather_compact(const DeviceLexicon* dl, const Batch* b, int dst_device, void* dst, size_t capacity, void* stream, std::string& err);
// test hook (adjacency.hip): the adjacency lists of the given signatures as the replica holds them
int adjacency_debug_lists(const DeviceLexicon* d, const uint64_t* sigs, size_t n, uint32_t* out_cum, uint32_t** out_ids, std::string& err);
void batch_stats(const Batch* b, anx_batch_stats* s);

// ---- search mode's lattice decoding on the device (lattice.hip) ------------------------------------------------------------------
// One lattice per stretch of text between hard boundaries (src/lib.rs:2088-2276): states = boundaries + start, arcs = the variants
// of the segments (+ out-of-vocabulary and fail-safe epsilon arcs), listed per DESTINATION state in (source state, arc number)
// order; state nstates is a virtual end state behind the final states (zero-cost epsilon arcs).  Indices inside a stretch are local.
struct LatStretch {
  uint32_t nstates;          // without the virtual end state
  uint32_t in_off0;          // first of its nstates + 2 entries of LatInput::in_off
  uint32_t arc0, sym0;       // first arc / symbol
  uint32_t btok_off0, btok0; // first of its nb + 1 entries of btok_off / first boundary token
  uint32_t out0;             // first slot of its chosen symbols in the output
  float best_cost_init;      // (nb - 1) * 2 (src/lib.rs:2319)
  uint32_t ring;             // 1 + the most states an arc of the stretch spans (virtual end arcs included): the cost lists a merge reads
  uint64_t node0;            // set by the driver: first node of the stretch in the launch's node pool
};
struct LatArc { float cost; uint32_t src; uint32_t sym; };   // sym: local symbol id, 0xFFFFFFFF = epsilon
struct LatSym { uint32_t vocab_id; uint32_t boundary; };     // OutputSymbol (src/search.rs:133-150): vocab id 0 = out of vocabulary
struct LatInput {
  std::vector<LatStretch> st;
  std::vector<uint32_t> in_off;
  std::vector<LatArc> arcs;
  std::vector<LatSym> syms;
  std::vector<uint32_t> btok_off;
  std::vector<int32_t> btok;   // LM tokens of the boundary texts (-1 = not in the vocabulary)
  size_t out_total = 0;
};
// the whole call's lattices as plain arrays (search.cpp lays them out in pinned host memory: host_result_alloc)
struct LatView {
  const LatStretch* st; size_t nst;
  const uint32_t* in_off; size_t nin;
  const LatArc* arcs; size_t narcs;
  const LatSym* syms; size_t nsyms;
  const uint32_t* btok_off; size_t nboff;
  const int32_t* btok; size_t nbtok;
  size_t out_total;
};
// out_n[i] = symbols on the chosen path of stretch i (0xFFFFFFFF: not decoded here -> the host decoder), out_syms[st[i].out0 ..]
// decodes the lattices [first, first + count) of the view on the replica `dl` (a multi-device model gives every replica a share)
int lattice_decode(const HostModel& m, const DeviceLexicon* dl, const LatView& in, size_t first, size_t count, const anx_search_params& p,
                   uint32_t* out_n, uint32_t* out_syms, std::string& err);
void batch_free(Batch*);

// ---- search mode in one device pass (lattice.hip): the lattices are built on the device from the rows of the part's batches --------------
// What the host knows without any result: the segments ("matches") of every stretch, in the order of the stretch's match list
// (order-major), the states they connect, and the layout of the arcs ("groups", listed per destination state in (source state,
// insertion) order: a segment's variants | the epsilon arc of a state | an arc into the virtual end state).
struct OnePassIn {
  size_t nmatch = 0, ngroup = 0, nin = 0, nst = 0;
  const uint32_t* m_q = nullptr;     // [nmatch] input index of the segment in its batch (order 1: the unigram batch, else the higher-order batch); 0xFFFFFFFF: none
  const uint32_t* m_u0 = nullptr;    // [nmatch] order > 1: the unigram matches [u0, u1) (part-wide match indices) that lie inside the segment (redundant_match)
  const uint32_t* m_u1 = nullptr;
  const uint32_t* m_pack = nullptr;  // [nmatch] source state | destination state << 12 | order << 24 | (ends at no boundary) << 31
  const uint32_t* m_lat = nullptr;   // [nmatch] lattice of its stretch (index into st), 0xFFFFFFFF: the stretch has no lattice
  const uint32_t* g_ref = nullptr;   // [ngroup] kind << 30 | value: 0 match index, 1 epsilon arc (value = source state), 2 arc into the end state (value = source state)
  const uint32_t* e_g0 = nullptr;    // [nin] first group of every (lattice, state) entry of in_off: nstates + 2 per lattice
  const uint32_t* e_lat = nullptr;   // [nin] its lattice
  const uint32_t* st_m0 = nullptr;   // [nst] first match of the lattice's stretch
  const uint32_t* st_e0 = nullptr;   // [nst] first in_off entry of the lattice (= LatStretch::in_off0)
  LatStretch* st = nullptr;          // [nst] nstates, in_off0, btok_off0, btok0, out0, best_cost_init, ring set; arc0 / sym0 are the device's
  const uint32_t* maxdeg = nullptr;  // [nst] upper bound of the incoming arcs of a state
  const uint32_t* btok_off = nullptr; size_t nboff = 0;
  const int32_t* btok = nullptr; size_t nbtok = 0;
  size_t out_total = 0;              // out slots (LatStretch::out0: nstates per lattice)
};
struct OnePassOut {   // host_result_alloc'd (block, rows): release with host_result_free
  char* block = nullptr;
  uint32_t* out_n = nullptr;    // [nst] symbols on the chosen path
  uint32_t* e_match = nullptr;  // [out_total] per out slot: match index within its stretch
  uint32_t* e_sel = nullptr;    // ... chosen variant, 0xFFFFFFFF = none (out of vocabulary)
  uint32_t* e_row0 = nullptr;   // [out_total + 1] ... first of the match's rows in `rows`
  anx_result* rows = nullptr;
  size_t n_rows = 0;
  bool handed_back = false;     // some lattice is beyond the device decoder's limits: nothing above is valid, take the classic path
};
struct OnePassState;
int search_onepass_prepare(const DeviceLexicon* dl, const Batch* bu, Batch* bh, const OnePassIn& in, const anx_search_params& p, OnePassState** out, std::string& err);
int search_onepass_finish(const HostModel& m, const DeviceLexicon* dl, OnePassState* s, const Batch* bu, const Batch* bh, OnePassIn& in, const anx_search_params& p,
                          OnePassOut& out, std::string& err);
int search_onepass_rows_wait(OnePassState* s, std::string& err);  // the row array of search_onepass_finish is complete
void search_onepass_free(OnePassState*);

}  // namespace anx
