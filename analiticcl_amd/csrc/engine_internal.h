// engine_internal.h -- shared by the HIP translation units of the engine (engine.hip, encode.hip); include after hip_runtime.h
#pragma once
#include <cstdlib>
#include <cstring>
#include "engine.h"
#include "sig_hash.h"

namespace anx {

// a tile probes the hash table instead of walking its window when the ball is cheaper (a probe step costs ~3 walk steps) and
// no group sum is large enough to overflow the byte-wise add of an offset
__host__ __device__ inline bool tile_probes(uint32_t balln, uint32_t window, unsigned long long sig) {
  if (balln == 0 || (unsigned long long)balln * 3ull >= window) return false;
  for (int g = 0; g < 8; ++g)
    if (((sig >> (8 * g)) & 0xFFu) > 120u) return false;
  return true;
}
inline bool probe_enabled() { return !switches().scan_walk_flat; }

// Kernel timer (engine.hip; off unless anx_debug_kernel_timer(1)): HIP events around the launches of the kernels that dominate the
// configurations bench.py reports (k_scan_bits, k_filter_score, k_conf_script, k_lattice), on the launch stream; resolved lazily
// when the totals are read.  ktimer_begin returns a handle (or -1 when off) for ktimer_end.
int ktimer_begin(const char* name, hipStream_t st);
void ktimer_end(int handle, hipStream_t st);
// per-device scratch pool (engine.hip): freed blocks are kept for the next batch
hipError_t pool_malloc(void** p, size_t bytes);
void pool_free(void* p);
// non-blocking stream for one call of the device-side encoder (kept per device, reused)
hipStream_t encoder_stream_acquire(int device);
void encoder_stream_release(int device, hipStream_t s);
// confusable weighting on the device (conf.hip)
struct DeviceConf;
void conf_free(DeviceConf*);
int conf_launch(const HostModel& m, const DeviceLexicon* dl, Batch* b, hipStream_t st, bool early, uint32_t row_cap, std::string& err);
// the signature adjacency lists built on the device (adjacency.hip); host copies of its table and headers for the host encoder
int adjacency_build_device(DeviceLexicon* d, const LexiconImage& img, int closure, size_t budget_bytes, AdjIndex& stats, std::string& err);
int adjacency_host_copies(const DeviceLexicon* d, std::string& err);
// search mode's LM tables of a replica (lattice.hip)
struct DeviceLm;
void lm_free(DeviceLm*);
// device-side query encoder (encode.hip): fills the query and tile arrays of `b` from the packed inputs
int batch_encode_device(const HostModel& m, const DeviceLexicon* dl, Batch* b, const char* blob, size_t blob_bytes, const uint32_t* off, size_t n,
                        const anx_params& p, std::string& err, bool blob_on_device = false, bool after_stream = false, void* src_stream = nullptr);

// ---- the small call: device buffers of its encoder (engine.hip small_find owns them; encode.hip launches the kernels) -----------------
struct Tile;
struct SmallEnc {
  uint8_t* codes = nullptr;
  uint32_t *meta = nullptr, *bits = nullptr, *kind = nullptr, *cv = nullptr, *blk = nullptr, *perm = nullptr;
  unsigned long long *key = nullptr, *sig = nullptr;
  uint4 *q_rec = nullptr, *q_rows = nullptr;
  uint32_t *q_bits = nullptr, *q_cv = nullptr, *q_meta = nullptr, *q_orig = nullptr, *qexact = nullptr, *s_kind = nullptr;
  unsigned long long* s_sig = nullptr;
  Tile* tiles = nullptr;  // [slots * inputs] (8 * SMALL_MAX in all): slot of (query s, part) = part * n + s
};
struct SmallZero { uint32_t* p[8]; uint32_t n[8]; };  // arrays k_small_tiles clears before the run (unused entries: n = 0)
// k_enc_strings -> k_enc_gather (identity order) -> k_small_tiles on `st`: no allocation, no host wait.  blob / off may be pinned host
// memory (read over PCIe; stage_lds: every block of k_enc_strings first copies its strings into LDS, inputs of <= 64 bytes).  qw: 16-byte words per query row (from the host's bound of the longest input)
int small_encode_launch(const HostModel& m, const DeviceLexicon* dl, const SmallEnc& e, const uint8_t* blob, const uint32_t* off, uint32_t n, uint32_t qw,
                        const anx_params& p, const SmallZero& z, uint32_t slots, bool stage_lds, const uint32_t* host_off, hipStream_t st, std::string& err);  // host_off: the offsets as the HOST reads them (stage_lds: the blocks' byte ranges go into the kernel arguments)  // slots: tile slots per query (>= 8)
int small_iota(uint32_t* perm, uint32_t n, hipStream_t st);

}  // namespace anx
