// kernels_common.hpp -- device-side structures shared by the kernels and the host driver (tiles, batch state, row records)
// Part of the single translation unit engine.hip (included inside namespace anx); gfx950 only.
#pragma once

// ------------------------------------------------------------------------------------------------
// Device structures
// ------------------------------------------------------------------------------------------------
struct Tile {           // <= SCAN_TQ queries of one length and one signature (bit-plane tiles: every kind 1..NBITPLANES, sorted by kind)
  uint32_t q0, nq;      // query range (queries are sorted by (scan kind, length, signature))
  uint32_t s0, s1;      // signature range [s0, s1) covering the +-k charcount window, aligned to 64-signature blocks
  uint32_t k;           // clamped anagram distance for this length
  uint32_t lq;          // query length in symbols
  uint32_t sig_lo, sig_hi;  // the tile's signature (per-group symbol counts, one byte each)
  uint32_t kind;        // 0 = SAD body (count vectors), 1 = bit-plane body
  uint32_t d;           // clamped edit distance for this length
  uint32_t kend;        // bit-plane tiles: end (within the tile) of the queries of kind 1 | kind 2 << 8 | kind 3 << 16; the rest are kind 4
  uint32_t ball0, balln;  // balln > 0: probe the signature hash table with the balln offsets ball[ball0 ..] instead of walking [s0, s1)
  uint32_t adj;           // > 0: header index + 1 of the signature's adjacency list (adjacency.h): the tile streams it (bit-plane tiles, k <= kAdjRadius)
  uint32_t flags;         // bit 0: the first tile of its (length, signature) group (statistics: list bytes a batch has to read at least once);
                          // bit 1: the tile streams only rows [s0, s1) of its adjacency list (relative to the list's first row; the small call)
};
constexpr uint32_t BALL_MAX = 4096;   // largest L1 ball of signature offsets enumerated (per k; larger k walk the window)
constexpr uint32_t SIG_BYTE_MAX = 120;  // group sums above this take the walk (byte-wise SWAR add of an offset must not overflow)

// Symbol PLANES of a string's first 16 symbols (round 6): plane b, bit i = bit b of (code of symbol i) + 1, zero from the string's end on --
// six planes, two to a word, for alphabets whose codes + 1 fit six bits (classes + the unknown code A + 1: A <= 61; kSymbolPlanesMaxA).
// The mismatch masks of kernels_score.hpp (bit i = s[i] != t[i + k]) are OR_b (S_b ^ T_b >> k): shifts and bit operations at the
// full issue rate, where the byte rows need v_alignbyte + a zero-byte test + v_dot4 per word and diagonal.
// Query: q_rec[q][1] = {meta, planes}.  Entry: e_planes[e] = {ent_meta, planes}.
constexpr int kSymbolPlanesMaxA = 61;
__host__ __device__ inline void symbol_planes16(const uint8_t* row, uint32_t len, uint32_t (&w)[3]) {
  uint32_t p[6] = {0u, 0u, 0u, 0u, 0u, 0u};
  for (uint32_t i = 0; i < 16u && i < len; ++i) {
    const uint32_t v = (uint32_t)row[i] + 1u;
    for (uint32_t b = 0; b < 6u; ++b) p[b] |= ((v >> b) & 1u) << i;
  }
  w[0] = p[0] | p[1] << 16; w[1] = p[2] | p[3] << 16; w[2] = p[4] | p[5] << 16;
}

constexpr int NBITPLANES = 4;            // thermometer planes stored per class / query
constexpr uint32_t SCAN_TQ = 64;         // most queries a tile (= a wave) can hold, compared in passes of 32 (one hit-mask bit each)
constexpr uint32_t SCAN_TQ_DEFAULT = 48; // (until round 6; now host_model.h default_scan_tq: 48 with 7 signature groups, 32 with 8) queries per tile the encoders cut groups into (ANX_SCAN_TQ overrides).  Measured on BASELINE configs[1], three interleaved
                                         // runs each: 64 -> k_scan_bits 1.55-1.57 ms, 56 -> 1.51-1.55, 48 -> 1.51-1.52, 40 -> 1.52 (more, smaller tiles of
                                         // the large groups balance the waves better than the longest-first order alone; configs[3] shares: 32 slightly ahead of 64)
constexpr uint32_t SCAN_HITS = 512;      // per-wave LDS hit list of the scan (entries); expanded when fewer than 256 are free
constexpr uint32_t SCAN_MASKW = 2048;    // stream positions per window of run-end bits in the scan's run staging (stage_runs)
constexpr uint32_t SCAN_PBUF = 128;      // per-wave LDS ring of dense pairs awaiting the fused filter (< 64 waiting + <= 64 new)
constexpr uint32_t SCAN_CHUNK = 256;     // pair slots a wave reserves per global atomic
constexpr uint32_t SCAN_CHUNK_FUSED = 128;  // ... in tiles whose pairs pass the fused prefilter first (a third survives)
constexpr uint32_t SCAN_REGIONS = 64;     // pair-list regions with one reservation counter each
constexpr uint32_t RC_STRIDE = 32;        // uint32 words per region counter block (128 B)
constexpr uint32_t RAW_INVALID = 0xFFFFFFFFu;
constexpr uint32_t RAW_ENTRY_MASK = 0x3FFFFFFFu;   // entry id of a pair (bit 31: exact anagram class, StopAtExactMatch)
constexpr uint32_t RAW_PREFILTERED = 0x40000000u;  // bit 30 of a pair's entry word: the scan applied the band-match bound already (entries < 2^26 then)
constexpr uint32_t META_SKIPPED = 0xFFFFFFFFu;

struct EntRec {   // per-entry attributes k_compact needs, one 16-B gather
  uint32_t vocab, freq, order, meta;
};

struct DevAlphabet {  // tables of the device-side query encoder (encode.hip), device pointers
  int16_t* fast = nullptr;         // [256] per first byte: class of a one-byte member that always matches, -1 none, -2 try the candidates
  uint32_t* coff = nullptr;        // [257] candidate range per first byte
  uint4* cand = nullptr;           // {class, characters, bytes, offset into `bytes`}, (class, member) file order per first byte
  uint8_t* bytes = nullptr;        // member byte pool
  uint8_t* sym_group = nullptr;    // [count-vector bytes] signature group of every symbol slot
  uint2* lower = nullptr;          // inclusive code point ranges of char::is_lowercase
  uint32_t nlower = 0;
  uint32_t* siglen_begin = nullptr;  // [kMaxSymbols + 2] signature range per charcount
};

// What a batch needs from the runtime besides device memory: its events and the pinned block its run reads back into.  Creating and
// destroying them per batch costs more than it looks -- hipHostFree waits for the DEVICE to go idle: a batch freed while the next one
// was running serialised the two (round 5: the step with the encoder inside, 3.56 ms, was encode + run back to back, never
// overlapped).  Freed batches hand theirs to the device's pool (engine.hip DevPool::shells); device_pool_trim releases them.
struct BatchShell {
  hipEvent_t ev[6] = {};
  hipEvent_t ev_scan0 = nullptr, ev_fs0 = nullptr, ev_fs1 = nullptr, ev_done = nullptr, ev_in = nullptr;
  uint32_t* h_read = nullptr;
};
// Sizes of the last finished first run on this replica, per query: the next batch of the same parameters sizes its pair list, its
// scoring grid and its survivor buffers from them (with a margin) instead of from worst-case estimates.  A wrong guess costs what a
// wrong estimate always cost: the run is repeated with the measured sizes (batch_finish).
struct RunHints {
  std::mutex mu;
  bool valid = false;
  anx_threshold kth = {}, dth = {};
  double score_threshold = 0.0;
  int stop = 0;
  double nq = 0, maxfill = 0, surv_fill = 0, list_fill = 0, total_surv = 0;
};
struct DeviceLexicon {
  int device = 0;
  int nplanes = 0;      // count-vector dwords (SAD path)
  int nsym = 0;
  uint32_t nclasses = 0, nentries = 0, cstride = 0, max_len = 0;
  uint32_t* cls_planes = nullptr;  // [nplanes][cstride] packed u8 counts
  uint32_t* cls_bits = nullptr;    // [NBITPLANES][cstride] thermometer planes (bit s of plane t: count_s > t), nsym <= 32
  uint8_t* cls_len = nullptr;      // [cstride]
  uint32_t* cls_off = nullptr;
  uint4* scan_rec = nullptr;       // [E + 1] per entry {plane 1, plane 2 of its class, len, class}: ScanArgs::scan_rec
  uint2* scan_rec34 = nullptr;     // [E + 1] {plane 3, plane 4}
  uint4* sig_e = nullptr;          // [nsig_pad] signature table with entry runs (ScanArgs::sig_e)
  uint4* sighash = nullptr;        // open-addressing table {sig lo, sig hi, first class of the run, classes}; empty slots are 0
  uint4* sighash_e = nullptr;      // the same slots with the run as scan records {.., .., first entry, entries} (bit-plane scan)
  uint32_t hash_mask = 0;
  unsigned long long* ball = nullptr;  // signature offsets (8 x int8) with sum |offset| <= k, for k = 0..12 back to back
  uint32_t* ball_tab = nullptr;    // device copy of ball_off[13] ++ ball_n[13]
  uint32_t ball_off[13] = {}, ball_n[13] = {};  // per k; ball_n = 0: no ball (walk)
  uint4* adj_hash = nullptr;       // adjacency lists (adjacency.h): table {sig lo, sig hi, header index + 1, rows}
  uint32_t adj_mask = 0;           // 0 = no lists
  uint32_t* adj_hdr = nullptr;     // [lists][8] {first row, cumulative rows of the 7 length sections}
  uint2* adj_planes = nullptr;     // [rows * 64] {plane 1, plane 2}
  uint32_t* adj_ids = nullptr;     // [rows * 64] entry ids (padding: nentries)
  uint32_t adj_nhdr = 0;           // lists
  mutable std::vector<AdjSlot> adj_hash_host;  // host copies for the host encoder (ANX_ENCODE=host): adjacency_host_copies, on first use
  mutable std::vector<AdjHdr> adj_hdr_host;
  uint4* sig = nullptr;            // [nsig_pad] signature table (see LexiconImage): {groups 0-3, groups 4-7, first class of the run, classes}
  uint32_t* sig_cbeg = nullptr;    // [nsig_pad+1]
  uint32_t* ent_vocab = nullptr;
  uint32_t* ent_freq = nullptr;
  uint32_t* ent_meta = nullptr;
  uint32_t* ent_rowoff = nullptr;
  uint32_t* ent_order = nullptr;
  EntRec* ent_rec = nullptr;           // {vocab, freq, order, meta} per entry
  uint4* e_rec = nullptr;              // [E][2] {first 16 symbols} {meta, row offset, freq, 0}: PairArgs::e_rec
  uint4* e_planes = nullptr;           // [E] {meta, symbol planes of the first 16 symbols (symbol_planes16)}: PairArgs::e_planes
  uint32_t* ent_var_off = nullptr;     // CSR entry -> VariantOf references (variant lists, src/lib.rs:1677-1727)
  uint32_t* var_target = nullptr;      // vocab id of the reference item
  uint32_t* var_target_freq = nullptr;
  double* var_score = nullptr;
  int any_variants = 0;
  uint4* rows = nullptr;
  DevAlphabet alpha;
  size_t bytes = 0;
  double* quot = nullptr;              // [33][33] IEEE quotients x / L computed on the host (ScoreArgs::quot)
  mutable RunHints hints;
  mutable struct DeviceLm* dlm = nullptr;      // bigram terms + token lists of the vocabulary (lattice.hip), built on first use
  mutable struct DeviceConf* dconf = nullptr;  // confusable patterns + vocabulary texts of this replica (conf.hip), built on first use
};

enum { CTR_SKIPPED = 2, CTR_MAXROWS = 3 /* longest ranked list of the run */, CTR_OVERFLOW = 4 /* survivor records dropped */, CTR_N = 8 };

struct SurvRow {  // one candidate result row of a query (k_compact -> k_rank); 32 B, written / read as two 16-B words
  double score;            // dist_score (times the variant score for expanded rows)
  unsigned long long ord;  // enumeration-order key: ent_order << 20 | position inside the expansion
  uint32_t vocab, freq;    // vocab id, absolute frequency of the row
  uint32_t via, pad;       // vocab id of the variant the row was reached through, 0xFFFFFFFF = none
};
struct SurvRec {   // one scored pair that passed the score threshold (k_score_* -> k_compact), appended per wave
  uint32_t q, e;
  double score;
};
struct DevRow {   // one ranked result row (device) for download / gather
  uint32_t vocab_id, via;
  double dist_score, freq_score;
};

struct Batch {
  int device = 0;
  size_t nq = 0;            // encoded queries
  anx_params params;
  // host side
  std::vector<uint32_t> order;     // sorted position -> original index (downloaded from q_orig on first use)
  std::vector<int32_t> status;     // per original query: 0 ok, ANX_EEMPTY, ANX_ELIMIT
  size_t n_input = 0;
  std::vector<Tile> tiles;         // host encoder only, emptied after the upload
  uint32_t ntiles = 0;             // d_tiles, in launch order: bit-plane tiles, then the count-vector (SAD) ones; each by decreasing cost
  uint32_t n_sad_tiles = 0;
  uint32_t n_adj_tiles = 0;        // the first n_adj_tiles bit-plane tiles stream an adjacency list (k_scan_adj)
  uint32_t qw = 1;                 // uint4 words per query row
  uint32_t dmax = 0;
  uint64_t n_class_tests = 0;
  uint64_t n_tests_kind[NBITPLANES + 1] = {};
  // device: queries
  uint32_t* q_cv = nullptr;        // [nq][nplanes]
  uint32_t* q_bits = nullptr;      // [nq][NBITPLANES]
  uint4* q_rows = nullptr;         // [nq][qw]
  uint4* q_rec = nullptr;          // [nq][2] {first 16 symbols} {meta, 0, 0, 0}: PairArgs::q_rec
  uint32_t* q_meta = nullptr;      // len | k<<8 | d<<16 | first_is_lower<<24
  uint32_t* q_orig = nullptr;      // original index
  Tile* d_tiles = nullptr;
  // device: pipeline
  uint32_t* counters = nullptr;
  uint32_t* rctr = nullptr;        // [SCAN_REGIONS][RC_STRIDE] per-region reservation / statistics counters
  uint32_t region_shift = 0;       // log2(slots per region); raw_cap = SCAN_REGIONS << region_shift
  uint32_t region_fill[SCAN_REGIONS] = {};  // host copy of rctr[r][RC_RAW] after the last run
  uint32_t* qexact = nullptr;      // per query: its exact-anagram class, 0xFFFFFFFF = none (StopAtExactMatch; host lookup)
  uint32_t* qsurv = nullptr;       // per query: pairs with score >= threshold
  uint32_t* soff = nullptr;        // nq+1, exclusive scan of qsurv
  uint32_t* qcur = nullptr;
  uint32_t* qmaxfreq = nullptr;
  void* d_cold = nullptr;          // FsCold of the last launch (k_filter_score's rarely used arguments)
  uint32_t* qpairs = nullptr;      // per query: scored pairs of the last run (only when count_pairs is set)
  bool count_pairs = false;
  // confusable weighting on the device (conf.hip): 0 none, 1 late (after the crop, then re-rank + cutoff), 2 early (before the crop)
  int conf_mode = 0;
  bool keep_text = false;          // keep the inputs' bytes + offsets on the device (conf_mode != 0)
  uint8_t* d_text = nullptr;       // the inputs as uploaded: input i = d_text[d_textoff[i] .. d_textoff[i + 1] - 1)
  uint32_t* d_textoff = nullptr;   // [n_input + 1]
  size_t text_bytes = 0;
  double* cf_weight = nullptr;     // per row slot
  uint32_t* cf_need = nullptr;     // row slots that need an edit script
  uint32_t* cf_ctr = nullptr;      // [0] their number, [1] rows the device could not weight (host fallback)
  void* cf_work = nullptr;         // per-lane working memory of k_conf_script
  uint32_t cf_work_blocks = 0;     // ... for this many one-wave blocks in flight
  uint32_t* cf_sort = nullptr;     // [4][cf_cap] shape keys / list positions before and after the sort; cf_sort_tmp: the sort's scratch
  void* cf_sort_tmp = nullptr;
  size_t cf_sort_tmp_bytes = 0;
  size_t cf_cap = 0;
  bool conf_fallback = false;      // the last run raised cf_ctr[1] (or could not allocate the device working set: conf_skipped)
  bool conf_skipped = false;
  uint32_t* scan_tmp = nullptr;
  uint2* raw = nullptr;            // flat pair list (query, entry | exact<<31), in wave chunks
  double* p_score = nullptr;       // per pair-list slot: score of the pairs that went through a DL kernel
  uint32_t* p_meta = nullptr;      // per pair-list slot: skipped / rejected / ld | samecase<<7 | lcs<<8 | prefix<<16 | suffix<<24
  uint32_t* list8 = nullptr;       // slot lists of the selected pairs the fused kernel leaves to k_score_fast8 / k_score_pairs
  uint32_t* listg = nullptr;
  uint32_t* listw = nullptr;       // wide pairs (a string of 17..32 symbols) awaiting the 8-word prefilter of k_filter_wide
  uint32_t* lctr = nullptr;        // [3][SCAN_REGIONS][RC_STRIDE] their fills
  size_t list_cap = 0;             // slots per region in list8 / listg
  size_t raw_cap = 0;
  SurvRec* surv = nullptr;         // survivor records in SCAN_REGIONS regions of surv_region_cap (order arbitrary)
  uint32_t* sctr = nullptr;        // [SCAN_REGIONS][RC_STRIDE] fill of every survivor region
  size_t surv_region_cap = 0;
  SurvRow* c_rows = nullptr;       // candidate result rows grouped by query (survivors, expanded by variant lists)
  uint32_t* qexpand = nullptr;     // per query: some DL survivor has variant references (has_expandable_variants)
  DevRow* r_rows = nullptr;        // ranked rows, per query at soff[q] .. soff[q] + r_count[q]
  double* t_key = nullptr;
  size_t surv_cap = 0;
  uint32_t* r_count = nullptr;
  uint32_t* r_off = nullptr;       // nq+1
  uint32_t n_raw = 0;
  uint64_t n_sel = 0;
  uint64_t n_pairs = 0, n_surv = 0, n_results = 0;
  mutable uint32_t *x_cnt = nullptr, *x_tmp = nullptr;  // export_compact temporaries (input-order counts, scan sums)
  uint32_t max_rows = 0;           // longest ranked list of the last run (a fixed-stride export needs stride >= max_rows)
  void* last_stream = nullptr;     // stream of the last run (batch_fetch continues on it)
  mutable void* async_stream = nullptr;   // stream the last asynchronous export was launched on
  mutable bool async_pending = false;     // ... and batch_free has to wait for
  bool ran = false;
  bool keep_all_pairs = false;     // also materialise the pairs whose lengths differ by more than d (debug fetch of every pair)
  bool ran_keep_all = false;       // what the last run did
  hipEvent_t ev[6] = {};
  hipEvent_t ev_scan0 = nullptr;   // just before the scan kernels (after the counter memsets)
  hipEvent_t ev_fs0 = nullptr, ev_fs1 = nullptr;  // around k_filter_score
  hipEvent_t ev_done = nullptr;    // after the read-back of a launched run
  hipEvent_t ev_in = nullptr;      // the caller's stream at the time of an asynchronous run on one of the library's streams
  uint32_t* h_read = nullptr;      // pinned: counters of the launched run (engine.hip HR_*)
  bool launched = false;           // a run is enqueued and not yet finished
  uint32_t fill_cap_launched = 0;  // slots per region the scoring grid of the launched run covers
  uint32_t prev_maxfill = 0, prev_surv_fill = 0, prev_list_fill = 0;  // largest fills seen by earlier runs of this batch
  bool hinted = false;             // ... or taken from the replica's RunHints (first run)
  size_t hint_rows = 0;            // candidate rows of the whole batch the hints expect (0: none)
  int runs_finished = 0;
  anx_batch_stats stats = {};
};

// Timing switches that skip parts of the kernels (ANX_SCAN_DBG / ANX_SCORE_DBG / ANX_SCAN_CHUNK: results are WRONG when set) exist
// only in builds with -DANX_DEBUG_SWITCHES (tools/build_variant.sh); the release library compiles them to constants.
#ifdef ANX_DEBUG_SWITCHES
#define ANX_DBG(x) (x)
#else
#define ANX_DBG(x) 0
#endif

typedef const __attribute__((address_space(4))) uint32_t* cptr_u32;  // constant address space: s_load

