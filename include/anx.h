/* anx.h -- C ABI of the MI355X-native variant-scoring engine ("anx") for analiticcl.
 *
 * This header is the drop-in boundary for analiticcl's variant-query hot path.  The reference
 * (proycon/analiticcl, Rust) has no FFI of its own; the functions below are what a Rust `extern "C"`
 * block (or cgo / ctypes / N-API) binds in place of the reference's Rust methods.  Each entry point cites
 * the reference interface it replaces (paths relative to the reference tree).  INTEGRATION.md shows the
 * Rust-side binding.
 *
 * Conventions: plain pointers and sizes only; UTF-8, NUL-terminated strings borrowed for the call;
 * every function returning int returns 0 on success and a negative ANX_E* code on failure, with a
 * thread-local message available from anx_last_error(); nothing aborts across the ABI.
 * The compute path is HIP-only (gfx950): there is no CPU fallback -- calls that need the device fail
 * with ANX_ENODEVICE when no GPU is present.
 */
#ifndef ANX_H
#define ANX_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* 1: rounds 1-3.
 * 2: anx_batch_stats grew (n_prefiltered_in_scan, ...) and anx_batch_get_stats(batch, out, struct_size) takes the caller's struct
 *    size as its third argument (the struct only grows at its end; a size below the version-2 struct's first 152 bytes is refused
 *    with ANX_EINVAL: a caller built against ABI 1 passes no size at all), anx_shutdown, asynchronous fetch, length-partitioned
 *    sharding.  A binding compares anx_abi_version() with the ANX_ABI_VERSION it was compiled against before it calls anything else.
 * 3: anx_pipeline_submit_packed NEVER blocks: with `depth` jobs in flight it returns ANX_ELIMIT and submits nothing (until the middle
 *    of ABI 2's life it waited instead -- a binding written against the blocking contract sees failing submits, hence the new
 *    version); anx_batch_encode_packed_device_on (the caller's stream orders the encoder behind the producer of the buffer);
 *    anx_debug_search_stats, anx_debug_small_stats (anx_find_variants_batch's path for small calls); ANX_ADJ_CLOSURE is 0..2 for every builder.  Nothing was removed; every struct of version 2 is unchanged. */
#define ANX_ABI_VERSION 3

enum {
  ANX_OK = 0,
  ANX_EINVAL = -1,    /* bad argument */
  ANX_EIO = -2,       /* file could not be read (reference: io::Result from the readers) */
  ANX_ENOTBUILT = -3, /* find_variants before build() (reference prints an error and returns [], src/lib.rs:973-976) */
  ANX_ENODEVICE = -4, /* no HIP device / HIP runtime error */
  ANX_ELIMIT = -5,    /* input exceeds a documented limit (255 symbols per string, 166 alphabet classes) */
  ANX_EEMPTY = -6     /* empty query (reference panics: assert!(input_length > 0), src/lib.rs:1420) */
};

typedef struct anx_model anx_model; /* VariantModel, src/lib.rs:50-100 */
typedef struct anx_batch anx_batch; /* a batch of encoded queries + its results, resident in HBM */

/* Weights, src/types.rs:40-73 (defaults .5/.125/.125/.125/.125) */
typedef struct anx_weights {
  double ld, lcs, prefix, suffix, casew;
} anx_weights;

/* DistanceThreshold, src/types.rs:76-83 */
enum { ANX_ABSOLUTE = 0, ANX_RATIO = 1, ANX_RATIO_WITH_LIMIT = 2 };
typedef struct anx_threshold {
  uint8_t kind;
  uint8_t value; /* Absolute(value) or the limit of RatioWithLimit(ratio, value) */
  float ratio;
} anx_threshold;

/* SearchParameters, src/types.rs:112-192 (fields used by find_variants) */
typedef struct anx_params {
  anx_threshold max_anagram_distance; /* default Absolute(3) */
  anx_threshold max_edit_distance;    /* default Absolute(3) */
  uint64_t max_matches;               /* default 20; 0 = unlimited */
  double score_threshold;             /* default 0.25 */
  double cutoff_threshold;            /* default 2.0 */
  int32_t stop_at_exact_match;        /* StopCriterion::StopAtExactMatch, src/types.rs:308-313 */
  float freq_weight;                  /* default 0.0 */
} anx_params;

/* VocabParams, src/vocab.rs:108-131 */
enum { ANX_FREQ_SUM = 0, ANX_FREQ_MAX = 1, ANX_FREQ_MIN = 2, ANX_FREQ_REPLACE = 3 };
enum { ANX_VOCAB_NONE = 0, ANX_VOCAB_INDEXED = 1, ANX_VOCAB_LM = 2, ANX_VOCAB_TRANSPARENT = 4 };
typedef struct anx_vocab_params {
  uint8_t text_column;   /* default 0 */
  int16_t freq_column;   /* default 1; -1 = None */
  uint8_t freq_handling; /* default ANX_FREQ_MAX */
  uint8_t vocab_type;    /* default ANX_VOCAB_INDEXED */
} anx_vocab_params;

/* VariantResult, src/types.rs:326-332 */
#define ANX_NO_VIA UINT64_MAX
typedef struct anx_result {
  uint64_t vocab_id;
  double dist_score;
  double freq_score;
  uint64_t via; /* Option<VocabId>; ANX_NO_VIA = None */
} anx_result;

/* One scored (query, candidate) pair -- the reference's Distance (src/types.rs:289-305) for every instance
 * on which damerau_levenshtein is invoked (src/lib.rs:1343); ld = -1 where it returned None.
 * Exposed for parity tests and the "scored pairs" metric. */
typedef struct anx_pair {
  uint32_t query;    /* index into the batch */
  uint32_t vocab_id;
  int16_t ld;
  uint16_t lcs, prefixlen, suffixlen;
  uint8_t samecase;
  uint8_t _pad;
  double score;      /* dist score (src/lib.rs:1443-1452); 0 when ld < 0 */
} anx_pair;

/* Fixed-stride ranked record for device-to-device export / RCCL gather (16 B). */
typedef struct anx_topk_record {
  uint32_t vocab_id; /* UINT32_MAX = empty slot */
  float freq_score;
  double dist_score;
} anx_topk_record;

void anx_default_weights(anx_weights *);           /* Weights::default() */
void anx_default_params(anx_params *);             /* SearchParameters::default() */
void anx_default_vocab_params(anx_vocab_params *); /* VocabParams::default() */
const char *anx_last_error(void);
int anx_last_error_code(void); /* the ANX_E* code that goes with anx_last_error() (entry points that return a pointer set it too) */
int anx_abi_version(void);

/* ---- model construction (host) ------------------------------------------------------------------ */
/* VariantModel::new(alphabet_file, weights, debug), src/lib.rs:104 (+ read_alphabet :369) */
anx_model *anx_model_new(const char *alphabet_path, const anx_weights *weights, int debug);
/* VariantModel::new_with_alphabet, src/lib.rs:132; the alphabet is passed as the TSV text */
anx_model *anx_model_new_with_alphabet(const char *alphabet_tsv, const anx_weights *weights, int debug);
void anx_model_free(anx_model *);
/* read_vocabulary(filename, &VocabParams), src/lib.rs:519 */
int anx_model_read_vocabulary(anx_model *, const char *path, const anx_vocab_params *);
/* add_to_vocabulary(text, Option<u32>, &VocabParams) -> VocabId, src/lib.rs:900. Returns UINT64_MAX on error. */
uint64_t anx_model_add_to_vocabulary(anx_model *, const char *utf8, int has_frequency, uint32_t frequency,
                                     const anx_vocab_params *);
/* add_variant(ref_id, variant, score, Option<u32>, &VocabParams) -> bool, src/lib.rs:460 (+ add_variant_by_id :478).
 * Returns 1 if linked, 0 if variant == reference, negative on error. */
int anx_model_add_variant(anx_model *, uint64_t ref_id, const char *variant_utf8, double score, int has_frequency,
                          uint32_t frequency, const anx_vocab_params *);
/* read_variants(filename, Some(&VocabParams), transparent), src/lib.rs:772: weighted variant / error lists */
int anx_model_read_variants(anx_model *, const char *path, const anx_vocab_params *, int transparent);
/* build(), src/lib.rs:192: anagram classes, sorted secondary index, then the device-resident SoA lexicon.
 * `device` = HIP device ordinal to upload to; -1 = build the host index only (queries then fail with
 * ANX_ENODEVICE until anx_model_to_device succeeds). */
int anx_model_build(anx_model *, int device);
/* On-disk image of a built model: save after anx_model_build; a later run creates the
 * model with the same alphabet and calls anx_model_load_index INSTEAD of read_vocabulary / read_variants / build
 * (device >= 0 also makes it resident, like anx_model_build).  The reference has no counterpart: it rebuilds its index
 * on every start (src/lib.rs:192-245).  Confusables are not part of the image. */
int anx_model_save_index(const anx_model *, const char *path);
int anx_model_load_index(anx_model *, const char *path, int device);
/* The image stores a caller-chosen tag (any UTF-8 string; default empty) describing what it was built from -- e.g. the list of
 * resource files with sizes and content hashes -- so that a front end can tell a stale image from a current one before loading
 * it (`analiticcl_amd` CLI: --index-cache rebuilds when the tag differs).  set: before anx_model_save_index; read: the tag of
 * an image file without loading it, as a malloc'd string (anx_string_free), or NULL + anx_last_error(). */
int anx_model_set_index_tag(anx_model *, const char *tag_utf8);
char *anx_index_read_tag(const char *path);
/* names of the lexicons read so far (VariantModel::lexicons, src/lib.rs:84): bit i of a vocab item's lexindex */
uint64_t anx_model_num_lexicons(const anx_model *);
const char *anx_model_lexicon_name(const anx_model *, uint64_t i);
int anx_model_to_device(anx_model *, int device);
/* Multi-GPU, one host process (the reference's fan-out over inputs inside one process: rayon par_iter,
 * src/bin/analiticcl.rs:445-448, src/lib.rs:1883): replicate the built lexicon on the n listed HIP devices (normally every GPU of
 * the node once; an ordinal may be listed more than once = several replicas on one GPU).  Every batch call below then splits its
 * inputs over the replicas (by length, see anx_batch_shard_info), each part driven by the replica's own host thread on the replica's
 * own stream, and returns the rows in input order -- there is no collective on this path, the rows are consumed on the host.  Calls
 * with fewer than ANX_SHARD_MIN (8192) inputs per replica use fewer replicas.  Replaces the model's previous replicas.
 * anx_model_to_device(m, d) == anx_model_to_devices(m, &d, 1). */
int anx_model_to_devices(anx_model *, const int *devices, int n);
int anx_model_num_replicas(const anx_model *);
int anx_model_replica_device(const anx_model *, int replica); /* -1 when out of range */
/* has(text), src/lib.rs:331; get_vocab(id), src/lib.rs:341 */
int anx_model_has(const anx_model *, const char *utf8);
uint64_t anx_model_vocab_size(const anx_model *);
const char *anx_model_vocab_text(const anx_model *, uint64_t vocab_id);
uint32_t anx_model_vocab_frequency(const anx_model *, uint64_t vocab_id);
uint32_t anx_model_vocab_lexindex(const anx_model *, uint64_t vocab_id);
/* index statistics ("Found N instances / N anagrams / N anagrams of length L", src/lib.rs:211,220,244) */
uint64_t anx_model_num_instances(const anx_model *);
uint64_t anx_model_num_classes(const anx_model *);
uint64_t anx_model_bucket_size(const anx_model *, int charcount);
int anx_model_alphabet_size(const anx_model *); /* alphabet_size(), src/lib.rs:163 (includes UNK) */
/* normalize_to_alphabet, src/anahash.rs:50; returns length or a negative error */
int anx_model_normalize(const anx_model *, const char *utf8, uint8_t *out, int cap);
/* anahash, src/anahash.rs:16, as a decimal string; returns length or a negative error */
int anx_model_anahash(const anx_model *, const char *utf8, char *out, int cap);

/* ---- the hot path: find_variants --------------------------------------------------------------- */
/* VariantModel::find_variants(&self, &str, &SearchParameters) -> Vec<VariantResult>, src/lib.rs:972,
 * for n inputs at once (the reference's callers fan out one call per input: rayon par_iter in
 * src/bin/analiticcl.rs:445-448, src/lib.rs:1883, bindings/python/src/lib.rs:727).
 * Results are CSR: rows of query i are (*out_rows)[(*out_offsets)[i] .. (*out_offsets)[i+1]).
 * Both arrays are allocated by the library; release with anx_results_free. Thread-safe for concurrent
 * read-only use of the model. */
int anx_find_variants_batch(const anx_model *, const char *const *utf8, size_t n, const anx_params *,
                            anx_result **out_rows, size_t **out_offsets);
void anx_results_free(anx_result *rows, size_t *offsets);

/* ---- staged form of the same call (inputs/results resident in HBM between stages) --------------------
 * encode : upload of the input bytes, then ON THE DEVICE: normalisation to the alphabet (src/anahash.rs:16-80), count
 *          vectors / signatures, threshold clamps (src/lib.rs:982-1012), sort by (length, signature), tiles.
 * run    : the device pipeline (candidate scan -> pair list -> scoring -> rank [-> confusable weighting]).
 * fetch  : download + convert.  anx_find_variants_batch == encode + run + fetch + free. */
anx_batch *anx_batch_encode(const anx_model *, const char *const *utf8, size_t n, const anx_params *);
/* the same with the n inputs packed into one buffer, each terminated by a NUL byte (saves building a pointer array
 * in bindings: 1 M Python strings take 0.27 s to marshal one by one, 0.03 s packed).  The buffer goes to the device as it
 * is and the strings are found there (the first n NUL-terminated spans; anything behind them is ignored; fewer than n spans or
 * a buffer that does not end with a NUL byte: ANX_EINVAL).  An input may not contain a NUL byte itself (as in the char** form). */
anx_batch *anx_batch_encode_packed(const anx_model *, const char *blob, size_t blob_len, size_t n, const anx_params *);
/* the same for inputs that already sit in HBM: device_blob is memory of the model's (one) device, laid out like the packed buffer above
 * (every input followed by a NUL byte).  Nothing crosses PCIe; the bytes are copied device to device, so the caller's buffer is free
 * again when the call returns.  Models with several replicas and the host-side rescoring path (ANX_CONFUSABLES=host): ANX_EINVAL.
 * A trailing NUL byte is the caller's responsibility (the host cannot look).
 * Ordering: the encoder reads the buffer on a stream of its own.  anx_batch_encode_packed_device takes no stream, so the buffer must
 * be COMPLETE and visible to the device when the call is made (the work that produced it -- a kernel, an asynchronous copy -- has been
 * waited for: hipStreamSynchronize / hipDeviceSynchronize).  anx_batch_encode_packed_device_on(..., stream) instead orders the
 * encoder behind everything `stream` (a hipStream_t; NULL = the default stream) holds at the time of the call (an event recorded on it
 * that the encoder's stream waits for): the producer needs no host synchronisation. */
anx_batch *anx_batch_encode_packed_device(const anx_model *, const void *device_blob, size_t blob_len, size_t n, const anx_params *);
anx_batch *anx_batch_encode_packed_device_on(const anx_model *, const void *device_blob, size_t blob_len, size_t n, const anx_params *, void *stream);
/* `stream` is a hipStream_t (NULL = the default stream).  A model with several replicas runs every shard of the batch on its
 * replica's own stream: `stream` must then be NULL. */
int anx_batch_run(const anx_model *, anx_batch *, void *stream);
/* The same in two halves.  run_async enqueues the whole pipeline on `stream` and returns (no host round trip inside a run: grids
 * and buffers are sized from the batch's previous run or from estimates, every kernel bounds-checks, one read-back at the end);
 * wait completes it -- in the rare case an estimate did not hold it regrows the buffers and repeats the run synchronously.
 * Batches in flight overlap: the latency-bound tail of one run (compaction, ranking) runs under the scan of the next.  For a
 * single-replica model run_async therefore enqueues on one of two streams of the library's own, alternately, ordered behind what
 * `stream` holds at the time of the call (so one caller stream is enough to get the overlap; ANX_RUN_OVERLAP=0: on `stream` itself);
 * results are read through the calls below after anx_batch_wait.  anx_batch_run == launch on `stream` + wait.  (The reference's counterpart is the rayon fan-out over inputs,
 * src/bin/analiticcl.rs:445-448: independent calls in flight at once.) */
int anx_batch_run_async(const anx_model *, anx_batch *, void *stream);
int anx_batch_wait(const anx_model *, anx_batch *);
int anx_batch_fetch(const anx_batch *, anx_result **out_rows, size_t **out_offsets);
/* The same ranked rows as 16-byte anx_topk_record {vocab_id u32, freq_score f32, dist_score f64} with uint32 offsets[n + 1]: half
 * the bytes of anx_batch_fetch over PCIe (BASELINE configs[1]: 70 MB instead of 141 MB per million queries).  A record has no
 * `via`, so models with variant lists are refused (ANX_EINVAL; so are models with confusables when their lists are rescored on the
 * host -- the ANX_CONFUSABLES=host A/B path, or after a batch fell back to it --: the device-side weighting is served): use
 * anx_batch_fetch.  Rows and offsets live in one cached pinned block: release both with anx_compact_free.
 * anx_compact_to_results writes the anx_result view (via = ANX_NO_VIA) of n_rows records into caller storage. */
int anx_batch_fetch_compact(const anx_batch *, anx_topk_record **out_rows, uint32_t **out_offsets);
void anx_compact_free(anx_topk_record *rows, uint32_t *offsets);
void anx_compact_to_results(const anx_topk_record *rows, size_t n_rows, anx_result *out);
/* The staged calls as an asynchronous pipeline for ONE caller thread (the reference's counterpart: independent find_variants calls in
 * flight on rayon's pool, src/bin/analiticcl.rs:445-448): submit hands over a packed buffer (as anx_batch_encode_packed; it must stay
 * valid until that job's results were returned) and returns at once -- ANX_ELIMIT, nothing submitted, when `depth` jobs are in flight
 * already: a job counts until anx_pipeline_next has returned its results, so the caller takes a result first; three library
 * threads encode, run and download the jobs on separate HIP streams, so the upload + encoding of batch i + 2, the device pipeline of
 * batch i + 1 and the download of batch i overlap; anx_pipeline_next returns the oldest job's ranked rows (compact records, as
 * anx_batch_fetch_compact; release with anx_compact_free) in submission order, or that job's error.  Models with variant lists or
 * host-side confusable rescoring are refused by the fetch stage (use the staged calls).  anx_pipeline_free waits for the jobs in
 * flight and drops their results. */
typedef struct anx_pipeline anx_pipeline;
anx_pipeline *anx_pipeline_new(const anx_model *, int depth /* jobs in flight; <= 0: 6 (three stages, each working on one job with one queued) */);
int anx_pipeline_submit_packed(anx_pipeline *, const char *blob, size_t blob_len, size_t n, const anx_params *);
int anx_pipeline_pending(const anx_pipeline *); /* jobs submitted and not yet returned */
int anx_pipeline_next(anx_pipeline *, anx_topk_record **out_rows, uint32_t **out_offsets, size_t *out_n);
void anx_pipeline_free(anx_pipeline *);
/* every scored pair of the batch (order unspecified within a query) */
int anx_batch_fetch_pairs(const anx_batch *, anx_pair **out_pairs, size_t *out_n);
void anx_pairs_free(anx_pair *);
/* Scored pairs per input (counts[n], malloc'd, release with anx_counts_free): the number of damerau_levenshtein calls the
 * reference's gather_instances makes for that input (src/lib.rs:1311-1402, one per instance of every anagram class
 * find_nearest_anahashes returned; StopAtExactMatch: of the exact class only when it exists, src/lib.rs:1164-1173).
 * Counted by the scan kernel of a PRODUCTION run -- pairs that fail the DL's length test are only counted there, never
 * materialised -- so this is the per-query check of the production pair list (anx_batch_fetch_pairs re-runs with every
 * pair materialised).  Re-runs the batch. */
int anx_batch_pair_counts(anx_batch *, uint32_t **out_counts);
void anx_counts_free(uint32_t *);
/* write fixed-stride ranked records (stride records per query, batch order) into a DEVICE buffer of
 * n*stride*sizeof(anx_topk_record) bytes (e.g. a torch tensor) -- the payload of the multi-GPU gather */
int anx_batch_export_topk(const anx_batch *, void *device_dst, uint32_t stride, void *stream);
/* the same records without padding, into a DEVICE buffer of `capacity` bytes: uint32 offsets[n+1] in input order
 * (padded to a multiple of 16 bytes), then the records of all inputs back to back.  *used = bytes written (known on
 * the host without a device round trip), also set when the call fails with ANX_ELIMIT because capacity is too small;
 * the multi-GPU gather then moves the used bytes only. */
int anx_batch_export_compact(const anx_batch *, void *device_dst, size_t capacity, void *stream, size_t *used);
/* The final top-k gather of a query-sharded job behind the C ABI (what analiticcl_amd/shard.py does for one rank per GPU with RCCL):
 * the compact export (as anx_batch_export_compact) of EVERY shard of a batch of a multi-device model, in ONE buffer on device
 * dst_device.  Shard g's records start at shard_offsets[g] (shard_offsets[0 .. shards], 256-byte aligned sections; may be NULL);
 * a section = u32 offsets[n_g + 1] padded to 16 bytes, then the 16-byte anx_topk_record rows of the shard's inputs IN THE SHARD'S
 * ORDER (anx_batch_shard_info / anx_batch_shard_inputs map them to the call's input indices).  A shard that lives on dst_device is
 * exported in place; the others are exported on their own device and copied device to device (hipMemcpyPeerAsync: over xGMI where
 * the devices see each other, staged by the runtime otherwise), every shard from its replica's own host thread and stream.  Returns
 * when all sections are in place; *used = bytes needed (also when ANX_ELIMIT says capacity is too small).  The reference's
 * counterpart is the collect() of its rayon fan-out (src/bin/analiticcl.rs:445-448). */
int anx_batch_gather_compact(const anx_batch *, int dst_device, void *device_dst, size_t capacity, size_t *shard_offsets, size_t *used);
typedef struct anx_batch_stats {
  uint64_t n_queries;
  uint64_t n_pairs;          /* scored (query,candidate) pairs = DL invocations of the reference */
  uint64_t n_class_tests;    /* (query,class) count-vector tests executed by the scan kernel */
  uint64_t n_results;        /* ranked results returned */
  uint64_t n_scan_blocks;    /* query tiles (= waves) of the scan kernels */
  uint64_t n_tests_kind[5];  /* class tests by scan body: [0] v_sad_u8 count vectors, [T] T thermometer bit planes */
  uint64_t n_pair_slots;     /* pair-list slots written (scored pairs + unused chunk tails) */
  uint64_t n_survivors;      /* pairs with score >= score_threshold */
  float ms_scan, ms_group, ms_score, ms_rank, ms_total; /* HIP-event times of the last run (stages incl. read-backs) */
  float ms_scan_kernel;      /* HIP events directly around the k_scan_bits launch (the dominant kernel) */
  uint64_t n_selected;       /* pairs that passed the prefilter and went through the DL kernels */
  float ms_filter_score_kernel; /* HIP events directly around the k_filter_score launch */
  uint64_t n_prefiltered_in_scan; /* scored pairs the scan's fused expansion tested against the band-match bound itself (their
                                   * survivors are the pair-list slots; the others of n_pairs failed the DL's length test or were
                                   * left to k_filter_score) */
  uint64_t n_conf_scripts;   /* ABI 2: ranked rows whose edit script the device-side confusable weighting computed (the other rows
                              * were screened out: no pattern can match them) */
  uint64_t n_adj_tiles;      /* scan tiles that streamed a signature adjacency list (k_scan_adj) instead of probing their ball themselves */
  uint64_t n_adj_records;    /* records (12 bytes each, padding included) those tiles streamed ... */
  uint64_t n_adj_records_first; /* ... and the share of the first tile of every (length, signature) group: what the batch reads at least once */
} anx_batch_stats;
/* counts summed over the shards of the batch, times of the slowest replica.  struct_size = sizeof(anx_batch_stats) as the CALLER
 * was compiled: the library writes at most that many bytes, so a caller built against an older, shorter struct stays in bounds
 * (the struct only ever grows at its end). */
int anx_batch_get_stats(const anx_batch *, anx_batch_stats *, size_t struct_size);
/* The replicas a batch is spread over.  Default split of a multi-replica call (ANX_SHARD_POLICY=length): the inputs are ordered by
 * (byte length, a cheap function of the signature) and cut into cost-balanced pieces, one per replica, so that a replica owns whole
 * (length, signature) groups: full scan tiles on every device (BASELINE configs[3]); a shard then holds n_inputs scattered inputs,
 * *first_input is its smallest index and anx_batch_shard_inputs returns all of them (ascending; valid until the batch is freed).
 * ANX_SHARD_POLICY=range (and calls whose rows are rescored on the host): shard i holds the consecutive inputs [first_input,
 * first_input + n_inputs) and anx_batch_shard_inputs returns NULL.  Either way every fetch returns rows in the call's input order.
 * Any out pointer may be NULL.  anx_batch_export_topk / _compact need a batch with exactly one shard. */
int anx_batch_num_shards(const anx_batch *);
int anx_batch_shard_info(const anx_batch *, int shard, int *device, size_t *first_input, size_t *n_inputs);
int anx_batch_shard_inputs(const anx_batch *, int shard, const uint32_t **indices);
/* Waits for asynchronous exports of the batch (anx_batch_export_topk / _compact on the caller's stream), then releases it. */
void anx_batch_free(anx_batch *);
/* Device scratch (pair lists, survivor rows: several GB per million queries) comes from a per-device pool that keeps freed
 * blocks for the next batch (up to 96 GB, ANX_POOL_CACHE_MB overrides; the reference has no counterpart: its scratch is
 * Vec storage inside find_variants, src/lib.rs:1311-1402).  This hands the cached blocks back to the driver, e.g. before
 * another library needs the memory. */
void anx_device_pool_trim(int device);
/* Joins the host threads the library keeps between calls (the pool search mode's parallel loops run on; a model's replica threads
 * end with anx_model_free) and releases the output blocks anx_matches_free keeps for the next search call (at most
 * ANX_SEARCH_OUT_CACHE_MB, default 1024).  Call it with no library call in flight -- before dlclose() or at the end of main(); the library also
 * does it from a destructor function when it is unloaded.  The next call that needs the pool starts a fresh one.  After fork() the
 * child starts with no pool (a pthread_atfork handler forgets the parent's), so forked workers may use the library independently;
 * device state (models, batches) is NOT usable across fork(): create models in the child. */
void anx_shutdown(void);

/* A/B, test and tuning switches (DESIGN.md section 8 "Switches").  The library reads them ONCE from the environment when it is first
 * used; a variable set later has no effect.  This sets one at run time: name = the variable's name ("ANX_ENCODE", "ANX_SHARD_MIN",
 * ...), value = what the variable would hold (NULL = unset).  None of them changes results.  ANX_EINVAL: unknown name. */
int anx_debug_set_switch(const char *name, const char *value);
/* Measurement hook (bench.py: live kernel times of the configurations that are not the headline one): while enabled, the launches of
 * "k_conf_script" (confusable weighting) and "k_lattice" (search mode's lattice decoding) are bracketed by HIP events on their launch
 * stream.  anx_debug_kernel_timer(1) clears the totals and starts, (0) stops; anx_debug_kernel_time waits for the recorded launches
 * and returns their summed duration and count (ANX_EINVAL: none recorded).  k_scan_bits / k_filter_score are always timed: anx_batch_stats. */
void anx_debug_kernel_timer(int enable);
int anx_debug_kernel_time(const char *name, double *total_ms, uint64_t *launches);
/* Diagnosis of search mode's output path since the library was loaded: out[0] = calls that ran as several parts, out[1] = of those, the
 * calls whose output arrays were written while later parts were still on the device, out[2] = calls that were eligible for that but
 * had to write at the end (an upper bound did not hold), out[3] = 0 (reserved). */
int anx_debug_search_stats(uint64_t out[4]);
/* The small call: anx_find_variants_batch answers calls of at most 4096 inputs of at most 64 bytes each (single-device models without
 * variant lists, confusables or StopAtExactMatch) through a path of nine launches and one host wait with preallocated buffers (the
 * reference's own granularity: one string per call, src/lib.rs:972; 1 000 per batch, src/bin/analiticcl.rs:416) instead of the batch
 * pipeline; results are identical.  ANX_SMALL=0 switches it off (A/B).  out[0] = calls it answered since the library was loaded,
 * out[1] = calls it handed to the batch pipeline because a fixed capacity did not hold. */
int anx_debug_small_stats(uint64_t out[2]);
/* The length-partitioned split by itself (no device needed): which of n_shards replicas each of the n inputs
 * would go to (out_shard[i] in 0 .. n_shards - 1; see anx_batch_shard_info).  bench.py and the tests use it to build one GPU's share of
 * a larger job (BASELINE configs[3]) on a one-GPU box.  learn_ms (may be NULL): the device times of THOSE shares, measured by the caller
 * one after the other -- fed to the split's per-length cost corrections exactly as a multi-replica run feeds its own shard times, so
 * that the next split is the one a real N-GPU job would see after this call. */
int anx_debug_length_split(const anx_model *, const char *const *utf8, size_t n, const anx_params *, int n_shards, uint8_t *out_shard,
                           const double *learn_ms);
/* Test hooks of the signature adjacency lists (analiticcl_amd/csrc/adjacency.h; no device needed).  anx_debug_signature: the group-sum
 * signature the scan prunes with (one byte per symbol group) of a string.  anx_debug_adjacency builds the lists of the model's lexicon
 * (closure 0..2, budget in bytes) and returns, for each of the n signatures, out_cum[i][8] = {first row of its list in *out_ids, rows
 * of the length sections L-3 .. L+3 cumulated} (all 0xFFFFFFFF: no list) and the lists' entry ids in rows of 64 (padding = number of
 * entries); *out_ids is released with free().  out_stats (may be NULL): uint64[7] {lexicon signatures, closure, lists kept, records, rows, ms, rows an
 * unlimited budget would keep}.
 * tests/test_adjacency_cpu.py compares them with find_nearest_anahashes' candidate set (src/lib.rs:1143-1308) by brute force. */
int anx_debug_signature(const anx_model *, const char *utf8, uint64_t *out_sig);
/* the index's entries (class-major order = the entry ids of the pair list and of the adjacency lists) as vocabulary ids; free() */
int anx_debug_entries(const anx_model *, uint32_t **out_vocab_ids, size_t *n);
int anx_debug_adjacency(const anx_model *, int closure, uint64_t budget_bytes, const uint64_t *sigs, size_t n, uint32_t *out_cum,
                        uint32_t **out_ids, uint64_t *out_stats);
/* The same lists as the model's first replica HOLDS them (built on the device by default, analiticcl_amd/csrc/adjacency.hip): out_cum /
 * out_ids as anx_debug_adjacency (the order of the ids inside a section is the builder's own). */
int anx_debug_adjacency_device(const anx_model *, const uint64_t *sigs, size_t n, uint32_t *out_cum, uint32_t **out_ids);
/* Test hook: the band-match bound the scan's fused filter and k_filter_score apply before damerau_levenshtein (src/distance.rs:101-179)
 * on n (query, candidate) pairs of <= 16 symbols: rows of 16 bytes (alphabet-indexed symbols, the query padded with 0xFE, the
 * candidate with 0xFF), lengths, d <= 3.  form: 0 the scan's (7-bit symbols, wave-uniform d), 1 k_filter_score's (7-bit symbols),
 * 2 the general one.  out_reject[i] = 1: the bound claims damerau_levenshtein(q, c) > d -- tests/test_gpu_switches.py checks that claim
 * against the oracle's DL.  Runs on HIP device `device`. */
int anx_debug_band_bound(int device, const uint8_t *q_rows, const uint8_t *c_rows, const uint8_t *lq, const uint8_t *lc, size_t n,
                         int d, int form, uint8_t *out_reject);

/* ---- output of `analiticcl query` --------------------------------------------------
 * The TSV lines / JSON items of output_matches_as_tsv / output_matches_as_json (src/bin/analiticcl.rs:21-187) for n
 * inputs and their ranked rows as anx_find_variants_batch returns them: one malloc'd UTF-8 buffer (not NUL-safe:
 * use *out_len), released with anx_string_free.  JSON items are numbered from first_seqnr (1 = first of the run, which
 * gets no leading comma).  Scores print like Rust's `{}` for f64 (shortest round-trip digits, no exponent). */
int anx_format_query_output(const anx_model *, const char *const *utf8_inputs, size_t n, const anx_result *rows,
                            const size_t *offsets, float freq_weight, int json, int output_lexmatch,
                            uint64_t first_seqnr, char **out, size_t *out_len);
void anx_string_free(char *);

/* ---- confusables --------------------------------------------------------------
 * add_to_confusables / read_confusablelist / set_confusables_before_pruning: src/lib.rs:446-458, :409-443, :157-159.
 * Patterns are sesdiff edit scripts ("-[y]+[i]", "=[c|k]-[y]+[i]", "^...", "...$"; src/confusables.rs:13-44).  With
 * confusables loaded the ranked rows are weighted ON THE DEVICE as part of anx_batch_run (late: after the crop, then re-rank and
 * cutoff, src/lib.rs:1591-1622; or before the crop when set_confusables_before_pruning was called, :1505-1508), so the exports
 * and anx_batch_fetch_compact work as without confusables.  A batch with an input or candidate beyond the device kernel's fixed
 * working memory (64 code points) is redone with the host-side weighting (same results); the exports are refused for such a batch.
 * The edit script restates sesdiff 0.3.1 / dissimilar (diff-match-patch): parity unpinned beyond tests/main.rs:914-1020. */
int anx_model_add_to_confusables(anx_model *, const char *editscript, double weight);
int anx_model_read_confusablelist(anx_model *, const char *path);
void anx_model_set_confusables_before_pruning(anx_model *);
/* shortest_edit_script(source, target, false, false, false) in sesdiff notation, e.g. "=[hu]-[y]+[i]=[s]" */
int anx_edit_script(const char *source, const char *target, char *out, int cap);
/* compute_confusable_weight(input, candidate), src/lib.rs:1733-1756: the product of the weights of the patterns found in the
 * edit script input -> text of the vocabulary item (1.0 if none).  Host only. */
int anx_model_confusable_weight(const anx_model *, const char *input_utf8, uint64_t vocab_id, double *out_weight);

/* ---- search mode: the main caller of the hot path -------------------------------
 * VariantModel::find_all_matches(&self, text, &SearchParameters) -> Vec<Match>, src/lib.rs:1790, for n texts at once.
 * Host side: boundaries / n-gram windows / redundancy filter (src/search.rs:190-336), lattice decoding and bigram-LM
 * rerank (src/lib.rs:2088-2495, 2580-2674) with context rules (below).  Device side: every segment of one
 * n-gram order, over all texts, is ONE anx_find_variants_batch call. */
typedef struct anx_search_params {
  anx_params base;
  uint8_t max_ngram;          /* default 3 */
  uint32_t max_seq;           /* default 250: candidate sequences taken to the rerank stage */
  float lm_weight;            /* default 1.0 */
  float variantmodel_weight;  /* default 3.0 */
  float contextrules_weight;  /* default 1.0 */
  int32_t unicodeoffsets;     /* offsets in code points instead of UTF-8 bytes */
} anx_search_params;
/* Match, src/search.rs:40-68 */
typedef struct anx_match {
  uint64_t begin, end;        /* offset of the matched text in its input text */
  uint32_t n;                 /* tokens (boundaries) spanned */
  int32_t selected;           /* index of the chosen variant, -1 = none (out-of-vocabulary, copied from the input) */
  uint64_t var_begin, var_end; /* its ranked variants: rows [var_begin, var_end) of the row array */
  uint32_t tag_begin, tag_end; /* Match.tag / Match.seqnr: entries [tag_begin, tag_end) of the tag array */
} anx_match;
/* one (tag, sequence number) of a match, assigned by a context rule: src/search.rs:60-66 */
typedef struct anx_match_tag {
  uint16_t tag;   /* index for anx_model_tag_name */
  uint8_t seqnr;  /* position of the match inside the tagged span */
  uint8_t pad_;
} anx_match_tag;
void anx_default_search_params(anx_search_params *);
/* matches of text i are (*out_matches)[(*out_offsets)[i] .. (*out_offsets)[i+1]); out_tags may be NULL (tags are then
 * not reported); release with anx_matches_free */
int anx_find_all_matches_batch(const anx_model *, const char *const *utf8_texts, size_t n, const anx_search_params *,
                               anx_match **out_matches, size_t **out_offsets, anx_result **out_rows, size_t *out_n_rows,
                               anx_match_tag **out_tags);
void anx_matches_free(anx_match *matches, size_t *offsets, anx_result *rows, anx_match_tag *tags);   /* (the two large arrays are kept for the next call: see anx_shutdown) */
/* the text `analiticcl search` prints for the matches of n texts (src/bin/analiticcl.rs:21-187, 600-630): per match the
 * input slice, its offsets, tags, and the variants with the selected one first.  Needs byte offsets (unicodeoffsets
 * = 0).  *n_matches = matches formatted (JSON items are numbered from first_seqnr). */
int anx_format_search_output(const anx_model *, const char *const *utf8_texts, size_t n, const anx_match *matches,
                             const size_t *offsets, const anx_result *rows, const anx_match_tag *tags,
                             float freq_weight, int json, int output_lexmatch, uint64_t first_seqnr, char **out,
                             size_t *out_len);

/* ---- context rules of search mode -----------------------------------------------------------------------------
 * VariantModel::add_contextrule(pattern, score, tag, tagoffset), src/lib.rs:658-765; read_contextrules, :570-656.
 * pattern: ';'-separated expressions -- a vocabulary word, "?" (anything), "^" (in no lexicon), "@lexicon", "!x"
 * (negation), "a|b" (disjunction), "!(a|b)"; src/search.rs:413-459.  score > 1 favours, < 1 penalises a candidate
 * sequence the pattern occurs in (src/lib.rs:2501-2578).  tags / tagoffsets ("begin:length", either may be empty):
 * n_tags strings, n_tagoffsets strings.  Rules are applied to every candidate sequence of the lattice on the host. */
int anx_model_add_contextrule(anx_model *, const char *pattern, float score, const char *const *tags, size_t n_tags,
                              const char *const *tagoffsets, size_t n_tagoffsets);
int anx_model_read_contextrules(anx_model *, const char *path);
size_t anx_model_num_tags(const anx_model *);
const char *anx_model_tag_name(const anx_model *, size_t index); /* NULL when out of range */

#ifdef __cplusplus
}
#endif
#endif
