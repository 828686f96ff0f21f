// host_model.h -- host side of the anx engine: alphabet, vocabulary, anagram-class index and the
// SoA lexicon image that is uploaded to HBM.  Mirrors the reference's VariantModel for the query path
// (/root/reference/src/lib.rs:50-245, 369-407, 519-568, 900-967; src/anahash.rs:16-80; src/vocab.rs).
#pragma once
#include <cstdint>
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/anx.h"
#include "confusables_core.hpp"

namespace anx {

constexpr int kMaxSymbols = 255;   // CharIndexType = u8 lengths/charcounts (src/types.rs:13)
constexpr int kMaxAlphabet = 166;  // PRIMES has 168 entries (src/types.rs:20-30): classes + UNK + 1
constexpr int kMaxAnagramDistance = 12;  // src/lib.rs:43
constexpr int kMaxEditDistance = 12;     // src/lib.rs:46

struct AlphabetMember {
  std::string bytes;
  int nchars;
};

struct Alphabet {
  std::vector<std::vector<AlphabetMember>> classes;  // file order; first byte-prefix match wins
  int size() const { return (int)classes.size(); }
  // members whose first byte is b, in (class, member) file order: the scan only has to try these
  struct Cand { int16_t cls; const AlphabetMember* m; };
  std::vector<Cand> by_first[256];
  // per first byte: the class when the FIRST candidate is a one-byte member (it always matches), -1 when no member
  // starts with the byte, -2 when the candidates have to be tried (multi-byte members first)
  int16_t fast[256];
  void index();  // (re)builds by_first; called by parse_alphabet
  // Walks `text` exactly like str::anahash / str::normalize_to_alphabet (src/anahash.rs:16-80) and
  // returns, per consumed position, the class index or -1 for an unmatched character.
  // Returns false if more than kMaxSymbols symbols are produced.
  bool scan(const char* text, size_t nbytes, std::vector<int16_t>& out) const;
  // allocation-free variant: writes into out[cap], returns the symbol count or -1 if it exceeds cap
  int scan_into(const char* text, size_t nbytes, int16_t* out, int cap) const;
};

bool parse_alphabet(const std::string& tsv, Alphabet& out, std::string& err);  // src/lib.rs:369-407
// The alphabet flattened for the device-side query encoder (encode.hip), plus the char::is_lowercase ranges
struct EncodeTables {
  int16_t fast[256];            // Alphabet::fast
  uint32_t coff[257];           // candidate range per first byte
  std::vector<uint32_t> cand;   // 4 words per candidate: class, characters, bytes, offset into `bytes`; (class, member) file order
  std::vector<uint8_t> bytes;   // member byte pool
  std::vector<uint32_t> lower;  // 2 words per inclusive code point range
};
void build_encode_tables(const Alphabet& a, EncodeTables& out);
bool first_char_is_lowercase(const char* utf8);  // char::is_lowercase on text.chars().next()
bool is_alphabetic_cp(uint32_t cp);              // char::is_alphabetic (L* + Nl; see oracle/gen_unicode.py)
const uint32_t (*alphabetic_ranges(uint32_t* n))[2];  // the table behind it: sorted inclusive code point ranges
uint32_t utf8_decode_at(const char* s, size_t avail, int* len);
std::string trim_whitespace(const std::string& s);  // str::trim()
// Host threads worth starting: hardware threads, limited by the affinity mask and the cgroup CPU quota (the GPU boxes
// show 256 hardware threads and grant 16 CPUs).
unsigned usable_hw_threads();

// A/B, test and tuning switches.  Read ONCE from the environment (variable names in the comments) when the library is first
// used -- a variable set later has no effect -- and changed at run time only through anx_debug_set_switch (tests, tools).  None of
// them changes results; the part-skipping timing switches exist only in -DANX_DEBUG_SWITCHES builds.
struct Switches {
  int encode_host = 0;       // ANX_ENCODE=host: the threaded host encoder instead of the device one (A/B reference)
  int scan_sad = 0;          // ANX_SCAN=sad: count-vector scan kernel for every tile
  int scan_walk_flat = 0;    // ANX_SCAN_WALK=flat: window walk instead of the hash-probed ball
  int scan_tq = 0;           // ANX_SCAN_TQ=1..64: queries per scan tile (0 = default)
  int scan_adj = 1;          // ANX_SCAN_ADJ=0: no signature adjacency lists (adjacency.h): every tile walks / probes its ball itself (A/B reference)
  int adj_build_host = 0;    // ANX_ADJ_BUILD=host: the lists are built by the host threads and uploaded (adjacency.cpp: the reference of the device builder)
  int adj_closure = 2;       // ANX_ADJ_CLOSURE=0..2: lists for the signatures within this distance of a lexicon signature (models put on a device afterwards)
  long adj_budget_mb = 16384; // ANX_ADJ_MB: most HBM the lists may take per replica (16 GB of 288: every list of the closure of a 1 M-entry lexicon, 12.8 GB)
  int sig_groups = 0;        // ANX_SIG_GROUPS=1..8: signature groups of a model built afterwards (0 = default)
  int prefilter = 1;         // ANX_PREFILTER=0: no SWAR bound, every length-compatible pair goes through the DL
  int score_fast = 1;        // ANX_SCORE_FAST=0: general k_score_pairs for every pair
  int fs_planes = 1;         // ANX_FS_PLANES=0: byte rows instead of symbol planes in k_filter_score (A/B; alphabets beyond 61 classes always take the rows)
  int fs_split = 1;          // ANX_FS_SPLIT=0: the 8-word prefilter of the wide pairs inline in k_filter_score (A/B; until round 6 what batches with long queries ran)
  int fs_b7 = 1;             // ANX_FS_B7=0: general zero test in the prefilter
  int fuse_prefilter = 1;    // ANX_SCAN_FUSE=0: the scan's expansion does not apply the SWAR bound (k_filter_score's phase 1 does)
  long cap_div = 1;          // ANX_CAP_DIV=n: first-run capacity estimates divided by n (regrow-and-repeat path)
  long max_batch = 4l << 20; // ANX_MAX_BATCH: inputs per device batch of anx_find_variants_batch
  int run_overlap = 1;       // ANX_RUN_OVERLAP=0: anx_batch_run_async enqueues on the caller's stream (no library streams: clean per-kernel times for profiling)
  int shard_by_length = 1;   // ANX_SHARD_POLICY=range: consecutive input ranges per replica instead of the length-partitioned split (A/B reference)
  long shard_min = 8192;     // ANX_SHARD_MIN: fewest inputs a replica of a multi-device model gets (smaller calls use fewer replicas)
  int confusables_host = 0;  // ANX_CONFUSABLES=host: confusable weighting on the host threads (A/B reference of the device kernel)
  int lattice_host = 0;      // ANX_LATTICE=host: lattice decoding on the host threads (A/B reference of the device kernel)
  int search_onepass = 1;    // ANX_SEARCH_ONEPASS=0: search mode downloads every ranked row and builds the lattice input on the host (the path until round 4; A/B reference)
  int scan_chunk_fused = 0;  // ANX_SCAN_CHUNK_FUSED=32..1024: pair-list slots a wave of a fused-filter tile reserves per atomic (0 = default)
  int adj_fail = 0;          // ANX_ADJ_FAIL=1 (test hook): the device build of the adjacency lists fails after its allocations: the replica must load without lists
  int small_path = 1;        // ANX_SMALL=0: calls of a few inputs take the batch pipeline like the large ones (A/B reference of small_path.hpp)
  int enc_priority = 1;      // ANX_ENC_PRIORITY=0: the encoder's streams get normal instead of the highest stream priority (A/B)
  int hints = 1;             // ANX_HINTS=0: a batch's first run sizes its grids and buffers from worst-case estimates instead of the last finished batch of the same parameters
  int encode_timing = 0;     // ANX_ENCODE_TIMING, ANX_SEARCH_TIMING: host phase times on stderr
  int search_timing = 0;
  int search_parts = 4;      // ANX_SEARCH_PARTS: parts of a large find_all_matches call in flight at a time (search.cpp)
  long search_parts_min = 2l << 20;  // ANX_SEARCH_PARTS_MIN: bytes of text from which a call is split
  long search_part_bytes = 4l << 20;  // ANX_SEARCH_PART_BYTES: text per part
  int search_prio = 1;               // ANX_SEARCH_PRIO=0: the host pool serves the loops of a call's parts first come, first served; 1: the earlier part first
  int search_early_output = 1;       // ANX_SEARCH_EARLY_OUTPUT=0: a call's output arrays are written when its last part is done (until round 5)
  int search_first_pct = 50;         // ANX_SEARCH_FIRST_PCT: size of a call's FIRST part in percent of an even share: the device idles until the first part's host phase
                                     // is done (same-box best calls 226-247 MB/s with even parts, 246-263 / 243-273 with 40 / 60 %; medians 222 -> 226 / 232)
};
Switches& switches();
// name = the environment variable's name, value = what the variable would hold; false: unknown name
bool set_switch(const char* name, const char* value);

struct VariantRef {  // VariantReference, src/types.rs:315-324
  bool variant_of;  // true = VariantOf((id, score)), false = ReferenceFor((id, score))
  uint64_t id;
  double score;
};

struct VocabEntry {  // VocabValue, src/vocab.rs:8-29
  std::string text;
  std::vector<uint8_t> norm;  // normalize_to_alphabet: UNK = alphabet.len()+1
  uint32_t frequency;
  uint32_t lexindex;
  uint8_t tokencount;
  uint8_t vocabtype;
  bool has_variants = false;         // variants.is_some()
  std::vector<VariantRef> variants;  // src/vocab.rs:23-26
};

// Little-endian arbitrary-precision unsigned integer; only what ordering anagram values needs
// (AnaValue = product of primes, src/anahash.rs:16-47; ordering src/lib.rs:243).
struct BigVal {
  std::vector<uint32_t> w;
  void set_one() { w.assign(1, 1u); }
  void mul_small(uint32_t m);
  int cmp(const BigVal& o) const;
  std::string to_decimal() const;
};

// SoA image of the lexicon, in "class rank" order: classes sorted by (charcount, anagram value ascending)
// = the order of sortedindex (src/lib.rs:222-245) and of BTreeSet<&AnaValue> iteration within a charcount;
// entries sorted by (class rank, vocab id) = the enumeration order of gather_instances (src/lib.rs:1327-1332).
struct LexiconImage {
  int nsym = 0;     // alphabet.len()+1 hash symbols (UNK = alphabet.len())
  int nplanes = 0;  // count-vector dwords per class (4 symbols per dword), padded to a kernel variant
  uint32_t nclasses = 0, nentries = 0;
  std::vector<uint32_t> cls_planes;     // [nplanes][cstride] packed u8 counts
  uint32_t cstride = 0;
  std::vector<uint32_t> cls_bits;       // [4][cstride] thermometer planes: bit s of plane t = (count_s > t); nsym <= 32 only
  std::vector<uint8_t> cls_len;         // [cstride] charcount per class (padding classes: 255)
  std::vector<uint32_t> cls_off;        // CSR class -> entries, nclasses+1
  uint32_t bucket_begin[kMaxSymbols + 2];  // class-rank range per charcount
  std::vector<uint32_t> ent_vocab;      // vocab id
  std::vector<uint32_t> ent_freq;
  std::vector<uint32_t> ent_meta;       // len | first_is_lower<<8 | has_variants<<9 | transparent<<10
  std::vector<uint32_t> ent_var_off;    // CSR entry -> its VariantOf references (nentries+1)
  std::vector<uint32_t> var_target;     // vocab id of the reference item
  std::vector<uint32_t> var_target_freq;
  std::vector<double> var_score;
  bool any_variants = false;
  std::vector<uint32_t> ent_rowoff;     // offset of the token row in 16-byte units
  std::vector<uint32_t> ent_order;      // position in the reference's enumeration order: classes by ascending
                                        // anagram value over ALL charcounts (BTreeSet<&AnaValue>, src/lib.rs:1148),
                                        // then vocab id (src/lib.rs:1327-1332); last key of the ranking order
  std::vector<uint8_t> rows;            // token rows padded to 16-byte multiples with 0xFF
  std::vector<BigVal> cls_value;        // anagram value per class (host only)
  // Signature pruning of the window scan: the count-vector slots are partitioned into kSigGroups groups of about
  // equal total frequency; sig(c) = per-group symbol counts (one byte each).  L1(sig(q), sig(c)) <= L1(cv_q, cv_c),
  // so a class whose signature is further than k from the query's cannot be within anagram distance k.  Classes are
  // stored in (charcount, signature, anagram value) order: all classes of one signature are one contiguous run.
  std::vector<uint8_t> sym_group;       // [nplanes*4] group of each count-vector slot
  uint32_t nsigs = 0;
  std::vector<uint32_t> sig_lo, sig_hi; // [nsig_pad] groups 0-3 / 4-7 packed as bytes; padding = 0xFFFFFFFF
  std::vector<uint32_t> sig_cbeg;       // [nsig_pad+1] first class of the run (padding: nclasses)
  uint32_t siglen_begin[kMaxSymbols + 2];  // signature range per charcount
};
// Round 6, with the signature adjacency lists (no ball walk per tile any more; every kernel alone on the GPU, ms: scan / filter+score /
// device pass): eng.aspell k=3 d=2 1 M queries, 7 groups 1.105 / 0.824 / 2.59 (4.65 G record tests, 60.6 k tiles of <= 48 queries),
// 8 groups 0.913 / 0.812 / 2.39 (2.78 G tests, 106 k tiles of <= 32); nld.aspell d=3: 1.431 / 2.56 / 4.89 -> 1.277 / 2.58 / 4.80;
// the 1 M-entry lexicon of BASELINE configs[3]: 4.39 / 2.96 / 9.32 -> 4.03 / 3.16 / 9.38 (its emptier tiles cost the scoring kernel
// what the scan gains).  So: 8 groups up to kSigGroupsWideMax entries, kSigGroups above.  ANX_SIG_GROUPS overrides.
constexpr int kSigGroupsWide = 8;
constexpr size_t kSigGroupsWideMax = 500000;
inline uint32_t default_scan_tq(int ngroups) { return ngroups >= 8 ? 32u : 48u; }  // queries per scan tile the encoders cut groups into (ANX_SCAN_TQ overrides)
constexpr int kSigGroups = 7;  // measured (round 2, hash-probe walk): eng.aspell k=3 d=2, 1 M queries: 6 groups 5.45 k record tests per
                               // query, 37.8 k tiles, 3.33 ms per step; 7: 4.14 k tests, 56.6 k tiles, 3.22 ms; 8: 2.35 k tests,
                               // 96 k tiles, 3.48 ms.  1 M-entry lexicon, 1.25 M queries: 13.5 / 12.2 / 13.8 ms
uint64_t signature_of(const uint8_t* cv, size_t n, const std::vector<uint8_t>& sym_group);

struct Confusable {  // src/confusables.rs:5-11; one edit-script pattern with '|' options per instruction
  std::vector<char> ops;                               // '=', '+', '-'
  std::vector<std::vector<std::u32string>> options;
  double weight = 1.0;
  bool strictbegin = false, strictend = false;
  // per instruction, for the screen that decides whether an edit script is worth computing (confusables.cpp may_match): the
  // options as ASCII presence bits when every option is one ASCII character (`simple`)
  struct Screen { uint64_t bits[2] = {0, 0}; bool simple = false; };
  std::vector<Screen> screen;
};
std::string edit_script_string(const std::string& source, const std::string& target);  // sesdiff notation, for tests

// Context rules (src/search.rs:338-524): a pattern over the (vocab id, lexicon mask) pairs of a candidate sequence with a
// bonus (> 1) or penalty (< 1) score and optional tags.
struct PatternMatch {  // src/search.rs:339-353
  enum Kind : uint8_t { Vocab, Any, NoLexicon, FromLexicon, Not, Disjunction } kind = Any;
  uint64_t vocab_id = 0;   // Vocab
  uint8_t lexicon = 0;     // FromLexicon
  std::vector<PatternMatch> sub;  // Not (one), Disjunction (any)
  bool matches(uint64_t vocab_id, uint32_t lexindex) const;  // src/search.rs:373-411
};
struct ContextRule {  // src/search.rs:355-364
  std::vector<PatternMatch> pattern;
  float score = 1.0f;
  std::vector<uint16_t> tag;
  std::vector<std::pair<uint8_t, uint8_t>> tagoffset;  // begin, length
};
struct PatternMatchResult {  // src/search.rs:366-371
  float score;
  int32_t tag;  // -1 = None
  uint8_t seqnr;
};

class HostModel {
 public:
  std::vector<ContextRule> context_rules;  // src/lib.rs:82
  std::vector<std::string> tags;           // src/lib.rs:84
  int add_contextrule(const std::string& pattern, float score, const std::vector<std::string>& tag,
                      const std::vector<std::string>& tagoffset, std::string& err);  // src/lib.rs:658-765
  int read_contextrules(const std::string& path, std::string& err);                  // src/lib.rs:570-656
  // test_context_rules (src/lib.rs:2501-2578) over the (vocab id, lexindex) pairs of one candidate sequence
  double test_context_rules(const std::vector<std::pair<uint64_t, uint32_t>>& sequence,
                            std::vector<std::vector<PatternMatchResult>>& results) const;
  Alphabet alphabet;
  anx_weights weights;
  int debug = 0;
  std::vector<VocabEntry> decoder;                       // VocabDecoder
  std::unordered_map<std::string, uint64_t> encoder;     // VocabEncoder
  std::vector<std::string> lexicons;
  bool have_freq = false;
  bool built = false;
  std::vector<Confusable> confusables;
  bool confusables_before_pruning = false;             // set_confusables_before_pruning (src/lib.rs:157)
  int add_to_confusables(const std::string& script, double weight, std::string& err);
  int read_confusablelist(const std::string& path, std::string& err);
  double confusable_weight(const std::string& input, uint64_t candidate) const;
  // the same for the n ranked rows of one input (ids[k] -> out[k]): the input is decoded once, the vocabulary texts come from a
  // decoded copy built on first use
  void confusable_weights(const char* input, size_t len, const uint64_t* ids, size_t n, double* out) const;
  void confusable_weights(const std::string& input, const uint64_t* ids, size_t n, double* out) const { confusable_weights(input.data(), input.size(), ids, n, out); }
  // flattened pattern tables (confusables_core.hpp Patterns): what the host matcher and the device kernel read
  struct ConfTables {
    std::vector<cdiff::FlatConf> conf;
    std::vector<cdiff::FlatOp> ops;
    std::vector<cdiff::FlatOpt> opts;
    std::vector<uint32_t> pool;
  } conf_tables;
  void rebuild_conf_tables();
  cdiff::Patterns conf_patterns() const;
  struct ConfCache;                                   // UTF-32 texts + character sets of the vocabulary (confusables.cpp)
  const ConfCache& conf_vocab() const;                // built on first use, rebuilt when the vocabulary grew
  // its arrays for the device upload: code point pool, offsets [V + 1], cdiff::CharSet [V]
  void conf_vocab_arrays(const uint32_t** pool, size_t* npool, const uint32_t** off, const void** cs, size_t* n) const;
  mutable std::atomic<const ConfCache*> conf_cache{nullptr};      // current copy (readers take no lock and no reference count)
  mutable std::vector<std::shared_ptr<ConfCache>> conf_cache_owned;  // every copy ever published, released with the model
  mutable std::mutex conf_cache_mu;
  // position of every vocabulary item in the reference's gather order (classes by ascending anagram value, then vocab id:
  // LexiconImage::ent_order), UINT32_MAX for items that are not indexed: what the host-side EARLY confusable rescoring sorts by
  // before it weights, because the reference weights its candidates in that order (src/lib.rs:1505-1535)
  const std::vector<uint32_t>& vocab_gather_order() const;
  // published copies: readers take the current one without a lock; a copy is built for one index generation (build_index /
  // load_index bump it) and one vocabulary size, and every copy ever published lives as long as the model
  struct VocabOrder { uint64_t generation; std::vector<uint32_t> order; };
  mutable std::atomic<const VocabOrder*> vocab_order{nullptr};
  mutable std::vector<std::unique_ptr<VocabOrder>> vocab_order_owned;
  std::atomic<uint64_t> index_generation{0};
  bool have_lm = false;
  std::unordered_map<std::string, uint32_t> ngrams;  // LM n-gram counts keyed by the packed vocab ids (src/lib.rs:68-70)
  std::unordered_map<uint64_t, uint32_t> unigrams, bigrams;  // the two orders lm_score_tokens looks up (id, id1 << 32 | id2)
  // into_ngram of every vocabulary item, precomputed (CSR; an item with more than 5 tokens has no tokens): the lattice
  // rerank asks for it once per symbol of each of up to max_seq paths
  std::vector<uint32_t> ngram_off, ngram_ids;
  LexiconImage lex;
  std::unordered_map<std::string, uint32_t> class_of_cv;  // count vector bytes -> class rank

  HostModel();
  int alphabet_size() const { return alphabet.size() + 1; }  // src/lib.rs:163-165
  uint64_t add_to_vocabulary(const char* text, bool has_freq, uint32_t freq, const anx_vocab_params& p,
                             uint8_t lexicon_index);  // src/lib.rs:900-967
  int read_vocabulary(const char* path, const anx_vocab_params& p, std::string& err);  // src/lib.rs:519-568
  int add_variant(uint64_t ref_id, const char* variant, double score, bool has_freq, uint32_t freq,
                  const anx_vocab_params& p, uint8_t lexicon_index);  // src/lib.rs:460-514
  int read_variants(const char* path, const anx_vocab_params& p, bool transparent, std::string& err);  // :772-897
  int build_index(std::string& err);  // src/lib.rs:192-245
  std::string index_tag;  // stored in / read from the index image: what the caller built it from (anx_model_set_index_tag)
  int save_index(const std::string& path, std::string& err) const;  // index_cache.cpp: image of the built model
  int load_index(const std::string& path, std::string& err);        // instead of read_vocabulary + build_index
  bool has(const char* text) const;   // src/lib.rs:331-338
  // encode one string: norm codes (UNK = len+1), hash-class count vector (UNK = len), symbol count
  bool encode(const char* text, std::vector<uint8_t>& norm, std::vector<uint8_t>& cv) const;
  bool anahash(const char* text, BigVal& out) const;
  // language model (src/lib.rs:247-296, 2632-2729)
  bool into_ngram(uint64_t vocab_id, std::vector<uint64_t>& out) const;
  static std::string ngram_key(const uint64_t* ids, size_t n);
  void build_lm();
};

// threshold clamps of find_variants (src/lib.rs:982-994, 1000-1012)
int index_read_tag(const std::string& path, std::string* tag, std::string& err);  // index_cache.cpp
int clamp_threshold(const anx_threshold& t, int len, int absolute_max);

// offsets of the first n NUL-terminated spans of blob[0, len) (n + 1 values); false if there are fewer (the host-side twin of
// k_nul_count / k_nul_emit in encode.hip: used when the host needs the strings itself)
bool packed_offsets(const char* blob, size_t len, size_t n, std::vector<uint32_t>& off);

}  // namespace anx
