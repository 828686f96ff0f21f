/* anx_oracle.h -- C API of the CPU oracle.  TEST INFRASTRUCTURE ONLY (see anx_oracle.c header).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library. */
#ifndef ANX_ORACLE_H
#define ANX_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_model orc_model;

/* DistanceThreshold, src/types.rs:76-83 */
enum { ORC_ABSOLUTE = 0, ORC_RATIO = 1, ORC_RATIO_WITH_LIMIT = 2 };
typedef struct {
  uint8_t kind;
  uint8_t value; /* Absolute(x) or the limit of RatioWithLimit */
  float ratio;
} orc_threshold;

/* SearchParameters (query-path subset), src/types.rs:112-192 */
typedef struct {
  orc_threshold max_anagram_distance;
  orc_threshold max_edit_distance;
  uint64_t max_matches;
  double score_threshold;
  double cutoff_threshold;
  int32_t stop_at_exact_match;
  float freq_weight;
} orc_params;

/* VariantResult, src/types.rs:326-332 */
typedef struct {
  uint64_t vocab_id;
  double dist_score;
  double freq_score;
  uint64_t via; /* Option<VocabId>: UINT64_MAX = None (set by expand_variants, src/lib.rs:1677-1727) */
} orc_result;

/* One scored (query, candidate) pair: the Distance of src/types.rs:289-305 for every instance on which
 * damerau_levenshtein was invoked (ld = -1 when it returned None). Enumeration order = reference order. */
typedef struct {
  uint64_t vocab_id;
  int16_t ld;
  uint16_t lcs, prefixlen, suffixlen;
  uint8_t samecase;
} orc_pair;

orc_model *orc_model_new(const char *alphabet_path);
orc_model *orc_model_new_from_text(const char *alphabet_tsv);
void orc_model_free(orc_model *);
void orc_set_weights(orc_model *, double ld, double lcs, double prefix, double suffix, double casew);
int orc_alphabet_len(const orc_model *);
/* add_to_vocabulary(text, Some(freq)|None, VocabParams::default()) -- does NOT set have_freq */
uint64_t orc_add(orc_model *, const char *text, int has_freq, uint32_t freq);
/* read_vocabulary(path, VocabParams::default()) -- sets have_freq */
int orc_read_lexicon(orc_model *, const char *path);
/* add_variant(ref_id, variant, score, freq, params[+TRANSPARENT]) src/lib.rs:460-514; returns 1 if linked */
int orc_add_variant(orc_model *, uint64_t ref_id, const char *variant, double score, int has_freq, uint32_t freq,
                    int transparent);
/* read_variants(path, Some(&VocabParams::default()), transparent) src/lib.rs:772-897 */
int orc_read_variants(orc_model *, const char *path, int transparent);
void orc_build(orc_model *);
uint64_t orc_vocab_size(const orc_model *);
const char *orc_vocab_text(const orc_model *, uint64_t id);
uint64_t orc_n_classes(const orc_model *);
uint64_t orc_n_instances(const orc_model *);
uint64_t orc_bucket_size(const orc_model *, int charcount);
int orc_has(const orc_model *, const char *text);
/* texts of get_anagram_instances(text), '\n' joined */
int orc_anagram_instances(const orc_model *, const char *text, char *out, int cap);

/* unit functions (known-answer tests) */
int orc_normalize(const orc_model *, const char *text, uint8_t *out, int cap);
int orc_anahash_decimal(const orc_model *, const char *text, char *out, int cap);
int orc_upper_bound(const orc_model *, const char *text, int alphabet_size, int *maxcharindex, int *count);
int orc_contains(const orc_model *, const char *a, const char *b);
/* writes lines "decimal depth charindex\n"; returns number of items or -1 on overflow */
int orc_iter_parents(const orc_model *, const char *text, int alphabet_size, char *out, int cap);
int orc_iter_recursive(const orc_model *, const char *text, int alphabet_size, int singlebeam, int mindepth,
                       int maxdepth, int breadthfirst, int unique, int empty_leaves, int max_items, char *out,
                       int cap);
int orc_damerau_levenshtein(const uint8_t *s, int ls, const uint8_t *t, int lt, int maxd);
int orc_levenshtein(const uint8_t *s, int ls, const uint8_t *t, int lt, int maxd);
int orc_lcs(const uint8_t *s, int ls, const uint8_t *t, int lt);
int orc_prefix(const uint8_t *s, int ls, const uint8_t *t, int lt);
int orc_suffix(const uint8_t *s, int ls, const uint8_t *t, int lt);
int orc_clamp_threshold(orc_threshold th, int len, int absmax);

/* find_variants: returns number of results written (<= cap) or -1 if cap too small / error.
 * pairs (optional): every scored pair in enumeration order; *n_pairs in: capacity, out: count.
 * n_classes (optional): |find_nearest_anahashes| */
int orc_find_variants(const orc_model *, const char *text, const orc_params *, orc_result *out, int cap,
                      orc_pair *pairs, int *n_pairs, int *n_classes);
/* find_nearest_anahashes only: class keys as decimal strings, '\n' joined, ascending */
int orc_find_nearest(const orc_model *, const char *text, int max_distance, int stop_at_exact, char *out,
                     int cap);
/* Timed batch for the CPU baseline: one task per query (rayon par_iter, src/bin/analiticcl.rs:445-448)
 * via OpenMP dynamic schedule. Fills counts[i] = #results, returns total scored pairs via *total_pairs. */
int orc_find_variants_batch(const orc_model *, const char *const *texts, size_t n, const orc_params *,
                            int nthreads, orc_result *out, int stride, int32_t *counts,
                            uint64_t *total_pairs, uint64_t *total_classes);
/* ---- search mode (anx_oracle_search.inc): find_all_matches, src/lib.rs:1790-1957 (no context rules) -------------------------------- */
typedef struct {
  orc_params base;
  uint8_t max_ngram;
  uint32_t max_seq;
  float lm_weight, variantmodel_weight, contextrules_weight;
} orc_search_params;
typedef struct {
  uint64_t begin, end;          /* byte offsets of the matched text */
  uint32_t n;                   /* tokens spanned */
  int32_t selected;             /* index of the chosen variant, -1 = none */
  uint64_t var_begin, var_end;  /* its variants: rows [var_begin, var_end) */
  int32_t has_variants;         /* 0: variants == None */
} orc_match;
/* add_to_vocabulary(text, freq, VocabParams{vocab_type: LM}), src/lib.rs:900-967 */
uint64_t orc_add_lm(orc_model *, const char *text, int has_freq, uint32_t freq);
/* returns the number of matches (-1: a capacity did not hold); *n_pairs (optional): scored pairs of all find_variants calls made */
int orc_find_all_matches(const orc_model *, const char *text, const orc_search_params *, orc_match *out, int cap, orc_result *rows, int rows_cap, int *n_rows,
                         uint64_t *n_pairs);
/* one OpenMP task per text (the reference's rayon fan-out, src/bin/analiticcl.rs:445-448): the CPU baseline of search mode */
int orc_find_all_matches_batch(const orc_model *, const char *const *texts, size_t n, const orc_search_params *, int nthreads, int32_t *counts, uint64_t *total_matches,
                               uint64_t *total_rows, uint64_t *total_pairs);
const char *orc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
