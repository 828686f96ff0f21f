/* anx_oracle.c -- CPU ORACLE for the variant-query hot path of proycon/analiticcl.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  It is a plain-C restatement of the reference's CPU
 * algorithm (bigint anagram values, BFS deletion enumeration, per-charcount bucket containment scan
 * with `%`, full-matrix unrestricted Damerau-Levenshtein, naive LCS, f64 score, stable rank), kept
 * deliberately literal so that (a) it is the checker for the HIP path and (b) it can be timed as the
 * "port" CPU baseline.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link,
 * load or call it.  The product library (analiticcl_amd/csrc) never does.
 *
 * The Rust reference cannot be built here (no cargo/rustc), so parity is pinned by:
 *   - the reference's own known-answer tests (tests/main.rs 01xx-04xx), transcribed in
 *     tests/test_oracle_c.py, and
 *   - the recorded outputs of the reference in tutorial.ipynb (tests/golden/tutorial_outputs.json),
 *   - plus a differential test against the independent Python twin (oracle/twin.py).
 * Parity status: PINNED ("Oracle").
 *
 * Third-party arithmetic on the path: ibig 0.3.x (Cargo.toml:21) -- exact unsigned bigint mul/div/rem/cmp;
 * restated below with 32-bit limbs (schoolbook multiply by a limb, Knuth algorithm D remainder).
 *
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 * Build: gcc -O2 -fopenmp -ffp-contract=off -shared -fPIC (oracle/Makefile).
 */
#define _GNU_SOURCE
#include "anx_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "unicode_tables.inc"

static __thread char g_err[256];
const char *orc_last_error(void) { return g_err; }
static void set_err(const char *m) { snprintf(g_err, sizeof g_err, "%s", m); }

/* src/types.rs:20-30 */
static const uint32_t PRIMES[168] = {
    2,   3,   5,   7,   11,  13,  17,  19,  23,  29,  31,  37,  41,  43,  47,  53,  59,  61,  67,  71,  73,
    79,  83,  89,  97,  101, 103, 107, 109, 113, 127, 131, 137, 139, 149, 151, 157, 163, 167, 173, 179, 181,
    191, 193, 197, 199, 211, 223, 227, 229, 233, 239, 241, 251, 257, 263, 269, 271, 277, 281, 283, 293, 307,
    311, 313, 317, 331, 337, 347, 349, 353, 359, 367, 373, 379, 383, 389, 397, 401, 409, 419, 421, 431, 433,
    439, 443, 449, 457, 461, 463, 467, 479, 487, 491, 499, 503, 509, 521, 523, 541, 547, 557, 563, 569, 571,
    577, 587, 593, 599, 601, 607, 613, 617, 619, 631, 641, 643, 647, 653, 659, 661, 673, 677, 683, 691, 701,
    709, 719, 727, 733, 739, 743, 751, 757, 761, 769, 773, 787, 797, 809, 811, 821, 823, 827, 829, 839, 853,
    857, 859, 863, 877, 881, 883, 887, 907, 911, 919, 929, 937, 941, 947, 953, 967, 971, 977, 983, 991, 997};

#define MAX_ANAGRAM_DISTANCE 12 /* src/lib.rs:43 */
#define MAX_EDIT_DISTANCE 12    /* src/lib.rs:46 */

/* ------------------------------------------------------------------------------------------------
 * Unsigned big integers (restating ibig's exact semantics; AnaValue = UBig, src/types.rs:33)
 * ---------------------------------------------------------------------------------------------- */
#define BIG_WORDS 80 /* 2560 bits: 255 symbols of < 2^10 each */
typedef struct {
  uint16_t n; /* significant limbs; value 0 has n == 0 */
  uint32_t w[BIG_WORDS];
} big;

static void big_set(big *a, uint32_t v) {
  a->n = v ? 1 : 0;
  a->w[0] = v;
}
static void big_copy(big *d, const uint32_t *w, int n) {
  d->n = (uint16_t)n;
  memcpy(d->w, w, (size_t)n * 4);
}
static int big_cmp(const uint32_t *a, int an, const uint32_t *b, int bn) {
  if (an != bn) return an < bn ? -1 : 1;
  for (int i = an - 1; i >= 0; i--)
    if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return 0;
}
static int big_is_one(const uint32_t *w, int n) { return n == 1 && w[0] == 1; }
/* a *= m; returns 0 on overflow */
static int big_mul_u32(big *a, uint32_t m) {
  uint64_t carry = 0;
  for (int i = 0; i < a->n; i++) {
    uint64_t t = (uint64_t)a->w[i] * m + carry;
    a->w[i] = (uint32_t)t;
    carry = t >> 32;
  }
  if (carry) {
    if (a->n >= BIG_WORDS) return 0;
    a->w[a->n++] = (uint32_t)carry;
  }
  return 1;
}
/* q = a / d, returns a % d */
static uint32_t big_divmod_u32(const uint32_t *a, int an, uint32_t d, big *q) {
  uint64_t rem = 0;
  for (int i = an - 1; i >= 0; i--) {
    uint64_t cur = (rem << 32) | a[i];
    if (q) q->w[i] = (uint32_t)(cur / d);
    rem = cur % d;
  }
  if (q) {
    int n = an;
    while (n > 0 && q->w[n - 1] == 0) n--;
    q->n = (uint16_t)n;
  }
  return (uint32_t)rem;
}
static uint32_t big_mod_u32(const uint32_t *a, int an, uint32_t d) {
  uint64_t rem = 0;
  for (int i = an - 1; i >= 0; i--) rem = ((rem << 32) | a[i]) % d;
  return (uint32_t)rem;
}
/* (a % b) == 0 for multi-limb b: Knuth TAOCP vol.2 4.3.1 algorithm D, remainder only. a >= b assumed. */
static int big_mod_is_zero(const uint32_t *a, int an, const uint32_t *b, int bn) {
  if (bn == 1) return big_mod_u32(a, an, b[0]) == 0;
  uint32_t u[BIG_WORDS + 1], v[BIG_WORDS];
  int s = __builtin_clz(b[bn - 1]);
  for (int i = bn - 1; i > 0; i--) v[i] = s ? (b[i] << s) | (b[i - 1] >> (32 - s)) : b[i];
  v[0] = b[0] << s;
  u[an] = s ? a[an - 1] >> (32 - s) : 0;
  for (int i = an - 1; i > 0; i--) u[i] = s ? (a[i] << s) | (a[i - 1] >> (32 - s)) : a[i];
  u[0] = a[0] << s;
  for (int j = an - bn; j >= 0; j--) {
    uint64_t num = ((uint64_t)u[j + bn] << 32) | u[j + bn - 1];
    uint64_t qhat = num / v[bn - 1], rhat = num % v[bn - 1];
    while (qhat >= (1ull << 32) || qhat * v[bn - 2] > ((rhat << 32) | u[j + bn - 2])) {
      qhat--;
      rhat += v[bn - 1];
      if (rhat >= (1ull << 32)) break;
    }
    int64_t borrow = 0;
    uint64_t carry = 0;
    for (int i = 0; i < bn; i++) {
      uint64_t p = qhat * v[i] + carry;
      carry = p >> 32;
      int64_t t = (int64_t)u[i + j] - borrow - (int64_t)(p & 0xFFFFFFFFull);
      u[i + j] = (uint32_t)t;
      borrow = t < 0 ? 1 : 0;
    }
    int64_t t = (int64_t)u[j + bn] - borrow - (int64_t)carry;
    u[j + bn] = (uint32_t)t;
    if (t < 0) { /* add back */
      uint64_t c = 0;
      for (int i = 0; i < bn; i++) {
        uint64_t x = (uint64_t)u[i + j] + v[i] + c;
        u[i + j] = (uint32_t)x;
        c = x >> 32;
      }
      u[j + bn] += (uint32_t)c;
    }
  }
  for (int i = 0; i < bn; i++)
    if (u[i]) return 0;
  return 1;
}
static uint64_t big_hash(const uint32_t *w, int n) {
  uint64_t h = 1469598103934665603ull;
  for (int i = 0; i < n; i++) {
    h ^= w[i];
    h *= 1099511628211ull;
    h ^= h >> 29;
  }
  return h;
}
static int big_to_decimal(const uint32_t *w, int n, char *out, int cap) {
  if (n == 0) {
    if (cap < 2) return -1;
    out[0] = '0';
    out[1] = 0;
    return 1;
  }
  big t, q;
  big_copy(&t, w, n);
  char tmp[BIG_WORDS * 10 + 2];
  int len = 0;
  while (t.n > 0) {
    uint32_t r = big_divmod_u32(t.w, t.n, 1000000000u, &q);
    t = q;
    for (int k = 0; k < 9; k++) {
      tmp[len++] = (char)('0' + r % 10);
      r /= 10;
      if (t.n == 0 && r == 0) break;
    }
  }
  while (len > 1 && tmp[len - 1] == '0') len--;
  if (len + 1 > cap) return -1;
  for (int i = 0; i < len; i++) out[i] = tmp[len - 1 - i];
  out[len] = 0;
  return len;
}

/* Anahash trait (src/anahash.rs:139-171) */
/* contains: value > self -> false, else self % value == 0 (src/anahash.rs:165-171) */
static int av_contains(const uint32_t *self, int sn, const uint32_t *val, int vn) {
  if (big_cmp(val, vn, self, sn) > 0) return 0;
  if (vn == 0) return 0;
  return big_mod_is_zero(self, sn, val, vn);
}
/* delete(character(ci)) (src/anahash.rs:156-162 specialised to a single prime, as every call site uses) */
static int av_delete_char(const uint32_t *self, int sn, int charindex, big *out) {
  uint32_t p = PRIMES[charindex];
  if (sn == 0) return 0;
  if (sn == 1 && self[0] < p) return 0;
  if (big_mod_u32(self, sn, p) != 0) return 0;
  big_divmod_u32(self, sn, p, out);
  return 1;
}
static int av_is_empty(const big *a) { return a->n == 0 || big_is_one(a->w, a->n); }

/* ------------------------------------------------------------------------------------------------
 * UTF-8 helpers and Unicode properties
 * ---------------------------------------------------------------------------------------------- */
static int utf8_len(unsigned char c) { return c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 1; }
static uint32_t utf8_decode(const char *s, int *len) {
  const unsigned char *p = (const unsigned char *)s;
  int l = utf8_len(p[0]);
  *len = l;
  switch (l) {
    case 1: return p[0];
    case 2: return ((p[0] & 0x1Fu) << 6) | (p[1] & 0x3Fu);
    case 3: return ((p[0] & 0x0Fu) << 12) | ((p[1] & 0x3Fu) << 6) | (p[2] & 0x3Fu);
    default: return ((p[0] & 0x07u) << 18) | ((p[1] & 0x3Fu) << 12) | ((p[2] & 0x3Fu) << 6) | (p[3] & 0x3Fu);
  }
}
static int in_ranges(const unsigned int (*r)[2], int n, uint32_t cp) {
  int lo = 0, hi = n - 1;
  while (lo <= hi) {
    int mid = (lo + hi) / 2;
    if (cp < r[mid][0]) hi = mid - 1;
    else if (cp > r[mid][1]) lo = mid + 1;
    else return 1;
  }
  return 0;
}
static int first_char_is_lowercase(const char *s) { /* char::is_lowercase on text.chars().next() */
  int l;
  if (!s[0]) return 0;
  return in_ranges(orc_uc_lower, orc_uc_lower_n, utf8_decode(s, &l));
}
static int utf8_count(const char *s, size_t bytes) {
  int n = 0;
  for (size_t i = 0; i < bytes; i += (size_t)utf8_len((unsigned char)s[i])) n++;
  return n;
}

/* ------------------------------------------------------------------------------------------------
 * Model
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  char *s;
  int bytelen, charlen;
} member;
typedef struct {
  member *m;
  int n;
} aclass;

typedef struct { /* VariantReference, src/types.rs:315-324 */
  uint8_t variant_of; /* 1 = VariantOf, 0 = ReferenceFor */
  uint64_t id;
  double score;
} varref;
typedef struct {
  char *text;
  uint8_t *norm;
  uint16_t normlen;
  uint32_t freq;
  uint8_t indexed;
  uint8_t transparent;
  uint8_t has_variants; /* variants.is_some() */
  varref *variants;
  uint32_t nvariants;
} vocab;

typedef struct {
  uint32_t *w;
  uint16_t n;
  uint16_t charcount;
  uint32_t *inst;
  uint32_t ninst, capinst;
} klass;

/* the language model of search mode (anx_oracle_search.inc): LM vocabulary ids and the n-gram counts built from them */
typedef struct { uint32_t a, b; uint64_t count; uint8_t used; } ngslot;
typedef struct {
  uint64_t *lm_ids; size_t nlm, caplm;
  ngslot *tab; size_t cap, n;
  int have_lm;
} orc_lm;
struct orc_model {
  orc_lm lm;
  aclass *alpha;
  int nalpha;
  double w_ld, w_lcs, w_prefix, w_suffix, w_case;
  vocab *voc;
  uint64_t nvoc, capvoc;
  uint64_t *enc; /* open addressing: vocab id + 1 */
  uint64_t enccap;
  int have_freq;
  klass *cls;
  uint64_t ncls, capcls;
  uint64_t *ctab; /* class idx + 1 */
  uint64_t ctabcap;
  uint32_t *bucket[256]; /* sortedindex: class idx ascending by value */
  uint32_t nbucket[256];
  uint32_t *bflat[256];  /* the same values stored contiguously (stride bstride[c] limbs) for the containment scan */
  uint8_t *bflat_n[256];
  uint32_t bstride[256];
  uint64_t ninstances;
};

static uint64_t str_hash(const char *s) {
  uint64_t h = 1469598103934665603ull;
  for (; *s; s++) {
    h ^= (unsigned char)*s;
    h *= 1099511628211ull;
  }
  return h;
}

/* src/lib.rs:369-407 */
static int parse_alphabet(orc_model *m, const char *data) {
  const char *p = data;
  while (*p) {
    const char *e = strchr(p, '\n');
    size_t len = e ? (size_t)(e - p) : strlen(p);
    size_t l2 = len;
    if (l2 > 0 && p[l2 - 1] == '\r') l2--;
    if (l2 > 0) {
      m->alpha = realloc(m->alpha, sizeof(aclass) * (size_t)(m->nalpha + 1));
      aclass *c = &m->alpha[m->nalpha++];
      c->m = NULL;
      c->n = 0;
      size_t i = 0;
      while (i <= l2) {
        size_t j = i;
        while (j < l2 && p[j] != '\t') j++;
        const char *f = p + i;
        size_t fl = j - i;
        char buf[64];
        size_t bl = 0;
        if (fl == 2 && f[0] == '\\' && f[1] == 's') { buf[0] = ' '; bl = 1; }
        else if (fl == 2 && f[0] == '\\' && f[1] == 't') { buf[0] = '\t'; bl = 1; }
        else if (fl == 2 && f[0] == '\\' && f[1] == 'n') { buf[0] = '\n'; bl = 1; }
        else { /* trim() by White_Space, drop if empty */
          size_t b = 0, en = fl;
          while (b < en) {
            int l;
            uint32_t cp = utf8_decode(f + b, &l);
            if (!in_ranges(orc_uc_ws, orc_uc_ws_n, cp)) break;
            b += (size_t)l;
          }
          while (en > b) {
            size_t k = en - 1;
            while (k > b && ((unsigned char)f[k] & 0xC0) == 0x80) k--;
            int l;
            uint32_t cp = utf8_decode(f + k, &l);
            if (!in_ranges(orc_uc_ws, orc_uc_ws_n, cp)) break;
            en = k;
          }
          bl = en - b;
          if (bl >= sizeof buf) { set_err("alphabet member too long"); return 0; }
          memcpy(buf, f + b, bl);
        }
        if (bl > 0) {
          c->m = realloc(c->m, sizeof(member) * (size_t)(c->n + 1));
          member *mm = &c->m[c->n++];
          mm->s = malloc(bl + 1);
          memcpy(mm->s, buf, bl);
          mm->s[bl] = 0;
          mm->bytelen = (int)bl;
          mm->charlen = utf8_count(buf, bl);
        }
        i = j + 1;
      }
    }
    if (!e) break;
    p = e + 1;
  }
  if (m->nalpha + 1 >= 168) { set_err("alphabet too large for PRIMES"); return 0; }
  return 1;
}

static orc_model *model_alloc(void) {
  orc_model *m = calloc(1, sizeof *m);
  m->w_ld = 0.5; /* Weights::default(), src/types.rs:57-67 */
  m->w_lcs = m->w_prefix = m->w_suffix = m->w_case = 0.125;
  m->enccap = 1 << 16;
  m->enc = calloc(m->enccap, sizeof(uint64_t));
  return m;
}
static uint64_t vocab_push(orc_model *m, const char *text, uint32_t freq, int indexed);
static void init_vocab(orc_model *m) { /* src/vocab.rs:145-181 */
  vocab_push(m, "<bos>", 0, 0);
  vocab_push(m, "<eos>", 0, 0);
  vocab_push(m, "<unk>", 0, 0);
}
orc_model *orc_model_new_from_text(const char *tsv) {
  orc_model *m = model_alloc();
  if (!parse_alphabet(m, tsv)) { orc_model_free(m); return NULL; }
  init_vocab(m);
  return m;
}
static char *slurp(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  char *buf = malloc((size_t)sz + 1);
  if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(buf); return NULL; }
  buf[sz] = 0;
  fclose(f);
  return buf;
}
orc_model *orc_model_new(const char *path) {
  char *d = slurp(path);
  if (!d) { set_err("cannot read alphabet file"); return NULL; }
  orc_model *m = orc_model_new_from_text(d);
  free(d);
  return m;
}
void orc_model_free(orc_model *m) {
  if (!m) return;
  for (int i = 0; i < m->nalpha; i++) {
    for (int j = 0; j < m->alpha[i].n; j++) free(m->alpha[i].m[j].s);
    free(m->alpha[i].m);
  }
  free(m->alpha);
  for (uint64_t i = 0; i < m->nvoc; i++) { free(m->voc[i].text); free(m->voc[i].norm); free(m->voc[i].variants); }
  free(m->voc);
  free(m->enc);
  for (uint64_t i = 0; i < m->ncls; i++) { free(m->cls[i].w); free(m->cls[i].inst); }
  free(m->cls);
  free(m->ctab);
  for (int i = 0; i < 256; i++) { free(m->bucket[i]); free(m->bflat[i]); free(m->bflat_n[i]); }
  free(m->lm.lm_ids);
  free(m->lm.tab);
  free(m);
}
void orc_set_weights(orc_model *m, double ld, double lcs, double prefix, double suffix, double casew) {
  m->w_ld = ld; m->w_lcs = lcs; m->w_prefix = prefix; m->w_suffix = suffix; m->w_case = casew;
}
int orc_alphabet_len(const orc_model *m) { return m->nalpha; }

/* Shared scan of anahash()/normalize_to_alphabet() (src/anahash.rs:16-80): class index per consumed
 * position, -1 where nothing matched. Returns count, or -1 if cap exceeded. */
static int encode(const orc_model *m, const char *text, int16_t *out, int cap) {
  size_t tl = strlen(text);
  int n = 0, skip = 0;
  for (size_t pos = 0; pos < tl; pos += (size_t)utf8_len((unsigned char)text[pos])) {
    if (skip > 0) { skip--; continue; }
    int matched = -1;
    for (int c = 0; c < m->nalpha && matched < 0; c++)
      for (int e = 0; e < m->alpha[c].n; e++) {
        const member *mm = &m->alpha[c].m[e];
        if (pos + (size_t)mm->bytelen <= tl && memcmp(text + pos, mm->s, (size_t)mm->bytelen) == 0) {
          matched = c;
          skip = mm->charlen - 1;
          break;
        }
      }
    if (n >= cap) return -1;
    out[n++] = (int16_t)matched;
  }
  return n;
}
#define MAXLEN 255
/* src/anahash.rs:50-80: UNK -> alphabet.len()+1 */
static int normalize(const orc_model *m, const char *text, uint8_t *out, int cap) {
  int16_t tmp[MAXLEN];
  int n = encode(m, text, tmp, cap < MAXLEN ? cap : MAXLEN);
  if (n < 0) return -1;
  for (int i = 0; i < n; i++) out[i] = (uint8_t)(tmp[i] >= 0 ? tmp[i] : m->nalpha + 1);
  return n;
}
/* src/anahash.rs:16-47: UNK -> PRIMES[alphabet.len()] */
static int anahash(const orc_model *m, const char *text, big *h) {
  int16_t tmp[MAXLEN];
  int n = encode(m, text, tmp, MAXLEN);
  if (n < 0) return 0;
  big_set(h, 1);
  for (int i = 0; i < n; i++)
    if (!big_mul_u32(h, PRIMES[tmp[i] >= 0 ? tmp[i] : m->nalpha])) return 0;
  return 1;
}
int orc_normalize(const orc_model *m, const char *text, uint8_t *out, int cap) { return normalize(m, text, out, cap); }
int orc_anahash_decimal(const orc_model *m, const char *text, char *out, int cap) {
  big h;
  if (!anahash(m, text, &h)) return -1;
  return big_to_decimal(h.w, h.n, out, cap);
}
int orc_contains(const orc_model *m, const char *a, const char *b) {
  big x, y;
  if (!anahash(m, a, &x) || !anahash(m, b, &y)) return -1;
  return av_contains(x.w, x.n, y.w, y.n);
}

/* -- vocabulary (src/lib.rs:900-967, src/vocab.rs) ------------------------------------------------ */
static void enc_insert(orc_model *m, uint64_t id) {
  uint64_t mask = m->enccap - 1, h = str_hash(m->voc[id].text) & mask;
  while (m->enc[h]) h = (h + 1) & mask;
  m->enc[h] = id + 1;
}
static int64_t enc_find(const orc_model *m, const char *text) {
  uint64_t mask = m->enccap - 1, h = str_hash(text) & mask;
  while (m->enc[h]) {
    if (strcmp(m->voc[m->enc[h] - 1].text, text) == 0) return (int64_t)m->enc[h] - 1;
    h = (h + 1) & mask;
  }
  return -1;
}
static uint64_t vocab_push(orc_model *m, const char *text, uint32_t freq, int indexed) {
  if (m->nvoc == m->capvoc) {
    m->capvoc = m->capvoc ? m->capvoc * 2 : 1024;
    m->voc = realloc(m->voc, m->capvoc * sizeof(vocab));
  }
  if ((m->nvoc + 1) * 2 > m->enccap) {
    m->enccap *= 2;
    free(m->enc);
    m->enc = calloc(m->enccap, sizeof(uint64_t));
    for (uint64_t i = 0; i < m->nvoc; i++) enc_insert(m, i);
  }
  vocab *v = &m->voc[m->nvoc];
  v->text = strdup(text);
  uint8_t tmp[MAXLEN];
  int n = indexed ? normalize(m, text, tmp, MAXLEN) : 0;
  if (n < 0) n = 0;
  v->norm = malloc((size_t)n + 1);
  memcpy(v->norm, tmp, (size_t)n);
  v->normlen = (uint16_t)n;
  v->freq = freq;
  v->indexed = (uint8_t)indexed;
  v->transparent = 0;
  v->has_variants = 0;
  v->variants = NULL;
  v->nvariants = 0;
  enc_insert(m, m->nvoc);
  return m->nvoc++;
}
static uint64_t add_vocab(orc_model *m, const char *text, int has_freq, uint32_t freq, int transparent) {
  uint32_t f = has_freq ? freq : 1;
  int64_t id = enc_find(m, text);
  if (id >= 0) { /* FrequencyHandling::Max (VocabParams::default, src/vocab.rs:121-131) */
    if (f > m->voc[id].freq) m->voc[id].freq = f;
    if (id > 2 && m->voc[id].transparent && !transparent) m->voc[id].transparent = 0; /* src/lib.rs:935-940 */
    return (uint64_t)id;
  }
  uint64_t nid = vocab_push(m, text, f, 1);
  m->voc[nid].transparent = (uint8_t)(transparent != 0);
  return nid;
}
uint64_t orc_add(orc_model *m, const char *text, int has_freq, uint32_t freq) { return add_vocab(m, text, has_freq, freq, 0); }
static void push_varref(vocab *v, int variant_of, uint64_t id, double score) {
  v->variants = realloc(v->variants, (size_t)(v->nvariants + 1) * sizeof(varref));
  v->variants[v->nvariants].variant_of = (uint8_t)variant_of;
  v->variants[v->nvariants].id = id;
  v->variants[v->nvariants].score = score;
  v->nvariants++;
  v->has_variants = 1;
}
/* add_variant + add_variant_by_id (src/lib.rs:460-514), including the reference's duplicate checks as written
 * (the VariantOf side compares the stored reference id with `variantid`) */
int orc_add_variant(orc_model *m, uint64_t ref_id, const char *variant, double score, int has_freq, uint32_t freq,
                    int transparent) {
  uint64_t variantid = add_vocab(m, variant, has_freq, freq, transparent);
  if (variantid == ref_id) return 0;
  vocab *r = &m->voc[ref_id];
  int dup = 0;
  for (uint32_t i = 0; i < r->nvariants; i++)
    if (!r->variants[i].variant_of && r->variants[i].id == variantid) dup = 1;
  if (!dup) push_varref(r, 0, variantid, score);
  vocab *v = &m->voc[variantid];
  dup = 0;
  for (uint32_t i = 0; i < v->nvariants; i++)
    if (v->variants[i].variant_of && v->variants[i].id == variantid) dup = 1;
  if (!dup) push_varref(v, 1, ref_id, score);
  return 1;
}
int orc_read_variants(orc_model *m, const char *path, int transparent) {
  char *d = slurp(path);
  if (!d) { set_err("cannot read variant list"); return -1; }
  int has_freq = -1; /* None */
  char *p = d;
  while (*p) {
    char *e = strchr(p, '\n');
    if (e) *e = 0;
    size_t len = strlen(p);
    if (len > 0 && p[len - 1] == '\r') p[--len] = 0;
    if (len > 0) {
      char *fields[4096];
      int nf = 0;
      for (char *f = p; f && nf < 4096;) {
        fields[nf++] = f;
        char *tab = strchr(f, '\t');
        if (tab) { *tab = 0; f = tab + 1; } else f = NULL;
      }
      int havef = 0;
      uint32_t freq = 0;
      if (has_freq < 0) {
        if (nf >= 2 && (nf - 2) % 3 == 0) {
          char *endp;
          unsigned long long v = strtoull(fields[1], &endp, 10);
          if (fields[1][0] && !*endp && fields[1][0] != '-' && fields[1][0] != '+' && v <= 0xFFFFFFFFull) {
            has_freq = 1; havef = 1; freq = (uint32_t)v;
          }
        } else has_freq = 0;
      } else if (has_freq == 1) { havef = 1; freq = (uint32_t)strtoul(fields[1], NULL, 10); }
      uint64_t ref_id = add_vocab(m, fields[0], havef, freq, 0);
      if (has_freq == 1) {
        for (int i = 2; i + 2 < nf; i += 3)
          orc_add_variant(m, ref_id, fields[i], strtod(fields[i + 1], NULL), 1, (uint32_t)strtoul(fields[i + 2], NULL, 10), transparent);
      } else {
        for (int i = 1; i + 1 < nf; i += 2) orc_add_variant(m, ref_id, fields[i], strtod(fields[i + 1], NULL), 0, 0, transparent);
      }
    }
    if (!e) break;
    p = e + 1;
  }
  free(d);
  return 0;
}
/* src/lib.rs:519-568 with VocabParams::default(): text column 0, freq column 1 (missing -> "1") */
int orc_read_lexicon(orc_model *m, const char *path) {
  char *d = slurp(path);
  if (!d) { set_err("cannot read lexicon file"); return -1; }
  char *p = d;
  while (*p) {
    char *e = strchr(p, '\n');
    if (e) *e = 0;
    size_t len = strlen(p);
    if (len > 0 && p[len - 1] == '\r') p[--len] = 0;
    if (len > 0) {
      char *tab = strchr(p, '\t');
      uint32_t freq = 1;
      if (tab) {
        *tab = 0;
        char *f = tab + 1, *tab2 = strchr(f, '\t');
        if (tab2) *tab2 = 0;
        freq = (uint32_t)strtoul(f, NULL, 10);
      }
      m->have_freq = 1;
      orc_add(m, p, 1, freq);
    }
    if (!e) break;
    p = e + 1;
  }
  free(d);
  return 0;
}
uint64_t orc_vocab_size(const orc_model *m) { return m->nvoc; }
const char *orc_vocab_text(const orc_model *m, uint64_t id) { return id < m->nvoc ? m->voc[id].text : NULL; }

/* -- index (src/lib.rs:192-245, src/index.rs) ----------------------------------------------------- */
static int64_t cls_find(const orc_model *m, const uint32_t *w, int n) {
  if (!m->ctab) return -1;
  uint64_t mask = m->ctabcap - 1, h = big_hash(w, n) & mask;
  while (m->ctab[h]) {
    const klass *k = &m->cls[m->ctab[h] - 1];
    if (k->n == n && memcmp(k->w, w, (size_t)n * 4) == 0) return (int64_t)m->ctab[h] - 1;
    h = (h + 1) & mask;
  }
  return -1;
}
static void ctab_insert(orc_model *m, uint64_t idx) {
  uint64_t mask = m->ctabcap - 1, h = big_hash(m->cls[idx].w, m->cls[idx].n) & mask;
  while (m->ctab[h]) h = (h + 1) & mask;
  m->ctab[h] = idx + 1;
}
/* char_count (src/anahash.rs:108-110) = number of prime factors */
static int char_count(const orc_model *m, const big *v) {
  big t = *v, q;
  int count = 0;
  for (int ci = m->nalpha; ci >= 0; ci--)
    while (av_delete_char(t.w, t.n, ci, &q)) { t = q; count++; }
  return count;
}
static const orc_model *g_sort_model;
static int cmp_cls(const void *a, const void *b) {
  const klass *x = &g_sort_model->cls[*(const uint32_t *)a], *y = &g_sort_model->cls[*(const uint32_t *)b];
  return big_cmp(x->w, x->n, y->w, y->n);
}
static void lm_build(orc_model *m);
void orc_build(orc_model *m) {
  for (uint64_t i = 0; i < m->ncls; i++) { free(m->cls[i].w); free(m->cls[i].inst); }
  m->ncls = 0;
  free(m->ctab);
  m->ctabcap = 1 << 12;
  while (m->ctabcap < m->nvoc * 2 + 16) m->ctabcap *= 2;
  m->ctab = calloc(m->ctabcap, sizeof(uint64_t));
  m->ninstances = 0;
  for (uint64_t id = 0; id < m->nvoc; id++) {
    if (!m->voc[id].indexed) continue;
    big h;
    if (!anahash(m, m->voc[id].text, &h)) continue;
    int64_t ci = cls_find(m, h.w, h.n);
    if (ci < 0) {
      if (m->ncls == m->capcls) {
        m->capcls = m->capcls ? m->capcls * 2 : 1024;
        m->cls = realloc(m->cls, m->capcls * sizeof(klass));
      }
      klass *k = &m->cls[m->ncls];
      k->w = malloc((size_t)(h.n ? h.n : 1) * 4);
      memcpy(k->w, h.w, (size_t)h.n * 4);
      k->n = h.n;
      k->charcount = (uint16_t)char_count(m, &h);
      k->inst = NULL;
      k->ninst = k->capinst = 0;
      ci = (int64_t)m->ncls++;
      ctab_insert(m, (uint64_t)ci);
    }
    klass *k = &m->cls[ci];
    if (k->ninst == k->capinst) {
      k->capinst = k->capinst ? k->capinst * 2 : 2;
      k->inst = realloc(k->inst, k->capinst * sizeof(uint32_t));
    }
    k->inst[k->ninst++] = (uint32_t)id;
    m->ninstances++;
  }
  for (int c = 0; c < 256; c++) { free(m->bucket[c]); m->bucket[c] = NULL; m->nbucket[c] = 0; }
  for (uint64_t i = 0; i < m->ncls; i++) m->nbucket[m->cls[i].charcount & 255]++;
  for (int c = 0; c < 256; c++) {
    if (m->nbucket[c]) m->bucket[c] = malloc(m->nbucket[c] * sizeof(uint32_t));
    m->nbucket[c] = 0;
  }
  for (uint64_t i = 0; i < m->ncls; i++) {
    int c = m->cls[i].charcount & 255;
    m->bucket[c][m->nbucket[c]++] = (uint32_t)i;
  }
  g_sort_model = m;
  for (int c = 0; c < 256; c++) {
    free(m->bflat[c]); free(m->bflat_n[c]);
    m->bflat[c] = NULL; m->bflat_n[c] = NULL; m->bstride[c] = 0;
    if (!m->nbucket[c]) continue;
    qsort(m->bucket[c], m->nbucket[c], sizeof(uint32_t), cmp_cls);
    uint32_t stride = 1;
    for (uint32_t i = 0; i < m->nbucket[c]; i++) if (m->cls[m->bucket[c][i]].n > stride) stride = m->cls[m->bucket[c][i]].n;
    m->bstride[c] = stride;
    m->bflat[c] = calloc((size_t)m->nbucket[c] * stride, 4);
    m->bflat_n[c] = malloc(m->nbucket[c]);
    for (uint32_t i = 0; i < m->nbucket[c]; i++) {
      const klass *k = &m->cls[m->bucket[c][i]];
      memcpy(m->bflat[c] + (size_t)i * stride, k->w, (size_t)k->n * 4);
      m->bflat_n[c][i] = (uint8_t)k->n;
    }
  }
  lm_build(m);  /* the n-gram counts of the LM vocabulary (src/lib.rs:252-277) */
}
uint64_t orc_n_classes(const orc_model *m) { return m->ncls; }
uint64_t orc_n_instances(const orc_model *m) { return m->ninstances; }
uint64_t orc_bucket_size(const orc_model *m, int c) { return (c >= 0 && c < 256) ? m->nbucket[c] : 0; }
int orc_anagram_instances(const orc_model *m, const char *text, char *out, int cap) { /* src/lib.rs:305-318 */
  big h;
  int n = 0, pos = 0;
  out[0] = 0;
  if (!anahash(m, text, &h)) return -1;
  int64_t ci = cls_find(m, h.w, h.n);
  if (ci < 0) return 0;
  for (uint32_t i = 0; i < m->cls[ci].ninst; i++) {
    const char *t = m->voc[m->cls[ci].inst[i]].text;
    int l = (int)strlen(t);
    if (pos + l + 2 > cap) return -1;
    memcpy(out + pos, t, (size_t)l);
    pos += l;
    out[pos++] = '\n';
    out[pos] = 0;
    n++;
  }
  return n;
}
int orc_has(const orc_model *m, const char *text) { /* src/lib.rs:331-338 */
  big h;
  if (!anahash(m, text, &h)) return 0;
  int64_t ci = cls_find(m, h.w, h.n);
  if (ci < 0) return 0;
  for (uint32_t i = 0; i < m->cls[ci].ninst; i++)
    if (strcmp(m->voc[m->cls[ci].inst[i]].text, text) == 0) return 1;
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Iterators (src/iterators.rs)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  big value;
  uint8_t charindex;
  uint32_t depth;
} qnode;
typedef struct { /* VecDeque<(DeletionResult,u32)> */
  qnode *a;
  size_t head, len, cap;
} dq;
static void dq_grow(dq *q) {
  size_t ncap = q->cap ? q->cap * 2 : 64;
  qnode *na = malloc(ncap * sizeof(qnode));
  for (size_t i = 0; i < q->len; i++) na[i] = q->a[(q->head + i) % q->cap];
  free(q->a);
  q->a = na;
  q->head = 0;
  q->cap = ncap;
}
static void dq_push_back(dq *q, const qnode *n) {
  if (q->len == q->cap) dq_grow(q);
  q->a[(q->head + q->len) % q->cap] = *n;
  q->len++;
}
static int dq_pop_front(dq *q, qnode *out) {
  if (!q->len) return 0;
  *out = q->a[q->head];
  q->head = (q->head + 1) % q->cap;
  q->len--;
  return 1;
}
static int dq_pop_back(dq *q, qnode *out) {
  if (!q->len) return 0;
  *out = q->a[(q->head + q->len - 1) % q->cap];
  q->len--;
  return 1;
}
typedef struct { /* HashSet<AnaValue> */
  big *items;
  size_t n, cap;
  uint32_t *tab;
  size_t tabcap;
} bigset;
static int bigset_contains(const bigset *s, const big *v) {
  if (!s->tabcap) return 0;
  size_t mask = s->tabcap - 1, h = big_hash(v->w, v->n) & mask;
  while (s->tab[h]) {
    const big *x = &s->items[s->tab[h] - 1];
    if (x->n == v->n && memcmp(x->w, v->w, (size_t)v->n * 4) == 0) return 1;
    h = (h + 1) & mask;
  }
  return 0;
}
static void bigset_insert(bigset *s, const big *v) {
  if (bigset_contains(s, v)) return;
  if (s->n == s->cap) {
    s->cap = s->cap ? s->cap * 2 : 64;
    s->items = realloc(s->items, s->cap * sizeof(big));
  }
  s->items[s->n++] = *v;
  if (s->n * 2 > s->tabcap) {
    s->tabcap = s->tabcap ? s->tabcap * 2 : 256;
    free(s->tab);
    s->tab = calloc(s->tabcap, sizeof(uint32_t));
    for (size_t i = 0; i < s->n; i++) {
      size_t mask = s->tabcap - 1, h = big_hash(s->items[i].w, s->items[i].n) & mask;
      while (s->tab[h]) h = (h + 1) & mask;
      s->tab[h] = (uint32_t)i + 1;
    }
  } else {
    size_t mask = s->tabcap - 1, h = big_hash(v->w, v->n) & mask;
    while (s->tab[h]) h = (h + 1) & mask;
    s->tab[h] = (uint32_t)s->n;
  }
}
static void bigset_free(bigset *s) { free(s->items); free(s->tab); }

/* RecurseDeletionIterator (src/iterators.rs:95-235) */
typedef struct {
  dq queue;
  int alphabet_size, singlebeam, breadthfirst, unique, empty_leaves;
  uint32_t mindepth;
  int has_max;
  uint32_t maxdepth;
  bigset visited;
} rdi;
static void rdi_init(rdi *it, const big *value, int alphabet_size, int singlebeam, int mindepth, int maxdepth,
                     int breadthfirst, int unique, int empty_leaves) {
  memset(it, 0, sizeof *it);
  qnode n;
  n.value = *value;
  n.charindex = 0;
  n.depth = 0;
  dq_push_back(&it->queue, &n);
  it->alphabet_size = alphabet_size;
  it->singlebeam = singlebeam;
  it->breadthfirst = breadthfirst;
  it->unique = unique;
  it->empty_leaves = empty_leaves;
  it->mindepth = mindepth < 0 ? 1u : (uint32_t)mindepth;
  it->has_max = maxdepth >= 0;
  it->maxdepth = maxdepth >= 0 ? (uint32_t)maxdepth : 0;
}
static void rdi_free(rdi *it) { free(it->queue.a); bigset_free(&it->visited); }
/* DeletionIterator::next (src/iterators.rs:51-70), resumable: *iter is the iteration counter */
static int deletion_next(const big *value, int alphabet_size, int *iter, qnode *child) {
  if (big_is_one(value->w, value->n)) return 0;
  while (*iter < alphabet_size) {
    int ci = alphabet_size - *iter - 1;
    (*iter)++;
    if (ci < 168 && av_delete_char(value->w, value->n, ci, &child->value)) {
      child->charindex = (uint8_t)ci;
      return 1;
    }
  }
  return 0;
}
static int rdi_next(rdi *it, qnode *out) {
  for (;;) {
    qnode node;
    if (it->breadthfirst) { /* src/iterators.rs:154-187 */
      if (!dq_pop_front(&it->queue, &node)) return 0;
      if (it->unique && bigset_contains(&it->visited, &node.value)) continue;
      if (!it->has_max || node.depth < it->maxdepth) {
        int iter = 0;
        qnode child;
        while (deletion_next(&node.value, it->alphabet_size, &iter, &child)) {
          if (it->unique && bigset_contains(&it->visited, &child.value)) continue;
          child.depth = node.depth + 1;
          dq_push_back(&it->queue, &child);
        }
      }
    } else { /* src/iterators.rs:188-234 */
      if (!dq_pop_back(&it->queue, &node)) return 0;
      if (!it->has_max || node.depth < it->maxdepth) {
        if (it->unique && bigset_contains(&it->visited, &node.value)) continue;
        int iter = 0;
        qnode child;
        if (it->singlebeam) {
          if (deletion_next(&node.value, it->alphabet_size, &iter, &child)) {
            child.depth = node.depth + 1;
            dq_push_back(&it->queue, &child);
          }
        } else {
          qnode *kids = NULL;
          int nk = 0, capk = 0;
          while (deletion_next(&node.value, it->alphabet_size, &iter, &child)) {
            if (nk == capk) { capk = capk ? capk * 2 : 16; kids = realloc(kids, (size_t)capk * sizeof(qnode)); }
            child.depth = node.depth + 1;
            kids[nk++] = child;
          }
          for (int i = nk - 1; i >= 0; i--) {
            if (it->unique && bigset_contains(&it->visited, &kids[i].value)) continue;
            dq_push_back(&it->queue, &kids[i]);
          }
          free(kids);
        }
      }
    }
    if (node.depth < it->mindepth || (!it->empty_leaves && av_is_empty(&node.value))) continue;
    if (it->unique) bigset_insert(&it->visited, &node.value);
    *out = node;
    return 1;
  }
}
static int emit_node(const qnode *n, char *out, int cap, int *pos) {
  char dec[BIG_WORDS * 10 + 2];
  if (big_to_decimal(n->value.w, n->value.n, dec, (int)sizeof dec) < 0) return 0;
  int w = snprintf(out + *pos, (size_t)(cap - *pos), "%s %u %u\n", dec, n->depth, (unsigned)n->charindex);
  if (w < 0 || w >= cap - *pos) return 0;
  *pos += w;
  return 1;
}
int orc_iter_parents(const orc_model *m, const char *text, int alphabet_size, char *out, int cap) {
  big h;
  if (!anahash(m, text, &h)) return -1;
  int iter = 0, n = 0, pos = 0;
  qnode child;
  out[0] = 0;
  while (deletion_next(&h, alphabet_size, &iter, &child)) {
    child.depth = 1;
    if (!emit_node(&child, out, cap, &pos)) return -1;
    n++;
  }
  return n;
}
int orc_iter_recursive(const orc_model *m, const char *text, int alphabet_size, int singlebeam, int mindepth,
                       int maxdepth, int breadthfirst, int unique, int empty_leaves, int max_items, char *out,
                       int cap) {
  big h;
  if (!anahash(m, text, &h)) return -1;
  rdi it;
  rdi_init(&it, &h, alphabet_size, singlebeam, mindepth, maxdepth, breadthfirst, unique, empty_leaves);
  qnode node;
  int n = 0, pos = 0;
  out[0] = 0;
  while ((max_items <= 0 || n < max_items) && rdi_next(&it, &node)) {
    if (!emit_node(&node, out, cap, &pos)) { rdi_free(&it); return -1; }
    n++;
  }
  rdi_free(&it);
  return n;
}
/* alphabet_upper_bound (src/anahash.rs:126-136) via iter() = single-beam DFS (:192-204) */
static void upper_bound(const big *v, int alphabet_size, int *maxci, int *count) {
  rdi it;
  rdi_init(&it, v, alphabet_size, 1, -1, -1, 0, 0, 1);
  qnode node;
  *maxci = 0;
  *count = 0;
  while (rdi_next(&it, &node)) {
    (*count)++;
    if (node.charindex > *maxci) *maxci = node.charindex;
  }
  rdi_free(&it);
}
int orc_upper_bound(const orc_model *m, const char *text, int alphabet_size, int *maxci, int *count) {
  big h;
  if (!anahash(m, text, &h)) return -1;
  upper_bound(&h, alphabet_size, maxci, count);
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Distances (src/distance.rs)
 * ---------------------------------------------------------------------------------------------- */
int orc_levenshtein(const uint8_t *a, int la, const uint8_t *b, int lb, int maxd) { /* :7-82 */
  if (la == lb && memcmp(a, b, (size_t)la) == 0) return 0;
  if (la == 0) return lb > maxd ? -1 : lb;
  else if (la > lb && la - lb > maxd) return -1;
  if (lb == 0) return la > maxd ? -1 : la;
  else if (lb > la && lb - la > maxd) return -1;
  size_t cache[MAXLEN + 1];
  for (int i = 0; i < la; i++) cache[i] = (size_t)i + 1;
  size_t result = 0;
  for (int ib = 0; ib < lb; ib++) {
    result = (size_t)ib;
    size_t da = (size_t)ib;
    for (int ia = 0; ia < la; ia++) {
      size_t db = a[ia] == b[ib] ? da : da + 1;
      da = cache[ia];
      if (da > result) result = db > result ? result + 1 : db;
      else if (db > da) result = da + 1;
      else result = db;
      cache[ia] = result;
    }
  }
  return result > (size_t)maxd ? -1 : (int)result;
}
static size_t min4(size_t a, size_t b, size_t c, size_t d) {
  size_t m = a < b ? a : b;
  m = m < c ? m : c;
  return m < d ? m : d;
}
/* damerau_levenshtein (src/distance.rs:101-179): unrestricted DL, full (len_s+2)x(len_t+2) matrix,
 * char_map = last row per symbol, db = last matching column in the row. Returns -1 for None. */
int orc_damerau_levenshtein(const uint8_t *s, int len_s, const uint8_t *t, int len_t, int maxd) {
  if (len_s == 0) return len_t > maxd ? -1 : len_t;
  else if (len_s > len_t && len_s - len_t > maxd) return -1;
  if (len_t == 0) return len_s > maxd ? -1 : len_s;
  else if (len_t > len_s && len_t - len_s > maxd) return -1;
  size_t ub = (size_t)(len_t + len_s);
  int W = len_t + 2;
  size_t *mat = calloc((size_t)(len_s + 2) * (size_t)W, sizeof(size_t));
#define M(i, j) mat[(size_t)(i) * (size_t)W + (size_t)(j)]
  M(0, 0) = ub;
  for (int i = 0; i < len_s + 1; i++) { M(i + 1, 0) = ub; M(i + 1, 1) = (size_t)i; }
  for (int i = 0; i < len_t + 1; i++) { M(0, i + 1) = ub; M(1, i + 1) = (size_t)i; }
  uint8_t char_map[256];
  memset(char_map, 0, sizeof char_map);
  for (int i0 = 0; i0 < len_s; i0++) {
    size_t db = 0;
    size_t i = (size_t)i0 + 1;
    for (int j0 = 0; j0 < len_t; j0++) {
      size_t j = (size_t)j0 + 1;
      size_t last = char_map[t[j0]];
      size_t cost = s[i0] == t[j0] ? 0 : 1;
      M(i + 1, j + 1) = min4(M(i + 1, j) + 1, M(i, j + 1) + 1, M(i, j) + cost,
                             M(last, db) + (i - last - 1) + 1 + (j - db - 1));
      if (cost == 0) db = j;
    }
    char_map[s[i0]] = (uint8_t)i;
  }
  size_t result = M(len_s + 1, len_t + 1);
#undef M
  free(mat);
  return result > (size_t)maxd ? -1 : (int)result;
}
int orc_lcs(const uint8_t *s1, int n1, const uint8_t *s2, int n2) { /* :181-205 */
  int lcs = 0;
  for (int i = 0; i < n1; i++)
    for (int j = 0; j < n2; j++)
      if (s1[i] == s2[j]) {
        int tmp = 1, ti = i + 1, tj = j + 1;
        while (ti < n1 && tj < n2 && s1[ti] == s2[tj]) { tmp++; ti++; tj++; }
        if (tmp > lcs) lcs = tmp;
      }
  return lcs;
}
int orc_prefix(const uint8_t *s1, int n1, const uint8_t *s2, int n2) { /* :208-218 */
  int n = 0, m = n1 < n2 ? n1 : n2;
  for (int i = 0; i < m; i++) {
    if (s1[i] == s2[i]) n++;
    else break;
  }
  return n;
}
int orc_suffix(const uint8_t *s1, int n1, const uint8_t *s2, int n2) { /* :221-231 */
  int n = 0, m = n1 < n2 ? n1 : n2;
  for (int i = 0; i < m; i++) {
    if (s1[n1 - i - 1] == s2[n2 - i - 1]) n++;
    else break;
  }
  return n;
}

/* ------------------------------------------------------------------------------------------------
 * Query pipeline (src/lib.rs:972-1653)
 * ---------------------------------------------------------------------------------------------- */
/* src/lib.rs:982-994 / :1000-1012. `as u8` saturates. */
int orc_clamp_threshold(orc_threshold th, int len, int absmax) {
  if (th.kind == ORC_RATIO || th.kind == ORC_RATIO_WITH_LIMIT) {
    float v = floorf((float)len * th.ratio);
    int x = v < 0.0f ? 0 : v > 255.0f ? 255 : (int)v;
    int lim = th.kind == ORC_RATIO ? absmax : th.value;
    return x < lim ? x : lim;
  }
  int half = (int)floor((double)len / 2.0);
  if (half > 255) half = 255;
  return th.value < half ? th.value : half;
}
typedef struct {
  uint32_t *a;
  size_t n, cap;
} u32vec;
static void u32vec_push(u32vec *v, uint32_t x) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 64; v->a = realloc(v->a, v->cap * sizeof(uint32_t)); }
  v->a[v->n++] = x;
}
typedef struct {
  big *a;
  size_t n, cap;
} bigvec;
static void bigvec_push(bigvec *v, const big *x) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 8; v->a = realloc(v->a, v->cap * sizeof(big)); }
  v->a[v->n++] = *x;
}
static int cmp_cls_r(const void *a, const void *b, void *arg) {
  const orc_model *m = arg;
  const klass *x = &m->cls[*(const uint32_t *)a], *y = &m->cls[*(const uint32_t *)b];
  return big_cmp(x->w, x->n, y->w, y->n);
}
/* find_nearest_anahashes (src/lib.rs:1143-1308), literal. Output: class indices ascending by value, unique. */
static void find_nearest(const orc_model *m, const big *focus, int max_distance, int stop_at_exact, u32vec *nearest) {
  int64_t ci = cls_find(m, focus->w, focus->n);
  if (ci >= 0) {
    u32vec_push(nearest, (uint32_t)ci);
    if (stop_at_exact && m->cls[ci].ninst > 0) return; /* :1164-1173 */
  }
  int ub, focus_charcount;
  upper_bound(focus, m->nalpha + 1, &ub, &focus_charcount); /* :1176 */
  int focus_alphabet_size = ub + 1;
  bigvec lookups[256];
  memset(lookups, 0, sizeof lookups);
  for (int distance = 1; distance <= max_distance; distance++) /* :1187-1200 */
    bigvec_push(&lookups[(focus_charcount + distance) & 255], focus);
  rdi it; /* :1202-1213: bfs, unique, no empty leaves, maxdepth = max_distance */
  rdi_init(&it, focus, focus_alphabet_size + 1, 0, -1, max_distance, 1, 1, 0);
  qnode node;
  while (rdi_next(&it, &node)) { /* :1217-1261 */
    int64_t di = cls_find(m, node.value.w, node.value.n);
    if (di >= 0) u32vec_push(nearest, (uint32_t)di);
    int deletion_charcount = focus_charcount - (int)node.depth;
    for (int sd = 1; sd <= max_distance - (int)node.depth; sd++)
      bigvec_push(&lookups[(deletion_charcount + sd) & 255], &node.value);
  }
  rdi_free(&it);
  for (int cc = 0; cc < 256; cc++) { /* :1268-1281 the containment scan */
    if (!lookups[cc].n) continue;
    const uint32_t stride = m->bstride[cc];
    for (uint32_t bi = 0; bi < m->nbucket[cc]; bi++) {
      const uint32_t *cw = m->bflat[cc] + (size_t)bi * stride;
      const int cn = m->bflat_n[cc][bi];
      for (size_t a = 0; a < lookups[cc].n; a++)
        if (av_contains(cw, cn, lookups[cc].a[a].w, lookups[cc].a[a].n)) {
          u32vec_push(nearest, m->bucket[cc][bi]);
          break;
        }
    }
    free(lookups[cc].a);
  }
  /* BTreeSet<&AnaValue>: ascending, unique */
  qsort_r(nearest->a, nearest->n, sizeof(uint32_t), cmp_cls_r, (void *)m);
  size_t w = 0;
  for (size_t i = 0; i < nearest->n; i++)
    if (w == 0 || nearest->a[w - 1] != nearest->a[i]) nearest->a[w++] = nearest->a[i];
  nearest->n = w;
}
int orc_find_nearest(const orc_model *m, const char *text, int max_distance, int stop_at_exact, char *out, int cap) {
  big h;
  if (!anahash(m, text, &h)) return -1;
  u32vec nearest = {0};
  find_nearest(m, &h, max_distance, stop_at_exact, &nearest);
  int pos = 0;
  out[0] = 0;
  for (size_t i = 0; i < nearest.n; i++) {
    char dec[BIG_WORDS * 10 + 2];
    big_to_decimal(m->cls[nearest.a[i]].w, m->cls[nearest.a[i]].n, dec, (int)sizeof dec);
    int w = snprintf(out + pos, (size_t)(cap - pos), "%s\n", dec);
    if (w < 0 || w >= cap - pos) { free(nearest.a); return -1; }
    pos += w;
  }
  int n = (int)nearest.n;
  free(nearest.a);
  return n;
}

static float f32(float x) { return x; }
/* VariantResult::score (src/types.rs:335-341) */
static double vr_score(const orc_result *r, float fw) {
  if (fw == 0.0f) return r->dist_score;
  return (r->dist_score + ((double)f32(fw) * r->freq_score)) / (1.0 + (double)f32(fw));
}
/* rank_cmp (src/types.rs:344-365): <0 if a ranks before b */
static int rank_cmp(const orc_result *a, const orc_result *b, float fw) {
  if (fw > 0.0f) {
    double sa = vr_score(a, fw), sb = vr_score(b, fw);
    return sb < sa ? -1 : sb > sa ? 1 : 0;
  }
  if (a->dist_score > b->dist_score) return -1;
  if (a->dist_score < b->dist_score) return 1;
  if (a->freq_score > b->freq_score) return -1;
  if (a->freq_score < b->freq_score) return 1;
  return 0;
}
/* slice::sort_by is a stable sort (src/lib.rs:1667-1669): bottom-up merge sort */
static void stable_sort(orc_result *a, size_t n, float fw) {
  if (n < 2) return;
  orc_result *tmp = malloc(n * sizeof *tmp), *src = a, *dst = tmp;
  for (size_t width = 1; width < n; width *= 2) {
    for (size_t lo = 0; lo < n; lo += 2 * width) {
      size_t mid = lo + width < n ? lo + width : n, hi = lo + 2 * width < n ? lo + 2 * width : n;
      size_t i = lo, j = mid, k = lo;
      while (i < mid && j < hi) dst[k++] = rank_cmp(&src[j], &src[i], fw) < 0 ? src[j++] : src[i++];
      while (i < mid) dst[k++] = src[i++];
      while (j < hi) dst[k++] = src[j++];
    }
    orc_result *t = src;
    src = dst;
    dst = t;
  }
  if (src != a) memcpy(a, src, n * sizeof *a);
  free(tmp);
}

int orc_find_variants(const orc_model *m, const char *text, const orc_params *p, orc_result *out, int cap,
                      orc_pair *pairs, int *n_pairs, int *n_classes) {
  if (m->ncls == 0) { set_err("model has not been built"); return 0; } /* :973-976 */
  uint8_t q[MAXLEN];
  int lq = normalize(m, text, q, MAXLEN); /* :979 */
  big h;
  if (lq < 0 || !anahash(m, text, &h)) { set_err("input too long"); return -1; }
  if (lq == 0) { set_err("empty input (reference panics: assert!(input_length > 0), src/lib.rs:1420)"); return -1; }
  int k = orc_clamp_threshold(p->max_anagram_distance, lq, MAX_ANAGRAM_DISTANCE); /* :982-994 */
  u32vec nearest = {0};
  find_nearest(m, &h, k, p->stop_at_exact_match, &nearest); /* :997 */
  int d = orc_clamp_threshold(p->max_edit_distance, lq, MAX_EDIT_DISTANCE); /* :1000-1012 */
  if (n_classes) *n_classes = (int)nearest.n;

  /* gather_instances (:1311-1402) fused with the scoring loop of score_and_rank (:1430-1503) */
  int q_lower = first_char_is_lowercase(text);
  int pair_cap = n_pairs ? *n_pairs : 0, np = 0;
  orc_result *res = NULL;
  size_t nres = 0, capres = 0;
  double max_freq = 0.0;
  int has_expandable = 0;
  double weights_sum = m->w_ld + m->w_lcs + m->w_prefix + m->w_suffix + m->w_case; /* src/types.rs:69-73 */
  for (size_t a = 0; a < nearest.n; a++) {
    const klass *kl = &m->cls[nearest.a[a]];
    for (uint32_t ii = 0; ii < kl->ninst; ii++) {
      uint32_t vid = kl->inst[ii];
      const vocab *v = &m->voc[vid];
      int ld = orc_damerau_levenshtein(q, lq, v->norm, v->normlen, d);
      int lcs = 0, pre = 0, suf = 0, samecase = 1;
      if (ld >= 0) {
        if (m->w_lcs > 0.0) lcs = orc_lcs(q, lq, v->norm, v->normlen);
        if (m->w_prefix > 0.0) pre = orc_prefix(q, lq, v->norm, v->normlen);
        if (m->w_suffix > 0.0) suf = orc_suffix(q, lq, v->norm, v->normlen);
        if (m->w_case > 0.0) samecase = first_char_is_lowercase(v->text) == q_lower;
      }
      if (pairs && np < pair_cap) {
        orc_pair *pp = &pairs[np];
        pp->vocab_id = vid;
        pp->ld = (int16_t)ld;
        pp->lcs = (uint16_t)lcs;
        pp->prefixlen = (uint16_t)pre;
        pp->suffixlen = (uint16_t)suf;
        pp->samecase = (uint8_t)samecase;
      }
      np++;
      if (ld < 0) continue;
      double input_length = (double)lq;
      double distance_score = ld > lq ? 0.0 : 1.0 - ((double)ld / input_length); /* :1433-1437 */
      double lcs_score = (double)lcs / input_length;
      double prefix_score = (double)pre / input_length;
      double suffix_score = (double)suf / input_length;
      double score = (m->w_ld * distance_score + m->w_lcs * lcs_score + m->w_prefix * prefix_score +
                      m->w_suffix * suffix_score + (samecase ? m->w_case : 0.0)) /
                     weights_sum; /* :1443-1452 */
      double freq_score = m->have_freq ? (double)v->freq : 1.0; /* :1454-1459 */
      if (freq_score > max_freq) max_freq = freq_score;
      if (v->has_variants) has_expandable = 1; /* :1464-1466 */
      if (score >= p->score_threshold) { /* :1475 */
        if (nres == capres) { capres = capres ? capres * 2 : 64; res = realloc(res, capres * sizeof *res); }
        res[nres].vocab_id = vid;
        res[nres].dist_score = score;
        res[nres].freq_score = freq_score;
        res[nres].via = UINT64_MAX;
        nres++;
      }
    }
  }
  free(nearest.a);
  if (n_pairs) *n_pairs = np;
  if (has_expandable) { /* expand_variants :1510-1518, :1677-1727 */
    orc_result *ex = NULL;
    size_t nex = 0, capex = 0;
    for (size_t i = 0; i < nres; i++) {
      const vocab *it = &m->voc[res[i].vocab_id];
      for (uint32_t j = 0; j < it->nvariants; j++) {
        if (!it->variants[j].variant_of) continue;
        if (nex == capex) { capex = capex ? capex * 2 : 64; ex = realloc(ex, capex * sizeof *ex); }
        double tf = (double)m->voc[it->variants[j].id].freq;
        ex[nex].vocab_id = it->variants[j].id;
        ex[nex].dist_score = res[i].dist_score * it->variants[j].score;
        ex[nex].freq_score = tf < res[i].freq_score ? tf : res[i].freq_score;
        ex[nex].via = res[i].vocab_id;
        nex++;
      }
      if (!it->transparent) {
        if (nex == capex) { capex = capex ? capex * 2 : 64; ex = realloc(ex, capex * sizeof *ex); }
        ex[nex++] = res[i];
      }
    }
    free(res);
    res = ex;
    nres = nex;
    for (size_t i = 0; i < nres; i++)
      if (res[i].freq_score > max_freq) max_freq = res[i].freq_score;
  }
  if (max_freq > 0.0) /* :1521-1525 */
    for (size_t i = 0; i < nres; i++) res[i].freq_score = res[i].freq_score / max_freq;
  float fw = p->freq_weight;
  stable_sort(res, nres, fw); /* :1528 */
  if (has_expandable) { /* dedup_by_key(vocab_id): consecutive duplicates, first kept :1530-1533 */
    size_t w = 0;
    for (size_t i = 0; i < nres; i++)
      if (w == 0 || res[w - 1].vocab_id != res[i].vocab_id) res[w++] = res[i];
    nres = w;
  }
  size_t mm = (size_t)p->max_matches;
  if (mm > 0 && nres > mm) { /* :1536-1589 */
    double last_score = vr_score(&res[mm - 1], fw), cropped_score = vr_score(&res[mm], fw);
    if (cropped_score < last_score) nres = mm;
    else {
      size_t early = 0, late = 0;
      for (size_t i = 0; i < nres; i++) {
        if (res[i].dist_score == cropped_score && early == 0) early = i;
        if (res[i].dist_score < cropped_score) { late = i; break; }
      }
      if (early > 0) nres = early + 1;
      else if (late > 0) nres = late + 1;
    }
  }
  size_t cutoff = 0; /* :1598-1622 */
  if (p->cutoff_threshold >= 1.0) {
    int have_best = 0;
    double best = 0.0;
    for (size_t i = 0; i < nres; i++) {
      if (have_best) {
        if (vr_score(&res[i], fw) <= best / p->cutoff_threshold) { cutoff = i; break; }
      } else { best = vr_score(&res[i], fw); have_best = 1; }
    }
  }
  if (cutoff > 0) nres = cutoff;
  int ret;
  if ((size_t)cap < nres) { set_err("result capacity too small"); ret = -1; }
  else { memcpy(out, res, nres * sizeof *res); ret = (int)nres; }
  free(res);
  return ret;
}

int orc_find_variants_batch(const orc_model *m, const char *const *texts, size_t n, const orc_params *p,
                            int nthreads, orc_result *out, int stride, int32_t *counts, uint64_t *total_pairs,
                            uint64_t *total_classes) {
  uint64_t tp = 0, tc = 0;
  int err = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : tp, tc) reduction(| : err)
  for (size_t i = 0; i < n; i++) {
    int np = 0, nc = 0;
    int r = orc_find_variants(m, texts[i], p, out + i * (size_t)stride, stride, NULL, &np, &nc);
    if (r < 0) { err |= 1; r = 0; }
    counts[i] = r;
    tp += (uint64_t)np;
    tc += (uint64_t)nc;
  }
  if (total_pairs) *total_pairs = tp;
  if (total_classes) *total_classes = tc;
  return err ? -1 : 0;
}

#include "anx_oracle_search.inc"
