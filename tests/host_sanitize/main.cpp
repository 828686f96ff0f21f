// Drives the host side of libanx through its C ABI under ASan / UBSan (no device): usage: host_sanitize <alphabet.tsv>
// <lexicon.tsv> <tmpdir>.  Prints "OK <checks>" and exits 0; any sanitizer report aborts with a non-zero status.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/anx.h"

static int checks = 0;
#define CHECK(c) do { ++checks; if (!(c)) { fprintf(stderr, "CHECK failed line %d: %s (%s)\n", __LINE__, #c, anx_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const std::string alphabet = argv[1], lexicon = argv[2], tmp = argv[3];
  anx_weights w; anx_default_weights(&w);
  anx_vocab_params vp; anx_default_vocab_params(&vp);
  anx_model* m = anx_model_new(alphabet.c_str(), &w, 0);
  CHECK(m != nullptr);
  CHECK(anx_model_read_vocabulary(m, lexicon.c_str(), &vp) == ANX_OK);
  CHECK(anx_model_read_vocabulary(m, (tmp + "/missing.tsv").c_str(), &vp) == ANX_EIO);
  const uint64_t ref = anx_model_add_to_vocabulary(m, "separate", 1, 7, &vp);
  CHECK(ref != UINT64_MAX);
  CHECK(anx_model_add_variant(m, ref, "seperate", 0.9, 0, 0, &vp) == 1);
  CHECK(anx_model_add_variant(m, ref, "separate", 0.9, 0, 0, &vp) == 0);
  { FILE* f = fopen((tmp + "/variants.tsv").c_str(), "w"); fputs("receive\trecieve\t1.0\trecive\t0.8\nbelieve\t10\tbeleive\t0.9\t2\n", f); fclose(f); }
  CHECK(anx_model_read_variants(m, (tmp + "/variants.tsv").c_str(), &vp, 1) == ANX_OK);
  { FILE* f = fopen((tmp + "/conf.tsv").c_str(), "w"); fputs("-[y]+[i]\t1.1\n=[c|k]-[y]+[i]\t1.2\n^-[x]\t0.5\n+[e]$\t0.9\n", f); fclose(f); }
  CHECK(anx_model_read_confusablelist(m, (tmp + "/conf.tsv").c_str()) == ANX_OK);
  CHECK(anx_model_add_to_confusables(m, "-[a]+[e]", 1.05) == ANX_OK);
  CHECK(anx_model_add_to_confusables(m, "bogus", 1.0) != ANX_OK);
  const char* tags[2] = {"a", "b"};
  const char* offs[2] = {"0:1", ":"};
  CHECK(anx_model_add_contextrule(m, "separate; ?; ^", 1.1f, tags, 2, offs, 2) == ANX_OK);
  CHECK(anx_model_add_contextrule(m, "!(separate|receive); @nolexicon", 1.0f, nullptr, 0, nullptr, 0) != ANX_OK);
  CHECK(anx_model_add_contextrule(m, "notinthelexiconzz", 1.0f, nullptr, 0, nullptr, 0) != ANX_OK);
  CHECK(anx_model_num_tags(m) == 2 && strcmp(anx_model_tag_name(m, 1), "b") == 0 && anx_model_tag_name(m, 2) == nullptr);
  CHECK(anx_model_build(m, -1) == ANX_OK);
  CHECK(anx_model_num_classes(m) > 1000 && anx_model_num_instances(m) >= anx_model_num_classes(m));
  CHECK(anx_model_has(m, "separate") == 1 && anx_model_has(m, "zzzzzzzzzzzz") == 0);
  char buf[4096];
  uint8_t norm[300];
  CHECK(anx_model_normalize(m, "separate", norm, sizeof norm) == 8);
  CHECK(anx_model_normalize(m, std::string(300, 'a').c_str(), norm, sizeof norm) < 0);
  CHECK(anx_model_anahash(m, "separate", buf, sizeof buf) > 0);
  CHECK(anx_model_anahash(m, std::string(200, 'z').c_str(), buf, sizeof buf) > 100);
  CHECK(anx_model_anahash(m, "abc", buf, 2) < 0);
  // edit scripts (confusables.cpp: diff-match-patch restatement) over neighbouring vocabulary items
  const uint64_t nv = anx_model_vocab_size(m);
  for (uint64_t i = 3; i + 1 < nv && i < 6000; ++i) {
    const int n = anx_edit_script(anx_model_vocab_text(m, i), anx_model_vocab_text(m, i + 1), buf, sizeof buf);
    CHECK(n > 0);
  }
  CHECK(anx_edit_script("", "abc", buf, sizeof buf) > 0 && anx_edit_script("abc", "", buf, sizeof buf) > 0);
  CHECK(anx_edit_script("h\xc3\xa9llo w\xc3\xb6rld", "hello world", buf, sizeof buf) > 0);
  CHECK(anx_edit_script("abc", "abd", buf, 3) < 0);
  // index image round trip
  const std::string img = tmp + "/model.idx";
  CHECK(anx_model_save_index(m, img.c_str()) == ANX_OK);
  anx_model* m2 = anx_model_new(alphabet.c_str(), &w, 0);
  CHECK(m2 != nullptr);
  CHECK(anx_model_load_index(m2, img.c_str(), -1) == ANX_OK);
  CHECK(anx_model_num_classes(m2) == anx_model_num_classes(m) && anx_model_vocab_size(m2) == nv);
  CHECK(anx_model_num_lexicons(m2) == anx_model_num_lexicons(m));
  { FILE* f = fopen((tmp + "/trunc.idx").c_str(), "w"); fputs("ANXnot an image", f); fclose(f); }
  anx_model* m3 = anx_model_new(alphabet.c_str(), &w, 0);
  CHECK(anx_model_load_index(m3, (tmp + "/trunc.idx").c_str(), -1) != ANX_OK);
  // the query path must fail loudly without a device
  const char* q[2] = {"seperate", ""};
  anx_params p; anx_default_params(&p);
  anx_result* rows = nullptr; size_t* ro = nullptr;
  CHECK(anx_find_variants_batch(m, q, 2, &p, &rows, &ro) == ANX_ENODEVICE);
  // packed buffers: the host-side offset scan (confusables are loaded, so the host needs the strings for rescoring)
  {
    const char packed[] = "seperate\0\0\x01x\0acommodate\0longer than eight bytes\0";  // sizeof counts the terminator too
    CHECK(anx_batch_encode_packed(m, packed, sizeof packed - 1, 5, &p) == nullptr && strstr(anx_last_error(), "stub") != nullptr);
    CHECK(anx_batch_encode_packed(m, packed, sizeof packed - 1, 6, &p) == nullptr && strstr(anx_last_error(), "fewer strings") != nullptr);
    CHECK(anx_batch_encode_packed(m, packed, sizeof packed - 2, 2, &p) == nullptr && strstr(anx_last_error(), "must end with a NUL") != nullptr);
    CHECK(anx_batch_encode_packed(m, packed, 0, 0, &p) == nullptr);
  }
  anx_search_params sp; anx_default_search_params(&sp);
  anx_match* ms = nullptr; size_t* mo = nullptr; size_t nr = 0; anx_match_tag* tg = nullptr;
  CHECK(anx_find_all_matches_batch(m, q, 2, &sp, &ms, &mo, &rows, &nr, &tg) != ANX_OK);
  // formatters on hand-made rows
  anx_result r[3] = {{ref, 0.734375, 1.0, ANX_NO_VIA}, {ref, 1.0, 0.5, ref}, {nv + 5, 1e-9, 0.0, ANX_NO_VIA}};
  size_t o3[3] = {0, 2, 3};
  char* out = nullptr; size_t outlen = 0;
  for (int js = 0; js < 2; ++js)
    for (int lm = 0; lm < 2; ++lm) {
      CHECK(anx_format_query_output(m, q, 2, r, o3, 0.5f, js, lm, 1, &out, &outlen) == ANX_OK && outlen > 20);
      anx_string_free(out);
    }
  anx_match mt[2] = {{0, 8, 1, 1, 0, 2, 0, 2}, {0, 0, 1, -1, 2, 3, 2, 2}};
  anx_match_tag mtags[2] = {{0, 0, 0}, {1, 1, 0}};
  size_t mo2[3] = {0, 1, 2};
  for (int js = 0; js < 2; ++js) {
    CHECK(anx_format_search_output(m, q, 2, mt, mo2, r, mtags, 0.0f, js, 1, 1, &out, &outlen) == ANX_OK && outlen > 20);
    anx_string_free(out);
  }
  anx_match badm[1] = {{0, 99, 1, -1, 0, 0, 0, 0}};
  size_t bo[2] = {0, 1};
  CHECK(anx_format_search_output(m, q, 1, badm, bo, r, nullptr, 0.0f, 0, 0, 1, &out, &outlen) != ANX_OK);
  anx_model_free(m3);
  anx_model_free(m2);
  anx_model_free(m);
  printf("OK %d\n", checks);
  return 0;
}
