// Drives the host side of libanx through its C ABI under ASan / UBSan (no device): usage: host_sanitize <alphabet.tsv>
// <lexicon.tsv> <tmpdir>.  Prints "OK <checks>" and exits 0; any sanitizer report aborts with a non-zero status.
#include <cstdio>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/anx.h"

static int checks = 0;
#define CHECK(c) do { ++checks; if (!(c)) { fprintf(stderr, "CHECK failed line %d: %s (%s)\n", __LINE__, #c, anx_last_error()); return 1; } } while (0)

// Multi-replica sharding of the batch calls against the fake devices of stub_engine.cpp (ANX_STUB_FAKE=1): the results of a call
// must not depend on the number of replicas or on the form the inputs are passed in.
static int shards_mode(const std::string& alphabet, const std::string& lexicon) {
  anx_weights w; anx_default_weights(&w);
  anx_vocab_params vp; anx_default_vocab_params(&vp);
  anx_params p; anx_default_params(&p);
  std::vector<std::string> in;
  uint64_t x = 88172645463325252ull;
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  for (int i = 0; i < 5000; ++i) {
    std::string s;
    const size_t len = i % 97 == 0 ? 0 : (i > 4000 ? 1 + rnd() % 3 : rnd() % 40);  // empty strings, a short-string tail: uneven byte split
    for (size_t j = 0; j < len; ++j) s.push_back((char)('a' + rnd() % 26));
    in.push_back(s);
  }
  std::vector<const char*> ptrs;
  std::string packed;
  for (const std::string& s : in) { ptrs.push_back(s.c_str()); packed += s; packed.push_back('\0'); }
  std::vector<anx_result> ref_rows, conf_rows;
  std::vector<size_t> ref_off, conf_off;
  std::vector<uint32_t> ref_counts;
  size_t ref_pairs = 0;
  // both split policies of a multi-replica call: consecutive input ranges, and the length-partitioned split (the default: a replica
  // holds scattered inputs, the rows come back in input order all the same) -- against the same reference rows
  for (const char* policy : {"range", "length"})
  for (int nrep : {1, 2, 3, 4}) {
    const bool by_length = strcmp(policy, "length") == 0;
    // ANX_HARNESS_QUICK (the ThreadSanitizer run, several times slower): the two-replica and four-replica models under the default policy
    if (getenv("ANX_HARNESS_QUICK") && !(by_length && (nrep == 2 || nrep == 4))) continue;
    CHECK(anx_debug_set_switch("ANX_SHARD_POLICY", policy) == ANX_OK);
    anx_model* m = anx_model_new(alphabet.c_str(), &w, 0);
    CHECK(m != nullptr);
    CHECK(anx_model_read_vocabulary(m, lexicon.c_str(), &vp) == ANX_OK);
    // the host rescoring path keeps the inputs on the host (ANX_CONFUSABLES=host: the device-side weighting is the default)
    CHECK(anx_debug_set_switch("ANX_CONFUSABLES", nrep == 3 ? "host" : nullptr) == ANX_OK);
    if (nrep == 3) CHECK(anx_model_add_to_confusables(m, "-[a]+[e]", 1.05) == ANX_OK);
    CHECK(anx_model_build(m, -1) == ANX_OK);
    const int devs[4] = {0, 1, 1, 3};
    // the signature adjacency lists are built (threaded) when the model goes to its devices: the lexicon's signatures only, their
    // 1-neighbourhood too, and a budget that drops most of the lists
    CHECK(anx_debug_set_switch("ANX_ADJ_CLOSURE", nrep == 2 ? "1" : "0") == ANX_OK);
    CHECK(anx_debug_set_switch("ANX_ADJ_MB", (nrep == 3 || getenv("ANX_HARNESS_QUICK")) ? "16" : nullptr) == ANX_OK);
    CHECK(anx_model_to_devices(m, devs, nrep) == ANX_OK);
    if (nrep == 1 && !by_length) {  // the lists by themselves: every list holds its own signature's entries (section 3 = the same length)
      uint64_t sig = 0, stats[7] = {};
      CHECK(anx_debug_signature(m, "separate", &sig) == ANX_OK);
      uint32_t cum[8] = {}, *ids = nullptr, *ent = nullptr;
      size_t nent = 0;
      CHECK(anx_debug_entries(m, &ent, &nent) == ANX_OK && nent > 100000);
      CHECK(anx_debug_adjacency(m, 1, (uint64_t)1 << 40, &sig, 1, cum, &ids, stats) == ANX_OK);
      CHECK(cum[0] != 0xFFFFFFFFu && stats[2] == stats[1] && stats[1] > stats[0] && cum[7] > cum[1]);
      size_t real = 0;
      for (size_t i = (size_t)cum[0] * 64; i < (size_t)(cum[0] + cum[7]) * 64; ++i) real += ids[i] < nent;
      CHECK(real > 100 && real <= (size_t)cum[7] * 64);
      free(ids);
      free(ent);
    }
    CHECK(anx_model_num_replicas(m) == nrep && anx_model_replica_device(m, nrep - 1) == devs[nrep - 1] && anx_model_replica_device(m, nrep) == -1);
    CHECK(anx_debug_set_switch("ANX_SHARD_MIN", nrep == 4 ? "2000" : "1") == ANX_OK);  // 2000: 5000 inputs use 2 of the 4 replicas
    CHECK(anx_debug_set_switch("ANX_NO_SUCH_SWITCH", "1") == ANX_EINVAL);
    for (int form = 0; form < 3; ++form) {
      anx_result* rows = nullptr; size_t* off = nullptr;
      anx_batch* b = nullptr;
      if (form == 0) {
        CHECK(anx_find_variants_batch(m, ptrs.data(), ptrs.size(), &p, &rows, &off) == ANX_OK);
      } else {
        b = form == 1 ? anx_batch_encode(m, ptrs.data(), ptrs.size(), &p) : anx_batch_encode_packed(m, packed.data(), packed.size(), in.size(), &p);
        CHECK(b != nullptr);
        CHECK(anx_batch_num_shards(b) == (nrep == 4 ? 2 : nrep));
        int dev = -1; size_t lo = 99, cnt = 0, total = 0;
        std::vector<char> seen(in.size(), 0);
        for (int g = 0; g < anx_batch_num_shards(b); ++g) {
          CHECK(anx_batch_shard_info(b, g, &dev, &lo, &cnt) == ANX_OK && dev == devs[g]);
          const uint32_t* ix = nullptr;
          CHECK(anx_batch_shard_inputs(b, g, &ix) == ANX_OK);
          // the host-rescoring model (nrep == 3) and one-shard calls keep consecutive ranges under either policy
          if (by_length && nrep != 3 && anx_batch_num_shards(b) > 1) {
            CHECK(ix != nullptr && cnt > 0 && ix[0] == lo);
            for (size_t i = 0; i < cnt; ++i) { CHECK(ix[i] < in.size() && !seen[ix[i]] && (i == 0 || ix[i] > ix[i - 1])); seen[ix[i]] = 1; }
            // whole lengths stay together: apart from the lengths a cut runs through, a shard's lengths are a range of their own
          } else {
            CHECK(ix == nullptr && lo == total);
          }
          total += cnt;
        }
        CHECK(total == in.size());
        CHECK(anx_batch_run_async(m, b, nullptr) == ANX_OK && anx_batch_wait(m, b) == ANX_OK);
        if (nrep > 1) CHECK(anx_batch_run(m, b, (void*)0x10) == ANX_EINVAL);  // a caller stream with several replicas
        CHECK(anx_batch_fetch(b, &rows, &off) == ANX_OK);
        if (nrep != 3) {  // the top-k gather behind the C ABI: every shard's compact section in one buffer, rows equal the fetched ones
          const int S = anx_batch_num_shards(b);
          std::vector<size_t> so((size_t)S + 1, 0);
          size_t used = 0;
          CHECK(anx_batch_gather_compact(b, 0, (void*)&used, 0, so.data(), &used) == ANX_ELIMIT && used > 0);   // too small: says what it needs
          std::vector<char> buf(used);
          size_t used2 = 0;
          CHECK(anx_batch_gather_compact(b, 0, buf.data(), buf.size(), so.data(), &used2) == ANX_OK && used2 == used && so[(size_t)S] == used);
          for (int g = 0; g < S; ++g) {
            size_t lo2 = 0, cnt2 = 0;
            const uint32_t* ix = nullptr;
            CHECK(anx_batch_shard_info(b, g, nullptr, &lo2, &cnt2) == ANX_OK && anx_batch_shard_inputs(b, g, &ix) == ANX_OK);
            const uint32_t* go = reinterpret_cast<const uint32_t*>(buf.data() + so[(size_t)g]);
            const anx_topk_record* gr = reinterpret_cast<const anx_topk_record*>(buf.data() + so[(size_t)g] + (((cnt2 + 1) * 4 + 15) & ~(size_t)15));
            for (size_t i = 0; i < cnt2; ++i) {
              const size_t inp = ix ? ix[i] : lo2 + i;
              CHECK(go[i + 1] - go[i] == off[inp + 1] - off[inp]);
              for (uint32_t k = go[i]; k < go[i + 1]; ++k)
                CHECK(gr[k].vocab_id == (uint32_t)rows[off[inp] + (k - go[i])].vocab_id && gr[k].dist_score == rows[off[inp] + (k - go[i])].dist_score);
            }
          }
        } else {
          size_t used = 0;
          char dummy[16];
          CHECK(anx_batch_gather_compact(b, 0, dummy, sizeof dummy, nullptr, &used) == ANX_EINVAL);  // host-rescored rows have no device export
        }
        anx_batch_stats st;
        CHECK(anx_batch_get_stats(b, &st, sizeof st) == ANX_OK && st.n_queries == in.size() && (nrep == 3 || st.n_results == off[in.size()]));
        uint32_t* counts = nullptr;
        CHECK(anx_batch_pair_counts(b, &counts) == ANX_OK);
        anx_pair* pairs = nullptr; size_t npairs = 0;
        CHECK(anx_batch_fetch_pairs(b, &pairs, &npairs) == ANX_OK);
        if (ref_counts.empty()) { ref_counts.assign(counts, counts + in.size()); ref_pairs = npairs; }
        CHECK(memcmp(ref_counts.data(), counts, in.size() * sizeof(uint32_t)) == 0 && npairs == ref_pairs);
        // pairs come shard by shard, each shard's in its own input order, with call-wide query indices
        if (!by_length || nrep == 3) {
          size_t w0 = 0;
          for (size_t i = 0; i < in.size(); ++i)
            for (size_t j = 0; j < in[i].size(); ++j, ++w0) CHECK(pairs[w0].query == i && pairs[w0].vocab_id == (uint32_t)(unsigned char)in[i][j]);
        } else {
          std::vector<uint32_t> at(in.size(), 0);
          bool okp = true;
          for (size_t k = 0; k < npairs && okp; ++k) {
            const uint32_t q = pairs[k].query;
            okp = q < in.size() && at[q] < in[q].size() && pairs[k].vocab_id == (uint32_t)(unsigned char)in[q][at[q]];
            if (okp) ++at[q];
          }
          CHECK(okp);
          for (size_t i = 0; i < in.size() && okp; ++i) okp = at[i] == in[i].size();
          CHECK(okp);
        }
        anx_counts_free(counts);
        anx_pairs_free(pairs);
        if (nrep != 3) {  // compact records over the shards == the anx_result rows
          anx_topk_record* cr = nullptr; uint32_t* co = nullptr;
          CHECK(anx_batch_fetch_compact(b, &cr, &co) == ANX_OK);
          std::vector<anx_result> view(off[in.size()] + 1);
          anx_compact_to_results(cr, off[in.size()], view.data());
          for (size_t i = 0; i <= in.size(); ++i) CHECK(co[i] == off[i]);
          CHECK(memcmp(view.data(), rows, off[in.size()] * sizeof(anx_result)) == 0);
          anx_compact_free(cr, co);
        } else {
          anx_topk_record* cr = nullptr; uint32_t* co = nullptr;
          CHECK(anx_batch_fetch_compact(b, &cr, &co) == ANX_EINVAL);
        }
        size_t used = 0;
        if (anx_batch_num_shards(b) > 1) CHECK(anx_batch_export_compact(b, &used, 8, nullptr, &used) == ANX_EINVAL);
        if (form == 2 && nrep != 3) {  // the asynchronous pipeline over the same packed buffer: three jobs in flight, rows as the staged calls'
          anx_pipeline* pl = anx_pipeline_new(m, 2);
          CHECK(pl != nullptr);
          for (int j = 0; j < 3; ++j) {
            if (j == 2) { anx_topk_record* cr = nullptr; uint32_t* co = nullptr; size_t cn = 0; CHECK(anx_pipeline_next(pl, &cr, &co, &cn) == ANX_OK && cn == in.size()); anx_compact_free(cr, co); }
            CHECK(anx_pipeline_submit_packed(pl, packed.data(), packed.size(), in.size(), &p) == ANX_OK);
          }
          CHECK(anx_pipeline_pending(pl) == 2);
          anx_topk_record* cr = nullptr; uint32_t* co = nullptr; size_t cn = 0;
          CHECK(anx_pipeline_next(pl, &cr, &co, &cn) == ANX_OK && cn == in.size());
          bool samep = true;
          for (size_t i = 0; i <= in.size() && samep; ++i) samep = co[i] == off[i];
          for (size_t i = 0; i < off[in.size()] && samep; ++i) samep = cr[i].vocab_id == rows[i].vocab_id && cr[i].dist_score == rows[i].dist_score;
          CHECK(samep);
          anx_compact_free(cr, co);
          CHECK(anx_pipeline_submit_packed(pl, packed.data(), packed.size(), in.size() + 9, &p) == ANX_OK);  // fails in its encode stage
          CHECK(anx_pipeline_next(pl, &cr, &co, &cn) == ANX_OK);
          anx_compact_free(cr, co);
          CHECK(anx_pipeline_next(pl, &cr, &co, &cn) == ANX_EINVAL);
          CHECK(anx_pipeline_next(pl, &cr, &co, &cn) == ANX_EINVAL && anx_pipeline_pending(pl) == 0);  // nothing in flight
          CHECK(anx_pipeline_submit_packed(pl, packed.data(), packed.size(), in.size(), &p) == ANX_OK);  // left in flight: freed with the pipeline
          anx_pipeline_free(pl);
        }
      }
      // confusables loaded (nrep == 3): the host rescoring applies the cutoff -- compared across the three input forms only
      std::vector<size_t>& roff = nrep == 3 ? conf_off : ref_off;
      std::vector<anx_result>& rrows = nrep == 3 ? conf_rows : ref_rows;
      if (roff.empty()) { roff.assign(off, off + in.size() + 1); rrows.assign(rows, rows + off[in.size()]); }
      CHECK(memcmp(roff.data(), off, (in.size() + 1) * sizeof(size_t)) == 0);
      CHECK(off[in.size()] == rrows.size() && memcmp(rrows.data(), rows, rrows.size() * sizeof(anx_result)) == 0);
      anx_results_free(rows, off);
      anx_batch_free(b);
    }
    // packed form: more strings announced than present / trailing strings beyond n are ignored
    CHECK(anx_batch_encode_packed(m, packed.data(), packed.size(), in.size() + 1, &p) == nullptr && strstr(anx_last_error(), "fewer strings") != nullptr);
    {
      anx_batch* b = anx_batch_encode_packed(m, packed.data(), packed.size(), in.size() - 1234, &p);
      CHECK(b != nullptr);
      anx_result* rows = nullptr; size_t* off = nullptr;
      CHECK(anx_batch_run(m, b, nullptr) == ANX_OK && anx_batch_fetch(b, &rows, &off) == ANX_OK);
      if (nrep != 3) CHECK(memcmp(ref_off.data(), off, (in.size() - 1234 + 1) * sizeof(size_t)) == 0);
      anx_results_free(rows, off);
      anx_batch_free(b);
    }
    {  // a call below the shard minimum uses one replica; an empty call works
      CHECK(anx_debug_set_switch("ANX_SHARD_MIN", nullptr) == ANX_OK);
      anx_batch* b = anx_batch_encode(m, ptrs.data(), 100, &p);
      CHECK(b != nullptr && anx_batch_num_shards(b) == 1);
      anx_batch_free(b);
      anx_result* rows = nullptr; size_t* off = nullptr;
      CHECK(anx_find_variants_batch(m, ptrs.data(), 0, &p, &rows, &off) == ANX_OK && off[0] == 0);
      anx_results_free(rows, off);
    }
    // search mode drives the same staged calls: texts over several replicas
    if (nrep == 2) {
      CHECK(anx_debug_set_switch("ANX_SHARD_MIN", "1") == ANX_OK);
      anx_search_params sp; anx_default_search_params(&sp);
      const char* texts[2] = {"I tink you are rihgt", "so it is"};
      anx_match* ms = nullptr; size_t* mo = nullptr; anx_result* rr = nullptr; size_t nr = 0; anx_match_tag* tg = nullptr;
      CHECK(anx_find_all_matches_batch(m, texts, 2, &sp, &ms, &mo, &rr, &nr, &tg) == ANX_OK);
      anx_matches_free(ms, mo, rr, tg);
      // a large call runs as concurrent parts whose arrays are merged: the same matches, offsets, rows and tags as one pass
      std::vector<std::string> many;
      for (int i = 0; i < 41; ++i) many.push_back(i % 7 == 3 ? std::string() : std::string(texts[i % 2]) + (i % 3 ? " and teh " : "\n") + in[(size_t)i * 37 % in.size()]);
      std::vector<const char*> mp;
      for (auto& x : many) mp.push_back(x.c_str());
      CHECK(anx_debug_set_switch("ANX_SEARCH_PARTS_MIN", "1") == ANX_OK);
      anx_match* ref_m = nullptr; size_t* ref_o = nullptr; anx_result* ref_r = nullptr; size_t ref_nr = 0; anx_match_tag* ref_t = nullptr;
      CHECK(anx_debug_set_switch("ANX_SEARCH_PARTS", "1") == ANX_OK);
      CHECK(anx_find_all_matches_batch(m, mp.data(), mp.size(), &sp, &ref_m, &ref_o, &ref_r, &ref_nr, &ref_t) == ANX_OK);
      for (const char* np : {"2", "3", "8"}) {
        CHECK(anx_debug_set_switch("ANX_SEARCH_PARTS", np) == ANX_OK);
        CHECK(anx_find_all_matches_batch(m, mp.data(), mp.size(), &sp, &ms, &mo, &rr, &nr, &tg) == ANX_OK);
        CHECK(nr == ref_nr && memcmp(mo, ref_o, (mp.size() + 1) * sizeof(size_t)) == 0);
        const size_t nm = ref_o[mp.size()];
        bool same = true;
        for (size_t k = 0; k < nm && same; ++k)
          same = ms[k].begin == ref_m[k].begin && ms[k].end == ref_m[k].end && ms[k].n == ref_m[k].n && ms[k].selected == ref_m[k].selected &&
                 ms[k].var_begin == ref_m[k].var_begin && ms[k].var_end == ref_m[k].var_end && ms[k].tag_begin == ref_m[k].tag_begin && ms[k].tag_end == ref_m[k].tag_end;
        CHECK(same);
        for (size_t k = 0; k < nr && same; ++k)
          same = rr[k].vocab_id == ref_r[k].vocab_id && rr[k].dist_score == ref_r[k].dist_score && rr[k].freq_score == ref_r[k].freq_score && rr[k].via == ref_r[k].via;
        CHECK(same);
        anx_matches_free(ms, mo, rr, tg);
        // without a tag array the output is written while later parts are still at work (arrays sized by upper bounds, cut back at the
        // end), with ANX_SEARCH_EARLY_OUTPUT=0 when the last part is done: the same arrays either way
        for (const char* early : {"1", "0"}) {
          CHECK(anx_debug_set_switch("ANX_SEARCH_EARLY_OUTPUT", early) == ANX_OK);
          CHECK(anx_find_all_matches_batch(m, mp.data(), mp.size(), &sp, &ms, &mo, &rr, &nr, nullptr) == ANX_OK);
          CHECK(nr == ref_nr && memcmp(mo, ref_o, (mp.size() + 1) * sizeof(size_t)) == 0);
          same = true;
          for (size_t k = 0; k < nm && same; ++k)
            same = ms[k].begin == ref_m[k].begin && ms[k].end == ref_m[k].end && ms[k].n == ref_m[k].n && ms[k].selected == ref_m[k].selected &&
                   ms[k].var_begin == ref_m[k].var_begin && ms[k].var_end == ref_m[k].var_end && ms[k].tag_begin == 0 && ms[k].tag_end == 0;
          for (size_t k = 0; k < nr && same; ++k)
            same = rr[k].vocab_id == ref_r[k].vocab_id && rr[k].dist_score == ref_r[k].dist_score && rr[k].freq_score == ref_r[k].freq_score && rr[k].via == ref_r[k].via;
          CHECK(same);
          anx_matches_free(ms, mo, rr, nullptr);
        }
        CHECK(anx_debug_set_switch("ANX_SEARCH_EARLY_OUTPUT", nullptr) == ANX_OK);
      }
      // anx_shutdown joins the pool's threads; the next call starts a fresh pool and gives the same answer
      anx_shutdown();
      anx_shutdown();  // idempotent
      CHECK(anx_find_all_matches_batch(m, mp.data(), mp.size(), &sp, &ms, &mo, &rr, &nr, &tg) == ANX_OK);
      CHECK(nr == ref_nr && memcmp(mo, ref_o, (mp.size() + 1) * sizeof(size_t)) == 0);
      anx_matches_free(ms, mo, rr, tg);
      // a fork()ed child has none of the parent's threads: its atfork handler forgets the parent's pool, so host-only calls work
      // there (models are not usable across fork: the child only touches the pool through a model-free entry point)
      {
        fflush(stdout);
        const pid_t pid = fork();
        CHECK(pid >= 0);
        if (pid == 0) {
          anx_shutdown();           // nothing to join in the child
          char buf[64];
          _exit(anx_edit_script("huys", "huis", buf, sizeof buf) > 0 ? 0 : 3);
        }
        int status = 0;
        CHECK(waitpid(pid, &status, 0) == pid && WIFEXITED(status) && WEXITSTATUS(status) == 0);
      }
      anx_matches_free(ref_m, ref_o, ref_r, ref_t);
      CHECK(anx_debug_set_switch("ANX_SEARCH_PARTS", nullptr) == ANX_OK && anx_debug_set_switch("ANX_SEARCH_PARTS_MIN", nullptr) == ANX_OK);
    }
    anx_model_free(m);
  }
  CHECK(anx_debug_set_switch("ANX_SHARD_POLICY", nullptr) == ANX_OK);
  printf("OK %d\n", checks);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const std::string alphabet = argv[1], lexicon = argv[2], tmp = argv[3];
  if (argc > 4 && strcmp(argv[4], "shards") == 0) return shards_mode(alphabet, lexicon);
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
    CHECK(anx_debug_set_switch("ANX_CONFUSABLES", "host") == ANX_OK);  // (the default weights on the device and needs no host-side scan)
    const char packed[] = "seperate\0\0\x01x\0acommodate\0longer than eight bytes\0";  // sizeof counts the terminator too
    CHECK(anx_batch_encode_packed(m, packed, sizeof packed - 1, 5, &p) == nullptr && strstr(anx_last_error(), "not resident") != nullptr);
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
