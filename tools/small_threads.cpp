// small_threads.cpp -- T native host threads, each issuing `calls` anx_find_variants_batch calls of n inputs on ONE model (the reference's
// fan-out: independent find_variants calls on a shared, read-only VariantModel, src/bin/analiticcl.rs:445-448).  bench.py's by_batch_size
// measures the same with Python threads, whose per-call interpreter work is serialised by the GIL; this is the library's own figure.
//   build: g++ -O2 -std=c++17 -pthread -I include tools/small_threads.cpp -o build/small_threads -L analiticcl_amd -lanx -Wl,-rpath,$PWD/analiticcl_amd
//   run:   build/small_threads <alphabet.tsv> <lexicon> <queries.txt> [threads=8] [n=1000] [calls=200]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "anx.h"

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: small_threads alphabet lexicon queries [threads] [n] [calls]\n"); return 2; }
  const int T = argc > 4 ? atoi(argv[4]) : 8, n = argc > 5 ? atoi(argv[5]) : 1000, calls = argc > 6 ? atoi(argv[6]) : 200;
  anx_weights w;
  anx_default_weights(&w);
  anx_model* m = anx_model_new(argv[1], &w, 0);
  if (!m) { fprintf(stderr, "model: %s\n", anx_last_error()); return 1; }
  anx_vocab_params vp;
  anx_default_vocab_params(&vp);
  if (anx_model_read_vocabulary(m, argv[2], &vp) || anx_model_build(m, 0)) { fprintf(stderr, "build: %s\n", anx_last_error()); return 1; }
  std::vector<std::string> qs;
  { std::ifstream f(argv[3]); std::string line; while (std::getline(f, line)) qs.push_back(line); }
  if ((int)qs.size() < T * n) { fprintf(stderr, "need %d queries, have %zu\n", T * n, qs.size()); return 1; }
  anx_params p;
  anx_default_params(&p);
  p.max_anagram_distance.value = 3; p.max_edit_distance.value = 2; p.max_matches = 10;
  std::atomic<int> failed{0};
  auto work = [&](int t, int reps) {
    std::vector<const char*> ptr(n);
    for (int i = 0; i < n; ++i) ptr[i] = qs[(size_t)t * n + i].c_str();
    for (int r = 0; r < reps; ++r) {
      anx_result* rows = nullptr;
      size_t* offs = nullptr;
      if (anx_find_variants_batch(m, ptr.data(), (size_t)n, &p, &rows, &offs) != 0) { failed++; return; }
      anx_results_free(rows, offs);
    }
  };
  double best = 0.0;
  for (int pass = 0; pass < 4; ++pass) {   // the first pass warms the contexts
    std::vector<std::thread> th;
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < T; ++t) th.emplace_back(work, t, pass ? calls : 20);
    for (auto& x : th) x.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (pass) best = std::max(best, (double)T * calls * n / dt);
  }
  uint64_t st[2] = {0, 0};
  anx_debug_small_stats(st);
  printf("{\"threads\": %d, \"n\": %d, \"calls_per_thread\": %d, \"queries_per_s\": %.0f, \"failed\": %d, \"small_path_calls\": %llu, \"handed_to_batch_path\": %llu}\n", T, n, calls, best,
         failed.load(), (unsigned long long)st[0], (unsigned long long)st[1]);
  anx_model_free(m);
  anx_shutdown();
  return failed.load() ? 1 : 0;
}
