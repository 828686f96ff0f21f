// Host-side confusable rescoring in isolation (no device): HostModel::confusable_weights over (query, ranked vocab ids) lines
// written by dump_pairs.py.  usage: conf_bench <alphabet.tsv> <lexicon.tsv> <confusables.tsv> <pairs.tsv> [repeats]
// build: tools/conf_bench/build.sh
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/anx.h"
#include "../../analiticcl_amd/csrc/host_model.h"

const anx::HostModel& anx_host_of(const anx_model* m);

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const int reps = argc > 5 ? atoi(argv[5]) : 20;
  anx_weights w; anx_default_weights(&w);
  anx_vocab_params vp; anx_default_vocab_params(&vp);
  anx_model* m = anx_model_new(argv[1], &w, 0);
  if (!m || anx_model_read_vocabulary(m, argv[2], &vp) != ANX_OK || anx_model_read_confusablelist(m, argv[3]) != ANX_OK ||
      anx_model_build(m, -1) != ANX_OK) { fprintf(stderr, "setup failed: %s\n", anx_last_error()); return 1; }
  const anx::HostModel& hm = anx_host_of(m);
  std::vector<std::string> qs;
  std::vector<std::vector<uint64_t>> ids;
  std::ifstream in(argv[4]);
  std::string line;
  size_t rows = 0;
  while (std::getline(in, line)) {
    const size_t tab = line.find('\t');
    if (tab == std::string::npos) continue;
    qs.push_back(line.substr(0, tab));
    ids.emplace_back();
    std::istringstream ss(line.substr(tab + 1));
    uint64_t v;
    while (ss >> v) ids.back().push_back(v);
    rows += ids.back().size();
  }
  std::vector<double> wts(64);
  double acc = 0.0;
  size_t changed = 0;
  hm.confusable_weights(qs[0], ids[0].data(), ids[0].size(), wts.data());  // builds the decoded-vocabulary cache
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r)
    for (size_t i = 0; i < qs.size(); ++i) {
      wts.resize(std::max<size_t>(wts.size(), ids[i].size()));
      hm.confusable_weights(qs[i], ids[i].data(), ids[i].size(), wts.data());
      for (size_t k = 0; k < ids[i].size(); ++k) { acc += wts[k]; changed += wts[k] != 1.0; }
    }
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("%zu queries, %zu rows, %d repeats: %.1f ns per row, %.2f M rows/s on one thread; %.1f %% of the rows reweighted (checksum %.6f)\n",
         qs.size(), rows, reps, dt / (double)(rows * (size_t)reps) * 1e9, (double)(rows * (size_t)reps) / dt / 1e6,
         100.0 * (double)changed / (double)(rows * (size_t)reps), acc);
  anx_model_free(m);
  return 0;
}
