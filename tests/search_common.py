"""Search-mode checker shared by the GPU tests: the oracle twin's segmentation / lattice / LM code with its per-segment
find_variants answered by the C oracle (same results as the twin's own, tests/test_oracle_c.py), so that thousands of
segments finish in seconds.  TEST INFRASTRUCTURE."""
from oracle import cwrap as O
from oracle import twin as T


def twin_matches_parallel(alphabet, lexicon, lm, texts, workers=16, max_ngram=3, timeout=900):
    """find_all_matches of the oracle twin over `texts`, spread over `workers` child processes (tests/search_twin_worker.py; a text
    of 1 KB takes the twin ~0.5 s, nearly all of it the C oracle's per-segment find_variants) -> (per text: [[text, begin, end, n,
    selected, [[vocab_id, dist, freq]]]], the slowest worker's seconds over its texts)."""
    import json
    import os
    import subprocess
    import sys
    import tempfile
    workers = max(1, min(workers, len(texts)))
    tmp = tempfile.mkdtemp(prefix="anx_twin_")
    procs = []
    for w in range(workers):
        job = os.path.join(tmp, f"job{w}.json")
        with open(job, "w") as f:
            json.dump({"alphabet": alphabet, "lexicon": lexicon, "lm": [[t, fr] for t, fr in lm], "texts": texts[w::workers], "max_ngram": max_ngram}, f)
        out = os.path.join(tmp, f"out{w}.json")
        env = dict(os.environ, OMP_NUM_THREADS="1")
        procs.append((subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "search_twin_worker.py"), job, out],
                                       env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE), out))
    res = [None] * len(texts)
    slowest = 0.0
    for w, (p, out) in enumerate(procs):
        _o, err = p.communicate(timeout=timeout)
        if p.returncode != 0:
            raise RuntimeError(f"twin worker {w} failed: {err.decode(errors='replace')[-2000:]}")
        with open(out) as f:
            r = json.load(f)
        slowest = max(slowest, r["seconds"])
        for k, m in enumerate(r["matches"]):
            res[w + k * workers] = m
    return res, slowest


class TwinOverOracle(T.SearchModel):
    """ids are aligned: twin and C oracle number the vocabulary in insertion order after BOS/EOS/UNK."""

    def attach(self, orc):
        self.orc = orc

    def find_variants(self, text, params, trace=None):
        cp = O.make_params(params.max_anagram_distance, params.max_edit_distance, params.max_matches,
                           params.score_threshold, params.cutoff_threshold, params.stop_at_exact_match,
                           params.freq_weight)
        return [T.VariantResult(v, d, f, via) for v, d, f, via in self.orc.find_variants_via(text, cp)]
