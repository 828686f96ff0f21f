"""Worker process of the parallel twin check (tests/test_gpu_config4.py, bench.py's search baseline): reads a JSON job
{alphabet, lexicon, lm: [[text, freq]], texts: [...], max_ngram}, runs the oracle twin's find_all_matches (per-segment find_variants
answered by the C oracle) over the texts and writes, per text, [[matched text, begin, end, n, selected, [[vocab_id, dist, freq]]]] +
the seconds the texts took (without the model build).  TEST INFRASTRUCTURE: started as a child process, never imported by the product."""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from oracle import cwrap as O  # noqa: E402
from oracle import twin as T  # noqa: E402
from search_common import TwinOverOracle  # noqa: E402


def main(job_path, out_path):
    with open(job_path) as f:
        job = json.load(f)
    tw = TwinOverOracle(T.read_alphabet(job["alphabet"]))
    tw.read_vocabulary(job["lexicon"])
    for t, fr in job["lm"]:
        tw.add_lm(t, fr)
    tw.build()
    orc = O.OracleModel(alphabet_path=job["alphabet"])
    orc.read_lexicon(job["lexicon"])
    orc.build()
    tw.attach(orc)
    tp = T.SearchParams(("abs", 3), ("abs", 2), 10, 0.25, 2.0, False, 0.0, max_ngram=job.get("max_ngram", 3))
    out = []
    t0 = time.perf_counter()
    for text in job["texts"]:
        exp = tw.find_all_matches(text, tp)
        out.append([[e.text, e.begin, e.end, e.n, e.selected if e.variants else -1,
                     [[v.vocab_id, v.dist_score, v.freq_score] for v in (e.variants or [])]] for e in exp])
    dt = time.perf_counter() - t0
    with open(out_path, "w") as f:
        json.dump({"matches": out, "seconds": dt, "bytes": sum(len(t.encode("utf-8")) for t in job["texts"])}, f)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
