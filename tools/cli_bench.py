"""End-to-end `query` command line: N synthetic queries through `python -m analiticcl_amd query` (TSV to /dev/null)."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from analiticcl_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
d = synth.materialize_golden("/tmp/anxdata")
qs = synth.make_queries(synth.load_lexicon_words(d["eng"]), N, max_len=16, seed=7)
open("/tmp/anx_cli_in.txt", "w", encoding="utf-8").write("\n".join(qs) + "\n")
for extra in ([], ["--json"]):
    t = time.time()
    with open("/tmp/anx_cli_out.txt", "w") as out:
        rc = subprocess.call([sys.executable, "-m", "analiticcl_amd", "query", "--lexicon", d["eng"], "--alphabet", d["alphabet"],
                              "/tmp/anx_cli_in.txt"] + extra, stdout=out, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    dt = time.time() - t
    print("query %s: rc %d, %d lines in %.1f s (incl. start-up and model build) = %.0f queries/s, %.1f MB out"
          % (" ".join(extra) or "tsv", rc, N, dt, N / dt, os.path.getsize("/tmp/anx_cli_out.txt") / 1e6))
