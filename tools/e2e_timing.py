import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import analiticcl_amd as A
from analiticcl_amd import synth
d = synth.materialize_golden("/tmp/anxdata")
m = A.VariantModel(d["alphabet"], A.Weights(), device=0); m.read_lexicon(d["eng"]); m.build()
qs = synth.make_queries(synth.load_lexicon_words(d["eng"]), 1000000, max_len=16)
p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10, score_threshold=0.25, cutoff_threshold=2.0)
for rep in range(2):
    t = time.time(); b = m.encode_batch(qs, p); t1 = time.time(); b.run(); t2 = time.time(); r = b.fetch_arrays() if hasattr(b, "fetch_arrays") else b.fetch(); t3 = time.time()
    print("encode %.3f run %.3f fetch %.3f" % (t1 - t, t2 - t1, t3 - t2))
