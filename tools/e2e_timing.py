"""End-to-end timing of one 1 M-query call from host strings (encode / device run / download), optionally BASELINE
config 3 (nld.aspell, len <= 24, d = 3, ~10 confusable patterns rescored on the host).
usage: e2e_timing.py [eng|nld] [confusables]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import analiticcl_amd as A
from analiticcl_amd import synth

lexname = sys.argv[1] if len(sys.argv) > 1 else "eng"
conf = len(sys.argv) > 2 and sys.argv[2] == "confusables"
d = synth.materialize_golden("/tmp/anxdata")
m = A.VariantModel(d["alphabet"], A.Weights(), device=0)
m.read_lexicon(d[lexname])
if conf:
    for script, w in (("-[y]+[i]", 1.1), ("-[i]+[y]", 1.1), ("-[ck]+[k]", 1.05), ("-[c]+[k]", 1.05), ("-[ae]+[e]", 1.1), ("-[s]+[z]", 1.05),
                      ("-[z]+[s]", 1.05), ("=[c|k]-[y]+[i]", 1.1), ("+[e]$", 0.95), ("^-[h]", 0.9)):
        m.add_to_confusables(script, w)
m.build()
maxlen, dd = (24, 3) if lexname == "nld" else (16, 2)
qs = synth.make_queries(synth.load_lexicon_words(d[lexname]), 1000000, max_len=maxlen)
p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=dd, max_matches=10, score_threshold=0.25, cutoff_threshold=2.0)
for rep in range(3):
    t = time.time(); b = m.encode_batch(qs, p); t1 = time.time(); b.run(); t2 = time.time(); r = b.fetch_arrays(); t3 = time.time()
    print("from a list of str: encode %.3f s  run %.3f s  fetch%s %.3f s  -> %.2f M queries/s end to end" % (t1 - t, t2 - t1, " + rescoring" if conf else "", t3 - t2, 1.0 / (t3 - t)))
    b.free(); del r
packed = ("\0".join(qs) + "\0").encode("utf-8")
for rep in range(4):
    t = time.time(); b = m.encode_packed(packed, len(qs), p); t1 = time.time(); b.run(); t2 = time.time(); r = b.fetch_arrays(); t3 = time.time()
    print("from a packed buffer: encode %.4f s  run %.4f s  fetch%s %.4f s  -> %.2f M queries/s end to end" % (t1 - t, t2 - t1, " + rescoring" if conf else "", t3 - t2, 1.0 / (t3 - t)))
    b.free(); del r
