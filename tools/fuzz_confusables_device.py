"""Device-vs-host soak of the confusable weighting (GPU box): random pattern sets over the golden lexicons, random parameters,
late and early mode; every ranked row (ids, order, f64 scores) of the device path (conf.hip) must equal the host path
(ANX_CONFUSABLES=host) -- both compile confusables_core.hpp.  Prints the number of rows and of edit scripts compared.
usage: fuzz_confusables_device.py [seconds] [seed]"""
import os, random, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import analiticcl_amd as A
from analiticcl_amd import synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
d = synth.materialize_golden("/tmp/anxdata")
letters = "abcdefghijklmnopqrstuvwxyz"
t0 = time.time()
rows = rounds = 0
while time.time() - t0 < budget:
    lex = rng.choice(["eng", "nld"])
    words = synth.load_lexicon_words(d[lex])
    g = A.VariantModel(d["alphabet"], A.Weights(), device=0)
    g.read_lexicon(d[lex])
    npat = rng.randrange(1, 14)
    pats = []
    for _ in range(npat):
        ops = []
        for k in range(rng.randrange(1, 4)):
            op = rng.choice("-+=")
            opts = "|".join("".join(rng.choice(letters + "ëéï") for _ in range(rng.choice([1, 1, 1, 2]))) for _ in range(rng.choice([1, 1, 2, 3])))
            ops.append(f"{op}[{opts}]")
        script = ("^" if rng.random() < 0.15 else "") + "".join(ops) + ("$" if rng.random() < 0.15 else "")
        pats.append((script, rng.choice([0.8, 0.9, 0.95, 1.05, 1.1, 1.2])))
        g.add_to_confusables(*pats[-1])
    early = rng.random() < 0.4
    if early:
        g.set_confusables_before_pruning()
    g.build()
    qs = synth.make_queries(words, rng.choice([20_000, 100_000, 300_000]), max_len=rng.choice([12, 16, 24, 30]), seed=rng.randrange(1 << 30))
    p = A.SearchParameters(max_anagram_distance=rng.choice([2, 3]), max_edit_distance=rng.choice([1, 2, 3]), max_matches=rng.choice([0, 1, 3, 10, 20]),
                           score_threshold=rng.choice([0.0, 0.25, 0.5]), cutoff_threshold=rng.choice([0.0, 1.5, 2.0]), freq_weight=rng.choice([0.0, 0.0, 0.5]))
    out = {}
    for mode in ("device", "host"):
        A.set_switch("ANX_CONFUSABLES", "host" if mode == "host" else None)
        b = g.encode_batch(qs, p)
        b.run()
        out[mode] = b.fetch_arrays()
        b.free()
    A.set_switch("ANX_CONFUSABLES", None)
    if any(not np.array_equal(x, y) for x, y in zip(out["device"], out["host"])):
        print("DIFFERENT", lex, "early" if early else "late", pats, {k: v for k, v in p.__dict__.items() if k in ("max_anagram_distance", "max_edit_distance", "max_matches", "score_threshold", "cutoff_threshold", "freq_weight")})
        (doff, dvid, ddist, dfreq), (hoff, hvid, hdist, hfreq) = out["device"], out["host"]
        shown = 0
        for i in range(len(qs)):
            a = list(zip(dvid[doff[i]:doff[i + 1]].tolist(), ddist[doff[i]:doff[i + 1]].tolist(), dfreq[doff[i]:doff[i + 1]].tolist()))
            h = list(zip(hvid[hoff[i]:hoff[i + 1]].tolist(), hdist[hoff[i]:hoff[i + 1]].tolist(), hfreq[hoff[i]:hoff[i + 1]].tolist()))
            if a != h:
                print(repr(qs[i]), "\n dev ", [(g.vocab_text(v), d_, f_) for v, d_, f_ in a], "\n host", [(g.vocab_text(v), d_, f_) for v, d_, f_ in h])
                shown += 1
                if shown == 3:
                    break
        sys.exit(1)
    rows += int(out["device"][0][-1])
    rounds += 1
    del g
print(f"{rounds} rounds, {rows} ranked rows: device == host in {time.time() - t0:.0f} s")
