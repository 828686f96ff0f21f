"""Randomised search-mode soak on the GPU: the small world of tests/test_gpu_search.py (3000 words + phrases, optional bigram LM
and context rules), fresh random texts per round (short stretches, long stretches, varying max_seq) -- every Match field of
anx_find_all_matches_batch against the Python twin (test infrastructure).  usage: fuzz_search.py [seconds] [first seed]"""
import os, random, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import test_gpu_search as T
from analiticcl_amd import synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t0 = time.time()
rounds = 0
worlds = {}
while time.time() - t0 < budget:
    rng = random.Random(seed)
    with_lm, with_rules = rng.random() < 0.6, rng.random() < 0.4
    if (with_lm, with_rules) not in worlds:
        worlds[(with_lm, with_rules)] = T.build_world(with_lm, with_rules)
    g, tw, words, phrases = worlds[(with_lm, with_rules)]
    texts = T.random_texts(words[:400] if with_lm else words, phrases, 60, seed)
    pool = synth.make_queries(words[:400], 400, max_len=14, seed=seed + 7)
    texts += [" ".join(pool[i:i + rng.randrange(10, 80)]) for i in range(0, 300, 80)]  # stretches without a hard boundary
    max_seq = rng.choice((1, 2, 5, 20, 40, 250))
    n_multi, n_tagged = T.compare_with_twin(g, tw, texts, max_seq)
    rounds += 1
    print(f"seed {seed}: ok  lm {with_lm} rules {with_rules} max_seq {max_seq} texts {len(texts)} n-gram matches {n_multi} tagged {n_tagged}", flush=True)
    seed += 1
print(f"{rounds} rounds identical to the twin in {time.time() - t0:.0f} s")
