"""CPU soak of the host-side diff (csrc/confusables.cpp through anx_edit_script, no GPU) against the twin's
(oracle/sesdiff_twin.py, test infrastructure): word pairs of the golden lexicons at small edit distances, unrelated words,
strings over a tiny alphabet (many repeats: the clean-up passes' shifts and overlaps), multi-word strings, non-ASCII.
usage: fuzz_editscript.py [seconds] [seed]"""
import os, random, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import analiticcl_amd as A
from analiticcl_amd import synth
from oracle.sesdiff_twin import shortest_edit_script, script_to_str

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
d = synth.materialize_golden("/tmp/anxdata")
words = synth.load_lexicon_words(d["eng"]) + synth.load_lexicon_words(d["nld"])[::2]
t0 = time.time()
n = 0
while time.time() - t0 < budget:
    kind = rng.randrange(6)
    if kind == 0:
        w = rng.choice(words)
        a, b = synth.make_queries([w], 1, max_len=40, seed=rng.randrange(1 << 30))[0], w
    elif kind == 1:
        a, b = rng.choice(words), rng.choice(words)
    elif kind == 2:
        a = "".join(rng.choice("ab") for _ in range(rng.randrange(0, 14)))
        b = "".join(rng.choice("ab") for _ in range(rng.randrange(0, 14)))
    elif kind == 3:
        a = "".join(rng.choice("abc \n") for _ in range(rng.randrange(0, 20)))
        b = "".join(rng.choice("abc \n") for _ in range(rng.randrange(0, 20)))
    elif kind == 4:
        ws = [rng.choice(words) for _ in range(rng.randrange(1, 5))]
        a = " ".join(ws)
        ws2 = [synth.make_queries([x], 1, max_len=40, seed=rng.randrange(1 << 30))[0] if rng.random() < 0.5 else x for x in ws]
        if rng.random() < 0.3 and len(ws2) > 1:
            ws2.pop(rng.randrange(len(ws2)))
        b = " ".join(ws2)
    else:
        a = "".join(rng.choice("éèaeßss日本e") for _ in range(rng.randrange(0, 12)))
        b = "".join(rng.choice("éèaeßss日本e") for _ in range(rng.randrange(0, 12)))
    if "\0" in a or "\0" in b:
        continue
    got, want = A.edit_script(a, b), script_to_str(shortest_edit_script(a, b))
    if got != want:
        print("DIFFERENT", repr(a), repr(b), got, want)
        sys.exit(1)
    n += 1
print(f"{n} pairs: identical edit scripts in {time.time() - t0:.0f} s")
