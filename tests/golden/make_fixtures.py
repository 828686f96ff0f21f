#!/usr/bin/env python3
"""Generate the committed golden fixtures from the reference tree (run in the build container only).

Inputs (read-only, /root/reference):
  * tutorial.ipynb        -- RECORDED OUTPUTS of the reference (analiticcl 0.4.4 wheel) for
                             build(), find_variants("separate"/"seperate") and
                             find_all_matches("We would like seperate beds"); we extract the
                             printed values only (data, not code).
  * examples/simple.alphabet.tsv, examples/eng.aspell.lexicon, examples/nld.aspell.lexicon
                          -- the data files BASELINE.json's configs name; stored gzip'ed under
                             tests/golden/data/ (GPL-3.0 data, see tests/golden/data/README.md).
Outputs:
  tests/golden/tutorial_outputs.json
  tests/golden/data/{simple_alphabet.tsv, eng_aspell.lexicon.gz, nld_aspell.lexicon.gz}
  tests/golden/twin_eng_queries.json  -- extra vectors produced by oracle/twin.py (our own restatement,
                             validated against the two sources above by tests/test_twin_golden.py);
                             they widen coverage for the C oracle and the GPU path, they do not pin the twin.

/root/reference does not exist on the GPU box; nothing at test/bench time reads it.
"""
import ast
import gzip
import json
import os
import re
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)


def extract_tutorial():
    nb = json.load(open(os.path.join(REF, "tutorial.ipynb"), encoding="utf-8"))
    out = {"source": "tutorial.ipynb recorded outputs (analiticcl 0.4.4)", "find_variants": [],
           "find_all_matches": []}
    for cell in nb["cells"]:
        if cell["cell_type"] != "code":
            continue
        src = "".join(cell["source"])
        text = ""
        for o in cell.get("outputs", []):
            t = o.get("text") or o.get("data", {}).get("text/plain")
            if t:
                text += "".join(t)
        if src.strip() == "model.build()":
            hist = {}
            for m in re.finditer(r"Found (\d+) anagrams of length (\d+)", text):
                hist[int(m.group(2))] = int(m.group(1))
            out["build"] = {
                "instances": int(re.search(r"Found (\d+) instances", text).group(1)),
                "anagrams": int(re.search(r"Found (\d+) anagrams\n", text).group(1)),
                "histogram": hist,
            }
        m = re.match(r'variants\s*=\s*model\.find_variants\("(\w+)", SearchParameters\(\)\)', src)
        if m and text:
            rows = [ast.literal_eval(line) for line in text.strip().split("\n")]
            out["find_variants"].append({
                "input": m.group(1), "params": "python-default",
                "results": [[r["text"], r["score"], r["dist_score"], r["freq_score"]] for r in rows]})
        m = re.match(r'matches = model\.find_all_matches\("([^"]+)", SearchParameters\(unicodeoffsets=True\)\)\nfor',
                     src)
        if m and text:
            rows = [ast.literal_eval(line) for line in text.strip().split("\n")]
            out["find_all_matches"].append({
                "input": m.group(1), "params": "python-default,unicodeoffsets",
                "matches": [{"input": r["input"], "begin": r["offset"]["begin"], "end": r["offset"]["end"],
                             "variants": [[v["text"], v["score"], v["dist_score"], v["freq_score"]]
                                          for v in r["variants"]]} for r in rows]})
        m = re.match(r'matches = model\.find_all_matches\("([^"]+)", SearchParameters\(unicodeoffsets=True\)\)\nprint\(matches\[3\]\)',
                     src)
        if m and text:
            r = ast.literal_eval(text.strip())
            out["find_all_matches"].append({
                "input": m.group(1), "params": "python-default,unicodeoffsets", "only_match_index": 3,
                "matches": [{"input": r["input"], "begin": r["offset"]["begin"], "end": r["offset"]["end"],
                             "variants": [[v["text"], v["score"], v["dist_score"], v["freq_score"]]
                                          for v in r["variants"]]}]})
    # cells 27-32: model2 built from a one-line transparent variant list; recorded outputs of cells 29 and 32
    out["variant_list"] = {"source": "tutorial.ipynb cells 27-32 (recorded outputs)",
                           "file_content": "separate\tseperate\t1.0\tseprate\t1.0\n", "transparent": True,
                           "params": {"max_anagram_distance": 2, "max_edit_distance": 2, "max_matches": 1}, "cases": []}
    for cell in nb["cells"]:
        if cell["cell_type"] != "code":
            continue
        src = "".join(cell["source"])
        m = re.match(r'variants = model2\.find_variants\("(\w+)"', src)
        if m:
            text = "".join("".join(o.get("text", "")) for o in cell.get("outputs", []))
            rows = [ast.literal_eval(line) for line in text.strip().split("\n") if line]
            out["variant_list"]["cases"].append({"input": m.group(1), "results": [
                [r["text"], r["score"], r["dist_score"], r["freq_score"], r.get("via")] for r in rows]})
    return out


def copy_data():
    d = os.path.join(HERE, "data")
    os.makedirs(d, exist_ok=True)
    shutil.copyfile(os.path.join(REF, "examples", "simple.alphabet.tsv"), os.path.join(d, "simple_alphabet.tsv"))
    for name in ("eng", "nld"):
        src = os.path.join(REF, "examples", f"{name}.aspell.lexicon")
        with open(src, "rb") as f, open(os.path.join(d, f"{name}_aspell.lexicon.gz"), "wb") as raw:
            with gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0, compresslevel=9) as g:
                g.write(f.read())


def twin_vectors():
    """Extra vectors from our own twin (after it is pinned): full distance tuples per pair."""
    import random
    from oracle import twin
    alphabet = twin.read_alphabet(os.path.join(REF, "examples", "simple.alphabet.tsv"))
    model = twin.VariantModel(alphabet)
    model.read_vocabulary(os.path.join(REF, "examples", "eng.aspell.lexicon"))
    model.build()
    rng = random.Random(20240601)
    words = [v.text for v in model.decoder[3:]]
    queries = ["separate", "seperate", "We", "would", "like", "beds", "a", "I", "Fo", "Kafka's", "xyzzyq",
               "it's", "O'Neil", "étude", "naïve", "co-op", "the", "teh", "recieve", "acommodate",
               "definately", "untill", "wich", "occured", "goverment", "1st", "b2b"]
    for _ in range(60):
        w = rng.choice(words)
        e = rng.choices([0, 1, 2], [0.2, 0.5, 0.3])[0]
        cs = list(w)
        for _ in range(e):
            op = rng.randrange(4)
            pos = rng.randrange(len(cs) + (1 if op == 1 else 0)) if cs else 0
            ch = chr(ord("a") + rng.randrange(26))
            if op == 0 and len(cs) > 1:
                del cs[pos]
            elif op == 1:
                cs.insert(pos, ch)
            elif op == 2 and cs:
                cs[pos] = ch
            elif op == 3 and len(cs) > 1:
                p = min(pos, len(cs) - 2)
                cs[p], cs[p + 1] = cs[p + 1], cs[p]
        if cs:
            queries.append("".join(cs))
    paramsets = {
        "lib-default": twin.SearchParameters(),
        "cli-default": twin.SearchParameters(("abs", 3), ("abs", 2), 10, 0.25, 2.0, False, 0.0),
        "test": twin.test_searchparams(),
        "ratio": twin.SearchParameters(("ratio", 0.3), ("ratiolimit", 0.25, 3), 5, 0.5, 0.0, False, 0.0),
        "exact-stop": twin.SearchParameters(("abs", 3), ("abs", 3), 20, 0.25, 2.0, True, 0.0),
        "unlimited": twin.SearchParameters(("abs", 2), ("abs", 2), 0, 0.0, 0.0, False, 0.0),
        "one": twin.SearchParameters(("abs", 3), ("abs", 3), 1, 0.25, 2.0, False, 0.0),
    }
    out = {"alphabet": "simple_alphabet.tsv", "lexicon": "eng_aspell.lexicon.gz", "cases": []}
    for pname, p in paramsets.items():
        for q in queries:
            trace = {}
            res = model.find_variants(q, p, trace)
            out["cases"].append({
                "params": pname, "input": q, "n_classes": trace["n_classes"], "n_pairs": trace["n_pairs"],
                "results": [[model.decoder[r.vocab_id].text, r.vocab_id, r.dist_score, r.freq_score] for r in res]})
    out["paramsets"] = {k: {"max_anagram_distance": list(v.max_anagram_distance),
                            "max_edit_distance": list(v.max_edit_distance), "max_matches": v.max_matches,
                            "score_threshold": v.score_threshold, "cutoff_threshold": v.cutoff_threshold,
                            "stop_at_exact_match": v.stop_at_exact_match, "freq_weight": v.freq_weight}
                        for k, v in paramsets.items()}
    return out


if __name__ == "__main__":
    json.dump(extract_tutorial(), open(os.path.join(HERE, "tutorial_outputs.json"), "w"), indent=1, ensure_ascii=False)
    copy_data()
    if "--twin" in sys.argv:
        json.dump(twin_vectors(), open(os.path.join(HERE, "twin_eng_queries.json"), "w"), indent=0, ensure_ascii=False)
    print("fixtures written")
