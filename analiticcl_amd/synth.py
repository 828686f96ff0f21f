"""Synthetic query generator of SURVEY.md section 8(d) (bench + tests input; not part of the scored path).

seed 20240601; pick a lexicon entry uniformly among entries whose length is in range; apply
e in {0,1,2} edits (p = .2/.5/.3) drawn uniformly from {delete, insert a-z, substitute a-z, adjacent
transpose}; reject empty / over-long results; original casing is kept.
"""
from __future__ import annotations

import gzip
import os
import random
from typing import List, Sequence

SEED = 20240601
GOLDEN_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")


def load_lexicon_words(path: str) -> List[str]:
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt", encoding="utf-8", newline="") as f:
        return [line.split("\t")[0] for line in f.read().split("\n") if line]


def materialize_golden(dst_dir: str) -> dict:
    """Decompress the golden data files (tests/golden/data) into dst_dir; returns their paths."""
    os.makedirs(dst_dir, exist_ok=True)
    out = {"alphabet": os.path.join(dst_dir, "simple.alphabet.tsv")}
    with open(os.path.join(GOLDEN_DATA, "simple_alphabet.tsv"), "rb") as f, open(out["alphabet"], "wb") as g:
        g.write(f.read())
    for name in ("eng", "nld"):
        out[name] = os.path.join(dst_dir, f"{name}.aspell.lexicon")
        if not os.path.exists(out[name]):
            with gzip.open(os.path.join(GOLDEN_DATA, f"{name}_aspell.lexicon.gz"), "rb") as f, open(out[name], "wb") as g:
                g.write(f.read())
    return out


def make_queries(words: Sequence[str], n: int, max_len: int = 16, min_len: int = 1, seed: int = SEED) -> List[str]:
    rng = random.Random(seed)
    pool = [w for w in words if min_len <= len(w) <= max_len]
    letters = "abcdefghijklmnopqrstuvwxyz"
    out: List[str] = []
    choice, rnd, randrange = rng.choice, rng.random, rng.randrange
    while len(out) < n:
        cs = list(choice(pool))
        r = rnd()
        e = 0 if r < 0.2 else (1 if r < 0.7 else 2)
        for _ in range(e):
            op = randrange(4)
            if op == 0:
                if len(cs) > 1:
                    del cs[randrange(len(cs))]
            elif op == 1:
                cs.insert(randrange(len(cs) + 1), letters[randrange(26)])
            elif op == 2:
                cs[randrange(len(cs))] = letters[randrange(26)]
            elif len(cs) > 1:
                p = randrange(len(cs) - 1)
                cs[p], cs[p + 1] = cs[p + 1], cs[p]
        if cs and len(cs) <= max_len:
            out.append("".join(cs))
    return out


def make_queries_with_quota(words: Sequence[str], quota: dict, max_len: int = 16, min_len: int = 1, seed: int = SEED) -> List[str]:
    """Queries from the law of make_queries(words, ., max_len, min_len) CONDITIONED on their length: quota[L] queries of every length
    L in quota (one GPU's share of a length-partitioned job, BASELINE configs[3]).  Rejection sampling from the entries that can
    reach those lengths (an edit changes the length by at most one, <= 2 edits), same edit process: exactly the conditional law."""
    rng = random.Random(seed)
    lo, hi = min(quota), max(quota)
    pool = [w for w in words if max(min_len, lo - 2) <= len(w) <= min(max_len, hi + 2)]
    left = dict(quota)
    need = sum(left.values())
    letters = "abcdefghijklmnopqrstuvwxyz"
    out: List[str] = []
    choice, rnd, randrange = rng.choice, rng.random, rng.randrange
    while need > 0:
        cs = list(choice(pool))
        r = rnd()
        e = 0 if r < 0.2 else (1 if r < 0.7 else 2)
        for _ in range(e):
            op = randrange(4)
            if op == 0:
                if len(cs) > 1:
                    del cs[randrange(len(cs))]
            elif op == 1:
                cs.insert(randrange(len(cs) + 1), letters[randrange(26)])
            elif op == 2:
                cs[randrange(len(cs))] = letters[randrange(26)]
            elif len(cs) > 1:
                p = randrange(len(cs) - 1)
                cs[p], cs[p + 1] = cs[p + 1], cs[p]
        n = len(cs)
        if left.get(n, 0) > 0 and n <= max_len:
            left[n] -= 1
            need -= 1
            out.append("".join(cs))
    return out


def make_lexicon(words: Sequence[str], n: int, min_len: int = 4, max_len: int = 32, seed: int = SEED) -> List[str]:
    """BASELINE.json configs[3]: the given words plus seeded order-2 Markov-chain words (trained on them) up to n
    distinct entries; lengths min_len..max_len."""
    rng = random.Random(seed)
    trans: dict = {}
    for w in words:
        lw = "^^" + w.lower() + "$"
        for i in range(len(lw) - 2):
            trans.setdefault(lw[i:i + 2], []).append(lw[i + 2])
    out = list(dict.fromkeys(w for w in words if len(w) <= max_len))
    seen = set(out)
    while len(out) < n:
        target = rng.randrange(min_len, max_len + 1)
        cs, ctx = [], "^^"
        while len(cs) < target:
            nxt = rng.choice(trans.get(ctx) or "$")
            if nxt == "$":
                if len(cs) >= min_len and rng.random() < 0.5:
                    break
                ctx = "^^" if not cs else ctx[1] + rng.choice("aeiou")
                cs.append(ctx[1])
                continue
            cs.append(nxt)
            ctx = ctx[1] + nxt
        w = "".join(cs)
        if len(w) >= min_len and w not in seen:
            seen.add(w)
            out.append(w)
    return out[:n]


def make_running_text(words: Sequence[str], megabytes: float, seed: int = SEED, sentences_per_text: int = 8) -> List[str]:
    """BASELINE.json configs[4] shape: running text of sentences of 5-25 perturbed lexicon words (make_queries: 0-2 edits
    each) ending in ". " / newline / ", " / an empty line; `sentences_per_text` sentences per input text."""
    rng = random.Random(seed)
    pert = make_queries(words, int(megabytes * 1e6 / 7) + 100, max_len=16, seed=seed + 1)
    texts: List[str] = []
    cur: List[str] = []
    size, k = 0, 0
    while size < megabytes * 1e6:
        n = rng.randrange(5, 26)
        if k + n > len(pert):
            k = 0
        cur.append(" ".join(pert[k:k + n]) + rng.choice([". ", "\n", ", ", "\n\n"]))
        k += n
        if len(cur) == sentences_per_text:
            texts.append("".join(cur))
            size += len(texts[-1].encode("utf-8"))
            cur = []
    return texts
