"""Late confusable rescoring on top of the C oracle (TEST INFRASTRUCTURE -- never imported by the product).

The pure-Python twin (oracle/twin.py) restates the reference's whole find_variants() including confusables, but its bigint
anagram search takes seconds per query on a 200 k-entry lexicon.  For full-size spot checks this module composes the same
steps from the two pinned parts:
  1. oracle/anx_oracle.c: find_variants() up to and including the crop (src/lib.rs:1536-1589), run with
     cutoff_threshold = 0 so that the cutoff of src/lib.rs:1598-1622 is NOT applied yet;
  2. oracle/sesdiff_twin.py: shortest_edit_script + Confusable.found_in (src/confusables.rs, src/lib.rs:1733-1756):
     dist_score *= product of the weights of the matching patterns (src/lib.rs:1656-1663), late mode (:1591-1595);
  3. rank_results again (stable, src/types.rs:344-365) and the cutoff (src/lib.rs:1598-1622) -- as in twin.score_and_rank.
tests/test_confusables_cpu.py checks this composition against the full twin.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

from oracle.sesdiff_twin import Confusable, shortest_edit_script
from oracle.twin import VariantResult


def read_confusables(path: str) -> List[Confusable]:
    out = []
    with open(path, encoding="utf-8", newline="") as f:
        for line in f.read().split("\n"):
            line = line.rstrip("\r")
            if line:
                fields = line.split("\t")
                out.append(Confusable(fields[0], float(fields[1]) if len(fields) >= 2 else 1.0))
    return out


def confusable_weight(confusables: Sequence[Confusable], text: str, candidate_text: str) -> float:
    weight = 1.0
    script = shortest_edit_script(text, candidate_text)
    for c in confusables:
        if c.found_in(script):
            weight *= c.weight
    return weight


def late_rescore(rows: Sequence[Tuple[int, float, float]], text: str, confusables: Sequence[Confusable],
                 vocab_text: Callable[[int], str], freq_weight: float, cutoff_threshold: float
                 ) -> List[Tuple[int, float, float]]:
    """rows: the cropped, ranked (vocab_id, dist_score, freq_score) list WITHOUT the cutoff applied."""
    res = [VariantResult(v, d, f) for v, d, f in rows]
    for r in res:
        r.dist_score *= confusable_weight(confusables, text, vocab_text(r.vocab_id))
    if freq_weight > 0.0:
        res.sort(key=lambda r: -r.score(freq_weight))
    else:
        res.sort(key=lambda r: (-r.dist_score, -r.freq_score))
    cutoff, best = 0, None
    if cutoff_threshold >= 1.0:
        for i, r in enumerate(res):
            if best is not None:
                if r.score(freq_weight) <= best / cutoff_threshold:
                    cutoff = i
                    break
            else:
                best = r.score(freq_weight)
    if cutoff > 0:
        del res[cutoff:]
    return [(r.vocab_id, r.dist_score, r.freq_score) for r in res]
