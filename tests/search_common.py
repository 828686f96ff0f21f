"""Search-mode checker shared by the GPU tests: the oracle twin's segmentation / lattice / LM code with its per-segment
find_variants answered by the C oracle (same results as the twin's own, tests/test_oracle_c.py), so that thousands of
segments finish in seconds.  TEST INFRASTRUCTURE."""
from oracle import cwrap as O
from oracle import twin as T


class TwinOverOracle(T.SearchModel):
    """ids are aligned: twin and C oracle number the vocabulary in insertion order after BOS/EOS/UNK."""

    def attach(self, orc):
        self.orc = orc

    def find_variants(self, text, params, trace=None):
        cp = O.make_params(params.max_anagram_distance, params.max_edit_distance, params.max_matches,
                           params.score_threshold, params.cutoff_threshold, params.stop_at_exact_match,
                           params.freq_weight)
        return [T.VariantResult(v, d, f, via) for v, d, f, via in self.orc.find_variants_via(text, cp)]
