"""Python twin of the CPU oracle (TEST INFRASTRUCTURE, not product code).

A literal, pure-Python restatement of the reference's variant-query hot path,
using Python's native big integers for the anagram values.  It exists to
  (1) pin the semantics against the reference's own known-answer tests and the
      recorded outputs in tutorial.ipynb (tests/test_twin_golden.py),
  (2) cross-check the C oracle (oracle/anx_oracle.c) on random inputs,
  (3) generate the committed fixtures under tests/golden/ (tests/golden/make_fixtures.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Every function cites the reference file:line (relative to /root/reference) it follows.
Parity status: PINNED by tests/main.rs 01xx-04xx known answers and tutorial.ipynb outputs.
"""
from __future__ import annotations

import math
from collections import deque
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

# src/types.rs:20-30
PRIMES = [
    2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97,
    101, 103, 107, 109, 113, 127, 131, 137, 139, 149, 151, 157, 163, 167, 173, 179, 181, 191, 193,
    197, 199, 211, 223, 227, 229, 233, 239, 241, 251, 257, 263, 269, 271, 277, 281, 283, 293, 307,
    311, 313, 317, 331, 337, 347, 349, 353, 359, 367, 373, 379, 383, 389, 397, 401, 409, 419, 421,
    431, 433, 439, 443, 449, 457, 461, 463, 467, 479, 487, 491, 499, 503, 509, 521, 523, 541, 547,
    557, 563, 569, 571, 577, 587, 593, 599, 601, 607, 613, 617, 619, 631, 641, 643, 647, 653, 659,
    661, 673, 677, 683, 691, 701, 709, 719, 727, 733, 739, 743, 751, 757, 761, 769, 773, 787, 797,
    809, 811, 821, 823, 827, 829, 839, 853, 857, 859, 863, 877, 881, 883, 887, 907, 911, 919, 929,
    937, 941, 947, 953, 967, 971, 977, 983, 991, 997,
]

MAX_ANAGRAM_DISTANCE = 12  # src/lib.rs:43
MAX_EDIT_DISTANCE = 12  # src/lib.rs:46

# Rust's str::trim() strips chars with the Unicode White_Space property.
_RUST_WS = set(chr(c) for c in (
    list(range(0x09, 0x0E)) + [0x20, 0x85, 0xA0, 0x1680] + list(range(0x2000, 0x200B))
    + [0x2028, 0x2029, 0x202F, 0x205F, 0x3000]))


def rust_trim(s: str) -> str:
    b, e = 0, len(s)
    while b < e and s[b] in _RUST_WS:
        b += 1
    while e > b and s[e - 1] in _RUST_WS:
        e -= 1
    return s[b:e]


def rust_lines(data: str) -> List[str]:
    """BufRead::lines(): split on '\n', strip one trailing '\r'; no final empty line."""
    parts = data.split("\n")
    if parts and parts[-1] == "":
        parts.pop()
    return [p[:-1] if p.endswith("\r") else p for p in parts]


_LOWERCASE = None


def is_lowercase(ch: str) -> bool:
    """char::is_lowercase = the Unicode derived property Lowercase (Ll + Other_Lowercase); same database as is_alphabetic
    (Python's str.islower() is the same property of Unicode 13: three code points differ)."""
    global _LOWERCASE
    if _LOWERCASE is None:
        import regex
        _LOWERCASE = regex.compile(r"\p{Lowercase}")
    return not (0xD800 <= ord(ch) <= 0xDFFF) and _LOWERCASE.match(ch) is not None


# ---------------------------------------------------------------------------------------------
# Alphabet  (src/lib.rs:369-407)
# ---------------------------------------------------------------------------------------------
Alphabet = List[List[str]]


def parse_alphabet(data: str) -> Alphabet:
    alphabet: Alphabet = []
    for line in rust_lines(data):
        if line == "":
            continue
        fields = []
        for x in line.split("\t"):
            if x == "\\s":
                fields.append(" ")
            elif x == "\\t":
                fields.append("\t")
            elif x == "\\n":
                fields.append("\n")
            else:
                t = rust_trim(x)
                if t != "":
                    fields.append(t)
        alphabet.append(fields)
    return alphabet


def read_alphabet(path: str) -> Alphabet:
    with open(path, "r", encoding="utf-8", newline="") as f:
        return parse_alphabet(f.read())


# src/test.rs:3-31 (values of the 27-class test alphabet)
TEST_ALPHABET: Alphabet = [[c, c.upper()] for c in "abcdefghijklmnopqrstuvwxyz"] + [[".", ","]]


def _encode(text: str, alphabet: Alphabet) -> List[int]:
    """Shared scan of anahash()/normalize_to_alphabet() (src/anahash.rs:16-80).

    Returns class indices; unmatched characters are reported as -1.
    Matching is on UTF-8 byte slices; we emulate with str.startswith on code points, which is
    equivalent because alphabet members and the text are both valid UTF-8 starting at a char boundary.
    """
    out: List[int] = []
    skip = 0
    n = len(text)
    for pos in range(n):
        if skip > 0:
            skip -= 1
            continue
        matched = False
        for seqnr, chars in enumerate(alphabet):
            for element in chars:
                if text.startswith(element, pos):
                    out.append(seqnr)
                    matched = True
                    skip = len(element) - 1
                    break
            if matched:
                break
        if not matched:
            out.append(-1)
    return out


def normalize_to_alphabet(text: str, alphabet: Alphabet) -> List[int]:
    """src/anahash.rs:50-80: UNK -> alphabet.len()+1."""
    unk = len(alphabet) + 1
    return [c if c >= 0 else unk for c in _encode(text, alphabet)]


def anahash(text: str, alphabet: Alphabet) -> int:
    """src/anahash.rs:16-47: product of primes; UNK -> PRIMES[alphabet.len()]; empty = 1."""
    h = 1
    unk = len(alphabet)
    for c in _encode(text, alphabet):
        h *= PRIMES[c if c >= 0 else unk]
    return h


# ---------------------------------------------------------------------------------------------
# Anahash trait (src/anahash.rs:139-261)
# ---------------------------------------------------------------------------------------------
def av_character(seqnr: int) -> int:
    return PRIMES[seqnr]


def av_insert(a: int, b: int) -> int:
    return b if a == 0 else a * b


def av_contains(a: int, b: int) -> bool:
    if b > a:
        return False
    return a % b == 0


def av_delete(a: int, b: int) -> Optional[int]:
    return a // b if av_contains(a, b) else None


def av_is_empty(a: int) -> bool:
    return a == 1 or a == 0


# ---------------------------------------------------------------------------------------------
# Iterators (src/iterators.rs)
# ---------------------------------------------------------------------------------------------
def deletion_iterator(value: int, alphabet_size: int):
    """src/iterators.rs:51-70: yields (value_after_deletion, charindex), descending charindex."""
    if value == 1:
        return
    for iteration in range(alphabet_size):
        charindex = alphabet_size - iteration - 1
        r = av_delete(value, av_character(charindex))
        if r is not None:
            yield (r, charindex)


def recurse_deletion_iterator(value: int, alphabet_size: int, singlebeam: bool = False,
                              mindepth: Optional[int] = None, maxdepth: Optional[int] = None,
                              breadthfirst: bool = False, unique: bool = False,
                              empty_leaves: bool = True):
    """src/iterators.rs:95-235: yields ((value, charindex), depth)."""
    queue = deque([((value, 0), 0)])
    mind = 1 if mindepth is None else mindepth
    visited = set()
    while queue:
        if breadthfirst:
            node, depth = queue.popleft()
            if unique and node[0] in visited:
                continue
            if maxdepth is None or depth < maxdepth:
                for child in deletion_iterator(node[0], alphabet_size):
                    if unique and child[0] in visited:
                        continue
                    queue.append((child, depth + 1))
            if depth < mind or ((not empty_leaves) and av_is_empty(node[0])):
                continue
            if unique:
                visited.add(node[0])
            yield (node, depth)
        else:
            node, depth = queue.pop()
            if maxdepth is None or depth < maxdepth:
                if unique and node[0] in visited:
                    continue
                children = deletion_iterator(node[0], alphabet_size)
                if singlebeam:
                    for child in children:
                        queue.append((child, depth + 1))
                        break
                else:
                    children = list(children)[::-1]
                    for child in children:
                        if unique and child[0] in visited:
                            continue
                        queue.append((child, depth + 1))
            if depth < mind or ((not empty_leaves) and av_is_empty(node[0])):
                continue
            if unique:
                visited.add(node[0])
            yield (node, depth)


def av_iter(value: int, alphabet_size: int):
    """src/anahash.rs:192-204."""
    return recurse_deletion_iterator(value, alphabet_size, True, None, None, False, False, True)


def av_iter_parents(value: int, alphabet_size: int):
    return deletion_iterator(value, alphabet_size)


def av_iter_recursive(value: int, alphabet_size: int, min_distance=None, max_distance=None,
                      breadthfirst=False, allow_duplicates=True, allow_empty_leaves=True):
    """src/anahash.rs:212-228 with SearchParams defaults :272-282."""
    return recurse_deletion_iterator(value, alphabet_size, False, min_distance, max_distance,
                                     breadthfirst, not allow_duplicates, allow_empty_leaves)


def char_count(value: int, alphabet_size: int) -> int:
    return sum(1 for _ in av_iter(value, alphabet_size))


def alphabet_upper_bound(value: int, alphabet_size: int) -> Tuple[int, int]:
    """src/anahash.rs:126-136."""
    maxc, count = 0, 0
    for (node, _depth) in av_iter(value, alphabet_size):
        count += 1
        if node[1] > maxc:
            maxc = node[1]
    return maxc, count


# ---------------------------------------------------------------------------------------------
# Distances (src/distance.rs)
# ---------------------------------------------------------------------------------------------
def levenshtein(a: Sequence[int], b: Sequence[int], max_distance: int) -> Optional[int]:
    """src/distance.rs:7-82."""
    a, b = list(a), list(b)
    if a == b:
        return 0
    la, lb = len(a), len(b)
    if la == 0:
        return None if lb > max_distance else lb
    elif la > lb:
        if la - lb > max_distance:
            return None
    if lb == 0:
        return None if la > max_distance else la
    elif lb > la:
        if lb - la > max_distance:
            return None
    cache = list(range(1, la + 1))
    result = 0
    for index_b, elem_b in enumerate(b):
        result = index_b
        distance_a = index_b
        for index_a, elem_a in enumerate(a):
            distance_b = distance_a if elem_a == elem_b else distance_a + 1
            distance_a = cache[index_a]
            if distance_a > result:
                result = result + 1 if distance_b > result else distance_b
            elif distance_b > distance_a:
                result = distance_a + 1
            else:
                result = distance_b
            cache[index_a] = result
    return None if result > max_distance else result


def damerau_levenshtein(s: Sequence[int], t: Sequence[int], max_distance: int) -> Optional[int]:
    """src/distance.rs:101-179 (unrestricted DL, full matrix, literal)."""
    len_s, len_t = len(s), len(t)
    if len_s == 0:
        return None if len_t > max_distance else len_t
    elif len_s > len_t:
        if len_s - len_t > max_distance:
            return None
    if len_t == 0:
        return None if len_s > max_distance else len_s
    elif len_t > len_s:
        if len_t - len_s > max_distance:
            return None
    ub = len_t + len_s
    mat = [[0] * (len_t + 2) for _ in range(len_s + 2)]
    mat[0][0] = ub
    for i in range(len_s + 1):
        mat[i + 1][0] = ub
        mat[i + 1][1] = i
    for i in range(len_t + 1):
        mat[0][i + 1] = ub
        mat[1][i + 1] = i
    char_map: Dict[int, int] = {}
    for i0, s_char in enumerate(s):
        db = 0
        i = i0 + 1
        for j0, t_char in enumerate(t):
            j = j0 + 1
            last = char_map.get(t_char, 0)
            cost = 0 if s_char == t_char else 1
            mat[i + 1][j + 1] = min(
                mat[i + 1][j] + 1,
                mat[i][j + 1] + 1,
                mat[i][j] + cost,
                mat[last][db] + (i - last - 1) + 1 + (j - db - 1),
            )
            if cost == 0:
                db = j
        char_map[s_char] = i
    result = mat[len_s + 1][len_t + 1]
    return None if result > max_distance else result


def longest_common_substring_length(s1: Sequence[int], s2: Sequence[int]) -> int:
    """src/distance.rs:181-205."""
    lcs = 0
    n1, n2 = len(s1), len(s2)
    for i in range(n1):
        for j in range(n2):
            if s1[i] == s2[j]:
                tmp, ti, tj = 1, i + 1, j + 1
                while ti < n1 and tj < n2 and s1[ti] == s2[tj]:
                    tmp += 1
                    ti += 1
                    tj += 1
                if tmp > lcs:
                    lcs = tmp
    return lcs


def common_prefix_length(s1: Sequence[int], s2: Sequence[int]) -> int:
    """src/distance.rs:208-218."""
    n = 0
    for i in range(min(len(s1), len(s2))):
        if s1[i] == s2[i]:
            n += 1
        else:
            break
    return n


def common_suffix_length(s1: Sequence[int], s2: Sequence[int]) -> int:
    """src/distance.rs:221-231."""
    n = 0
    for i in range(min(len(s1), len(s2))):
        if s1[len(s1) - i - 1] == s2[len(s2) - i - 1]:
            n += 1
        else:
            break
    return n


# ---------------------------------------------------------------------------------------------
# Parameters (src/types.rs)
# ---------------------------------------------------------------------------------------------
@dataclass
class Weights:  # src/types.rs:40-73
    ld: float = 0.5
    lcs: float = 0.125
    prefix: float = 0.125
    suffix: float = 0.125
    case: float = 0.125

    def sum(self) -> float:
        return self.ld + self.lcs + self.prefix + self.suffix + self.case


# DistanceThreshold (src/types.rs:76-83) as a tuple: ("abs", x) | ("ratio", r) | ("ratiolimit", r, limit)
def _f32(x: float) -> float:
    import struct
    return struct.unpack("f", struct.pack("f", x))[0]


def clamp_threshold(th, length: int, absolute_max: int) -> int:
    """src/lib.rs:982-994 / 1000-1012.  Ratio arithmetic is done in f32 like the reference."""
    kind = th[0]
    if kind == "ratio":
        v = math.floor(_f32(_f32(float(length)) * _f32(th[1])))
        return min(min(max(int(v), 0), 255), absolute_max)
    if kind == "ratiolimit":
        v = math.floor(_f32(_f32(float(length)) * _f32(th[1])))
        return min(min(max(int(v), 0), 255), th[2])
    return min(th[1], min(int(math.floor(length / 2.0)), 255))


@dataclass
class SearchParameters:  # src/types.rs:112-192 (query-path subset)
    max_anagram_distance: tuple = ("abs", 3)
    max_edit_distance: tuple = ("abs", 3)
    max_matches: int = 20
    score_threshold: float = 0.25
    cutoff_threshold: float = 2.0
    stop_at_exact_match: bool = False
    freq_weight: float = 0.0


def test_searchparams() -> SearchParameters:
    """src/test.rs:48-68."""
    return SearchParameters(("abs", 2), ("abs", 2), 10, 0.0, 0.0, False, 0.0)


@dataclass
class VocabValue:  # src/vocab.rs:8-29 (query-path subset)
    text: str
    norm: List[int]
    frequency: int
    indexed: bool = True
    transparent: bool = False
    # Option<Vec<VariantReference>> (src/types.rs:315-324): ("ref_for" | "variant_of", vocab_id, score)
    variants: Optional[List[Tuple[str, int, float]]] = None
    lexindex: int = 1  # bit i: present in lexicon i (src/vocab.rs:19, src/lib.rs:941,958)


@dataclass
class VariantResult:  # src/types.rs:326-365
    vocab_id: int
    dist_score: float
    freq_score: float
    via: Optional[int] = None

    def score(self, freq_weight: float) -> float:
        if freq_weight == 0.0:
            return self.dist_score
        fw = _f32(freq_weight)
        return (self.dist_score + (fw * self.freq_score)) / (1.0 + fw)


@dataclass
class Distance:  # src/types.rs:289-305
    ld: int
    lcs: int
    prefixlen: int
    suffixlen: int
    samecase: bool


class VariantModel:
    """src/lib.rs:50-100, query path only (no LM, no variant lists, no confusables)."""

    def __init__(self, alphabet: Alphabet, weights: Optional[Weights] = None):
        self.alphabet = alphabet
        self.weights = weights or Weights()
        self.decoder: List[VocabValue] = []
        self.encoder: Dict[str, int] = {}
        self.have_freq = False
        self.lexicons: List[str] = []
        self.index: Dict[int, Tuple[List[int], int]] = {}  # anavalue -> (instances, charcount)
        self.sortedindex: Dict[int, List[int]] = {}
        # src/vocab.rs:145-181: ids 0,1,2 reserved, not INDEXED
        for t in ("<bos>", "<eos>", "<unk>"):
            self.encoder[t] = len(self.decoder)
            self.decoder.append(VocabValue(t, [], 0, indexed=False, lexindex=0))

    def alphabet_size(self) -> int:  # src/lib.rs:163-165
        return len(self.alphabet) + 1

    def add_to_vocabulary(self, text: str, frequency: Optional[int] = None,
                          freq_handling: str = "max", transparent: bool = False, lexicon_index: int = 0) -> int:
        """src/lib.rs:900-967 (INDEXED entries only)."""
        frequency = 1 if frequency is None else frequency
        vid = self.encoder.get(text)
        if vid is not None:
            item = self.decoder[vid]
            item.lexindex |= 1 << lexicon_index
            if item.transparent and not transparent and vid > 2:
                item.transparent = False  # src/lib.rs:935-940
            if freq_handling == "sum":
                item.frequency += frequency
            elif freq_handling == "max":
                item.frequency = max(item.frequency, frequency)
            elif freq_handling == "min":
                item.frequency = min(item.frequency, frequency)
            else:
                item.frequency = frequency
            return vid
        self.encoder[text] = len(self.decoder)
        self.decoder.append(VocabValue(text, normalize_to_alphabet(text, self.alphabet), frequency,
                                       transparent=transparent, lexindex=1 << lexicon_index))
        return len(self.decoder) - 1

    def add_variant(self, ref_id: int, variant: str, score: float, freq: Optional[int] = None,
                    transparent: bool = False, lexicon_index: int = 0) -> bool:
        """src/lib.rs:460-514."""
        variantid = self.add_to_vocabulary(variant, freq, transparent=transparent, lexicon_index=lexicon_index)
        if variantid == ref_id:
            return False
        ref = self.decoder[ref_id]
        if ref.variants is None:
            ref.variants = [("ref_for", variantid, score)]
        elif not any(k == "ref_for" and y == variantid for k, y, _ in ref.variants):
            ref.variants.append(("ref_for", variantid, score))
        var = self.decoder[variantid]
        if var.variants is None:
            var.variants = [("variant_of", ref_id, score)]
        elif not any(k == "variant_of" and y == variantid for k, y, _ in var.variants):  # sic: compares with variantid
            var.variants.append(("variant_of", ref_id, score))
        return True

    def read_variants(self, path: str, transparent: bool = False) -> None:
        """src/lib.rs:772-897 with VocabParams::default()."""
        with open(path, "r", encoding="utf-8", newline="") as f:
            data = f.read()
        has_freq = None
        lexicon_index = len(self.lexicons)  # src/lib.rs:784
        for line in rust_lines(data):
            if line == "":
                continue
            fields = line.split("\t")
            freq = None
            if has_freq is None:
                if (len(fields) - 2) % 3 == 0:
                    try:
                        freq = int(fields[1])
                        if freq < 0:
                            raise ValueError
                        has_freq = True
                    except ValueError:
                        freq = None
                else:
                    has_freq = False
            elif has_freq:
                freq = int(fields[1])
            ref_id = self.add_to_vocabulary(fields[0], freq, lexicon_index=lexicon_index)
            if has_freq:
                rest = fields[2:]
                for i in range(0, len(rest) - 2, 3):
                    self.add_variant(ref_id, rest[i], float(rest[i + 1]), int(rest[i + 2]), transparent, lexicon_index)
            else:
                rest = fields[1:]
                for i in range(0, len(rest) - 1, 2):
                    self.add_variant(ref_id, rest[i], float(rest[i + 1]), None, transparent, lexicon_index)
        self.lexicons.append(path)  # src/lib.rs:895

    def expand_variants(self, results: List["VariantResult"]) -> List["VariantResult"]:
        """src/lib.rs:1677-1727."""
        out = []
        for r in results:
            item = self.decoder[r.vocab_id]
            if item.variants is not None:
                for kind, target, vscore in item.variants:
                    if kind == "variant_of":
                        tf = float(self.decoder[target].frequency)
                        out.append(VariantResult(target, r.dist_score * vscore,
                                                 tf if tf < r.freq_score else r.freq_score, r.vocab_id))
            if not item.transparent:
                out.append(r)
        return out

    def read_vocabulary(self, path: str, text_column: int = 0, freq_column: Optional[int] = 1,
                        freq_handling: str = "max") -> None:
        """src/lib.rs:519-568."""
        with open(path, "r", encoding="utf-8", newline="") as f:
            data = f.read()
        lexicon_index = len(self.lexicons)  # src/lib.rs:536
        for line in rust_lines(data):
            if line == "":
                continue
            fields = line.split("\t")
            text = fields[text_column]
            if freq_column is not None:
                self.have_freq = True
                frequency = int(fields[freq_column]) if freq_column < len(fields) else 1
            else:
                frequency = 1
            self.add_to_vocabulary(text, frequency, freq_handling, lexicon_index=lexicon_index)
        self.lexicons.append(path)  # src/lib.rs:566

    def build(self) -> None:
        """src/lib.rs:192-245."""
        self.index = {}
        for vid, value in enumerate(self.decoder):
            if value.indexed:
                av = anahash(value.text, self.alphabet)
                node = self.index.get(av)
                if node is None:
                    node = ([], char_count(av, self.alphabet_size()))
                    self.index[av] = node
                node[0].append(vid)
        self.sortedindex = {}
        for av, node in self.index.items():
            self.sortedindex.setdefault(node[1], []).append(av)
        for keys in self.sortedindex.values():
            keys.sort()

    def get_anagram_instances(self, text: str) -> List[VocabValue]:  # src/lib.rs:305-318
        node = self.index.get(anahash(text, self.alphabet))
        return [self.decoder[v] for v in node[0]] if node else []

    def has(self, text: str) -> bool:  # src/lib.rs:331-338
        return any(i.text == text for i in self.get_anagram_instances(text))

    # -- hot path ---------------------------------------------------------------------------
    def find_nearest_anahashes(self, focus: int, max_distance: int,
                               stop_at_exact_match: bool = False) -> List[int]:
        """src/lib.rs:1143-1308, literal.  Returns ascending list (BTreeSet iteration order)."""
        nearest = set()
        if focus in self.index:
            nearest.add(focus)
            if stop_at_exact_match and self.index[focus][0]:
                return sorted(nearest)
        focus_upper_bound, focus_charcount = alphabet_upper_bound(focus, self.alphabet_size())
        focus_alphabet_size = focus_upper_bound + 1
        lookups: Dict[int, List[int]] = {}
        for distance in range(1, max_distance + 1):
            lookups.setdefault((focus_charcount + distance) & 0xFF, []).append(focus)
        for (node, distance) in av_iter_recursive(focus, focus_alphabet_size + 1,
                                                  max_distance=max_distance, breadthfirst=True,
                                                  allow_empty_leaves=False, allow_duplicates=False):
            deletion = node[0]
            if deletion in self.index:
                nearest.add(deletion)
            deletion_charcount = focus_charcount - distance
            for search_distance in range(1, max_distance - distance + 1):
                lookups.setdefault((deletion_charcount + search_distance) & 0xFF, []).append(deletion)
        for search_charcount, anavalues in lookups.items():
            bucket = self.sortedindex.get(search_charcount)
            if bucket is not None:
                for candidate in bucket:
                    for av in anavalues:
                        if av_contains(candidate, av):
                            nearest.add(candidate)
                            break
        return sorted(nearest)

    def gather_instances(self, nearest: List[int], querystring: List[int], query: str,
                         max_edit_distance: int) -> List[Tuple[int, Distance]]:
        """src/lib.rs:1311-1402."""
        found = []
        w = self.weights
        for av in nearest:
            for vocab_id in self.index[av][0]:
                item = self.decoder[vocab_id]
                ld = damerau_levenshtein(querystring, item.norm, max_edit_distance)
                if ld is not None:
                    found.append((vocab_id, Distance(
                        ld,
                        longest_common_substring_length(querystring, item.norm) if w.lcs > 0.0 else 0,
                        common_prefix_length(querystring, item.norm) if w.prefix > 0.0 else 0,
                        common_suffix_length(querystring, item.norm) if w.suffix > 0.0 else 0,
                        (is_lowercase(item.text[0]) == is_lowercase(query[0])) if w.case > 0.0 else True,
                    )))
        return found

    # -- confusables (src/lib.rs:409-458, 1656-1663, 1733-1756; src/confusables.rs) -------------------------------
    def add_to_confusables(self, editscript: str, weight: float) -> None:
        from oracle.sesdiff_twin import Confusable
        if not hasattr(self, "confusables"):
            self.confusables, self.confusables_before_pruning = [], False
        self.confusables.append(Confusable(editscript, weight))

    def read_confusablelist(self, filename: str) -> None:
        with open(filename, encoding="utf-8", newline="") as f:
            for line in rust_lines(f.read()):
                if line != "":
                    fields = line.split("\t")
                    self.add_to_confusables(fields[0], float(fields[1]) if len(fields) >= 2 else 1.0)

    def set_confusables_before_pruning(self) -> None:
        if not hasattr(self, "confusables"):
            self.confusables = []
        self.confusables_before_pruning = True

    def compute_confusable_weight(self, input: str, candidate: int) -> float:
        from oracle.sesdiff_twin import shortest_edit_script
        weight = 1.0
        script = shortest_edit_script(input, self.decoder[candidate].text)
        for c in self.confusables:
            if c.found_in(script):
                weight *= c.weight
        return weight

    def rescore_confusables(self, results, input: str) -> None:
        for r in results:
            r.dist_score *= self.compute_confusable_weight(input, r.vocab_id)

    def score_and_rank(self, instances, input_length: int, max_matches: int,
                       score_threshold: float, cutoff_threshold: float,
                       freq_weight: float, input: Optional[str] = None) -> List[VariantResult]:
        """src/lib.rs:1405-1653."""
        confusables = getattr(self, "confusables", [])
        early = getattr(self, "confusables_before_pruning", False)
        results: List[VariantResult] = []
        max_freq = 0.0
        has_expandable_variants = False
        w = self.weights
        weights_sum = w.sum()
        assert input_length > 0
        for vocab_id, d in instances:
            item = self.decoder[vocab_id]
            distance_score = 0.0 if d.ld > input_length else 1.0 - (float(d.ld) / float(input_length))
            lcs_score = float(d.lcs) / float(input_length)
            prefix_score = float(d.prefixlen) / float(input_length)
            suffix_score = float(d.suffixlen) / float(input_length)
            score = (w.ld * distance_score + w.lcs * lcs_score + w.prefix * prefix_score
                     + w.suffix * suffix_score + (w.case if d.samecase else 0.0)) / weights_sum
            freq_score = float(item.frequency) if self.have_freq else 1.0
            if freq_score > max_freq:
                max_freq = freq_score
            if item.variants is not None:
                has_expandable_variants = True  # src/lib.rs:1464-1466
            if score >= score_threshold:
                results.append(VariantResult(vocab_id, score, freq_score))
        if confusables and early:  # src/lib.rs:1505-1508
            self.rescore_confusables(results, input)
        if has_expandable_variants:  # src/lib.rs:1510-1518
            results = self.expand_variants(results)
            for r in results:
                if r.freq_score > max_freq:
                    max_freq = r.freq_score
        if max_freq > 0.0:
            for r in results:
                r.freq_score = r.freq_score / max_freq
        # rank_results: stable sort with rank_cmp (src/types.rs:344-365)
        def rank_results(rs):
            if freq_weight > 0.0:
                rs.sort(key=lambda r: -r.score(freq_weight))
            else:
                rs.sort(key=lambda r: (-r.dist_score, -r.freq_score))
        rank_results(results)
        if has_expandable_variants:  # Vec::dedup_by_key: consecutive duplicates only (src/lib.rs:1530-1533)
            ded = []
            for r in results:
                if not ded or ded[-1].vocab_id != r.vocab_id:
                    ded.append(r)
            results = ded
        if max_matches > 0 and len(results) > max_matches:
            last_score = results[max_matches - 1].score(freq_weight)
            cropped_score = results[max_matches].score(freq_weight)
            if cropped_score < last_score:
                del results[max_matches:]
            else:
                early_cutoff = 0
                late_cutoff = 0
                for i, r in enumerate(results):
                    if r.dist_score == cropped_score and early_cutoff == 0:
                        early_cutoff = i
                    if r.dist_score < cropped_score:
                        late_cutoff = i
                        break
                if early_cutoff > 0:
                    del results[early_cutoff + 1:]
                elif late_cutoff > 0:
                    del results[late_cutoff + 1:]
        if confusables and not early:  # late rescoring, the default (src/lib.rs:1591-1595)
            self.rescore_confusables(results, input)
            rank_results(results)
        cutoff = 0
        bestscore = None
        if cutoff_threshold >= 1.0:
            for i, r in enumerate(results):
                if bestscore is not None:
                    if r.score(freq_weight) <= bestscore / cutoff_threshold:
                        cutoff = i
                        break
                else:
                    bestscore = r.score(freq_weight)
        if cutoff > 0:
            del results[cutoff:]
        return results

    def find_variants(self, text: str, params: SearchParameters, trace: Optional[dict] = None
                      ) -> List[VariantResult]:
        """src/lib.rs:972-1027."""
        if not self.index:
            return []
        normstring = normalize_to_alphabet(text, self.alphabet)
        av = anahash(text, self.alphabet)
        k = clamp_threshold(params.max_anagram_distance, len(normstring), MAX_ANAGRAM_DISTANCE)
        nearest = self.find_nearest_anahashes(av, k, params.stop_at_exact_match)
        d = clamp_threshold(params.max_edit_distance, len(normstring), MAX_EDIT_DISTANCE)
        variants = self.gather_instances(nearest, normstring, text, d)
        if trace is not None:
            trace["n_classes"] = len(nearest)
            trace["n_pairs"] = sum(len(self.index[a][0]) for a in nearest)
            trace["distances"] = variants
        return self.score_and_rank(variants, len(normstring), params.max_matches,
                                   params.score_threshold, params.cutoff_threshold,
                                   params.freq_weight, text)


# =============================================================================================================
# Search mode: find_all_matches and what it needs.  Host-side logic in the
# reference; the hot path is called once per n-gram segment (src/lib.rs:1864-1899).
# Offsets are UTF-8 byte offsets like the reference's (unicodeoffsets remaps at the end).
# =============================================================================================================
import numpy as _np

TRANSITION_SMOOTHING_LOGPROB = _np.float32(-13.815510557964274)  # src/search.rs:4
BOS, EOS, UNK = 0, 1, 2  # src/vocab.rs:145-147


@dataclass
class SearchParams(SearchParameters):  # the search-mode fields of SearchParameters (src/types.rs:132-168)
    max_ngram: int = 3
    max_seq: int = 250
    lm_weight: float = 1.0
    variantmodel_weight: float = 3.0
    contextrules_weight: float = 1.0
    unicodeoffsets: bool = False


def test_searchparams_search() -> "SearchParams":
    """src/test.rs:48-68 (max_ngram 2)."""
    return SearchParams(("abs", 2), ("abs", 2), 10, 0.0, 0.0, False, 0.0, max_ngram=2)


@dataclass
class Match:  # src/search.rs:40-68
    text: str
    begin: int
    end: int
    variants: Optional[List[VariantResult]] = None
    selected: Optional[int] = None
    n: int = 0
    tag: List[int] = field(default_factory=list)    # src/search.rs:60-66, set by context rules
    seqnr: List[int] = field(default_factory=list)


_ALPHABETIC = None


def is_alphabetic(ch: str) -> bool:
    """char::is_alphabetic = the Unicode derived property Alphabetic (L* + Nl + Other_Alphabetic), from the `regex` module's
    Unicode database (the newest in this image); tests/test_unicode_tables_cpu.py cross-checks the compiled range tables
    against perl's database and Python's unicodedata."""
    global _ALPHABETIC
    if _ALPHABETIC is None:
        import regex
        _ALPHABETIC = regex.compile(r"\p{Alphabetic}")
    return not (0xD800 <= ord(ch) <= 0xDFFF) and _ALPHABETIC.match(ch) is not None


def _byte_offsets(text: str) -> List[int]:
    """code point index -> UTF-8 byte offset (len+1 entries)"""
    out, b = [], 0
    for ch in text:
        out.append(b)
        b += len(ch.encode("utf-8"))
    out.append(b)
    return out


def find_boundaries(text: str) -> List[Match]:
    """src/search.rs:190-233."""
    bo = _byte_offsets(text)
    boundaries: List[Match] = []
    begin = None
    for i, c in enumerate(text):
        if begin is not None:
            if is_alphabetic(c):
                boundaries.append(Match(text[begin:i], bo[begin], bo[i]))
                begin = None
        elif not is_alphabetic(c):
            begin = i
    if begin is not None:
        boundaries.append(Match(text[begin:], bo[begin], bo[len(text)]))
    else:
        boundaries.append(Match("", bo[len(text)], bo[len(text)]))
    return boundaries


def classify_boundaries(boundaries: List[Match]) -> List[str]:
    """src/search.rs:238-258; `boundary.text.len() > 1` is a BYTE length."""
    out = []
    for i, b in enumerate(boundaries):
        if i == len(boundaries) - 1:
            out.append("hard")
        elif len(b.text.encode("utf-8")) > 1:
            out.append("hard")
        elif b.text in ("'", "-", "_"):
            out.append("weak")
        else:
            out.append("normal")
    return out


def _slice_bytes(text: str, bo: List[int], b0: int, b1: int) -> str:
    i0, i1 = bo.index(b0), bo.index(b1)
    return text[i0:i1]


def internal_boundaries(m: Match, boundaries: List[Match]) -> List[Match]:
    """src/search.rs:103-120 (including its behaviour for exactly one internal boundary: empty slice)."""
    begin, end = None, 0
    for i, b in enumerate(boundaries):
        if b.begin > m.begin and b.end < m.end:
            if begin is None:
                begin = i
            else:
                end = i + 1
    if begin is None or begin >= end:
        return []
    return boundaries[begin:end]


def find_match_ngrams(text: str, boundaries: List[Match], order: int, begin: int, end: Optional[int]) -> List[Match]:
    """src/search.rs:262-313 (byte offsets)."""
    bo = _byte_offsets(text)
    ngrams: List[Match] = []
    end = bo[-1] if end is None else end
    i = 0
    while i + order - 1 < len(boundaries):
        boundary = boundaries[i + order - 1]
        if boundary.begin > end:
            break
        mt = _slice_bytes(text, bo, begin, boundary.begin) if boundary.begin >= begin else ""
        if mt != "" and mt != " ":
            ngrams.append(Match(mt, begin, boundary.begin, n=order))
        begin = boundaries[i].end
        i += 1
    if begin < end:
        mt = _slice_bytes(text, bo, begin, end)
        if mt != "" and mt != " ":
            ng = Match(mt, begin, end, n=order)
            if len(internal_boundaries(ng, boundaries)) == order:
                ngrams.append(ng)
    return ngrams


def redundant_match(candidate: Match, matches: List[Match]) -> bool:
    """src/search.rs:317-336."""
    for ref in matches:
        if ref.n == 1:
            if ref.begin >= candidate.begin and ref.end <= candidate.end:
                if ref.variants is not None:
                    if not ref.variants or ref.variants[0].dist_score < 1.0:
                        return False
                else:
                    return False
        else:
            break
    return True


def _ln(x: float) -> float:
    """f64::ln: ln(0) = -inf, ln(negative) = ln(NaN) = NaN instead of Python's ValueError."""
    if x != x or x < 0.0:
        return math.nan
    return -math.inf if x == 0.0 else math.log(x)


def _div(a: float, b: float) -> float:
    """f64 division: x/0 = +-inf, 0/0 = NaN instead of ZeroDivisionError."""
    if b == 0.0:
        return math.nan if a == 0.0 or a != a else math.copysign(math.inf, a)
    return a / b


def _parse_u8(s: str, msg: str) -> int:
    if not (s.isascii() and s.lstrip("+").isdigit() and len(s) - len(s.lstrip("+")) <= 1) or int(s) > 255:
        raise ValueError(msg)
    return int(s)


def parse_pattern(s: str, lexicons: List[str], encoder: Dict[str, int]):
    """PatternMatch::parse (src/search.rs:413-459) -> ("any",) | ("nolex",) | ("vocab", id) | ("lex", i) |
    ("not", pm) | ("or", [pm])."""
    s = rust_trim(s)
    if s == "?":
        return ("any",)
    if s == "^":
        return ("nolex",)
    if s.startswith("!(") and s.endswith(")"):
        return ("not", parse_pattern(s[2:-1], lexicons, encoder))
    if "|" in s:
        return ("or", [parse_pattern(item, lexicons, encoder) for item in s.split("|")])
    if s.startswith("!"):
        return ("not", parse_pattern(s[1:], lexicons, encoder))
    if s.startswith("@"):
        source = s[1:]
        for i, lexicon in enumerate(lexicons):
            if source == lexicon or lexicon.endswith("/" + source):
                return ("lex", i)
        raise ValueError("WARNING: Context rule references lexicon or variant list '%s' but this source was not loaded" % source)
    if s in encoder:
        return ("vocab", encoder[s])
    raise ValueError("WARNING: Context rule references word '%s' but this word does not occur in any lexicon" % s)


def pattern_matches(pm, item: Tuple[int, int]) -> bool:
    """PatternMatch::matches (src/search.rs:373-411) on one (vocab_id, lexindex)."""
    vocab_id, lexindex = item
    kind = pm[0]
    if kind == "any":
        return True
    if kind == "nolex":
        return lexindex == 0 or vocab_id == 0
    if kind == "vocab":
        return vocab_id == pm[1]
    if kind == "lex":
        return lexindex & (1 << pm[1]) == 1 << pm[1]
    if kind == "not":
        return not pattern_matches(pm[1], item)
    return any(pattern_matches(x, item) for x in pm[1])


@dataclass
class ContextRule:  # src/search.rs:355-364
    pattern: list
    score: float
    tag: List[int]
    tagoffset: List[Tuple[int, int]]


class SearchModel(VariantModel):
    """VariantModel + language-model vocabulary + context rules + find_all_matches."""

    def __init__(self, alphabet, weights=None):
        super().__init__(alphabet, weights)
        self.context_rules: List[ContextRule] = []
        self.tags: List[str] = []
        self.ngrams: Dict[tuple, int] = {}
        self.have_lm = False
        self.lm_ids: List[int] = []

    def add_lm(self, text: str, frequency: Optional[int] = None) -> int:
        """add_to_vocabulary(text, freq, VocabParams{vocab_type: LM}) (src/lib.rs:900-967)."""
        frequency = 1 if frequency is None else frequency
        vid = self.encoder.get(text)
        if vid is not None:
            item = self.decoder[vid]
            item.frequency = max(item.frequency, frequency)
            if vid not in self.lm_ids and vid > 2:
                pass  # the reference keeps the first vocab type; BOS/EOS/UNK become LM by definition
            return vid
        self.encoder[text] = len(self.decoder)
        self.decoder.append(VocabValue(text, normalize_to_alphabet(text, self.alphabet), frequency, indexed=False))
        self.lm_ids.append(len(self.decoder) - 1)
        return len(self.decoder) - 1

    def _into_ngram(self, vid: int) -> Optional[tuple]:
        """src/lib.rs:2688-2729 with use_unk = true; None for > 5 tokens."""
        text = self.decoder[vid].text
        tokencount = text.count(" ") + 1
        if tokencount > 5:
            return None
        return tuple(self.encoder.get(tok, UNK) for tok in text.split(" "))

    def build(self) -> None:
        super().build()
        self.ngrams = {}
        for vid in self.lm_ids:  # src/lib.rs:252-277
            ng = self._into_ngram(vid)
            if ng is not None:
                self.ngrams[ng] = self.ngrams.get(ng, 0) + self.decoder[vid].frequency
        self.have_lm = bool(self.ngrams)

    # -- context rules ---------------------------------------------------------------------------------------
    def add_contextrule(self, pattern: str, score: float, tag: Sequence[str] = (), tagoffset: Sequence[str] = ()) -> None:
        """src/lib.rs:658-765."""
        pms = [parse_pattern(rust_trim(e), self.lexicons, self.encoder) for e in pattern.split(";")]
        tags = []
        empty = False
        for t in tag:
            if t == "":
                empty = True
            if t not in self.tags:
                self.tags.append(t)
            tags.append(self.tags.index(t))
        if empty:
            raise ValueError("tag is empty")
        offs = []
        for spec in tagoffset:
            fields = spec.split(":")
            begin = 0 if fields[0] == "" else _parse_u8(fields[0], "tag offset should be an integer")
            if len(fields) > 1 and fields[1] != "":
                length = _parse_u8(fields[1], "tag length should be an integer")
            else:
                length = (len(pms) - begin) & 0xFF
            offs.append((begin, length))
        while len(offs) < len(tags):
            offs.append((0, len(pms)))
        self.context_rules.append(ContextRule(pms, float(_np.float32(score)), tags, offs))

    def read_contextrules(self, filename: str) -> None:
        """src/lib.rs:570-656."""
        with open(filename, "r", encoding="utf-8", newline="") as f:
            data = f.read()

        def tagfields(s):
            return [w for w in (rust_trim(x) for x in s.split(";")) if w != ""]
        for linenr, line in enumerate(rust_lines(data), 1):
            if line == "" or line.startswith("#"):
                continue
            fields = line.split("\t")
            if len(fields) < 2:
                raise ValueError("Expected at least two columns in context rules file %s, line %d" % (filename, linenr))
            if fields[0] == "":
                continue
            score = float(fields[1])
            tag = tagfields(fields[2]) if len(fields) > 2 else []
            tagoffset = tagfields(fields[3]) if len(fields) > 3 else []
            if len(tag) == 1 and not tagoffset:
                tagoffset.append("0:")
            elif len(tag) != len(tagoffset):
                raise ValueError("Multiple tags are specified for a context rule, expected the same number of tag offsets!")
            self.add_contextrule(fields[0], score, tag, tagoffset)

    def test_context_rules(self, sequence: List[Tuple[int, int]]):
        """src/lib.rs:2501-2578 with ContextRule::matches (src/search.rs:472-524): (context score, per position
        [(score, tag | None, seqnr)])."""
        results: List[list] = [[] for _ in sequence]
        found = False
        for begin in range(len(sequence)):
            for rule in self.context_rules:
                n = len(rule.pattern)
                if begin + n > len(sequence):
                    continue
                if any(results[begin + c] or not pattern_matches(rule.pattern[c], sequence[begin + c]) for c in range(n)):
                    continue
                found = True
                for c in range(n):
                    if not rule.tag:
                        results[begin + c] = [(rule.score, None, c)]
                    else:
                        results[begin + c] = [(rule.score, t, c - b) for t, (b, l) in zip(rule.tag, rule.tagoffset)
                                              if b <= c < b + l]
        if not found:
            return 1.0, results
        total = _np.float32(0.0)
        for r in results:
            total = _np.float32(total + _np.float32(r[0][0] if r else 1.0))
        return float(total) / float(len(sequence)), results

    # -- LM ------------------------------------------------------------------------------------------------
    def lm_score_tokens(self, tokens: List[Optional[int]]) -> Tuple[float, float]:
        """src/lib.rs:2632-2674: f32 logprob, f64 perplexity."""
        logprob = _np.float32(0.0)
        n = 0
        for i in range(1, len(tokens)):
            a, b = tokens[i - 1], tokens[i]
            if a is not None and b is not None:
                priorcount = self.ngrams.get((a,), 1)
                joint = self.ngrams.get((a, b))
                if joint is not None:
                    if priorcount < joint:
                        logprob = _np.float32(logprob + _np.log(_np.float32(joint)))
                    else:
                        logprob = _np.float32(logprob + _np.log(_np.float32(joint) / _np.float32(priorcount)))
                else:
                    logprob = _np.float32(logprob + TRANSITION_SMOOTHING_LOGPROB)
            else:
                logprob = _np.float32(logprob + TRANSITION_SMOOTHING_LOGPROB)
            n += 1
        return float(logprob), -1.0 / float(n) * float(logprob)

    def lm_score(self, symbols, boundaries: List[Match]) -> Tuple[float, float]:
        """src/lib.rs:2580-2629. symbols: list of (vocab_id, match_index, variant_index, boundary_index)."""
        tokens: List[Optional[int]] = [BOS]
        for vocab_id, _mi, _vi, bidx in symbols:
            nb = boundaries[bidx]
            if vocab_id == 0:
                tokens.append(None)
            else:
                ng = self._into_ngram(vocab_id)
                if ng is not None:
                    tokens.extend(ng)
            bt = rust_trim(nb.text)
            if bt != "":
                bid = self.encoder.get(bt)
                if bid is not None:
                    ng = self._into_ngram(bid)
                    if ng is not None:
                        tokens.extend(ng)
                else:
                    tokens.append(None)
        tokens.append(EOS)
        return self.lm_score_tokens(tokens)

    # -- lattice -------------------------------------------------------------------------------------------
    def most_likely_sequence(self, matches: List[Match], boundaries: List[Match], begin_offset: int,
                             end_offset: int, params: "SearchParams") -> List[Match]:
        """src/lib.rs:2088-2495 without context rules.  The reference decodes with rustfst
        shortest_path(nshortest = max_seq); here: exact k-best over the boundary DAG.  Order among equal-cost
        paths is rustfst-internal in the reference and NOT pinned (documented)."""
        f32 = _np.float32
        nstates = len(boundaries) + 1  # state 0 = start, state i+1 = boundary i
        finals = [i + 1 for i, b in enumerate(boundaries) if b.begin == end_offset or b.end == end_offset]
        assert finals, "no final state found"
        arcs: List[List[tuple]] = [[] for _ in range(nstates)]  # per source state: (cost, dst, symbol or None)
        symbols = [None]  # output symbols: (vocab_id, match_index, variant_index, boundary_index)
        for mi, m in enumerate(matches):
            prevb = nextb = None
            for i, b in enumerate(boundaries):
                if m.begin == b.end:
                    prevb = i
                elif m.end == b.begin:
                    nextb = i
            assert nextb is not None
            n = nextb - prevb if prevb is not None else nextb + 1
            src = prevb + 1 if prevb is not None else 0
            dst = nextb + 1
            if m.variants:
                for vi, vr in enumerate(m.variants):
                    symbols.append((vr.vocab_id, mi, vi, nextb))
                    cost = f32(f32(n) + f32(f32(1.0) - f32(vr.score(params.freq_weight))))
                    arcs[src].append((cost, dst, len(symbols) - 1))
            elif n == 1:
                symbols.append((0, mi, None, nextb))
                arcs[src].append((f32(f32(n) + f32(1.0)), dst, len(symbols) - 1))
        for i in range(len(boundaries)):  # failsafe epsilon transitions
            arcs[i].append((f32(100.0), i + 1, None))
        if len(symbols) == 1:
            return matches
        # k-best paths into every state (states are topologically ordered by index)
        K = params.max_seq
        best: List[List[tuple]] = [[] for _ in range(nstates)]  # (cost, symbol list)
        best[0] = [(f32(0.0), ())]
        for s in range(nstates):
            if not best[s]:
                continue
            best[s].sort(key=lambda t: float(t[0]))
            best[s] = best[s][:K]
            for cost, dst, sym in arcs[s]:
                for c0, syms in best[s]:
                    best[dst].append((f32(c0 + cost), syms + ((sym,) if sym is not None else ())))
        paths: List[tuple] = []
        for fstate in finals:
            paths.extend(best[fstate])
        paths.sort(key=lambda t: float(t[0]))
        paths = paths[:K]
        # rerank (src/lib.rs:2318-2425)
        seqs = []
        best_ppl, best_cost, best_ctx = 999999.0, f32((len(boundaries) - 1) * 2.0), 0.0
        use_lm = self.have_lm and params.lm_weight > 0.0
        for cost, syms in paths:
            osyms = [symbols[s] for s in syms]
            logprob, ppl = (0.0, 0.0)
            ctx, tags = 1.0, []
            if use_lm:
                logprob, ppl = self.lm_score(osyms, boundaries)
                best_ppl = min(best_ppl, ppl)
            if self.context_rules:  # src/lib.rs:2345-2363
                ctx, results = self.test_context_rules(
                    [(vid, self.decoder[vid].lexindex if vid != 0 else 0) for vid, _mi, _vi, _b in osyms])
                tags = [[(t, nr) for _sc, t, nr in r if t is not None] for r in results]
            if cost < best_cost:
                best_cost = cost
            if ctx > best_ctx:
                best_ctx = ctx
            seqs.append((cost, osyms, ppl, ctx, tags))
        best_score, best_seq, best_tags = -99999999.0, None, []
        lw, vw, cw = float(f32(params.lm_weight)), float(f32(params.variantmodel_weight)), float(f32(params.contextrules_weight))
        shortcut = (not self.have_lm or lw == 0.0) and (not self.context_rules or cw == 0.0)
        for cost, osyms, ppl, ctx, tags in seqs:
            norm_lm = _ln(_div(best_ppl, ppl)) if use_lm else 0.0
            # a stretch with ONE boundary leaves best_cost at 0.0 (src/lib.rs:2320): ln(0) = -inf for every path, and
            # the first path the FST yields wins (src/lib.rs:2419); here that is the cheapest one
            norm_var = _ln(_div(float(best_cost), float(cost)))
            norm_ctx = _ln(_div(ctx, best_ctx))
            if shortcut:
                score = norm_var
            else:
                with _np.errstate(all="ignore"):
                    score = float((_np.float64(lw) * norm_lm + _np.float64(vw) * norm_var + _np.float64(cw) * norm_ctx)
                                  / _np.float64(lw + vw + cw))
            if score > best_score or best_seq is None:
                best_score, best_seq, best_tags = score, osyms, tags
        out = []
        for i, (vocab_id, mi, vi, _b) in enumerate(best_seq):
            m = matches[mi]
            r = Match(m.text, m.begin, m.end, m.variants, vi, m.n)
            if best_tags:
                r.tag = [t for t, _nr in best_tags[i]]
                r.seqnr = [nr for _t, nr in best_tags[i]]
            out.append(r)
        return out

    def find_all_matches(self, text: str, params: "SearchParams") -> List[Match]:
        """src/lib.rs:1790-1957."""
        matches: List[Match] = []
        if text == "" or not self.index:
            return matches
        boundaries = find_boundaries(text)
        strengths = classify_boundaries(boundaries)
        bo = _byte_offsets(text)
        begin, begin_index = 0, 0
        for i, (strength, boundary) in enumerate(zip(strengths, boundaries)):
            if strength == "hard" and boundary.begin != begin:
                bslice = boundaries[begin_index:i + 1]
                batch: List[Match] = []
                for order in range(1, params.max_ngram + 1):
                    cur = find_match_ngrams(text, bslice, order, begin, boundary.begin)
                    for seg in cur:
                        if order == 1 or not redundant_match(seg, batch):
                            seg.variants = self.find_variants(seg.text, params)
                    batch.extend(cur)
                if params.max_ngram > 1 or self.have_lm or self.context_rules:  # src/lib.rs:1912
                    matches.extend(self.most_likely_sequence(batch, bslice, begin, boundary.begin, params))
                else:
                    for m in batch:
                        m.selected = 0
                    matches.extend(batch)
                begin = boundary.end
                begin_index = i + 1
        if params.unicodeoffsets:  # remap_offsets_to_unicodepoints (src/search.rs:527-546)
            for m in matches:
                m.begin, m.end = bo.index(m.begin), bo.index(m.end)
        return matches

    def match_to_str(self, m: Match) -> str:  # src/lib.rs:2757-2763
        if m.selected is not None and m.variants:
            return self.decoder[m.variants[m.selected].vocab_id].text
        return m.text
