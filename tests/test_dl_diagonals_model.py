"""The recurrence behind k_filter_score's Damerau-Levenshtein (analiticcl_amd/csrc/kernels_score.hpp dl_diag) as a Python model, against
the twin's restatement of the reference's loop (oracle/twin.py damerau_levenshtein <- /root/reference/src/distance.rs:101-179).

L[e][k] = the furthest row on diagonal k (column = row + k) reachable with e edits.  Everything runs on MISMATCH MASKS, one per diagonal
of the band (bit i = s[i] != t[i + k], bits from 16 on set): slide(r, k) = r + ctz(M[k] >> r); the unrestricted transposition (a symbols
of s deleted and b of t inserted between the swapped pair, cost 1 + a + b) is tried at the furthest point of its source diagonal only and
is two single-bit tests of ~M.  The model below is the kernel's code line by line (same candidate terms, same order, no clamping of rows
beyond the strings' ends); the masks are built symbol by symbol here, from byte rows or symbol planes on the device.  The device code
itself is compared with the C oracle in tests/test_gpu_parity.py and the full-size tests."""
import random

import pytest

from oracle.twin import damerau_levenshtein


def masks(S, T, D, W=16):
    out = {}
    for k in range(-D, D + 1):
        m = (~0 << W) & 0xFFFFFFFF
        for i in range(W):
            j = i + k
            if not (i < len(S) and 0 <= j < len(T) and S[i] == T[j]):
                m |= 1 << i
        out[k] = m
    return out


def ctz(x):
    return (x & -x).bit_length() - 1


def dl_diag(S, T, D):
    lq, lc = len(S), len(T)
    kf = lc - lq
    if abs(kf) > D:
        return None
    M = masks(S, T, D)
    N = {k: ~M[k] & 0xFFFFFFFF for k in M}

    def slide(r, k):
        return r + ctz(M[k] >> r)
    L = [dict() for _ in range(D + 1)]
    L[0][0] = slide(0, 0)
    for e in range(1, D + 1):
        for k in range(-e, e + 1):
            v = 0
            if (k - 1) in L[e - 1]:
                v = max(v, L[e - 1][k - 1])            # insertion: same row, next column
            if (k + 1) in L[e - 1]:
                v = max(v, L[e - 1][k + 1] + 1)        # deletion
            for a in range(0, e):
                for b in range(0, e - a):
                    x, kp = 1 + a + b, k - b + a
                    if kp not in L[e - x]:
                        continue
                    r = L[e - x][kp]
                    both = N[kp + 1 + b] & (N[kp - 1 - a] >> (1 + a))
                    v = max(v, r + 1 + a + ((both >> r) & 1))   # substitution + a deletions + b insertions, one row more with the transposition
            L[e][k] = slide(v, k)
    for e in range(0, D + 1):
        if -e <= kf <= e and L[e][kf] >= lq:
            return e
    return None


def pairs(rng, n, alpha, maxlen, D):
    for _ in range(n):
        S = [rng.randrange(alpha) for _ in range(rng.randint(1, maxlen))]
        if rng.random() < 0.7:
            T = list(S)
            for _ in range(rng.randint(0, D + 1)):
                op = rng.randrange(4)
                if op == 0 and len(T) > 1:
                    T.pop(rng.randrange(len(T)))
                elif op == 1 and len(T) < maxlen:
                    T.insert(rng.randint(0, len(T)), rng.randrange(alpha))
                elif op == 2:
                    T[rng.randrange(len(T))] = rng.randrange(alpha)
                elif op == 3 and len(T) > 1:
                    i = rng.randrange(len(T) - 1)
                    T[i], T[i + 1] = T[i + 1], T[i]
        else:
            T = [rng.randrange(alpha) for _ in range(rng.randint(1, maxlen))]
        yield S, T


@pytest.mark.parametrize("D", [1, 2, 3])
@pytest.mark.parametrize("alpha", [2, 3, 5, 26])
def test_diagonal_recurrence_equals_the_reference_loop(D, alpha):
    rng = random.Random(1000 * D + alpha)
    for S, T in pairs(rng, 6000, alpha, 16, D):
        assert dl_diag(S, T, D) == damerau_levenshtein(S, T, D), (S, T)


def test_transpositions_with_gaps_and_string_ends():
    # the cases the furthest-point argument is about: a transposition whose partner lies behind a deleted / inserted symbol, at the very
    # start and the very end of 16-symbol strings, repeated symbols (several candidate partners)
    cases = [("ca", "abc", 3), ("abcdefghijklmnop", "bacdefghijklmnpo", 2), ("abcdefghijklmnop", "abcdefghijklmnpo", 1), ("aab", "aba", 1),
             ("abab", "baba", 2), ("abcd", "cbad", 3), ("abcdefgh", "abdcefhg", 2), ("abcdefgh", "abdxcefgh", 3), ("aaaaab", "baaaaa", 2),
             ("xabcdefghijklmno", "abcdefghijklmnox", 2), ("ab", "ba", 1), ("a", "b", 1), ("a", "ab", 1), ("abc", "ca", 3)]
    for s, t, D in cases:
        S, T = [ord(c) for c in s], [ord(c) for c in t]
        for d in range(1, D + 1):
            assert dl_diag(S, T, d) == damerau_levenshtein(S, T, d), (s, t, d)
