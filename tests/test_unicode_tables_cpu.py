"""The Unicode range tables the product compiles in (analiticcl_amd/csrc/unicode_tables.inc: char::is_alphabetic for the
boundaries of search mode, /root/reference/src/search.rs:190-233; char::is_lowercase for `samecase`, src/lib.rs:1367-1377)
checked against sources they were NOT generated from: the `regex` module's Unicode database, Python's str methods, and
hand-listed boundary cases from the Unicode categories (Lo, Lt, Lm, Nl, Other_Alphabetic, Other_Lowercase).
The tables come from the `regex` module's Unicode database (oracle/gen_unicode.py); the checks use perl's Unicode database,
Python's unicodedata / str methods and literal code points."""
import os
import re
import unicodedata

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _table(name):
    src = open(os.path.join(REPO, "analiticcl_amd", "csrc", "unicode_tables.inc")).read()
    body = src[src.index(f"anx_uc_{name}[][2]"):]
    body = body[:body.index("};")]
    rs = [(int(a, 16), int(b, 16)) for a, b in re.findall(r"\{0x([0-9A-F]+),0x([0-9A-F]+)\}", body)]
    assert rs == sorted(rs) and all(a <= b for a, b in rs) and all(rs[i][1] < rs[i + 1][0] for i in range(len(rs) - 1))
    n = int(re.search(rf"anx_uc_{name}_n = (\d+)", src).group(1))
    assert n == len(rs)
    member = bytearray(0x110000)
    for a, b in rs:
        for c in range(a, b + 1):
            member[c] = 1
    return member


def _perl_property(prop):
    import subprocess
    out = subprocess.check_output(["perl", "-e", r'''
        my $p = shift; my $s = -1;
        for my $c (0 .. 0x110000) {
          my $ok = ($c < 0x110000 && !($c >= 0xD800 && $c <= 0xDFFF) && chr($c) =~ /\p{$p}/) ? 1 : 0;
          if ($ok && $s < 0) { $s = $c; } elsif (!$ok && $s >= 0) { printf "%X %X\n", $s, $c - 1; $s = -1; }
        }''', prop], text=True)
    member = bytearray(0x110000)
    for line in out.split("\n"):
        if line:
            a, b = (int(x, 16) for x in line.split())
            for c in range(a, b + 1):
                member[c] = 1
    return member


def test_alphabetic_table_vs_perl_and_unicodedata():
    """The table comes from the regex module's database (Unicode 17); perl's and Python's are Unicode 13.  For a code point
    assigned in 13 the property is stable except for documented corrections (Unicode 16 added Other_Alphabetic to the combining
    Latin letters U+0363..036F, U+1DD3..1DE6 and a few signs): the tables may differ there -- only there, only by ADDING."""
    import shutil
    alpha = _table("alpha")
    for cp in range(0xD800, 0xE000):
        assert not alpha[cp]
    # every letter and letter number of Unicode 13 is alphabetic
    for cp in range(0x110000):
        if not (0xD800 <= cp <= 0xDFFF) and (unicodedata.category(chr(cp))[0] == "L" or unicodedata.category(chr(cp)) == "Nl"):
            assert alpha[cp], hex(cp)
    if shutil.which("perl") is None:
        return
    ref = _perl_property("Alphabetic")
    assert sum(ref) > 130000
    added, removed = [], []
    for cp in range(0x110000):
        if 0xD800 <= cp <= 0xDFFF or unicodedata.category(chr(cp)) == "Cn":
            continue                            # unassigned in Unicode 13: perl knows nothing about it
        if alpha[cp] and not ref[cp]:
            added.append(cp)
        elif ref[cp] and not alpha[cp]:
            removed.append(cp)
    assert removed == [], [hex(c) for c in removed[:20]]
    assert len(added) <= 64 and all(unicodedata.category(chr(c)) in ("Mn", "Mc") for c in added), [hex(c) for c in added[:20]]


def test_alphabetic_hand_listed_cases():
    alpha = _table("alpha")
    yes = ["a", "Z", "é", "ß", "ª", "º", "µ",            # Ll / Lo in Latin-1
           "ǅ", "ᾈ",                                       # Lt
           "ʰ", "ˑ", "ᴬ",                                  # Lm
           "א", "ش", "अ", "日", "한", "ぁ",                 # Lo
           "Ⅷ", "ⅷ", "〇", "ᛮ",                           # Nl
           "ͅ", "ְ", "ؐ", "ा", "ઁ", "ั", "ᜒ",   # Other_Alphabetic: marks, vowel signs
           "Ⓐ", "ⓩ", "\U0001F130"]                        # Other_Alphabetic: circled / squared letters (So)
    no = ["0", "9", " ", "-", "_", "'", ".", "€", "²", "½", "٣", "́", "‍", "­", "⃝", "①", "☃", "\U0001F600",
          "͸", "\U000E0041"]                           # digits, punctuation, No, Nd, Mn without Other_Alphabetic, Cf, Me, So, Cn, tag
    for ch in yes:
        assert alpha[ord(ch)], (ch, hex(ord(ch)))
    for ch in no:
        assert not alpha[ord(ch)], (ch, hex(ord(ch)))


def test_lowercase_table():
    lower = _table("lower")
    diff = [cp for cp in range(0x110000) if not (0xD800 <= cp <= 0xDFFF) and unicodedata.category(chr(cp)) != "Cn" and
            bool(lower[cp]) != chr(cp).islower()]    # str.islower(): the Lowercase property (Ll + Other_Lowercase) of Unicode 13
    assert len(diff) <= 8, [hex(c) for c in diff]    # reclassified since: U+0295, U+10FC, U+AB69
    for ch in ("a", "ß", "ª", "º", "ʰ", "ⓐ", "ⅷ", "ͅ"):
        assert lower[ord(ch)], ch
    for ch in ("A", "ǅ", "1", "日", "Ⅷ", "Ⓐ", " "):
        assert not lower[ord(ch)], ch


def test_twin_and_product_agree_on_boundaries():
    """find_boundaries of the twin (regex-module Alphabetic) on text mixing the hand-listed characters: every maximal run of
    non-alphabetic characters is a boundary (src/search.rs:190-233); the product's table must classify each character alike."""
    from oracle import twin as T
    alpha = _table("alpha")
    text = "abͅc Ⓐ-①x ٣३ Ⅷ, é́日 ½²"
    for ch in text:
        assert T.is_alphabetic(ch) == bool(alpha[ord(ch)]), (ch, hex(ord(ch)))
    bounds = [(b.begin, b.end) for b in T.find_boundaries(text)]
    assert bounds and all(not T.is_alphabetic(c) for b, e in bounds for c in text.encode()[b:e].decode())
