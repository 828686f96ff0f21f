"""Formatting helpers of the command line (no GPU): Rust's `{}` for f64 and the TSV / JSON shapes of
/root/reference/src/bin/analiticcl.rs:21-187 (README.md:121-124 is the recorded TSV line)."""
from analiticcl_amd import cli


def test_rust_f64_display():
    assert cli.rust_f64(1.0) == "1"
    assert cli.rust_f64(0.734375) == "0.734375"
    assert cli.rust_f64(0.7499999999999999) == "0.7499999999999999"
    assert cli.rust_f64(1e-7) == "0.0000001"
    assert cli.rust_f64(0.0) == "0"
    assert cli.rust_f64(2.5e20) == "250000000000000000000"


def test_tsv_and_json_shapes():
    v = [{"text": "separate", "score": 0.734375, "dist_score": 0.734375, "freq_score": 1.0, "lexicons": ["l.tsv"]},
         {"text": 'o"perate', "score": 0.6875, "dist_score": 0.6875, "freq_score": 1.0, "via": "x", "lexicons": ["l.tsv"]}]
    assert cli.tsv_line("seperate", v) == 'seperate\tseparate\t0.734375\t\to"perate\t0.6875\t'
    assert cli.tsv_line("seperate", v, (3, 11)) .startswith("seperate\t3:11\tseparate\t0.734375\t")
    assert cli.tsv_line("x", [], None) == "x"
    assert cli.tsv_line("seperate", v[:1], None, True) == 'seperate\tseparate\t0.734375\t\t"l.tsv"'
    j = cli.json_item("sep", v, 2, (0, 3), True)
    assert j.startswith('    ,{ "input": "sep", "begin": 0, "end": 3, "variants": [ \n        { "text": "separate", "score": 0.734375, "dist_score": 0.734375, "freq_score": 1, "lexicons": [ "l.tsv" ] },\n')
    assert '"text": "o\\"perate"' in j and '"via": "x"' in j and j.endswith("\n    ] }\n")


def test_threshold_parsing():
    assert cli._threshold("3") == 3 and cli._threshold("0.3") == 0.3 and cli._threshold("0.25;3") == (0.25, 3)
