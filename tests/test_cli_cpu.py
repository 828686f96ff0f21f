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


def test_reference_flags_are_accepted():
    """Every option of the reference's query / search subcommands parses (src/bin/analiticcl.rs:656-895, 944-949), including
    the command line performance.md used (`query --progress`) and the flags whose consumers are dead in the reference."""
    from analiticcl_amd import cli
    p = cli.build_parser()
    a = p.parse_intermixed_args(["--debug", "1", "query", "--alphabet", "a.tsv", "--lexicon", "l.tsv", "--progress", "--allow-overlap",
                                 "--lm-order", "2", "--weight-context", "0.5", "-k", "2", "-d", "0.3;2", "--devices", "0,1"])
    assert a.progress and a.debug == 1 and a.allow_overlap and a.lm_order == 2 and a.weight_context == 0.5 and a.devices == "0,1"
    a = p.parse_intermixed_args(["search", "-a", "a.tsv", "-l", "l.tsv", "-D", "2", "-L", "3", "--per-line"])
    assert a.debug == 2 and a.lm_order == 3 and not a.progress


def test_progress_lines(capsys):
    from analiticcl_amd import cli
    pr = cli.Progress(True)
    pr.show(1, 1)
    t = [100.0]
    pr.clock = lambda: t[0]
    pr.last = 99.0
    pr.show(2001, 2000)     # 2000 items in 1000 ms
    err = capsys.readouterr().err.splitlines()
    assert err[0] == "@ 1" and err[1] == "@ 2001 - processing speed was 2000 items per second"
    cli.Progress(False).show(5, 5)
    assert capsys.readouterr().err == ""


def test_index_tag_sees_equals_spelling(tmp_path, monkeypatch):
    import json
    import sys
    from analiticcl_amd import cli
    f = tmp_path / "l.tsv"
    f.write_text("a\n")
    a = cli.build_parser().parse_intermixed_args(["query", "-a", "x", f"--lexicon={f}"])
    monkeypatch.setattr(sys, "argv", ["prog", "query", "-a", "x", f"--lexicon={f}"])
    assert json.loads(cli._index_tag(a))["argv_order"] == ["--lexicon"]
