"""`python -m analiticcl_amd query|search ...` -- the `analiticcl query` / `analiticcl search` command line
(/root/reference/src/bin/analiticcl.rs) on top of the MI355X engine.

Same option names and defaults (note the CLI's own weight defaults 0.5/0.125/0.125/0.125/0.125, bin:760-800, and
k=3 d=2 n=10, bin:800-817), same TSV / JSON output (bin:21-187).  `query` reads one input per line and runs every
`--batch-size` lines as ONE device batch (the reference: 1000-line rayon batches, bin:416-448); `search` groups lines
into texts like bin:561-636 and decodes them with find_all_matches.  Not mirrored: learn / index modes,
--interactive buffering semantics (output is flushed per batch).  `--progress` prints the reference's "@ N - processing speed
was R items per second" lines to stderr after every batch (bin:638-654).  `--unicode-offsets` is accepted and, as in the reference
(the flag is looked up under the wrong name, bin:1175), has no effect; `--allow-overlap`, `--lm-order` and `--weight-context` are
accepted and ignored: the reference parses them into SearchParameters fields whose consumers are commented out or absent
(src/lib.rs:1905-1907).  `--debug` / `-D` may stand before or after the mode."""
import argparse
import sys
from decimal import Decimal
from typing import List, Optional

from .model import SearchParameters, VariantModel, VocabParams, Weights


def rust_f64(x: float) -> str:
    """Rust's `{}` for f64: shortest round-trip digits, never an exponent, no trailing `.0`."""
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "inf" if x > 0 else "-inf"
    s = format(Decimal(repr(x)), "f")
    if "." in s:
        s = s.rstrip("0").rstrip(".")
    return s if s not in ("", "-") else "0"


def _threshold(v: str):
    """DistanceThreshold::from_str (src/types.rs:85-108): "r;limit", integer, or ratio."""
    if ";" in v:
        r, l = v.split(";", 1)
        return (float(r), int(l))
    try:
        return int(v)
    except ValueError:
        return float(v)


def _esc(s: str) -> str:
    return s.replace('"', '\\"')


def tsv_line(inp: str, variants: Optional[List[dict]], offset=None, output_lexmatch=False) -> str:
    """output_matches_as_tsv / output_result_as_tsv (bin:21-76); `variants` already has the selected one first."""
    out = [inp]
    if offset is not None:
        out.append(f"\t{offset[0]}:{offset[1]}")
    for v in variants or []:
        out.append(f"\t{v['text']}\t{rust_f64(v['score'])}\t")
        if output_lexmatch:
            out.append('\t"' + ";".join(v["lexicons"]) + '"')
    return "".join(out)


def json_item(inp: str, variants: Optional[List[dict]], seqnr: int, offset=None, output_lexmatch=False,
              tag=(), tag_seqnr=()) -> str:
    """output_matches_as_json / output_result_as_json (bin:78-187)."""
    out = ["    ," if seqnr > 1 else "    ", '{ "input": "%s"' % _esc(inp)]
    if offset is not None:
        out.append(f', "begin": {offset[0]}, "end": {offset[1]}')
    if tag:  # bin:99-121
        out.append(', "tag": [%s], "seqnr": [ %s]' % (",".join('"%s"' % t for t in tag), ",".join(str(n) for n in tag_seqnr)))
    if variants is None:
        out.append(" }\n")
        return "".join(out)
    out.append(', "variants": [ \n')
    items = []
    for v in variants:
        s = '        { "text": "%s", "score": %s, "dist_score": %s, "freq_score": %s' % (
            _esc(v["text"]), rust_f64(v["score"]), rust_f64(v["dist_score"]), rust_f64(v["freq_score"]))
        if "via" in v:
            s += ', "via": "%s"' % _esc(v["via"])
        if output_lexmatch:
            s += ', "lexicons": [ %s ]' % ", ".join('"%s"' % _esc(x) for x in v["lexicons"])
        items.append(s + " }")
    out.append(",\n".join(items))
    out.append("\n    ] }\n")
    return "".join(out)


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="python -m analiticcl_amd", description=__doc__.split("\n\n")[0])
    p.add_argument("mode", choices=["query", "search"])
    p.add_argument("files", nargs="*", help="input files (default: standard input)")
    p.add_argument("--lexicon", "-l", action="append", default=[])
    p.add_argument("--variants", "-V", action="append", default=[])
    p.add_argument("--errors", "-E", action="append", default=[])
    p.add_argument("--alphabet", "-a", required=True)
    p.add_argument("--confusables", "-C", action="append", default=[])
    p.add_argument("--early-confusables", action="store_true")
    p.add_argument("--contextrules", "-R", action="append", default=[])
    p.add_argument("--lm", action="append", default=[])
    p.add_argument("--output-lexmatch", action="store_true")
    p.add_argument("--json", "-j", action="store_true")
    p.add_argument("--stop-exact", "-s", action="store_true")
    p.add_argument("--score-threshold", "-t", type=float, default=0.25)
    p.add_argument("--cutoff-threshold", "-T", type=float, default=2.0)
    p.add_argument("--freq-ranking", "-F", type=float, default=None)
    p.add_argument("--weight-ld", type=float, default=0.5)
    p.add_argument("--weight-lcs", type=float, default=0.125)
    p.add_argument("--weight-prefix", type=float, default=0.125)
    p.add_argument("--weight-suffix", type=float, default=0.125)
    p.add_argument("--weight-case", type=float, default=0.125)
    p.add_argument("--max-anagram-distance", "-k", default="3")
    p.add_argument("--max-edit-distance", "-d", default="2")
    p.add_argument("--max-matches", "-n", type=int, default=10)
    p.add_argument("--unicode-offsets", "-u", action="store_true")
    p.add_argument("--per-line", action="store_true")
    p.add_argument("--retain-linebreaks", action="store_true")
    p.add_argument("--max-ngram-order", "-N", type=int, default=3)
    p.add_argument("--max-seq", "-Q", type=int, default=250)
    p.add_argument("--weight-lm", type=float, default=1.0)
    p.add_argument("--weight-variant-model", type=float, default=3.0)
    p.add_argument("--weight-contextrules", type=float, default=1.0)
    p.add_argument("--batch-size", type=int, default=100000, help="query mode: lines per device batch")
    p.add_argument("--index-cache", default=None, help="image of the built model: loaded if the file exists (lexicons are then not read), written after build() otherwise")
    p.add_argument("--device", type=int, default=None)
    p.add_argument("--single-thread", "-1", action="store_true", help="accepted for compatibility, no effect")
    p.add_argument("--interactive", "-x", action="store_true", help="one device batch per input line")
    p.add_argument("--progress", action="store_true", help="Show progress (items per second on stderr after every batch)")
    p.add_argument("--debug", "-D", type=int, default=0, help="debug level 0-4 (passed to the model)")
    p.add_argument("--allow-overlap", action="store_true", help="accepted for compatibility, no effect (as in the reference)")
    p.add_argument("--lm-order", "-L", type=int, default=3, help="accepted for compatibility: the LM is a bigram model (src/lib.rs:2580-2674)")
    p.add_argument("--weight-context", type=float, default=0.0, help="accepted for compatibility, no effect (as in the reference)")
    p.add_argument("--devices", default=None, help="comma-separated HIP device ordinals: one replica of the lexicon per device, batches are "
                                                    "sharded over them inside this process (the reference's rayon fan-out, bin:445-448)")
    return p


class Progress:
    """show_progress (bin:638-654)"""

    def __init__(self, enabled: bool):
        import time
        self.enabled, self.last, self.clock = enabled, time.time(), time.time

    def show(self, seqnr: int, batchsize: int) -> None:
        if not self.enabled:
            return
        now = self.clock()
        elapsed_ms = int((now - self.last) * 1000)
        if now <= self.last or seqnr <= 1 or elapsed_ms <= 0:
            sys.stderr.write(f"@ {seqnr}\n")
        else:
            sys.stderr.write(f"@ {seqnr} - processing speed was {batchsize / (elapsed_ms / 1000.0):.0f} items per second\n")
        self.last = now


def _index_tag(a) -> str:
    """What --index-cache binds the image to: every resource file that enters build() (kind, path, size, SHA-256 of the content),
    in the order given.  (The vocabulary parameters are the CLI's fixed defaults per kind, so they are not part of the tag.)"""
    import hashlib
    import json
    import os
    items = []
    for kind in ("lexicon", "variants", "errors", "lm"):
        for f in getattr(a, kind):
            h = hashlib.sha256()
            with open(f, "rb") as fh:
                for chunk in iter(lambda: fh.read(1 << 20), b""):
                    h.update(chunk)
            items.append([kind, os.path.abspath(f), os.path.getsize(f), h.hexdigest()])
    flags = ("--lexicon", "-l", "--variants", "-V", "--errors", "-E")
    order = [t.split("=", 1)[0] for t in sys.argv if t in flags or any(t.startswith(f + "=") for f in flags if f.startswith("--"))]
    return json.dumps({"resources": items, "argv_order": order}, sort_keys=True)


def make_model(a) -> VariantModel:
    weights = Weights(ld=a.weight_ld, lcs=a.weight_lcs, prefix=a.weight_prefix, suffix=a.weight_suffix, case=a.weight_case)
    devices = [int(x) for x in a.devices.split(",")] if a.devices else None
    model = VariantModel(a.alphabet, weights, debug=a.debug, device=a.device, devices=devices)
    import os
    # The image is bound to what it was built from: alphabet (checked by the library) and the resource files in command-line
    # order with their sizes and content hashes -- an edited lexicon, another variant list or LM makes the tag differ and the
    # model is rebuilt (the reference rebuilds on every start, so it can never be stale)
    tag = _index_tag(a) if a.index_cache else None
    if a.index_cache and os.path.exists(a.index_cache) and VariantModel.index_tag_of(a.index_cache) == tag:
        try:
            model.load_index(a.index_cache)
            loaded = True
        except Exception as e:  # noqa: BLE001 -- another layout (e.g. signature groups), another alphabet, a damaged file: rebuild
            sys.stderr.write(f"[analiticcl_amd] {a.index_cache}: {e}; rebuilding the index\n")
            loaded = False
            model = VariantModel(a.alphabet, weights, debug=a.debug, device=a.device, devices=devices)
        if loaded:
            for filename in a.confusables:
                model.read_confusablelist(filename)
            for filename in a.contextrules:
                model.read_contextrules(filename)
            if a.early_confusables:
                model.set_confusables_before_pruning()
            return model
    # resources in command-line order (bin:1020-1068): lexicons, variant lists, error lists
    order = []
    argv = sys.argv
    for flagset, kind in ((("--lexicon", "-l"), "lexicon"), (("--variants", "-V"), "variants"), (("--errors", "-E"), "errors")):
        values = iter(getattr(a, kind))
        for i, tok in enumerate(argv):
            if tok in flagset or any(tok.startswith(f + "=") for f in flagset if f.startswith("--")):
                try:
                    order.append((i, kind, next(values)))
                except StopIteration:
                    pass
    if len(order) != len(a.lexicon) + len(a.variants) + len(a.errors):  # called programmatically: fall back to kind order
        order = [(0, "lexicon", f) for f in a.lexicon] + [(1, "variants", f) for f in a.variants] + [(2, "errors", f) for f in a.errors]
    for _i, kind, filename in sorted(order, key=lambda t: t[0]):
        if kind == "lexicon":
            model.read_lexicon(filename)
        else:
            model.read_variants(filename, transparent=(kind == "errors"))
    for filename in a.lm:
        model.read_vocabulary(filename, VocabParams(vocabtype="LM"))
    for filename in a.confusables:
        model.read_confusablelist(filename)
    for filename in a.contextrules:  # bin:1097-1109
        model.read_contextrules(filename)
    model.build()
    if a.index_cache:
        model.set_index_tag(tag)
        model.save_index(a.index_cache)
    if a.early_confusables:
        model.set_confusables_before_pruning()
    return model


def make_params(a) -> SearchParameters:
    if a.cutoff_threshold < 1.0 and a.cutoff_threshold != 0.0:
        sys.stderr.write("ERROR: Cutoff-threshold must be >= 1.0, or 0 to disable\n")
        sys.exit(2)
    return SearchParameters(max_anagram_distance=_threshold(a.max_anagram_distance),
                            max_edit_distance=_threshold(a.max_edit_distance), max_matches=a.max_matches,
                            score_threshold=a.score_threshold, cutoff_threshold=a.cutoff_threshold,
                            stop_criterion=a.stop_exact, freq_weight=a.freq_ranking if a.freq_ranking is not None else 0.0,
                            max_ngram=a.max_ngram_order, max_seq=a.max_seq, lm_weight=a.weight_lm,
                            variantmodel_weight=a.weight_variant_model, contextrules_weight=a.weight_contextrules,
                            unicodeoffsets=False)  # the reference reads the flag under the wrong name: inert (bin:1175)


def _lines(files):
    streams = [open(f, encoding="utf-8", newline="") for f in files] or [sys.stdin]
    for st in streams:
        for line in st:
            yield line[:-1] if line.endswith("\n") else line


def run_query(model, params, a, out) -> None:
    seqnr = 0
    batch: List[str] = []
    progress = Progress(a.progress)

    def flush():
        nonlocal seqnr
        if not batch:
            return
        # one device batch, formatted natively (anx_format_query_output = tsv_line / json_item of this module)
        out.write(model.query_output(batch, params, a.json, a.output_lexmatch, seqnr + 1))
        seqnr += len(batch)
        out.flush()
        progress.show(seqnr, len(batch))  # bin:477-479
        batch.clear()

    limit = 1 if a.interactive else max(1, a.batch_size)
    for line in _lines(a.files):
        batch.append(line)
        if len(batch) >= limit:
            flush()
    flush()


def run_search(model, params, a, out) -> None:
    """process_search (bin:561-636): consecutive lines form one text up to an empty line (or one text per line)."""
    seqnr = 0
    progress = Progress(a.progress)
    MAX_BATCHSIZE_SEARCH = 100  # bin:17
    lines = _lines(a.files)
    eof = False
    while not eof:
        text = ""
        for i in range(MAX_BATCHSIZE_SEARCH):
            try:
                inp = next(lines)
            except StopIteration:
                eof = True
                break
            if i > 0:
                text += "\n" if a.retain_linebreaks else " "
            text += inp
            if inp == "" or a.per_line:
                break
            if text == "":
                break
        # one text through find_all_matches, formatted natively (anx_format_search_output = tsv_line / json_item above)
        text_out, nmatches = model.search_output([text], params, a.json, a.output_lexmatch, seqnr + 1) if text else ("", 0)
        if seqnr > 0 and nmatches:
            out.write("\n")
        out.write(text_out)
        seqnr += nmatches
        out.flush()
        progress.show(seqnr, nmatches)  # bin:631-634


def main(argv=None) -> int:
    a = build_parser().parse_intermixed_args(argv)
    model = make_model(a)
    params = make_params(a)
    out = sys.stdout
    if a.json:
        out.write("[\n")
    (run_query if a.mode == "query" else run_search)(model, params, a, out)
    if a.json:
        out.write("]\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
