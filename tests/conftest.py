import gzip
import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
if os.path.join(REPO, "tests") not in sys.path:  # shared helpers of the test modules (fullsize_common, search_common)
    sys.path.insert(0, os.path.join(REPO, "tests"))
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def data_dir(tmp_path_factory):
    """Golden data files, decompressed once per session into a temp dir."""
    d = tmp_path_factory.mktemp("anxdata")
    src = os.path.join(GOLDEN, "data")
    with open(os.path.join(src, "simple_alphabet.tsv"), "rb") as f:
        (d / "simple.alphabet.tsv").write_bytes(f.read())
    for name in ("eng", "nld"):
        with gzip.open(os.path.join(src, f"{name}_aspell.lexicon.gz"), "rb") as f:
            (d / f"{name}.aspell.lexicon").write_bytes(f.read())
    return str(d)


@pytest.fixture(scope="session")
def tutorial_outputs():
    with open(os.path.join(GOLDEN, "tutorial_outputs.json"), encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="session")
def twin_vectors():
    with open(os.path.join(GOLDEN, "twin_eng_queries.json"), encoding="utf-8") as f:
        return json.load(f)
