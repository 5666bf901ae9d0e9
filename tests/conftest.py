import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Harness tests (subprocess runs of bench.py) are collected after every parity test, whatever the file order: under
    `pytest -x` a flake there must not hide a comparison with the oracle.  Plain `pytest tests/` on a box without a GPU
    skips the GPU tests instead of failing them."""
    items.sort(key=lambda it: 1 if "bench_harness" in it.nodeid else 0)       # stable: everything else keeps its order
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs a GPU (MI355X)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def golden_manifest():
    return json.load(open(os.path.join(GOLDEN, "manifest.json")))


def golden_records(entry):
    """The input of a golden case as a list of uint8 code arrays (regenerated or read)."""
    from debwt_amd import fasta, synth
    src = entry["source"]
    if src["kind"] == "fasta":
        return fasta.read_fasta(os.path.join(GOLDEN, entry["name"] + ".fa"))[1]
    recs = getattr(synth, src["fn"])(*src["args"])
    return [recs] if src.get("wrap") else recs


def golden_outputs(entry):
    """(bwt_words, hash_rows, dollar_row) of a golden case, or None when only hashes are stored."""
    if not entry.get("files"):
        return None
    stem = os.path.join(GOLDEN, f"{entry['name']}.k{entry['k']}")
    return (np.fromfile(stem + ".bwt", dtype=np.uint64), np.fromfile(stem + ".hash", dtype=np.uint64),
            int(np.fromfile(stem + ".dollar", dtype=np.uint64)[0]))


def golden_id(entry):
    return f"{entry['name']}-k{entry['k']}"


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O
