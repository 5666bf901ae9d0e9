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


def outside_domain_cases():
    """Inputs the reference does not handle (it crashes or mis-orders on them; 'the build must not imitate the
    failures'): a base missing from the whole collection, records of one symbol, short exact duplicates, so few branching
    positions that the SP code is shorter than one 32-symbol word."""
    rng = np.random.default_rng(4242)
    ac = lambda m: rng.integers(0, 2, size=m).astype(np.uint8)                       # noqa: E731 -- only A and C
    cases = {
        "only_A_and_C": [ac(400), ac(333), ac(90)],
        "only_G_and_T": [ac(257) + 2, ac(64) + 2],
        "one_symbol_per_record": [np.full(40, 0, np.uint8), np.full(77, 3, np.uint8), np.full(33, 1, np.uint8), np.full(120, 0, np.uint8)],
        "homopolymer_only_one_record": [np.full(500, 2, np.uint8)],
        "three_identical_40_base_records": [rng.integers(0, 4, size=40).astype(np.uint8)] * 3,
        "identical_records_of_34_bases_and_a_prefix_of_them": (lambda r: [r, r.copy(), r.copy(), np.concatenate([r, r[:7]])])(
            rng.integers(0, 4, size=34).astype(np.uint8)),
        "sp_code_shorter_than_32_symbols": [rng.integers(0, 4, size=60).astype(np.uint8)],   # a unique text: no branching node at all
        "two_records_one_branch": (lambda r: [r, np.concatenate([r[:50], (r[50:51] + 1) % 4, rng.integers(0, 4, size=40).astype(np.uint8)])])(
            rng.integers(0, 4, size=100).astype(np.uint8)),
        "tandem_repeat_of_two_symbols": [np.tile(np.array([0, 3], np.uint8), 150), np.tile(np.array([0, 3], np.uint8), 40)],
    }
    return cases
