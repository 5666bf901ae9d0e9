"""The CPU oracle against the committed golden vectors (produced by the reference's own stage
functions, tests/golden/make_golden.py), the naive definition and the inverse BWT.  CPU only."""
import hashlib

import numpy as np
import pytest

from conftest import golden_id, golden_manifest, golden_outputs, golden_records, outside_domain_cases

MANIFEST = golden_manifest()


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("entry", MANIFEST, ids=golden_id)
def test_oracle_matches_reference_golden(oracle, entry):
    recs = golden_records(entry)
    sym = oracle.sym_from_codes(recs)
    assert len(sym) == entry["n"]
    words, hrows, drow, st = oracle.build_bwt(sym, entry["k"])
    sha = entry["sha256"]
    assert _sha(words) == sha["bwt"]
    assert _sha(hrows) == sha["hash"]
    assert _sha(np.array([drow], dtype=np.uint64)) == sha["dollar"]
    files = golden_outputs(entry)
    if files:
        assert np.array_equal(words, files[0]) and np.array_equal(hrows, files[1]) and drow == files[2]
    c = entry["counters"]
    assert (st["n"], st["nrec"]) == (c["BWTLEN"], c["countRead"])
    assert st["case3num"] == c["case3num"] and st["blue_bound_num"] == c["blueBoundNum"]
    assert st["red_capacity"] == c["redCapacity"] and st["blue_capacity"] == c["blueCapacity"]
    assert st["sp_len"] + 32 == c["spCodeLen"]            # src/generateSP.c:215-216
    assert st["special_branch_num"] == c["specialBranchNum"]
    assert st["distinct_kmers"] == c["distinctKmers"]


@pytest.mark.parametrize("entry", MANIFEST, ids=golden_id)
def test_oracle_kmer_sort_matches_reference_kmerinfo(oracle, entry):
    sym = oracle.sym_from_codes(golden_records(entry))
    km, ct = oracle.kmer_count(sym, entry["k"])
    pairs = np.stack([km, ct], axis=1)                    # kmerInfo layout, src/mySort.c:193-195
    assert _sha(pairs) == entry["sha256"]["kmerInfo"]
    assert (np.diff(km.astype(object)) > 0).all() if len(km) < 5000 else (km[1:] > km[:-1]).all()


@pytest.mark.parametrize("entry", [e for e in MANIFEST if e["n"] <= 10000], ids=golden_id)
def test_golden_equals_naive_definition(oracle, entry):
    """Inside the reference's valid domain its output is the BWT by definition."""
    files = golden_outputs(entry)
    sym = oracle.sym_from_codes(golden_records(entry))
    got = oracle.unpack_bwt(files[0], entry["n"], files[1], files[2])
    assert np.array_equal(got, oracle.naive_bwt(sym))


@pytest.mark.parametrize("entry", MANIFEST, ids=golden_id)
def test_kmer_counts_equal_naive_definition(oracle, entry):
    """a-1 by definition, independent of the oracle's counter: np.unique over the k-windows of every record.  The
    reference's mySort output (the golden kmerInfo hash) and the oracle's counter must both equal it -- so the
    golden kmerInfo is the reference applied to counts that are right by definition (the generator asserts the same
    for the dump ref_driver feeds to mySort)."""
    import refformat as RF
    recs = golden_records(entry)
    km, ct = RF.naive_kmer_counts(recs, entry["k"])
    assert _sha(np.stack([km, ct], axis=1)) == entry["sha256"]["kmerInfo"]           # reference mySort == definition
    okm, oct_ = oracle.kmer_count(oracle.sym_from_codes(recs), entry["k"])
    assert np.array_equal(km, okm) and np.array_equal(ct, oct_)                      # oracle counter == definition
    assert len(km) == entry["counters"]["distinctKmers"]


@pytest.mark.parametrize("entry", MANIFEST, ids=golden_id)
def test_oracle_intermediates_match_reference(oracle, entry):
    """The oracle's red table and SP code against what the reference's generateBlocks / generateSP wrote
    (redSeq, src/INandOut.c:396-404; spCode + spSpecialIndex, src/generateSP.c:626-660)."""
    import refformat as RF
    sym = oracle.sym_from_codes(golden_records(entry))
    out = oracle.build_bwt(sym, entry["k"], want_intermediates=True)
    sp, red = out[4], out[5]
    assert _sha(RF.red_seq(red, entry["k"])) == entry["sha256"]["redSeq"]
    assert _sha(sp) == entry["sha256"]["spSymbols"]


def test_naive_kmer_counts_on_adversarial_inputs(oracle):
    import refformat as RF
    for seed in range(10):
        rng = np.random.default_rng(7000 + seed)
        recs = _random_collection(rng)
        for k in (12, 21, 32):
            km, ct = RF.naive_kmer_counts(recs, k)
            okm, oct_ = oracle.kmer_count(oracle.sym_from_codes(recs), k)
            assert np.array_equal(km, okm) and np.array_equal(ct, oct_)
            assert int(ct.sum()) == sum(len(r) - k + 1 for r in recs)


def _random_collection(rng):
    nrec = int(rng.integers(1, 6))
    base = rng.integers(0, 4, size=int(rng.integers(60, 300))).astype(np.uint8)
    recs = []
    for _ in range(nrec):
        L = int(rng.integers(33, 700))
        x = rng.integers(0, 4, size=L).astype(np.uint8)
        if rng.random() < 0.7:
            seg = base[:min(len(base), L)]
            p = int(rng.integers(0, L - len(seg) + 1))
            x[p:p + len(seg)] = seg
        if rng.random() < 0.3:
            x[-min(L, 40):] = base[:min(L, 40)]
        if rng.random() < 0.2:
            x[:min(L, 50)] = 0
        recs.append(x)
    if rng.random() < 0.3:
        recs.append(recs[0].copy())
    if rng.random() < 0.2:
        recs.append(recs[-1][:max(33, len(recs[-1]) // 2)].copy())
    return recs


@pytest.mark.parametrize("seed", range(25))
def test_oracle_equals_naive_on_adversarial_inputs(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    recs = _random_collection(rng)
    sym = oracle.sym_from_codes(recs)
    ref = oracle.naive_bwt(sym)
    outs = []
    for k in (12, 17, 32):
        words, hrows, drow, _ = oracle.build_bwt(sym, k)
        got = oracle.unpack_bwt(words, len(sym), hrows, drow)
        assert np.array_equal(got, ref), f"k={k}"
        outs.append(words)
        rc, inv = oracle.inverse_bwt(words, len(sym), hrows, drow)
        assert rc == 0 and np.array_equal(inv, sym)
    assert all(np.array_equal(outs[0], w) for w in outs[1:])   # k-invariance


def test_oracle_roundtrip_midsize(oracle):
    from debwt_amd import synth
    recs = synth.chromosomes(400_000, 7)
    sym = oracle.sym_from_codes(recs)
    words, hrows, drow, st = oracle.build_bwt(sym, 32)
    assert len(hrows) == 6 and (np.diff(hrows.astype(np.int64)) > 0).all()
    rc, inv = oracle.inverse_bwt(words, len(sym), hrows, drow)
    assert rc == 0 and np.array_equal(inv, sym)
    w2, h2, d2, _ = oracle.build_bwt(sym, 20)
    assert np.array_equal(words, w2) and np.array_equal(hrows, h2) and drow == d2


def test_oracle_rejects_bad_input(oracle):
    with pytest.raises(ValueError):
        oracle.make_text(["ACGT" * 8])            # 32 bases: src/collect#$.c:41-45
    with pytest.raises(ValueError):
        oracle.make_text(["ACGTN" * 10])


def test_pack_text_layout(oracle):
    sym = oracle.make_text(["ACGT" * 10, "TTGCA" * 8])
    w = oracle.pack_text(sym)
    n = len(sym)
    assert len(w) == (n + 32 + 31) // 32
    for j in range(n + 32):
        c = int(w[j >> 5] >> np.uint64(2 * (31 - (j & 31)))) & 3
        assert c == (int(sym[j]) if j < n and sym[j] < 4 else 3)


@pytest.mark.parametrize("name", sorted(outside_domain_cases()))
def test_oracle_equals_the_definition_outside_the_reference_domain(oracle, name):
    """inputs the reference crashes or mis-orders on (a base that never occurs, homopolymer-only records, short
    exact duplicates, an SP code below 32 symbols): the oracle -- the checker of the GPU path on exactly these inputs,
    tests/test_gpu_parity.py -- gives the BWT by definition there, for k = 12, 20 and 32."""
    recs = outside_domain_cases()[name]
    sym = oracle.sym_from_codes(recs)
    want = oracle.naive_bwt(sym)
    for k in (12, 20, 32):
        w, h, d, _ = oracle.build_bwt(sym, k)
        assert np.array_equal(oracle.unpack_bwt(w, len(sym), h, d), want), (name, k)
