"""GPU parity: the HIP path (through the C ABI of libdebwt_hip.so) against the CPU oracle, the
committed reference golden vectors and size-independent properties.  Bit-exact everywhere: the path is
64-bit integer arithmetic only."""
import hashlib

import numpy as np
import pytest

from conftest import outside_domain_cases, golden_id, golden_manifest, golden_outputs, golden_records

pytestmark = pytest.mark.gpu
MANIFEST = golden_manifest()


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def api():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    from debwt_amd import api as A
    return A


def _run(api, recs, k, algo=0):
    d = api.DeBWT(k=k, sort_algo=algo)
    d.load_records(recs)
    d.build()
    out = d.fetch()
    st = d.stats()
    return d, out, st


@pytest.mark.parametrize("entry", MANIFEST, ids=golden_id)
def test_hip_matches_reference_golden(api, entry):
    recs = golden_records(entry)
    d, (words, hrows, drow), st = _run(api, recs, entry["k"])
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
    assert st["sp_len"] + 32 == c["spCodeLen"] and st["special_branch_num"] == c["specialBranchNum"]
    d.close()


@pytest.mark.parametrize("entry", MANIFEST, ids=golden_id)
def test_hip_intermediates_match_reference_files(api, entry):
    """the arrays the stages hand to each other, fetched from the HIP build (debwt_fetch_array) and put
    into the reference's formats, against the files the reference's own generateBlocks / generateSP wrote
    (redSeq, redPoint, blueBound, case3bound: src/INandOut.c:347-366,396-417; spCode, blueTable:
    src/generateSP.c:626-672) -- sha256 in the golden manifest, made by tests/golden/make_golden.py."""
    import refformat as RF
    d = api.DeBWT(k=entry["k"])
    d.load_records(golden_records(entry))
    d.kmer_sort_rle()
    d.classify()
    red = d.fetch_array(api.ARR_RED)
    bb = d.fetch_array(api.ARR_BLUE_BOUND)
    sha = entry["sha256"]
    assert _sha(RF.red_seq(red, entry["k"])) == sha["redSeq"]
    assert _sha(bb) == sha["blueBound"]
    assert _sha(RF.red_point(red, bb)) == sha["redPoint"]
    assert _sha(d.fetch_array(api.ARR_CASE3_BOUND)) == sha["case3bound"]
    d.sp_generate()
    assert _sha(d.fetch_array(api.ARR_SP_SYMBOLS)) == sha["spSymbols"]
    assert _sha(RF.blue_blocks_sorted(d.fetch_array(api.ARR_BLUE), bb)) == sha["blueBlocks"]
    d.close()


@pytest.mark.parametrize("entry", [e for e in MANIFEST if e["n"] < 400000], ids=golden_id)
def test_hip_kmer_count_matches_reference_kmerinfo(api, entry):
    d = api.DeBWT(k=entry["k"])
    d.load_records(golden_records(entry))
    km, ct = d.kmer_count_sorted()
    assert _sha(np.stack([km, ct], axis=1)) == entry["sha256"]["kmerInfo"]     # src/mySort.c:193-195
    d.close()


@pytest.mark.parametrize("entry", [e for e in MANIFEST if e["records"] >= 2000], ids=golden_id)
def test_many_record_goldens_take_the_parallel_special_region_path(api, entry):
    """collections of thousands of records put N*(k-1) >= 2^14 suffixes through the special-region module's
    parallel form (host threads or the device) -- checked against the reference's own output for 2,000 contigs and
    20,000 reads (tests/golden: made by the reference's stage functions, -t 1), and again with the module forced onto
    one host thread: same bytes."""
    import os
    recs = golden_records(entry)
    assert entry["records"] * (entry["k"] - 1) >= 1 << 14
    outs = []
    for force_serial in (False, True):
        if force_serial:
            os.environ["DEBWT_SPECIAL_PAR_MIN"] = str(1 << 62)
            os.environ["DEBWT_SPECIAL_DEVICE_MIN"] = str(1 << 62)
        try:
            d, (words, hrows, drow), st = _run(api, recs, entry["k"])
        finally:
            os.environ.pop("DEBWT_SPECIAL_PAR_MIN", None)
            os.environ.pop("DEBWT_SPECIAL_DEVICE_MIN", None)
        assert (st["special_path"] == 0) == force_serial, st
        assert force_serial or st["special_path"] == 2 or st["special_threads"] > 1, st
        sha = entry["sha256"]
        assert _sha(words) == sha["bwt"] and _sha(hrows) == sha["hash"]
        assert _sha(np.array([drow], dtype=np.uint64)) == sha["dollar"]
        assert st["special_branch_num"] == entry["counters"]["specialBranchNum"]
        outs.append(words)
        d.close()
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("entry", [e for e in MANIFEST if e["records"] >= 2000 and e["k"] == 32], ids=golden_id)
def test_special_region_module_falls_back_to_host_when_its_workspace_does_not_fit(api, entry, monkeypatch):
    """The device module is refused when its workspace (72 bytes per special suffix) does not fit beside the build
    (special_wants_device; DEBWT_SPECIAL_FAKE_FREE_BYTES stands in for hipMemGetInfo): the host threads build the tables
    and the result is the reference's all the same; with the real figure the device module runs."""
    recs = golden_records(entry)
    sha = entry["sha256"]
    for fake, want_path in (("1000000", (0, 1)), (None, (2,))):
        if fake:
            monkeypatch.setenv("DEBWT_SPECIAL_FAKE_FREE_BYTES", fake)
        else:
            monkeypatch.delenv("DEBWT_SPECIAL_FAKE_FREE_BYTES", raising=False)
        d, (words, hrows, drow), st = _run(api, recs, 32)
        assert st["special_path"] in want_path, st
        assert _sha(words) == sha["bwt"] and _sha(hrows) == sha["hash"]
        d.close()


def test_special_compare_leaves_the_last_build_alone(api):
    """debwt_special_compare overwrites the special-region tables of a build in progress: the context is back at "text
    loaded" afterwards (stage calls out of order are refused, a new build is right) and the counters of the last build stay."""
    from debwt_amd import synth
    recs = synth.read_set(3000, 60, 300, 200_000, seed=11)
    d = api.DeBWT(k=32)
    d.load_records(recs)
    d.build()
    want = d.fetch()
    d.kmer_sort_rle()
    st = d.stats()
    assert d.special_compare() == [0] * 6
    st2 = d.stats()
    assert st2["special_path"] == st["special_path"] and st2["ms_host_special"] == st["ms_host_special"]
    with pytest.raises(api.DebwtError):
        d.classify()                                     # the sort of the interrupted build is gone
    d.build()
    got = d.fetch()
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2] == want[2]
    d.close()


def test_hundred_thousand_records_equal_oracle(api, oracle):
    """10^5 reads (beyond what the reference's O(N)-per-record insert, src/INandOut.c:91-108, finishes in test time):
    3.1e6 special suffixes through the parallel special-region path against the oracle, whose special-region code is
    pinned by the 2,000- and 20,000-record goldens."""
    from debwt_amd import synth
    recs = synth.read_set(100_000, 60, 300, 3_000_000, seed=0xBEEF5)
    ow, oh, od, ost = oracle.build_bwt(oracle.sym_from_codes(recs), 32)
    d, (words, hrows, drow), st = _run(api, recs, 32)
    assert st["special_path"] == 2 or st["special_threads"] > 1, st
    assert np.array_equal(words, ow) and np.array_equal(hrows, oh) and drow == od
    assert st["special_branch_num"] == ost["special_branch_num"] and st["nrec"] == 100_000
    assert d.verify_device()["inverse_bwt_ok"]
    d.close()


@pytest.mark.parametrize("case", ["adversarial", "reads", "contigs_dup", "reads_20000", "single_record"])
@pytest.mark.parametrize("max_rounds", [None, 2, 0, "2 on the host"])
def test_special_region_tables_device_equals_host(api, case, max_rounds, monkeypatch):
    """the special-region tables (suffix order of the N*K special suffixes, their keys and BWT symbols, the
    special branches, head and tail nodes: src/collect#$.c:118-157,428-455,468-598) built by the device module against
    the host module, element by element (debwt_special_compare); max_rounds = 2 (0) puts every tie group of record starts
    that is still tied after 42 symbols (every group) through the JUMP rounds -- the path of records identical for thousands
    of symbols: the group's depth advances by what all its members share with its head, then one window round -- and
    "2 on the host" through the host comparison that did this job until round 5 (DEBWT_SPECIAL_HOST_TIES)."""
    from debwt_amd import synth
    if max_rounds is not None:
        monkeypatch.setenv("DEBWT_SPECIAL_MAX_ROUNDS", str(max_rounds).split()[0])
        if "host" in str(max_rounds):
            monkeypatch.setenv("DEBWT_SPECIAL_HOST_TIES", "1")
    rng = np.random.default_rng(4242)
    if case == "adversarial":
        sets = [(_adversarial(rng), int(rng.choice([12, 16, 20, 27, 32]))) for _ in range(25)]
    elif case == "reads":
        sets = [(synth.read_set(int(rng.integers(2, 600)), 50, int(rng.integers(60, 300)), 20_000, seed=int(rng.integers(1, 1 << 30)),
                                snap=int(rng.choice([1, 4, 8])), dup_every=int(rng.choice([0, 3, 16]))), int(rng.choice([12, 21, 32])))
                for _ in range(15)]
    elif case == "contigs_dup":       # exact duplicates of 3-8 kb: tied for hundreds of windows
        sets = [(synth.read_set(2000, 3000, 8000, 4_000_000), 32), (synth.read_set(300, 3000, 8000, 400_000, dup_every=2), 20)]
    elif case == "reads_20000":
        sets = [(synth.read_set(20000, 60, 400, 1_000_000), 32), (synth.read_set(20000, 60, 400, 1_000_000), 16)]
    else:
        sets = [([synth.uniform_codes(5000)], 32), (synth.pan_genome(3000, 2), 12)]
    for recs, k in sets:
        d = api.DeBWT(k=k)
        d.load_records(recs)
        assert d.special_compare() == [0, 0, 0, 0, 0, 0], (case, k, len(recs))
        d.close()


def test_device_special_region_module_forced_on_small_inputs_equals_oracle(api, oracle, monkeypatch):
    """The device module at any size (DEBWT_SPECIAL_DEVICE_MIN=0) inside full builds, single and multi-range, against
    the oracle: adversarial collections (duplicates, prefix-duplicates, shared ends, homopolymer ends) and read sets."""
    from debwt_amd import synth
    monkeypatch.setenv("DEBWT_SPECIAL_DEVICE_MIN", "0")
    rng = np.random.default_rng(777)
    for c in range(60):
        recs = _adversarial(rng) if c % 2 else synth.read_set(int(rng.integers(2, 300)), 50, 200, 10_000,
                                                              seed=int(rng.integers(1, 1 << 30)), snap=4, dup_every=5)
        k = int(rng.choice([12, 16, 21, 32]))
        ow, oh, od, ost = oracle.build_bwt(oracle.sym_from_codes(recs), k)
        d = api.DeBWT(k=k)
        if c % 3 == 0:
            d.set_range_cap(4096)
        d.load_records(recs)
        d.build()
        w, h, dr = d.fetch()
        st = d.stats()
        assert st["special_path"] == 2 and st["special_branch_num"] == ost["special_branch_num"]
        assert np.array_equal(w, ow) and np.array_equal(h, oh) and dr == od, (c, k)
        d.close()


def test_sorted_keys_are_refused_once_their_buffer_is_reused(api):
    """DEBWT_ARR_SORTED_KEYS names a buffer the SP stage reuses as scratch: after that stage the fetch is an error,
    not stale bytes."""
    from debwt_amd import synth
    d = api.DeBWT(k=32)
    d.load_records(synth.pan_genome(50_000, 2))
    d.kmer_sort_rle()
    sk = d.fetch_array(api.ARR_SORTED_KEYS)
    d.classify()
    assert np.array_equal(d.fetch_array(api.ARR_SORTED_KEYS), sk)
    d.sp_generate()
    with pytest.raises(RuntimeError):
        d.fetch_array(api.ARR_SORTED_KEYS)
    d.build()
    with pytest.raises(RuntimeError):
        d.fetch_array(api.ARR_SORTED_KEYS)
    assert len(d.fetch_array(api.ARR_DISTINCT_KEYS)) == d.stats()["distinct_keys"]
    d.close()


def _adversarial(rng):
    nrec = int(rng.integers(1, 7))
    base = rng.integers(0, 4, size=int(rng.integers(60, 400))).astype(np.uint8)
    recs = []
    for _ in range(nrec):
        L = int(rng.integers(33, 900))
        x = rng.integers(0, 4, size=L).astype(np.uint8)
        if rng.random() < 0.7:
            seg = base[:min(len(base), L)]
            p = int(rng.integers(0, L - len(seg) + 1))
            x[p:p + len(seg)] = seg
        if rng.random() < 0.3:
            x[-min(L, 40):] = base[:min(L, 40)]
        if rng.random() < 0.2:
            x[:min(L, 50)] = rng.integers(0, 4)
        recs.append(x)
    if rng.random() < 0.4:
        recs.append(recs[0].copy())
    if rng.random() < 0.3:
        recs.append(recs[-1][:max(33, len(recs[-1]) // 2)].copy())
    return recs


@pytest.mark.parametrize("seed", range(20))
def test_hip_stagewise_equals_oracle_on_adversarial_inputs(api, oracle, seed):
    rng = np.random.default_rng(7000 + seed)
    recs = _adversarial(rng)
    sym = oracle.sym_from_codes(recs)
    k = int(rng.choice([12, 15, 20, 31, 32]))
    ow, oh, od, ost, osp, ored = oracle.build_bwt(sym, k, want_intermediates=True)
    d = api.DeBWT(k=k)
    d.load_records(recs)
    d.kmer_sort_rle()
    keys = d.fetch_array(api.ARR_SORTED_KEYS)
    assert (keys[1:] >= keys[:-1]).all()
    d.classify()
    assert np.array_equal(d.fetch_array(api.ARR_RED), ored)
    d.sp_generate()
    assert np.array_equal(d.fetch_array(api.ARR_SP_SYMBOLS), osp)
    d.blue_sort()
    d.bwt_assemble()
    words, hrows, drow = d.fetch()
    assert np.array_equal(words, ow) and np.array_equal(hrows, oh) and drow == od
    st = d.stats()
    for a, b in (("red_capacity", "red_capacity"), ("blue_capacity", "blue_capacity"),
                 ("blue_bound_num", "blue_bound_num"), ("sp_len", "sp_len"),
                 ("special_branch_num", "special_branch_num")):
        assert st[a] == ost[b], a
    rows = d.fetch_array(api.ARR_ROW_SYMBOLS)
    assert np.array_equal(rows, oracle.unpack_bwt(ow, len(sym), oh, od))
    km, ct = d.kmer_count_sorted()
    okm, oct_ = oracle.kmer_count(sym, k)
    assert np.array_equal(km, okm) and np.array_equal(ct, oct_)
    d.close()


@pytest.mark.parametrize("name,k", [("pan_16M_4", 32), ("uniform_16M", 32), ("pan_16M_4", 21)])
def test_hip_equals_oracle_midsize(api, oracle, name, k):
    from debwt_amd import synth
    recs = synth.make_workload(name)
    sym = oracle.sym_from_codes(recs)
    ow, oh, od, ost = oracle.build_bwt(sym, k)
    d, (words, hrows, drow), st = _run(api, recs, k)
    assert np.array_equal(words, ow) and np.array_equal(hrows, oh) and drow == od
    assert st["red_capacity"] == ost["red_capacity"] and st["sp_len"] == ost["sp_len"]
    assert st["blue_capacity"] == ost["blue_capacity"]
    d.close()


def test_hip_block_size_classes(api, oracle):
    """Blocks of 65..2048 rows: the 2/4/8-entries-per-lane wave sorts and the LDS workgroup sort."""
    rng = np.random.default_rng(11)
    parts = []
    for copies in (70, 100, 130, 200, 300, 500, 600, 1100, 1900):
        unit = rng.integers(0, 4, size=36).astype(np.uint8)
        for i in range(copies):
            parts.append(unit)
            parts.append(rng.integers(0, 4, size=int(rng.integers(2, 7))).astype(np.uint8))
    recs = [np.concatenate(parts), rng.integers(0, 4, size=300).astype(np.uint8)]
    sym = oracle.sym_from_codes(recs)
    ow, oh, od, ost = oracle.build_bwt(sym, 32)
    d, (words, hrows, drow), st = _run(api, recs, 32)
    assert np.array_equal(words, ow) and np.array_equal(hrows, oh) and drow == od
    blue = d.fetch_array(api.ARR_BLUE)
    bound = d.fetch_array(api.ARR_BLUE_BOUND)
    sizes = np.diff(np.concatenate([[-1], bound.astype(np.int64)]))
    assert (sizes > 64).sum() >= 9 and (sizes > 512).sum() >= 3 and len(blue) == st["blue_capacity"]
    d.close()


@pytest.mark.parametrize("tune", [1048576, 1048576 + 128, 524288])
def test_blue_blocks_by_classes(api, oracle, tune):
    """Blocks of 257..2048 rows sorted by classes around sampled splitters (k_blue_classify; tune bit 20 takes that kernel
    for any number of such blocks, bit 19 never): copies that tie beyond 42 SP symbols with several BWT symbols (queued one
    pair of windows deeper), copies that differ early, a block of one symbol, and -- with the queue switched off (bit 7) --
    every block left to the kernel behind it."""
    rng = np.random.default_rng(31)
    parts = []
    for copies, tail in ((700, 3), (900, 40), (600, 400), (1000, 0), (550, 90), (300, 30), (450, 200), (1500, 60), (1900, 5)):
        unit = rng.integers(0, 4, size=60).astype(np.uint8)
        variants = [rng.integers(0, 4, size=max(tail, 1)).astype(np.uint8) for _ in range(7)]
        for i in range(copies):
            parts.append(rng.integers(0, 4, size=1).astype(np.uint8) if tail else np.array([i % 2], dtype=np.uint8))
            parts.append(unit)
            if tail:
                v = variants[i % 7].copy()
                if tail > 50 and i % 3 == 0:
                    v[tail // 2] = (v[tail // 2] + 1) & 3
                parts.append(v)
            parts.append(rng.integers(0, 4, size=int(rng.integers(3, 9))).astype(np.uint8))
    recs = [np.concatenate(parts), rng.integers(0, 4, size=500).astype(np.uint8)]
    ow, oh, od, ost = oracle.build_bwt(oracle.sym_from_codes(recs), 32)
    d = api.DeBWT(k=32, tune=tune)
    d.load_records(recs)
    for _ in range(2):
        d.build()
        words, hrows, drow = d.fetch()
        assert np.array_equal(words, ow) and np.array_equal(hrows, oh) and drow == od, tune
    d.close()


@pytest.mark.parametrize("tune", [0, 1024, 128])
def test_hip_large_block_path(api, oracle, tune):
    """A node with more occurrences than one workgroup's LDS holds: split in HBM into ranges for the LDS kernels (0),
    the bitonic network in HBM alone (1024), and the network as the fall-back when nothing can be queued (128)."""
    rng = np.random.default_rng(5)
    unit = rng.integers(0, 4, size=40).astype(np.uint8)
    parts = []
    for i in range(5000):
        parts.append(unit)
        parts.append(rng.integers(0, 4, size=int(rng.integers(3, 9))).astype(np.uint8))
    recs = [np.concatenate(parts), rng.integers(0, 4, size=500).astype(np.uint8)]
    sym = oracle.sym_from_codes(recs)
    ow, oh, od, _ = oracle.build_bwt(sym, 32)
    d = api.DeBWT(k=32, tune=tune)
    d.load_records(recs)
    d.build()
    words, hrows, drow = d.fetch()
    st = d.stats()
    assert st["blue_large_blocks"] >= 1 and st["blue_max_block"] > 2048
    assert np.array_equal(words, ow) and np.array_equal(hrows, oh) and drow == od
    d.close()


def test_hip_large_blocks_with_long_ties(api, oracle):
    """Large blocks whose rows stay tied beyond the first two SP windows (long exact repeats of a branching unit): the
    split queues what it separates and hands ranges it cannot separate to the network."""
    rng = np.random.default_rng(9)
    core = rng.integers(0, 4, size=36).astype(np.uint8)
    # a long segment built from the core with small variable spacers: many branching nodes, repeated exactly
    seg_parts = []
    for _ in range(60):
        seg_parts.append(core)
        seg_parts.append(rng.integers(0, 4, size=int(rng.integers(2, 6))).astype(np.uint8))
    seg = np.concatenate(seg_parts)
    parts = []
    for i in range(70):                                       # 70 exact copies of the segment: 4200 copies of the core
        parts.append(seg)
        parts.append(rng.integers(0, 4, size=int(rng.integers(40, 90))).astype(np.uint8))
    recs = [np.concatenate(parts), rng.integers(0, 4, size=300).astype(np.uint8)]
    ow, oh, od, _ = oracle.build_bwt(oracle.sym_from_codes(recs), 32)
    for tune in (0, 1024):
        d = api.DeBWT(k=32, tune=tune)
        d.load_records(recs)
        d.build()
        words, hrows, drow = d.fetch()
        st = d.stats()
        assert st["blue_large_blocks"] >= 1
        assert np.array_equal(words, ow) and np.array_equal(hrows, oh) and drow == od, tune
        d.close()


@pytest.mark.parametrize("seed,runs,maxrun", [(5, 40, 9000), (6, 12, 30000)])
def test_hip_periodic_stretches_pivot_rounds(api, oracle, seed, runs, maxrun):
    """Runs of one symbol and tandem repeats of 2..6-symbol units, thousands of symbols long: the rows of their nodes
    tie for as long as the stretch lasts, in blocks far above the LDS capacity.  The pivot rounds of the large-block
    split (default), window rounds only (reserved bit 13) and the bitonic network alone (bit 10) must all give the
    oracle's BWT."""
    rng = np.random.default_rng(seed)
    parts = []
    for _ in range(runs):
        parts.append(np.full(int(rng.integers(200, maxrun)), int(rng.integers(0, 4)), dtype=np.uint8))
        parts.append(np.tile(rng.integers(0, 4, size=int(rng.integers(2, 7))).astype(np.uint8), int(rng.integers(50, maxrun // 10))))
        parts.append(rng.integers(0, 4, size=int(rng.integers(100, 3000))).astype(np.uint8))
    recs = [np.concatenate(parts[:len(parts) // 2]), np.concatenate(parts[len(parts) // 2:])]
    ow, oh, od, _ = oracle.build_bwt(oracle.sym_from_codes(recs), 32)
    for tune in (0, 8192) + ((1024,) if maxrun < 10000 else ()):       # the network's comparator walks the whole tie
        d = api.DeBWT(k=32, tune=tune)
        d.load_records(recs)
        d.build()
        words, hrows, drow = d.fetch()
        st = d.stats()
        assert st["blue_large_blocks"] >= 1 and st["blue_max_block"] > 2048
        assert np.array_equal(words, ow) and np.array_equal(hrows, oh) and drow == od, tune
        d.close()


@pytest.mark.parametrize("k", [32, 20])
def test_hip_long_unfit_stretch_of_the_key_sort(api, oracle, k):
    """A run of one symbol with 10 % substitutions: nearly half of its k-mers share their first 8 symbols -- one bucket of
    the two top radix passes with 10^5 keys that differ further down.  The bucket leaves through the all-HBM path and its
    run-length encoding is shared by several workgroups along the raster of the wave tiles (rs_unfit_rle_kernel<*, 1>)."""
    rng = np.random.default_rng(17)
    run = np.zeros(260_000, dtype=np.uint8)
    hit = rng.random(len(run)) < 0.10
    run[hit] = rng.integers(1, 4, size=int(hit.sum()))
    recs = [np.concatenate([rng.integers(0, 4, size=5000).astype(np.uint8), run, rng.integers(0, 4, size=5000).astype(np.uint8)]),
            rng.integers(0, 4, size=700).astype(np.uint8)]
    ow, oh, od, _ = oracle.build_bwt(oracle.sym_from_codes(recs), k)
    d = api.DeBWT(k=k)
    d.load_records(recs)
    for _ in range(2):
        d.build()
        words, hrows, drow = d.fetch()
        assert np.array_equal(words, ow) and np.array_equal(hrows, oh) and drow == od
    d.close()


def _node_keys(recs, k):
    """(node << 2 | pred) of every main position, by definition (numpy)."""
    K = k - 1
    out = []
    for x in recs:
        x = np.asarray(x, dtype=np.uint64)
        m = len(x) - K + 1                                   # windows that end at or before the separator
        node = np.zeros(m, dtype=np.uint64)
        for j in range(K):
            node = (node << np.uint64(2)) | x[j:j + m]
        pred = np.concatenate([np.array([3], dtype=np.uint64), x[:m - 1]])
        out.append((node << np.uint64(2)) | pred)
    return np.concatenate(out)


def _unfit_case(name):
    rng = np.random.default_rng(4242)
    if name == "families":
        # repeat families with 2 % divergence: a few hundred keys per 8-mer bucket (two top passes at this size), most
        # of them copies of a few dozen distinct keys -- what the classifying finish is made for
        recs = []
        for r in range(4):
            parts = []
            for f in range(3):
                cons = rng.integers(0, 4, size=600).astype(np.uint8)
                for _ in range(400):
                    cp = cons.copy()
                    hit = rng.random(600) < 0.02
                    cp[hit] = (cp[hit] + rng.integers(1, 4, size=int(hit.sum()))) & 3
                    parts.append(cp)
                    parts.append(rng.integers(0, 4, size=int(rng.integers(5, 200))).astype(np.uint8))
            recs.append(np.concatenate(parts))
        return recs
    if name == "runs":
        # runs of one symbol with 10 % substitutions: buckets of 1000 to 5000 keys that are nearly all distinct (gap classes
        # between the splitters do the work; the longest go to the all-HBM passes)
        parts = []
        for ln in (1400, 2600, 3900, 14000, 1800, 3100):
            run = np.full(ln, int(rng.integers(0, 4)), dtype=np.uint8)
            hit = rng.random(ln) < 0.10
            run[hit] = (run[hit] + rng.integers(1, 4, size=int(hit.sum()))) & 3
            parts += [rng.integers(0, 4, size=3000).astype(np.uint8), run]
        return [np.concatenate(parts), rng.integers(0, 4, size=900).astype(np.uint8)]
    # exact copies: every key of a stretch equals a splitter, or the stretch is in order already
    seg = rng.integers(0, 4, size=500).astype(np.uint8)
    return [np.concatenate([seg] * 700 + [rng.integers(0, 4, size=4000).astype(np.uint8)]), np.concatenate([seg[100:]] * 50)]


@pytest.mark.parametrize("tune", [0, 32768])
@pytest.mark.parametrize("k", [32, 21])
@pytest.mark.parametrize("name", ["families", "runs", "copies"])
def test_unfit_stretches_of_the_key_sort_by_classes(api, name, k, tune):
    """Stretches above a wave tile are sorted by classes around sampled splitters (rs_local_unfit_kernel); tune bit 15
    sends every stretch with keys between the splitters to the 4096-key network instead.  Sorted keys, distinct keys
    and row symbols against the definition."""
    recs = _unfit_case(name)
    want = np.sort(_node_keys(recs, k))
    d = api.DeBWT(k=k, tune=tune)
    d.load_records(recs)
    for _ in range(2):
        d.kmer_sort_rle()
        got = d.fetch_array(api.ARR_SORTED_KEYS)
        assert np.array_equal(got, want)
        assert np.array_equal(d.fetch_array(api.ARR_DISTINCT_KEYS), np.unique(want))
        st = d.stats()
        assert st["sort_unfit_stretches"] > 0, st
        if tune and name != "copies":
            assert st["sort_unfit_network"] > 0, st
        if name == "runs":
            assert st["sort_over_stretches"] > 0, st
    d.close()


def test_hip_large_tie_ranges_go_deeper(api, oracle):
    """3000 exact copies of a segment full of branching nodes: the rows of a node early in the segment tie for more than
    the 42 SP symbols a split looks at, in groups above the LDS capacity -- those ranges are split again one pair of
    windows deeper (and must give what the network alone gives)."""
    rng = np.random.default_rng(21)
    core = rng.integers(0, 4, size=36).astype(np.uint8)
    seg_parts = []
    for _ in range(24):
        seg_parts.append(core)
        seg_parts.append(rng.integers(0, 4, size=int(rng.integers(2, 6))).astype(np.uint8))
    seg = np.concatenate(seg_parts)
    parts = []
    for i in range(3000):
        parts.append(seg)
        parts.append(rng.integers(0, 4, size=int(rng.integers(40, 90))).astype(np.uint8))
    recs = [np.concatenate(parts), rng.integers(0, 4, size=300).astype(np.uint8)]
    ow, oh, od, _ = oracle.build_bwt(oracle.sym_from_codes(recs), 32)
    for tune in (0, 1024):
        d = api.DeBWT(k=32, tune=tune)
        d.load_records(recs)
        d.build()
        words, hrows, drow = d.fetch()
        st = d.stats()
        assert st["blue_large_blocks"] >= 1 and st["blue_max_block"] >= 3000 * 20
        assert np.array_equal(words, ow) and np.array_equal(hrows, oh) and drow == od, tune
        d.close()


def test_hip_properties_at_bench_size(api):
    """BASELINE configs[1]-sized input: inverse BWT reproduces the text, k-invariance, symbol census."""
    from debwt_amd import synth
    recs = synth.make_workload("pan_100M_4")
    d, (words, hrows, drow), st = _run(api, recs, 32)
    n = st["n"]
    rc, inv = api.verify_inverse(words, n, hrows, drow)
    assert rc == 0
    o = 0
    for i, r in enumerate(recs):
        assert np.array_equal(inv[o:o + len(r)], r)
        assert inv[o + len(r)] == (5 if i + 1 == len(recs) else 4)
        o += len(r) + 1
    assert (np.diff(hrows.astype(np.int64)) > 0).all()
    d.close()
    d2, (w2, h2, dr2), _ = _run(api, recs, 24)
    assert np.array_equal(words, w2) and np.array_equal(hrows, h2) and drow == dr2
    d2.close()


@pytest.mark.parametrize("count,bits", [(1, 64), (2, 64), (4095, 64), (4097, 40), (1 << 20, 64), (3_000_001, 62),
                                        (5_000_000, 24)])
def test_radix_sort_primitive(api, count, bits):
    import torch
    g = torch.Generator(device="cpu").manual_seed(count)
    hi = torch.randint(0, 2 ** 31, (count,), generator=g, dtype=torch.int64)
    lo = torch.randint(0, 2 ** 31, (count,), generator=g, dtype=torch.int64)
    keys = (hi << 33) ^ (lo << 2) ^ (hi >> 7)
    if bits < 64:
        keys &= (1 << bits) - 1
    ref = np.sort(keys.numpy().view(np.uint64))
    dk = keys.cuda()
    tmp = torch.empty_like(dk)
    d = api.DeBWT(k=32)
    d.radix_sort_device(dk.data_ptr(), tmp.data_ptr(), count, bits)
    assert np.array_equal(dk.cpu().numpy().view(np.uint64), ref)
    d.close()


@pytest.mark.parametrize("case", ["all_equal", "few_values", "medium_buckets", "huge_and_small", "sorted", "reverse"])
@pytest.mark.parametrize("algo", [1, 3])
def test_radix_sort_skewed(api, case, algo):
    """Bucket shapes the hybrid sort treats differently: long runs of equal keys (HBM fallback), buckets of
    hundreds of keys (LSD inside LDS), tiny buckets (rank by counting)."""
    import torch
    rng = np.random.default_rng(99)
    n = 3_000_000
    if case == "all_equal":
        keys = np.full(n, 0x0123456789ABCDEF, dtype=np.uint64)
    elif case == "few_values":
        keys = rng.choice(rng.integers(0, 2 ** 63, size=7, dtype=np.uint64), size=n)
    elif case == "medium_buckets":
        keys = (rng.integers(0, 3000, size=n, dtype=np.uint64) << np.uint64(44)) | rng.integers(0, 2 ** 30, size=n, dtype=np.uint64)
    elif case == "huge_and_small":
        keys = rng.integers(0, 2 ** 63, size=n, dtype=np.uint64)
        keys[:700_000] = (np.uint64(0xABCDEF) << np.uint64(40)) | rng.integers(0, 2 ** 20, size=700_000, dtype=np.uint64)
        keys[700_000:760_000] = np.uint64(42)
        rng.shuffle(keys)
    elif case == "sorted":
        keys = np.sort(rng.integers(0, 2 ** 63, size=n, dtype=np.uint64))
    else:
        keys = np.sort(rng.integers(0, 2 ** 63, size=n, dtype=np.uint64))[::-1].copy()
    ref = np.sort(keys)
    dk = torch.from_numpy(keys.view(np.int64)).cuda()
    tmp = torch.empty_like(dk)
    d = api.DeBWT(k=32, sort_algo=algo)
    d.radix_sort_device(dk.data_ptr(), tmp.data_ptr(), n, 64)
    assert np.array_equal(dk.cpu().numpy().view(np.uint64), ref)
    d.close()


def test_sort_algorithms_agree_end_to_end(api):
    from debwt_amd import synth
    recs = synth.pan_genome(1_500_000, 3)
    outs = []
    for algo in (1, 3):
        d, out, st = _run(api, recs, 32, algo=algo)
        outs.append(out)
        d.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]


def test_stage_order_and_errors(api):
    from debwt_amd import synth
    with pytest.raises(api.DebwtError):
        api.DeBWT(k=33)                                    # src/main.c:45-46
    d = api.DeBWT(k=32)
    with pytest.raises(api.DebwtError):
        d.build()                                          # nothing loaded
    d.load_records(synth.pan_genome(5000, 2))
    with pytest.raises(api.DebwtError):
        d.classify()                                       # out of order
    with pytest.raises(ValueError):
        api.pack_records([np.zeros(32, np.uint8)])         # src/collect#$.c:41-45
    d.build()
    a = d.fetch()
    d.build()                                              # a context is reusable
    b = d.fetch()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    d.load_ascii(["ACGT" * 30 + "TTGACCA" * 9, "acgtacgttgca" * 11])
    d.build()
    d.close()


# ---- key ranges: texts whose node instances exceed the range cap are sorted and classified range by range ------

@pytest.mark.parametrize("entry", [e for e in MANIFEST if e["k"] in (12, 32)], ids=golden_id)
def test_multi_range_build_matches_reference_golden(api, entry):
    """The same golden vectors with the key space cut into many prefix ranges (cap = 4096 instances)."""
    recs = golden_records(entry)
    d = api.DeBWT(k=entry["k"])
    d.set_range_cap(4096 if entry["n"] < 2_000_000 else entry["n"] // 40)     # the Mbp-sized goldens: ~40 ranges
    d.load_records(recs)
    d.build()
    words, hrows, drow = d.fetch()
    st = d.stats()
    sha = entry["sha256"]
    assert _sha(words) == sha["bwt"] and _sha(hrows) == sha["hash"]
    assert _sha(np.array([drow], dtype=np.uint64)) == sha["dollar"]
    c = entry["counters"]
    assert st["case3num"] == c["case3num"] and st["blue_bound_num"] == c["blueBoundNum"]
    assert st["red_capacity"] == c["redCapacity"] and st["blue_capacity"] == c["blueCapacity"]
    d.close()


def _build_to_host(api, d):
    """debwt_build_to_host into page-locked arrays; returns (words, hash rows, dollar row) as fetch() does."""
    from debwt_amd import synth_native as SN
    w = SN.PinnedArray((d.n + 31) // 32); h = SN.PinnedArray(max(d.nrec - 1, 1)); dr = SN.PinnedArray(1)
    w.a[:] = 0xDEADBEEFDEADBEEF
    try:
        d.build_into(w.a, h.a, dr.a)
        return w.a.copy(), h.a[:d.nrec - 1].copy(), int(dr.a[0])
    finally:
        w.free(); h.free(); dr.free()


@pytest.mark.parametrize("entry", [e for e in MANIFEST if e["k"] in (12, 32)], ids=golden_id)
def test_build_to_host_streams_the_reference_rows(api, entry):
    """debwt_build_to_host on the golden vectors cut into many key ranges: the rows of a range leave for the host while
    the blocks of the next range are sorted -- the reference's bytes; then once more in one range (plain build + fetch
    inside) and with the overlap switched off (tune bit 16)."""
    recs = golden_records(entry)
    sha = entry["sha256"]
    for cap, tune in ((4096 if entry["n"] < 2_000_000 else entry["n"] // 40, 0), (None, 0), (4096 if entry["n"] < 2_000_000 else entry["n"] // 7, 65536)):
        d = api.DeBWT(k=entry["k"], tune=tune)
        if cap:
            d.set_range_cap(cap)
        d.load_records(recs)
        for _ in range(2):
            words, hrows, drow = _build_to_host(api, d)
            assert _sha(words) == sha["bwt"] and _sha(hrows) == sha["hash"], (cap, tune)
            assert _sha(np.array([drow], dtype=np.uint64)) == sha["dollar"]
        w2, h2, d2 = d.fetch()                                   # the device copy is complete as well
        assert np.array_equal(w2, words) and np.array_equal(h2, hrows) and d2 == drow
        d.close()


def test_build_to_host_with_large_blocks_and_deep_ties(api, oracle):
    """Blocks above the LDS capacity, queued tie groups and special suffixes spread over the key ranges of a streamed build."""
    from debwt_amd import synth
    rng = np.random.default_rng(78)
    unit = rng.integers(0, 4, size=50).astype(np.uint8)
    recs = [np.concatenate([unit] * 120 + [rng.integers(0, 4, size=2000).astype(np.uint8)]) for _ in range(25)]
    recs += synth.pan_genome(150_000, 6)
    ow, oh, od, _ = oracle.build_bwt(oracle.sym_from_codes(recs), 32)
    for cap in (50_000, 300_000):
        d = api.DeBWT(k=32)
        d.set_range_cap(cap)
        d.load_records(recs)
        words, hrows, drow = _build_to_host(api, d)
        assert d.stats()["blue_large_blocks"] >= 1
        assert np.array_equal(words, ow) and np.array_equal(hrows, oh) and drow == od, cap
        d.close()


@pytest.mark.parametrize("entry", [e for e in MANIFEST if e["k"] in (12, 32)], ids=golden_id)
def test_blue_entries_routed_by_key_range(api, entry):
    """The path of collections with 2^32 blue rows and more (tune bit 18 takes it at any size): the routed entries of a text
    slice are bucketed by key range, every range's entries sorted by block id on their own -- the reference's bytes, and
    the blue table itself equal to the one array sort's."""
    recs = golden_records(entry)
    sha = entry["sha256"]
    cap = 4096 if entry["n"] < 2_000_000 else entry["n"] // 9
    blues = []
    for tune in (262144, 0):
        d = api.DeBWT(k=entry["k"], tune=tune)
        d.set_range_cap(cap)
        d.load_records(recs)
        d.kmer_sort_rle(); d.classify(); d.sp_generate()
        blues.append(d.fetch_array(api.ARR_BLUE).copy())
        d.blue_sort(); d.bwt_assemble()
        words, hrows, drow = d.fetch()
        assert _sha(words) == sha["bwt"] and _sha(hrows) == sha["hash"], tune
        d.close()
    # (the order of the entries inside a block is the order of arrival, which differs between the two ways: compare as sets per table)
    assert len(blues[0]) == len(blues[1]) and np.array_equal(np.sort(blues[0]), np.sort(blues[1]))


@pytest.mark.parametrize("entry", [e for e in MANIFEST if e["k"] == 32 and e["n"] > 300_000], ids=golden_id)
def test_sp_stage_in_slices(api, entry):
    """The SP stage slice by slice (what a text above 2^31 positions takes; tune bit 22 cuts slices of 2^17 positions so that
    the goldens have several): the reference's bytes, and SP symbols and blue table equal to the unsliced build's."""
    recs = golden_records(entry)
    sha = entry["sha256"]
    sp, blue = [], []
    for tune in (4194304, 0):
        d = api.DeBWT(k=32, tune=tune)
        d.load_records(recs)
        d.kmer_sort_rle(); d.classify(); d.sp_generate()
        sp.append(d.fetch_array(api.ARR_SP_SYMBOLS).copy()); blue.append(d.fetch_array(api.ARR_BLUE).copy())
        d.blue_sort(); d.bwt_assemble()
        words, hrows, drow = d.fetch()
        assert _sha(words) == sha["bwt"] and _sha(hrows) == sha["hash"], tune
        d.close()
    assert np.array_equal(sp[0], sp[1])
    assert len(blue[0]) == len(blue[1]) and np.array_equal(np.sort(blue[0]), np.sort(blue[1]))


@pytest.mark.parametrize("cap", [1 << 20, 3_000_000, 1 << 23])
def test_multi_range_equals_single_range_midsize(api, cap):
    from debwt_amd import synth
    recs = synth.make_workload("pan_16M_4")
    _, (w1, h1, d1), st1 = _run(api, recs, 32)
    # blue fill: 0 = entries routed and sorted by block id; 32 = cursor atomics; 48 = cursor atomics with
    # block-relative cursors + 64-bit block starts (what a context with >= 2^32 blue rows uses)
    d = api.DeBWT(k=32, tune={1 << 20: 0, 3_000_000: 48, 1 << 23: 32}[cap])
    d.set_range_cap(cap)
    d.load_records(recs)
    d.build()
    w2, h2, d2 = d.fetch()
    st2 = d.stats()
    assert np.array_equal(w1, w2) and np.array_equal(h1, h2) and d1 == d2
    for key in ("red_capacity", "blue_capacity", "blue_bound_num", "sp_len", "distinct_keys", "blue_large_blocks"):
        assert st1[key] == st2[key], key
    # stage-wise arrays that do not depend on the ranges
    for which in (api.ARR_RED, api.ARR_SP_SYMBOLS, api.ARR_BLUE_BOUND, api.ARR_CASE3_BOUND):
        assert np.array_equal(_run(api, recs, 32)[0].fetch_array(which), d.fetch_array(which)), which
    with pytest.raises(api.DebwtError):
        d.fetch_array(api.ARR_SORTED_KEYS)           # not kept by a multi-range build
    d.close()


def test_multi_range_with_large_blocks_and_many_records(api, oracle):
    """Blocks above the LDS capacity and special suffixes spread over several ranges."""
    rng = np.random.default_rng(77)
    units = [rng.integers(0, 4, size=40).astype(np.uint8) for _ in range(3)]
    recs = []
    for i in range(30):
        parts = []
        for j in range(300):                                 # 9000 copies of each unit over the collection
            parts.append(units[j % 3])
            parts.append(rng.integers(0, 4, size=int(rng.integers(3, 9))).astype(np.uint8))
        recs.append(np.concatenate(parts))
    d = api.DeBWT(k=20)
    d.set_range_cap(20000)
    d.load_records(recs)
    d.build()
    words, hrows, drow = d.fetch()
    st = d.stats()
    ow, oh, od, ost = oracle.build_bwt(oracle.sym_from_codes(recs), 20)
    assert st["blue_large_blocks"] > 0
    assert np.array_equal(words, ow) and np.array_equal(hrows, oh) and drow == od
    d.close()


def test_context_reuse_across_range_caps_and_repeated_stage_calls(api, oracle):
    """One context through what a host may do with it: a build in many key ranges, then the cap lifted and a single-range
    build (the range fields of the last multi-range build must not leak into it), classification called twice, a load
    with rejected arguments in between."""
    from debwt_amd import synth
    recs = synth.pan_genome(60_000, 3)
    ow, oh, od, _ = oracle.build_bwt(oracle.sym_from_codes(recs), 32)
    d = api.DeBWT(k=32)
    d.load_records(recs)
    d.set_range_cap(4096)
    d.build()
    a = d.fetch()
    d.set_range_cap(1 << 31)
    d.build()
    b = d.fetch()
    for w, h, dr in (a, b):
        assert np.array_equal(w, ow) and np.array_equal(h, oh) and dr == od
    d.kmer_sort_rle()
    d.classify()
    q1 = d.stats()["blue_bound_num"]
    d.classify()                                               # again: the block tables start over, nothing is appended twice
    assert d.stats()["blue_bound_num"] == q1
    d.sp_generate(); d.blue_sort(); d.bwt_assemble()
    w, h, dr = d.fetch()
    assert np.array_equal(w, ow) and np.array_equal(h, oh) and dr == od
    with pytest.raises(api.DebwtError):
        d.load_packed(np.zeros(8, dtype=np.uint64), 100, np.array([50], dtype=np.uint64))   # last separator is not n-1
    d.build()                                                  # rejected arguments leave the loaded text untouched
    w, h, dr = d.fetch()
    assert np.array_equal(w, ow) and np.array_equal(h, oh) and dr == od
    d.close()


def test_shard_world_limit(api):
    from debwt_amd import _lib, synth
    d = api.DeBWT(k=32)
    d.load_records(synth.pan_genome(5000, 2))
    L = _lib.lib()
    assert L.debwt_shard_begin(d._h, 0, 256) == -1             # owner tables hold bytes, 0xFF = none
    assert L.debwt_shard_begin(d._h, 3, 3) == -1
    assert L.debwt_shard_begin(d._h, 0, 255) == 0
    d.close()


@pytest.mark.parametrize("entry", [e for e in MANIFEST if e["k"] >= 24], ids=golden_id)
def test_minimizer_prefilter_matches_reference_golden(api, entry):
    """The SP stage's prefilter indexed by minimizers (the form large texts use; tune bit 12 forces it at any size), with
    the filter at its default size and cut to 1/4 (saturated words send more positions to the node table); the node
    table hashed by node (the default) and addressed by (minimizer, offset) pairs (bit 14)."""
    for tune in (4096, 4096 + 6, 4096 + 16384):
        d = api.DeBWT(k=entry["k"], tune=tune)
        d.load_records(golden_records(entry))
        d.build()
        words, hrows, drow = d.fetch()
        st = d.stats()
        assert _sha(words) == entry["sha256"]["bwt"] and _sha(hrows) == entry["sha256"]["hash"]
        assert st["sp_len"] + 32 == entry["counters"]["spCodeLen"] and st["blue_capacity"] == entry["counters"]["blueCapacity"]
        assert _sha(d.fetch_array(api.ARR_SP_SYMBOLS)) == entry["sha256"]["spSymbols"]
        d.close()


@pytest.mark.parametrize("tune", [0, 128])
def test_pan_genome_deep_tie_groups(api, oracle, tune):
    """Ten near-identical genomes with repeat families: blocks of hundreds of rows whose tie groups (one locus in
    several genomes) stay tied for hundreds of SP symbols -- with (0) and without (128) the hand-off of those groups to
    the wave-per-block kernel."""
    from debwt_amd import synth
    recs = synth.pan_genome(400_000, 10)
    d = api.DeBWT(k=32, tune=tune)
    d.load_records(recs)
    d.build()
    words, hrows, drow = d.fetch()
    bound = d.fetch_array(api.ARR_BLUE_BOUND).astype(np.int64)
    sizes = np.diff(np.concatenate([[-1], bound]))
    assert (sizes > 128).sum() > 100
    ow, oh, od, _ = oracle.build_bwt(oracle.sym_from_codes(recs), 32)
    assert np.array_equal(words, ow) and np.array_equal(hrows, oh) and drow == od
    d.close()


@pytest.mark.parametrize("kind", ["pan", "lowcomplexity"])
def test_rle_paths_agree(api, kind):
    """The bucket finish that counts distinct keys per tile (default) against the separate count + emit passes over the
    sorted keys (tune 256): same sorted keys, distinct keys and BWT -- on repeat families (stretches above a wave tile)
    and on low-complexity runs (buckets that go through the 4096-key tiles and the HBM path)."""
    from debwt_amd import synth
    if kind == "pan":
        recs = synth.pan_genome(600_000, 6)
    else:
        rng = np.random.default_rng(5)
        parts = []
        for _ in range(300):
            parts.append(np.full(int(rng.integers(200, 30000)), int(rng.integers(0, 4)), dtype=np.uint8))
            parts.append(np.tile(rng.integers(0, 4, size=int(rng.integers(2, 7))).astype(np.uint8), int(rng.integers(50, 3000))))
            parts.append(rng.integers(0, 4, size=int(rng.integers(100, 5000))).astype(np.uint8))
        recs = [np.concatenate(parts)]
    got = []
    for tune in (0, 256):
        d = api.DeBWT(k=32, tune=tune)
        d.load_records(recs)
        d.kmer_sort_rle()
        sk = d.fetch_array(api.ARR_SORTED_KEYS)
        dk = d.fetch_array(api.ARR_DISTINCT_KEYS)
        assert bool((sk[1:] >= sk[:-1]).all())
        assert np.array_equal(np.unique(sk), dk)
        d.build()
        got.append((sk, dk) + tuple(d.fetch()))
        d.close()
    a, b = got
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and a[4] == b[4]


@pytest.mark.parametrize("cap", [0, 1 << 20])
def test_tiles_that_stage_their_distinct_keys_need_not_write_their_sorted_keys_back(api, cap):
    """Inside debwt_build a tile of the bucket finish that staged its distinct keys keeps its sorted keys to itself
    (RleSink::drop_sorted): nobody reads them -- the sorted keys can only be fetched after the sort stage of a one-range
    build driven stage by stage, where they are written as before.  Same BWT with the write-back restored (tune bit 23) and
    with the separate count + emit passes (tune 256), in one key range and in several, on repeat families (tiles and unfit
    stretches that stage) and low-complexity runs (oversize stretches, re-sorted in place)."""
    from debwt_amd import synth
    rng = np.random.default_rng(23)
    parts = []
    for _ in range(120):
        parts.append(np.full(int(rng.integers(200, 20000)), int(rng.integers(0, 4)), dtype=np.uint8))
        parts.append(rng.integers(0, 4, size=int(rng.integers(100, 5000))).astype(np.uint8))
    recs = synth.pan_genome(400_000, 8) + [np.concatenate(parts)]
    got = []
    for tune in (0, 1 << 23, 256):
        d = api.DeBWT(k=32, tune=tune)
        if cap:
            d.set_range_cap(cap)
        d.load_records(recs)
        d.build()
        got.append(tuple(d.fetch()) + (d.fetch_array(api.ARR_DISTINCT_KEYS) if not cap else None,))
        with pytest.raises(api.DebwtError):
            d.fetch_array(api.ARR_SORTED_KEYS)       # not after a whole build, in one range or several
        d.close()
    for other in got[1:]:
        assert np.array_equal(got[0][0], other[0]) and np.array_equal(got[0][1], other[1]) and got[0][2] == other[2]
        if not cap:
            assert np.array_equal(got[0][3], other[3])
    if not cap:
        # driven stage by stage the sorted keys are there, whole and in order
        d = api.DeBWT(k=32)
        d.load_records(recs)
        d.kmer_sort_rle()
        sk = d.fetch_array(api.ARR_SORTED_KEYS)
        assert bool((sk[1:] >= sk[:-1]).all()) and np.array_equal(np.unique(sk), got[0][3])
        d.close()


def test_blocks_of_up_to_32_rows_several_to_a_wave_leave_their_rows_as_a_wave_each_does(api, oracle):
    """k_blue_tiny holds a block of up to 16 rows in a 16-lane group (four blocks to a wave; 17..32 rows: two) and runs the
    rounds of k_blue_refine for tie groups of up to 32 rows, stable ranks included: the sorted rows of those blocks -- not only
    the BWT -- equal those of the wave-per-block kernel (tune bit 24), on ten near-identical genomes (blocks of ten rows that tie
    for hundreds of SP symbols), on reads (blocks of a few rows) and on periodic stretches."""
    from debwt_amd import synth
    rng = np.random.default_rng(31)
    per = []
    for _ in range(40):
        per.append(np.tile(rng.integers(0, 4, size=int(rng.integers(2, 9))).astype(np.uint8), int(rng.integers(20, 300))))
        per.append(rng.integers(0, 4, size=int(rng.integers(40, 900))).astype(np.uint8))
    genome = rng.integers(0, 4, size=30_000).astype(np.uint8)
    reads = [genome[p:p + 100].copy() for p in rng.integers(0, len(genome) - 100, size=3000)]
    n16 = n32 = 0
    for recs in (synth.pan_genome(600_000, 10), reads, [np.concatenate(per)]):
        got = []
        for tune in (0, 1 << 24):
            d = api.DeBWT(k=32, tune=tune)
            d.load_records(recs)
            d.build()
            got.append(tuple(d.fetch()) + (d.fetch_array(api.ARR_BLUE),))
            if tune == 0:
                bound = d.fetch_array(api.ARR_BLUE_BOUND).astype(np.int64)
                sizes = np.diff(np.concatenate([[-1], bound]))
                n16 += int((sizes <= 16).sum()); n32 += int(((sizes > 16) & (sizes <= 32)).sum())
                small = np.repeat(sizes <= 32, sizes)          # rows of the blocks k_blue_tiny takes
            d.close()
        a, b = got
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
        # (the rows of larger blocks may differ in order: a class of one symbol is finished wherever its rows land)
        assert np.array_equal(a[3][:len(small)][small], b[3][:len(small)][small])
    assert n16 > 100 and n32 > 0, (n16, n32)
    ow, oh, od, _ = oracle.build_bwt(oracle.sym_from_codes(reads), 32)
    d = api.DeBWT(k=32)
    d.load_records(reads)
    d.build()
    w, h, dr = d.fetch()
    assert np.array_equal(w, ow) and np.array_equal(h, oh) and dr == od
    d.close()


def test_randomised_parity_sweep(api, oracle):
    """200 random small collections x random k x random key-range caps x the alternative device paths (cursor atomics,
    64-bit cursors, no tie-group hand-off, separate run-length passes), two builds per context, against the oracle (scripts/gpu_fuzz.py runs the
    same sweep for thousands of cases)."""
    from debwt_amd import synth
    rng = np.random.default_rng(31337)
    for c in range(200):
        kind = int(rng.integers(0, 3))
        if kind == 0:
            recs = _adversarial(rng)
        elif kind == 1:
            recs = synth.pan_genome(int(rng.integers(2000, 30000)), int(rng.integers(1, 9)), seed=int(rng.integers(1, 1 << 30)))
        else:
            unit = rng.integers(0, 4, size=int(rng.integers(34, 60))).astype(np.uint8)
            parts = []
            for _ in range(int(rng.integers(50, 1500))):
                parts.append(unit)
                parts.append(rng.integers(0, 4, size=int(rng.integers(1, 9))).astype(np.uint8))
            recs = [np.concatenate(parts), rng.integers(0, 4, size=int(rng.integers(33, 500))).astype(np.uint8)]
        k = int(rng.choice([12, 13, 16, 20, 24, 27, 31, 32]))
        tune = int(rng.choice([0, 0, 32, 48, 128, 160, 256]))
        cap = int(rng.choice([0, 0, 4096, 20000]))
        ow, oh, od, _ = oracle.build_bwt(oracle.sym_from_codes(recs), k)
        d = api.DeBWT(k=k, tune=tune)
        if cap:
            d.set_range_cap(cap)
        d.load_records(recs)
        for rep in range(2):
            d.build()
            w, h, dr = d.fetch()
            assert np.array_equal(w, ow) and np.array_equal(h, oh) and dr == od, (c, kind, k, tune, cap, rep)
        d.close()


@pytest.mark.parametrize("name", sorted(outside_domain_cases()))
def test_inputs_outside_the_reference_domain_equal_the_definition(api, oracle, name, monkeypatch):
    """outside the reference's valid domain (a base that never occurs, homopolymer-only records, short
    exact duplicates, an SP code below 32 symbols) the reference itself crashes or mis-orders, so there is nothing of it
    to be bit-exact with; the contract there is the BWT of r0#r1#...$ under A<C<G<T<#<$ BY DEFINITION (naive suffix sort,
    orc_naive_bwt).  k = 12 and 32, one key range and ranges of 4096 instances, host and device special-region module."""
    recs = outside_domain_cases()[name]
    sym = oracle.sym_from_codes(recs)
    want = oracle.naive_bwt(sym)
    n = len(sym)
    for k in (12, 32):
        for cap in (0, 4096):
            for device_special in (False, True):
                if device_special:
                    monkeypatch.setenv("DEBWT_SPECIAL_DEVICE_MIN", "0")
                else:
                    monkeypatch.delenv("DEBWT_SPECIAL_DEVICE_MIN", raising=False)
                d = api.DeBWT(k=k)
                if cap:
                    d.set_range_cap(cap)
                d.load_records(recs)
                d.build()
                w, h, dr = d.fetch()
                st = d.stats()
                d.close()
                assert (st["special_path"] == 2) == device_special, (name, k, cap, st["special_path"])
                got = oracle.unpack_bwt(w, n, h, dr)
                assert np.array_equal(got, want), (name, k, cap, device_special)
                assert len(h) == len(recs) - 1 and (len(h) < 2 or bool((np.diff(h.astype(np.int64)) > 0).all()))


@pytest.mark.parametrize("entry", [e for e in MANIFEST if e["k"] == 32 and (e["records"] >= 2000 or e["name"] in ("special_branches", "ecoli_4.6M"))][:4],
                         ids=golden_id)
def test_hash_rows_by_list_and_by_mask_are_the_reference_rows(api, entry):
    """The '#' rows (OUT.#, src/insertCase3.c:86-97): collections of up to 2^20 records have the assembly kernel append them to
    a list that is sorted afterwards, larger ones (and cfg.reserved bit 21) mark them in a mask per 32 rows and count -- the
    same rows, the reference's, either way; single- and multi-range, streamed to the host or fetched."""
    recs = golden_records(entry)
    sha = entry["sha256"]
    for tune in (0, 1 << 21):
        for cap in (0, 4096):
            d = api.DeBWT(k=32, tune=tune)
            if cap:
                d.set_range_cap(cap)
            d.load_records(recs)
            d.build()
            words, hrows, drow = d.fetch()
            assert _sha(hrows) == sha["hash"] and _sha(words) == sha["bwt"] and _sha(np.array([drow], dtype=np.uint64)) == sha["dollar"], (tune, cap)
            n = sum(len(r) for r in recs) + len(recs)
            w2 = np.zeros((n + 31) // 32, dtype=np.uint64); h2 = np.zeros(max(len(recs) - 1, 1), dtype=np.uint64); d2 = np.zeros(1, dtype=np.uint64)
            d.build_into(w2, h2, d2)
            assert _sha(h2[:len(recs) - 1]) == sha["hash"] and _sha(w2) == sha["bwt"], (tune, cap, "streamed")
            d.close()
