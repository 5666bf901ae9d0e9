"""The device inverse-BWT verifier (debwt_verify_device: rank structure, backward search for segment starts, one LF walk
per segment -- the job of the reference's dead LFsearch path, src/LFsearch.c:49-166) and the BASELINE.json
configurations on one GPU: chr1-sized against the oracle, GRCh38-sized by its size-independent properties."""
import hashlib

import numpy as np
import pytest

from conftest import golden_id, golden_manifest, golden_records

pytestmark = pytest.mark.gpu
MANIFEST = golden_manifest()


@pytest.fixture(scope="module")
def api():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    from debwt_amd import api as A
    return A


@pytest.mark.parametrize("entry", [e for e in MANIFEST if e["k"] in (12, 32)], ids=golden_id)
def test_verify_device_accepts_reference_golden(api, entry):
    recs = golden_records(entry)
    d = api.DeBWT(k=entry["k"])
    d.load_records(recs)
    d.build()
    for segments in (0, 1, 7, 1000):
        r = d.verify_device(segments=segments)
        assert r["inverse_bwt_ok"], (segments, r)
        assert r["inverse_bwt"]["steps"] == entry["n"] - 1
    d.close()


def test_verify_device_agrees_with_host_walk_and_rejects_corruption(api):
    import torch
    from debwt_amd import synth
    recs = synth.pan_genome(150_000, 4)
    n = sum(len(r) for r in recs) + len(recs)
    d = api.DeBWT(k=32)
    d.load_records(recs)
    d.build()
    words, hrows, drow = d.fetch()
    rc, inv = api.verify_inverse(words, n, hrows, drow)                  # the host tool (one walk per record)
    assert rc == 0
    good = torch.from_numpy(words.view(np.int64)).cuda()
    r = d.verify_device(good.data_ptr(), hrows, drow, segments=64)
    assert r["inverse_bwt_ok"] and r["inverse_bwt"]["segments"] > 8, r   # several segments found by backward search
    rng = np.random.default_rng(3)
    # (a) two rows with different symbols swapped: the census survives, the walk does not
    for trial in range(4):
        bad = words.copy()
        while True:
            i, j = (int(x) for x in rng.integers(0, n, size=2))
            si = int(bad[i >> 5] >> np.uint64(2 * (31 - (i & 31)))) & 3
            sj = int(bad[j >> 5] >> np.uint64(2 * (31 - (j & 31)))) & 3
            if si != sj and i not in hrows and j not in hrows and drow not in (i, j):
                break
        for pos, s in ((i, sj), (j, si)):
            sh = np.uint64(2 * (31 - (pos & 31)))
            bad[pos >> 5] = (bad[pos >> 5] & ~(np.uint64(3) << sh)) | (np.uint64(s) << sh)
        t = torch.from_numpy(bad.view(np.int64)).cuda()
        r = d.verify_device(t.data_ptr(), hrows, drow, segments=64)
        assert not r["inverse_bwt_ok"], (trial, i, j, r)
    # (b) a wrong '#' row, (c) a wrong '$' row
    h2 = hrows.copy()
    h2[1] = h2[1] + 1 if h2[1] + 1 < h2[2] else h2[1] - 1
    assert not d.verify_device(good.data_ptr(), h2, drow, segments=64)["inverse_bwt_ok"]
    assert not d.verify_device(good.data_ptr(), hrows, (drow + 5) % n, segments=64)["inverse_bwt_ok"]
    d.close()


def test_config1_chr1_250M_equals_oracle(api, oracle):
    """BASELINE.json configs[1]: chr1-sized (250 Mbp, k = 32), the whole BWT against the CPU oracle."""
    from debwt_amd import synth_native as SN
    syn = SN.Synth.named("chr1_250M")
    words, census = syn.words()
    d = api.DeBWT(k=32)
    d.load_packed(words, syn.n, syn.sep())
    d.build()
    w, h, dr = d.fetch()
    st = d.stats()
    assert d.verify_device()["inverse_bwt_ok"]
    d.close()
    ow, oh, od, ost = oracle.build_bwt(oracle.sym_from_codes([syn.codes(0)]), 32)
    assert np.array_equal(w, ow) and np.array_equal(h, oh) and dr == od
    for a, b in (("red_capacity", "red_capacity"), ("blue_capacity", "blue_capacity"), ("blue_bound_num", "blue_bound_num"),
                 ("sp_len", "sp_len"), ("case3num", "case3num")):
        assert st[a] == ost[b]


@pytest.mark.parametrize("workload", ["grch38_3.1G", "real_small"])
def test_config2_properties(api, workload):
    """BASELINE.json configs[2]: GRCh38-sized (3.1 Gbp in 24 records) on one GPU -- beyond what the oracle holds, so the
    size-independent properties: the inverse BWT reproduces the text (device LF walks), the symbol census equals the
    text's, the '#' rows ascend, and the result does not depend on k (k = 32 and k = 24 give the identical BWT).
    real_small: the same checks on distribution R (Alu-like family, satellites, homopolymers)."""
    from debwt_amd import synth_native as SN
    syn = SN.Synth.named(workload)
    words, census = syn.words()
    sep = syn.sep()
    digests = []
    for k in (32, 24):
        d = api.DeBWT(k=k)
        d.load_packed(words, syn.n, sep)
        d.build()
        got = d.bwt_census().astype(np.int64)
        want = census.astype(np.int64).copy()
        want[3] += syn.nrec
        assert (got == want).all()
        r = d.verify_device()
        assert r["inverse_bwt_ok"], r
        w, h, dr = d.fetch()
        assert len(h) == syn.nrec - 1 and (np.diff(h.astype(np.int64)) > 0).all()
        digests.append((hashlib.sha256(w.tobytes()).hexdigest(), hashlib.sha256(h.tobytes()).hexdigest(), dr))
        d.close()
    assert digests[0] == digests[1]


def test_k_over_its_range_at_config2_size(api):
    """The reference takes any -k from 12 to 32 for any input (src/main.c:41-47).  GRCh38-sized text, k = 16 and k = 20 beside
    k = 32 (k = 24 is in test_config2_properties): the symbol census, the device inverse BWT, and the identical BWT (k-invariance,
    SURVEY 4.4).  k = 16 has 1.07 G branching 15-mers out of 4^15: the node table takes its 2^32 slots (64 GB) and the build
    releases the idle range workspace when that allocation does not fit (round 5 answered DEBWT_ERANGE); k = 20 takes the
    prefilter by minimizers of 12 symbols (round 5: a bitmap probe per position, 1.45 x the time of k = 32)."""
    import zlib
    from debwt_amd import synth_native as SN
    syn = SN.Synth.named("grch38_3.1G")
    words, census = syn.words()
    sep = syn.sep()
    want = census.astype(np.int64).copy()
    want[3] += syn.nrec
    ref = None
    for k in (32, "compact", 20, 16):
        # "compact": k = 32 behind debwt_reserve(..., ONE_SHOT | COMPACT), the plan cli/deBWT asks for -- key ranges of 2^29
        # instances (six here instead of one, a third of the device memory): the same rows
        d = api.DeBWT(k=32 if k == "compact" else k)
        if k == "compact":
            d.reserve(syn.n, syn.nrec, one_shot=True, compact=True)
        d.load_packed(words, syn.n, sep)
        d.build()
        assert (d.bwt_census().astype(np.int64) == want).all(), k
        r = d.verify_device()
        assert r["inverse_bwt_ok"] and r["inverse_bwt"]["mismatches"] == 0, (k, r)
        if k == "compact":                                   # (radix_pass_keys: the keys of the first key range)
            assert 0 < d.stats()["radix_pass_keys"] < syn.n // 2, d.stats()["radix_pass_keys"]
        w, h, dr = d.fetch()
        d.close()
        cur = (zlib.crc32(w.view(np.uint8)), zlib.crc32(h.view(np.uint8)), dr)
        if ref is None:
            ref = cur
        assert cur == ref, k


@pytest.mark.parametrize("workload", ["pan4x3.1G", "pan10x3G"])
def test_config3_config4_collections_on_one_gpu(api, workload):
    """BASELINE.json configs[3] and configs[4] name 4 and 8 GPUs; their collections -- 4 x GRCh38-sized = 12.4 Gbp and
    10 genomes x 3.0 Gbp = 30 Gbp, the size the metric is quoted on -- also fit ONE MI355X (key ranges one after the
    other), so the build itself is tested here at full size: symbol census = the text's, '#' rows complete and ascending,
    '$' row present, and the inverse BWT walked on the device reproduces the text (1.2e10 / 3e10 LF steps).  The sharded
    form of the same builds is what bench.py --gpus N runs (and tests/test_sharded.py checks at small sizes)."""
    from debwt_amd import synth_native as SN
    syn = SN.Synth.named(workload)
    text = SN.PinnedArray(syn.nwords)
    census = syn.words_into(text.ptr)
    d = api.DeBWT(k=32)
    d.load_packed(text.a, syn.n, syn.sep())
    d.build()
    st = d.stats()
    assert st["n"] == syn.n and st["nrec"] == syn.nrec and st["n_main"] == syn.n - syn.nrec * 31
    got = d.bwt_census().astype(np.int64)
    want = census.astype(np.int64).copy()
    want[3] += syn.nrec
    assert (got == want).all()
    _, h, dr = d.fetch_small()
    assert len(h) == syn.nrec - 1 and (np.diff(h.astype(np.int64)) > 0).all() and 0 <= dr < syn.n
    r = d.verify_device()
    assert r["inverse_bwt_ok"] and r["inverse_bwt"]["steps"] == syn.n - 1 and r["inverse_bwt"]["mismatches"] == 0, r
    d.close()
    text.free()


@pytest.mark.parametrize("devices,case,k,cap,tune", [([0, 0], "pan", 32, 0, 0), ([0, 0, 0], "chrom", 24, 150_000, 0),
                                                     ([0] * 8, "many", 32, 0, 0), ([0] * 5, "pan", 20, 60_000, 0),
                                                     ([0, 0, 0], "reads", 32, 0, 0), ([0, 0], "reads", 24, 40_000, 0),
                                                     ([0], "pan", 32, 0, 0), ([0, 0, 0], "pan", 32, 0, 32)])
def test_multi_gpu_host_from_one_process(api, oracle, devices, case, k, cap, tune, monkeypatch):
    """debwt_multi_build: one host thread per shard, exchanges as device-to-device copies -- here with all shards on
    the box's one GPU (on a node the ordinals differ and the copies cross xGMI).  2, 3, 5 and 8 shards, with several
    exchange rounds per shard where a range cap is set, against the oracle; then the device inverse BWT."""
    from debwt_amd import synth
    recs = {"pan": lambda: synth.pan_genome(300_000, 3), "many": lambda: synth.pan_genome(20_000, 9, seed=5),
            "reads": lambda: synth.read_set(3000, 60, 300, 200_000, seed=11),      # special-region module on the device in every shard
            "chrom": lambda: synth.chromosomes(2_000_000, 4)}[case]()
    ow, oh, od, _ = oracle.build_bwt(oracle.sym_from_codes(recs), k)
    m = api.MultiDeBWT(devices, k=k, tune=tune)               # tune 32: blue entries placed through block cursors
    m.load_records(recs)
    if cap:
        for r in range(len(devices)):
            api._lib.lib().debwt_set_range_cap(m._shard_ctx(r), cap)
    for mode, want in (("exchange", 0), ("rescan", 1), ("auto", 1), ("exchange", 0)):   # reusable, in either key mode
        m.set_key_mode(mode)
        m.build()
        w, h, dr = m.fetch()
        assert np.array_equal(w, ow) and np.array_equal(h, oh) and dr == od
        ms, s0 = m.stats()
        assert ms["ngpus"] == len(devices) and (ms["rounds"] > 1) == bool(cap) and ms["key_mode"] == want
        assert (ms["key_bytes_in"] > 0) == (want == 0 and len(devices) > 1)
    assert m.verify_device()["ok"] == 1
    m.close()

@pytest.mark.parametrize("mode", ["rescan", "exchange"])
def test_multi_gpu_host_serial_mode_reports_every_shard(api, oracle, mode):
    """debwt_multi_set_serial: the shards of a build take turns between the barriers (how per-shard times are measured on a box
    with one GPU) -- the same BWT as the oracle's and as the concurrent build; debwt_multi_get_shard_report: every shard's
    keys / blocks / rows add up to the whole, its bins are contiguous, what the shards sent is what the shards received, and
    every step it ran has a time."""
    from debwt_amd import synth
    recs = synth.pan_genome(300_000, 4, seed=9)
    sym = oracle.sym_from_codes(recs)
    ow, oh, od, ost = oracle.build_bwt(sym, 32)
    m = api.MultiDeBWT([0, 0, 0, 0, 0], k=32)
    m.load_records(recs)
    m.set_key_mode(mode)
    for serial in (True, False, True):
        m.set_serial(serial)
        m.build()
        w, h, dr = m.fetch()
        assert np.array_equal(w, ow) and np.array_equal(h, oh) and dr == od, (mode, serial)
    reps = [m.shard_report(r) for r in range(5)]
    ms, _ = m.stats()
    assert [r["bins"][0] for r in reps][0] == 0 and reps[-1]["bins"][1] == 4096
    assert all(a["bins"][1] == b["bins"][0] for a, b in zip(reps, reps[1:]))
    n_main = len(sym) - len(recs) * 31
    assert sum(r["keys"] for r in reps) == n_main and sum(r["rows"] for r in reps) == len(sym)
    assert sum(r["blocks"] for r in reps) == ost["blue_bound_num"] and sum(r["blue_rows"] for r in reps) == ost["blue_capacity"]
    for x in ("keys", "facts", "sp", "blue"):
        assert sum(r["bytes_in"][x] for r in reps) == sum(r["bytes_out"][x] for r in reps), x
    assert (sum(r["bytes_in"]["keys"] for r in reps) > 0) == (mode == "exchange")
    assert reps[0]["bytes_in"]["rows"] > 0 and all(r["bytes_in"]["rows"] == 0 for r in reps[1:])
    must = {"shard_histogram", "shard_classify_local", "shard_sp_flags", "shard_sp_emit", "shard_blue_route", "shard_blue_place", "blue_sort",
            "bwt_assemble", "shard_export"} | ({"kmer_sort_rle"} if mode == "rescan" else {"shard_partition_keys", "shard_sort_range"})
    for r in reps:
        assert must <= set(r["ms"]), (r["shard"], sorted(must - set(r["ms"])))
        assert r["ctx"]["n"] == len(sym) and r["key_ranges"] >= 1
    assert "concat_rows" in reps[0]["ms"] and "waiting" not in reps[0]["ms"]      # (serial: turns, not waits)
    assert m.verify_device()["ok"] == 1
    m.close()


@pytest.mark.parametrize("ngpus", [1, 2])
def test_multi_gpu_host_exchanges_over_rccl(api, oracle, ngpus):
    """The C host with its exchanges carried by RCCL (ncclCommInitAll in the one process, one grouped ncclSend / ncclRecv
    alltoallv per exchange; librccl loaded on demand): a communicator group of ONE on this box's GPU -- every message is a
    send to itself -- and of two where the box has two GPUs; a device ordinal that repeats is refused (RCCL wants
    distinct GPUs), peer copies stay available."""
    import torch
    from debwt_amd import synth
    if torch.cuda.device_count() < ngpus:
        pytest.skip("this box has fewer GPUs than shards")
    recs = synth.pan_genome(300_000, 3)
    ow, oh, od, _ = oracle.build_bwt(oracle.sym_from_codes(recs), 32)
    m = api.MultiDeBWT(list(range(ngpus)), k=32)
    m.load_records(recs)
    m.set_exchange("rccl")
    for mode in ("exchange", "rescan"):
        m.set_key_mode(mode)
        m.build()
        w, h, dr = m.fetch()
        assert np.array_equal(w, ow) and np.array_equal(h, oh) and dr == od, mode
        assert m.stats()[0]["exchange_backend"] == 1
    m.set_exchange("peer")
    m.build()
    assert np.array_equal(m.fetch()[0], ow) and m.stats()[0]["exchange_backend"] == 0
    m.close()
    if ngpus == 1:
        m2 = api.MultiDeBWT([0, 0], k=32)
        with pytest.raises(api.DebwtError):
            m2.set_exchange("rccl")
        m2.close()


def test_million_reads_property(api):
    """10^6 reads of 60..140 bases (3.1 * 10^7 special suffixes: the special-region module on the device): symbol census of
    the BWT against the text's, '#' rows ascending and complete, inverse BWT on the device -- the size the oracle cannot
    reach in test time."""
    from debwt_amd import synth
    recs = synth.read_set(1_000_000, 60, 140, 5_000_000, seed=21)
    d = api.DeBWT(k=32)
    d.load_records(recs)
    d.build()
    st = d.stats()
    assert st["special_path"] == 2 and st["nrec"] == 1_000_000, st
    census = np.bincount(np.concatenate(recs), minlength=4).astype(np.int64)
    census[3] += len(recs)
    assert np.array_equal(d.bwt_census().astype(np.int64), census)
    _, hrows, drow = d.fetch_small()
    assert len(hrows) == len(recs) - 1 and bool((np.diff(hrows.astype(np.int64)) > 0).all())
    rep = d.verify_device()
    assert rep["inverse_bwt_ok"], rep
    d.close()
