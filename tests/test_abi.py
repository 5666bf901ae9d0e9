"""CPU-side checks of the boundary: the library loads, exports every symbol include/debwt_hip.h declares,
and the host-side mirror (packing, output writer) is correct.  No GPU compute."""
import os
import re

import numpy as np

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from debwt_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "debwt_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(debwt_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(_lib.SYMBOLS)
    _lib.build()
    L = _lib.lib()
    for s in declared:
        assert hasattr(L, s), s


def test_strerror_and_create_validation():
    import ctypes
    from debwt_amd import _lib
    L = _lib.lib()
    assert L.debwt_strerror(0) == b"ok"
    assert L.debwt_strerror(-4) == b"stage called out of order"
    cfg = _lib.DebwtConfig(k=11, device=0, sort_algo=0, reserved=0)
    h = ctypes.c_void_p()
    assert L.debwt_create(ctypes.byref(cfg), ctypes.byref(h)) == -1        # k outside 12..32, src/main.c:45


def test_pack_records_matches_reference_layout(oracle):
    from debwt_amd import api, synth
    recs = synth.pan_genome(1000, 3)
    words, n, sep = api.pack_records(recs)
    sym = oracle.sym_from_codes(recs)
    ref = oracle.pack_text(sym)
    assert n == len(sym) and np.array_equal(words[:len(ref)], ref)
    assert list(sep) == list(np.nonzero(sym >= 4)[0])


def test_verify_inverse_host_tool(oracle):
    from debwt_amd import api, synth
    recs = synth.pan_genome(20000, 3)
    sym = oracle.sym_from_codes(recs)
    w, h, d, _ = oracle.build_bwt(sym, 32)
    rc, inv = api.verify_inverse(w, len(sym), h, d)
    assert rc == 0 and np.array_equal(inv, sym)
    w2 = w.copy()
    w2[3] ^= np.uint64(1 << 20)
    rc2, inv2 = api.verify_inverse(w2, len(sym), h, d)
    assert rc2 != 0 or not np.array_equal(inv2, sym)


def test_write_outputs_format(tmp_path, oracle):
    from debwt_amd import api, synth
    recs = synth.pan_genome(3000, 4)
    sym = oracle.sym_from_codes(recs)
    w, h, d, _ = oracle.build_bwt(sym, 32)
    p = str(tmp_path / "OUT")
    api.write_outputs(p, w, h, d)
    assert os.path.getsize(p) == 8 * ((len(sym) + 31) // 32)                # src/insertCase3.c:115-119
    assert os.path.getsize(p + ".#") == 8 * 3 and os.path.getsize(p + ".$") == 8


def test_verify_inverse_walks_records_in_parallel(oracle):
    """debwt_verify_inverse (host tool, the job of the reference's dead LFsearch path, src/LFsearch.c:14-48): one walk
    per record started from the '#' rows, chained afterwards -- against the oracle's BWT of multi-record texts."""
    import numpy as np
    from debwt_amd import api, synth
    rng = np.random.default_rng(9)
    cases = [synth.pan_genome(20000, 7), synth.pan_genome(30000, 1),
             [rng.integers(0, 4, size=int(L)).astype(np.uint8) for L in (40, 33, 1000, 77, 5000, 34)]]
    for recs in cases:
        sym = oracle.sym_from_codes(recs)
        w, h, d, _ = oracle.build_bwt(sym, 32)
        rc, inv = api.verify_inverse(w, len(sym), h, d)
        assert rc == 0 and np.array_equal(inv, sym)
        # a damaged BWT does not close
        w2 = w.copy(); w2[len(w2) // 2] ^= np.uint64(1 << 20)
        rc2, inv2 = api.verify_inverse(w2, len(sym), h, d)
        assert rc2 != 0 or not np.array_equal(inv2, sym)


def test_special_region_module_threads_agree(monkeypatch):
    """The host special-region module (src/collect#$.c:118-157,348-602) cut over threads gives the tables of its
    single-threaded run: many short records from a small motif pool, so that special suffixes tie in long runs."""
    import ctypes
    import numpy as np
    from debwt_amd import _lib, api
    rng = np.random.default_rng(4)
    pool = [rng.integers(0, 4, size=150).astype(np.uint8) for _ in range(30)]
    recs = []
    for _ in range(3000):
        p = pool[int(rng.integers(0, len(pool)))]
        off, L = int(rng.integers(0, 30)), int(rng.integers(40, 120))
        x = p[off:off + L].copy()
        flip = rng.random(L) < 0.02
        x[flip] = rng.integers(0, 4, size=int(flip.sum()))
        recs.append(x)
    words, n, sep = api.pack_records(recs)
    L_ = _lib.lib()
    p64 = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))

    def digest(threads, k):
        monkeypatch.setenv("DEBWT_SPECIAL_THREADS", str(threads))
        monkeypatch.setenv("DEBWT_SPECIAL_PAR_MIN", "0")
        out = np.zeros(4, dtype=np.uint64)
        assert L_.debwt_special_digest(p64(words), n, p64(sep), len(sep), k, p64(out)) == 0
        return out.tolist()

    for k in (32, 16):
        ref = digest(1, k)
        assert digest(3, k) == ref and digest(8, k) == ref
    # enough records for the runs of the shortest special suffixes (one per record and length) to take the
    # all-threads sort of long runs
    recs = []
    for _ in range(40000):
        p = pool[int(rng.integers(0, len(pool)))]
        off, L = int(rng.integers(0, 60)), int(rng.integers(34, 70))
        recs.append(p[off:off + L].copy())
    words, n, sep = api.pack_records(recs)
    ref = digest(1, 32)
    assert digest(5, 32) == ref and digest(8, 32) == ref


def test_key_mode_cost_model():
    """debwt_shard_key_mode (host only): on one node of 2..8 GPUs the keys are cheaper to re-read from each GPU's copy of
    the text than to ship; the exchange wins once a shard's slice is small against the text and the links are fast."""
    import ctypes
    from debwt_amd import _lib
    L = _lib.lib()
    x, r = ctypes.c_double(), ctypes.c_double()
    for world in (1, 2, 4, 8):
        assert L.debwt_shard_key_mode(30_000_000_000, world, 0.0, ctypes.byref(x), ctypes.byref(r)) == 1
        assert 0 < r.value < x.value
    assert L.debwt_shard_key_mode(30_000_000_000, 64, 0.0, ctypes.byref(x), ctypes.byref(r)) == 0
    slow = x.value
    assert L.debwt_shard_key_mode(30_000_000_000, 64, 400.0, ctypes.byref(x), ctypes.byref(r)) == 0 and x.value < slow
    one = L.debwt_shard_key_mode(3_100_000_000, 1, 0.0, ctypes.byref(x), ctypes.byref(r))
    assert one in (0, 1) and x.value > 0 and r.value > 0
