"""Host special-region module on read-like collections (no GPU work): python scripts/special_bench.py  (DEBWT_TRACE_SPECIAL=1: phases)"""
import sys, time, ctypes, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from debwt_amd import api, _lib
L = _lib.lib()
p64 = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
rng = np.random.default_rng(1)
for nrec, ln in ((100_000, 150), (1_000_000, 100)):
    recs = [rng.integers(0, 4, size=ln).astype(np.uint8) for _ in range(nrec)]
    t0 = time.time(); words, n, sep = api.pack_records(recs); tp = time.time() - t0
    out = np.zeros(4, dtype=np.uint64)
    t0 = time.time(); rc = L.debwt_special_digest(p64(words), n, p64(sep), len(sep), 32, p64(out)); dt = time.time() - t0
    print(f"records={nrec} len={ln}: pack {tp:.1f}s special module {dt*1e3:.0f} ms rc={rc}", flush=True)
