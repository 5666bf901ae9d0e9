"""Duplicated contigs (exact copies of 20-80 kb, copies that differ only at the end): the record starts tie for thousands of symbols --
the jump rounds of the device special-region module against the host comparison (DEBWT_SPECIAL_HOST_TIES).  python scripts/gpu_dup_contigs.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from debwt_amd import api
rng = np.random.default_rng(9)
base = [rng.integers(0, 4, size=int(rng.integers(20_000, 80_000))).astype(np.uint8) for _ in range(1500)]
recs = []
for i, r in enumerate(base):
    recs.append(r)
    if i % 2 == 0: recs.append(r.copy())          # exact duplicates: tied to the end
    if i % 5 == 0:                                 # and a copy that differs in its last 1000 bases
        q = r.copy(); q[-1000:] = rng.integers(0, 4, size=1000); recs.append(q)
print(len(recs), "contigs,", sum(len(r) for r in recs), "bases")
for env in ({}, {"DEBWT_SPECIAL_HOST_TIES": "1"}):
    os.environ.pop("DEBWT_SPECIAL_HOST_TIES", None); os.environ.update(env)
    d = api.DeBWT(k=32); d.load_records(recs); d.build()
    best = 1e9
    for _ in range(3):
        d.build(); st = d.stats(); best = min(best, st["ms_total"])
    import zlib
    w, h, dr = d.fetch()
    print("host comparison of the long ties" if env else "jump rounds on the device", "build %.2f ms" % best, "special path", st["special_path"],
          "ms_host_special %.2f" % st["ms_host_special"], "crc %08x" % zlib.crc32(w.tobytes()), d.special_compare())
    d.close()
