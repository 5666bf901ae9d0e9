"""PCIe-inclusive rate: packed text in host memory -> packed BWT + '#'/'$' rows in host memory (load + build + fetch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from debwt_amd import api, synth
recs = synth.make_workload("chr1_250M")
words, n, sep = api.pack_records(recs)
d = api.DeBWT(k=32)
best = None
for it in range(6):
    t0 = time.perf_counter(); d.load_packed(words, n, sep); t1 = time.perf_counter()
    d.build(); t2 = time.perf_counter()
    w, h, dr = d.fetch(); t3 = time.perf_counter()
    r = (t3 - t0, t1 - t0, t2 - t1, t3 - t2)
    if it >= 2 and (best is None or r[0] < best[0]): best = r
print(f"n={n}: load {best[1]*1e3:.1f} ms + build {best[2]*1e3:.1f} ms + fetch {best[3]*1e3:.1f} ms = {best[0]*1e3:.1f} ms -> {n/best[0]/1e9:.2f} Gbp/s "
      f"(pageable host buffers; build alone {n/best[2]/1e9:.2f} Gbp/s)")
