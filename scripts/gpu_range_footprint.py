"""Build time and device memory of ONE collection against the number of key ranges it is sorted in (debwt_set_range_cap): what a
one-shot run could trade for a smaller footprint (a cold process waits for the driver to clear what it is handed,
profiles/r06_malloc_cost.txt).  python scripts/gpu_range_footprint.py [workload = grch38_3.1G]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from debwt_amd import api, synth_native as SN
wl = sys.argv[1] if len(sys.argv) > 1 else "grch38_3.1G"
syn = SN.Synth.named(wl)
text = SN.PinnedArray(syn.nwords)
syn.words_into(text.ptr)
sep = syn.sep()
ref = None
for P in (1, 2, 4, 8, 16):
    torch.cuda.synchronize()
    free0, total = torch.cuda.mem_get_info()
    d = api.DeBWT(k=32)
    if P > 1: d.set_range_cap((syn.n + P - 1) // P + (1 << 20))
    d.load_packed(text.a, syn.n, sep)
    d.build()
    ms = []
    for _ in range(3):
        d.build(); ms.append(d.stats()["ms_total"])
    free1, _ = torch.cuda.mem_get_info()
    w, h, dr = d.fetch()
    if ref is None: ref = w.copy()
    st = d.stats()
    print(f"{wl}: cap n/{P:<2d}: {min(ms):8.1f} ms per build (sort {st['ms_sort']:.1f}), {(free0 - free1) / 2**30:6.1f} GiB of device memory, same BWT: {np.array_equal(w, ref)}", flush=True)
    d.close(); del d
    time.sleep(6)                                  # (what was released is being cleared)
