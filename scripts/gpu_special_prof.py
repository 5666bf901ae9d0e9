"""10^6 reads x 100 b: three builds with the special-region module on the device (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from debwt_amd import api, synth
rs = np.random.default_rng(5)
g = synth.base_genome(30_000_000, seed=77)
starts = rs.integers(0, len(g) - 100, size=1_000_000)
words, n, sep = api.pack_records([g[s:s + 100] for s in starts])
d = api.DeBWT(k=32)
d.load_packed(words, n, sep)
for rep in range(3):
    d.build()
    st = d.stats()
    print(f"special tables {st['ms_host_special']:.1f} ms (path {st['special_path']}), build {st['ms_total']:.1f} ms", flush=True)
d.close()
