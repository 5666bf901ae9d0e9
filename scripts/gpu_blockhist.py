"""Block-size census of a workload: python scripts/gpu_blockhist.py [bases_per_genome] [genomes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from debwt_amd import api, synth
L = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 10
recs = synth.pan_genome(L, G)
d = api.DeBWT(k=32); d.load_records(recs)
for it in range(2):
    t0 = time.time(); d.build(); dt = time.time() - t0
st = d.stats()
print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items()}, "wall %.1f ms" % (dt * 1e3))
bound = d.fetch_array(api.ARR_BLUE_BOUND).astype(np.int64)
sizes = np.diff(np.concatenate([[-1], bound]))
edges = [0, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 1 << 20, 1 << 40]
h, _ = np.histogram(sizes, bins=edges)
rows, _ = np.histogram(sizes, bins=edges, weights=sizes)
for a, b, c, r in zip(edges[:-1], edges[1:], h, rows):
    print(f"blocks of [{a},{b}) rows: {c:9d} blocks, {int(r):11d} rows")
