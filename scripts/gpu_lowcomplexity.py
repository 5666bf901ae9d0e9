"""Low-complexity input (runs of one symbol, tandem repeats): python scripts/gpu_lowcomplexity.py  (TUNE=1024: network only)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from debwt_amd import api
rng = np.random.default_rng(5)
parts = []
for _ in range(300):
    parts.append(np.full(int(rng.integers(200, 30000)), int(rng.integers(0, 4)), dtype=np.uint8))
    parts.append(np.tile(rng.integers(0, 4, size=int(rng.integers(2, 7))).astype(np.uint8), int(rng.integers(50, 3000))))
    parts.append(rng.integers(0, 4, size=int(rng.integers(100, 5000))).astype(np.uint8))
recs = [np.concatenate(parts)]
d = api.DeBWT(k=32, tune=int(os.environ.get("TUNE", "0"))); d.load_records(recs)
d.build(); t0 = time.time(); d.build(); dt = time.time() - t0
st = d.stats()
import zlib
w, h, dr = d.fetch()
print(f"tune={os.environ.get('TUNE','0')} n={st['n']} build {dt*1e3:.1f} ms, blue stage {st['ms_blue']:.1f} ms, large blocks {st['blue_large_blocks']} max {st['blue_max_block']} crc={zlib.crc32(w.tobytes()):08x}")
