"""Cost of the HBM path for heavy-tail blocks: python scripts/gpu_large_block.py [copies]"""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
from debwt_amd import api
from oracle import oracle as O
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
rng = np.random.default_rng(5)
unit = rng.integers(0, 4, size=40).astype(np.uint8)
parts = []
for i in range(copies):
    parts.append(unit); parts.append(rng.integers(0, 4, size=int(rng.integers(3, 9))).astype(np.uint8))
recs = [np.concatenate(parts), rng.integers(0, 4, size=500).astype(np.uint8)]
d = api.DeBWT(k=32, tune=int(__import__("os").environ.get("TUNE", "0"))); d.load_records(recs)
d.build(); t0 = time.time(); d.build(); dt = time.time() - t0
st = d.stats()
print(f"copies={copies} n={st['n']} build {dt*1e3:.1f} ms, blue stage {st['ms_blue']:.1f} ms, large blocks {st['blue_large_blocks']} max {st['blue_max_block']}")
if st['n'] < 3_000_000:
    ow, oh, od, _ = O.build_bwt(O.sym_from_codes(recs), 32)
    w, h, dr = d.fetch()
    print("equals oracle:", np.array_equal(w, ow) and np.array_equal(h, oh) and dr == od)
