"""Where the blue stage's rows sit and how deep they tie: block-size census + SP-suffix tie depth of adjacent rows in the
sorted blocks.  python scripts/gpu_blue_census.py [workload=pan_1G] (pan-genome 10 x 100 Mbp in 24 records each)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from debwt_amd import api, synth_native as SN
syn = SN.Synth(100_000_000, 10, 24)
words, _ = syn.words()
d = api.DeBWT(k=32); d.load_packed(words, syn.n, syn.sep())
d.build(); d.build()
st = d.stats()
print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items() if k.startswith("ms_") or k in ("blue_capacity", "blue_bound_num", "sp_len")})
bound = d.fetch_array(api.ARR_BLUE_BOUND).astype(np.int64)
sizes = np.diff(np.concatenate([[-1], bound]))
edges = [0, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 1 << 40]
h, _ = np.histogram(sizes, bins=edges)
rows, _ = np.histogram(sizes, bins=edges, weights=sizes)
for a, b, c, r in zip(edges[:-1], edges[1:], h, rows):
    print(f"blocks of [{a},{b}) rows: {c:9d} blocks, {int(r):11d} rows ({100*r/sizes.sum():.1f} %)")
# tie depth: for adjacent rows of the sorted blue table inside one block, the SP symbols they share
blue = d.fetch_array(api.ARR_BLUE)
sp = d.fetch_array(api.ARR_SP_SYMBOLS)
pos = (blue >> np.uint64(4)).astype(np.int64)
sym = (blue & np.uint64(15)).astype(np.int64)
first = np.zeros(len(blue), dtype=bool); first[np.concatenate([[0], bound[:-1] + 1])] = True
idx = np.nonzero(~first)[0]
rng = np.random.default_rng(1); idx = rng.choice(idx, size=min(len(idx), 2_000_000), replace=False)
a, b = pos[idx - 1], pos[idx]
lcp = np.zeros(len(idx), dtype=np.int64); live = np.ones(len(idx), dtype=bool)
for step in range(4000):
    ok = live & (a + step < len(sp)) & (b + step < len(sp))
    eq = np.zeros(len(idx), dtype=bool); eq[ok] = sp[a[ok] + step] == sp[b[ok] + step]
    live &= eq; lcp[live] += 1
    if not live.any(): break
same = sym[idx - 1] == sym[idx]
print("adjacent rows of a block (sample of %d): share >= 42 SP symbols %.1f %%, >= 84: %.1f %%, >= 210: %.1f %%, >= 1000: %.1f %%; mean %.1f"
      % (len(idx), 100 * (lcp >= 42).mean(), 100 * (lcp >= 84).mean(), 100 * (lcp >= 210).mean(), 100 * (lcp >= 1000).mean(), lcp.mean()))
print("  of those sharing >= 42 symbols: same BWT symbol %.1f %%" % (100 * same[lcp >= 42].mean()))
