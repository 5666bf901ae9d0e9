"""The device special-region module at scale: 10^7 reads x 100 b (N*K = 3.1e8 special suffixes, 1.01 Gbp), built once,
checked by the device inverse BWT and the symbol census.  python scripts/gpu_special_big.py [reads=10000000] [len=100]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from debwt_amd import api, synth

nreads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t0 = time.time()
g = synth.base_genome(100_000_000, seed=91)
rs = np.random.default_rng(17)
starts = rs.integers(0, len(g) - L, size=nreads)
starts[::50] = starts[1::50][: len(starts[::50])]           # 2 % exact duplicates
n = nreads * (L + 1)
total = n + 32
nwords = (total + 31) // 32 + 1
sym = np.full(nwords * 32, 3, dtype=np.uint8)                # 'T' at separators and behind the end
sym[total:] = 0
view = sym[:n].reshape(nreads, L + 1)
for a in range(0, nreads, 1 << 20):                          # rows = read + its separator slot
    b = min(nreads, a + (1 << 20))
    view[a:b, :L] = g[starts[a:b, None] + np.arange(L)[None, :]]
shifts = (np.uint64(62) - np.arange(32, dtype=np.uint64) * np.uint64(2))[None, :]
words = np.zeros(nwords, dtype=np.uint64)
for a in range(0, nwords, 1 << 22):
    b = min(nwords, a + (1 << 22))
    words[a:b] = (sym[a * 32:b * 32].reshape(-1, 32).astype(np.uint64) << shifts).sum(axis=1, dtype=np.uint64)
sep = (np.arange(nreads, dtype=np.uint64) + np.uint64(1)) * np.uint64(L + 1) - np.uint64(1)
census = np.bincount(view[:, :L].ravel(), minlength=4).astype(np.int64)
print(f"{nreads} reads x {L} b: n = {n}, text made in {time.time() - t0:.0f} s", flush=True)
d = api.DeBWT(k=32)
d.load_packed(words, n, sep)
for rep in range(2):
    t0 = time.time(); d.build(); dt = time.time() - t0
    st = d.stats()
    print(f"build {rep}: wall {dt * 1e3:.0f} ms, device {st['ms_total']:.1f} ms (sort {st['ms_sort']:.1f} incl. special tables {st['ms_host_special']:.1f} ms on path {st['special_path']}), "
          f"special branches {st['special_branch_num']}, blue rows {st['blue_capacity']}", flush=True)
got = d.bwt_census().astype(np.int64)
want = census.copy(); want[3] += nreads
rep = d.verify_device()
print("census equals text:", bool((got == want).all()), "inverse BWT ok:", rep["inverse_bwt_ok"], rep["inverse_bwt"], flush=True)
d.close()
sys.exit(0 if (got == want).all() and rep["inverse_bwt_ok"] else 1)
