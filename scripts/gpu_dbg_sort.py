import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from debwt_amd import api
from conftest import golden_manifest, golden_records
ent = [e for e in golden_manifest() if e["name"] == "contigs_2000" and e["k"] == 32][0]
recs = golden_records(ent)
K = 31
out = []
for x in recs:
    x = np.asarray(x, dtype=np.uint64); m = len(x) - K + 1
    node = np.zeros(m, dtype=np.uint64)
    for j in range(K): node = (node << np.uint64(2)) | x[j:j + m]
    pred = np.concatenate([np.array([3], dtype=np.uint64), x[:m - 1]])
    out.append((node << np.uint64(2)) | pred)
want = np.sort(np.concatenate(out))
d = api.DeBWT(k=32); d.load_records(recs); d.kmer_sort_rle()
got = d.fetch_array(api.ARR_SORTED_KEYS)
bad = np.nonzero(got != want)[0]
print("n", len(want), "mismatches", len(bad))
if len(bad):
    b = int(bad[0]); print("first", b, "tile", b // 896, "offset in raster", b % 896, "last", int(bad[-1]))
    T = 0
    while (len(want) >> (8 * T)) > 64 and T < 4: T += 1
    print("T", T)
    pre = want >> np.uint64(64 - 8 * T)
    # bucket containing b
    lo = b
    while lo > 0 and pre[lo - 1] == pre[b]: lo -= 1
    hi = b
    while hi + 1 < len(pre) and pre[hi + 1] == pre[b]: hi += 1
    print("bucket of first mismatch: [%d, %d] size %d" % (lo, hi, hi - lo + 1))
    print("is multiset of got == want in window?", np.array_equal(np.sort(got[lo:hi+1]), want[lo:hi+1]))
    runs = np.split(bad, np.nonzero(np.diff(bad) > 1)[0] + 1)
    print("runs of mismatches:", [(int(r[0]), int(r[-1])) for r in runs[:10]], len(runs))
dk = d.fetch_array(api.ARR_DISTINCT_KEYS); print("distinct ok", np.array_equal(dk, np.unique(want)), len(dk), len(np.unique(want)))
