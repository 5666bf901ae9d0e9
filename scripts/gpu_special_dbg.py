import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from debwt_amd import api, synth
# 1. the radix sort on keys shaped like the module's: heavily duplicated chunk in the high bits, unique payload below
d = api.DeBWT(k=32)
rng = np.random.default_rng(3)
for count, cb, pb, distinct in ((2000, 32, 11, 2000), (620000, 31, 20, 5), (620000, 31, 20, 3000), (620000, 20, 20, 40), (62000, 31, 16, 2), (3_100_000, 31, 22, 7)):
    chunk = rng.integers(0, 1 << cb, size=distinct, dtype=np.uint64)[rng.integers(0, distinct, size=count)]
    if distinct <= 7: chunk[: count // 2] = (1 << cb) - 1
    keys = (chunk << np.uint64(pb)) | np.arange(count, dtype=np.uint64)
    perm = rng.permutation(count)
    keys = keys[perm]
    a = torch.from_numpy(keys.view(np.int64)).cuda(); b = torch.empty_like(a)
    d.radix_sort_device(a.data_ptr(), b.data_ptr(), count, cb + pb)
    got = a.cpu().numpy().view(np.uint64)
    print("radix", count, cb, pb, distinct, "ok" if np.array_equal(got, np.sort(keys)) else "WRONG", flush=True)
d.close()
os.environ["DEBWT_SPECIAL_DEBUG"] = "1"
for name, recs in (("contigs_2000", synth.read_set(2000, 3000, 8000, 4_000_000)), ("reads_20000", synth.read_set(20000, 60, 400, 1_000_000))):
    d = api.DeBWT(k=32)
    d.load_records(recs)
    print(name, d.special_compare(), flush=True)
    d.close()
