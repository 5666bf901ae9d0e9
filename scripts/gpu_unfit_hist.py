"""Shape of the stretches the bucket finish of the key sort finds above a wave tile: sizes, largest bucket, distinct keys.
python scripts/gpu_unfit_hist.py [GENOME_LEN:GENOMES:CHROMS]   (one key range: keep the collection below 4 G positions)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from debwt_amd import api, synth_native as SN
gl, g, c = map(int, (sys.argv[1] if len(sys.argv) > 1 else "300000000:10:24").split(":"))
syn = SN.Synth(gl, g, c)
words, _ = syn.words()
d = api.DeBWT(k=32); d.load_packed(words, syn.n, syn.sep()); d.kmer_sort_rle()
keys = torch.from_numpy(d.fetch_array(api.ARR_SORTED_KEYS).view(np.int64)).cuda()
d.close()
n = keys.numel()
T = 0
while (n >> (8 * T)) > 64 and T < 4: T += 1
pre = keys >> (64 - 8 * T)
head = torch.ones(n, dtype=torch.bool, device="cuda"); head[1:] = pre[1:] != pre[:-1]
def nz(m):                                                              # nonzero in pieces (torch overflows above 2^31 elements)
    out = []
    for a in range(0, m.numel(), 1 << 30):
        out.append(torch.nonzero(m[a:a + (1 << 30)]).flatten() + a)
    return torch.cat(out)
bpos = nz(head)                                                         # bucket starts
H, CAP = 896, 1024
x = torch.arange(0, n, H, device="cuda")
idx = torch.searchsorted(bpos, x)                                      # first bucket start >= x
ends = torch.cat([bpos, torch.tensor([n], device="cuda")])
bnd = ends[idx.clamp(max=len(ends) - 1)]
size = torch.cat([bnd[1:], torch.tensor([n], device="cuda")]) - bnd
unfit = size > CAP
print(f"keys {n}, T {T}, tiles {len(x)}, unfit {int(unfit.sum())}, keys in unfit {int(size[unfit].sum())} ({float(size[unfit].sum()) / n:.3f})")
edges = [1024, 1280, 1536, 2048, 3072, 4096, 8192, 1 << 40]
su = size[unfit]
for a, b in zip(edges[:-1], edges[1:]):
    m = (su > a) & (su <= b)
    print(f"  stretch size ({a}, {b}]: {int(m.sum())} stretches, {int(su[m].sum())} keys")
# bucket sizes weighted by keys
bsz = ends[1:] - ends[:-1]
for a, b in zip([0, 16, 64, 128, 256, 512, 1024, 2048, 4096], [16, 64, 128, 256, 512, 1024, 2048, 4096, 1 << 40]):
    m = (bsz > a) & (bsz <= b)
    print(f"  bucket size ({a}, {b}]: {int(m.sum())} buckets, {int(bsz[m].sum())} keys ({float(bsz[m].sum()) / n:.4f})")
# distinct keys inside unfit stretches (sample of 2000 stretches)
dh = torch.ones(n, dtype=torch.bool, device="cuda"); dh[1:] = keys[1:] != keys[:-1]
cs = torch.empty(n, dtype=torch.int64, device="cuda")
run = 0
for a in range(0, n, 1 << 30):
    cs[a:a + (1 << 30)] = torch.cumsum(dh[a:a + (1 << 30)].to(torch.int64), 0) + run
    run = int(cs[min(a + (1 << 30), n) - 1])
ui = nz(unfit)
s0 = bnd[ui]; e0 = s0 + size[ui]
dist = cs[e0 - 1] - cs[s0] + 1
ratio = dist.double() / size[ui].double()
print("  distinct / keys in unfit stretches: mean %.3f, quantiles 10/50/90 %%: %.3f %.3f %.3f" % (
    float(ratio.mean()), *[float(torch.quantile(ratio, q)) for q in (0.1, 0.5, 0.9)]))
print("  distinct per unfit stretch: mean %.1f, max %d" % (float(dist.double().mean()), int(dist.max())))
