"""Size classes of the multi-in blocks of a named workload (rows per block, blocks and rows per class):
python scripts/gpu_block_sizes.py [workload=pan10x300M] [k=32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from debwt_amd import api, synth_native as SN

wl = sys.argv[1] if len(sys.argv) > 1 else "pan10x300M"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 32
syn = SN.Synth.named(wl)
text = SN.PinnedArray(syn.nwords)
syn.words_into(text.ptr)
d = api.DeBWT(k=k)
d.load_packed(text.a, syn.n, syn.sep())
d.build()
bound = d.fetch_array(api.ARR_BLUE_BOUND).astype(np.int64)
sizes = np.diff(np.concatenate([[-1], bound]))
print(f"{wl}: n = {syn.n}, {len(sizes)} blocks, {int(sizes.sum())} rows")
edges = [0, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 1 << 62]
for lo, hi in zip(edges[:-1], edges[1:]):
    sel = sizes[(sizes > lo) & (sizes <= hi)]
    if len(sel):
        print(f"  {lo + 1:5d}..{hi if hi < (1 << 60) else 'inf':>5}: {len(sel):10d} blocks ({100.0 * len(sel) / len(sizes):5.1f} %)  {int(sel.sum()):12d} rows ({100.0 * sel.sum() / sizes.sum():5.1f} %)")
d.close()
