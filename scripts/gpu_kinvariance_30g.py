"""k-invariance at full size: the BWT of a collection does not depend on k.  Builds the named workload (default pan10x3G =
30 Gbp) with k = 32 and k = 24 on one GPU and compares the results word for word (plus '#' rows and '$' row).
python scripts/gpu_kinvariance_30g.py [workload] [k1,k2,...]"""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from debwt_amd import api, synth_native as SN
wl = sys.argv[1] if len(sys.argv) > 1 else "pan10x3G"
ks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "32,24").split(",")]
syn = SN.Synth.named(wl)
text = SN.PinnedArray(syn.nwords)
t0 = time.time(); syn.words_into(text.ptr); print(f"{wl}: n={syn.n} generated in {time.time()-t0:.1f} s", flush=True)
sep = syn.sep()
ref = None
out = SN.PinnedArray((syn.n + 31) // 32); oh = SN.PinnedArray(max(syn.nrec - 1, 1)); od = SN.PinnedArray(1)
for k in ks:
    d = api.DeBWT(k=k)
    d.load_packed(text.a, syn.n, sep)
    t0 = time.time(); d.build(); t1 = time.time() - t0
    t0 = time.time(); d.build(); t2 = time.time() - t0           # warm: the workspace is there
    st = d.stats()
    ver = d.verify_device()                                       # inverse BWT on the device against the loaded text
    print(f"k={k}: warm build {t2 * 1e3:.1f} ms (sort {st['ms_sort']:.1f} classify {st['ms_classify']:.1f} SP {st['ms_sp']:.1f} blue {st['ms_blue']:.1f}), "
          f"inverse BWT ok: {ver['inverse_bwt_ok']}", flush=True)
    if not ver["inverse_bwt_ok"]: print("VERIFY FAILED"); sys.exit(1)
    d.fetch_into(out.a, oh.a, od.a)
    d.close()
    crc = zlib.crc32(out.a.view(np.uint8)); hcrc = zlib.crc32(oh.a[:syn.nrec - 1].view(np.uint8))
    print(f"k={k}: first build {t1:.2f} s, distinct keys {st['distinct_keys']}, red {st['red_capacity']}, blue {st['blue_capacity']}, "
          f"blocks {st['blue_bound_num']}, S {st['sp_len']}; crc32(bwt)={crc:08x} crc32(#rows)={hcrc:08x} $row={int(od.a[0])}", flush=True)
    cur = (crc, hcrc, int(od.a[0]))
    if ref is None: ref = cur
    elif cur != ref: print("MISMATCH between k values"); sys.exit(1)
print("identical for k in", ks)
