"""BASELINE configs[2]: GRCh38-sized collection (~3.1 Gbp, 24 records) on ONE MI355X, in core.
python scripts/gpu_config3.py [total_bases] [--inverse]"""
import hashlib, sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
from debwt_amd import api, synth

total = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3_100_000_000
t0 = time.time(); recs = synth.chromosomes(total, 24); tg = time.time() - t0
n = sum(len(r) for r in recs) + len(recs)
print(f"generated n={n} in {tg:.0f}s", flush=True)
res = {}
for k in (32, 24):
    d = api.DeBWT(k=k)
    t0 = time.time(); d.load_records(recs); tl = time.time() - t0
    d.build()                                     # first build allocates
    t0 = time.time(); d.build(); tb = time.time() - t0
    st = d.stats()
    w, h, dr = d.fetch()
    res[k] = (hashlib.sha256(w.tobytes()).hexdigest(), h.copy(), dr)
    print(f"k={k}: load {tl:.1f}s, steady-state build {tb*1e3:.1f} ms = {n/tb/1e9:.2f} Gbp/s; stages ms sort {st['ms_sort']:.1f} "
          f"classify {st['ms_classify']:.1f} sp {st['ms_sp']:.1f} blue {st['ms_blue']:.1f} asm {st['ms_assemble']:.1f}; "
          f"red={st['red_capacity']} blue={st['blue_capacity']} S={st['sp_len']} large={st['blue_large_blocks']}", flush=True)
    if k == 32 and "--inverse" in sys.argv:
        t0 = time.time(); rc, inv = api.verify_inverse(w, n, h, dr)
        ok = rc == 0
        o = 0
        for i, r in enumerate(recs):
            ok = ok and np.array_equal(inv[o:o + len(r)], r) and inv[o + len(r)] == (5 if i + 1 == len(recs) else 4)
            o += len(r) + 1
        print(f"inverse BWT reproduces the text: {ok} ({time.time()-t0:.0f}s)", flush=True)
    d.close()
print("k-invariance 32 vs 24:", res[32][0] == res[24][0] and np.array_equal(res[32][1], res[24][1]) and res[32][2] == res[24][2])
