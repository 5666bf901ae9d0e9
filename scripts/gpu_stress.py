"""Robustness runs beyond the unit tests: python scripts/gpu_stress.py [big|records|all]"""
import hashlib, sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
from debwt_amd import api, synth
from oracle import oracle as O

what = sys.argv[1] if len(sys.argv) > 1 else "all"

def run(recs, k, algo=0):
    d = api.DeBWT(k=k, sort_algo=algo)
    t0 = time.time(); d.load_records(recs); t1 = time.time()
    d.build(); t2 = time.time()
    out = d.fetch(); st = d.stats(); d.close()
    return out, st, (t1 - t0, t2 - t1)

if what in ("records", "all"):
    rng = np.random.default_rng(3)
    for nrec, lo, hi in ((2000, 3000, 8000), (20000, 60, 400)):
        base = synth.uniform_codes(200_000, seed=77)
        recs = []
        for r in range(nrec):
            L = int(rng.integers(lo, hi)); p = int(rng.integers(0, len(base) - L))
            x = base[p:p + L].copy()
            m = rng.random(L) < 0.01
            x[m] = (x[m] + 1) & 3
            recs.append(x)
        sym = O.sym_from_codes(recs)
        t0 = time.time(); ow, oh, od, ost = O.build_bwt(sym, 32); to = time.time() - t0
        (w, h, d_), st, tm = run(recs, 32)
        ok = np.array_equal(w, ow) and np.array_equal(h, oh) and d_ == od
        print(f"records={nrec} n={len(sym)} ok={ok} oracle {to:.1f}s gpu load {tm[0]:.2f}s build {tm[1]*1e3:.1f} ms "
              f"(host special {st['ms_host_special']:.1f} ms) special_branches={st['special_branch_num']} "
              f"large_blocks={st['blue_large_blocks']} max_block={st['blue_max_block']}", flush=True)
        assert ok

if what in ("big", "all"):
    for total, nrec in ((1_000_000_000, 24),):
        t0 = time.time(); recs = synth.chromosomes(total, nrec); tg = time.time() - t0
        (w, h, d_), st, tm = run(recs, 32)
        sha32 = hashlib.sha256(w.tobytes()).hexdigest()
        print(f"n={st['n']} gen {tg:.0f}s load {tm[0]:.1f}s build {tm[1]*1e3:.1f} ms -> {st['n']/tm[1]/1e9:.2f} Gbp/s; "
              f"stages ms: sort {st['ms_sort']:.1f} classify {st['ms_classify']:.1f} sp {st['ms_sp']:.1f} "
              f"blue {st['ms_blue']:.1f} asm {st['ms_assemble']:.1f}; red={st['red_capacity']} blue={st['blue_capacity']} "
              f"large_blocks={st['blue_large_blocks']}", flush=True)
        (w2, h2, d2), st2, tm2 = run(recs, 24)
        print("k-invariance (32 vs 24):", hashlib.sha256(w2.tobytes()).hexdigest() == sha32 and np.array_equal(h, h2) and d_ == d2,
              f"build k=24 {tm2[1]*1e3:.1f} ms", flush=True)
        assert len(h) == nrec - 1 and (np.diff(h.astype(np.int64)) > 0).all()
