"""Randomised parity run against the oracle: random small collections (the generator of tests/test_gpu_parity.py plus
larger repeat-heavy ones, runs of one symbol and tandem repeats), random k, random key-range caps and the alternative device paths (cursor atomics, 64-bit
cursors, no tie-group hand-off, either SP prefilter, no pivot rounds).  python scripts/gpu_fuzz.py [cases=300] [seed=1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from debwt_amd import api, synth
from oracle import oracle as O
from test_gpu_parity import _adversarial

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); bad = 0
for c in range(cases):
    kind = int(rng.integers(0, 5))
    if kind == 4:                            # periodic stretches: the pivot rounds of the large-block split
        parts = []
        for _ in range(int(rng.integers(2, 30))):
            parts.append(np.full(int(rng.integers(50, 6000)), int(rng.integers(0, 4)), dtype=np.uint8))
            parts.append(np.tile(rng.integers(0, 4, size=int(rng.integers(2, 7))).astype(np.uint8), int(rng.integers(20, 800))))
            parts.append(rng.integers(0, 4, size=int(rng.integers(40, 1500))).astype(np.uint8))
        cut = int(rng.integers(1, len(parts)))
        recs = [np.concatenate(parts[:cut]), np.concatenate(parts[cut:])] if rng.integers(0, 2) else [np.concatenate(parts)]
        recs = [r for r in recs if len(r) > 32]
    elif kind == 0:
        recs = _adversarial(rng)
    elif kind == 1:
        recs = synth.pan_genome(int(rng.integers(2000, 60000)), int(rng.integers(1, 9)), seed=int(rng.integers(1, 1 << 30)))
    elif kind == 2:
        unit = rng.integers(0, 4, size=int(rng.integers(34, 60))).astype(np.uint8)
        parts = []
        for _ in range(int(rng.integers(50, 3000))):
            parts.append(unit); parts.append(rng.integers(0, 4, size=int(rng.integers(1, 9))).astype(np.uint8))
        recs = [np.concatenate(parts)] + [rng.integers(0, 4, size=int(rng.integers(33, 500))).astype(np.uint8) for _ in range(int(rng.integers(0, 4)))]
    else:
        recs = [rng.integers(0, 4, size=int(rng.integers(33, 3000))).astype(np.uint8) for _ in range(int(rng.integers(1, 40)))]
    k = int(rng.choice([12, 13, 16, 20, 24, 27, 31, 32]))
    tune = int(rng.choice([0, 0, 0, 32, 48, 128, 160, 256, 2048, 4096, 4096 + 32, 4096 + 6, 8192, 8192 + 128]))
    cap = int(rng.choice([0, 0, 4096, 20000, 300000]))
    if rng.integers(0, 3) == 0: os.environ["DEBWT_SPECIAL_DEVICE_MIN"] = "0"      # special-region module on the device at any size
    else: os.environ.pop("DEBWT_SPECIAL_DEVICE_MIN", None)
    sym = O.sym_from_codes(recs)
    ow, oh, od, ost = O.build_bwt(sym, k)
    d = api.DeBWT(k=k, tune=tune)
    if cap: d.set_range_cap(cap)
    try:
        d.load_records(recs)
    except ValueError as ex:                 # the generator's own records failed validation: say which and how
        print(f"case {c} kind {kind} k {k} tune {tune} cap {cap}: {ex}; records: "
              + str([(len(r), str(r.dtype), int(r.max()) if len(r) else None, int((r > 3).sum())) for r in recs][:10]), flush=True)
        raise
    for rep in range(2):                     # a context is reusable
        d.build()
        w, h, dr = d.fetch()
        ok = np.array_equal(w, ow) and np.array_equal(h, oh) and dr == od
        if not ok:
            bad += 1
            print(f"MISMATCH case {c} kind {kind} k {k} tune {tune} cap {cap} n {len(sym)} rep {rep}", flush=True)
            break
    d.close()
    if c % 50 == 49: print(f"{c+1} cases, {bad} mismatches, {time.time()-t0:.0f}s", flush=True)
print(f"done: {cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
