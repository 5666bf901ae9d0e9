"""The special-region module on the device against the host module (debwt_special_compare: element-wise) and, through
full builds, against the oracle; then its time on read sets.  python scripts/gpu_special.py [cases=150]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from debwt_amd import api, synth
from oracle import oracle as O
from test_gpu_parity import _adversarial

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(11)
bad = 0
t0 = time.time()


def check(tag, recs, k, rounds=None):
    global bad
    if rounds is not None:
        os.environ["DEBWT_SPECIAL_MAX_ROUNDS"] = str(rounds)
    d = api.DeBWT(k=k)
    d.load_records(recs)
    mm = d.special_compare()
    d.close()
    os.environ.pop("DEBWT_SPECIAL_MAX_ROUNDS", None)
    if any(mm):
        bad += 1
        print(f"TABLE MISMATCH {tag} k={k} rounds={rounds} records={len(recs)}: {mm}", flush=True)


# 1. tables, element by element: adversarial small collections (duplicates, shared ends, prefix-duplicates), any k
for c in range(cases):
    recs = _adversarial(rng)
    if c % 3 == 0:          # many short records with shared ends and exact duplicates
        recs = synth.read_set(int(rng.integers(2, 400)), 50, int(rng.integers(60, 300)), 20_000, seed=int(rng.integers(1, 1 << 30)),
                              snap=int(rng.choice([1, 4, 8])), dup_every=int(rng.choice([0, 3, 16])))
    k = int(rng.choice([12, 13, 16, 20, 24, 27, 31, 32]))
    check(f"case{c}", recs, k, rounds=int(rng.choice([64, 64, 1, 2, 5])))
print(f"{cases} table comparisons, {bad} bad, {time.time()-t0:.0f}s", flush=True)
for name, recs in (("contigs_2000", synth.read_set(2000, 3000, 8000, 4_000_000)), ("reads_20000", synth.read_set(20000, 60, 400, 1_000_000)),
                   ("reads_1e5", synth.read_set(100_000, 60, 300, 3_000_000, seed=0xBEEF5))):
    for k in (32, 20):
        check(name, recs, k)
        check(name, recs, k, rounds=3)
print(f"named sets done, {bad} bad, {time.time()-t0:.0f}s", flush=True)

# 2. full builds with the device module forced at any size, against the oracle
os.environ["DEBWT_SPECIAL_DEVICE_MIN"] = "0"
for c in range(cases):
    recs = _adversarial(rng) if c % 2 else synth.read_set(int(rng.integers(2, 300)), 50, 200, 10_000, seed=int(rng.integers(1, 1 << 30)), snap=4, dup_every=5)
    k = int(rng.choice([12, 16, 21, 32]))
    cap = int(rng.choice([0, 0, 4096]))
    ow, oh, od, ost = O.build_bwt(O.sym_from_codes(recs), k)
    d = api.DeBWT(k=k)
    if cap: d.set_range_cap(cap)
    d.load_records(recs)
    d.build()
    w, h, dr = d.fetch()
    st = d.stats()
    if not (np.array_equal(w, ow) and np.array_equal(h, oh) and dr == od and st["special_path"] == 2
            and st["special_branch_num"] == ost["special_branch_num"]):
        bad += 1
        print(f"BUILD MISMATCH case {c} k {k} cap {cap} records {len(recs)} path {st['special_path']}", flush=True)
    d.close()
os.environ.pop("DEBWT_SPECIAL_DEVICE_MIN")
print(f"{cases} forced-device builds, {bad} bad, {time.time()-t0:.0f}s", flush=True)

# 3. time: 10^6 reads of 100 b (SURVEY 8f-1 / VERDICT: special tables < 50 ms), device against host threads
rs = np.random.default_rng(5)
g = synth.base_genome(30_000_000, seed=77)
starts = rs.integers(0, len(g) - 100, size=1_000_000)
recs = [g[s:s + 100] for s in starts]
words, n, sep = api.pack_records(recs)
for label, env in (("device", {}), ("host threads", {"DEBWT_SPECIAL_DEVICE_MIN": str(1 << 62)})):
    os.environ.update(env)
    d = api.DeBWT(k=32)
    d.load_packed(words, n, sep)
    for rep in range(2):
        d.build()
        st = d.stats()
        print(f"10^6 reads x 100 b, {label}: special tables {st['ms_host_special']:.1f} ms (path {st['special_path']}, "
              f"threads {st['special_threads']}), build {st['ms_total']:.1f} ms, branches {st['special_branch_num']}", flush=True)
    if label == "device":
        ok = d.verify_device()["inverse_bwt_ok"]
        dig = d.bwt_census().tolist()
    else:
        print("census equal:", dig == d.bwt_census().tolist(), "inverse ok (device path):", ok, flush=True)
    d.close()
    for k_ in env: os.environ.pop(k_)
print(f"done: {bad} bad")
sys.exit(1 if bad else 0)
