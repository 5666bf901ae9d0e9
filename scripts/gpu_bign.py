"""Texts beyond 2^32 positions on ONE MI355X: the key space is built in prefix ranges over the resident text.
python scripts/gpu_bign.py [total_bases=5000000000] [records=24] [--cap N] [--pan] [--k32only] [--inverse]
Checks: k-invariance (k=32 vs k=24 give the identical BWT), '#' rows ascending, symbol census = text census."""
import hashlib, sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
from debwt_amd import api, synth

args = [a for a in sys.argv[1:] if not a.startswith("--")]
total = int(args[0]) if args else 5_000_000_000
nrec = int(args[1]) if len(args) > 1 else 24
cap = int(sys.argv[sys.argv.index("--cap") + 1]) if "--cap" in sys.argv else None
if "--pan" in sys.argv:      # SURVEY 8d distribution P: nrec genomes = one base genome each with independent SNPs at 1e-3
    t0 = time.time(); recs = synth.pan_genome(total // nrec, nrec); tg = time.time() - t0
else:
    t0 = time.time(); recs = synth.chromosomes(total, nrec); tg = time.time() - t0
n = sum(len(r) for r in recs) + len(recs)
print(f"generated n={n} ({n / 2**32:.2f} x 2^32) in {tg:.0f}s", flush=True)
t0 = time.time(); words, n2, sep = api.pack_records(recs); tp = time.time() - t0
census = np.zeros(4, dtype=np.int64)
for r in recs:
    census += np.bincount(r, minlength=4)[:4]
del recs
print(f"packed in {tp:.0f}s", flush=True)
res = {}
for k in ((32,) if "--k32only" in sys.argv else (32, 24)):
    d = api.DeBWT(k=k, tune=int(__import__("os").environ.get("TUNE", "0")))
    if cap: d.set_range_cap(cap)
    t0 = time.time(); d.load_packed(words, n, sep); tl = time.time() - t0
    t0 = time.time(); d.build(); t1 = time.time() - t0      # first build allocates
    t0 = time.time(); d.build(); tb = time.time() - t0
    st = d.stats()
    w, h, dr = d.fetch()
    res[k] = (hashlib.sha256(w.tobytes()).hexdigest(), h.copy(), dr)
    print(f"k={k}: load {tl:.1f}s, first build {t1:.2f}s, steady-state build {tb*1e3:.1f} ms = {n/tb/1e9:.2f} Gbp/s; "
          f"stages ms sort+local classify {st['ms_sort']:.1f} global classify {st['ms_classify']:.1f} sp {st['ms_sp']:.1f} "
          f"blue {st['ms_blue']:.1f} asm {st['ms_assemble']:.1f}; distinct={st['distinct_keys']} red={st['red_capacity']} "
          f"blue={st['blue_capacity']} blocks={st['blue_bound_num']} S={st['sp_len']} large={st['blue_large_blocks']}", flush=True)
    if k == 32:
        # symbol census of the BWT = census of the text (+ '#'/'$' rows stored as 3)
        b = w.view(np.uint8)
        cnt = np.zeros(4, dtype=np.int64)
        lut = np.zeros((256, 4), dtype=np.int64)
        for v in range(256):
            for s in range(4):
                lut[v, (v >> (2 * s)) & 3] += 1
        hist = np.bincount(b, minlength=256)
        cnt = (hist[:, None] * lut).sum(axis=0)
        pad = (-n) % 32
        cnt[0] -= pad                                      # unused tail bits are 0
        cnt[3] -= len(sep)                                 # '#' and '$' rows are stored as 3
        print("symbol census matches the text:", bool((cnt == census).all()), "; '#' rows ascending:",
              bool((np.diff(h.astype(np.int64)) > 0).all()), flush=True)
    if k == 32 and "--inverse" in sys.argv:
        # inverse BWT by LF walk on the host (sequential: ~5 min per Gbp) with a heartbeat for the job runner
        import threading
        stop = threading.Event()
        def beat():
            t = 0
            while not stop.wait(60):
                t += 1; print(f"  ... inverse BWT running, {t} min", flush=True)
        th = threading.Thread(target=beat, daemon=True); th.start()
        t0 = time.time(); rc, inv = api.verify_inverse(w, n, h, dr); stop.set()
        ok = rc == 0
        if ok:
            # compare with the packed text: symbol j at bits 2*(31-(j&31)) of word j>>5, separators from sep
            txt = np.empty(n, dtype=np.uint8)
            wb = words[:(n + 31) // 32].byteswap().view(np.uint8)           # big-endian bytes: 4 symbols per byte, first on top
            q = np.empty((len(wb), 4), dtype=np.uint8)
            q[:, 0] = wb >> 6; q[:, 1] = (wb >> 4) & 3; q[:, 2] = (wb >> 2) & 3; q[:, 3] = wb & 3
            txt[:] = q.reshape(-1)[:n]
            txt[sep.astype(np.int64)] = 4; txt[n - 1] = 5
            ok = bool(np.array_equal(inv, txt))
        print(f"inverse BWT reproduces the text: {ok} (rc={rc}, {time.time()-t0:.0f}s)", flush=True)
    d.close()
if 24 in res: print("k-invariance 32 vs 24:", res[32][0] == res[24][0] and np.array_equal(res[32][1], res[24][1]) and res[32][2] == res[24][2], flush=True)
