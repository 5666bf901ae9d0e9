"""Kernel A/B on one box: python scripts/gpu_ab.py [--wl GENOME_LEN:GENOMES:CHROMS | NAME] [--tune T] [--reps R] LIB...
LIB = "default" (debwt_amd/libdebwt_hip.so) or a name in build/variants/ (scripts/build_variants.sh).  The collection is
generated once by the native generator into /dev/shm; every library then runs in its own process on that text and prints
its stage times (best of the timed builds), the scatter-pass mean, the sort counters and a checksum of the BWT + row
lists, so that variants can be compared for equality.  LIB may carry a tune value: name@TUNE."""
import argparse, os, subprocess, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHM = "/dev/shm/debwt_ab"


def child(label, tune, reps, cap):
    import numpy as np, time
    from debwt_amd import api
    words = np.load(SHM + "_words.npy"); sep = np.load(SHM + "_sep.npy"); n = int(np.load(SHM + "_n.npy")[0])
    d = api.DeBWT(k=32, tune=tune)
    if cap: d.set_range_cap(cap)
    t0 = time.perf_counter(); d.load_packed(words, n, sep); d.build(); first = time.perf_counter() - t0
    best, wall = None, 1e9
    for it in range(reps):
        t0 = time.perf_counter(); d.build(); dt = time.perf_counter() - t0
        st = d.stats()
        if best is None or st["ms_total"] < best["ms_total"]: best = dict(st)
        wall = min(wall, dt)
    w, h, dr = d.fetch()
    crc = zlib.crc32(h.tobytes(), zlib.crc32(w.tobytes())) ^ (dr & 0xFFFFFFFF)
    keys = ("ms_sort", "ms_classify", "ms_sp", "ms_blue", "ms_assemble", "ms_total")
    print(f"{label:>14} crc={crc:08x} " + " ".join(f"{k[3:]}={best[k]:.2f}" for k in keys) +
          " wall=%.2f first=%.2fs pass=%.3fms x%d unfit=%d net=%d over=%d" % (
              wall * 1e3, first, best["radix_pass_ms"] / max(best["radix_pass_launches"], 1), best["radix_pass_launches"],
              best.get("sort_unfit_stretches", -1), best.get("sort_unfit_network", -1), best.get("sort_over_stretches", -1)),
          flush=True)
    d.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--wl", default="300000000:10:24")
    ap.add_argument("--tune", type=int, default=0)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--child", default=None)
    ap.add_argument("--cap", type=int, default=0, help="range cap (node instances per key range): several ranges on a small text")
    ap.add_argument("--keep", action="store_true", help="leave the generated text in /dev/shm (for a profiled --child run)")
    ap.add_argument("libs", nargs="*")
    a = ap.parse_args()
    if a.child is not None:
        return child(a.child, a.tune, a.reps, a.cap)
    import numpy as np, time
    from debwt_amd import synth_native as SN
    t0 = time.perf_counter()
    if ":" in a.wl:
        gl, g, c = map(int, a.wl.split(":"))
        syn = SN.Synth(gl, g, c)
    else:
        syn = SN.Synth.named(a.wl)
    words, _ = syn.words()
    np.save(SHM + "_words.npy", words); np.save(SHM + "_sep.npy", syn.sep()); np.save(SHM + "_n.npy", np.array([syn.n], dtype=np.uint64))
    print(f"workload {a.wl}: n={syn.n} records={syn.nrec}, generated in {time.perf_counter() - t0:.1f} s", flush=True)
    del words
    try:
        for lib in a.libs:
            name, _, tune = lib.partition("@")
            env = dict(os.environ)
            if name != "default":
                env["DEBWT_HIP_LIB"] = os.path.join(ROOT, "build", "variants", f"libdebwt_{name}.so")
            rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--child", lib, "--tune", tune or str(a.tune),
                                  "--reps", str(a.reps), "--cap", str(a.cap)], env=env)
            if rc: print(lib, "FAILED rc", rc, flush=True)
    finally:
        for s in ("_words.npy", "_sep.npy", "_n.npy") if not a.keep else ():
            try: os.remove(SHM + s)
            except OSError: pass


if __name__ == "__main__":
    main()
