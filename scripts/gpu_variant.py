"""Kernel A/B on one box: python scripts/gpu_variant.py NAME [NAME...]  (libs in build/variants/, see build_variants.sh).
Every variant runs in its own process on the same cached workload; prints stage times, the scatter-pass mean and a
checksum of the BWT so that variants can be compared for equality."""
import os, subprocess, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child(workload):
    import numpy as np, time
    from debwt_amd import api, synth
    cache = f"/tmp/wl_{workload}.npy"
    if os.path.exists(cache):
        recs = [np.load(cache)]
    else:
        if workload.startswith("pan:"):                      # pan:<bases per genome>:<genomes>
            _, L, G = workload.split(":")
            recs = synth.pan_genome(int(L), int(G))
        else:
            recs = synth.make_workload(workload)
        if len(recs) == 1: np.save(cache, recs[0])
    d = api.DeBWT(k=32, tune=int(os.environ.get("TUNE", "0"))); d.load_records(recs)
    best = None
    for it in range(6):
        d.build(); st = d.stats()
        if it >= 2 and (best is None or st["ms_total"] < best["ms_total"]): best = dict(st)
    w, h, dr = d.fetch()
    crc = zlib.crc32(w.tobytes())
    keys = ("ms_sort", "ms_classify", "ms_sp", "ms_blue", "ms_assemble", "ms_total")
    print(os.environ.get("DEBWT_HIP_LIB", "default").split("/")[-1], workload, "crc=%08x" % crc,
          " ".join(f"{k[3:]}={best[k]:.2f}" for k in keys),
          "pass=%.3f ms x%d" % (best["radix_pass_ms"] / max(best["radix_pass_launches"], 1), best["radix_pass_launches"]), flush=True)
    d.close()

if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        wl = os.environ.get("WL", "chr1_250M")
        for name in sys.argv[1:]:
            env = dict(os.environ)
            if name != "default":
                env["DEBWT_HIP_LIB"] = os.path.join(ROOT, "build", "variants", f"libdebwt_{name}.so")
            rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--child", wl], env=env)
            if rc: print(name, "FAILED rc", rc, flush=True)
