"""Times debwt_kmer_sort_rle alone for the libraries named (build/variants): python scripts/gpu_sortonly.py NAME..."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "--child":
    import numpy as np
    from debwt_amd import api, synth
    wl = sys.argv[2]
    cache = f"/tmp/wl_{wl}.npy"
    if os.path.exists(cache): recs = [np.load(cache)]
    else:
        recs = synth.pan_genome(*map(int, wl.split(":")[1:])) if wl.startswith("pan:") else synth.make_workload(wl)
        if len(recs) == 1: np.save(cache, recs[0])
    d = api.DeBWT(k=32, tune=int(os.environ.get("TUNE", "0"))); d.load_records(recs)
    best = 1e9
    for it in range(6):
        t0 = time.perf_counter(); d.kmer_sort_rle(); dt = (time.perf_counter() - t0) * 1e3
        if it >= 2: best = min(best, dt)
    print(os.environ.get("DEBWT_HIP_LIB", "default").split("/")[-1], wl, "kmer_sort_rle %.2f ms" % best, flush=True)
else:
    wl = os.environ.get("WL", "chr1_250M")
    for name in sys.argv[1:]:
        env = dict(os.environ); env["DEBWT_HIP_LIB"] = os.path.join(ROOT, "build", "variants", f"libdebwt_{name}.so")
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", wl], env=env)
