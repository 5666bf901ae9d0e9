"""Host FASTA ingest rate (no GPU work): python scripts/ingest_bench.py [workload=chr1_250M] [line width=60]
Writes the workload as FASTA to /tmp, then times debwt_pack_fasta (file in the page cache) for 1..N threads."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from debwt_amd import api, synth

wl = sys.argv[1] if len(sys.argv) > 1 else "chr1_250M"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 60
path = f"/tmp/ingest_{wl}_{L}.fa"
if not os.path.exists(path):
    with open(path, "wb") as f:
        for r, codes in enumerate(synth.make_workload(wl)):
            asc = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
            nfull = len(asc) // L
            body = np.empty((nfull, L + 1), dtype=np.uint8)
            body[:, :L] = asc[:nfull * L].reshape(nfull, L); body[:, L] = 10
            f.write(b">rec%d synthetic\n" % r); f.write(body.tobytes()); f.write(asc[nfull * L:].tobytes() + b"\n")
sz = os.path.getsize(path)
ncpu = len(os.sched_getaffinity(0))
print(f"{path}: {sz/1e6:.0f} MB, {ncpu} CPUs", flush=True)
for th in sorted({1, 2, 4, 8, 16, ncpu}):
    if th > max(ncpu, 8): continue
    best = None
    for it in range(4):
        w, n, sep, sr, sp = api.pack_fasta(path, th)
        if best is None or sr + sp < best[0]: best = (sr + sp, sr, sp)
    print(f"threads={th:3d}: map {best[1]*1e3:6.1f} ms + parse/pack {best[2]*1e3:7.1f} ms -> {sz/best[0]/1e9:6.2f} GB/s", flush=True)
