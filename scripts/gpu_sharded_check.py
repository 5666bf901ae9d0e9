"""One BWT over several ranks against the single-GPU build of the same collection (ranks share the box's GPU, gloo):
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29577 \
      scripts/gpu_sharded_check.py [bases_per_record=25000000] [records=4]"""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from debwt_amd import api, synth, sharded
from debwt_amd import dist as D

rank, local_rank, world = D.init(backend="gloo")
torch.cuda.set_device(0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
recs = synth.pan_genome(L, G)
n = sum(len(r) for r in recs) + len(recs)
d = api.DeBWT(k=32, device=0)
d.load_records(recs)
ref = None
if rank == 0:
    d.build(); w, h, dr = d.fetch()
    ref = (zlib.crc32(w.tobytes()), zlib.crc32(h.tobytes()), dr)
for mode in ("scan", "exchange"):
    t0 = time.time(); sharded.build_sharded(d, mode=mode); dt = time.time() - t0
    res = sharded.gather_bwt(d, n)
    if rank == 0:
        w, h, dr = res
        got = (zlib.crc32(w.tobytes()), zlib.crc32(h.tobytes()), dr)
        print(f"{world} shards, mode {mode}: n={n}, {dt*1e3:.0f} ms (gloo, one GPU), equals the single-GPU build: {got == ref}", flush=True)
D.finalize(); d.close()
