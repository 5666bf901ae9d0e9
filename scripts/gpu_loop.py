"""A few steady-state builds of a workload (for profilers): python scripts/gpu_loop.py WORKLOAD [builds]  (pan:L:G or a synth name)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from debwt_amd import api, synth
wl = sys.argv[1]
if wl.startswith("pan:"):
    _, L, G = wl.split(":"); recs = synth.pan_genome(int(L), int(G))
else:
    recs = synth.make_workload(wl)
d = api.DeBWT(k=32, tune=int(os.environ.get("TUNE", "0"))); d.load_records(recs)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    d.build()
print({k: round(v, 2) for k, v in d.stats().items() if k.startswith("ms_")})
