import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from debwt_amd import synth_native as SN
syn = SN.Synth.named(sys.argv[1] if len(sys.argv) > 1 else "grch38_3.1G")
fa = "/tmp/cli_in.fa"
asc = np.frombuffer(b"ACGT", dtype=np.uint8)
with open(fa, "wb") as f:
    a = 0
    for j, ln in enumerate(syn._lens):
        f.write(b">chr%d\n" % j)
        for o in range(a, a + int(ln), 1 << 26):
            f.write(asc[syn.codes(0, o, min(a + int(ln), o + (1 << 26)))].tobytes())
        f.write(b"\n"); a += int(ln)
env = dict(os.environ, DEBWT_TRACE_ALLOC="1", AMD_LOG_LEVEL="0")
for rep in range(2):
    t0 = time.time()
    r = subprocess.run([os.path.join(ROOT, "cli", "deBWT"), "-o", "/tmp/cli_OUT", "-t", "16", fa], capture_output=True, text=True, env=env)
    print(f"run {rep}: wall {time.time() - t0:.2f} s"); print(r.stdout[-400:]); print(r.stderr[-6000:], flush=True)
