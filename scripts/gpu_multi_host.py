"""debwt_multi_build at size on the box's one GPU (the shards share it): python scripts/gpu_multi_host.py [workload] [shards]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from debwt_amd import api, synth_native as SN
wl = sys.argv[1] if len(sys.argv) > 1 else "grch38_3.1G"
shards = int(sys.argv[2]) if len(sys.argv) > 2 else 4
syn = SN.Synth.named(wl)
words, census = syn.words()
m = api.MultiDeBWT([0] * shards, k=32)
m.load_packed(words, syn.n, syn.sep())
for mode in ("rescan", "exchange"):
    m.set_key_mode(mode)
    t0 = time.time(); m.build(); t1 = time.time() - t0
    t0 = time.time(); m.build(); t2 = time.time() - t0
    ms, s0 = m.stats()
    rep = m.verify_device()
    print(f"{wl}: {shards} shards on one GPU, keys {mode}: first build {t1:.2f} s, second {t2*1e3:.0f} ms = {syn.n/t2/1e9:.2f} Gbp/s; {ms}; inverse BWT ok={rep['ok']} ({rep['segments']} segments, {rep['ms_walk']:.0f} ms)", flush=True)
m.close()
