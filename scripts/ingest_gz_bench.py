"""Ingest rate of plain, gzip and block-gzip (BGZF) FASTA on the host threads: python scripts/ingest_gz_bench.py [Mbp=1000] [threads]
(host only: no GPU is used)"""
import os, struct, sys, time, zlib, gzip
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from debwt_amd import api, synth_native as SN
mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else SN.default_threads()
syn = SN.Synth(mbp * 1_000_000, 1, 24)
asc = np.frombuffer(b"ACGT", dtype=np.uint8)
d = "/dev/shm/debwt_gzbench"; os.makedirs(d, exist_ok=True)
fa = f"{d}/x.fa"
with open(fa, "wb") as f:
    a = 0
    for j, ln in enumerate(syn._lens):
        f.write(b">chr%d\n" % j)
        s = asc[syn.codes(0, a, a + int(ln))].tobytes()
        for o in range(0, len(s), 1 << 24): f.write(s[o:o + (1 << 24)])
        f.write(b"\n"); a += int(ln)
data = open(fa, "rb").read()
only = os.environ.get("DEBWT_GZBENCH_ONLY")                # e.g. "x.gzip6,x.bgzf" (x.fa always runs: it is the text the others are compared with)
want = lambda name: not only or any(name.startswith(o) for o in only.split(","))
t0 = time.time()
for level in (1, 6):
    if not want(f"x.gzip{level}"): continue                                   # one member each: gzip -1 and gzip's default, -6
    with open(f"{d}/x.gzip{level}.fa.gz", "wb") as f:
        co = zlib.compressobj(level, zlib.DEFLATED, 31)
        for o in range(0, len(data), 1 << 24): f.write(co.compress(data[o:o + (1 << 24)]))
        f.write(co.flush())
t1 = time.time()
def bgzf_block(a, B=65280):
    chunk = data[a:a + B] if a < len(data) else b""
    co = zlib.compressobj(1, zlib.DEFLATED, -15); body = co.compress(chunk) + co.flush()
    return (b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1)
            + body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
from concurrent.futures import ThreadPoolExecutor
with open(f"{d}/x.bgzf.fa.gz", "wb") as f, ThreadPoolExecutor(threads) as pool:      # (zlib releases the interpreter lock)
    for blk in (pool.map(bgzf_block, list(range(0, len(data), 65280)) + [len(data)], chunksize=64) if want("x.bgzf") else ()): f.write(blk)
# several plain members: one per record (24 chromosome-like members) and the text cut into 3 (few, large members)
starts = [i for i in range(len(data)) if data[i:i + 1] == b">"] if len(data) < (1 << 20) else None
import re
starts = [m.start() for m in re.finditer(b">", data)] + [len(data)]
with open(f"{d}/x.members24.fa.gz", "wb") as f:
    for a, b_ in (zip(starts[:-1], starts[1:]) if want("x.members24") else ()): f.write(gzip.compress(data[a:b_], compresslevel=6, mtime=0))
third = [0, starts[len(starts) // 3], starts[2 * len(starts) // 3], len(data)]
with open(f"{d}/x.members3.fa.gz", "wb") as f:
    for a, b_ in (zip(third[:-1], third[1:]) if want("x.members3") else ()): f.write(gzip.compress(data[a:b_], compresslevel=6, mtime=0))
print(f"{mbp} Mbp FASTA: {len(data) / 1e9:.2f} GB; gzip -1 and -6 written in {t1 - t0:.0f} s, BGZF in {time.time() - t1:.0f} s; {threads} host threads", flush=True)
ref = None
for name, env in (("x.fa", {}), ("x.gzip1.fa.gz", {"DEBWT_GZ_SERIAL": "1"}), ("x.gzip1.fa.gz", {}), ("x.gzip6.fa.gz", {"DEBWT_GZ_SERIAL": "1"}),
                  ("x.gzip6.fa.gz", {}), ("x.bgzf.fa.gz", {}), ("x.members24.fa.gz", {"DEBWT_GZ_SERIAL": "1"}), ("x.members24.fa.gz", {}),
                  ("x.members3.fa.gz", {})):
    if only and name != "x.fa" and (env or not want(name)): continue
    best = None
    os.environ.pop("DEBWT_GZ_SERIAL", None)
    os.environ.update(env)
    label = name + (" (serial: zlib's gzread, as in round 4)" if env else "")
    for _ in range(2):
        time.sleep(0.5)                                   # (the buffers of the call before are released behind its back: gz_parallel.h release_later)
        t0 = time.time(); w, n, sep, s_read, s_pack = api.pack_fasta(f"{d}/{name}", threads); dt = time.time() - t0
        best = min(best, (dt, s_read, s_pack)) if best else (dt, s_read, s_pack)
    if ref is None: ref = (w.copy(), n)
    same = n == ref[1] and np.array_equal(w, ref[0])
    print(f"{label:56s} file {os.path.getsize(f'{d}/{name}') / 1e9:6.2f} GB: read/inflate {best[1]:6.2f} s + parse/pack {best[2]:5.2f} s = "
          f"{len(data) / 1e9 / (best[1] + best[2]):6.2f} GB/s of FASTA text ({mbp / 1e3 / (best[1] + best[2]):.2f} Gbp/s), same text: {same}", flush=True)
if not os.environ.get("DEBWT_GZBENCH_KEEP"):              # (kept for a trace of one file: DEBWT_TRACE_GZ=1)
    for name in os.listdir(d): os.remove(f"{d}/{name}")
    os.rmdir(d)
