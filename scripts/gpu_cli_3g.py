"""cli/deBWT -- the drop-in program -- end to end on a GRCh38-sized FASTA (3.1 Gbp, 24 records): wall time of the whole
process (context, FASTA ingest on the host threads, cold build, fetch, write of OUT / OUT.# / OUT.$) and its own
breakdown; the output is compared with a build of the same text through the API.  python scripts/gpu_cli_3g.py [workload] [gz [gz6]]"""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from debwt_amd import synth_native as SN

wl = sys.argv[1] if len(sys.argv) > 1 else "grch38_3.1G"
syn = SN.Synth.named(wl)
fa = "/tmp/cli_in.fa"
asc = np.frombuffer(b"ACGT", dtype=np.uint8)
t0 = time.time()
with open(fa, "wb") as f:
    for g in range(syn.genomes):
        a = 0
        for j, ln in enumerate(syn._lens):
            f.write(b">g%d_chr%d\n" % (g, j))
            for o in range(a, a + int(ln), 1 << 26):
                e = min(a + int(ln), o + (1 << 26))
                f.write(asc[syn.codes(g, o, e)].tobytes())
            f.write(b"\n")
            a += int(ln)
print(f"wrote {os.path.getsize(fa) / 1e9:.2f} GB FASTA in {time.time() - t0:.1f} s", flush=True)
for rep in range(2):                      # second run: the file is in the page cache for sure
    t0 = time.time()
    r = subprocess.run([os.path.join(ROOT, "cli", "deBWT"), "-o", "/tmp/cli_OUT", "-t", "16", fa], capture_output=True, text=True)
    dt = time.time() - t0
    print(f"run {rep}: exit {r.returncode}, wall {dt:.2f} s = {syn.n / dt / 1e9:.2f} Gbp/s end to end", flush=True)
    print(r.stdout.strip(), flush=True)
    if r.returncode: print(r.stderr[-2000:]); sys.exit(1)
sha_cli = hashlib.sha256(open("/tmp/cli_OUT", "rb").read()).hexdigest()
# A/B of the compact plan the program asks for (DEBWT_RESERVE_COMPACT: key ranges of 2^29 instances, a third of the device memory):
# two runs right behind each other without it
for rep in range(2):
    t0 = time.time()
    r = subprocess.run([os.path.join(ROOT, "cli", "deBWT"), "-o", "/tmp/cli_OUT", "-t", "16", fa], capture_output=True, text=True,
                       env=dict(os.environ, DEBWT_CLI_NO_COMPACT="1"))
    dt = time.time() - t0
    print(f"without the compact plan, run {rep}: exit {r.returncode}, wall {dt:.2f} s = {syn.n / dt / 1e9:.2f} Gbp/s end to end", flush=True)
    print(r.stdout.strip().split("\n")[-1], flush=True)
for rep in range(2, 4):
    t0 = time.time()
    r = subprocess.run([os.path.join(ROOT, "cli", "deBWT"), "-o", "/tmp/cli_OUT", "-t", "16", fa], capture_output=True, text=True)
    dt = time.time() - t0
    print(f"run {rep} (compact again): exit {r.returncode}, wall {dt:.2f} s = {syn.n / dt / 1e9:.2f} Gbp/s end to end", flush=True)
    print(r.stdout.strip().split("\n")[-1], flush=True)
# the same text block-gzipped (BGZF, written by 16 threads) and, with "gz6" as second argument, as ONE gzip -6 member (zlib on
# one thread: ~75 s per GB): the program inflates them on the host threads (fast_inflate.h); same output files
extra = []
if len(sys.argv) > 2:
    import struct, zlib
    from concurrent.futures import ThreadPoolExecutor
    data = open(fa, "rb").read()
    def bgzf_block(a, B=65280):
        chunk = data[a:a + B] if a < len(data) else b""
        co = zlib.compressobj(6, zlib.DEFLATED, -15); body = co.compress(chunk) + co.flush()
        return (b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1)
                + body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    t0 = time.time()
    with open("/tmp/cli_in.bgzf.fa.gz", "wb") as f, ThreadPoolExecutor(16) as pool:
        for blk in pool.map(bgzf_block, list(range(0, len(data), 65280)) + [len(data)], chunksize=64): f.write(blk)
    extra.append("/tmp/cli_in.bgzf.fa.gz")
    print(f"wrote BGZF (level 6) {os.path.getsize(extra[-1]) / 1e9:.2f} GB in {time.time() - t0:.1f} s", flush=True)
    if "gz6" in sys.argv[2:]:
        t0 = time.time()
        with open("/tmp/cli_in.gzip6.fa.gz", "wb") as f:
            co = zlib.compressobj(6, zlib.DEFLATED, 31)
            for o in range(0, len(data), 1 << 24): f.write(co.compress(data[o:o + (1 << 24)]))
            f.write(co.flush())
        extra.append("/tmp/cli_in.gzip6.fa.gz")
        print(f"wrote one gzip -6 member {os.path.getsize(extra[-1]) / 1e9:.2f} GB in {time.time() - t0:.1f} s", flush=True)
    del data
    for z in extra:
        for rep in range(2):
            t0 = time.time()
            r = subprocess.run([os.path.join(ROOT, "cli", "deBWT"), "-o", "/tmp/cli_OUT_gz", "-t", "16", z], capture_output=True, text=True)
            dt = time.time() - t0
            print(f"{os.path.basename(z)} run {rep}: exit {r.returncode}, wall {dt:.2f} s = {syn.n / dt / 1e9:.2f} Gbp/s end to end", flush=True)
            print(r.stdout.strip().split("\n")[-1], flush=True)
            if r.returncode: print(r.stderr[-2000:]); sys.exit(1)
        same = hashlib.sha256(open("/tmp/cli_OUT_gz", "rb").read()).hexdigest() == sha_cli
        print(f"{os.path.basename(z)}: same OUT as from the plain file: {same}", flush=True)
        if not same: sys.exit(1)
    for p in extra + ["/tmp/cli_OUT_gz", "/tmp/cli_OUT_gz.#", "/tmp/cli_OUT_gz.$"]: os.remove(p)
import torch
from debwt_amd import api
text = SN.PinnedArray(syn.nwords)
syn.words_into(text.ptr)
d = api.DeBWT(k=32)
d.load_packed(text.a, syn.n, syn.sep())
d.build()
w, h, dr = d.fetch()
ok = hashlib.sha256(w.tobytes()).hexdigest() == sha_cli and np.array_equal(np.fromfile("/tmp/cli_OUT.#", dtype=np.uint64), h) \
    and int(np.fromfile("/tmp/cli_OUT.$", dtype=np.uint64)[0]) == dr
print("CLI output == API build of the same text:", ok, flush=True)
for p in (fa, "/tmp/cli_OUT", "/tmp/cli_OUT.#", "/tmp/cli_OUT.$"): os.remove(p)
sys.exit(0 if ok else 1)
