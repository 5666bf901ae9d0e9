"""Ingest rate of a FASTQ read set on the host threads (no GPU is used): python scripts/ingest_fastq_bench.py [million reads = 10] [read length = 100] [threads] [bgzf]
The reads are rewritten as header-less FASTA by all threads (fasta_host.cpp, fastq_to_fasta_parallel), then parsed and packed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from debwt_amd import api, synth_native as SN
args = [a for a in sys.argv[1:] if a != "bgzf"]
mreads = float(args[0]) if len(args) > 0 else 10
L = int(args[1]) if len(args) > 1 else 100
threads = int(args[2]) if len(args) > 2 else SN.default_threads()
n = int(mreads * 1e6)
rng = np.random.default_rng(1)
p = "/dev/shm/debwt_fqbench.fq"
with open(p, "wb") as f:
    for a in range(0, n, 200000):
        m = min(200000, n - a)
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(m, L), dtype=np.uint8)]
        qual = rng.integers(33, 74, size=(m, L), dtype=np.uint8)                # ('@' and '>' among the first characters of quality lines)
        rows = np.empty((m, 13 + L + 3 + L + 1), dtype=np.uint8)                 # "@r%010d\n" SEQ "\n+\n" QUAL "\n"
        names = np.char.mod("@r%010d\n", np.arange(a, a + m)).astype("S13")
        rows[:, :13] = np.frombuffer(names.tobytes(), dtype=np.uint8).reshape(m, 13)
        rows[:, 13:13 + L] = seq
        rows[:, 13 + L:16 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
        rows[:, 16 + L:16 + 2 * L] = qual
        rows[:, 16 + 2 * L] = 10
        f.write(rows.tobytes())
size = os.path.getsize(p)
print(f"{n} reads x {L} b: {size / 1e9:.2f} GB of FASTQ, {threads} host threads", flush=True)
ref = None
for label, env in (("one thread rewrites (the serial walk, as until round 6)", {"DEBWT_FASTQ_SERIAL": "1"}), ("all threads rewrite", {})):
    os.environ.pop("DEBWT_FASTQ_SERIAL", None); os.environ.update(env)
    best = None
    for _ in range(2):
        time.sleep(0.5)
        w, nn, sep, s_read, s_pack = api.pack_fasta(p, threads)
        best = min(best, s_read + s_pack) if best else s_read + s_pack
    if ref is None: ref = (w.copy(), nn)
    print(f"{label:58s} {best:6.3f} s = {size / 1e9 / best:6.2f} GB/s of FASTQ = {n * L / 1e9 / best:6.2f} Gbp/s, same text: {nn == ref[1] and np.array_equal(w, ref[0])}", flush=True)
if "bgzf" in sys.argv[1:]:                               # the same reads block-gzipped (level 6, written by all threads): inflate, rewrite, parse
    import struct, zlib
    from concurrent.futures import ThreadPoolExecutor
    data = open(p, "rb").read()
    def bgzf_block(a, B=65280):
        chunk = data[a:a + B] if a < len(data) else b""
        co = zlib.compressobj(6, zlib.DEFLATED, -15); body = co.compress(chunk) + co.flush()
        return (b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1)
                + body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    pz = p + ".gz"
    with open(pz, "wb") as f, ThreadPoolExecutor(threads) as pool:
        for blk in pool.map(bgzf_block, list(range(0, len(data), 65280)) + [len(data)], chunksize=64): f.write(blk)
    del data
    os.environ.pop("DEBWT_FASTQ_SERIAL", None)
    best = None
    for _ in range(2):
        time.sleep(0.5)
        w, nn, sep, s_read, s_pack = api.pack_fasta(pz, threads)
        best = min(best, (s_read + s_pack, s_read, s_pack)) if best else (s_read + s_pack, s_read, s_pack)
    print(f"{'the same reads as BGZF (' + format(os.path.getsize(pz) / 1e9, '.2f') + ' GB): inflate ' + format(best[1], '.3f') + ' s + rewrite, parse ' + format(best[2], '.3f') + ' s':58s} "
          f"{best[0]:6.3f} s = {size / 1e9 / best[0]:6.2f} GB/s of FASTQ = {n * L / 1e9 / best[0]:6.2f} Gbp/s, same text: {nn == ref[1] and np.array_equal(w, ref[0])}", flush=True)
    os.remove(pz)
os.environ["DEBWT_TRACE_INGEST"] = "1"                    # where the time goes, on stderr
time.sleep(0.5); api.pack_fasta(p, threads)
os.remove(p)
