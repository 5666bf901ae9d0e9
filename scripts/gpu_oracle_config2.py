"""Byte-for-byte comparison with the oracle at BASELINE configs[2] size (3.1 Gbp): the single-threaded CPU restatement of the
reference's algorithm (oracle/) builds the BWT of the whole collection on the GPU box's host (~120 GB, several minutes) and
the HIP build's rows, '#' rows and '$' row are compared with it word for word.  The tests keep to sizes the oracle finishes
in seconds (up to 250 Mbp); this is the one-off that raises the largest byte-for-byte comparison to 3.1 Gbp.
python scripts/gpu_oracle_config2.py [workload=grch38_3.1G] [k=32]"""
import os, sys, threading, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from debwt_amd import api, synth_native as SN
from oracle import oracle as O

wl = sys.argv[1] if len(sys.argv) > 1 else "grch38_3.1G"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 32
syn = SN.Synth.named(wl)
t0 = time.time()
words, census = syn.words()
sep = syn.sep()
d = api.DeBWT(k=k)
d.load_packed(words, syn.n, sep)
d.build(); d.build()
st = d.stats()
ver = d.verify_device()
w, h, dr = d.fetch()
d.close()
print(f"{wl}: n={syn.n} records={syn.nrec} k={k}: HIP build {st['ms_total']:.1f} ms, inverse BWT ok {ver['inverse_bwt_ok']}, "
      f"crc32(rows)={zlib.crc32(w.view(np.uint8)):08x}; {time.time()-t0:.0f} s so far", flush=True)
# the symbols of the collection, record by record (records = the chromosome-like cuts of every genome)
starts = np.concatenate([[0], sep[:-1] + 1]).astype(np.int64)
sym = np.empty(syn.n, dtype=np.uint8)
glen = int(syn.genome_len)
pos_in_genome = 0
g = 0
for r in range(syn.nrec):
    ln = int(sep[r]) - int(starts[r])
    sym[starts[r]:starts[r] + ln] = syn.codes(g, pos_in_genome, pos_in_genome + ln)
    sym[sep[r]] = 5 if r + 1 == syn.nrec else 4
    pos_in_genome += ln
    if pos_in_genome >= glen: g += 1; pos_in_genome = 0
print(f"symbols ready after {time.time()-t0:.0f} s; the oracle starts (single thread)", flush=True)
done = threading.Event()
def beat():
    while not done.wait(45): print(f"  ... oracle running, {time.time()-t0:.0f} s", flush=True)
threading.Thread(target=beat, daemon=True).start()
t1 = time.time()
ow, oh, od, ost = O.build_bwt(sym, k)
done.set()
print(f"oracle: {time.time()-t1:.0f} s = {syn.n / (time.time()-t1) / 1e9:.4f} Gbp/s; red {ost['red_capacity']} blue {ost['blue_capacity']} "
      f"blocks {ost['blue_bound_num']} S {ost['sp_len']}", flush=True)
same = np.array_equal(w, ow) and np.array_equal(h, oh) and dr == od
cnt = all(st[a] == ost[b] for a, b in (("red_capacity", "red_capacity"), ("blue_capacity", "blue_capacity"), ("blue_bound_num", "blue_bound_num"),
                                        ("sp_len", "sp_len"), ("case3num", "case3num")))
print(f"HIP == oracle, word for word ({len(w)} words, {len(h)} '#' rows, '$' row {dr}): {same}; counters equal: {cnt}", flush=True)
sys.exit(0 if same and cnt else 1)
