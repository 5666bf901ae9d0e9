#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own stage
functions (oracle/_ref/ref_driver, built by oracle/Makefile from /root/reference/src) with
`-t 1` (the reference's -t >= 2 path corrupts short SP segments).

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py

Every case is data: an input (a small FASTA, or the parameters of a formula-defined input from
debwt_amd.synth) and the reference's outputs for it -- OUT / OUT.# / OUT.$ bytes for small
cases, their sha256 for all, the stage counters the reference prints
(src/generateSP.c:28-31), the sha256 of its sorted edge file kmerInfo (src/mySort.c:193-195), and the
sha256 of the intermediates its stages hand to each other (ref_driver copies them out between the
stage calls): redSeq, redPoint, blueBound, case3bound (src/INandOut.c:347-366,396-417), the SP code
as one byte per symbol, the blue table with every block's entries sorted (src/generateSP.c:626-672).
The k-mer dump that stands in for Jellyfish's is checked HERE against naive counting (np.unique over the
windows of every record): the golden kmerInfo is therefore the reference's mySort applied to counts that
are right by definition, not merely to what the oracle's counter produced.  All inputs stay inside the reference's valid domain (all
four bases present, SP code well over 32 symbols, records > 32 bases).
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from debwt_amd import fasta, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402
import refformat as RF  # noqa: E402

SMALL = 40_000  # store output bytes below this many rows


def sha(b):
    return hashlib.sha256(b).hexdigest()


def handmade():
    rng = np.random.default_rng(20261003)
    u = lambda n: rng.integers(0, 4, size=n).astype(np.uint8)  # noqa: E731
    cases = {}
    # three short records with a shared repeat (the survey's "t1-like" case)
    rep = u(120)
    r0, r1, r2 = u(340), u(400), u(440)
    r0[100:220] = rep; r1[30:150] = rep; r2[300:420] = rep; r1[200:320] = rep
    cases["t1_three_records"] = [r0, r1, r2]
    # records sharing prefixes and suffixes across '#', duplicates, a record ending in T...T
    core = u(300)
    a = np.concatenate([core, u(80)])
    b = np.concatenate([core, u(60), np.full(45, 3, np.uint8)])
    c = np.concatenate([u(70), core[-200:]])
    d = a.copy()                                   # identical record
    e = a[:250].copy()                             # proper prefix of another record
    f = np.concatenate([u(50), a[-120:]])          # shares a suffix
    cases["shared_ends_duplicates"] = [a, b, c, d, e, f]
    # homopolymer runs and tandem repeats longer than k
    g = np.concatenate([u(200), np.zeros(90, np.uint8), u(150), np.full(70, 2, np.uint8), u(100),
                        np.tile(np.array([0, 1, 3], np.uint8), 40), u(120), np.zeros(90, np.uint8), u(60)])
    h = np.concatenate([u(90), np.full(64, 1, np.uint8), u(300), np.tile(np.array([2, 3], np.uint8), 50), u(77)])
    cases["homopolymers_tandem"] = [g, h]
    # special branches (src/collect#$.c:534-598): records with one shared tail, followed by records
    # that start differently, so K-windows across '#' are equal but continue differently
    tail = u(60)
    cases["special_branches"] = [np.concatenate([u(200), rep, tail]), np.concatenate([u(150), rep]),
                                 np.concatenate([u(180), tail]), np.concatenate([rep[:40], u(90)]),
                                 np.concatenate([u(100), rep, tail]), np.concatenate([u(140), tail])]
    # single record
    s = u(3000); s[2000:2400] = s[500:900]; s[2600:2900] = s[100:400]
    cases["single_record"] = [s]
    return cases


def run_ref(driver, recs, k, lower=False):
    d = tempfile.mkdtemp(prefix="golden_", dir="/tmp")
    try:
        fa = os.path.join(d, "in.fa")
        fasta.write_fasta(fa, recs, lower=lower)
        out = os.path.join(d, "OUT")
        p = subprocess.run([driver, d, fa, out, str(k), "1"], capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError(f"reference failed ({p.returncode}):\n{p.stdout[-1500:]}\n{p.stderr[-1500:]}")
        res = {ext: open(out + ext, "rb").read() for ext in ("", ".#", ".$", ".kmerInfo", ".redSeq", ".redPoint",
                                                              ".blueBound", ".case3bound", ".spCode", ".blueTable",
                                                              ".spSpecialIndex")}
        ctr = {a: int(b) for a, b in (ln.split() for ln in open(out + ".counters"))}
        # the stand-in for the Jellyfish dump against the definition of the counts
        km, ct = RF.parse_kmer_dump(open(out + ".kmerdump").read(), k)
        nk, nc = RF.naive_kmer_counts(recs, k)
        assert np.array_equal(km, nk) and np.array_equal(ct, nc), "k-mer dump differs from naive counting"
        info = np.frombuffer(res[".kmerInfo"], dtype=np.uint64).reshape(-1, 2)
        assert np.array_equal(info[:, 0], nk) and np.array_equal(info[:, 1], nc), "mySort output differs from naive counting"
        u64 = lambda b: np.frombuffer(b, dtype=np.uint64)  # noqa: E731
        res["spSymbols"] = RF.sp_symbols(u64(res[".spCode"]), ctr["spCodeLen"] - 32, u64(res[".spSpecialIndex"])).tobytes()
        res["blueBlocks"] = RF.blue_blocks_sorted(u64(res[".blueTable"]), u64(res[".blueBound"])).tobytes()
        return res, ctr
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    """No argument: regenerate everything.  `--only NAME[,NAME...]`: (re)generate those cases and merge them into the
    committed manifest (the other entries stay as they are)."""
    driver = O.build_ref()
    if not driver:
        sys.exit("needs /root/reference (build container only)")
    manifest = []
    only = set(sys.argv[2].split(",")) if len(sys.argv) > 2 and sys.argv[1] == "--only" else None

    def emit(name, recs, ks, source, lower=False):
        if only is not None and name not in only:
            return
        if callable(recs):
            recs = recs()
        n = sum(len(r) for r in recs) + len(recs)
        if source["kind"] == "fasta":
            fasta.write_fasta(os.path.join(HERE, name + ".fa"), recs, lower=lower)
        for k in ks:
            res, ctr = run_ref(driver, recs, k, lower)
            entry = {"name": name, "k": k, "n": n, "records": len(recs), "source": source, "counters": ctr,
                     "sha256": {"bwt": sha(res[""]), "hash": sha(res[".#"]), "dollar": sha(res[".$"]),
                                "kmerInfo": sha(res[".kmerInfo"]), "redSeq": sha(res[".redSeq"]),
                                "redPoint": sha(res[".redPoint"]), "blueBound": sha(res[".blueBound"]),
                                "case3bound": sha(res[".case3bound"]), "spSymbols": sha(res["spSymbols"]),
                                "blueBlocks": sha(res["blueBlocks"])}}
            if n <= SMALL:
                stem = os.path.join(HERE, f"{name}.k{k}")
                open(stem + ".bwt", "wb").write(res[""])
                open(stem + ".hash", "wb").write(res[".#"])
                open(stem + ".dollar", "wb").write(res[".$"])
                entry["files"] = True
            manifest.append(entry)
            print(name, k, n, ctr)

    for name, recs in handmade().items():
        emit(name, recs, (12, 16, 24, 32), {"kind": "fasta"})
    recs = synth.pan_genome(2500, 3)
    emit("lowercase_3x2500", recs, (32,), {"kind": "fasta", "lower": True}, lower=True)
    emit("pan_4x20k", synth.pan_genome(20000, 4), (12, 16, 32),
         {"kind": "synth", "fn": "pan_genome", "args": [20000, 4]})
    emit("pan_6x60k", synth.pan_genome(60000, 6), (32,),
         {"kind": "synth", "fn": "pan_genome", "args": [60000, 6]})
    emit("chrom_1M_5", synth.chromosomes(1_000_000, 5), (32, 20),
         {"kind": "synth", "fn": "chromosomes", "args": [1_000_000, 5]})
    emit("uniform_300k", [synth.uniform_codes(300_000)], (32,),
         {"kind": "synth", "fn": "uniform_codes", "args": [300_000], "wrap": True})
    # BASELINE.json configs[0]: the E. coli-sized single record bench.py calls ecoli_4.6M
    emit("ecoli_4.6M", lambda: synth.pan_chromosomes(4_600_000, 1, 1), (32,),
         {"kind": "synth", "fn": "pan_chromosomes", "args": [4_600_000, 1, 1]})
    # many records: N*K special suffixes well above the 2^14 at which the special-region module of the
    # build goes to host threads / the device; the reference's insert() is O(N) per record (src/INandOut.c:91-108),
    # hence N <= 2*10^4 here
    emit("contigs_2000", lambda: synth.read_set(2000, 3000, 8000, 4_000_000), (32,),
         {"kind": "synth", "fn": "read_set", "args": [2000, 3000, 8000, 4_000_000]})
    emit("reads_20000", lambda: synth.read_set(20000, 60, 400, 1_000_000), (32, 20),
         {"kind": "synth", "fn": "read_set", "args": [20000, 60, 400, 1_000_000]})
    if only is not None:
        old = json.load(open(os.path.join(HERE, "manifest.json")))
        new = {(e["name"], e["k"]) for e in manifest}
        manifest = [e for e in old if (e["name"], e["k"]) not in new] + manifest
    json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1)
    print("wrote", len(manifest), "entries")


if __name__ == "__main__":
    main()
