"""The C host program keeps the reference's command line (src/main.c:25-58) and output files
(src/insertCase3.c:115-131)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden_manifest, golden_outputs, golden_records

CLI = os.path.join(ROOT, "cli", "deBWT")


def _have_cli():
    if not os.path.exists(CLI):
        subprocess.call(["make", "-C", os.path.join(ROOT, "cli")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return os.path.exists(CLI)


def test_cli_argument_errors(tmp_path):
    if not _have_cli():
        pytest.skip("cli not built")
    r = subprocess.run([CLI], capture_output=True, text=True)
    assert r.returncode == 1 and "usage" in r.stderr
    fa = tmp_path / "x.fa"
    fa.write_text(">a\n" + "ACGT" * 20 + "\n")
    r = subprocess.run([CLI, "-o", str(tmp_path / "o"), "-k", "40", str(fa)], capture_output=True, text=True)
    assert r.returncode == 1 and "k-mer length" in r.stderr
    r = subprocess.run([CLI, "-o", str(tmp_path / "o"), "-t", "zero", str(fa)], capture_output=True, text=True)
    assert r.returncode == 1 and "thread number" in r.stderr
    r = subprocess.run([CLI, "-o", "/nonexistent_dir/o", str(fa)], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot create" in r.stderr
    # --devices: a list that is not numeric, ends in a comma, or names another number of GPUs than --gpus is refused,
    # never silently dropped
    for devs, gpus, msg in (("0,x", "2", "comma-separated"), ("zero", "1", "comma-separated"), ("0,", "1", "comma-separated"),
                            ("0,0,0", "2", "names 3 GPUs")):
        r = subprocess.run([CLI, "-o", str(tmp_path / "o"), "--gpus", gpus, "--devices", devs, str(fa)],
                           capture_output=True, text=True)
        assert r.returncode == 1 and msg in r.stderr, (devs, r.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("name,k", [("special_branches", 32), ("shared_ends_duplicates", 16), ("lowercase_3x2500", 32)])
def test_cli_outputs_equal_reference_files(tmp_path, name, k):
    assert _have_cli()
    entry = next(e for e in golden_manifest() if e["name"] == name and e["k"] == k)
    out = str(tmp_path / "OUT")
    fa = os.path.join(ROOT, "tests", "golden", name + ".fa")
    r = subprocess.run([CLI, "-o", out, "-k", str(k), "-t", "4", "-j", "/ignored", fa], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    words, hrows, drow = golden_outputs(entry)
    assert np.array_equal(np.fromfile(out, dtype=np.uint64), words)
    assert np.array_equal(np.fromfile(out + ".#", dtype=np.uint64), hrows)
    assert int(np.fromfile(out + ".$", dtype=np.uint64)[0]) == drow
    assert f"the redCapacity is {entry['counters']['redCapacity']}" in r.stdout


@pytest.mark.gpu
def test_cli_gzip_input(tmp_path):
    import gzip
    assert _have_cli()
    entry = next(e for e in golden_manifest() if e["name"] == "t1_three_records" and e["k"] == 32)
    src = os.path.join(ROOT, "tests", "golden", "t1_three_records.fa")
    gz = tmp_path / "in.fa.gz"
    with open(src, "rb") as f, gzip.open(gz, "wb") as g:
        g.write(f.read())
    out = str(tmp_path / "OUT")
    r = subprocess.run([CLI, "-o", out, str(gz)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(np.fromfile(out, dtype=np.uint64), golden_outputs(entry)[0])


@pytest.mark.gpu
def test_cli_fastq_input_equals_reference_files(tmp_path):
    """FASTQ, which the reference's reader takes too (src/kseq.h:177-201): the 20,000-read golden collection written as
    a gzipped FASTQ (qualities with '@' / '>' line starts) gives the reference's OUT, OUT.# and OUT.$ -- and, being 20,000
    records, goes through the special-region module on the device."""
    import gzip
    import hashlib
    from conftest import golden_records
    assert _have_cli()
    entry = next(e for e in golden_manifest() if e["name"] == "reads_20000" and e["k"] == 32)
    recs = golden_records(entry)
    fq = tmp_path / "reads.fq.gz"
    asc = np.frombuffer(b"ACGT", dtype=np.uint8)
    with gzip.open(fq, "wb", compresslevel=1) as g:
        for i, r in enumerate(recs):
            q = bytearray(b"I" * len(r))
            q[0] = b"@>+I"[i % 4]
            g.write(b"@r%d\n" % i + asc[r].tobytes() + b"\n+\n" + bytes(q) + b"\n")
    out = str(tmp_path / "OUT")
    r = subprocess.run([CLI, "-o", out, "-t", "4", str(fq)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for ext, key in (("", "bwt"), (".#", "hash"), (".$", "dollar")):
        assert hashlib.sha256(open(out + ext, "rb").read()).hexdigest() == entry["sha256"][key], ext


@pytest.mark.gpu
def test_cli_iupac_option(tmp_path):
    """A FASTA with N runs: refused as the reference's reader refuses it, accepted with --iupac, and then the BWT is the
    BWT of the text the ingest produced (same seed through the API)."""
    from debwt_amd import api
    assert _have_cli()
    rng = np.random.default_rng(8)
    s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=20000)].copy()
    s[3000:3400] = ord("N"); s[9000:9007] = ord("r"); s[15000] = ord("Y")
    fa = tmp_path / "n.fa"
    with open(fa, "wb") as f:
        f.write(b">withN\n")
        for a in range(0, len(s), 70):
            f.write(s[a:a + 70].tobytes() + b"\n")
    out = str(tmp_path / "OUT")
    r = subprocess.run([CLI, "-o", out, str(fa)], capture_output=True, text=True)
    assert r.returncode == 1 and "not one of ACGTacgt" in r.stderr
    r = subprocess.run([CLI, "-o", out, "--iupac", "12345", str(fa)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = api.DeBWT(k=32)
    d.load_fasta(str(fa), threads=3, iupac_seed=12345)
    d.build()
    words, hrows, drow = d.fetch()
    d.close()
    assert np.array_equal(np.fromfile(out, dtype=np.uint64), words)
    assert int(np.fromfile(out + ".$", dtype=np.uint64)[0]) == drow


@pytest.mark.gpu
@pytest.mark.parametrize("name,k,gpus,keys", [("special_branches", 32, 2, "exchange"), ("shared_ends_duplicates", 16, 3, "auto"),
                                              ("pan_fa", 32, 4, "rescan"), ("pan_fa", 32, 2, "exchange")])
def test_cli_multi_gpu_outputs_equal_reference_files(tmp_path, name, k, gpus, keys):
    """deBWT --gpus G: the C host over G shards (all on GPU 0 of the test box: --devices 0,0,...) writes the same three
    files the reference writes."""
    assert _have_cli()
    out = str(tmp_path / "OUT")
    if name == "pan_fa":                                   # a formula-defined golden input, written as FASTA here
        from debwt_amd import fasta
        entry = next(e for e in golden_manifest() if e["name"] == "pan_4x20k" and e["k"] == k)
        fa = str(tmp_path / "pan.fa")
        fasta.write_fasta(fa, golden_records(entry))
    else:
        entry = next(e for e in golden_manifest() if e["name"] == name and e["k"] == k)
        fa = os.path.join(ROOT, "tests", "golden", name + ".fa")
    r = subprocess.run([CLI, "-o", out, "-k", str(k), "-t", "4", "--gpus", str(gpus), "--devices", ",".join(["0"] * gpus), "--keys", keys, fa],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    import hashlib
    sha = lambda p: hashlib.sha256(open(p, "rb").read()).hexdigest()      # noqa: E731
    assert sha(out) == entry["sha256"]["bwt"] and sha(out + ".#") == entry["sha256"]["hash"]
    assert sha(out + ".$") == entry["sha256"]["dollar"]
    assert f"{gpus} GPUs, exchanges by peer-to-peer copies, keys {'exchanged' if keys == 'exchange' else 'rescanned'}" in r.stdout


FILE_GOLDENS = [e for e in golden_manifest() if e["n"] < 400000 and e["k"] in (12, 32)]


@pytest.mark.gpu
@pytest.mark.parametrize("entry", FILE_GOLDENS, ids=lambda e: f"{e['name']}-k{e['k']}")
def test_cli_dump_writes_the_reference_files(tmp_path, entry):
    """SURVEY 8f-4 as the row is worded: `deBWT --dump DIR` leaves kmerInfo, redSeq, redPoint, blueBound and case3bound
    byte-identical to the files the reference's own mySort / generateBlocks wrote (sha256 in tests/golden/manifest.json;
    formats src/mySort.c:193-195, src/INandOut.c:347-366,396-417), spCode + spSpecialIndex that decode to the reference's
    SP symbols, and a blueTable holding the reference's entries block by block (src/generateSP.c:626-672; the order inside
    a block is the scan's arrival order in the reference too) -- and the build driven stage by stage still writes the
    reference's OUT, OUT.#, OUT.$.  --verify beside it: one summary line, exit status 0."""
    import hashlib
    import refformat as RF
    assert _have_cli()
    fa = os.path.join(ROOT, "tests", "golden", entry["name"] + ".fa")
    if not os.path.exists(fa):
        from debwt_amd import fasta
        fa = str(tmp_path / "in.fa")
        fasta.write_fasta(fa, golden_records(entry))
    out, dump = str(tmp_path / "OUT"), tmp_path / "dump"
    dump.mkdir()
    r = subprocess.run([CLI, "-o", out, "-k", str(entry["k"]), "--dump", str(dump), "--verify", fa], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "verify: inverse BWT ok" in r.stdout and f"{entry['n'] - 1} LF steps" in r.stdout and " 0 mismatches" in r.stdout
    sha = entry["sha256"]
    for name in ("kmerInfo", "redSeq", "redPoint", "blueBound", "case3bound"):
        assert hashlib.sha256((dump / name).read_bytes()).hexdigest() == sha[name], name
    u64 = lambda name: np.fromfile(dump / name, dtype=np.uint64)
    sp_len = entry["counters"]["spCodeLen"] - 32
    assert len(u64("spCode")) == (sp_len + 31) // 32 and len(u64("spSpecialIndex")) == entry["records"]
    assert hashlib.sha256(RF.sp_symbols(u64("spCode"), sp_len, u64("spSpecialIndex")).tobytes()).hexdigest() == sha["spSymbols"]
    assert hashlib.sha256(RF.blue_blocks_sorted(u64("blueTable"), u64("blueBound")).tobytes()).hexdigest() == sha["blueBlocks"]
    for ext, key in (("", "bwt"), (".#", "hash"), (".$", "dollar")):
        assert hashlib.sha256(open(out + ext, "rb").read()).hexdigest() == sha[key], ext


@pytest.mark.gpu
def test_cli_verify_rejects_a_result_that_is_not_the_bwt(tmp_path):
    """`deBWT --verify` is the reference's LF walk (src/LFsearch.c:14-48) as a tool: exit status 0 and 'ok' on a build; the
    same walk through the C ABI rejects the rows once two of them are swapped (the CLI verifies the context's own rows, so
    the rejection is shown on the ABI: tests/test_gpu_verify.py has the full set)."""
    assert _have_cli()
    fa = os.path.join(ROOT, "tests", "golden", "special_branches.fa")
    out = str(tmp_path / "OUT")
    r = subprocess.run([CLI, "--verify", "-o", out, fa], capture_output=True, text=True)
    assert r.returncode == 0 and "verify: inverse BWT ok" in r.stdout and "0 mismatches" in r.stdout, r.stderr
    r = subprocess.run([CLI, "-o", out, "--gpus", "2", "--devices", "0,0", "--verify", fa], capture_output=True, text=True)
    assert r.returncode == 0 and "verify: inverse BWT ok" in r.stdout, r.stderr
