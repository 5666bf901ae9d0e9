"""Host FASTA ingest (debwt_pack_fasta): the multi-threaded parser against the plain numpy packer on
the same records -- layout of src/collect#$.c:61-90 -- for every thread count, line shape and error the reference's
reader distinguishes (src/collect#$.c:34-45, src/main.c:18-23).  No GPU needed."""
import gzip
import os

import numpy as np
import pytest

from debwt_amd import api

ASC = b"ACGT"


def _write(path, recs, width=60, lower=False, crlf=False, blank_lines=False, spaces=False, gz=False):
    eol = b"\r\n" if crlf else b"\n"
    op = gzip.open if gz else open
    with op(path, "wb") as f:
        for i, r in enumerate(recs):
            f.write(b">rec%d some > description with >signs" % i + eol)
            s = bytes(ASC[c] for c in r)
            if lower and i % 2:
                s = s.lower()
            for a in range(0, len(s), width):
                line = s[a:a + width]
                if spaces and len(line) > 10:
                    line = line[:5] + b" " + line[5:9] + b"\t" + line[9:]
                f.write(line + eol)
                if blank_lines and a % (3 * width) == 0:
                    f.write(eol)


def _check(path, recs, threads):
    w0, n0, sep0 = api.pack_records(recs)
    w, n, sep, _, _ = api.pack_fasta(path, threads)
    nw = (n0 + 63) // 32
    assert n == n0 and np.array_equal(sep, sep0)
    assert np.array_equal(w[:nw], w0[:nw])


@pytest.fixture(scope="module")
def recs():
    rng = np.random.default_rng(3)
    return [rng.integers(0, 4, size=int(rng.integers(33, 40000))).astype(np.uint8) for _ in range(41)]


@pytest.mark.parametrize("threads", [1, 2, 3, 8, 64])
@pytest.mark.parametrize("shape", ["plain", "lower", "crlf", "blank", "spaces", "wide", "narrow", "gz"])
def test_pack_fasta_matches_numpy_packer(tmp_path, recs, threads, shape, monkeypatch):
    # the packed words are not cleared as a whole (fasta_host.cpp: only the words the packers OR into): start from 0xA5 bytes
    monkeypatch.setenv("DEBWT_INGEST_POISON", "1")
    p = str(tmp_path / ("t.fa.gz" if shape == "gz" else "t.fa"))
    kw = {"plain": {}, "lower": {"lower": True}, "crlf": {"crlf": True}, "blank": {"blank_lines": True},
          "spaces": {"spaces": True}, "wide": {"width": 100000}, "narrow": {"width": 1}, "gz": {"gz": True}}[shape]
    _write(p, recs, **kw)
    _check(p, recs, threads)


def test_pack_fasta_single_long_line_and_many_threads(tmp_path):
    rng = np.random.default_rng(5)
    recs = [rng.integers(0, 4, size=3_000_000).astype(np.uint8), rng.integers(0, 4, size=33).astype(np.uint8)]
    p = str(tmp_path / "l.fa")
    _write(p, recs, width=10_000_000)
    for t in (1, 7, 16):
        _check(p, recs, t)


def test_pack_fasta_headers_longer_than_a_chunk(tmp_path):
    """Chunk starts that fall inside header lines."""
    rng = np.random.default_rng(6)
    recs = [rng.integers(0, 4, size=40).astype(np.uint8) for _ in range(30)]
    p = str(tmp_path / "h.fa")
    with open(p, "wb") as f:
        for i, r in enumerate(recs):
            f.write(b">" + b"x>y" * 20000 + b"\n" + bytes(ASC[c] for c in r) + b"\n")
    for t in (1, 5, 16):
        _check(p, recs, t)


@pytest.mark.parametrize("content,msg", [
    (b">a\nACGTNACGT" + b"A" * 40 + b"\n", "not one of ACGTacgt"),
    (b"ACGT\n>a\n" + b"A" * 40 + b"\n", "before the first header"),
    (b">a\n" + b"A" * 32 + b"\n", "Length <= 32"),
    (b"@r\nACGT\n+\nIIII\n", "Length <= 32"),                                       # FASTQ is read; 4 bases are too few
    (b"@r\n" + b"A" * 40 + b"\n+\n" + b"I" * 39 + b"\n", "quality characters"),      # truncated quality string
    (b"@r\n" + b"A" * 30 + b"\n", "Length <= 32"),                                    # an '@' record without qualities is a record
    (b"@r\n" + b"A" * 40 + b"\n+\n" + b"I" * 40 + b"\nACGT\n", "does not start with '@'"),
    (b"", "empty"),
])
def test_pack_fasta_errors(tmp_path, content, msg):
    p = str(tmp_path / "e.fa")
    open(p, "wb").write(content)
    with pytest.raises(api.DebwtError) as ei:
        api.pack_fasta(p, 4)
    assert msg in str(ei.value)


@pytest.mark.parametrize("where", ["first", "middle", "last", "at_a_cut"])
def test_pack_fasta_short_record_among_many_chunks(tmp_path, where):
    """src/collect#$.c:41-45: a record of 32 bases or fewer is an error wherever it lies -- the chunks' censuses report their
    shortest complete record, and the one record that is open across a chunk end is followed by the serial combine."""
    rng = np.random.default_rng(77)
    recs = [rng.integers(0, 4, size=int(rng.integers(2000, 9000))).astype(np.uint8) for _ in range(200)]   # ~1.1 MB: 8 chunks and more
    p = str(tmp_path / "short.fa")
    _write(p, recs, width=60)
    size = os.path.getsize(p)
    _check(p, recs, 8)
    k = {"first": 0, "middle": 100, "last": 199}.get(where)
    if k is None:                                          # the record that lies across the cut between chunks 3 and 4 of 8
        data = open(p, "rb").read()
        k = data[:size // 8 * 4].count(b"\n>")             # (headers before the cut, less the first, which no newline precedes)
    recs2 = list(recs)
    recs2[k] = recs[k][:32]
    _write(p, recs2, width=60)
    for threads in (1, 8):
        with pytest.raises(api.DebwtError, match="Length <= 32!"):
            api.pack_fasta(p, threads)
    recs2[k] = recs[k][:33]
    _write(p, recs2, width=60)
    _check(p, recs2, 8)


def test_pack_fasta_missing_file():
    with pytest.raises(api.DebwtError):
        api.pack_fasta("/nonexistent/x.fa", 2)


def test_golden_fasta_files_pack_like_the_python_reader():
    from conftest import GOLDEN, golden_manifest
    from debwt_amd import fasta
    for e in golden_manifest():
        if e["source"]["kind"] != "fasta":
            continue
        path = os.path.join(GOLDEN, e["name"] + ".fa")
        _check(path, fasta.read_fasta(path)[1], 4)


def test_pack_fasta_no_trailing_newline_and_empty_lines_at_end(tmp_path, recs):
    p = str(tmp_path / "n.fa")
    with open(p, "wb") as f:
        for i, r in enumerate(recs[:5]):
            f.write(b">r%d\n" % i + bytes(ASC[c] for c in r) + (b"" if i == 4 else b"\n"))
    _check(p, recs[:5], 3)
    with open(p, "ab") as f:
        f.write(b"\n\n\n")
    _check(p, recs[:5], 3)


def test_pack_fasta_empty_record_is_an_error(tmp_path):
    p = str(tmp_path / "e.fa")
    open(p, "wb").write(b">a\n" + b"ACGT" * 20 + b"\n>empty\n>b\n" + b"ACGT" * 20 + b"\n")
    with pytest.raises(api.DebwtError) as ei:
        api.pack_fasta(p, 2)
    assert "Length <= 32" in str(ei.value)


IUPAC_SETS = {"N": "ACGT", "V": "ACG", "D": "ATG", "B": "TCG", "H": "ATC", "W": "AT", "S": "CG", "K": "TG", "M": "AC",
              "Y": "CT", "R": "AG"}           # otherTool/transferN.c:8-9,17-27


def _unpack(words, n):
    w = np.asarray(words, dtype=np.uint64)
    sh = np.uint64(62) - np.uint64(2) * (np.arange(32, dtype=np.uint64))
    return ((w[:, None] >> sh[None, :]) & np.uint64(3)).astype(np.uint8).reshape(-1)[:n]


@pytest.mark.parametrize("width", [60, 7, 200])
def test_pack_fasta_iupac_replacement(tmp_path, width):
    """transferN's job inside the ingest: every ambiguity letter becomes a base of its set, all other characters are
    untouched, the result depends on the seed and the text position only (not on the thread count), and without the
    option the same file is still an error."""
    rng = np.random.default_rng(11)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    texts = []
    for r in range(5):
        s = letters[rng.integers(0, 4, size=int(rng.integers(200, 30000)))].copy()
        for _ in range(int(rng.integers(1, 30))):
            a = int(rng.integers(0, len(s) - 1)); b = min(len(s), a + int(rng.integers(1, 400)))
            amb = list(IUPAC_SETS)[int(rng.integers(0, len(IUPAC_SETS)))]
            s[a:b] = ord(amb.lower() if rng.integers(0, 2) else amb)
        texts.append(s.tobytes())
    p = str(tmp_path / "amb.fa")
    with open(p, "wb") as f:
        for i, t in enumerate(texts):
            f.write(b">r%d\n" % i)
            for a in range(0, len(t), width):
                f.write(t[a:a + width] + b"\n")
    with pytest.raises(api.DebwtError, match="not one of ACGTacgt"):
        api.pack_fasta(p, 3)
    w1, n1, sep1, _, _ = api.pack_fasta(p, 1, iupac_seed=5)
    for threads in (2, 8, 64):
        w, n, sep, _, _ = api.pack_fasta(p, threads, iupac_seed=5)
        assert n == n1 and np.array_equal(sep, sep1) and np.array_equal(w, w1)
    assert n1 == sum(len(t) for t in texts) + len(texts)
    sym = _unpack(w1, n1)
    pos = 0
    for t in texts:
        got = sym[pos:pos + len(t)]
        src = np.frombuffer(t.upper(), dtype=np.uint8)
        for ch, code in zip(b"ACGT", range(4)):
            assert (got[src == ch] == code).all()
        for amb, allowed in IUPAC_SETS.items():
            m = src == ord(amb)
            if m.any():
                ok = np.isin(got[m], [b"ACGT".index(c) for c in allowed.encode()])
                assert ok.all()
                if m.sum() > 200:
                    assert len(np.unique(got[m])) == len(allowed)           # every member of the set is drawn
        pos += len(t)
        assert sym[pos] == 3                                                 # separator
        pos += 1
    w2, _, _, _, _ = api.pack_fasta(p, 4, iupac_seed=6)
    assert not np.array_equal(w2, w1)


@pytest.mark.parametrize("width", [1, 2, 5, 16, 30, 31, 32, 33, 63, 64, 65])
@pytest.mark.parametrize("novec", [False, True])
def test_pack_fasta_line_widths_around_the_vector_block(tmp_path, recs, width, novec, monkeypatch):
    """Lines shorter than, equal to and just above the 32-character block of the vectorised classifier (several line
    breaks per block, a break at the block's last byte, a header right behind it), with and without AVX2."""
    if novec:
        monkeypatch.setenv("DEBWT_INGEST_NO_AVX2", "1")
    sub = recs[:7]
    p = str(tmp_path / "w.fa")
    _write(p, sub, width=width)
    for threads in (1, 5):
        _check(p, sub, threads)


def _write_fastq(path, recs, width=0, crlf=False, gz=False, lower=False, nasty_quals=True, blank=False):
    """FASTQ as the reference's reader takes it (src/kseq.h:177-201): '@' header, sequence on one or several lines, a
    '+' line (with or without the name repeated), as many quality characters as bases -- quality lines may start with
    '@' or '>' (both are legal quality characters)."""
    eol = b"\r\n" if crlf else b"\n"
    op = gzip.open if gz else open
    rng = np.random.default_rng(9)
    with op(path, "wb") as f:
        for i, r in enumerate(recs):
            f.write(b"@read%d/1 length=%d" % (i, len(r)) + eol)
            s = bytes(ASC[c] for c in r)
            if lower and i % 2:
                s = s.lower()
            w = width or len(s)
            for a in range(0, len(s), w):
                f.write(s[a:a + w] + eol)
            f.write((b"+read%d/1" % i if i % 2 else b"+") + eol)
            q = bytes(rng.integers(33, 74, size=len(s)).astype(np.uint8))
            if nasty_quals:
                q = (b"@" if i % 3 == 0 else b">" if i % 3 == 1 else b"+") + q[1:]
            for a in range(0, len(q), w):
                f.write(q[a:a + w] + eol)
            if blank and i % 5 == 0:
                f.write(eol)


@pytest.mark.parametrize("threads", [1, 4, 16])
@pytest.mark.parametrize("shape", ["plain", "multiline", "crlf", "gz", "lower", "blank"])
def test_pack_fastq_matches_numpy_packer(tmp_path, threads, shape):
    """FASTQ input (reads: the collections the special-region module on the device exists for): same packed text as the
    numpy packer gives for the sequences, qualities dropped."""
    rng = np.random.default_rng(12)
    recs = [rng.integers(0, 4, size=int(rng.integers(33, 400))).astype(np.uint8) for _ in range(3000)]
    p = str(tmp_path / ("r.fq.gz" if shape == "gz" else "r.fq"))
    kw = {"plain": {}, "multiline": {"width": 37}, "crlf": {"crlf": True, "width": 50}, "gz": {"gz": True},
          "lower": {"lower": True}, "blank": {"blank": True}}[shape]
    _write_fastq(p, recs, **kw)
    _check(p, recs, threads)


@pytest.mark.parametrize("shape", ["plain", "multiline", "crlf", "blank", "mixed"])
@pytest.mark.parametrize("threads,chunk_min", [(2, 64), (5, 500), (16, 4096)])
def test_pack_fastq_by_all_threads(tmp_path, monkeypatch, capfd, shape, threads, chunk_min):
    """The FASTQ rewrite by all threads (fasta_host.cpp, fastq_to_fasta_parallel): every thread but the first GUESSES a record
    start behind its cut of the file (a quality line may start with '@': it cannot be recognised), walks like the serial walk,
    and must end exactly on the next thread's guess -- by induction from byte 0 the guesses are then record starts of the one
    serial walk.  Chunks down to 64 bytes (DEBWT_FASTQ_CHUNK_MIN: cuts inside headers, qualities that start with '@', '>'
    and '+', '+' lines that repeat the name), records without qualities and FASTA records mixed in (no guess in their
    chunks): same packed text as the serial walk and as the numpy packer; a malformed record gives the serial walk's message."""
    rng = np.random.default_rng(40 + threads)
    recs = [rng.integers(0, 4, size=int(rng.integers(33, 400))).astype(np.uint8) for _ in range(700)]
    p = str(tmp_path / "par.fq")
    if shape == "mixed":
        with open(p, "wb") as f:
            for i, r in enumerate(recs):
                s = bytes(ASC[c] for c in r)
                kind = i % 4 if i + 1 < len(recs) else 1
                f.write((b">" if kind == 2 else b"@") + b"rec%d\n" % i)
                for a in range(0, len(s), 61):
                    f.write(s[a:a + 61] + b"\n")
                if kind in (0, 3):
                    f.write(b"+\n" + (b"@" if kind == 0 else b">") + b"I" * (len(s) - 1) + b"\n")
    else:
        kw = {"plain": {}, "multiline": {"width": 37}, "crlf": {"crlf": True, "width": 50}, "blank": {"blank": True}}[shape]
        _write_fastq(p, recs, **kw)
    monkeypatch.setenv("DEBWT_FASTQ_CHUNK_MIN", str(chunk_min))
    monkeypatch.setenv("DEBWT_FASTQ_REQUIRE_PARALLEL", "1")          # (says on stderr how many threads took part)
    capfd.readouterr()
    _check(p, recs, threads)
    assert "fastq: %d threads" % threads in capfd.readouterr().err
    w, n, sep, _, _ = api.pack_fasta(p, threads)
    monkeypatch.setenv("DEBWT_FASTQ_SERIAL", "1")
    w1, n1, sep1, _, _ = api.pack_fasta(p, threads)
    assert n == n1 and np.array_equal(sep, sep1) and np.array_equal(w, w1)
    monkeypatch.delenv("DEBWT_FASTQ_SERIAL")
    # one quality character short in the middle: the message of the serial walk, with its record number
    data = open(p, "rb").read()
    if shape == "plain":
        lines = data.split(b"\n")
        k = 4 * 350 + 3                                              # quality line of record 351
        lines[k] = lines[k][:-1]
        bad = str(tmp_path / "bad.fq")
        open(bad, "wb").write(b"\n".join(lines))
        with pytest.raises(api.DebwtError, match="FASTQ record 351: "):
            api.pack_fasta(bad, threads)


def test_pack_fastq_records_without_qualities_and_fasta_records_mixed_in(tmp_path):
    """kseq_read ends a sequence at a line that starts with '+', '@' or '>' (src/kseq.h:188): an '@' record without a
    quality section and '>' records inside a FASTQ file are records like any other; the last record may end the file
    without qualities."""
    rng = np.random.default_rng(14)
    recs = [rng.integers(0, 4, size=int(rng.integers(40, 200))).astype(np.uint8) for _ in range(60)]
    p = str(tmp_path / "mixed.fq")
    with open(p, "wb") as f:
        for i, r in enumerate(recs):
            s = bytes(ASC[c] for c in r)
            kind = i % 4 if i + 1 < len(recs) else 1
            f.write((b">" if kind == 2 else b"@") + b"rec%d\n" % i)
            for a in range(0, len(s), 61):
                f.write(s[a:a + 61] + b"\n")
            if kind in (0, 3):                           # with qualities (first character legal but nasty)
                f.write(b"+\n" + (b"@" if kind == 0 else b">") + b"I" * (len(s) - 1) + b"\n")
    for threads in (1, 7):
        _check(p, recs, threads)


def _write_bgzf(path, data, block=65280):
    """Block gzip as bgzip writes it: one gzip member per block, each with the 'BC' extra subfield that holds its length,
    an empty member at the end."""
    import struct
    import zlib
    with open(path, "wb") as f:
        for a in list(range(0, len(data), block)) + [len(data)]:
            chunk = data[a:a + block] if a < len(data) else b""
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            body = co.compress(chunk) + co.flush()
            bsize = 12 + 6 + len(body) + 8
            f.write(b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1))
            f.write(body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))


@pytest.mark.parametrize("threads", [1, 3, 16])
def test_pack_block_gzip_in_parallel(tmp_path, threads):
    """BGZF input: the members are found by their headers and inflated by the ingest threads in parallel -- the same packed
    text as from the plain file; python's gzip module reads the file too (it is ordinary multi-member gzip); a member with
    a wrong CRC is an error."""
    rng = np.random.default_rng(15)
    recs = [rng.integers(0, 4, size=int(rng.integers(40, 90_000))).astype(np.uint8) for _ in range(40)]
    plain = str(tmp_path / "b.fa")
    _write(plain, recs, width=70)
    data = open(plain, "rb").read()
    p = str(tmp_path / "b.fa.gz")
    _write_bgzf(p, data)
    assert gzip.open(p, "rb").read() == data
    _check(p, recs, threads)
    raw = bytearray(open(p, "rb").read())
    raw[len(raw) // 2] ^= 0x55                                   # damage inside a member
    bad = str(tmp_path / "bad.fa.gz")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(api.DebwtError):
        api.pack_fasta(bad, threads)


def test_pack_fastq_with_ambiguity_letters(tmp_path):
    rng = np.random.default_rng(13)
    recs = [rng.integers(0, 4, size=150).astype(np.uint8) for _ in range(50)]
    p = str(tmp_path / "n.fq")
    _write_fastq(p, recs)
    data = bytearray(open(p, "rb").read())
    pos = data.index(b"\n") + 10                       # an N inside the first read
    data[pos] = ord("N")
    open(p, "wb").write(bytes(data))
    with pytest.raises(api.DebwtError):
        api.pack_fasta(p, 2)
    a = api.pack_fasta(p, 1, iupac_seed=5)
    b = api.pack_fasta(p, 8, iupac_seed=5)
    assert a[1] == b[1] and np.array_equal(a[0], b[0])


def _gz_members(parts, level=6):
    """Plain gzip members (no BGZF subfield), concatenated."""
    return b"".join(gzip.compress(p, compresslevel=level, mtime=0) for p in parts)


@pytest.mark.parametrize("level", [1, 6, 9])
@pytest.mark.parametrize("piece", [65536, 300000])
def test_pack_one_member_gzip_in_parallel(tmp_path, monkeypatch, level, piece):
    """A one-member gzip file (how genome FASTA is distributed) is inflated by all ingest threads: the compressed bytes are
    cut into pieces, a deflate block start is found in every piece by trial, pieces are decoded with an unknown 32 KB window
    (markers) until zlib can take over, markers are resolved in order, CRC32 and length are checked against the trailer
    (gz_parallel.cpp).  DEBWT_GZ_REQUIRE_PARALLEL: no silent fall-back to the serial path.  Same packed text as from the plain
    file, for compression levels 1 (markers never clear: our decoder alone), 6 and 9 and for pieces of 64 KB and 300 KB (the
    default, two pieces per thread and at least 1 MB each: test_pack_one_member_gzip_default_pieces)."""
    rng = np.random.default_rng(21 + level)
    recs = [rng.integers(0, 4, size=int(rng.integers(200_000, 1_500_000))).astype(np.uint8) for _ in range(5)]
    plain = str(tmp_path / "g.fa")
    _write(plain, recs, width=80)
    data = open(plain, "rb").read()
    p = str(tmp_path / "g.fa.gz")
    open(p, "wb").write(gzip.compress(data, compresslevel=level, mtime=0))
    monkeypatch.setenv("DEBWT_GZ_REQUIRE_PARALLEL", "1")
    monkeypatch.setenv("DEBWT_GZ_PIECE_BYTES", str(piece))
    for threads in (2, 8):
        _check(p, recs, threads)


def test_pack_one_member_gzip_default_pieces(tmp_path, monkeypatch):
    """40 MB of FASTA in one gzip member with the default cut (two pieces per thread, at least 1 MB of compressed bytes each):
    the same words as from the plain file."""
    rng = np.random.default_rng(77)
    codes = rng.integers(0, 4, size=40_000_000).astype(np.uint8)
    lines = np.frombuffer(b"ACGT", dtype=np.uint8)[codes].reshape(-1, 80)
    data = b">one record\n" + np.concatenate([lines, np.full((lines.shape[0], 1), 10, np.uint8)], axis=1).tobytes()
    plain, p = str(tmp_path / "d.fa"), str(tmp_path / "d.fa.gz")
    open(plain, "wb").write(data)
    open(p, "wb").write(gzip.compress(data, compresslevel=6, mtime=0))
    w0, n0, sep0, _, _ = api.pack_fasta(plain, 8)
    monkeypatch.setenv("DEBWT_GZ_REQUIRE_PARALLEL", "1")
    for threads in (3, 8):
        w, n, sep, _, _ = api.pack_fasta(p, threads)
        assert n == n0 == 40_000_001 and np.array_equal(sep, sep0) and np.array_equal(w, w0)


def test_pack_gzip_shapes_the_parallel_path_declines(tmp_path, monkeypatch):
    """What the parallel paths do not take is inflated serially as before, with the same result: stored blocks only (gzip -0:
    no dynamic block to start from), a file too small to cut (several plain members are taken since round 6:
    test_pack_several_plain_gzip_members_in_parallel); a FASTQ .gz of one member IS taken; a stream damaged in the middle is an error, not a wrong text."""
    rng = np.random.default_rng(31)
    recs = [rng.integers(0, 4, size=int(rng.integers(100_000, 400_000))).astype(np.uint8) for _ in range(6)]
    plain = str(tmp_path / "m.fa")
    _write(plain, recs, width=61)
    data = open(plain, "rb").read()
    monkeypatch.setenv("DEBWT_GZ_PIECE_BYTES", "65536")
    cut = data.index(b">", len(data) // 2)
    shapes = {"stored": gzip.compress(data, compresslevel=0, mtime=0),
              "tiny": gzip.compress(data[:data.index(b"\n>", 1) + 1], mtime=0)}
    for name, blob in shapes.items():
        p = str(tmp_path / (name + ".fa.gz"))
        open(p, "wb").write(blob)
        monkeypatch.setenv("DEBWT_GZ_REQUIRE_PARALLEL", "1")
        with pytest.raises(api.DebwtError, match="declined"):
            api.pack_fasta(p, 4)
        monkeypatch.delenv("DEBWT_GZ_REQUIRE_PARALLEL")
        _check(p, recs if name != "tiny" else recs[:1], 4)
    # FASTQ, one member
    reads = [rng.integers(0, 4, size=150).astype(np.uint8) for _ in range(20000)]
    fq = str(tmp_path / "r.fq")
    _write_fastq(fq, reads)
    fqz = str(tmp_path / "r.fq.gz")
    open(fqz, "wb").write(gzip.compress(open(fq, "rb").read(), mtime=0))
    monkeypatch.setenv("DEBWT_GZ_REQUIRE_PARALLEL", "1")
    _check(fqz, reads, 8)
    monkeypatch.delenv("DEBWT_GZ_REQUIRE_PARALLEL")
    # damage in the middle of the one-member stream: declined by the parallel path (its pieces no longer meet, or the CRC
    # differs), reported by the serial one
    raw = bytearray(gzip.compress(data, mtime=0))
    raw[len(raw) // 2] ^= 0x5A
    bad = str(tmp_path / "bad.fa.gz")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(api.DebwtError):
        api.pack_fasta(bad, 4)
    # ... and cut short
    open(bad, "wb").write(gzip.compress(data, mtime=0)[:-40000])
    with pytest.raises(api.DebwtError):
        api.pack_fasta(bad, 4)


@pytest.mark.parametrize("way", ["into_place", "general"])
@pytest.mark.parametrize("members,threads", [(2, 8), (3, 2), (7, 3), (40, 4)])
def test_pack_several_plain_gzip_members_in_parallel(tmp_path, monkeypatch, members, threads, way):
    """`cat a.fa.gz b.fa.gz ...` (one member per chromosome or genome; the reference reads it through gzread,
    src/collect#$.c:26,34-90): the member headers are found by their fixed bytes, every member is inflated on its own -- many
    members by as many threads, few large ones one after the other in pieces -- checked against its CRC32 and ISIZE, and the
    members must chain from byte 0 to the end.  Same packed text as from the plain file, no serial fall-back
    (DEBWT_GZ_REQUIRE_PARALLEL).  One member holds the bytes of a gzip header in its data (a false candidate): it is on no
    chain.  Bytes behind the last member: left to the serial path, which reads what gzread reads.
    way: when every candidate is a member, each member's ISIZE stands in front of the next candidate and the members are
    inflated straight into their places of one buffer ("into_place"; the file with the false candidate falls through to the
    general way by itself); DEBWT_GZ_MEMBERS_GENERAL forces the general way (a buffer per candidate, chained, copied)."""
    if way == "general":
        monkeypatch.setenv("DEBWT_GZ_MEMBERS_GENERAL", "1")
    rng = np.random.default_rng(500 + members)
    recs = [rng.integers(0, 4, size=int(rng.integers(60_000, 300_000))).astype(np.uint8) for _ in range(max(members, 6))]
    plain = str(tmp_path / "m.fa")
    _write(plain, recs, width=70)
    data = open(plain, "rb").read()
    starts = [0] + sorted({data.index(b">", len(data) * i // members) for i in range(1, members)}) + [len(data)]
    parts = [data[a:b] for a, b in zip(starts[:-1], starts[1:]) if b > a]
    blob = _gz_members(parts, level=6)
    p = str(tmp_path / "members.fa.gz")
    open(p, "wb").write(blob)
    assert gzip.open(p, "rb").read() == data
    monkeypatch.setenv("DEBWT_GZ_REQUIRE_PARALLEL", "1")
    monkeypatch.setenv("DEBWT_GZ_PIECE_BYTES", "65536")
    _check(p, recs, threads)
    # a false candidate: a comment line that holds the fixed bytes of a member header, in a member of stored blocks
    fake = b">x " + bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 2, 3]) + b" header bytes in a name\n" + b"ACGT" * 20 + b"\n"
    blob2 = gzip.compress(parts[0], mtime=0) + gzip.compress(fake, compresslevel=0, mtime=0) + _gz_members(parts[1:])
    p2 = str(tmp_path / "fake.fa.gz")
    open(p2, "wb").write(blob2)
    w, n, sep, _, _ = api.pack_fasta(p2, threads)
    plain2 = str(tmp_path / "fake.fa")
    open(plain2, "wb").write(parts[0] + fake + b"".join(parts[1:]))
    w0, n0, sep0, _, _ = api.pack_fasta(plain2, threads)
    assert n == n0 and np.array_equal(sep, sep0) and np.array_equal(w, w0)
    # bytes behind the last member: not this path's case
    p3 = str(tmp_path / "trail.fa.gz")
    open(p3, "wb").write(blob + b"\0" * 100)
    with pytest.raises(api.DebwtError, match="declined"):
        api.pack_fasta(p3, threads)
    monkeypatch.delenv("DEBWT_GZ_REQUIRE_PARALLEL")
    _check(p3, recs, threads)


def test_ingest_in_a_forked_child_of_a_process_that_has_ingested(tmp_path):
    """The library releases the buffers the gzip ingest gives up on a thread of its own (gz_parallel.cpp, release_later).  A
    process forked off one that has started that thread has the queue, the lock and the condition variable but not the thread:
    it must start its own, and must neither wait for nor join the parent's at its exit.  (Run in a process of its own, with a
    time limit: the failure is a child that never exits.)"""
    import subprocess, sys, textwrap
    rng = np.random.default_rng(8)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 400_000)].tobytes()
    rec = b"".join(seq[a:a + 80] + b"\n" for a in range(0, len(seq), 80))
    p = str(tmp_path / "members.fa.gz")
    with open(p, "wb") as f:
        for i in range(4):
            f.write(gzip.compress(b">r%d\n" % i + rec, mtime=0))
    code = textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
        import numpy as np
        from debwt_amd import api
        os.environ["DEBWT_GZ_REQUIRE_PARALLEL"] = "1"
        r = api.pack_fasta({p!r}, 4)
        pid = os.fork()
        if pid == 0:
            r2 = api.pack_fasta({p!r}, 4)
            sys.exit(0 if r2[1] == r[1] and np.array_equal(r2[0], r[0]) else 3)
        _, st = os.waitpid(pid, 0)
        r3 = api.pack_fasta({p!r}, 4)
        sys.exit(0 if os.WIFEXITED(st) and os.WEXITSTATUS(st) == 0 and r3[1] == r[1] else 4)
    """)
    done = subprocess.run([sys.executable, "-c", code], timeout=120, capture_output=True, text=True)
    assert done.returncode == 0, done.stderr[-2000:]


def test_gzip_files_damaged_at_random_are_taken_or_refused_as_the_serial_reader_does(tmp_path, monkeypatch):
    """Differential: gzip files of every shape the parallel paths take (BGZF, several plain members, one member in pieces) with a
    byte changed, a stretch cut out, the end cut off or bytes appended, at random: the threads' paths (fast_inflate.h,
    gz_parallel.cpp) and the serial reader (zlib's gzread, DEBWT_GZ_SERIAL) must both give the same packed text or both refuse
    the file.  (A change in a header's time stamp or OS byte is no damage; one in the deflate data or a trailer is.)"""
    rng = np.random.default_rng(2024)
    recs = [rng.integers(0, 4, size=int(rng.integers(20_000, 60_000))).astype(np.uint8) for _ in range(12)]
    plain = str(tmp_path / "d.fa")
    _write(plain, recs, width=70)
    data = open(plain, "rb").read()
    cuts = [0] + sorted({data.index(b">", len(data) * i // 5) for i in range(1, 5)}) + [len(data)]
    shapes = {
        "bgzf": None,
        "members": _gz_members([data[a:b] for a, b in zip(cuts[:-1], cuts[1:])], level=6),
        "one6": gzip.compress(data, compresslevel=6, mtime=0),
        "one1": gzip.compress(data, compresslevel=1, mtime=0),
    }
    pb = str(tmp_path / "b.gz")
    _write_bgzf(pb, data, block=3000)
    shapes["bgzf"] = open(pb, "rb").read()
    monkeypatch.setenv("DEBWT_GZ_PIECE_BYTES", "40000")
    outcomes = {"same": 0, "both refuse": 0}
    p = str(tmp_path / "x.fa.gz")
    for name, blob in shapes.items():
        for trial in range(24):
            z = bytearray(blob)
            how = trial % 6
            if how == 1: z[int(rng.integers(0, len(z)))] ^= 1 << int(rng.integers(0, 8))
            elif how == 2: a = int(rng.integers(0, len(z) - 200)); del z[a:a + int(rng.integers(1, 200))]
            elif how == 3: del z[int(rng.integers(len(z) // 2, len(z))):]
            elif how == 4: z += bytes(rng.integers(0, 256, size=int(rng.integers(1, 50))).astype(np.uint8))
            elif how == 5: z[int(rng.integers(len(z) - 40, len(z)))] ^= 0x10          # in the last member's data or trailer
            open(p, "wb").write(bytes(z))
            res = []
            for serial in (False, True):
                if serial: monkeypatch.setenv("DEBWT_GZ_SERIAL", "1")
                else: monkeypatch.delenv("DEBWT_GZ_SERIAL", raising=False)
                try:
                    w, n, sep, _, _ = api.pack_fasta(p, 5)
                    res.append((n, w.tobytes(), sep.tobytes()))
                except api.DebwtError:
                    res.append(None)
            assert (res[0] is None) == (res[1] is None), (name, trial, how)
            if res[0] is not None:
                assert res[0] == res[1], (name, trial, how)
                outcomes["same"] += 1
            else:
                outcomes["both refuse"] += 1
    assert outcomes["same"] >= 16 and outcomes["both refuse"] >= 30, outcomes


def test_text_bound_from_the_file_alone(tmp_path):
    """debwt_fasta_text_bound: what a host reserves device memory for while the file is still being parsed -- the file's bytes
    (plain), the members' ISIZE added up (block gzip), the ISIZE of a one-member gzip file; 0 where the framing cannot say
    (here: a file that is not there).  Never below the text's length for these shapes."""
    rng = np.random.default_rng(31)
    recs = [rng.integers(0, 4, size=int(rng.integers(5000, 50000))).astype(np.uint8) for _ in range(9)]
    plain = str(tmp_path / "b.fa")
    _write(plain, recs, width=61)
    data = open(plain, "rb").read()
    n = sum(len(r) for r in recs) + len(recs)
    assert api.fasta_text_bound(plain) == len(data) + 1 >= n
    one = str(tmp_path / "b1.fa.gz")
    open(one, "wb").write(gzip.compress(data, mtime=0))
    assert api.fasta_text_bound(one) == len(data) + 1
    bg = str(tmp_path / "b2.fa.gz")
    _write_bgzf(bg, data, block=3000)
    assert api.fasta_text_bound(bg) == len(data) + 1
    assert api.fasta_text_bound(str(tmp_path / "none.fa")) == 0

