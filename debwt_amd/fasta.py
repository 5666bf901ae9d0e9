"""Minimal FASTA reader/writer for the host side (the reference reads FASTA/FASTQ through
klib's kseq.h + zlib, src/collect#$.c:26,34-90).  Plain or gzip-compressed FASTA; sequence
letters must be ACGT in either case (README.md:37)."""
import gzip

import numpy as np

_LUT = np.full(256, 255, dtype=np.uint8)
for _c, _v in zip(b"ACGTacgt", (0, 1, 2, 3, 0, 1, 2, 3)):
    _LUT[_c] = _v


def _open(path, mode):
    with open(path, "rb") as f:
        magic = f.read(2)
    return gzip.open(path, mode) if magic == b"\x1f\x8b" else open(path, mode)


def read_fasta(path):
    """Returns (names, list of uint8 code arrays)."""
    names, recs, cur = [], [], None
    with _open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if not line:
                continue
            if line[:1] == b">":
                if cur is not None:
                    recs.append(cur)
                names.append(line[1:].decode())
                cur = []
            else:
                if cur is None:
                    raise ValueError("sequence before the first header")
                cur.append(line)
    if cur is not None:
        recs.append(cur)
    out = []
    for parts in recs:
        codes = _LUT[np.frombuffer(b"".join(parts), dtype=np.uint8)]
        if (codes == 255).any():
            raise ValueError("sequence holds letters other than ACGT (see README.md:37 of the reference)")
        out.append(codes)
    return names, out


def write_fasta(path, records, names=None, width=70, lower=False):
    """records: list of uint8 code arrays (0..3) or ASCII bytes/str."""
    alphabet = np.frombuffer(b"acgt" if lower else b"ACGT", dtype=np.uint8)
    with open(path, "wb") as f:
        for i, r in enumerate(records):
            name = names[i] if names else f"r{i}"
            f.write(b">" + name.encode() + b"\n")
            if isinstance(r, str):
                s = r.encode()
            elif isinstance(r, (bytes, bytearray)):
                s = bytes(r)
            else:
                s = alphabet[np.asarray(r, dtype=np.uint8)].tobytes()
            for a in range(0, len(s), width):
                f.write(s[a:a + width] + b"\n")
