"""Reference formats of the intermediates the stages hand to each other (SURVEY 8a "reference-compatible
intermediates"), as numpy conversions shared by the golden generator and the parity tests.  Test infrastructure."""
import numpy as np


def naive_kmer_counts(records, k):
    """Exact, non-canonical count of every k-mer inside each record (what `jellyfish count -m k` without -C followed
    by `dump -c -t` lists, src/kmercounting.sh:8,11), by definition: slide a window over every record, count with
    np.unique.  Returns (kmers left-aligned as src/mySort.c:61-75 packs them, ascending; counts)."""
    parts = []
    for r in records:
        r = np.asarray(r, dtype=np.uint64)
        if len(r) < k:
            continue
        v = np.zeros(len(r) - k + 1, dtype=np.uint64)
        for j in range(k):
            v = (v << np.uint64(2)) | r[j:len(r) - k + 1 + j]
        parts.append(v << np.uint64(64 - 2 * k))
    km, ct = np.unique(np.concatenate(parts), return_counts=True)
    return km, ct.astype(np.uint64)


def parse_kmer_dump(text, k):
    """'KMER<ws>COUNT' lines (any order) -> (kmers left-aligned ascending, counts)."""
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    km, ct = [], []
    for ln in text.split("\n"):
        if not ln:
            continue
        s, c = ln.split()
        v = 0
        for ch in s:
            v = (v << 2) | code[ch]
        km.append(v << (64 - 2 * k))
        ct.append(int(c))
    km, ct = np.array(km, dtype=np.uint64), np.array(ct, dtype=np.uint64)
    o = np.argsort(km, kind="stable")
    return km[o], ct[o]


def red_seq(red, k):
    """Red table (node << 2 | multiin << 1 | multiout, node = k-1 symbols right-aligned) -> redSeq entries
    (src/INandOut.c:396-404): only the low k-1-10 symbols of the node, the first 10 index blackTable."""
    red = np.asarray(red, dtype=np.uint64)
    extract = np.uint64((1 << (2 * (k - 1 - 10))) - 1)
    return (((red >> np.uint64(2)) & extract) << np.uint64(2)) | (red & np.uint64(3))


def red_point(red, blue_bound):
    """redPoint (src/INandOut.c:405): the inclusive end of the last multi-in block up to and including the entry's
    own (2^64-1 before the first)."""
    red = np.asarray(red, dtype=np.uint64)
    mi = ((red >> np.uint64(1)) & np.uint64(1)).astype(np.int64)
    q = np.cumsum(mi) - 1                                  # block of the last multi-in entry at or before r
    bb = np.concatenate([np.asarray(blue_bound, dtype=np.uint64), [np.uint64(0xFFFFFFFFFFFFFFFF)]])
    return bb[q]                                           # q = -1 -> the sentinel behind the array


def sp_symbols(sp_words, sp_len, sp_special):
    """spCode (2 bits per symbol, separators stored as 3) + spSpecialIndex (their SP positions, the last one is '$')
    -> one byte per symbol, A0 C1 G2 T3 #4 $5."""
    w = np.asarray(sp_words, dtype=np.uint64)
    j = np.arange(sp_len, dtype=np.uint64)
    sym = ((w[(j >> np.uint64(5)).astype(np.int64)] >> ((np.uint64(31) - (j & np.uint64(31))) * np.uint64(2))) & np.uint64(3)).astype(np.uint8)
    sp_special = np.asarray(sp_special, dtype=np.int64)
    if len(sp_special):
        sym[sp_special[:-1]] = 4
        sym[sp_special[-1]] = 5
    return sym


def blue_blocks_sorted(blue, blue_bound):
    """blueTable before the blue sort: the order of the entries inside a block is the order the text scan reached
    them (thread-dependent in the reference) -- only the set per block is defined.  Entries sorted inside each block."""
    blue = np.asarray(blue, dtype=np.uint64).copy()
    a = 0
    for e in np.asarray(blue_bound, dtype=np.int64):
        blue[a:e + 1] = np.sort(blue[a:e + 1])
        a = e + 1
    return blue
