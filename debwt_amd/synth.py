"""Formula-defined synthetic DNA: both the build container and the GPU box regenerate
the same inputs from (distribution, length, records, seed) -- no big file ever ships.

  U  uniform:     b(i) = (splitmix64(seed + (i >> 5)) >> (2 * (i & 31))) & 3
  P  pan-genome:  one base genome (uniform + repeat families: consensus 300..5999 bases,
                  5..200 copies, 2 % divergence, ~25 % coverage), G records = that genome with
                  independent SNPs at rate 1e-3 each.

All records are > 32 bases and ACGT only (reference input domain, src/collect#$.c:41-45,
README.md:37).  Arrays are uint8 codes A0 C1 G2 T3.
"""
import numpy as np

SEED_U = 0xDEB07
SEED_P = 0x5EEDBA5E
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """Vectorised splitmix64 finaliser of the uint64 array x (wrapping arithmetic)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _mix(*vals):
    h = np.uint64(0x243F6A8885A308D3)
    for v in vals:
        h = splitmix64(np.uint64(h) ^ np.uint64(v & 0xFFFFFFFFFFFFFFFF))
    return int(h)


def uniform_codes(length, seed=SEED_U, start=0):
    """Bases start..start+length of the uniform stream for `seed`."""
    w0, w1 = start >> 5, (start + length + 31) >> 5
    out = np.empty((w1 - w0) * 32, dtype=np.uint8)
    chunk = 1 << 20
    shifts = (np.arange(32, dtype=np.uint64) * np.uint64(2))[None, :]
    for a in range(w0, w1, chunk):
        b = min(a + chunk, w1)
        with np.errstate(over="ignore"):
            words = splitmix64(np.uint64(seed) + np.arange(a, b, dtype=np.uint64))
        out[(a - w0) * 32:(b - w0) * 32] = ((words[:, None] >> shifts) & np.uint64(3)).astype(np.uint8).ravel()
    off = start - (w0 << 5)
    return out[off:off + length]


def _mutate(codes, rate, key):
    """Substitute each base with probability `rate` (hash-defined), never by itself."""
    n = len(codes)
    thr = np.uint64(int(rate * float(1 << 64)))
    chunk = 1 << 22
    for a in range(0, n, chunk):
        b = min(a + chunk, n)
        with np.errstate(over="ignore"):
            h = splitmix64(np.uint64(key) + np.arange(a, b, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        idx = np.nonzero(h < thr)[0]
        if len(idx):
            delta = ((h[idx] >> np.uint64(61)) % np.uint64(3) + np.uint64(1)).astype(np.uint8)
            codes[a + idx] = (codes[a + idx] + delta) & 3
    return codes


def base_genome(length, seed=SEED_P, repeat_coverage=0.25, lowcx_fraction=0.0, alu_copies=0, alu_divergence=0.12):
    """Uniform genome with repeat families written over it.  Distribution R adds what real genomes put into the
    large blue blocks: one short interspersed family with very many diverged copies (Alu-like), satellite arrays
    (tandem copies of a 5..171-base unit, 1 % divergence, lowcx_fraction of the genome) and homopolymer /
    microsatellite tracts (one per 150 / lowcx_fraction bases)."""
    g = uniform_codes(length, seed).copy()
    if length < 2000:
        return g
    covered, target, f = 0, int(repeat_coverage * length), 0
    while covered < target:
        clen = 300 + _mix(seed, f, 1) % 5700
        clen = min(clen, max(64, length // 8))
        copies = 5 + _mix(seed, f, 2) % 196
        cons = uniform_codes(clen, _mix(seed, f, 3) & 0x7FFFFFFFFFFF)
        for c in range(copies):
            if covered >= target:
                break
            p = _mix(seed, f, 4, c) % (length - clen)
            cp = _mutate(cons.copy(), 0.02, _mix(seed, f, 5, c))
            g[p:p + clen] = cp
            covered += clen
        f += 1
    if alu_copies:
        cons = uniform_codes(300, _mix(seed, 0xA1, 0) & 0x7FFFFFFFFFFF)
        for c in range(alu_copies):
            p = _mix(seed, 0xA1, 1, c) % (length - 300)
            g[p:p + 300] = _mutate(cons.copy(), alu_divergence, _mix(seed, 0xA1, 2, c))
    if lowcx_fraction > 0:
        cap = max(64, length // 8)
        covered, target, t = 0, int(lowcx_fraction * length), 0
        while covered < target:                              # satellite arrays
            ulen = 5 + _mix(seed, 0x5A7, t, 1) % 167
            alen = min(1000 + _mix(seed, 0x5A7, t, 2) % 99000, cap)
            unit = uniform_codes(ulen, _mix(seed, 0x5A7, t, 3) & 0x7FFFFFFFFFFF)
            arr = np.tile(unit, (alen + ulen - 1) // ulen)[:alen].copy()
            p = _mix(seed, 0x5A7, t, 4) % (length - alen)
            g[p:p + alen] = _mutate(arr, 0.01, _mix(seed, 0x5A7, t, 5))
            covered += alen
            t += 1
        for t in range(int(length * lowcx_fraction / 150)):  # homopolymer and microsatellite tracts (exact repeats)
            kind = _mix(seed, 0x7AC, t, 0) % 3
            if kind < 2:
                ulen, tlen = 1, 20 + _mix(seed, 0x7AC, t, 1) % 181
                unit = np.array([(0, 3, 0, 3, 1, 2)[_mix(seed, 0x7AC, t, 2) % 6]], dtype=np.uint8)
            else:
                ulen, tlen = 2 + _mix(seed, 0x7AC, t, 1) % 5, 30 + _mix(seed, 0x7AC, t, 2) % 471
                unit = uniform_codes(ulen, _mix(seed, 0x7AC, t, 3) & 0x7FFFFFFFFFFF)
            tlen = min(tlen, cap)
            p = _mix(seed, 0x7AC, t, 4) % (length - tlen)
            g[p:p + tlen] = np.tile(unit, (tlen + ulen - 1) // ulen)[:tlen]
    return g


def pan_genome(length, records=1, seed=SEED_P, snp_rate=1e-3, repeat_coverage=0.25):
    """`records` records of `length` bases: the base genome with independent SNPs per record."""
    g = base_genome(length, seed, repeat_coverage)
    out = []
    for r in range(records):
        rec = g.copy() if records > 1 else g
        if r > 0 or records > 1:
            _mutate(rec, snp_rate, _mix(seed, 0xC0FFEE, r))
        out.append(rec)
    return out


def chromosomes(total, records, seed=SEED_P, repeat_coverage=0.25):
    """One genome of `total` bases cut into `records` chromosome-like records (sizes shrink
    geometrically like a karyotype), every record > 32 bases."""
    g = base_genome(total, seed, repeat_coverage)
    w = np.array([0.93 ** i for i in range(records)])
    cuts = np.floor(np.cumsum(w / w.sum()) * total).astype(np.int64)
    cuts[-1] = total
    out, a = [], 0
    for c in cuts:
        c = int(max(c, a + 33))
        out.append(g[a:c])
        a = c
    return out


def chromosome_lengths(total, records):
    """Record lengths of one genome of `total` bases cut into `records` chromosome-like pieces (sizes shrink
    geometrically like a karyotype), every record > 32 bases."""
    w = np.array([0.93 ** i for i in range(records)])
    cuts = np.floor(np.cumsum(w / w.sum()) * total).astype(np.int64)
    cuts[-1] = total
    out, a = [], 0
    for c in cuts:
        c = int(max(c, a + 33))
        out.append(c - a)
        a = c
    return out


def pan_chromosomes(genome_len, genomes, chroms, seed=SEED_P, snp_rate=1e-3, repeat_coverage=0.25, lowcx_fraction=0.0,
                    alu_copies=0, alu_divergence=0.12):
    """Distribution P at chromosome granularity ("10 genomes of 24 chromosome-like records"):
    `genomes` copies of one base genome, each with independent SNPs, each cut into the same `chroms` records.
    genomes == 1 gives chromosomes(), chroms == 1 gives pan_genome().  The native generator
    (csrc/synth_host.cpp, debwt_synth_*) produces the same text without materialising it in numpy."""
    g = base_genome(genome_len, seed, repeat_coverage, lowcx_fraction, alu_copies, alu_divergence)
    lens = chromosome_lengths(genome_len, chroms)
    out = []
    for j in range(genomes):
        gj = g.copy() if genomes > 1 else g
        if genomes > 1:
            _mutate(gj, snp_rate, _mix(seed, 0xC0FFEE, j))
        a = 0
        for ln in lens:
            out.append(gj[a:a + ln])
            a += ln
    return out


def read_set(records, min_len, max_len, genome_len, seed=SEED_P, snap=8, dup_every=16):
    """Many short records (the input the special-region module exists for, SURVEY 8f-1: contig-level assemblies, read
    sets): `records` substrings of one base genome with repeat families, lengths min_len..max_len (> 32), starts
    hash-defined; start and end positions are snapped to multiples of `snap`, so many records share their last or first
    bases with other records -- equal windows across '#' that continue differently (special branches,
    src/collect#$.c:534-598) and long ties among the special suffixes; every `dup_every`-th record repeats an earlier
    one exactly."""
    assert min_len > 32 + 2 * snap and max_len >= min_len and genome_len > 2 * max_len
    g = base_genome(genome_len, seed)
    i = np.arange(records, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h1 = splitmix64(np.uint64(_mix(seed, 0xAEAD5)) + i * np.uint64(0x9E3779B97F4A7C15))
        h2 = splitmix64(h1 ^ np.uint64(0x5851F42D4C957F2D))
    ln = (np.uint64(min_len) + h2 % np.uint64(max_len - min_len + 1)).astype(np.int64)
    st = (h1 % np.uint64(genome_len - max_len)).astype(np.int64)
    st = st // snap * snap
    en = np.minimum((st + ln) // snap * snap, genome_len)
    out = []
    for r in range(records):
        if dup_every and r % dup_every == dup_every - 1:
            out.append(out[int(h2[r] >> np.uint64(32)) % r].copy())
        else:
            out.append(g[int(st[r]):int(en[r])].copy())
    return out


def codes_to_ascii(codes):
    return np.frombuffer(b"ACGT", dtype=np.uint8)[np.asarray(codes, dtype=np.uint8)].tobytes()


def make_workload(name):
    """Named workloads used by bench.py and the tests; returns a list of uint8 code arrays."""
    table = {
        # BASELINE.json configs[0]: E. coli-sized, single record
        "ecoli_4.6M": lambda: pan_genome(4_600_000, 1),
        # BASELINE.json configs[1]: chr1-sized, single record, repeat families
        "chr1_250M": lambda: pan_genome(250_000_000, 1),
        "pan_100M_4": lambda: chromosomes(100_000_000, 4),
        "pan_16M_4": lambda: pan_genome(4_000_000, 4),
        "uniform_16M": lambda: [uniform_codes(16_000_000)],
        "tiny_64k_3": lambda: pan_genome(21_000, 3),
    }
    if name not in table:
        raise KeyError(f"unknown workload {name}; have {sorted(table)}")
    return table[name]()
