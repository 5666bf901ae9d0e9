"""Native generator of the synthetic collections (include/debwt_synth.h, csrc/synth_host.cpp): the text of
synth.pan_chromosomes written straight into the reference's 2-bit layout on host threads.  Host-only (no GPU needed)."""
import ctypes
import os

import numpy as np

from . import _lib, synth


class SynthSpec(ctypes.Structure):
    _fields_ = [("seed", ctypes.c_uint64), ("genome_len", ctypes.c_uint64), ("genomes", ctypes.c_uint32),
                ("nchrom", ctypes.c_uint32), ("chrom_len", ctypes.POINTER(ctypes.c_uint64)),
                ("snp_rate", ctypes.c_double), ("repeat_coverage", ctypes.c_double),
                ("lowcx_fraction", ctypes.c_double), ("alu_copies", ctypes.c_uint32),
                ("alu_divergence", ctypes.c_double)]


def _bind(L):
    if getattr(L, "_synth_bound", False):
        return L
    vp, u64p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)
    L.debwt_synth_open.restype = ctypes.c_int
    L.debwt_synth_open.argtypes = [ctypes.POINTER(SynthSpec), ctypes.c_int, ctypes.POINTER(vp)]
    L.debwt_synth_close.restype = None
    L.debwt_synth_close.argtypes = [vp]
    for name in ("debwt_synth_n", "debwt_synth_nrec", "debwt_synth_nwords"):
        getattr(L, name).restype = ctypes.c_uint64
        getattr(L, name).argtypes = [vp]
    L.debwt_synth_sep.restype = ctypes.c_int
    L.debwt_synth_sep.argtypes = [vp, u64p]
    L.debwt_synth_words.restype = ctypes.c_int
    L.debwt_synth_words.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, vp, u64p]
    L.debwt_synth_codes.restype = ctypes.c_int
    L.debwt_synth_codes.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64, vp]
    L.debwt_pinned_alloc.restype = ctypes.c_int
    L.debwt_pinned_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(vp)]
    L.debwt_pinned_free.restype = None
    L.debwt_pinned_free.argtypes = [vp]
    L._synth_bound = True
    return L


def default_threads():
    """Host threads this process may really use: the CPU affinity mask, capped by the cgroup's CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


# named workloads: (genome_len, genomes, chroms, extra spec fields)
WORKLOADS = {
    # BASELINE.json configs[4], the configuration the metric is quoted on (README.md:19 "10 human genomes"):
    # 10 genomes x 3.0 Gbp, each cut into 24 chromosome-like records
    "pan10x3G": (3_000_000_000, 10, 24, {}),
    # configs[3]: 4 x GRCh38-sized
    "pan4x3.1G": (3_100_000_000, 4, 24, {}),
    # configs[2]: one GRCh38-sized genome, 24 records
    "grch38_3.1G": (3_100_000_000, 1, 24, {}),
    # configs[1] / configs[0]
    "chr1_250M": (250_000_000, 1, 1, {}),
    "ecoli_4.6M": (4_600_000, 1, 1, {}),
    # distribution U: uniform, no repeats
    "uniform_3.1G": (3_100_000_000, 1, 24, {"repeat_coverage": 0.0, "seed": synth.SEED_U}),
    # distribution R: repeat families + one Alu-like family (10^6 copies per 3.1 Gbp, 12 % divergence) + 3 % satellite
    # arrays and homopolymer / microsatellite tracts
    "real_3.1G": (3_100_000_000, 1, 24, {"lowcx_fraction": 0.03, "alu_copies": 1_000_000}),
    # distribution R at the size the metric is quoted on: ten such genomes (SNPs at 1e-3 between them), 240 records
    "real10x3G": (3_000_000_000, 10, 24, {"lowcx_fraction": 0.03, "alu_copies": 1_000_000}),
    # the two headline distributions at a size whose shards fit next to each other in ONE GPU's HBM (~40 bytes per base over
    # all shards: N = 2 and 4 here; N = 8 ran out of memory and was measured at 10 x 400 Mbp): per-shard measurements on a
    # one-GPU box (scripts/gpu_shard_balance.py);
    # the same densities of repeat families, Alu-like copies and low-complexity tracts as the 3 Gbp genomes
    "pan10x600M": (600_000_000, 10, 24, {}),
    "real10x600M": (600_000_000, 10, 24, {"lowcx_fraction": 0.03, "alu_copies": 200_000}),
    # small shapes of the same kinds (tests)
    "pan_small": (300_000, 4, 3, {}),
    "real_small": (1_500_000, 2, 3, {"lowcx_fraction": 0.03, "alu_copies": 500}),
}


class Synth:
    """One collection = synth.pan_chromosomes(genome_len, genomes, chroms, ...)."""

    def __init__(self, genome_len, genomes=1, chroms=1, seed=synth.SEED_P, snp_rate=1e-3, repeat_coverage=0.25,
                 lowcx_fraction=0.0, alu_copies=0, alu_divergence=0.12, threads=None):
        self._L = _bind(_lib.lib())
        self.threads = threads or default_threads()
        self.genome_len, self.genomes, self.chroms = genome_len, genomes, chroms
        self.kw = dict(seed=seed, snp_rate=snp_rate, repeat_coverage=repeat_coverage, lowcx_fraction=lowcx_fraction,
                       alu_copies=alu_copies, alu_divergence=alu_divergence)
        self._lens = np.array(synth.chromosome_lengths(genome_len, chroms), dtype=np.uint64)
        spec = SynthSpec(seed, genome_len, genomes, chroms, self._lens.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                         snp_rate, repeat_coverage, lowcx_fraction, alu_copies, alu_divergence)
        h = ctypes.c_void_p()
        rc = self._L.debwt_synth_open(ctypes.byref(spec), self.threads, ctypes.byref(h))
        if rc:
            raise RuntimeError(f"debwt_synth_open failed: {rc}")
        self._h = h
        self.n = int(self._L.debwt_synth_n(h))
        self.nrec = int(self._L.debwt_synth_nrec(h))
        self.nwords = int(self._L.debwt_synth_nwords(h))

    @classmethod
    def named(cls, name, seed=None, threads=None):
        if name not in WORKLOADS:
            # pan<G>x<L>[M|G] / real<G>x<L>[M|G]: G genomes of L bases in 24 records each, distribution P or R at the
            # densities of the named 3 Gbp workloads (an Alu-like copy per 3,000 bases, 3 % low-complexity tracts)
            import re
            mt = re.fullmatch(r"(pan|real)(\d+)x(\d+(?:\.\d+)?)([MG])", name)
            if not mt:
                raise KeyError(f"unknown workload {name!r}")
            gl = int(float(mt.group(3)) * (1_000_000 if mt.group(4) == "M" else 1_000_000_000))
            extra = {"lowcx_fraction": 0.03, "alu_copies": gl // 3000} if mt.group(1) == "real" else {}
            WORKLOADS[name] = (gl, int(mt.group(2)), 24, extra)
        gl, g, c, extra = WORKLOADS[name]
        kw = dict(extra)
        if seed is not None:
            kw["seed"] = seed
        return cls(gl, g, c, threads=threads, **kw)

    def close(self):
        if getattr(self, "_h", None):
            self._L.debwt_synth_close(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def sep(self):
        out = np.empty(self.nrec, dtype=np.uint64)
        self._L.debwt_synth_sep(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)))
        return out

    def words_into(self, ptr, w0=0, w1=None):
        """Packed words [w0, w1) of the text to the host address `ptr`; returns the base census of those words."""
        w1 = self.nwords if w1 is None else w1
        census = np.zeros(4, dtype=np.uint64)
        rc = self._L.debwt_synth_words(self._h, w0, w1, self.threads, ctypes.c_void_p(ptr),
                                       census.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)))
        if rc:
            raise RuntimeError(f"debwt_synth_words failed: {rc}")
        return census

    def words(self, w0=0, w1=None):
        w1 = self.nwords if w1 is None else w1
        out = np.empty(w1 - w0, dtype=np.uint64)
        census = self.words_into(out.ctypes.data, w0, w1)
        return out, census

    def codes(self, genome=0, i0=0, i1=None):
        i1 = self.genome_len if i1 is None else i1
        out = np.empty(i1 - i0, dtype=np.uint8)
        rc = self._L.debwt_synth_codes(self._h, genome, i0, i1, out.ctypes.data_as(ctypes.c_void_p))
        if rc:
            raise RuntimeError(f"debwt_synth_codes failed: {rc}")
        return out

    def records_numpy(self):
        """The same collection from the numpy definition (small sizes: tests)."""
        return synth.pan_chromosomes(self.genome_len, self.genomes, self.chroms, **self.kw)


class PinnedArray:
    """uint64 array in page-locked host memory (debwt_pinned_alloc); .a is the numpy view."""

    def __init__(self, nwords):
        self._L = _bind(_lib.lib())
        p = ctypes.c_void_p()
        rc = self._L.debwt_pinned_alloc(int(nwords) * 8, ctypes.byref(p))
        if rc:
            raise MemoryError(f"debwt_pinned_alloc({nwords * 8} bytes) failed: {rc}")
        self.ptr = p.value
        self.a = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint64)), shape=(int(nwords),))

    def free(self):
        if getattr(self, "ptr", None):
            self.a = None
            self._L.debwt_pinned_free(ctypes.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        self.free()
