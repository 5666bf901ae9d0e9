"""Host-side mirror of the reference's stage interface (src/main.h:1-8, call order src/main.c:83-149)
over the C ABI of libdebwt_hip.so.  Same stage names and order; state lives in a context object instead
of globals and temp files; errors raise DebwtError instead of exit(1)."""
import ctypes

import numpy as np

from . import _lib

ARR_SORTED_KEYS, ARR_DISTINCT_KEYS, ARR_RED, ARR_SP_SYMBOLS, ARR_BLUE, ARR_BLUE_BOUND, ARR_CASE3_BOUND, \
    ARR_ROW_SYMBOLS = range(1, 9)
DUMP_KMERINFO, DUMP_BLOCKS, DUMP_SP = 1, 2, 3
_ARR_DTYPE = {ARR_SP_SYMBOLS: np.uint8, ARR_ROW_SYMBOLS: np.uint8}


class DebwtError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        msg = _lib.lib().debwt_strerror(code).decode()
        super().__init__(f"{msg} ({code})" + (f": {detail}" if detail else ""))


def _p64(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


def pack_records(records):
    """2-bit packed text in the reference layout (src/collect#$.c:61-90): base j at bit
    2*(31-(j&31)) of word j>>5, 'T' at every separator, 32 'T' of padding.  records: list of uint8
    code arrays (A0 C1 G2 T3).  Returns (words, n, sep)."""
    lens = np.array([len(r) for r in records], dtype=np.uint64)
    if (lens <= 32).any():
        raise ValueError("every record must be longer than 32 bases (src/collect#$.c:41-45)")
    n = int(lens.sum()) + len(records)
    total = n + 32
    nwords = (total + 31) // 32 + 1
    sym = np.full(nwords * 32, 3, dtype=np.uint8)
    sym[total:] = 0
    sep = np.empty(len(records), dtype=np.uint64)
    o = 0
    for i, r in enumerate(records):
        r = np.asarray(r, dtype=np.uint8)
        if r.size and r.max() > 3:
            raise ValueError("codes must be 0..3")
        sym[o:o + len(r)] = r
        o += len(r)
        sep[i] = o
        o += 1
    q = sym.reshape(-1, 4)
    b = (q[:, 0] << 6) | (q[:, 1] << 4) | (q[:, 2] << 2) | q[:, 3]
    words = np.ascontiguousarray(b).view(">u8").astype(np.uint64)
    return words, n, sep


class DeBWT:
    """One context = one GPU.  Typical use:
        d = DeBWT(k=32); d.load_records(records); d.build(); words, hash_rows, dollar_row = d.fetch()
    or stage by stage: kmer_sort_rle(), classify(), sp_generate(), blue_sort(), bwt_assemble()."""

    def __init__(self, k=32, device=0, sort_algo=0, tune=0):
        self._L = _lib.lib()
        cfg = _lib.DebwtConfig(k=k, device=device, sort_algo=sort_algo, reserved=tune)
        h = ctypes.c_void_p()
        rc = self._L.debwt_create(ctypes.byref(cfg), ctypes.byref(h))
        if rc:
            raise DebwtError(rc)
        self._h = h
        self.k = k
        self.n = 0
        self.nrec = 0
        self._keep = None

    def close(self):
        if getattr(self, "_h", None):
            self._L.debwt_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc:
            raise DebwtError(rc, self._L.debwt_last_error(self._h).decode())

    # -- loading ------------------------------------------------------------------------------------
    def load_packed(self, words, n, sep):
        words = np.ascontiguousarray(words, dtype=np.uint64)
        sep = np.ascontiguousarray(sep, dtype=np.uint64)
        self._keep = (words, sep)       # the library reads the host text during a run
        self._chk(self._L.debwt_load_text(self._h, _p64(words), n, _p64(sep), len(sep)))
        self.n, self.nrec = n, len(sep)

    def load_records(self, records):
        words, n, sep = pack_records(records)
        self.load_packed(words, n, sep)

    def load_ascii(self, records):
        recs = [r.encode() if isinstance(r, str) else bytes(r) for r in records]
        lens = np.array([len(r) for r in recs], dtype=np.uint64)
        self._chk(self._L.debwt_load_ascii(self._h, b"".join(recs), _p64(lens), len(recs)))
        self.n, self.nrec = int(lens.sum()) + len(recs), len(recs)

    def load_fasta(self, path, threads=8, iupac_seed=None):
        """FASTA (plain or gzip) parsed and packed by `threads` host threads, then loaded.  iupac_seed: replace N and
        the other ambiguity letters by pseudo-random bases of their sets (deterministic in seed and position)."""
        if iupac_seed is None:
            self._chk(self._L.debwt_load_fasta(self._h, str(path).encode(), int(threads)))
        else:
            self._chk(self._L.debwt_load_fasta_opts(self._h, str(path).encode(), int(threads), FASTA_IUPAC_RANDOM, int(iupac_seed)))
        st = self.stats()
        self.n, self.nrec = st["n"], st["nrec"]

    def reserve(self, n, nrec, branching=0.0, one_shot=False, compact=False):
        """Allocate the workspace of a text of up to n symbols in nrec records ahead of the load (debwt_reserve): call it
        on a thread of its own while the input is still being read -- ctypes releases the GIL for the call."""
        self._chk(self._L.debwt_reserve(self._h, int(n), int(nrec), float(branching), (1 if one_shot else 0) | (2 if compact else 0)))

    def set_range_cap(self, max_instances):
        """Largest number of node instances sorted in one go; larger texts are built in k-mer-prefix ranges."""
        self._chk(self._L.debwt_set_range_cap(self._h, int(max_instances)))

    # -- stages (src/main.c:83-149) -------------------------------------------------------------------
    def kmer_sort_rle(self):
        self._chk(self._L.debwt_kmer_sort_rle(self._h))

    def classify(self):
        self._chk(self._L.debwt_classify(self._h))

    def sp_generate(self):
        self._chk(self._L.debwt_sp_generate(self._h))

    def blue_sort(self):
        self._chk(self._L.debwt_blue_sort(self._h))

    def bwt_assemble(self):
        self._chk(self._L.debwt_bwt_assemble(self._h))

    def build(self):
        self._chk(self._L.debwt_build(self._h))

    # -- results --------------------------------------------------------------------------------------
    def fetch(self):
        words = np.empty((self.n + 31) // 32, dtype=np.uint64)
        hrows = np.empty(max(self.nrec - 1, 1), dtype=np.uint64)
        drow = np.empty(1, dtype=np.uint64)
        self._chk(self._L.debwt_fetch_bwt(self._h, _p64(words), _p64(hrows), _p64(drow)))
        return words, hrows[:self.nrec - 1], int(drow[0])

    def fetch_into(self, words, hrows, drow):
        """fetch() into caller-owned uint64 arrays (e.g. page-locked ones): ceil(n/32), nrec-1 (>= 1), 1 words."""
        self._chk(self._L.debwt_fetch_bwt(self._h, _p64(words), _p64(hrows), _p64(drow)))

    def build_into(self, words, hrows, drow):
        """build() + fetch_into() with the copy of finished row ranges hidden behind the blue sort of the following ones
        (debwt_build_to_host); page-locked arrays for the overlap."""
        self._chk(self._L.debwt_build_to_host(self._h, _p64(words), _p64(hrows), _p64(drow)))

    def fetch_small(self):
        """(None, hash_rows, dollar_row): the row lists only, the BWT words stay in HBM."""
        hrows = np.empty(max(self.nrec - 1, 1), dtype=np.uint64)
        drow = np.empty(1, dtype=np.uint64)
        self._chk(self._L.debwt_fetch_rows(self._h, _p64(hrows), _p64(drow)))
        return None, hrows[:self.nrec - 1], int(drow[0])

    def bwt_census(self):
        """Rows of the result per 2-bit code (computed on the device)."""
        c = np.zeros(4, dtype=np.uint64)
        self._chk(self._L.debwt_bwt_census(self._h, _p64(c)))
        return c

    def verify_device(self, d_words=None, hash_rows=None, dollar_row=0, segments=0):
        """Inverse BWT on the device against the loaded text (debwt_verify_device).  Default: the context's own result;
        otherwise d_words = device address of packed rows, hash_rows / dollar_row their row lists."""
        rep = _lib.DebwtVerifyReport()
        hp = None
        if d_words is not None and hash_rows is not None and len(hash_rows):
            hash_rows = np.ascontiguousarray(hash_rows, dtype=np.uint64)
            hp = _p64(hash_rows)
        self._chk(self._L.debwt_verify_device(self._h, ctypes.c_void_p(d_words) if d_words else None, hp, int(dollar_row),
                                              int(segments), ctypes.byref(rep)))
        r = rep.as_dict()
        return {"inverse_bwt_ok": bool(r["ok"]), "inverse_bwt": {k: (round(v, 2) if isinstance(v, float) else v)
                                                                  for k, v in r.items() if k != "ok"}}

    def special_compare(self):
        """Special-region tables of the loaded text, device module against host module: mismatching elements of
        (suffix order, keys, BWT symbols, special branches, head nodes, tail nodes) -- all zero when they agree."""
        mm = np.zeros(6, dtype=np.uint64)
        self._chk(self._L.debwt_special_compare(self._h, _p64(mm)))
        return [int(x) for x in mm]

    def stats(self):
        st = _lib.DebwtStats()
        self._chk(self._L.debwt_get_stats(self._h, ctypes.byref(st)))
        return st.as_dict()

    def fetch_array(self, which):
        cnt = ctypes.c_uint64()
        self._chk(self._L.debwt_fetch_array(self._h, which, None, 0, ctypes.byref(cnt)))
        out = np.empty(max(cnt.value, 1), dtype=_ARR_DTYPE.get(which, np.uint64))
        self._chk(self._L.debwt_fetch_array(self._h, which, out.ctypes.data_as(ctypes.c_void_p), cnt.value,
                                            ctypes.byref(cnt)))
        return out[:cnt.value]

    def dump_reference_files(self, directory, stage):
        """Intermediates of the stage just run as files in the reference's byte formats (debwt_dump_reference_files):
        stage DUMP_KMERINFO (kmerInfo), DUMP_BLOCKS after classify() (redSeq, redPoint, blueBound, case3bound),
        DUMP_SP after sp_generate() (spCode, spSpecialIndex, blueTable)."""
        self._chk(self._L.debwt_dump_reference_files(self._h, str(directory).encode(), int(stage)))

    def kmer_count_sorted(self):
        """(kmers left-aligned, counts): the contents of the reference's kmerInfo (src/mySort.c:193-195)."""
        d = ctypes.c_uint64()
        self._chk(self._L.debwt_kmer_count_sorted(self._h, None, None, 0, ctypes.byref(d)))
        km = np.empty(max(d.value, 1), dtype=np.uint64)
        ct = np.empty(max(d.value, 1), dtype=np.uint64)
        self._chk(self._L.debwt_kmer_count_sorted(self._h, _p64(km), _p64(ct), d.value, ctypes.byref(d)))
        return km[:d.value], ct[:d.value]

    def radix_sort_device(self, keys_ptr, tmp_ptr, count, key_bits=64, want_ms=False):
        """Sort `count` u64 keys resident in HBM (raw device pointers, e.g. torch tensor .data_ptr())."""
        ms = ctypes.c_float()
        self._chk(self._L.debwt_radix_sort_u64(self._h, ctypes.c_void_p(keys_ptr), ctypes.c_void_p(tmp_ptr), count,
                                               key_bits, ctypes.byref(ms) if want_ms else None))
        return ms.value

    def bwt_device_ptr(self):
        p = ctypes.c_void_p()
        self._chk(self._L.debwt_bwt_device_ptr(self._h, ctypes.byref(p)))
        return p.value


class MultiDeBWT:
    """One BWT over several GPUs from one process (debwt_multi_*): one host thread per GPU, peer-to-peer exchanges.
    devices: HIP ordinals of the shards (may repeat: several shards on one GPU)."""

    def __init__(self, devices, k=32, tune=0):
        self._L = _lib.lib()
        cfg = _lib.DebwtConfig(k=k, device=0, sort_algo=0, reserved=tune)
        dv = (ctypes.c_int * len(devices))(*devices)
        h = ctypes.c_void_p()
        rc = self._L.debwt_multi_create(ctypes.byref(cfg), dv, len(devices), ctypes.byref(h))
        if rc:
            raise DebwtError(rc)
        self._h, self.n, self.nrec, self._keep = h, 0, 0, None

    def close(self):
        if getattr(self, "_h", None):
            self._L.debwt_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc:
            raise DebwtError(rc, self._L.debwt_multi_last_error(self._h).decode())

    def _shard_ctx(self, shard):
        return ctypes.c_void_p(self._L.debwt_multi_shard(self._h, shard))

    def load_packed(self, words, n, sep):
        words = np.ascontiguousarray(words, dtype=np.uint64)
        sep = np.ascontiguousarray(sep, dtype=np.uint64)
        self._keep = (words, sep)
        self._chk(self._L.debwt_multi_load_text(self._h, _p64(words), n, _p64(sep), len(sep)))
        self.n, self.nrec = n, len(sep)

    def load_records(self, records):
        self.load_packed(*pack_records(records))

    def set_key_mode(self, mode):
        """"exchange", "rescan" or "auto" (the library's cost model decides): how the keys reach their shards."""
        self._chk(self._L.debwt_multi_set_key_mode(self._h, {"exchange": 0, "rescan": 1, "auto": -1}[mode]))

    def set_exchange(self, backend):
        """"peer" (device-to-device copies, the default) or "rccl" (grouped ncclSend / ncclRecv; one distinct GPU per shard)."""
        self._chk(self._L.debwt_multi_set_exchange(self._h, {"peer": 0, "rccl": 1}[backend]))

    def build(self):
        self._chk(self._L.debwt_multi_build(self._h))

    def fetch(self):
        words = np.empty((self.n + 31) // 32, dtype=np.uint64)
        hrows = np.empty(max(self.nrec - 1, 1), dtype=np.uint64)
        drow = np.empty(1, dtype=np.uint64)
        self._chk(self._L.debwt_multi_fetch_bwt(self._h, _p64(words), _p64(hrows), _p64(drow)))
        return words, hrows[:self.nrec - 1], int(drow[0])

    def stats(self):
        ms, s0 = _lib.DebwtMultiStats(), _lib.DebwtStats()
        self._chk(self._L.debwt_multi_get_stats(self._h, ctypes.byref(ms), ctypes.byref(s0)))
        return ms.as_dict(), s0.as_dict()

    def set_serial(self, on=True):
        """The shards of a build take turns between the barriers, one on its GPU at a time (debwt_multi_set_serial): how the
        per-shard times of N = 2, 4, 8 are measured on a box with one GPU."""
        self._chk(self._L.debwt_multi_set_serial(self._h, 1 if on else 0))

    def shard_report(self, shard):
        """What shard `shard` did in the last build: wall ms per step (by name), bytes in/out per exchange, its sizes, and the
        device-side stage times and counters of its context."""
        rep = _lib.DebwtShardReport()
        self._chk(self._L.debwt_multi_get_shard_report(self._h, int(shard), ctypes.byref(rep)))
        names = [self._L.debwt_multi_step_name(i).decode() for i in range(_lib.MULTI_STEPS)]
        xn = ("keys", "facts", "sp", "blue", "rows")
        st = _lib.DebwtStats()
        self._L.debwt_get_stats(self._shard_ctx(shard), ctypes.byref(st))
        return {"shard": int(shard), "bins": [int(rep.bin_lo), int(rep.bin_hi)], "keys": int(rep.keys),
                "key_ranges": int(rep.key_ranges), "blocks": int(rep.blocks), "blue_rows": int(rep.blue_rows), "rows": int(rep.rows),
                "ms": {nm: round(float(rep.ms[i]), 3) for i, nm in enumerate(names) if rep.ms[i] > 0},
                "bytes_in": {x: int(rep.bytes_in[i]) for i, x in enumerate(xn)},
                "bytes_out": {x: int(rep.bytes_out[i]) for i, x in enumerate(xn)},
                "ctx": st.as_dict()}

    def verify_device(self):
        rep = _lib.DebwtVerifyReport()
        self._chk(self._L.debwt_multi_verify(self._h, ctypes.byref(rep)))
        return rep.as_dict()


def verify_inverse(words, n, hash_rows, dollar_row):
    """Inverse BWT by LF walk on the host (the job of the reference's dead LFsearch path)."""
    L = _lib.lib()
    words = np.ascontiguousarray(words, dtype=np.uint64)
    hr = np.ascontiguousarray(hash_rows, dtype=np.uint64)
    nrec = len(hr) + 1
    if len(hr) == 0:
        hr = np.zeros(1, dtype=np.uint64)
    out = np.empty(n, dtype=np.uint8)
    rc = L.debwt_verify_inverse(_p64(words), n, _p64(hr), nrec, int(dollar_row),
                                out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    return rc, out


def write_outputs(path, words, hash_rows, dollar_row):
    """OUT, OUT.#, OUT.$ exactly as src/insertCase3.c:115-131 writes them."""
    np.ascontiguousarray(words, dtype=np.uint64).tofile(path)
    np.ascontiguousarray(hash_rows, dtype=np.uint64).tofile(path + ".#")
    np.array([dollar_row], dtype=np.uint64).tofile(path + ".$")


FASTA_IUPAC_RANDOM = 1


def fasta_text_bound(path):
    """Upper bound of the text length the file can hold, from its size and framing alone (0: not known); debwt_fasta_text_bound."""
    return int(_lib.lib().debwt_fasta_text_bound(str(path).encode()))


def pack_fasta(path, threads=8, iupac_seed=None):
    """Host-only: (words, n, sep, seconds_read, seconds_pack) of a FASTA file in the reference's 2-bit layout."""
    L = _lib.lib()
    pt = _lib.DebwtPackedText()
    err = ctypes.create_string_buffer(256)
    if iupac_seed is None:
        rc = L.debwt_pack_fasta(str(path).encode(), int(threads), ctypes.byref(pt), err, 256)
    else:
        rc = L.debwt_pack_fasta_opts(str(path).encode(), int(threads), FASTA_IUPAC_RANDOM, int(iupac_seed), ctypes.byref(pt), err, 256)
    if rc:
        raise DebwtError(rc, err.value.decode())
    try:
        words = np.ctypeslib.as_array(pt.words, shape=(pt.nwords,)).copy()
        sep = np.ctypeslib.as_array(pt.sep, shape=(pt.nrec,)).copy()
        return words, int(pt.n), sep, pt.seconds_read, pt.seconds_pack
    finally:
        L.debwt_free_packed(ctypes.byref(pt))
