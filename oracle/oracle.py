"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/debwt_oracle.h).  The product path (debwt_amd) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OrcStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in (
        "n", "nrec", "distinct_kmers", "red_capacity", "blue_capacity", "blue_bound_num",
        "case3num", "sp_len", "special_branch_num", "cmp_calls")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build(force=False):
    """Compile liboracle.so (and, where /root/reference exists, oracle/_ref/ref_driver)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("debwt_oracle.c", "debwt_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def build_ref():
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    p = os.path.join(_HERE, "_ref", "ref_driver")
    return p if os.path.exists(p) else None


def lib():
    global _LIB
    if _LIB is None:
        # DEBWT_ORACLE_LIB: another build of the same source (the ASan + UBSan one of tests/sanitize/Makefile)
        L = ctypes.CDLL(os.environ.get("DEBWT_ORACLE_LIB") or build())
        u8p, u64p = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint64)
        L.orc_make_text.restype = ctypes.c_uint64
        L.orc_make_text.argtypes = [ctypes.c_char_p, u64p, ctypes.c_uint64, u8p]
        L.orc_pack_text.restype = None
        L.orc_pack_text.argtypes = [u8p, ctypes.c_uint64, u64p]
        L.orc_kmer_count.restype = ctypes.c_uint64
        L.orc_kmer_count.argtypes = [u8p, ctypes.c_uint64, ctypes.c_int, u64p, u64p]
        L.orc_build_bwt_ex.restype = ctypes.c_int
        L.orc_build_bwt_ex.argtypes = [u8p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, u64p, u64p, u64p,
                                       ctypes.POINTER(OrcStats), u8p, u64p]
        L.orc_naive_bwt.restype = None
        L.orc_naive_bwt.argtypes = [u8p, ctypes.c_uint64, u8p]
        L.orc_unpack_bwt.restype = None
        L.orc_unpack_bwt.argtypes = [u64p, ctypes.c_uint64, u64p, ctypes.c_uint64, ctypes.c_uint64, u8p]
        L.orc_inverse_bwt.restype = ctypes.c_int
        L.orc_inverse_bwt.argtypes = [u64p, ctypes.c_uint64, u64p, ctypes.c_uint64, ctypes.c_uint64, u8p]
        _LIB = L
    return _LIB


def _p8(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def _p64(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


def make_text(records):
    """records: list of str/bytes (ACGT).  Returns the symbol array r0#r1#...$ (uint8 0..5)."""
    recs = [r.encode() if isinstance(r, str) else bytes(r) for r in records]
    seq = b"".join(recs)
    reclen = np.array([len(r) for r in recs], dtype=np.uint64)
    sym = np.empty(len(seq) + len(recs), dtype=np.uint8)
    n = lib().orc_make_text(seq, _p64(reclen), len(recs), _p8(sym))
    if n == 0:
        raise ValueError("invalid input: a record of <= 32 bases or a non-ACGT letter")
    return sym


def sym_from_codes(codes_list):
    """codes_list: list of uint8 arrays with values 0..3.  Returns the symbol array."""
    parts = []
    for i, c in enumerate(codes_list):
        parts.append(np.asarray(c, dtype=np.uint8))
        parts.append(np.array([5 if i + 1 == len(codes_list) else 4], dtype=np.uint8))
    return np.concatenate(parts)


def pack_text(sym):
    n = len(sym)
    words = np.zeros((n + 32 + 31) // 32, dtype=np.uint64)
    lib().orc_pack_text(_p8(sym), n, _p64(words))
    return words


def kmer_count(sym, k):
    n = len(sym)
    km = np.empty(n, dtype=np.uint64)
    ct = np.empty(n, dtype=np.uint64)
    d = lib().orc_kmer_count(_p8(sym), n, k, _p64(km), _p64(ct))
    return km[:d].copy(), ct[:d].copy()


def build_bwt(sym, k=32, threads=1, want_intermediates=False):
    """Returns (bwt_words, hash_rows, dollar_row, stats[, sp_sym, red_nodes])."""
    sym = np.ascontiguousarray(sym, dtype=np.uint8)
    n = len(sym)
    nrec = int(np.count_nonzero(sym >= 4))
    words = np.zeros((n + 31) // 32, dtype=np.uint64)
    hrows = np.zeros(max(nrec - 1, 1), dtype=np.uint64)
    drow = np.zeros(1, dtype=np.uint64)
    st = OrcStats()
    sp = np.zeros(n if want_intermediates else 1, dtype=np.uint8)
    red = np.zeros(n if want_intermediates else 1, dtype=np.uint64)
    rc = lib().orc_build_bwt_ex(_p8(sym), n, k, threads, _p64(words), _p64(hrows), _p64(drow),
                                ctypes.byref(st), _p8(sp) if want_intermediates else None,
                                _p64(red) if want_intermediates else None)
    if rc != 0:
        raise RuntimeError(f"orc_build_bwt failed: {rc}")
    out = (words, hrows[:nrec - 1], int(drow[0]), st.as_dict())
    if want_intermediates:
        out += (sp[:st.sp_len].copy(), red[:st.red_capacity].copy())
    return out


def naive_bwt(sym):
    sym = np.ascontiguousarray(sym, dtype=np.uint8)
    out = np.empty(len(sym), dtype=np.uint8)
    lib().orc_naive_bwt(_p8(sym), len(sym), _p8(out))
    return out


def unpack_bwt(words, n, hash_rows, dollar_row):
    out = np.empty(n, dtype=np.uint8)
    hr = np.ascontiguousarray(hash_rows, dtype=np.uint64)
    if len(hr) == 0:
        hr = np.zeros(1, dtype=np.uint64)
        nrec = 1
    else:
        nrec = len(hash_rows) + 1
    lib().orc_unpack_bwt(_p64(np.ascontiguousarray(words, dtype=np.uint64)), n, _p64(hr), nrec,
                         int(dollar_row), _p8(out))
    return out


def inverse_bwt(words, n, hash_rows, dollar_row):
    out = np.empty(n, dtype=np.uint8)
    hr = np.ascontiguousarray(hash_rows, dtype=np.uint64)
    nrec = len(hash_rows) + 1
    if len(hr) == 0:
        hr = np.zeros(1, dtype=np.uint64)
    rc = lib().orc_inverse_bwt(_p64(np.ascontiguousarray(words, dtype=np.uint64)), n, _p64(hr), nrec,
                               int(dollar_row), _p8(out))
    return rc, out
