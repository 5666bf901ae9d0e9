"""Randomised parity run against the oracle: random small collections (the generator of tests/test_gpu_parity.py plus
larger repeat-heavy ones, runs of one symbol and tandem repeats), random k, random key-range caps and the alternative device paths (cursor atomics, 64-bit
cursors, no tie-group hand-off, either SP prefilter, no pivot rounds).

    python scripts/gpu_fuzz.py [cases=300] [seed=1] [--oracle-only] [--no-arena]

Integrity harness (round 6; profiles/r05_experiments.txt item 31: a record once held a value above 3 when it was loaded,
after the oracle had consumed it -- somebody wrote into memory it does not own):
  * every buffer the library or the harness must not write is a page-granular mapping of its own whose END sits against a
    PROT_NONE guard page (an overrun faults at the store): records (read-only from their creation on), the packed text and
    separator list (hashed before the load and after the context is destroyed), the three fetch buffers;
  * the packed text and the fetch buffers of a finished case stay mapped behind PROT_NONE for two more cases: a late write
    (a copy still in flight, a host thread that outlived its call) faults in the thread that issues it, and
    scripts/fuzz_guard.c prints that thread's native backtrace;
  * every record is hashed at creation, after the oracle, before the load and after the context is destroyed.
--oracle-only runs the same cases (same generator stream) through the oracle alone: the leg that runs under ASan + UBSan on
the CPU (LD_PRELOAD=libasan.so DEBWT_ORACLE_LIB=tests/sanitize/liboracle_asan.so).
"""
import collections, ctypes, hashlib, mmap, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

args = [a for a in sys.argv[1:] if not a.startswith("--")]
ORACLE_ONLY = "--oracle-only" in sys.argv
ARENA = "--no-arena" not in sys.argv
cases = int(args[0]) if len(args) > 0 else 300
seed = int(args[1]) if len(args) > 1 else 1

from oracle import oracle as O
from debwt_amd import synth
if not ORACLE_ONLY:
    import torch  # noqa: F401  (the ROCm runtime of the torch wheel is the one libdebwt_hip.so binds)
    from debwt_amd import api
else:
    api = None

PAGE = mmap.PAGESIZE
_libc = ctypes.CDLL(None, use_errno=True)
_libc.mmap.restype = ctypes.c_void_p
_libc.mmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long]
_libc.mprotect.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
_libc.munmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
PROT_NONE, PROT_READ, PROT_RW = 0, 1, 3


class Guarded:
    """One buffer in a mapping of its own: [guard page][data pages][guard page], the data flush against the rear guard."""

    def __init__(self, count, dtype):
        dt = np.dtype(dtype)
        self.nbytes = max(int(count), 1) * dt.itemsize
        self.pages = (self.nbytes + PAGE - 1) // PAGE
        self.total = (self.pages + 2) * PAGE
        base = _libc.mmap(None, self.total, PROT_RW, mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS, -1, 0)
        if base in (None, ctypes.c_void_p(-1).value):
            raise MemoryError("mmap")
        self.base = base
        for off in (0, (self.pages + 1) * PAGE):
            if _libc.mprotect(base + off, PAGE, PROT_NONE):
                raise OSError(ctypes.get_errno(), "mprotect")
        self.addr = base + PAGE + self.pages * PAGE - self.nbytes
        raw = (ctypes.c_uint8 * self.nbytes).from_address(self.addr)
        self.arr = np.ctypeslib.as_array(raw).view(dt)[:max(int(count), 1)]
        self.count = int(count)

    def protect(self, prot):
        if _libc.mprotect(self.base + PAGE, self.pages * PAGE, prot):
            raise OSError(ctypes.get_errno(), "mprotect")

    def free(self):
        self.arr = None
        _libc.munmap(self.base, self.total)


def guarded_copy(a, prot=None):
    g = Guarded(a.size, a.dtype)
    g.arr[:a.size] = a.ravel()
    if prot is not None:
        g.protect(prot)
    return g


def digest(a):
    return hashlib.blake2b(np.ascontiguousarray(a).tobytes(), digest_size=8).hexdigest()


def install_guard():
    so = os.path.join("/tmp", f"fuzz_guard_{os.getpid()}.so")
    subprocess.check_call(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "scripts", "fuzz_guard.c")])
    g = ctypes.CDLL(so)
    if g.fuzz_guard_install():
        raise OSError("sigaction")
    return g


def gen_case(rng):
    from test_gpu_parity import _adversarial
    kind = int(rng.integers(0, 5))
    if kind == 4:                            # periodic stretches: the pivot rounds of the large-block split
        parts = []
        for _ in range(int(rng.integers(2, 30))):
            parts.append(np.full(int(rng.integers(50, 6000)), int(rng.integers(0, 4)), dtype=np.uint8))
            parts.append(np.tile(rng.integers(0, 4, size=int(rng.integers(2, 7))).astype(np.uint8), int(rng.integers(20, 800))))
            parts.append(rng.integers(0, 4, size=int(rng.integers(40, 1500))).astype(np.uint8))
        cut = int(rng.integers(1, len(parts)))
        recs = [np.concatenate(parts[:cut]), np.concatenate(parts[cut:])] if rng.integers(0, 2) else [np.concatenate(parts)]
        recs = [r for r in recs if len(r) > 32]
    elif kind == 0:
        recs = _adversarial(rng)
    elif kind == 1:
        recs = synth.pan_genome(int(rng.integers(2000, 60000)), int(rng.integers(1, 9)), seed=int(rng.integers(1, 1 << 30)))
    elif kind == 2:
        unit = rng.integers(0, 4, size=int(rng.integers(34, 60))).astype(np.uint8)
        parts = []
        for _ in range(int(rng.integers(50, 3000))):
            parts.append(unit); parts.append(rng.integers(0, 4, size=int(rng.integers(1, 9))).astype(np.uint8))
        recs = [np.concatenate(parts)] + [rng.integers(0, 4, size=int(rng.integers(33, 500))).astype(np.uint8) for _ in range(int(rng.integers(0, 4)))]
    else:
        recs = [rng.integers(0, 4, size=int(rng.integers(33, 3000))).astype(np.uint8) for _ in range(int(rng.integers(1, 40)))]
    k = int(rng.choice([12, 13, 16, 20, 24, 27, 31, 32]))
    tune = int(rng.choice([0, 0, 0, 32, 48, 128, 160, 256, 2048, 4096, 4096 + 32, 4096 + 6, 8192, 8192 + 128]))
    cap = int(rng.choice([0, 0, 4096, 20000, 300000]))
    special_dev = rng.integers(0, 3) == 0    # special-region module on the device at any size
    return kind, recs, k, tune, cap, special_dev


def main():
    rng = np.random.default_rng(seed)
    if ARENA:
        install_guard()
    retired = collections.deque()            # (case, [Guarded...]) behind PROT_NONE for two more cases
    t0 = time.time(); bad = 0; hash_changes = 0
    print(f"# gpu_fuzz: {cases} cases, seed {seed}, arena {'on' if ARENA else 'off'}, "
          f"{'oracle only' if ORACLE_ONLY else 'HIP against the oracle'}, oracle lib {os.environ.get('DEBWT_ORACLE_LIB', 'oracle/liboracle.so')}",
          flush=True)
    for c in range(cases):
        kind, recs, k, tune, cap, special_dev = gen_case(rng)
        if special_dev: os.environ["DEBWT_SPECIAL_DEVICE_MIN"] = "0"
        else: os.environ.pop("DEBWT_SPECIAL_DEVICE_MIN", None)
        where = f"case {c} kind {kind} k {k} tune {tune} cap {cap}"
        held = []
        if ARENA:                            # records live read-only behind guard pages from here on
            rg = [guarded_copy(r, PROT_READ) for r in recs]
            held += rg
            recs = [g.arr[:g.count] for g in rg]
        h0 = [digest(r) for r in recs]

        def check(stage):
            nonlocal hash_changes
            h = [digest(r) for r in recs]
            if h != h0:
                hash_changes += 1
                print(f"HASH CHANGE {where} at '{stage}': records "
                      + str([(i, len(r), int(r.max()), int((r > 3).sum())) for i, r in enumerate(recs) if h[i] != h0[i]][:10]), flush=True)
                sys.exit(2)

        sym = O.sym_from_codes(recs)
        ow, oh, od, ost = O.build_bwt(sym, k)
        check("after the oracle")
        if ORACLE_ONLY:
            if c % 500 == 499: print(f"{c+1} cases, oracle only, {hash_changes} hash changes, {time.time()-t0:.0f}s", flush=True)
            for g in held: g.free()
            continue
        words, n, sep = api.pack_records(recs)
        if ARENA:
            tw, ts = guarded_copy(words), guarded_copy(sep)
            held += [tw, ts]
            words, sep = tw.arr, ts.arr[:ts.count]
        ht = (digest(words), digest(sep))
        check("before the load")
        d = api.DeBWT(k=k, tune=tune)
        if cap: d.set_range_cap(cap)
        d.load_packed(words, n, sep)
        nrec = len(sep)
        for rep in range(2):                     # a context is reusable
            d.build()
            if ARENA:
                fw, fh, fd = Guarded((n + 31) // 32, np.uint64), Guarded(max(nrec - 1, 1), np.uint64), Guarded(1, np.uint64)
                held += [fw, fh, fd]
                d.fetch_into(fw.arr, fh.arr, fd.arr)
                w, h, dr = fw.arr, fh.arr[:nrec - 1], int(fd.arr[0])
            else:
                w, h, dr = d.fetch()
            ok = np.array_equal(w, ow) and np.array_equal(h, oh) and dr == od
            if not ok:
                bad += 1
                print(f"MISMATCH {where} n {len(sym)} rep {rep}", flush=True)
                break
        d.close()
        d._keep = None
        check("after the context is destroyed")
        if (digest(words), digest(sep)) != ht:
            hash_changes += 1
            print(f"HASH CHANGE {where}: the packed text or the separator list was written", flush=True)
            sys.exit(2)
        if ARENA:
            words = sep = w = h = None
            for g in held:
                g.protect(PROT_NONE)             # anybody still writing this case's buffers faults from now on
            retired.append(held)
            while len(retired) > 2:
                for g in retired.popleft(): g.free()
        if c % 50 == 49:
            print(f"{c+1} cases, {bad} mismatches, {hash_changes} hash changes, 0 faults, {time.time()-t0:.0f}s", flush=True)
    print(f"done: {cases} cases, seed {seed}, {bad} mismatches, {hash_changes} hash changes, 0 faults, {time.time()-t0:.0f}s")
    sys.exit(1 if bad else 0)


main()
