"""One BWT built by several GPUs: k-mer-prefix shards.

Every rank holds the whole 2-bit text (n/4 bytes: all-gathering the text costs 1/32 of an alltoallv of the
64-bit k-mers and removes the halo).  Rank r sorts and classifies the keys of one prefix range -- in several
key ranges, one exchange round each, when they do not fit its HBM at once -- and owns the contiguous BWT rows
of those nodes and their multi-in blocks.  The exchanges of a build whose keys travel ("exchange" mode):

  all_gather  4096-bin k-mer prefix census of every rank's text slice   -> splitters (balanced instance counts,
                                                                           the reference's segCount idea,
                                                                           src/mySort.c:104-110) and all send /
                                                                           receive counts of the key rounds
  all_gather  every shard's range cuts                                  -> the owner table of every round
  ALL_TO_ALL  8-byte k-mers of the slice -> their bucket owners, once per round      (the k-mer bucket exchange)
  all_gather  per-shard counts, then the classification facts (8 B per branching node) -> the red table, identical
                                                                           on every rank
  all_gather  slice SP lengths, then the slice SP symbols (<= 1 B per branching position)
  ALL_TO_ALL  8-byte blue entries of the slice -> the owners of their blocks
  gather      packed BWT row ranges -> shift-merge on rank 0                         (the final concat)

over torch.distributed ("nccl" = RCCL over xGMI on the GPU node, tensors stay in HBM; "gloo" in the tests, where
the ranks may even share one GPU and the tensors are staged through host memory).

"rescan" mode drops the first ALL_TO_ALL: every rank holds the text anyway, so it reads all of it once per key range
and keeps the keys of its ranges in the first radix pass -- n/4 bytes from its own HBM per range instead of
8 n / world bytes over the fabric; everything after the sort is sliced and exchanged as above.  "auto" (the default)
asks the library's cost model (debwt_shard_key_mode) which of the two is cheaper; on one node of up to 8 GPUs that
is "rescan".  "scan" mode has no bulk exchange at all: keys as in "rescan", and the SP code computed in full on every
rank (tests; O(n) text work per GPU).
"""
import ctypes
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .api import DebwtError

SHARD_BINS = 4096
MAX_RANGES = 64            # key ranges (exchange rounds) a shard may need
MODES = ("auto", "exchange", "rescan", "scan")


# ---- pure host logic (tested on CPU) ---------------------------------------------------------------------------

def plan_splitters(hist, world):
    """Cut the 4096 prefix bins into `world` contiguous ranges of (nearly) equal instance counts.
    Returns bins[world+1] (bins[0] = 0, bins[world] = 4096)."""
    hist = np.asarray(hist, dtype=np.uint64)
    cum = np.concatenate([[0], np.cumsum(hist.astype(np.int64))])
    total = int(cum[-1])
    bins = [0]
    for r in range(1, world):
        target = total * r // world
        b = int(np.searchsorted(cum, target, side="left"))
        b = min(max(b, bins[-1]), SHARD_BINS)
        bins.append(b)
    bins.append(SHARD_BINS)
    return bins, cum


def plan_round(hists, cuts, t, rank):
    """Exchange round t.  hists: (world, 4096) censuses of the ranks' text slices; cuts[s]: bin bounds of shard s's
    key ranges.  Returns (owner table: 4096 bytes, 0xFF = bin not in this round; send counts of `rank` per
    destination; receive counts of `rank` per source)."""
    world = len(cuts)
    tab = np.full(SHARD_BINS, 0xFF, dtype=np.uint8)
    send = [0] * world
    for s in range(world):
        if t + 1 < len(cuts[s]):
            lo, hi = int(cuts[s][t]), int(cuts[s][t + 1])
            tab[lo:hi] = s
            send[s] = int(hists[rank, lo:hi].sum())
    recv = [0] * world
    if t + 1 < len(cuts[rank]):
        lo, hi = int(cuts[rank][t]), int(cuts[rank][t + 1])
        recv = [int(hists[src, lo:hi].sum()) for src in range(world)]
    return tab, send, recv


def concat_rows(parts, n):
    """Host restatement of the final concat (the device version is debwt_concat_rows).  parts: list of
    (row_base, rows, words) with words packed from the shard's first row (row j at bit 2*(31-(j&31)) of word j>>5).
    Returns the ceil(n/32) words of the whole BWT (src/insertCase3.c:115-119)."""
    out = np.zeros((n + 31) // 32 + 1, dtype=np.uint64)
    for base, rows, words in parts:
        if rows == 0:
            continue
        w = np.asarray(words[:(rows + 31) // 32], dtype=np.uint64)
        sh = np.uint64(2 * (base & 31))
        w0 = base >> 5
        if sh == 0:
            out[w0:w0 + len(w)] |= w
        else:
            out[w0:w0 + len(w)] |= w >> sh
            out[w0 + 1:w0 + 1 + len(w)] |= w << (np.uint64(64) - sh)
    return out[:(n + 31) // 32]


# ---- collectives: device tensors over RCCL, or staged through the host for gloo ---------------------------------------

def _chk(d, rc):
    if rc:
        raise DebwtError(rc, _lib.lib().debwt_last_error(d._h).decode())


def _on_device():
    return dist.get_backend() == "nccl"


# RCCL (2.26.6, ROCm 7.0) silently drops the second half of any point-to-point message above 1 GiB -- measured on
# the GPU box in a process group of one (scripts/gpu_nccl_probe.py: all_to_all_single, all_to_all, batch_isend_irecv
# all lose elements [n/2, n) once n * 8 > 2^30; all_gather_into_tensor does not).  all_to_all and gather are built
# on send/recv, and a 30 Gbp build moves up to 3.75 GB per peer: every collective here is cut into calls of at most
# P2P_MAX bytes per peer message.
P2P_MAX = int(os.environ.get("DEBWT_P2P_MAX_BYTES", str(1 << 29)))
# sustained one-direction rate of one GPU-to-GPU link for the key-path cost model (0: the library's default)
LINK_GBYTES_PER_S = float(os.environ.get("DEBWT_LINK_GBYTES_PER_S", "0"))


def measure_link(device, mib_per_peer=1024, reps=2):
    """What a key exchange would run at on THIS node, measured instead of assumed (the cost model's link rate,
    debwt_shard_key_mode): an all_to_all of `mib_per_peer` MiB to every peer -- cut into P2P_MAX calls exactly like the
    exchanges of a build -- timed after one warm-up; the sustained one-direction rate per peer pair becomes
    LINK_GBYTES_PER_S unless DEBWT_LINK_GBYTES_PER_S fixes it.  With DEBWT_BIG_MESSAGE_PROBE=1 it also re-probes, between
    real peers, the RCCL message limit found in a group of one (a single message above 1 GiB arrives with its second half
    zeroed): one 1.25 GiB message per peer, checked at the receiver.  Collective: every rank calls it."""
    global LINK_GBYTES_PER_S
    world, rank = dist.get_world_size(), dist.get_rank()
    if world < 2:
        return None
    dev = device if _on_device() else torch.device("cpu")
    on_gpu = torch.device(device).type == "cuda"                # gloo tests may run it on host tensors
    sync = (lambda: torch.cuda.synchronize(device)) if on_gpu else (lambda: None)
    per = (mib_per_peer << 20) // 8
    src = torch.empty(per * world, dtype=torch.int64, device=device)
    dst = torch.empty(per * world, dtype=torch.int64, device=device)
    src.fill_(rank + 1)
    counts = [per] * world
    times = []
    for _ in range(reps + 1):
        sync()
        dist.barrier()
        t0 = time.perf_counter()
        _all_to_all(dst, src, counts, counts)
        sync()
        times.append(time.perf_counter() - t0)
    ok = all(int(dst[r * per]) == r + 1 and int(dst[(r + 1) * per - 1]) == r + 1 for r in range(world))
    best = min(times[1:])
    t = torch.tensor([best], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    best = float(t.item())
    per_peer = per * 8 / best / 1e9                                   # one direction, one peer pair
    res = {"all_to_all_mib_per_peer": mib_per_peer, "seconds": round(best, 4), "gbytes_per_s_per_peer": round(per_peer, 4),
           "gbytes_per_s_out_of_one_gpu": round(per_peer * (world - 1), 4), "content_ok": bool(ok),
           "calls_of_at_most_bytes": P2P_MAX}
    # one message of 1.25 GiB per peer in ONE call (the exchanges never do this: they stay below P2P_MAX).  Opt-in
    # (DEBWT_BIG_MESSAGE_PROBE=1): a library that mishandles such a message must not be able to take a bench run down
    if _on_device() and os.environ.get("DEBWT_BIG_MESSAGE_PROBE") == "1":
        try:
            big = (5 << 28) // 8
            a = torch.empty(big * world, dtype=torch.int64, device=device)
            b = torch.zeros(big * world, dtype=torch.int64, device=device)
            a.fill_(7)
            dist.all_to_all([b[r * big:(r + 1) * big] for r in range(world)], [a[r * big:(r + 1) * big] for r in range(world)])
            torch.cuda.synchronize(device)
            peer = (rank + 1) % world
            intact = bool((b[peer * big + big // 2:(peer + 1) * big] == 7).all().item())
            v = torch.tensor([1 if intact else 0], dtype=torch.int64, device=device)
            dist.all_reduce(v, op=dist.ReduceOp.MIN)
            res["message_above_1GiB_intact_between_peers"] = bool(v.item())
            del a, b
        except Exception as e:          # noqa: BLE001 -- a probe: report, do not fail the build
            res["message_above_1GiB_intact_between_peers"] = f"probe failed: {e}"
    del src, dst
    if on_gpu:
        torch.cuda.empty_cache()
    if not os.environ.get("DEBWT_LINK_GBYTES_PER_S") and _on_device():
        LINK_GBYTES_PER_S = per_peer
        res["fed_to_cost_model"] = True
    else:
        res["fed_to_cost_model"] = False
    return res


def _global_max(value):
    return int(_all_gather_small([int(value)]).max())


def _all_to_all(dst, src, recv_counts, send_counts):
    """dst[:sum(recv)] <- all_to_all of src[:sum(send)] (1-D tensors on the GPU), in rounds of at most P2P_MAX bytes
    per peer message.  Returns the number of elements received."""
    world = dist.get_world_size()
    ch = max(1, P2P_MAX // src.element_size())
    rounds = max(1, -(-_global_max(max(list(send_counts) + list(recv_counts))) // ch))
    soff = np.concatenate([[0], np.cumsum(send_counts)]).astype(np.int64)
    roff = np.concatenate([[0], np.cumsum(recv_counts)]).astype(np.int64)
    for c in range(rounds):
        ins = [src[int(soff[i]) + min(c * ch, send_counts[i]):int(soff[i]) + min((c + 1) * ch, send_counts[i])] for i in range(world)]
        outs = [dst[int(roff[i]) + min(c * ch, recv_counts[i]):int(roff[i]) + min((c + 1) * ch, recv_counts[i])] for i in range(world)]
        if _on_device():
            dist.all_to_all(outs, ins)                       # grouped ncclSend / ncclRecv on views: no staging
        else:
            ssz, rsz = [int(t.numel()) for t in ins], [int(t.numel()) for t in outs]
            h = torch.empty(sum(rsz), dtype=src.dtype)
            dist.all_to_all_single(h, torch.cat([t.cpu() for t in ins]) if sum(ssz) else torch.empty(0, dtype=src.dtype),
                                   output_split_sizes=rsz, input_split_sizes=ssz)
            o = 0
            for i in range(world):
                outs[i].copy_(h[o:o + rsz[i]])
                o += rsz[i]
    return int(roff[-1])


def _all_gather_small(values, dtype=torch.int64):
    """all_gather of a short list of numbers per rank -> numpy (world, len)."""
    world = dist.get_world_size()
    dev = torch.device("cuda", torch.cuda.current_device()) if _on_device() else torch.device("cpu")
    t = torch.tensor([int(v) for v in values], dtype=dtype, device=dev)
    out = torch.empty(world * t.numel(), dtype=dtype, device=dev)
    dist.all_gather_into_tensor(out, t)
    return out.cpu().numpy().reshape(world, -1)


def _all_gather_var(ws, name, part, count, counts):
    """all_gather of `count` leading elements of the 1-D GPU tensor `part` (counts: every rank's count), in calls of
    at most P2P_MAX bytes per rank.  Returns (the concatenation in rank order as a GPU tensor, its length)."""
    world = dist.get_world_size()
    cap = max(max(counts), 1)
    ch = max(1, min(cap, P2P_MAX // part.element_size()))
    total = sum(counts)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    out = ws.get(name + "_cat", max(total, 1), part.dtype)
    send = ws.get(name + "_send", ch, part.dtype)
    recv = ws.get(name + "_recv", ch * world, part.dtype)
    for o in range(0, cap, ch):
        m = min(ch, cap - o)
        k = max(0, min(m, count - o))
        if k:
            send[:k].copy_(part[o:o + k])
        if _on_device():
            dist.all_gather_into_tensor(recv[:m * world], send[:m])
        else:
            h = torch.empty(m * world, dtype=part.dtype)
            dist.all_gather_into_tensor(h, send[:m].cpu())
            recv[:m * world].copy_(h)
        for r in range(world):
            kr = max(0, min(m, counts[r] - o))
            if kr:
                out[int(offs[r]) + o:int(offs[r]) + o + kr].copy_(recv[r * m:r * m + kr])
    return out, total


class Workspace:
    """Exchange buffers of one rank: GPU tensors that grow on demand and live across builds, so that a steady-state
    build allocates nothing and the library's own buffers never compete with a caching allocator's leftovers."""

    def __init__(self, d, device=None, mode="auto"):
        if mode not in MODES:
            raise ValueError(f"mode {mode!r}: one of {MODES}")
        self.d, self.mode = d, mode
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.buf = {}
        self.blue_bufs = None       # (send, receive) buffers of the last blue-entry exchange
        self.result = None          # rank 0 after a build: (words tensor on the GPU, hash_rows, dollar_row)
        self.built = False          # build_sharded ran (on every rank: the final concat is a collective)
        self.n = d.n

    def held_bytes(self):
        return sum(t.numel() * t.element_size() for t in self.buf.values())

    def get(self, name, numel, dtype):
        t = self.buf.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype:
            self.buf.pop(name, None)
            t = None
            torch.cuda.empty_cache()
            t = torch.empty(int(numel + numel // 16 + 64), dtype=dtype, device=self.device)
            self.buf[name] = t
        return t


class _DevPtr:
    """A device buffer of the library seen through __cuda_array_interface__ (torch.as_tensor wraps it without a copy)."""

    def __init__(self, ptr, numel):
        self.__cuda_array_interface__ = {"shape": (int(numel),), "typestr": "<i8", "data": (int(ptr), False), "version": 2}


def _scratch_tensor(d, which, numel, device):
    """The context's free key buffer `which` (0 send, 1 receive; debwt_shard_scratch) as an int64 tensor of all its words,
    or None when there is none of at least `numel` words."""
    p, nbytes = ctypes.c_void_p(), ctypes.c_uint64()
    _chk(d, _lib.lib().debwt_shard_scratch(d._h, which, ctypes.byref(p), ctypes.byref(nbytes)))
    if not p.value or nbytes.value < numel * 8 or os.environ.get("DEBWT_NO_SCRATCH_ALIAS"):
        return None
    return torch.as_tensor(_DevPtr(p.value, nbytes.value // 8), device=device)


def generate_text_all_gather(syn, text, device):
    """Every rank packs 1/world of the text words with the native generator; the pieces are all-gathered (over RCCL:
    through HBM) and land in the page-locked host array `text` (synth_native.PinnedArray).  Returns the base census."""
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (syn.nwords + world - 1) // world
    w0, w1 = min(rank * per, syn.nwords), min((rank + 1) * per, syn.nwords)
    mine = np.zeros(per, dtype=np.uint64)
    census = syn.words_into(mine.ctypes.data, w0, w1) if w1 > w0 else np.zeros(4, dtype=np.uint64)
    dev = device if _on_device() else torch.device("cpu")
    host = torch.from_numpy(text.a.view(np.int64))
    mine_t = torch.from_numpy(mine.view(np.int64))
    ch = max(1, min(per, P2P_MAX // 8))
    for o in range(0, per, ch):
        m = min(ch, per - o)
        full = torch.empty(m * world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(full, mine_t[o:o + m].to(dev))
        for r in range(world):
            lo = r * per + o
            hi = min(lo + m, syn.nwords)
            if hi > lo:
                host[lo:hi].copy_(full[r * m:r * m + (hi - lo)])
        del full
    c = torch.from_numpy(census.astype(np.int64)).to(dev)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    torch.cuda.empty_cache()
    return c.cpu().numpy().astype(np.uint64)


# ---- the sharded build -------------------------------------------------------------------------------------------

def build_sharded(d, ws=None, mode=None, device=None):
    """Run the sharded stage sequence on this rank's context `d` (text already loaded, the same text on every rank)
    and concatenate the shards' rows on rank 0 (ws.result).  Returns a dict of what the exchanges moved and took."""
    if ws is None:
        ws = Workspace(d, device, mode or "auto")
    L = _lib.lib()
    rank, world = dist.get_rank(), dist.get_world_size()
    device = ws.device
    est_x, est_r = ctypes.c_double(), ctypes.c_double()
    choice = L.debwt_shard_key_mode(d.n, world, LINK_GBYTES_PER_S, ctypes.byref(est_x), ctypes.byref(est_r))
    if ws.mode == "auto" and world > 1:
        # the link rate behind the choice is measured per rank (measure_link) and a probe may fail on one rank alone: every
        # rank takes rank 0's choice, or the ranks would issue different collectives and hang
        choice = int(_all_gather_small(np.array([choice], dtype=np.int64))[0, 0])
    keys = ws.mode if ws.mode != "auto" else ("exchange" if choice == 0 else "rescan")
    exchange = keys == "exchange"               # the keys travel
    sliced = keys != "scan"                     # SP code and blue entries by text slice, then exchanged
    u64p, u8p, u32p = (ctypes.POINTER(t) for t in (ctypes.c_uint64, ctypes.c_uint8, ctypes.c_uint32))
    sync = lambda: torch.cuda.synchronize(device)            # torch's stream <-> the context's own stream
    info = {"key_exchange_ms": 0.0, "key_exchange_GB": 0.0, "blue_exchange_ms": 0.0, "blue_exchange_GB": 0.0,
            "facts_sp_gather_ms": 0.0, "concat_ms": 0.0, "rounds": 0, "keys": keys,
            "model_ms": {"exchange": round(est_x.value, 1), "rescan": round(est_r.value, 1)}}

    # 1. census of the slices -> splitters over the shards, key ranges inside every shard
    _chk(d, L.debwt_shard_begin(d._h, rank, world))
    hist = np.zeros(SHARD_BINS, dtype=np.uint64)
    _chk(d, L.debwt_shard_histogram(d._h, hist.ctypes.data_as(u64p)))
    hists = _all_gather_small(hist.astype(np.int64))                           # (world, 4096)
    total = hists.sum(axis=0).astype(np.uint64)
    bins, cum = plan_splitters(total, world)
    nr = ctypes.c_uint32()
    _chk(d, L.debwt_shard_plan(d._h, np.ascontiguousarray(total).ctypes.data_as(u64p), bins[rank], bins[rank + 1],
                               int(cum[bins[rank]]), 1 if exchange else (2 if sliced else 0), ws.held_bytes(),
                               ctypes.byref(nr)))
    bounds = np.zeros(MAX_RANGES + 1, dtype=np.uint32)
    mkeys = np.zeros(MAX_RANGES, dtype=np.uint64)
    _chk(d, L.debwt_shard_ranges(d._h, bounds.ctypes.data_as(u32p), mkeys.ctypes.data_as(u64p), MAX_RANGES))
    allcuts = _all_gather_small([nr.value] + bounds.tolist())
    cuts = [allcuts[s, 1:2 + int(allcuts[s, 0])] for s in range(world)]
    rounds = max(int(allcuts[s, 0]) for s in range(world))
    info["rounds"] = rounds

    if exchange:
        # 2. the k-mer bucket exchange, one round per key range: keys of my text slice -> their owners -> local sort
        _chk(d, L.debwt_shard_sort_begin(d._h))
        for t in range(rounds):
            tab, send, recv = plan_round(hists, cuts, t, rank)
            ns, nrv = sum(send), sum(recv)
            xa = ws.get("xa", ns + 64, torch.int64)
            offs = np.zeros(world + 1, dtype=np.uint64)
            _chk(d, L.debwt_shard_partition_keys(d._h, tab.ctypes.data_as(u8p), ctypes.c_void_p(xa.data_ptr()),
                                                 xa.numel(), offs.ctypes.data_as(u64p)))
            if [int(offs[i + 1] - offs[i]) for i in range(world)] != send:
                raise RuntimeError("key partition differs from the census")
            xb = ws.get("xb", nrv + 64, torch.int64)
            t0 = time.perf_counter()
            _all_to_all(xb, xa, recv, send)
            sync()
            info["key_exchange_ms"] += (time.perf_counter() - t0) * 1e3
            info["key_exchange_GB"] += 8e-9 * (ns - send[rank])
            if t + 1 < len(cuts[rank]):
                _chk(d, L.debwt_shard_sort_range(d._h, t, ctypes.c_void_p(xb.data_ptr()), nrv))
        _chk(d, L.debwt_shard_sort_end(d._h))
    else:
        _chk(d, L.debwt_kmer_sort_rle(d._h))

    # 3. local classification totals, red table from everybody's facts
    nf, nb, br = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    _chk(d, L.debwt_shard_classify_local(d._h, ctypes.byref(nf), ctypes.byref(nb), ctypes.byref(br)))
    allc = _all_gather_small([nf.value, nb.value, br.value])
    first_block = np.concatenate([[0], np.cumsum(allc[:, 1])]).astype(np.uint32)
    qbase, blue_total = int(first_block[rank]), int(allc[:, 2].sum())
    t0 = time.perf_counter()
    myfacts = ws.get("facts", max(nf.value, 1), torch.int64)
    _chk(d, L.debwt_shard_facts_export(d._h, ctypes.c_void_p(myfacts.data_ptr()), myfacts.numel()))
    facts, nfacts = _all_gather_var(ws, "facts", myfacts, nf.value, [int(x) for x in allc[:, 0]])
    sync()
    info["facts_sp_gather_ms"] += (time.perf_counter() - t0) * 1e3
    _chk(d, L.debwt_shard_classify_global(d._h, ctypes.c_void_p(facts.data_ptr()), nfacts, qbase, blue_total))

    if sliced:
        # 4. SP code: flags of my slice, offsets from the slice lengths (the reference's spSplit prefix,
        #    src/generateSP.c:152-157), symbols all-gathered
        s_loc, b_loc = ctypes.c_uint64(), ctypes.c_uint64()
        _chk(d, L.debwt_shard_sp_flags(d._h, ctypes.byref(s_loc), ctypes.byref(b_loc)))
        lens = [int(x) for x in _all_gather_small([s_loc.value])[:, 0]]
        sp_off, sp_total = sum(lens[:rank]), sum(lens)
        mysp = ws.get("sp", max(s_loc.value, 1), torch.uint8)
        _chk(d, L.debwt_shard_sp_emit(d._h, sp_off, ctypes.c_void_p(mysp.data_ptr()), mysp.numel()))
        t0 = time.perf_counter()
        allsp, got_sp = _all_gather_var(ws, "sp", mysp, s_loc.value, lens)
        sync()
        info["facts_sp_gather_ms"] += (time.perf_counter() - t0) * 1e3
        if got_sp != sp_total:
            raise RuntimeError(f"gathered {got_sp} SP symbols, the slices announced {sp_total}")
        _chk(d, L.debwt_shard_sp_import(d._h, ctypes.c_void_p(allsp.data_ptr()), sp_total))

        # 5. blue entries of my slice -> the owners of their blocks.  Send and receive buffer: the context's own key buffers
        #    where they are free and large enough (keys read off the text; debwt_shard_scratch), else tensors of this workspace
        xa = _scratch_tensor(d, 0, b_loc.value + 64, device) if not exchange else None
        if xa is None:
            xa = ws.get("xa", b_loc.value + 64, torch.int64)
        boffs = np.zeros(world + 1, dtype=np.uint64)
        _chk(d, L.debwt_shard_blue_route(d._h, first_block.ctypes.data_as(u32p), ctypes.c_void_p(xa.data_ptr()),
                                         xa.numel(), boffs.ctypes.data_as(u64p)))
        send = [int(boffs[i + 1] - boffs[i]) for i in range(world)]
        recv = [int(x) for x in _all_gather_small(send)[:, rank]]
        xb = _scratch_tensor(d, 1, sum(recv) + 64, device) if not exchange else None     # (its routed entries are in xa now)
        if xb is None:
            xb = ws.get("xb", sum(recv) + 64, torch.int64)
        ws.blue_bufs = (xa, xb)                  # the final concat reuses them (both are free once the entries are placed)
        t0 = time.perf_counter()
        got = _all_to_all(xb, xa, recv, send)
        sync()
        info["blue_exchange_ms"] += (time.perf_counter() - t0) * 1e3
        info["blue_exchange_GB"] += 8e-9 * (sum(send) - send[rank])
        if os.environ.get("DEBWT_DEBUG"):
            print(f"[shard {rank}] rounds {rounds} cuts {[c.tolist() for c in cuts]} facts {nf.value} blocks {nb.value} "
                  f"owned_rows {br.value} first_block {first_block.tolist()} slice S {s_loc.value} mi {b_loc.value} "
                  f"sent {send} got {got}", flush=True)
        _chk(d, L.debwt_shard_blue_place(d._h, ctypes.c_void_p(xb.data_ptr()), got))
    else:
        _chk(d, L.debwt_sp_generate(d._h))

    # 6. owned blocks and rows
    _chk(d, L.debwt_blue_sort(d._h))
    _chk(d, L.debwt_bwt_assemble(d._h))

    # 7. final concat on rank 0: gather of the packed row ranges, shift-merge by row offset
    t0 = time.perf_counter()
    ws.result = _concat_on_rank0(d, ws)
    ws.built = True
    info["concat_ms"] = (time.perf_counter() - t0) * 1e3
    return info


def _shard_rows(d):
    L = _lib.lib()
    rb, rows, nh = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    _chk(d, L.debwt_shard_info(d._h, ctypes.byref(rb), ctypes.byref(rows), ctypes.byref(nh)))
    return rb.value, rows.value, nh.value


def _concat_on_rank0(d, ws, dst=0):
    L = _lib.lib()
    u64p = ctypes.POINTER(ctypes.c_uint64)
    rank, world = dist.get_rank(), dist.get_world_size()
    n = d.n
    rb, rows, nh = _shard_rows(d)
    allr = _all_gather_small([rb, rows, nh])
    maxw = int(max((int(r) + 31) // 32 for r in allr[:, 1])) + 1               # one spare word behind every part
    # after the exchanges the key buffers are free: the shard's rows leave from xb, the parts arrive in xa
    xa, xb = getattr(ws, "blue_bufs", None) or (ws.buf.get("xa"), ws.buf.get("xb"))
    mine = xb if xb is not None and xb.numel() >= maxw else ws.get("part", maxw, torch.int64)
    _chk(d, L.debwt_shard_export(d._h, ctypes.c_void_p(mine.data_ptr()), maxw))
    torch.cuda.synchronize(ws.device)
    # '#' rows and the '$' row: short lists, global row numbers
    hr = np.zeros(max(nh, 1), dtype=np.uint64)
    dr = np.zeros(1, dtype=np.uint64)
    _chk(d, L.debwt_shard_fetch(d._h, None, hr.ctypes.data_as(u64p), dr.ctypes.data_as(u64p)))
    maxh = int(allr[:, 2].max()) + 1
    lists = _all_gather_small([int(dr[0]) if dr[0] != 0xFFFFFFFFFFFFFFFF else -1] + hr[:nh].astype(np.int64).tolist()
                              + [0] * (maxh - 1 - nh))
    # gather of the packed row ranges as an all_to_all whose only receiver is `dst` (RCCL's gather is send/recv too)
    need = maxw * world if rank == dst else 1
    parts = xa if xa is not None and xa.numel() >= need else ws.get("parts", need, torch.int64)
    send = [maxw if i == dst else 0 for i in range(world)]
    recv = [maxw] * world if rank == dst else [0] * world
    _all_to_all(parts, mine, recv, send)
    if rank != dst:
        return None
    torch.cuda.synchronize(ws.device)
    out = ws.get("out", (n + 31) // 32 + 1, torch.int64)
    offs = np.arange(world, dtype=np.uint64) * np.uint64(maxw)
    base = np.ascontiguousarray(allr[:, 0].astype(np.uint64))
    rws = np.ascontiguousarray(allr[:, 1].astype(np.uint64))
    _chk(d, L.debwt_concat_rows(d._h, ctypes.c_void_p(parts.data_ptr()), world, offs.ctypes.data_as(u64p),
                                base.ctypes.data_as(u64p), rws.ctypes.data_as(u64p), n, ctypes.c_void_p(out.data_ptr())))
    hrows = np.sort(np.concatenate([lists[r, 1:1 + int(allr[r, 2])] for r in range(world)])).astype(np.uint64)
    dollars = [int(lists[r, 0]) for r in range(world) if lists[r, 0] >= 0]
    if len(dollars) != 1:
        raise RuntimeError(f"{len(dollars)} shards report the '$' row: exactly one holds it")
    return out[:(n + 31) // 32], hrows, dollars[0]


def gather_bwt(d, n, ws=None, dst=0):
    """(words, hash_rows, dollar_row) as host arrays on rank `dst` (None elsewhere) -- the contents of OUT, OUT.#,
    OUT.$ (src/insertCase3.c:115-131) of the last sharded build."""
    res = ws.result if ws is not None and ws.built else _concat_on_rank0(d, ws or Workspace(d), dst)
    if res is None:
        return None
    out, hrows, dollar = res
    return out.cpu().numpy().view(np.uint64), hrows, dollar


def check_result(d, ws, census, n, nrec):
    """Outside the timed region (rank 0 holds the result): symbol census of the concatenated BWT against the text's,
    '#' rows ascending and complete."""
    L = _lib.lib()
    if ws.result is None:
        return None
    out, hrows, dollar = ws.result
    got = np.zeros(4, dtype=np.uint64)
    _chk(d, L.debwt_census_words(d._h, ctypes.c_void_p(out.data_ptr()), n, got.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))))
    want = census.astype(np.int64).copy()
    want[3] += nrec
    res = {"census_equals_text": bool((got.astype(np.int64) == want).all()),
           "hash_rows_ascending": bool((np.diff(hrows.astype(np.int64)) > 0).all()) if nrec > 2 else True,
           "hash_rows": int(len(hrows)), "dollar_row": int(dollar)}
    # the inverse BWT of the concatenated rows against the text this rank holds (one LF walk per text segment)
    res.update(d.verify_device(out.data_ptr(), hrows, dollar))
    return res
