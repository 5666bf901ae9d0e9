"""One BWT built by several GPUs: k-mer-prefix shards.

Every rank holds the whole 2-bit text (n/4 bytes: all-gathering the text costs 1/32 of an alltoallv of the
64-bit k-mers and removes the halo).  Rank r sorts and classifies the keys of one prefix range, owns the
contiguous BWT rows of those nodes and their multi-in blocks.  The exchanges are

  all_reduce  4096-bin k-mer prefix census             -> splitters (balanced instance counts)
  all_gather  per-shard counts (facts, blocks, rows)   -> offsets
  all_gather  classification facts (8 bytes per branching node)  -> the red table, identical on every rank
  gather      packed BWT row ranges + '#' rows         -> final concatenation by row on rank 0

over torch.distributed ("nccl" = RCCL over xGMI on the GPU node; "gloo" in the tests, where the ranks may even
share one GPU).  The SP stage (text scan) is replicated on every rank in this version: each rank scans the
whole text and keeps the blue entries of the blocks it owns, so no alltoallv is needed yet.
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .api import DebwtError

SHARD_BINS = 4096


def plan_splitters(hist, world):
    """Cut the 4096 prefix bins into `world` contiguous ranges of (nearly) equal instance counts.
    Returns bins[world+1] (bins[0] = 0, bins[world] = 4096).  Pure host logic (tested on CPU)."""
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


def concat_rows(parts, n):
    """parts: list of (row_base, rows, words) with words packed from the shard's first row (row j at bit
    2*(31-(j&31)) of word j>>5).  Returns the ceil(n/32) words of the whole BWT (src/insertCase3.c:115-119)."""
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


def _chk(d, rc):
    if rc:
        raise DebwtError(rc, _lib.lib().debwt_last_error(d._h).decode())


def _dev_for_comm(device):
    return device if dist.get_backend() == "nccl" else torch.device("cpu")


def build_sharded(d, device=None):
    """Run the sharded stage sequence on this rank's context `d` (text already loaded, same text on every
    rank).  Returns (row_base, rows) of this shard; results stay in HBM until fetch_shard()."""
    L = _lib.lib()
    rank, world = dist.get_rank(), dist.get_world_size()
    device = device or torch.device("cuda", torch.cuda.current_device())
    cdev = _dev_for_comm(device)
    u64p = ctypes.POINTER(ctypes.c_uint64)

    _chk(d, L.debwt_shard_begin(d._h, rank, world))
    hist = np.zeros(SHARD_BINS, dtype=np.uint64)
    _chk(d, L.debwt_shard_histogram(d._h, hist.ctypes.data_as(u64p)))
    t = torch.from_numpy(hist.astype(np.int64)).to(cdev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)                                  # census -> every rank
    bins, cum = plan_splitters(t.cpu().numpy(), world)
    m_keys = int(cum[bins[rank + 1]] - cum[bins[rank]])
    m_base = int(cum[bins[rank]])
    _chk(d, L.debwt_shard_set_range(d._h, bins[rank], bins[rank + 1], m_keys, m_base))

    _chk(d, L.debwt_kmer_sort_rle(d._h))
    nf, nb, br = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    _chk(d, L.debwt_shard_classify_local(d._h, ctypes.byref(nf), ctypes.byref(nb), ctypes.byref(br)))

    counts = torch.tensor([nf.value, nb.value, br.value], dtype=torch.int64, device=cdev)
    allc = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(allc, counts)
    allc = torch.stack(allc).cpu().numpy()
    nfacts = allc[:, 0]
    qbase = int(allc[:rank, 1].sum())
    blue_total = int(allc[:, 2].sum())

    # all-gather of the fact lists (variable length: padded to the longest)
    cap = max(int(nfacts.max()), 1)
    mine = torch.zeros(cap, dtype=torch.int64, device=device)
    _chk(d, L.debwt_shard_facts_export(d._h, ctypes.c_void_p(mine.data_ptr()), cap))
    send = mine.to(cdev)
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    facts = torch.cat([recv[r][:int(nfacts[r])] for r in range(world)]).to(device).contiguous()
    torch.cuda.synchronize(device)       # torch's stream -> the context's own stream reads `facts` next
    _chk(d, L.debwt_shard_classify_global(d._h, ctypes.c_void_p(facts.data_ptr()), facts.numel(), qbase, blue_total))

    _chk(d, L.debwt_sp_generate(d._h))
    _chk(d, L.debwt_blue_sort(d._h))
    _chk(d, L.debwt_bwt_assemble(d._h))
    rb, rows, nh = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    _chk(d, L.debwt_shard_info(d._h, ctypes.byref(rb), ctypes.byref(rows), ctypes.byref(nh)))
    return rb.value, rows.value, nh.value


def fetch_shard(d):
    L = _lib.lib()
    u64p = ctypes.POINTER(ctypes.c_uint64)
    rb, rows, nh = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    _chk(d, L.debwt_shard_info(d._h, ctypes.byref(rb), ctypes.byref(rows), ctypes.byref(nh)))
    words = np.zeros((rows.value + 31) // 32 + 1, dtype=np.uint64)
    hrows = np.zeros(max(nh.value, 1), dtype=np.uint64)
    drow = np.zeros(1, dtype=np.uint64)
    _chk(d, L.debwt_shard_fetch(d._h, words.ctypes.data_as(u64p), hrows.ctypes.data_as(u64p), drow.ctypes.data_as(u64p)))
    return rb.value, rows.value, words, hrows[:nh.value], int(drow[0])


def gather_bwt(d, n, dst=0):
    """Final concat: every rank ships its packed row range and '#' rows to `dst`; returns
    (words, hash_rows, dollar_row) there, None elsewhere."""
    part = fetch_shard(d)
    world, rank = dist.get_world_size(), dist.get_rank()
    parts = [None] * world if rank == dst else None
    dist.gather_object(part, parts, dst=dst)
    if rank != dst:
        return None
    words = concat_rows([(p[0], p[1], p[2]) for p in parts], n)
    hrows = np.sort(np.concatenate([p[3] for p in parts])) if any(len(p[3]) for p in parts) else np.zeros(0, np.uint64)
    dollars = [p[4] for p in parts if p[4] != 0xFFFFFFFFFFFFFFFF]
    assert len(dollars) == 1, "exactly one shard holds the '$' row"
    assert sum(p[1] for p in parts) == n, "shard rows must add up to n"
    return words, hrows.astype(np.uint64), dollars[0]
