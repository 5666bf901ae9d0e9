"""One BWT built by several GPUs: k-mer-prefix shards.

Every rank holds the whole 2-bit text (n/4 bytes: all-gathering the text costs 1/32 of an alltoallv of the
64-bit k-mers and removes the halo).  Rank r sorts and classifies the keys of one prefix range, owns the
contiguous BWT rows of those nodes and their multi-in blocks.  The exchanges are

  all_reduce  4096-bin k-mer prefix census             -> splitters (balanced instance counts)
  all_gather  per-shard counts (facts, blocks, rows)   -> offsets
  all_gather  classification facts (8 bytes per branching node)  -> the red table, identical on every rank
  gather      packed BWT row ranges + '#' rows         -> final concatenation by row on rank 0

over torch.distributed ("nccl" = RCCL over xGMI on the GPU node; "gloo" in the tests, where the ranks may even
share one GPU).

Two ways to feed a shard (mode=):
  "scan"      every rank scans the whole text: keys outside its prefix range are dropped in the first radix pass
              and the SP stage is computed in full everywhere -- no bulk exchange at all, best for 2-4 GPUs;
  "exchange"  every rank scans only its 1/world slice of the text: its keys go to their owners by
              all_to_all (the k-mer bucket exchange, 8 bytes per base), the slices' SP symbols are all-gathered
              (<= 1 byte per branching position), and the blue entries go to the owners of their blocks by a
              second all_to_all (8 bytes per multi-in position).  Per-rank work is O(n / world).
"""
import ctypes
import os

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


def _all_to_all_var(send, send_offs, cdev, device):
    """Variable all_to_all of int64 words: send[send_offs[i]:send_offs[i+1]] goes to rank i.
    Returns the received words (device tensor) concatenated in rank order."""
    world = dist.get_world_size()
    scount = torch.tensor([int(send_offs[i + 1] - send_offs[i]) for i in range(world)], dtype=torch.int64, device=cdev)
    rcount = torch.empty_like(scount)
    dist.all_to_all_single(rcount, scount)
    ssz, rsz = [int(x) for x in scount.cpu()], [int(x) for x in rcount.cpu()]
    src = send[:int(send_offs[world])].to(cdev).contiguous()
    dst = torch.empty(sum(rsz), dtype=torch.int64, device=cdev)
    dist.all_to_all_single(dst, src, output_split_sizes=rsz, input_split_sizes=ssz)
    return dst.to(device).contiguous()


def _all_gather_var(part, count, dtype, cdev, device):
    """all_gather of variable-length 1-D tensors; returns (concatenation on `device`, counts)."""
    world = dist.get_world_size()
    c = torch.tensor([count], dtype=torch.int64, device=cdev)
    allc = [torch.zeros_like(c) for _ in range(world)]
    dist.all_gather(allc, c)
    counts = [int(x.item()) for x in allc]
    cap = max(max(counts), 1)
    send = torch.zeros(cap, dtype=dtype, device=cdev)
    send[:count] = part[:count].to(cdev)
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    return torch.cat([recv[r][:counts[r]] for r in range(world)]).to(device).contiguous(), counts


def build_sharded(d, device=None, mode="scan"):
    """Run the sharded stage sequence on this rank's context `d` (text already loaded, same text on every
    rank).  Returns (row_base, rows, hash_rows) of this shard; results stay in HBM until fetch_shard()."""
    if mode == "exchange":
        return _build_exchange(d, device)
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
    mine = torch.empty(cap, dtype=torch.int64, device=device)     # empty, not zeros: a fill kernel on torch's
                                                                  # stream could land after the library's copy
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


def _build_exchange(d, device=None):
    L = _lib.lib()
    rank, world = dist.get_rank(), dist.get_world_size()
    device = device or torch.device("cuda", torch.cuda.current_device())
    cdev = _dev_for_comm(device)
    u64p, u8p, u32p = (ctypes.POINTER(t) for t in (ctypes.c_uint64, ctypes.c_uint8, ctypes.c_uint32))
    sync = lambda: torch.cuda.synchronize(device)            # torch's stream <-> the context's own stream

    # 1. census of the slice -> splitters
    _chk(d, L.debwt_shard_begin(d._h, rank, world))
    hist = np.zeros(SHARD_BINS, dtype=np.uint64)
    _chk(d, L.debwt_shard_histogram(d._h, hist.ctypes.data_as(u64p)))
    t = torch.from_numpy(hist.astype(np.int64)).to(cdev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    bins, cum = plan_splitters(t.cpu().numpy(), world)
    m_keys, m_base = int(cum[bins[rank + 1]] - cum[bins[rank]]), int(cum[bins[rank]])
    _chk(d, L.debwt_shard_set_range(d._h, bins[rank], bins[rank + 1], m_keys, m_base))
    shard_of_bin = np.zeros(SHARD_BINS, dtype=np.uint8)
    for r in range(world):
        shard_of_bin[bins[r]:bins[r + 1]] = r

    # 2. k-mer bucket exchange: keys of my text slice -> their owners
    cap = d.n // world + 64 + 32 * world
    part = torch.empty(cap, dtype=torch.int64, device=device)
    offs = np.zeros(world + 1, dtype=np.uint64)
    _chk(d, L.debwt_shard_partition_keys(d._h, shard_of_bin.ctypes.data_as(u8p), ctypes.c_void_p(part.data_ptr()),
                                         cap, offs.ctypes.data_as(u64p)))
    mine = _all_to_all_var(part, offs, cdev, device)
    sync()
    assert mine.numel() == m_keys, (mine.numel(), m_keys)
    _chk(d, L.debwt_shard_import_keys(d._h, ctypes.c_void_p(mine.data_ptr()), mine.numel()))
    del part, mine

    # 3. local sort + classification, red table from everybody's facts
    _chk(d, L.debwt_kmer_sort_rle(d._h))
    nf, nb, br = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    _chk(d, L.debwt_shard_classify_local(d._h, ctypes.byref(nf), ctypes.byref(nb), ctypes.byref(br)))
    counts = torch.tensor([nb.value, br.value], dtype=torch.int64, device=cdev)
    allc = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(allc, counts)
    allc = torch.stack(allc).cpu().numpy()
    first_block = np.concatenate([[0], np.cumsum(allc[:, 0])]).astype(np.uint32)
    qbase, blue_total = int(first_block[rank]), int(allc[:, 1].sum())
    myfacts = torch.empty(max(nf.value, 1), dtype=torch.int64, device=device)
    _chk(d, L.debwt_shard_facts_export(d._h, ctypes.c_void_p(myfacts.data_ptr()), max(nf.value, 1)))
    facts, _ = _all_gather_var(myfacts, nf.value, torch.int64, cdev, device)
    sync()
    _chk(d, L.debwt_shard_classify_global(d._h, ctypes.c_void_p(facts.data_ptr()), facts.numel(), qbase, blue_total))

    # 4. SP code: flags of my slice, offsets from the slice lengths (the reference's spSplit prefix,
    #    src/generateSP.c:152-157), symbols all-gathered
    s_loc, b_loc = ctypes.c_uint64(), ctypes.c_uint64()
    _chk(d, L.debwt_shard_sp_flags(d._h, ctypes.byref(s_loc), ctypes.byref(b_loc)))
    sl = torch.tensor([s_loc.value], dtype=torch.int64, device=cdev)
    alls = [torch.zeros_like(sl) for _ in range(world)]
    dist.all_gather(alls, sl)
    lens = [int(x.item()) for x in alls]
    sp_off, sp_total = sum(lens[:rank]), sum(lens)
    mysp = torch.empty(max(s_loc.value, 1), dtype=torch.uint8, device=device)
    _chk(d, L.debwt_shard_sp_emit(d._h, sp_off, ctypes.c_void_p(mysp.data_ptr()), max(s_loc.value, 1)))
    allsp, _ = _all_gather_var(mysp, s_loc.value, torch.uint8, cdev, device)
    sync()
    assert allsp.numel() == sp_total
    _chk(d, L.debwt_shard_sp_import(d._h, ctypes.c_void_p(allsp.data_ptr()), sp_total))

    # 5. blue entries of my slice -> the owners of their blocks
    routed = torch.empty(max(b_loc.value, 1), dtype=torch.int64, device=device)
    boffs = np.zeros(world + 1, dtype=np.uint64)
    _chk(d, L.debwt_shard_blue_route(d._h, first_block.ctypes.data_as(u32p), ctypes.c_void_p(routed.data_ptr()),
                                     max(b_loc.value, 1), boffs.ctypes.data_as(u64p)))
    got = _all_to_all_var(routed, boffs, cdev, device)
    sync()
    if os.environ.get("DEBWT_DEBUG"):
        print(f"[shard {rank}] keys {m_keys} facts {nf.value} blocks {nb.value} owned_rows {br.value} first_block "
              f"{first_block.tolist()} slice S {s_loc.value} mi {b_loc.value} sent {boffs.tolist()} got {got.numel()} "
              f"q range of got {int((got >> 36).min()) if got.numel() else -1}..{int((got >> 36).max()) if got.numel() else -1}",
              flush=True)
    _chk(d, L.debwt_shard_blue_place(d._h, ctypes.c_void_p(got.data_ptr()), got.numel()))

    # 6. owned blocks and rows
    _chk(d, L.debwt_blue_sort(d._h))
    _chk(d, L.debwt_bwt_assemble(d._h))
    rb, rows, nh = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    _chk(d, L.debwt_shard_info(d._h, ctypes.byref(rb), ctypes.byref(rows), ctypes.byref(nh)))
    return rb.value, rows.value, nh.value
