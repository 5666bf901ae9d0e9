"""Host-side orchestration of the multi-GPU runs: one process per GPU, torch.distributed for rendezvous and
the two small reductions of the timing protocol (backend "nccl" = RCCL on the GPU box, "gloo" in CPU tests).

This module holds the launch and timing protocol only (rendezvous, barrier-bracketed max-over-ranks timing, the
whole-job sum); the data path of a multi-GPU build -- ONE collection as k-mer-prefix shards with the key and
blue-entry all_to_all exchanges -- is debwt_amd/sharded.py.  `collection_seed` / `assign_collections`
serve the alternative of independent collections per GPU (bench.py --mode replicas).  Nothing here touches device
memory.
"""
import os
import time


def env_world():
    """(rank, local_rank, world_size) as torch.distributed.run exports them; (0, 0, 1) when run directly."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None, device_id=None, force=False):
    """Join the process group when WORLD_SIZE > 1 (force: also a group of one).  Returns (rank, local_rank, world)."""
    rank, local_rank, world = env_world()
    if world > 1 or force:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            import torch
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device_id is not None:
            kw["device_id"] = device_id
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def finalize():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def collection_seed(base_seed, rank):
    """Seed of the collection rank `rank` builds: same shape on every rank, different content."""
    return (base_seed + 7919 * rank) & 0x7FFFFFFFFFFFFFFF


def assign_collections(n_collections, rank, world):
    """Independent collections -> ranks, round-robin (collection c goes to rank c % world)."""
    return [c for c in range(n_collections) if c % world == rank]


def _sync(device_sync):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if device_sync:
        device_sync()


def timed_steps(step_fn, steps, warmup, device_sync=None, tensor_device="cpu"):
    """The bench timing protocol: `warmup` untimed steps, then exactly `steps` steps bracketed by
    barrier + device synchronisation on both sides; returns the MAX over ranks of the elapsed seconds."""
    import torch
    import torch.distributed as dist
    for _ in range(warmup):
        step_fn()
    _sync(device_sync)
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    _sync(device_sync)
    dt = time.perf_counter() - t0
    if dist.is_available() and dist.is_initialized():
        t = torch.tensor([dt], dtype=torch.float64, device=tensor_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def sum_over_ranks(value, tensor_device="cpu"):
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        t = torch.tensor([float(value)], dtype=torch.float64, device=tensor_device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())
    return float(value)
