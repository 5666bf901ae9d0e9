#!/usr/bin/env python3
"""bench.py -- Gbp/s of BWT construction on MI355X (BASELINE.json metric), one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--k 32]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the whole hot path (key extraction, radix sort, classification, SP code, blue-block
sort, assembly) over one synthetic collection whose packed text is already resident in HBM.  At N=1 the
workload is BASELINE.json configs[1] (chr1-sized, 250 Mbp, k=32).  At N>1 every rank builds the BWT of its
own collection of that size (independent objects: weak scaling, no data-path collective); the timed region
is bracketed by a barrier + device synchronisation on both sides and the slowest rank's time is used.

The one JSON line also carries
  roofline     -- the dominant kernel (one radix scatter pass): algorithmic bytes (16 B per key moved:
                  8 read + 8 written) / mean launch time from hipEvents recorded inside the timed region on
                  the stream the kernel runs on, against the 8 TB/s HBM peak;
  cpu_baseline -- the CPU oracle (a single-threaded port of the reference path) timed on this box's host
                  cores on a bounded prefix of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
METRIC = "Gbp/sec BWT build (30 Gbp synthetic DNA); bit-exact vs CPU ref; 1/2/4/8-GPU"


def cpu_baseline(recs, k, budget_bases=150_000_000):
    """Oracle on a prefix of the first record (same generator, same repeat structure)."""
    from oracle import oracle as O
    sample = [np.ascontiguousarray(recs[0][:budget_bases])]
    sym = O.sym_from_codes(sample)
    t0 = time.perf_counter()
    O.build_bwt(sym, k, threads=1)
    dt = time.perf_counter() - t0
    return {"value": round(len(sym) / dt / 1e9, 6), "unit": "Gbp/s", "cores": 1, "kind": "port",
            "sample": f"first {len(sample[0])} bases of record 0 of the workload, k={k}, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="chr1_250M")
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--sort-algo", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from debwt_amd import api, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert world == args.gpus or (world == 1 and args.gpus == 1), "launch one process per GPU"
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # every rank gets its own collection of the same shape (seed differs by rank)
    if rank == 0:
        recs = synth.make_workload(args.workload)
    else:
        recs = _workload_for_rank(synth, args.workload, rank)
    n = sum(len(r) for r in recs) + len(recs)

    d = api.DeBWT(k=args.k, device=local_rank, sort_algo=args.sort_algo)
    d.load_records(recs)                      # text -> HBM before the timed region

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        d.build()
    sync()
    pass_ms, pass_launches, stage = 0.0, 0, {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        d.build()                             # synchronous: returns after the stream drained
        st = d.stats()
        pass_ms += st["radix_pass_ms"]
        pass_launches += st["radix_pass_launches"]
        for key in ("ms_extract", "ms_sort", "ms_classify", "ms_sp", "ms_blue", "ms_assemble", "ms_total"):
            stage[key] = stage.get(key, 0.0) + st[key] / args.steps
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([float(n)], dtype=torch.float64, device="cuda")
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_bases = float(tot.item())
    else:
        total_bases = float(n)
    st = d.stats()

    if rank == 0:
        ms_per_step = dt * 1e3 / args.steps
        value = total_bases / (dt / args.steps) / 1e9
        keys = st["radix_pass_keys"]
        mean_pass_ms = pass_ms / max(pass_launches, 1)
        achieved = 16.0 * keys / (mean_pass_ms * 1e-3) / 1e9 if mean_pass_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("rs_scatter_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": METRIC, "value": round(value, 4), "unit": "Gbp/s", "n_gpus": args.gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"{args.workload} (BASELINE configs[1]: chr1-sized synthetic, repeat families)",
                       "k": args.k, "bases_per_gpu": n, "records_per_gpu": len(recs),
                       "parallelism": f"{args.gpus} independent collections, one per GPU"},
            "roofline": {"bound": "hbm", "kernel": "rs_scatter_kernel (one 8-bit radix pass over the keys)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "bytes_per_launch": 16 * keys, "mean_launch_ms": round(mean_pass_ms, 4),
                         "launches_timed": pass_launches},
            "stages_ms": {k_: round(v, 3) for k_, v in stage.items()},
            "counters": {k_: st[k_] for k_ in ("n", "nrec", "n_main", "distinct_keys", "red_capacity",
                                              "blue_capacity", "blue_bound_num", "sp_len")},
        }
        if args.gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(recs, args.k)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    d.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _workload_for_rank(synth, name, rank):
    """Same shape as make_workload(name), different seed."""
    seed = synth.SEED_P + 7919 * rank
    if name == "chr1_250M":
        return synth.pan_genome(250_000_000, 1, seed=seed)
    if name == "ecoli_4.6M":
        return synth.pan_genome(4_600_000, 1, seed=seed)
    if name == "pan_100M_4":
        return synth.chromosomes(100_000_000, 4, seed=seed)
    if name == "pan_16M_4":
        return synth.pan_genome(4_000_000, 4, seed=seed)
    return synth.make_workload(name)


if __name__ == "__main__":
    main()
