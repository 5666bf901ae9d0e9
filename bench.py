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
                  cores on a bounded prefix of the same workload (rank 0, N=1 only);
  cpu_reference -- the reference's own stage functions (oracle/_ref, 8 threads) on a shorter prefix, when built.
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


def cpu_reference(recs, k, budget_bases=10_000_000):
    """The reference's OWN stage functions (oracle/_ref/ref_driver: mySort ... insertCase3 compiled from the reference
    sources where they lie; only the Jellyfish dump in front of them is supplied by the oracle's counter) on a short
    prefix, with the reference's default of 8 threads.  None where the binary was not built."""
    import shutil
    import subprocess
    import tempfile
    from debwt_amd import fasta
    driver = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if not os.path.exists(driver):
        return None
    d = tempfile.mkdtemp(prefix="debwt_ref_")
    try:
        sample = np.ascontiguousarray(recs[0][:budget_bases])
        fa, out = os.path.join(d, "in.fa"), os.path.join(d, "OUT")
        fasta.write_fasta(fa, [sample])
        threads = min(8, os.cpu_count() or 1)
        t0 = time.perf_counter()
        p = subprocess.run([driver, d, fa, out, str(k), str(threads)], capture_output=True, text=True, timeout=600)
        dt = time.perf_counter() - t0
        if p.returncode:
            return None
        return {"value": round((len(sample) + 1) / dt / 1e9, 6), "unit": "Gbp/s", "cores": threads, "kind": "reference",
                "sample": f"first {len(sample)} bases of record 0 of the workload, k={k}, {dt:.1f} s (includes writing and "
                          "parsing the k-mer dump that stands in for Jellyfish)"}
    except Exception:
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="chr1_250M")
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--sort-algo", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tune", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)       # gloo: ranks may share one GPU (tests)
    ap.add_argument("--mode", choices=["replicas", "sharded", "sharded-scan"], default="replicas",
                    help="N>1: 'replicas' = one independent collection per GPU (default); 'sharded' = ONE "
                         "collection of N x the per-GPU size built by all GPUs as k-mer-prefix shards with the "
                         "key and blue-entry all_to_all exchanges; 'sharded-scan' = the same shards, every GPU "
                         "scanning the whole text instead of exchanging")
    args = ap.parse_args()

    import torch
    from debwt_amd import api, synth
    from debwt_amd import dist as D

    rank, local_rank, world = D.env_world()
    assert world == args.gpus or (world == 1 and args.gpus == 1), "launch one process per GPU"
    if args.backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    D.init(backend=args.backend, device_id=torch.device("cuda", local_rank))
    tdev = "cuda" if args.backend == "nccl" else "cpu"

    sharded_mode = args.mode.startswith("sharded") and world > 1
    shard_feed = "scan" if args.mode == "sharded-scan" else "exchange"
    if sharded_mode:
        # ONE collection, `world` records of the per-GPU size, the same text on every rank
        from debwt_amd import sharded as SH
        recs = []
        for r in range(world):
            recs += _workload_for_rank(synth, args.workload, r)
    else:
        # every rank builds its own collection of the same shape (independent objects)
        recs = _workload_for_rank(synth, args.workload, rank)
    n = sum(len(r) for r in recs) + len(recs)

    d = api.DeBWT(k=args.k, device=local_rank, sort_algo=args.sort_algo, tune=args.tune)
    d.load_records(recs)                      # text -> HBM before the timed region

    acc = {"pass_ms": 0.0, "pass_launches": 0, "stage": {}, "timed": False}

    def step():
        if sharded_mode:
            SH.build_sharded(d, torch.device("cuda", local_rank), mode=shard_feed)   # collectives inside
        else:
            d.build()                         # synchronous: returns after the context's stream drained
        if acc["timed"]:
            st_ = d.stats()
            acc["pass_ms"] += st_["radix_pass_ms"]
            acc["pass_launches"] += st_["radix_pass_launches"]
            for key in ("ms_extract", "ms_sort", "ms_classify", "ms_sp", "ms_blue", "ms_assemble", "ms_total"):
                acc["stage"][key] = acc["stage"].get(key, 0.0) + st_[key] / args.steps

    for _ in range(args.warmup):
        step()
    acc["timed"] = True
    dt = D.timed_steps(step, steps=args.steps, warmup=0, device_sync=torch.cuda.synchronize, tensor_device=tdev)
    total_bases = float(n) if sharded_mode else D.sum_over_ranks(n, tensor_device=tdev)
    pass_ms, pass_launches, stage = acc["pass_ms"], acc["pass_launches"], acc["stage"]
    st = d.stats()

    if rank == 0:
        ms_per_step = dt * 1e3 / args.steps
        value = total_bases / (dt / args.steps) / 1e9
        keys = st["radix_pass_keys"]
        mean_pass_ms = pass_ms / max(pass_launches, 1)
        achieved = 16.0 * keys / (mean_pass_ms * 1e-3) / 1e9 if mean_pass_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc) and args.workload == "chr1_250M" and args.k == 32:      # the PMC passes were run on this workload
            try:
                traffic = json.load(open(pmc)).get("rs_scatter_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": METRIC, "value": round(value, 4), "unit": "Gbp/s", "n_gpus": args.gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": (f"{args.workload} (BASELINE configs[1]: chr1-sized synthetic, repeat families)"
                                    if args.workload == "chr1_250M" else args.workload),
                       "k": args.k, "bases_per_gpu": n // world if sharded_mode else n,
                       "records_per_gpu": len(recs) // world if sharded_mode else len(recs),
                       "parallelism": (f"one collection of {len(recs)} records ({n} bases) built by {world} "
                                       f"k-mer-prefix shards ({shard_feed} mode)"
                                       if sharded_mode else f"{args.gpus} independent collections, one per GPU")},
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
            line["cpu_reference"] = cpu_reference(recs, args.k)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    d.close()
    D.finalize()


def _workload_for_rank(synth, name, rank):
    """Same shape as make_workload(name), different seed."""
    from debwt_amd import dist as D
    seed = D.collection_seed(synth.SEED_P, rank)
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
