#!/usr/bin/env python3
"""bench.py -- Gbp/s of BWT construction on MI355X (BASELINE.json metric), one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--k 32]      (N > 1: starts its own N ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (default, every N): BASELINE.json configs[4], the configuration the metric is quoted on -- 10 genomes x
3.0 Gbp = 30 Gbp of synthetic DNA in 240 chromosome-like records (distribution P of SURVEY 8d: repeat families,
SNPs at 1e-3 between the genomes; /root/reference/README.md:19 "10 human genomes").  The text is produced by the
native generator (include/debwt_synth.h) from the formula of debwt_amd/synth.py, straight into page-locked host
memory in the reference's 2-bit layout.

A step = one pass of the whole hot path (key extraction, radix sort, classification, SP code, blue-block sort,
assembly) over that collection.
  N = 1   the one GPU builds the whole BWT: the key space is cut into k-mer-prefix ranges that are sorted one after
          the other over the resident text.
  N > 1   ONE collection, built by N k-mer-prefix shards (strong scaling: the same 30 Gbp at every N): census
          all-gather; the keys of a shard's ranges either arrive by all_to_all of the 8-byte k-mers (RCCL over xGMI,
          --mode exchange) or are read from the shard's own copy of the text (--mode rescan) -- --mode auto, the
          default, takes the cheaper one by the library's cost model, which on one node is the second; local sort and
          classification, all-gather of the branching-node facts and of the SP code, all_to_all of the blue entries,
          local blue sort and assembly, final concatenation of the row ranges on rank 0.
`value` = bases / wall time of a step with the packed text resident in HBM (on every GPU) and the result left in
HBM (rank 0), barrier + device synchronisation on both sides, slowest rank.  Extra keys:
  host_to_host  -- N = 1: the same build from the page-locked host text to the BWT and its '#'/'$' rows back in
                   page-locked host memory (SURVEY 8d's region: load + build + fetch), MEAN over as many consecutive
                   steps as `steps`;
  link_probe    -- N > 1: all_to_all rate per peer pair measured on this node before the first build, fed to the
                   key-path cost model (debwt_shard_key_mode) in place of its assumed link rate;
  key_modes_ms  -- N > 1, --mode auto: ms per build of the chosen key path (the timed steps) AND of the other one;
  first_build_s -- the cold first build (allocates the workspace);
  check         -- outside the timed region: symbol census of the result against the text's (device kernel), '#' rows
                   ascending, and the inverse BWT on the device (one LF walk per text segment, debwt_verify_device);
  roofline      -- the dominant kernel (one 8-bit radix scatter pass over the keys of a range): algorithmic bytes
                   (16 B per key: 8 read + 8 written) / mean launch time from hipEvents recorded around every such
                   launch of the first key range on the stream it runs on, against the 8 TB/s HBM peak;
  cpu_baseline  -- the reference's own stage functions (oracle/_ref, compiled from /root/reference/src in the build
                   container) on this box's host cores on a bounded prefix of record 0 (rank 0, N = 1 only);
  cpu_port      -- the single-threaded CPU oracle on a prefix of the same record;
  cpu_baseline_at_configs -- the reference on the whole ecoli_4.6M and chr1_250M collections (BASELINE configs[0] and [1];
                   --no-cpu-configs leaves out the second, a minute of CPU).
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
WORKLOAD_NOTE = {
    "pan10x3G": "BASELINE configs[4]: 10 genomes x 3.0 Gbp in 240 chromosome-like records, repeat families, SNP 1e-3",
    "pan4x3.1G": "BASELINE configs[3]: 4 x GRCh38-sized genomes in 96 records",
    "grch38_3.1G": "BASELINE configs[2]: one GRCh38-sized genome in 24 records",
    "chr1_250M": "BASELINE configs[1]: chr1-sized single record, repeat families",
    "ecoli_4.6M": "BASELINE configs[0]: E. coli-sized single record",
    "uniform_3.1G": "distribution U: uniform 3.1 Gbp in 24 records",
    "real_3.1G": "distribution R: 3.1 Gbp in 24 records with an Alu-like family, satellite arrays, homopolymer tracts",
    "pan10x600M": "distribution P, 10 genomes x 600 Mbp in 240 records (up to 4 shards fit next to each other on one GPU; 8 do at 10 x 400 Mbp)",
    "real10x600M": "distribution R, 10 genomes x 600 Mbp in 240 records (Alu-like family, satellites, homopolymer tracts at the 3 Gbp densities)",
    "real10x3G": "distribution R at the headline size: 10 genomes x 3.0 Gbp in 240 records, each with an Alu-like family "
                 "(10^6 copies), satellite arrays and homopolymer tracts, SNP 1e-3 between the genomes",
}


def cpu_port(codes, k):
    """The CPU oracle (single-threaded restatement of the reference path) on a sample."""
    from oracle import oracle as O
    sym = O.sym_from_codes([codes])
    t0 = time.perf_counter()
    O.build_bwt(sym, k, threads=1)
    dt = time.perf_counter() - t0
    return {"value": round(len(sym) / dt / 1e9, 6), "unit": "Gbp/s", "cores": 1, "kind": "port",
            "sample": f"first {len(codes)} bases of record 0 of the workload, k={k}, {dt:.1f} s"}


def cpu_reference(codes, k, threads):
    """The reference's OWN stage functions (oracle/_ref/ref_driver: mySort ... insertCase3 compiled from the reference
    sources where they lie) with -t `threads`.  The k-mer dump in front of them stands in for Jellyfish (absent third
    party tool) and is not timed; mySort's parse of that dump is reference code and is.  None where the binary was
    not built."""
    import shutil
    import subprocess
    import tempfile
    from debwt_amd import fasta
    driver = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if not os.path.exists(driver):
        return None
    d = tempfile.mkdtemp(prefix="debwt_ref_")
    try:
        fa, out = os.path.join(d, "in.fa"), os.path.join(d, "OUT")
        fasta.write_fasta(fa, [codes])
        t0 = time.perf_counter()
        p = subprocess.run([driver, d, fa, out, str(k), str(threads)], capture_output=True, text=True, timeout=900)
        dt = time.perf_counter() - t0
        if p.returncode:
            return None
        stages = {}
        if os.path.exists(out + ".timing"):
            for ln in open(out + ".timing"):
                a, b = ln.split()
                stages[a] = float(b)
        ref_s = sum(v for s_, v in stages.items() if s_ != "kmer_dump_standin") if stages else dt
        return {"value": round((len(codes) + 1) / ref_s / 1e9, 6), "unit": "Gbp/s", "cores": threads, "kind": "reference",
                "sample": f"first {len(codes)} bases of record 0 of the workload, k={k}, -t {threads}: {ref_s:.1f} s in the "
                          f"reference's stage functions (mySort incl. its ~3 s of lock/histogram set-up ... insertCase3); "
                          f"the {stages.get('kmer_dump_standin', 0):.1f} s k-mer dump that stands in for Jellyfish is not counted",
                "stages_s": {s_: round(v, 2) for s_, v in stages.items()}}
    except Exception:
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def emit_line(line):
    """The one JSON line, object and newline in a single write (print() issues two: another process on the same pipe can
    get in between)."""
    sys.stdout.flush()
    os.write(1, (json.dumps(line) + "\n").encode())


def promote_c_host(py_line, c_line):
    """N > 1 on real GPUs: north_star's host is C -- one process, one host thread per GPU, exchanges as grouped ncclSend / ncclRecv
    over RCCL (debwt_multi_build, what `cli/deBWT --gpus N --exchange rccl` runs).  When that run came back whole, ITS line is the
    line of the bench and the python ranks' measurement (one process per GPU, torch.distributed) goes under `host_python`;
    otherwise the python ranks' line stays, with what became of the C host under `host_c`."""
    ok = (isinstance(c_line, dict) and c_line.get("host") == "c" and isinstance(c_line.get("value"), (int, float)) and c_line["value"] > 0
          and c_line.get("steps") == py_line.get("steps") and (c_line.get("check") is None or c_line["check"].get("inverse_bwt_ok", True)))
    if not ok:
        return py_line
    out = dict(c_line)
    out["host"] = "c (one process, one host thread per GPU, debwt_multi_build); host_python: the same collection by one process per GPU"
    out["host_python"] = {k_: py_line[k_] for k_ in ("value", "unit", "ms_per_step", "steps", "warmup", "stages_ms", "exchange", "link_probe",
                                                    "key_modes_ms", "first_build_s", "check", "config") if k_ in py_line}
    for k_ in ("cpu_baseline", "vs_baseline"):
        out.setdefault(k_, py_line.get(k_))
    return out


def run_c_host(args):
    """--host c: ONE process, one host thread per GPU inside the library (debwt_multi_build -- what `cli/deBWT --gpus N`
    runs), the same collection, the same checks, the same line with "host": "c".  With --backend gloo (the one-GPU test
    box) all shards sit on GPU 0."""
    import ctypes
    import numpy as np
    from debwt_amd import api, _lib
    from debwt_amd import synth_native as SN
    L = _lib.lib()
    ndev = ctypes.c_int(0)
    ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(ndev))
    devices = list(range(args.gpus)) if (ndev.value >= args.gpus and args.backend == "nccl") else [0] * args.gpus
    t0 = time.perf_counter()
    syn = SN.Synth.named(args.workload)
    n, nrec = syn.n, syn.nrec
    sep = syn.sep()
    text = SN.PinnedArray(syn.nwords)
    census = syn.words_into(text.ptr)
    t_gen = time.perf_counter() - t0
    m = api.MultiDeBWT(devices, k=args.k, tune=args.tune)
    if args.exchange == "rccl":
        m.set_exchange("rccl")
    m.set_key_mode({"auto": "auto", "exchange": "exchange", "rescan": "rescan"}.get(args.mode, "auto"))
    if args.serial_shards:
        m.set_serial(True)                                 # the shards take turns on the GPU: per-shard times that are each shard's own
    t0 = time.perf_counter()
    m.load_packed(text.a, n, sep)
    t_load = time.perf_counter() - t0
    t0 = time.perf_counter()
    m.build()
    first_build_s = time.perf_counter() - t0
    for _ in range(max(args.warmup - 1, 0)):
        m.build()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m.build()                                          # synchronous: returns when every shard's stream has drained
    dt = time.perf_counter() - t0
    ms, s0 = m.stats()
    check = None
    if not args.no_check:
        rep = m.verify_device()                            # inverse BWT of the concatenated result on the first GPU
        check = {"inverse_bwt_ok": bool(rep["ok"]),
                 "inverse_bwt": {k_: (round(v, 2) if isinstance(v, float) else v) for k_, v in rep.items() if k_ != "ok"}}
        if n < 4_000_000_000:                              # the rows on the host as well: census and row lists
            w, h, dr = m.fetch()
            cnt = np.zeros(4, dtype=np.int64)
            for a in range(0, len(w), 1 << 22):
                x = w[a:a + (1 << 22)]
                for sh_ in range(0, 64, 2):
                    cnt += np.bincount(((x >> np.uint64(sh_)) & np.uint64(3)).astype(np.uint8), minlength=4)[:4]
            cnt[0] -= (-n) % 32                            # unused tail bits of the last word are zero
            want = census.astype(np.int64).copy(); want[3] += nrec
            check.update({"census_equals_text": bool((cnt == want).all()),
                          "hash_rows_ascending": bool((np.diff(h.astype(np.int64)) > 0).all()) if nrec > 2 else True,
                          "hash_rows": int(len(h)), "dollar_row": int(dr)})
    keys = s0["radix_pass_keys"]
    mean_pass_ms = s0["radix_pass_ms"] / max(s0["radix_pass_launches"], 1)
    achieved = 16.0 * keys / (mean_pass_ms * 1e-3) / 1e9 if mean_pass_ms > 0 else 0.0
    line = {
        "metric": METRIC, "value": round(n / (dt / args.steps) / 1e9, 4), "unit": "Gbp/s", "n_gpus": args.gpus,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3 / args.steps, 3), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic", "host": "c",
        "config": {"workload": f"{args.workload} ({WORKLOAD_NOTE.get(args.workload, 'custom')})", "k": args.k, "bases": n,
                   "records": nrec, "bases_per_gpu": n // args.gpus,
                   "parallelism": f"one process, one host thread per GPU (debwt_multi_build): {args.gpus} k-mer-prefix shards on devices "
                                  f"{devices}, text replicated in the HBM of every GPU, result concatenated on the first"},
        "exchange": {"backend": "RCCL: grouped ncclSend/ncclRecv per exchange (ncclCommInitAll in the one process)"
                     if ms["exchange_backend"] == 1 else "peer copies: every shard pulls its segments with hipMemcpyAsync device-to-device",
                     "keys": "exchange" if ms["key_mode"] == 0 else "rescan", "key_rounds": ms["rounds"],
                     "key_bytes_into_shard0": ms["key_bytes_in"], "blue_bytes_into_shard0": ms["blue_bytes_in"]},
        "roofline": {"bound": "hbm", "kernel": "rs_scatter_kernel<0,0,1> of shard 0 (one 8-bit radix pass over the keys of a key range)",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": None, "bytes_per_launch": 16 * keys, "mean_launch_ms": round(mean_pass_ms, 4),
                     "launches_timed": s0["radix_pass_launches"]},
        "first_build_s": round(first_build_s, 3), "setup_s": {"generate_text": round(t_gen, 2), "load_to_hbm": round(t_load, 3)},
        "check": check, "cpu_baseline": None,
    }
    if args.serial_shards:
        line["shards"] = [m.shard_report(r) for r in range(args.gpus)]
        line["serial_shards"] = ("the shards took turns between the barriers, one on the GPU at a time (debwt_multi_set_serial): "
                                 "`value` is then the SUM of the shards' times, not a parallel build")
    emit_line(line)
    m.close()
    text.free()
    return 0


def _strip_flags(argv, flags):
    """argv without the given value-carrying flags, in both forms (`--flag value`, `--flag=value`)."""
    out, skip = [], False
    for a_ in argv:
        if skip:
            skip = False
            continue
        if a_ in flags:
            skip = True
            continue
        if any(a_.startswith(f + "=") for f in flags):
            continue
        out.append(a_)
    return out


try:                                 # (optional: ends the whole process tree of an extra; without it the launcher alone is ended)
    import psutil
except ImportError:
    psutil = None

_CHILD = {"p": None}                 # the child launch that is running now (launch_ranks' signal handler ends exactly it)


def _end_tree(p, grace=15.0):
    """End the child `p` and every process it started (exact PIDs: the children stay in this process group, so whoever
    ends bench.py by group ends them too): SIGTERM to the launcher -- torch.distributed.run hands it on to its workers --
    then SIGKILL to whatever is left after `grace` seconds."""
    if psutil is None:                                        # no psutil: the launcher alone (it hands SIGTERM on to its workers)
        try:
            p.terminate()
            p.wait(grace)
        except Exception:                                     # noqa: BLE001 -- still there after `grace`, or already gone
            try:
                p.kill()
            except OSError:
                pass
        return
    try:
        tree = [psutil.Process(p.pid)] + psutil.Process(p.pid).children(recursive=True)
    except psutil.Error:
        tree = []
    try:
        p.terminate()
    except OSError:
        pass
    _, alive = psutil.wait_procs(tree, timeout=grace)
    for q in alive:
        try:
            q.kill()
        except psutil.Error:
            pass


def _run_child(cmd, env, timeout=None, on_object=None):
    """Run one child command, relaying its stdout: JSON objects that start with {"metric" are parsed (the last one is
    returned), everything else goes to stderr.  Returns (exit code or None on timeout, object or None)."""
    import subprocess
    import threading
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    _CHILD["p"] = p
    timed_out = {"v": False}

    def _expire():
        timed_out["v"] = True
        _end_tree(p)

    timer = threading.Timer(timeout, _expire) if timeout else None
    if timer:
        timer.daemon = True
        timer.start()
    got = None
    for ln in p.stdout:
        # the ranks share one pipe: another rank's output (gloo prints its connection messages to stdout in pieces) may
        # stand in front of rank 0's object or between it and its newline: take the object from where it starts to where it ends
        at = ln.find('{"metric"')
        obj, end = None, 0
        if at >= 0:
            try:
                obj, end = json.JSONDecoder().raw_decode(ln[at:])
            except ValueError:
                obj = None
        if obj is not None:
            rest = ln[:at] + ln[at + end:]
            got = obj
            if on_object:
                on_object(obj)
            if rest.strip():
                sys.stderr.write(rest if rest.endswith("\n") else rest + "\n")
        else:
            sys.stderr.write(ln)
        sys.stderr.flush()
    rc = p.wait()
    if timer:
        timer.cancel()
    _CHILD["p"] = None
    return (None if timed_out["v"] else rc), got


def _start_ranks(n_ranks, argv, env, timeout=None):
    """One `python -m torch.distributed.run --nproc-per-node N bench.py <argv>` child: (exit code or None on timeout, rank
    0's object or None)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return _run_child(cmd, env, timeout=timeout)


def launch_ranks(n_ranks, args):
    """`python bench.py --gpus N` run directly (no launcher, WORLD_SIZE unset): start the N ranks as CHILD processes
    of this one -- `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` -- before anything here
    has touched the GPU (no torch import yet, and never an exec of a process that has), hand rank 0's one JSON line
    through to stdout (everything else the ranks print goes to stderr) and return THAT launch's exit code.

    Extras (extra information, each a FRESH child launch after the ranks of the measurement have exited and released their
    GPUs; their failures and timeouts are recorded as text in the line and never touch the exit code):
      key_modes_ms -- --mode auto: the key path the cost model did NOT choose, `bench.py --gpus N --mode <other>` over <= 3 steps;
      host_c       -- the same collection through the C host (one process, one thread per GPU: what cli/deBWT --gpus runs).
    The measured line is held back while the extras run (at most --extras-timeout seconds in total); it is written to stderr
    at once, and a SIGTERM/SIGINT to this process prints it to stdout before leaving."""
    import signal
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    nested = os.environ.get("DEBWT_BENCH_NESTED") == "1"      # this IS an extra of another bench.py: no extras of its own
    own = _strip_flags(sys.argv[1:], ("--host",))
    rc, held = _start_ranks(n_ranks, own + ["--host", "python"], env)
    if held is None:
        return rc if rc else 1
    want_other = (not nested and rc == 0 and args.mode == "auto" and not args.no_other_mode and held.get("exchange") is not None)
    want_c = not nested and rc == 0 and args.host in (None, "both")
    if not (want_other or want_c):
        emit_line(held)
        return rc

    sys.stderr.write("bench.py: measured line (held back for the extras): " + json.dumps(held) + "\n")
    sys.stderr.flush()
    state = {"out": False}

    def flush_held(*_sig):
        if not state["out"]:
            state["out"] = True
            emit_line(held)
        if _sig:                                              # asked to leave: the line is out; end the extra that is running
            if _CHILD["p"] is not None:
                _end_tree(_CHILD["p"], grace=5.0)
            os._exit(rc)

    old = {sg: signal.signal(sg, flush_held) for sg in (signal.SIGTERM, signal.SIGINT)}
    deadline = time.perf_counter() + args.extras_timeout
    env2 = dict(env, DEBWT_BENCH_NESTED="1")
    base = _strip_flags(own, ("--steps", "--warmup", "--mode", "--extras-timeout", "--cpu-sample", "--port-sample"))
    no_check = ["--no-check"] if "--no-check" in base else []
    base = [a_ for a_ in base if a_ not in ("--no-check", "--no-cpu-baseline", "--no-other-mode")]
    steps = str(max(1, min(3, held.get("steps", 1))))
    c_full = None
    try:
        if want_other:
            chosen = held["exchange"].get("keys", "rescan")
            other = "exchange" if chosen == "rescan" else "rescan"
            key_modes = {chosen: held["ms_per_step"]}
            left = deadline - time.perf_counter()
            try:
                if left <= 0:
                    raise TimeoutError
                rc2, j = _start_ranks(n_ranks, base + ["--host", "python", "--mode", other, "--steps", steps, "--warmup", "1",
                                                      "--no-check", "--no-cpu-baseline"], env2, timeout=left)
                if rc2 is None:
                    raise TimeoutError
                if rc2 == 0 and j is not None:
                    key_modes[other] = j["ms_per_step"]
                    key_modes["note"] = (f"ms per build; '{chosen}' is the cost model's choice and the timed steps of `value`, '{other}' "
                                         f"was timed over {steps} steps after one warm-up by a fresh launch of the ranks afterwards")
                else:
                    key_modes[other] = f"dropped: the extra launch ended with exit code {rc2}"
            except TimeoutError:
                key_modes[other] = f"dropped: not back within the {args.extras_timeout} s the extras may take"
            except Exception as e:                            # noqa: BLE001
                key_modes[other] = f"dropped: {type(e).__name__}: {e}"
            held["key_modes_ms"] = key_modes
        if want_c:
            left = deadline - time.perf_counter()
            # on real GPUs (RCCL between the ranks) the C host runs the SAME steps over RCCL -- its line becomes the line of the
            # bench when it comes back whole (promote_c_host); on the one-GPU test box (gloo) it stays a short extra over peer copies
            real = args.backend == "nccl" and not args.launch_probe
            c_steps = [str(held.get("steps", 1)), str(held.get("warmup", 1))] if real else [steps, "1"]
            child = ([sys.executable, os.path.abspath(__file__)] + base + ["--host", "c", "--steps", c_steps[0], "--warmup", c_steps[1], "--no-cpu-baseline"]
                     + (["--exchange", "rccl"] if real and "--exchange" not in base else []) + no_check)
            try:
                if left <= 0:
                    raise TimeoutError
                rc3, j = _run_child(child, env2, timeout=left)
                if rc3 is None:
                    raise TimeoutError
                if rc3 == 0 and j is not None:
                    c_full = j if real else None
                    held["host_c"] = {k_: j[k_] for k_ in ("value", "unit", "ms_per_step", "steps", "warmup", "host", "exchange",
                                                          "first_build_s", "check", "config") if k_ in j}
                else:
                    held["host_c"] = {"error": f"exit code {rc3}"}
            except TimeoutError:
                held["host_c"] = {"error": f"not back within the {args.extras_timeout} s the extras may take"}
            except Exception as e:                            # noqa: BLE001
                held["host_c"] = {"error": f"{type(e).__name__}: {e}"}
            held["host"] = "python (one process per GPU, torch.distributed); host_c: the C host on the same collection"
            if c_full is not None:
                promoted = promote_c_host(held, c_full)
                if promoted is not held:
                    held.clear(); held.update(promoted)
    finally:
        for sg, h in old.items():
            signal.signal(sg, h)
        flush_held()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="pan10x3G")
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--sort-algo", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--h2h-reps", type=int, default=-1,
                    help="host-to-host steps (N = 1; load + build + fetch, mean over them); default = --steps, 0 = skip")
    ap.add_argument("--host", choices=["python", "c", "both"], default=None,
                    help="which host drives the GPUs.  python: one process per GPU, torch.distributed (RCCL all_to_all); c: ONE "
                         "process, one host thread per GPU inside the library (debwt_multi_build, what cli/deBWT --gpus runs), "
                         "exchanges by --exchange; both (default for N > 1 when bench.py starts its own ranks): the python ranks "
                         "give `value`, then the C host runs as a child process and its line becomes the key `host_c`")
    ap.add_argument("--serial-shards", action="store_true",
                    help="--host c: the shards take turns, one on the GPU at a time, and the line carries every shard's step times "
                         "(`shards`): load balance of N = 2, 4, 8 measured on a box with one GPU")
    ap.add_argument("--exchange", choices=["peer", "rccl"], default="peer",
                    help="--host c: device-to-device copies (default) or grouped ncclSend/ncclRecv (needs one GPU per shard)")
    ap.add_argument("--extras-timeout", type=float, default=420.0,
                    help="N > 1 started by bench.py itself: seconds the extras (other key path, C host) may take in total")
    ap.add_argument("--no-reserve", action="store_true", help="no debwt_reserve beside the text generation (A/B of the cold path)")
    ap.add_argument("--h2h-plain", action="store_true", help="host-to-host steps with build + fetch one after the other (A/B)")
    ap.add_argument("--cpu-configs", action="store_true", help=argparse.SUPPRESS)          # (the default now)
    ap.add_argument("--no-cpu-configs", action="store_true",
                    help="do not time the reference on the whole chr1_250M collection (BASELINE configs[1]; a minute of CPU)")
    ap.add_argument("--no-other-mode", action="store_true",
                    help="N>1 with --mode auto: do not also time the key path the cost model did not choose")
    ap.add_argument("--cpu-sample", type=int, default=100_000_000, help="bases of record 0 given to the reference")
    ap.add_argument("--port-sample", type=int, default=100_000_000, help="bases of record 0 given to the oracle port")
    ap.add_argument("--tune", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)       # gloo: ranks may share one GPU (tests)
    ap.add_argument("--force-sharded", action="store_true", help=argparse.SUPPRESS)   # the N>1 code path at N = 1 (tests)
    ap.add_argument("--launch-probe", action="store_true", help=argparse.SUPPRESS)    # rendezvous only, no GPU (CPU tests)
    ap.add_argument("--mode", choices=["auto", "exchange", "rescan", "scan", "replicas"], default="auto",
                    help="N>1: ONE collection built by N k-mer-prefix shards; SP symbols, facts, blue entries and the "
                         "final rows always travel over RCCL.  'exchange' = the 8-byte keys travel too (all_to_all per "
                         "key range); 'rescan' = every GPU reads its own copy of the text once per key range instead; "
                         "'auto' (default) = the cheaper of the two by the library's cost model (debwt_shard_key_mode); "
                         "'scan' = no bulk exchange at all; 'replicas' = N independent collections (no collective)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if os.environ.get("DEBWT_BENCH_NESTED") == "1" and ("WORLD_SIZE" in os.environ or args.host == "c"):
        # test hooks, seen by the extras only (an extra = a child launch that carries DEBWT_BENCH_NESTED): fail, or be slow
        if os.environ.get("DEBWT_BENCH_FAIL_EXTRA"):
            raise RuntimeError("injected failure of an extra launch (tests)")
        time.sleep(float(os.environ.get("DEBWT_BENCH_EXTRA_SLEEP", "0")))
    if args.host == "c" and args.launch_probe:              # the CPU suite's stand-in for the C host's line (no GPU touched)
        emit_line({"metric": METRIC, "probe": True, "host": "c", "n_gpus": args.gpus, "value": 0.0, "ms_per_step": 0.0, "steps": args.steps})
        return
    if args.host == "c":
        sys.exit(run_c_host(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, args))

    from debwt_amd import dist as D
    rank, local_rank, world = D.env_world()
    if args.launch_probe:                      # what the CPU suite can check of the N > 1 start: ranks up, one line out
        D.init(backend="gloo")
        total = D.sum_over_ranks(1)
        if rank == 0:
            emit_line({"metric": METRIC, "probe": True, "n_gpus": args.gpus, "ranks_joined": int(total), "steps": args.steps,
                       "ms_per_step": 0.0, "mode": args.mode, "exchange": {"keys": "rescan"}})
        else:
            print(f"rank {rank} joined", flush=True)
        D.finalize()
        return

    import torch
    from debwt_amd import api
    from debwt_amd import synth_native as SN

    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks (one process per GPU)")
    if args.backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    # RCCL prints a version banner to stdout when its first communicator comes up: send it to stderr, stdout carries
    # the one JSON line
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        D.init(backend=args.backend, device_id=torch.device("cuda", local_rank), force=args.force_sharded)
        if world > 1 or args.force_sharded:
            D.sum_over_ranks(1, tensor_device="cuda" if args.backend == "nccl" else "cpu")
            torch.cuda.synchronize()
    finally:
        sys.stdout.flush()
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
    tdev = "cuda" if args.backend == "nccl" else "cpu"
    device = torch.device("cuda", local_rank)
    sharded = (world > 1 and args.mode != "replicas") or args.force_sharded

    # ---- the collection: formula-defined, generated natively into page-locked host memory --------------------------
    t0 = time.perf_counter()
    seed = None if (world == 1 or sharded) else D.collection_seed(0x5EEDBA5E, rank)
    syn = SN.Synth.named(args.workload, seed=seed)
    n, nrec, nwords = syn.n, syn.nrec, syn.nwords
    sep = syn.sep()
    d = api.DeBWT(k=args.k, device=local_rank, sort_algo=args.sort_algo, tune=args.tune)
    # one GPU: the workspace is allocated on a thread of its own while the text is still being produced (debwt_reserve) --
    # what a one-shot host does while it reads its input.  The driver clears device memory another process released at
    # ~33 GiB/s (profiles/r04_alloc_probe.txt); a 30 Gbp build holds ~250 GB.
    one_shot_mode = world == 1 and not sharded and not args.no_reserve
    reserve = {"s": 0.0, "error": None}

    def _reserve():
        t_r = time.perf_counter()
        try:
            d.reserve(n, nrec)                             # (a context that is reused: the default key ranges)
        except Exception as e:                                # noqa: BLE001 -- the build allocates what is missing
            reserve["error"] = str(e)
        reserve["s"] = time.perf_counter() - t_r

    import threading
    rthread = threading.Thread(target=_reserve) if one_shot_mode else None
    if rthread:
        rthread.start()
    text = SN.PinnedArray(nwords)
    census = np.zeros(4, dtype=np.uint64)
    if sharded:
        from debwt_amd import sharded as SH
        census = SH.generate_text_all_gather(syn, text, device)       # every rank packs 1/world of the words
    else:
        census = syn.words_into(text.ptr)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    if rthread:
        rthread.join()
    t_reserve_exposed = time.perf_counter() - t0

    t0 = time.perf_counter()
    d.load_packed(text.a, n, sep)                 # text -> HBM before the timed region
    t_load = time.perf_counter() - t0

    acc = {"pass_ms": 0.0, "pass_launches": 0, "stage": {}, "timed": False, "xfer": {}, "info": {}}
    shard_ws = SH.Workspace(d, device, mode=args.mode) if sharded else None
    # N > 1: what the links of THIS node sustain, measured before the first build and fed to the key-path cost model
    link_probe = None
    if sharded and world > 1:
        try:
            link_probe = SH.measure_link(device, mib_per_peer=1024 if args.backend == "nccl" else 16)
        except Exception as e:                                # noqa: BLE001 -- the probe informs the cost model, nothing depends on it
            link_probe = {"failed": str(e)}

    def step():
        if sharded:
            info = SH.build_sharded(d, shard_ws)          # collectives and the final concat inside
            if acc["timed"]:
                for key, v in info.items():
                    if isinstance(v, (int, float)):
                        acc["xfer"][key] = acc["xfer"].get(key, 0.0) + v / args.steps
                    else:
                        acc["info"][key] = v
        else:
            d.build()                                     # synchronous: returns after the context's stream drained
        if acc["timed"]:
            st_ = d.stats()
            if os.environ.get("DEBWT_BENCH_TRACE"):
                print("step: " + " ".join("%s=%.1f" % (k_[3:], st_[k_]) for k_ in ("ms_sort", "ms_classify", "ms_sp", "ms_blue", "ms_total")),
                      file=sys.stderr, flush=True)
            acc["pass_ms"] += st_["radix_pass_ms"]
            acc["pass_launches"] += st_["radix_pass_launches"]
            for key in ("ms_extract", "ms_sort", "ms_classify", "ms_sp", "ms_blue", "ms_assemble", "ms_total"):
                acc["stage"][key] = acc["stage"].get(key, 0.0) + st_[key] / args.steps

    one_shot = None
    t0 = time.perf_counter()
    if one_shot_mode:
        # the cold first build as a one-shot host runs it: host text (loaded above: t_load) -> BWT and row lists in page-locked
        # host memory through debwt_build_to_host, nothing warm
        os_words = SN.PinnedArray((n + 31) // 32); os_hash = SN.PinnedArray(max(nrec - 1, 1)); os_dollar = SN.PinnedArray(1)
        t0 = time.perf_counter()
        d.build_into(os_words.a, os_hash.a, os_dollar.a)
        torch.cuda.synchronize()
        first_build_s = time.perf_counter() - t0
        one_shot = {"seconds": round(t_load + first_build_s, 3), "value": round(n / (t_load + first_build_s) / 1e9, 4), "unit": "Gbp/s",
                    "load_s": round(t_load, 3), "build_to_host_s": round(first_build_s, 3),
                    "reserve_s": round(reserve["s"], 3), "reserve_exposed_s": round(t_reserve_exposed, 3),
                    "reserve_error": reserve["error"],
                    "note": "the FIRST build of this process, nothing warm: page-locked host text -> debwt_load_text -> "
                            "debwt_build_to_host -> BWT + row lists in page-locked host memory; the workspace was allocated by "
                            "debwt_reserve on a helper thread while the text was generated (reserve_s; reserve_exposed_s = what "
                            "of it outlasted the generation and is NOT in `seconds`)"}
        os_words.free(); os_hash.free(); os_dollar.free()
    else:
        step()                                            # cold build: allocates the workspace (reported, not timed)
        torch.cuda.synchronize()
        first_build_s = time.perf_counter() - t0
    # (after a one-shot first build one plain build more stays untimed: debwt_build sizes a few buffers for the whole text that
    # debwt_build_to_host sized per key range -- the batch buffer of the large-block split above all, 7.5 GB on distribution R --
    # and the timed steps are to hold no first-use allocation)
    for _ in range(max(args.warmup - 1, 0) + (1 if one_shot_mode else 0)):
        step()
    acc["timed"] = True
    dt = D.timed_steps(step, steps=args.steps, warmup=0, device_sync=torch.cuda.synchronize, tensor_device=tdev)
    total_bases = float(n) if (sharded or world == 1) else D.sum_over_ranks(n, tensor_device=tdev)
    pass_ms, pass_launches, stage = acc["pass_ms"], acc["pass_launches"], acc["stage"]
    st = d.stats()

    # ---- outside the timed region: checks, host-to-host figure, CPU baselines ----------------------------------------
    check = None
    if not args.no_check:
        if sharded:
            check = SH.check_result(d, shard_ws, census, n, nrec)
        elif world == 1:
            got = d.bwt_census().astype(np.int64)
            want = census.astype(np.int64).copy()
            want[3] += nrec                                # '#' and '$' rows are stored as 3
            _, hrows, drow = d.fetch_small()
            check = {"census_equals_text": bool((got == want).all()),
                     "hash_rows_ascending": bool((np.diff(hrows.astype(np.int64)) > 0).all()) if nrec > 2 else True,
                     "hash_rows": int(len(hrows)), "dollar_row": int(drow)}
            if hasattr(d, "verify_device"):
                check.update(d.verify_device())
    h2h = None
    h2h_reps = args.steps if args.h2h_reps < 0 else args.h2h_reps
    if world == 1 and not sharded and h2h_reps > 0:
        # SURVEY 8d's region: page-locked host text -> debwt_load_text -> debwt_build -> debwt_fetch_bwt into page-locked
        # host memory; MEAN over h2h_reps consecutive steps bracketed by device synchronisation, like the timed steps above
        out_words = SN.PinnedArray((n + 31) // 32)
        out_hash = SN.PinnedArray(max(nrec - 1, 1))
        out_dollar = SN.PinnedArray(1)
        parts = np.zeros(3)
        torch.cuda.synchronize()
        t_begin = time.perf_counter()
        for _ in range(h2h_reps):
            t0 = time.perf_counter()
            d.load_packed(text.a, n, sep)
            t1 = time.perf_counter()
            if args.h2h_plain:
                d.build()
                t2 = time.perf_counter()
                d.fetch_into(out_words.a, out_hash.a, out_dollar.a)
            else:
                d.build_into(out_words.a, out_hash.a, out_dollar.a)      # debwt_build_to_host: rows leave range by range
                t2 = time.perf_counter()
            t3 = time.perf_counter()
            parts += (t1 - t0, t2 - t1, t3 - t2)
        torch.cuda.synchronize()
        mean_s = (time.perf_counter() - t_begin) / h2h_reps
        parts /= h2h_reps
        h2h = {"value": round(n / mean_s / 1e9, 4), "unit": "Gbp/s", "seconds": round(mean_s, 4), "steps": h2h_reps,
               "load_s": round(parts[0], 4), "build_s": round(parts[1], 4), "fetch_s": round(parts[2], 4),
               "call": "debwt_load_text + debwt_build + debwt_fetch_bwt" if args.h2h_plain else
                       "debwt_load_text + debwt_build_to_host (build_s holds the copy of the rows: finished key ranges leave "
                       "for the host while the blocks of the next range are sorted)",
               "note": "MEAN over `steps` consecutive steps of: page-locked host text -> HBM, build, BWT + '#'/'$' rows -> "
                       "page-locked host memory (SURVEY 8d's region; PCIe both ways inside).  A fresh load also re-plans the "
                       "key ranges (prefix census of the text).  `value` above is the same build with the text resident in HBM"}
        if not (out_words.a[:4] != 0).any() and n > 4096:
            h2h["error"] = "the host buffer holds no rows"
        out_words.free(); out_hash.free(); out_dollar.free()

    line = None
    if rank == 0:
        ms_per_step = dt * 1e3 / args.steps
        value = total_bases / (dt / args.steps) / 1e9
        keys = st["radix_pass_keys"]
        mean_pass_ms = pass_ms / max(pass_launches, 1)
        achieved = 16.0 * keys / (mean_pass_ms * 1e-3) / 1e9 if mean_pass_ms > 0 else 0.0
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc) and args.k == 32 and world == 1:
            try:
                import hashlib
                j = json.load(open(pmc))
                src = os.path.join(ROOT, "debwt_amd", "csrc", "radix_sort.hip")
                now = hashlib.sha256(open(src, "rb").read()).hexdigest() if os.path.exists(src) else None
                then = (j.get("kernel_source") or {}).get("sha256")
                if j.get("workload") != args.workload:      # the PMC passes were run on another workload
                    pass
                elif then is None or then != now:           # ... or on another version of the kernel: a stale counter is no counter
                    traffic_source = ("dropped: profiles/pmc_latest.json was measured on radix_sort.hip "
                                      f"{(then or 'unknown')[:12]}, this tree holds {(now or 'none')[:12]} -- rerun scripts/pmc_30g.sh")
                else:
                    traffic = j.get("rs_scatter_bytes_per_launch")
                    traffic_source = ("profiles/pmc_latest.json: the builder's rocprofv3 --pmc passes on this workload "
                                      f"({j.get('source', 'scripts/pmc_30g.sh')}) with radix_sort.hip {now[:12]} = the kernel of "
                                      "this tree; NOT a counter of this run")
            except Exception:
                traffic = None
        if sharded:
            par = (f"one collection of {nrec} records ({n} bases) built by {world} k-mer-prefix shards "
                   f"(keys: {acc['info'].get('keys', args.mode)}), text replicated in the HBM of every GPU, result concatenated on rank 0")
        elif world > 1:
            par = f"{world} independent collections, one per GPU"
        else:
            par = "one GPU, k-mer-prefix key ranges sorted one after the other over the resident text"
        line = {
            "metric": METRIC, "value": round(value, 4), "unit": "Gbp/s", "n_gpus": args.gpus,
            "timed_region": "packed text and result resident in HBM (debwt_build); SURVEY 8d's host DRAM -> host DRAM region, "
                            "PCIe both ways inside, is the key `host_to_host`",
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak" if (world > 1 and not sharded) else "strong",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{args.workload} ({WORKLOAD_NOTE.get(args.workload, 'custom')})",
                       "k": args.k, "bases": n, "records": nrec, "bases_per_gpu": n // world if sharded else n,
                       "parallelism": par},
            "roofline": {"bound": "hbm", "kernel": "rs_scatter_kernel<0,0,1> (one 8-bit radix pass over the keys of a key range)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "bytes_per_launch": 16 * keys, "mean_launch_ms": round(mean_pass_ms, 4),
                         "launches_timed": pass_launches},
            "stages_ms": {k_: round(v, 3) for k_, v in stage.items()},
            "counters": {k_: st[k_] for k_ in ("n", "nrec", "n_main", "distinct_keys", "red_capacity",
                                              "blue_capacity", "blue_bound_num", "sp_len", "blue_large_blocks")},
            "first_build_s": round(first_build_s, 3),
            "one_shot": one_shot,
            "setup_s": {"generate_text": round(t_gen, 2), "load_to_hbm": round(t_load, 3)},
            "check": check,
            "host_to_host": h2h,
        }
        if sharded:
            line["exchange"] = {k_: round(v, 3) for k_, v in acc["xfer"].items()}
            line["exchange"].update(acc["info"])
            line["link_probe"] = link_probe
            line["key_modes_ms"] = None
        if args.gpus == 1 and not args.no_cpu_baseline:
            threads = SN.default_threads()
            ref = cpu_reference(syn.codes(0, 0, min(args.cpu_sample, int(syn._lens[0]))), args.k, threads)
            port = cpu_port(syn.codes(0, 0, min(args.port_sample, int(syn._lens[0]))), args.k)
            line["cpu_baseline"] = ref if ref else port
            line["cpu_port"] = port
            # the reference on WHOLE BASELINE configurations (configs[0] always; configs[1] = minutes of CPU,
            # on request -- profiles/ holds the run): same stage functions, same thread count
            at = {}
            for wl in ["ecoli_4.6M"] + ([] if args.no_cpu_configs else ["chr1_250M"]):
                s2 = SN.Synth.named(wl)
                at[wl] = cpu_reference(s2.codes(0, 0, int(s2._lens[0])), args.k, threads)
                s2.close()
            line["cpu_baseline_at_configs"] = at
        else:
            line["cpu_baseline"] = None

    # N > 1 under an external launcher (the driver's command) on real GPUs: after the ranks' measurement, rank 0 runs the C host
    # over RCCL on the same collection as a CHILD process -- once every rank has released its GPU (the others simply exit: the
    # launcher waits for rank 0) -- and its line becomes the line of the bench when it comes back whole (promote_c_host).  The
    # measured line is held back meanwhile; SIGTERM / SIGINT print it first.  --host python: no such child.
    # (DEBWT_BENCH_FORCE_C_AFTER=1: the same choreography over gloo on the one-GPU test box, the C host then over peer copies)
    forced = os.environ.get("DEBWT_BENCH_FORCE_C_AFTER") == "1"
    c_after = (rank == 0 and world > 1 and sharded and (args.backend == "nccl" or forced) and args.host in (None, "both")
               and os.environ.get("DEBWT_BENCH_NESTED") != "1" and "TORCHELASTIC_RUN_ID" in os.environ)
    if rank == 0 and not c_after:
        emit_line(line)
    d.close()
    text.free()
    shard_ws = None
    D.finalize()
    if c_after:
        import signal
        torch.cuda.empty_cache()
        state = {"out": False}

        def flush_line(*_sig):
            if not state["out"]:
                state["out"] = True
                emit_line(line)
            if _sig:
                if _CHILD["p"] is not None:
                    _end_tree(_CHILD["p"], grace=5.0)
                os._exit(0)

        old = {sg: signal.signal(sg, flush_line) for sg in (signal.SIGTERM, signal.SIGINT)}
        try:
            env2 = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                                       "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
            env2["DEBWT_BENCH_NESTED"] = "1"
            child = [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--host", "c", "--backend", args.backend,
                     "--exchange", "rccl" if args.backend == "nccl" else "peer",
                     "--steps", str(args.steps), "--warmup", str(args.warmup), "--workload", args.workload, "--k", str(args.k),
                     "--mode", args.mode if args.mode in ("auto", "exchange", "rescan") else "auto", "--no-cpu-baseline"] + (["--no-check"] if args.no_check else [])
            sys.stderr.write("bench.py: measured line of the ranks (held back for the C host): " + json.dumps(line) + "\n")
            sys.stderr.flush()
            rc3, j = _run_child(child, env2, timeout=args.extras_timeout)
            if rc3 == 0 and j is not None:
                promoted = promote_c_host(line, j)
                if promoted is line:
                    line["host_c"] = {"error": "came back without a usable line", "line": {k_: j.get(k_) for k_ in ("value", "steps", "check")}}
                line = promoted
            else:
                line["host_c"] = {"error": "not back within %.0f s" % args.extras_timeout if rc3 is None else "exit code %s" % rc3}
        except Exception as e:                                # noqa: BLE001 -- the C host is extra information here
            line["host_c"] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            for sg, h in old.items():
                signal.signal(sg, h)
            flush_line()


if __name__ == "__main__":
    main()
