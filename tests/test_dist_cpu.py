"""The N>1 host path on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.  Compute is stood in by the
oracle on tiny collections (tests may use it); what is checked is the orchestration bench.py relies on:
collection assignment, per-rank seeds, the barrier-bracketed max-over-ranks timing and the sum reduction."""
import json
import os
import subprocess
import sys
import textwrap
import time

from conftest import ROOT

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import numpy as np
    from debwt_amd import dist as D, synth
    from oracle import oracle as O
    rank, local_rank, world = D.init(backend="gloo")
    seed = D.collection_seed(synth.SEED_P, rank)
    recs = synth.pan_genome(6000 + 0 * rank, 2, seed=seed)
    sym = O.sym_from_codes(recs)
    out = {}
    def step():
        out["bwt"] = O.build_bwt(sym, 32)
        time.sleep(0.05 * (rank + 1))                 # rank 1 is the slow one
    dt = D.timed_steps(step, steps=2, warmup=1)
    total = D.sum_over_ranks(len(sym))
    w, h, d, st = out["bwt"]
    rc, inv = O.inverse_bwt(w, len(sym), h, d)
    res = {"rank": rank, "world": world, "dt": dt, "total": total, "n": int(len(sym)), "seed": seed,
           "ok": bool(rc == 0 and np.array_equal(inv, sym)), "mine": D.assign_collections(5, rank, world),
           "sha": __import__("hashlib").sha256(w.tobytes()).hexdigest()}
    open(os.path.join(sys.argv[1], f"rank{rank}.json"), "w").write(json.dumps(res))
    D.finalize()
""") % ROOT


def test_two_rank_gloo_protocol(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", str(script), str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = [json.load(open(tmp_path / f"rank{i}.json")) for i in range(2)]
    assert [x["world"] for x in res] == [2, 2] and all(x["ok"] for x in res)
    assert res[0]["seed"] != res[1]["seed"] and res[0]["sha"] != res[1]["sha"]      # different collections
    assert res[0]["total"] == res[1]["total"] == res[0]["n"] + res[1]["n"]          # whole-job units
    assert abs(res[0]["dt"] - res[1]["dt"]) < 1e-9                                   # both hold the MAX
    assert res[0]["dt"] >= 2 * 0.10 - 0.01                                           # the slow rank's time
    assert res[0]["mine"] == [0, 2, 4] and res[1]["mine"] == [1, 3]


def test_single_process_defaults():
    from debwt_amd import dist as D
    assert D.env_world() == (0, 0, 1) or int(os.environ.get("WORLD_SIZE", "1")) > 1
    assert D.assign_collections(3, 0, 1) == [0, 1, 2]
    calls = []
    dt = D.timed_steps(lambda: calls.append(1), steps=3, warmup=2)
    assert len(calls) == 5 and dt >= 0
    assert D.sum_over_ranks(7) == 7.0


# ---- the host logic of the sharded build (debwt_amd/sharded.py) over gloo on CPU tensors: round planning from the
# slice censuses, the variable all_to_all of the key rounds and the variable all_gather of facts / SP symbols.  The
# "keys" are plain integers whose top 12 bits are their prefix bin; the device work is not part of this test.
SHARD_WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch
    import torch.distributed as dist
    from debwt_amd import dist as D, sharded as SH
    rank, local_rank, world = D.init(backend="gloo")
    rng = np.random.default_rng(100 + rank)
    keys = rng.integers(0, 1 << 20, size=5000 + 700 * rank, dtype=np.int64)        # my slice; bin = key >> 8
    keys[:300] = (77 << 8) | (keys[:300] & 255)                                    # a heavy bin
    hist = np.bincount(keys >> 8, minlength=4096).astype(np.int64)
    hists = SH._all_gather_small(hist)
    assert hists.shape == (world, 4096) and (hists[rank] == hist).all()
    total = hists.sum(axis=0)
    bins, cum = SH.plan_splitters(total, world)
    # every shard cuts its bins into key ranges of at most `cap` keys (the job of debwt_shard_plan)
    cap = int(total.sum()) // (world * 3) + 1
    def cut(lo, hi):
        b, acc, out = lo, 0, [lo]
        for x in range(lo, hi):
            if acc and acc + total[x] > cap:
                out.append(x); acc = 0
            acc += int(total[x])
        return out + [hi]
    cuts = [np.array(cut(bins[s], bins[s + 1])) for s in range(world)]
    rounds = max(len(c) - 1 for c in cuts)
    ws = SH.Workspace(type("Ctx", (), {"n": 0})(), torch.device("cpu"))
    got_all = []
    for t in range(rounds):
        tab, send, recv = SH.plan_round(hists, cuts, t, rank)
        dest = tab[keys >> 8]
        order = np.argsort(dest, kind="stable")
        part = keys[order][: int((dest != 0xFF).sum())]                             # grouped by owner, 0xFF last
        assert [int((dest == s).sum()) for s in range(world)] == send
        xa = torch.from_numpy(part.copy())
        xb = torch.empty(max(sum(recv), 1), dtype=torch.int64)
        n = SH._all_to_all(xb, xa, recv, send)
        got = xb[:n].numpy()
        if t + 1 < len(cuts[rank]):
            lo, hi = cuts[rank][t], cuts[rank][t + 1]
            assert n == int(total[lo:hi].sum()) and ((got >> 8) >= lo).all() and ((got >> 8) < hi).all()
        else:
            assert n == 0
        got_all.append(got.copy())
    mine = np.concatenate(got_all) if got_all else np.zeros(0, np.int64)
    assert len(mine) == int(total[bins[rank]:bins[rank + 1]].sum())
    # variable all_gather (facts, SP symbols)
    part = torch.arange(10 + 5 * rank, dtype=torch.int64) + 1000 * rank
    counts = [int(x) for x in SH._all_gather_small([part.numel()])[:, 0]]
    cat, tot = SH._all_gather_var(ws, "t", part, part.numel(), counts)
    want = np.concatenate([np.arange(10 + 5 * r) + 1000 * r for r in range(world)])
    assert tot == len(want) and (cat[:tot].numpy() == want).all()
    sp = torch.full((7 * (rank + 1),), rank, dtype=torch.uint8)
    counts = [int(x) for x in SH._all_gather_small([sp.numel()])[:, 0]]
    cat, tot = SH._all_gather_var(ws, "sp", sp, sp.numel(), counts)
    assert tot == sum(counts) and (cat[:tot].numpy() == np.concatenate([np.full(7 * (r + 1), r) for r in range(world)])).all()
    # the link probe of bench.py (N > 1): an all_to_all of 1 MiB per peer in calls of P2P_MAX bytes, content checked
    lp = SH.measure_link(torch.device("cpu"), mib_per_peer=1, reps=1)
    assert lp["content_ok"] and lp["gbytes_per_s_per_peer"] > 0 and lp["fed_to_cost_model"] is False, lp
    open(os.path.join(sys.argv[1], f"shard{rank}.json"), "w").write(json.dumps(
        {"rank": rank, "rounds": rounds, "keys": int(len(mine)), "sum": int(mine.sum()), "sent": int(len(keys)), "sent_sum": int(keys.sum())}))
    D.finalize()
""") % ROOT


def test_sharded_host_logic_over_gloo(tmp_path):
    script = tmp_path / "shard_worker.py"
    script.write_text(SHARD_WORKER)
    # (world 8 = the size the metric is quoted at: on the GPU box at most 6 processes may share the card, so the device-side
    # test of the sharded build stops at 4 ranks, tests/test_sharded.py, and the C host covers 8 shards in one process)
    for world, port, p2p_max in ((2, "29535", None), (3, "29536", "4096"), (8, "29537", "4096")):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        if p2p_max:
            env["DEBWT_P2P_MAX_BYTES"] = p2p_max           # every collective cut into many calls (RCCL's 1 GiB limit)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", port, str(script), str(tmp_path)]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
        res = [json.load(open(tmp_path / f"shard{i}.json")) for i in range(world)]
        assert res[0]["rounds"] >= 3                                               # several exchange rounds
        assert sum(x["keys"] for x in res) == sum(x["sent"] for x in res)           # every key reached exactly one owner
        assert sum(x["sum"] for x in res) == sum(x["sent_sum"] for x in res)


def test_bench_starts_its_own_ranks_when_run_directly():
    """`python bench.py --gpus N` with no launcher (the shape of the driver's N = 1 command) starts N child ranks,
    hands rank 0's one JSON line through on stdout and returns the launcher's exit code (--launch-probe: rendezvous
    only, no GPU)."""
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-probe"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                     # exactly one line on stdout; the ranks' chatter is on stderr
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_joined"] == 2 and "rank 1 joined" in r.stderr
    # a launcher that started the wrong number of ranks is an error message and a non-zero exit, not an assert
    env2 = dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-check"],
                       env=env2, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "one process per GPU" in r.stderr


def _probe(extra, env_extra=None, timeout=300):
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-probe"] + extra,
                          env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_extras_are_child_launches_that_cannot_own_the_exit_code():
    """The N > 1 extras of bench.py (the other key path, the C host) are fresh child launches after the measurement's ranks
    have exited (--launch-probe: rendezvous only).  Recorded when they come back; text in the line and exit code 0 when they
    fail or are not back in time; `--host=both` (one argument) is taken like `--host both` and does not recurse."""
    r = _probe(["--host=both"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["ranks_joined"] == 2 and set(j["key_modes_ms"]) >= {"exchange", "rescan"} and j["host_c"]["host"] == "c", j
    assert r.stderr.count("rank 1 joined") == 2          # the measurement's launch and the other key path's, nothing nested

    r = _probe([], env_extra={"DEBWT_BENCH_FAIL_EXTRA": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    assert "exit code" in j["key_modes_ms"]["exchange"] and "exit code" in j["host_c"]["error"], j
    assert "injected failure" in r.stderr

    t0 = time.perf_counter()
    r = _probe(["--extras-timeout", "2"], env_extra={"DEBWT_BENCH_EXTRA_SLEEP": "60"})
    assert r.returncode == 0 and time.perf_counter() - t0 < 50, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert "not back within" in j["key_modes_ms"]["exchange"] and "not back within" in j["host_c"]["error"], j

    # started by a launcher (WORLD_SIZE set): no extras at all; --host python / --no-other-mode: none either
    r = _probe(["--host", "python", "--no-other-mode"])
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    assert r.returncode == 0 and "key_modes_ms" not in j and "host_c" not in j, j


def test_bench_asked_to_leave_during_the_extras_prints_the_measured_line_first():
    """SIGTERM while an extra is running: the held line goes to stdout, the extra's processes are ended, exit code = the
    measurement's."""
    import signal
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["DEBWT_BENCH_EXTRA_SLEEP"] = "120"
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-probe"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    err = []
    for ln in p.stderr:                                   # wait for the measurement to be over and the extras to start
        err.append(ln)
        if "held back for the extras" in ln:
            break
    time.sleep(3.0)
    p.send_signal(signal.SIGTERM)
    out, _ = p.communicate(timeout=120)
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, out, "".join(err)[-1500:])
    j = json.loads(lines[0])
    assert j["ranks_joined"] == 2 and "key_modes_ms" not in j


def test_bench_promotes_the_c_host_line_only_when_it_came_back_whole():
    """N > 1 on real GPUs: the C host over RCCL (north_star's host) gives the line of the bench when its run is whole -- same
    steps, a positive value, the inverse BWT accepted -- and the python ranks' measurement moves under `host_python`; anything
    less leaves the python ranks' line as it is (bench.promote_c_host: pure dict logic, no GPU)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)
    py = {"metric": B.METRIC, "value": 40.0, "unit": "Gbp/s", "n_gpus": 8, "steps": 5, "warmup": 1, "ms_per_step": 750.0, "host": "python",
          "exchange": {"keys": "rescan"}, "check": {"inverse_bwt_ok": True}, "cpu_baseline": None, "vs_baseline": None}
    c = {"metric": B.METRIC, "value": 55.0, "unit": "Gbp/s", "n_gpus": 8, "steps": 5, "warmup": 1, "ms_per_step": 545.0, "host": "c",
         "exchange": {"backend": "RCCL"}, "check": {"inverse_bwt_ok": True}}
    out = B.promote_c_host(py, c)
    assert out is not py and out["value"] == 55.0 and out["ms_per_step"] == 545.0 and out["host"].startswith("c (")
    assert out["host_python"]["value"] == 40.0 and out["host_python"]["exchange"] == {"keys": "rescan"} and "cpu_baseline" in out
    for bad in (None, {}, dict(c, value=0.0), dict(c, steps=3), dict(c, host="python"), dict(c, check={"inverse_bwt_ok": False})):
        assert B.promote_c_host(py, bad) is py
