"""The N>1 host path on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.  Compute is stood in by the
oracle on tiny collections (tests may use it); what is checked is the orchestration bench.py relies on:
collection assignment, per-rank seeds, the barrier-bracketed max-over-ranks timing and the sum reduction."""
import json
import os
import subprocess
import sys
import textwrap

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
