"""One BWT over several ranks (k-mer-prefix shards).  CPU: the host-side planning and concatenation logic.
GPU: 2 and 3 ranks sharing the box's single GPU, collectives over gloo -- the same orchestration the 8-GPU node
runs over RCCL -- checked bit for bit against the oracle."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT


def test_plan_splitters_balances_and_covers():
    from debwt_amd.sharded import plan_splitters
    rng = np.random.default_rng(1)
    hist = rng.integers(0, 1000, size=4096).astype(np.uint64)
    hist[100:130] = 50000                                  # skewed prefixes
    for world in (1, 2, 3, 8):
        bins, cum = plan_splitters(hist, world)
        assert bins[0] == 0 and bins[-1] == 4096 and all(a <= b for a, b in zip(bins, bins[1:]))
        sizes = [int(cum[bins[r + 1]] - cum[bins[r]]) for r in range(world)]
        assert sum(sizes) == int(hist.sum())
        assert max(sizes) <= int(hist.sum()) / world + int(hist.max())


def test_concat_rows_bit_exact(oracle):
    from debwt_amd import synth
    from debwt_amd.sharded import concat_rows
    sym = oracle.sym_from_codes(synth.pan_genome(5000, 3))
    w, h, d, _ = oracle.build_bwt(sym, 32)
    n = len(sym)
    rows = (w[np.arange(n) >> 5] >> ((31 - (np.arange(n) & 31)).astype(np.uint64) * np.uint64(2))) & np.uint64(3)
    cuts = [0, 37, 4100, 4101, 9000, n]
    parts = []
    for a, b in zip(cuts, cuts[1:]):
        m = b - a
        loc = np.zeros((m + 31) // 32 + 1, dtype=np.uint64)
        j = np.arange(m)
        np.bitwise_or.at(loc, j >> 5, rows[a:b] << ((31 - (j & 31)).astype(np.uint64) * np.uint64(2)))
        parts.append((a, m, loc))
    assert np.array_equal(concat_rows(parts, n), w)


WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch
    import torch.distributed as dist
    from debwt_amd import api, synth, sharded
    from debwt_amd import dist as D
    rank, local_rank, world = D.init(backend="gloo")
    torch.cuda.set_device(0)                              # every rank on the one GPU of the test box
    case, k, out, mode, cap = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5])
    if case == "pan":
        recs = synth.pan_genome(300_000, 3)
    elif case == "many":
        recs = synth.pan_genome(20_000, 9, seed=5)
    elif case == "reads":                                 # 3000 records: the special-region module runs on the device
        recs = synth.read_set(3000, 60, 300, 200_000, seed=11)
    else:
        recs = synth.chromosomes(2_000_000, 4)
    n = sum(len(r) for r in recs) + len(recs)
    d = api.DeBWT(k=k, device=0)
    d.load_records(recs)
    if cap:
        d.set_range_cap(cap)                              # several key ranges (exchange rounds) per shard
    ws = sharded.Workspace(d, torch.device("cuda", 0), mode=mode)
    for it in range(2):                                   # a context is reusable in sharded mode too
        info = sharded.build_sharded(d, ws)
    res = sharded.gather_bwt(d, n, ws)
    if rank == 0:
        w, h, dr = res
        np.savez(out, w=w, h=h, d=np.array([dr], dtype=np.uint64))
    D.finalize()
    d.close()
""") % ROOT


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["scan", "exchange", "rescan", "auto"])
@pytest.mark.parametrize("world,case,k,cap", [(2, "pan", 32, 0), (3, "pan", 20, 0), (2, "many", 32, 0), (4, "chrom", 32, 0),
                                              (2, "pan", 32, 100_000), (3, "chrom", 24, 150_000), (4, "many", 32, 8192),
                                              (3, "reads", 32, 0), (2, "reads", 20, 50_000)])
# (world <= 4: the GPU box allows 6 processes on its card at once, and the runner, the launcher's agent and the ranks all
#  count -- five ranks were killed by its process guard.  World 8, the size the metric is quoted at: the host logic over gloo
#  on CPU in tests/test_dist_cpu.py, and 8 shards in ONE process through debwt_multi_build in tests/test_gpu_verify.py)
def test_sharded_build_equals_oracle(tmp_path, oracle, world, case, k, mode, cap):
    from debwt_amd import synth
    if case == "reads" and mode in ("scan", "auto"):
        pytest.skip("the many-record collection runs in the two key modes only (suite time)")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = str(tmp_path / "res.npz")
    port = str(29540 + world + 10 * ["scan", "exchange", "rescan", "auto"].index(mode) + (40 if cap else 0))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    if cap:
        env["DEBWT_P2P_MAX_BYTES"] = "65536"               # collectives in many calls, as above RCCL's 1 GiB message limit
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", port, str(script), case, str(k), out, mode, str(cap)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    recs = {"pan": lambda: synth.pan_genome(300_000, 3), "many": lambda: synth.pan_genome(20_000, 9, seed=5),
            "reads": lambda: synth.read_set(3000, 60, 300, 200_000, seed=11),
            "chrom": lambda: synth.chromosomes(2_000_000, 4)}[case]()
    ow, oh, od, _ = oracle.build_bwt(oracle.sym_from_codes(recs), k)
    z = np.load(out)
    assert np.array_equal(z["w"], ow) and np.array_equal(z["h"], oh) and int(z["d"][0]) == od
