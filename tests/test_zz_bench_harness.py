"""bench.py's own orchestration at N > 1 on the one-GPU box (ranks started by bench.py itself, the extras, the C host):
subprocess tests of the HARNESS, collected last (file name + conftest) so that no flake here can stand in front of a parity
test under `pytest -x`."""
import pytest

pytestmark = pytest.mark.gpu


def _run_bench_direct(extra, timeout=900, env_extra=None):
    """`python bench.py --gpus N ...` exactly as the driver types it for N = 1: no launcher in front."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_started_by_bench_itself_gloo():
    """bench.py --gpus 2 run DIRECTLY starts its two ranks as child processes (they share this box's one GPU, so the
    collectives go over gloo), builds ONE chr1-sized collection as two k-mer-prefix shards and prints one line whose
    result passed the census and the device inverse BWT."""
    j = _run_bench_direct(["--gpus", "2", "--backend", "gloo", "--workload", "chr1_250M", "--steps", "1", "--warmup", "1",
                           "--no-cpu-baseline"])
    assert j["n_gpus"] == 2 and j["steps"] == 1 and j["scaling"] == "strong" and j["value"] > 0
    assert j["check"]["census_equals_text"] and j["check"]["inverse_bwt_ok"], j["check"]
    assert "chr1_250M" in j["config"]["workload"] and j["config"]["bases_per_gpu"] * 2 <= j["config"]["bases"]
    # the link probe ran and both key paths were timed (--mode auto times the one the cost model did not choose as well)
    assert j["link_probe"]["content_ok"] and set(j["key_modes_ms"]) >= {"exchange", "rescan"}, (j["link_probe"], j["key_modes_ms"])
    assert all(isinstance(j["key_modes_ms"][m], float) for m in ("exchange", "rescan")), j["key_modes_ms"]


def test_bench_c_host_two_shards_on_one_gpu():
    """bench.py --gpus 2 --host c: ONE process, debwt_multi_build over two shards (both on this box's GPU with --backend
    gloo), the line says which host and which exchange its number belongs to; and the default N > 1 run (python ranks first,
    then the C host as a child process) carries the C host's line under `host_c`."""
    j = _run_bench_direct(["--gpus", "2", "--backend", "gloo", "--host", "c", "--workload", "chr1_250M", "--steps", "2", "--warmup", "1",
                           "--no-cpu-baseline"])
    assert j["host"] == "c" and j["n_gpus"] == 2 and j["steps"] == 2 and j["value"] > 0
    assert j["check"]["inverse_bwt_ok"] and j["check"]["census_equals_text"], j["check"]
    assert "peer copies" in j["exchange"]["backend"] and j["exchange"]["keys"] in ("exchange", "rescan")
    j = _run_bench_direct(["--gpus", "2", "--backend", "gloo", "--workload", "ecoli_4.6M", "--steps", "1", "--warmup", "1",
                           "--no-cpu-baseline", "--no-other-mode"])
    assert j["n_gpus"] == 2 and j["check"]["inverse_bwt_ok"] and "python" in j["host"]
    hc = j["host_c"]
    assert hc.get("host") == "c" and hc["value"] > 0 and hc["check"]["inverse_bwt_ok"], hc


def _run_bench_raw(extra, env_extra=None, timeout=900):
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_extras_cannot_cost_the_result_when_they_are_slow():
    """The second key path of --mode auto and the C host are extra information, each a fresh child launch after the ranks of
    the measurement have exited: when they are not back in time (here: 3 s for extras that sleep 30 s first) the run still
    ends with exit code 0 and exactly the measured line, the extras recorded as dropped."""
    import json
    r = _run_bench_raw(["--gpus", "2", "--backend", "gloo", "--workload", "ecoli_4.6M", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline", "--extras-timeout", "3"], env_extra={"DEBWT_BENCH_EXTRA_SLEEP": "30"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["check"]["inverse_bwt_ok"], j
    dropped = [v for v in j["key_modes_ms"].values() if isinstance(v, str) and v.startswith("dropped")]
    assert len(dropped) == 1 and "not back within" in dropped[0], j["key_modes_ms"]
    assert "not back within" in j["host_c"]["error"], j["host_c"]


def test_bench_extras_that_fail_are_recorded_and_cost_nothing():
    """An extra launch that RAISES (every rank of the other key path's launch, and the C host) is text in the line --
    `dropped: ... exit code` -- and the exit code of the run stays the measurement's: 0."""
    import json
    r = _run_bench_raw(["--gpus", "2", "--backend", "gloo", "--workload", "ecoli_4.6M", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline"], env_extra={"DEBWT_BENCH_FAIL_EXTRA": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["value"] > 0 and j["check"]["inverse_bwt_ok"]
    dropped = [v for v in j["key_modes_ms"].values() if isinstance(v, str) and v.startswith("dropped")]
    assert len(dropped) == 1 and "exit code" in dropped[0], j["key_modes_ms"]
    assert "exit code" in j["host_c"]["error"], j["host_c"]
    assert "injected failure" in r.stderr


def test_bench_two_ranks_started_by_bench_itself_rccl():
    """The same over RCCL when the box has two GPUs (the driver's 8-GPU node; skipped on a one-GPU box)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box: two RCCL ranks need two devices")
    for mode in ("exchange", "rescan"):
        j = _run_bench_direct(["--gpus", "2", "--workload", "chr1_250M", "--steps", "1", "--warmup", "1",
                               "--no-cpu-baseline", "--mode", mode], env_extra={"DEBWT_BIG_MESSAGE_PROBE": "1"})
        assert j["n_gpus"] == 2 and j["check"]["census_equals_text"] and j["check"]["inverse_bwt_ok"], j
        assert j["link_probe"]["content_ok"] and j["link_probe"]["gbytes_per_s_per_peer"] > 0, j["link_probe"]


def test_bench_under_a_launcher_hands_the_line_to_the_c_host_when_it_comes_back_whole():
    """The driver's N > 1 command (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) on real GPUs: after the
    ranks' measurement rank 0 runs the C host on the same collection as a child process -- the other ranks have exited and
    released their GPU -- and its line becomes the line of the bench (`host` starts with "c", the ranks' measurement under
    `host_python`).  Rehearsed here over gloo on the one GPU (DEBWT_BENCH_FORCE_C_AFTER: the C host then exchanges by peer
    copies); exactly one JSON line, exit code 0, both results verified by the device inverse BWT."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["DEBWT_BENCH_FORCE_C_AFTER"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload", "ecoli_4.6M",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["host"].startswith("c (") and j["n_gpus"] == 2 and j["steps"] == 2 and j["value"] > 0, j.get("host")
    assert j["check"]["inverse_bwt_ok"] and j["host_python"]["value"] > 0 and j["host_python"]["check"]["inverse_bwt_ok"], j
    # ... and when the C host fails, the ranks' line stays and says what happened
    env["DEBWT_BENCH_FAIL_EXTRA"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload", "ecoli_4.6M",
                        "--steps", "1", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert "host_python" not in j and j["check"]["inverse_bwt_ok"] and "exit code" in j["host_c"]["error"], j.get("host_c")
