#!/bin/bash
# round 6: the GPU suite, then the 30 Gbp bench with the first pass of a key range by waves (default) and in lockstep (A/B)
set -o pipefail
mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider > gpurun_out/r06/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06/pytest_gpu.txt
tail -n 5 gpurun_out/r06/pytest_gpu.txt
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --h2h-reps 0 > gpurun_out/r06/bench30_waves.json 2> gpurun_out/r06/bench30_waves.err || exit 1
DEBWT_SPARSE_LOCKSTEP=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --h2h-reps 0 --no-check > gpurun_out/r06/bench30_lockstep.json 2> gpurun_out/r06/bench30_lockstep.err || exit 1
python - <<'PY'
import json
for f in ("waves", "lockstep"):
    j = json.load(open(f"gpurun_out/r06/bench30_{f}.json"))
    print(f, j["ms_per_step"], j["stages_ms"], (j.get("check") or {}).get("inverse_bwt_ok"))
PY
