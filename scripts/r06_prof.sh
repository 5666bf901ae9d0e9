#!/bin/bash
# round 6: GPU suite, kernel statistics of the 30 Gbp build (first pass by waves / in lockstep), SQ counters, gather calibration,
# the sharded code path at N = 1
set -o pipefail
mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider > gpurun_out/r06/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06/pytest_gpu.txt
tail -n 4 gpurun_out/r06/pytest_gpu.txt
bash scripts/prof_30g.sh r06a > gpurun_out/r06/prof_waves.txt 2>&1 || exit 1
DEBWT_SPARSE_LOCKSTEP=1 bash scripts/prof_30g.sh r06b > gpurun_out/r06/prof_lockstep.txt 2>&1 || exit 1
grep -h "sparse" gpurun_out/r06/prof_waves.txt gpurun_out/r06/prof_lockstep.txt
bash scripts/pmc_sq.sh r06a > gpurun_out/r06/sq_waves.txt 2>&1 || exit 1
grep -h "sparse\|rs_scatter_kernel<0, 0, 1>" gpurun_out/r06/sq_waves.txt
build/gather16 > gpurun_out/r06/gather16.txt 2>&1 || exit 1
(cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OLDPWD/gpurun_out/r06/pmc_gather16 -o c -- $OLDPWD/build/gather16 > /dev/null 2>&1) || exit 1
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r06/pmc_gather16/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "FETCH_SIZE": acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
with open("gpurun_out/r06/gather16.txt", "a") as o:
    for k, v in acc.items():
        line = "FETCH_SIZE %-28s launches %d mean %.1f KiB" % (k, len(v), sum(v) / len(v))
        print(line); o.write(line + "\n")
PY
python bench.py --force-sharded --mode rescan --steps 3 --warmup 1 --no-cpu-baseline --h2h-reps 0 > gpurun_out/r06/bench30_sharded_rescan.json 2> gpurun_out/r06/bench30_sharded_rescan.err || exit 1
python -c "
import json; j=json.load(open('gpurun_out/r06/bench30_sharded_rescan.json')); print('sharded N=1', j['ms_per_step'], j['stages_ms'], j['exchange'], (j.get('check') or {}).get('inverse_bwt_ok'))"
bash scripts/prof_30g.sh r06c --force-sharded --mode rescan > gpurun_out/r06/prof_sharded.txt 2>&1 || exit 1
head -n 30 gpurun_out/r06/prof_sharded.txt
