#!/bin/bash
# rocprofv3 kernel statistics of the default bench (30 Gbp on one GPU); summaries land in gpurun_out/ -> copy to profiles/
# usage: scripts/prof_30g.sh TAG [extra bench args]
set -e
TAG=${1:-r02}; shift || true
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$TAG -o $TAG -- python3 $ROOT/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-check --h2h-reps 0 "$@" > $ROOT/gpurun_out/${TAG}_bench_under_rocprof.json 2> $ROOT/gpurun_out/${TAG}_under.err
cd $ROOT
F=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
cp $F gpurun_out/${TAG}_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${TAG}_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms over the run: %.1f"%(tot/1e6))
for r in rows[:32]:
    print("%-90s calls %5s total %9.1f ms avg %9.3f ms %5.1f%%"%(r["Name"][:90], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6, float(r["Percentage"])))
PY
