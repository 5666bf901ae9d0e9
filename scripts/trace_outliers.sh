#!/bin/bash
# Kernel trace of one build, read launch by launch: kernels whose slowest launch is far from their median (one key range or
# slice unlike the others) and long launches of small grids (tails).   usage: scripts/trace_outliers.sh WORKLOAD
set -e
W=${1:-real10x3G}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/ktr_$W -o t -- python3 $ROOT/bench.py --gpus 1 --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-check --h2h-reps 0 > $ROOT/gpurun_out/ktr_$W.json 2> $ROOT/gpurun_out/ktr_$W.err
cd $ROOT
F=$(find gpurun_out/ktr_$W -name "*kernel_trace.csv" | head -1)
python - "$F" > gpurun_out/outliers_$W.txt <<'PY'
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
by = {}
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    wg = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1) // max(1, int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1))
    by.setdefault(r["Kernel_Name"], []).append((d, wg))
print("kernels whose slowest launch is more than 1.5 x their median and 0.5 ms above it")
for k, v in sorted(by.items(), key=lambda kv: -sum(d for d, _ in kv[1])):
    ds = sorted(d for d, _ in v)
    med = st.median(ds)
    if len(ds) >= 3 and ds[-1] > 1.5 * med and ds[-1] - med > 0.5:
        print("  %-70s n %4d total %8.1f median %7.3f max %7.3f   launches: %s" % (k[:70], len(ds), sum(ds), med, ds[-1], " ".join("%.2f" % d for d, _ in v[:16])))
print("launches of fewer than 1024 workgroups that take more than 0.5 ms")
for k, v in sorted(by.items(), key=lambda kv: -sum(d for d, _ in kv[1])):
    sm = [(d, wg) for d, wg in v if wg < 1024 and d > 0.5]
    if sm: print("  %-70s n %4d total %8.1f  e.g. %.2f ms with %d workgroups" % (k[:70], len(sm), sum(d for d, _ in sm), sm[0][0], sm[0][1]))
PY
find gpurun_out/ktr_$W -name "*.csv" -delete
cat gpurun_out/outliers_$W.txt
