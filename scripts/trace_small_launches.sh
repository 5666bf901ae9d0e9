#!/bin/bash
# Long launches of small grids in the kernel trace of a run (three builds), with start time and neighbours
# usage: scripts/trace_small_launches.sh WORKLOAD
set -e
W=$1
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/ktr_$W -o t -- python3 $ROOT/bench.py --gpus 1 --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-check --h2h-reps 0 > $ROOT/gpurun_out/ktr_$W.json 2> $ROOT/gpurun_out/ktr_$W.err
cd $ROOT
F=$(find gpurun_out/ktr_$W -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"])//max(1,int(r["Workgroup_Size_X"]))) for r in rows)
t0 = ev[0][0]
for i,(s,e,n,wg) in enumerate(ev):
    d=(e-s)/1e6
    if wg < 1024 and d > 0.5 and "rocclr" not in n:
        print("%9.1f ms  %6.2f ms  wg %5d  %s   | before: %s | after: %s" % ((s-t0)/1e6, d, wg, n[:40], ev[i-1][2][:36], ev[i+1][2][:36] if i+1<len(ev) else ""))
PY
find gpurun_out/ktr_$W -name "*.csv" -delete
