#!/bin/bash
# Kernel trace of one build: the all-HBM passes over the oversize stretches of the first key range, launch by launch
# usage: scripts/trace_aux_passes.sh WORKLOAD   (result: gpurun_out/aux_passes_WORKLOAD.txt)
set -e
W=${1:-real10x3G}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/ktr_$W -o t -- python3 $ROOT/bench.py --gpus 1 --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-check --h2h-reps 0 > $ROOT/gpurun_out/ktr_$W.json 2> $ROOT/gpurun_out/ktr_$W.err
cd $ROOT
F=$(find gpurun_out/ktr_$W -name "*kernel_trace.csv" | head -1)
python - "$F" > gpurun_out/aux_passes_$W.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = 0
for i, r in enumerate(rows):
    nm = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if "rs_over_move" in nm or "rs_hist_kernel<0, 1>" in nm or "rs_scatter_kernel<0, 1, 0>" in nm or "rs_local_count" in nm or "rs_unfit_rle" in nm or "rs_local_unfit" in nm:
        print("%8.3f ms  grid %-10s %s" % (d, r.get("Grid_Size_X", r.get("Grid_Size", "")), nm[:60]))
        n += 1
        if n > 140: break
PY
find gpurun_out/ktr_$W -name "*.csv" -delete
head -70 gpurun_out/aux_passes_$W.txt
