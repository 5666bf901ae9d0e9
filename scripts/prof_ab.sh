#!/bin/bash
# rocprofv3 kernel statistics of one library on a generated collection:
#   scripts/prof_ab.sh OUTNAME WL [LIB[@TUNE]]     -> gpurun_out/OUTNAME_kernel_stats.csv
set -e
cd "$(dirname "$0")/.."
out=$1; wl=${2:-300000000:10:24}; lib=${3:-default}
export TMPDIR=/tmp
python3 scripts/gpu_ab.py --wl "$wl" --keep
name=${lib%@*}; tune=0; [[ "$lib" == *@* ]] && tune=${lib#*@}
if [ "$name" != default ]; then export DEBWT_HIP_LIB=$PWD/build/variants/libdebwt_$name.so; fi
rm -rf /tmp/prof_$out
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$out -o p -- python3 scripts/gpu_ab.py --child "$lib" --tune "$tune" --reps 2 > gpurun_out/${out}_run.log 2>&1
cp $(find /tmp/prof_$out -name '*kernel_stats.csv' | head -1) gpurun_out/${out}_kernel_stats.csv
rm -f /dev/shm/debwt_ab_*
python3 - "$out" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(f"gpurun_out/{sys.argv[1]}_kernel_stats.csv")))
for r in rows[:28]:
    print("%-60s calls %5s avg %10.3f ms total %9.2f ms %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
