#!/bin/bash
# HBM traffic of the default bench (30 Gbp) from two separate counter passes (FETCH_SIZE, WRITE_SIZE), summarised by
# scripts/pmc_summarize.py into gpurun_out/pmc_$TAG.json (copy to profiles/ and profiles/pmc_latest.json)
set -e
TAG=${1:-r02}; shift || true
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}_$C -o c -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-check --h2h-reps 0 "$@" > $ROOT/gpurun_out/pmc_${TAG}_$C.json 2> $ROOT/gpurun_out/pmc_${TAG}_$C.err
done
cd $ROOT
KEYS=$(python -c "import json; print(json.load(open('gpurun_out/pmc_${TAG}_FETCH_SIZE.json'))['roofline']['bytes_per_launch']//16)")
WL=$(python -c "import json; print(json.load(open('gpurun_out/pmc_${TAG}_FETCH_SIZE.json'))['config']['workload'].split()[0])")
python scripts/pmc_summarize.py $(find gpurun_out/pmc_${TAG}_FETCH_SIZE -name "*counter_collection.csv") $(find gpurun_out/pmc_${TAG}_WRITE_SIZE -name "*counter_collection.csv") $KEYS gpurun_out/pmc_$TAG.json $WL
find gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE -name "*.csv" -size +2M -delete
