#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider > gpurun_out/r06/pytest_gpu2.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06/pytest_gpu2.txt
tail -n 4 gpurun_out/r06/pytest_gpu2.txt
DEBWT_TRACE_SORT=1 DEBWT_TRACE_LARGE=1 python bench.py --workload real10x3G --steps 1 --warmup 1 --no-cpu-baseline --h2h-reps 0 --no-check > gpurun_out/r06/bench_real10x3G_trace.json 2> gpurun_out/r06/bench_real10x3G_trace.err; echo "R rc $?"
grep -c . gpurun_out/r06/bench_real10x3G_trace.err
bash scripts/prof_30g.sh r06R --workload real10x3G > gpurun_out/r06/prof_R.txt 2>&1 || exit 1
head -n 40 gpurun_out/r06/prof_R.txt
