#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r06
DEBWT_TRACE_ALLOC=1 timeout -k 10 900 python scripts/gpu_kinvariance_30g.py grch38_3.1G 32,16,12 > gpurun_out/r06/k_range_grch38_b.txt 2>&1; echo "k range rc $?"; grep -v "^ensure" gpurun_out/r06/k_range_grch38_b.txt | tail -n 12; grep -c "^ensure" gpurun_out/r06/k_range_grch38_b.txt; grep "reclaim" gpurun_out/r06/k_range_grch38_b.txt | head
timeout -k 10 600 python scripts/ingest_gz_bench.py 1000 > gpurun_out/r06/ingest_gz.txt 2>&1; tail -n 10 gpurun_out/r06/ingest_gz.txt
