#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=5 -p no:cacheprovider > gpurun_out/r06/pytest_mz.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06/pytest_mz.txt
tail -n 3 gpurun_out/r06/pytest_mz.txt
python scripts/gpu_fuzz.py 2000 95 > gpurun_out/r06/fuzz_mz.txt 2>&1; tail -n 1 gpurun_out/r06/fuzz_mz.txt
timeout -k 10 600 python scripts/gpu_kinvariance_30g.py grch38_3.1G 32,24,20,17 > gpurun_out/r06/k_range_grch38_c.txt 2>&1; echo "k range rc $?"; tail -n 10 gpurun_out/r06/k_range_grch38_c.txt
