#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_parity.py tests/test_gpu_verify.py tests/test_sharded.py -m gpu -q --maxfail=5 -p no:cacheprovider > gpurun_out/r06/pytest_ab2.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06/pytest_ab2.txt
tail -n 3 gpurun_out/r06/pytest_ab2.txt
python scripts/gpu_fuzz.py 1500 98 > gpurun_out/r06/fuzz_ab2.txt 2>&1; tail -n 1 gpurun_out/r06/fuzz_ab2.txt
bash scripts/prof_30g.sh r06f > gpurun_out/r06/prof_f.txt 2>&1 || exit 1
grep -h "sparse\|k_blue\|total kernel" gpurun_out/r06/prof_f.txt
python -c "
import json
j=json.load(open('gpurun_out/r06f_bench_under_rocprof.json')); print(j['ms_per_step'], j['stages_ms'])"
