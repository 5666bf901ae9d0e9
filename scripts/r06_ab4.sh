#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=5 -p no:cacheprovider -k "range or golden or multi or midsize or adversarial" > gpurun_out/r06/pytest_ab4.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06/pytest_ab4.txt
tail -n 3 gpurun_out/r06/pytest_ab4.txt
python scripts/gpu_fuzz.py 1500 94 > gpurun_out/r06/fuzz_ab4.txt 2>&1; tail -n 1 gpurun_out/r06/fuzz_ab4.txt
bash scripts/prof_30g.sh r06h > gpurun_out/r06/prof_h.txt 2>&1 || exit 1
grep -h "sparse\|total kernel" gpurun_out/r06/prof_h.txt
python -c "
import json
j=json.load(open('gpurun_out/r06h_bench_under_rocprof.json')); print(j['ms_per_step'], j['stages_ms'])"
