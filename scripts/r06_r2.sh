#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_parity.py tests/test_gpu_verify.py -m gpu -q --maxfail=5 -p no:cacheprovider > gpurun_out/r06/pytest_r2.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06/pytest_r2.txt
tail -n 3 gpurun_out/r06/pytest_r2.txt
python scripts/gpu_lowcomplexity.py > gpurun_out/r06/lowcx.txt 2>&1; echo "lowcx rc $?"; tail -n 3 gpurun_out/r06/lowcx.txt
python scripts/gpu_fuzz.py 1000 96 > gpurun_out/r06/fuzz_r2.txt 2>&1; tail -n 1 gpurun_out/r06/fuzz_r2.txt
DEBWT_TRACE_SORT=1 python bench.py --workload real10x3G --steps 3 --warmup 1 --no-cpu-baseline --h2h-reps 0 > gpurun_out/r06/bench_real10x3G_b.json 2> gpurun_out/r06/bench_real10x3G_b.err; echo "R rc $?"
python -c "
import json
j=json.load(open('gpurun_out/r06/bench_real10x3G_b.json')); print('R', j['ms_per_step'], j['stages_ms'], (j.get('check') or {}).get('inverse_bwt_ok'))"
DEBWT_OVER_PLAIN=1 python bench.py --workload real10x3G --steps 3 --warmup 1 --no-cpu-baseline --h2h-reps 0 --no-check > gpurun_out/r06/bench_real10x3G_plain.json 2> /dev/null; echo "R plain rc $?"
python -c "
import json
j=json.load(open('gpurun_out/r06/bench_real10x3G_plain.json')); print('R plain', j['ms_per_step'], j['stages_ms'])"
