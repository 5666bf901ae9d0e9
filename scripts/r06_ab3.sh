#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=5 -p no:cacheprovider -k "range or golden or multi or midsize or adversarial" > gpurun_out/r06/pytest_ab3.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06/pytest_ab3.txt
tail -n 3 gpurun_out/r06/pytest_ab3.txt
python scripts/gpu_fuzz.py 1500 97 > gpurun_out/r06/fuzz_ab3.txt 2>&1; tail -n 1 gpurun_out/r06/fuzz_ab3.txt
bash scripts/prof_30g.sh r06g > gpurun_out/r06/prof_g.txt 2>&1 || exit 1
grep -h "sparse\|total kernel" gpurun_out/r06/prof_g.txt
python -c "
import json
j=json.load(open('gpurun_out/r06g_bench_under_rocprof.json')); print(j['ms_per_step'], j['stages_ms'])"
timeout -k 10 600 python scripts/gpu_kinvariance_30g.py grch38_3.1G 32,24,20,16 > gpurun_out/r06/k_range_grch38.txt 2>&1; echo "k range rc $?"; tail -n 12 gpurun_out/r06/k_range_grch38.txt
python bench.py --workload real10x3G --steps 3 --warmup 1 --no-cpu-baseline --h2h-reps 0 > gpurun_out/r06/bench_real10x3G_a.json 2> gpurun_out/r06/bench_real10x3G_a.err; echo "R rc $?"
python -c "
import json
j=json.load(open('gpurun_out/r06/bench_real10x3G_a.json')); print('R', j['ms_per_step'], j['stages_ms'], (j.get('check') or {}).get('inverse_bwt_ok'))"
