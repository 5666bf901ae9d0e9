#!/bin/bash
# round 6: parity subset that exercises the changed kernels, then the 30 Gbp bench (kernel statistics) by waves and in lockstep
set -o pipefail
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_parity.py tests/test_gpu_verify.py -m gpu -q --maxfail=5 -p no:cacheprovider > gpurun_out/r06/pytest_ab.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06/pytest_ab.txt
tail -n 3 gpurun_out/r06/pytest_ab.txt
python scripts/gpu_fuzz.py 1500 99 > gpurun_out/r06/fuzz_ab.txt 2>&1; tail -n 1 gpurun_out/r06/fuzz_ab.txt
bash scripts/prof_30g.sh r06d > gpurun_out/r06/prof_waves2.txt 2>&1 || exit 1
DEBWT_SPARSE_LOCKSTEP=1 bash scripts/prof_30g.sh r06e > gpurun_out/r06/prof_lockstep2.txt 2>&1 || exit 1
grep -h "sparse\|k_sp_flags\|hist_ranges\|prefix_hist" gpurun_out/r06/prof_waves2.txt gpurun_out/r06/prof_lockstep2.txt
python -c "
import json
for t in ('r06d','r06e'):
    j=json.load(open('gpurun_out/%s_bench_under_rocprof.json'%t)); print(t, j['ms_per_step'], j['stages_ms'])"
