#!/bin/bash
# The bench-harness tests of N > 1 (extras that are slow, extras that fail) N times in a row on one box: the evidence that
# the orchestration around the measurement has no race left (round 4's watchdog race made one of them fail 1 in ~3).
# usage: scripts/gpu_harness_loop.sh [N=50] [logfile]
N=${1:-50}
LOG=${2:-gpurun_out/r05_harness_loop.txt}
mkdir -p "$(dirname "$LOG")"
: > "$LOG"
pass=0
for i in $(seq 1 "$N"); do
    t0=$(date +%s.%N)
    if python -m pytest tests/test_zz_bench_harness.py -x -q -k "extras" >> "$LOG.detail" 2>&1; then
        pass=$((pass + 1)); verdict=passed
    else
        verdict=FAILED
    fi
    printf "iteration %d: %s (%s s)\n" "$i" "$verdict" "$(python3 -c "import time; print(round(time.time() - $t0, 1))")" | tee -a "$LOG"
done
echo "$pass / $N iterations passed (each: test_bench_extras_cannot_cost_the_result_when_they_are_slow + test_bench_extras_that_fail_are_recorded_and_cost_nothing)" | tee -a "$LOG"
[ "$pass" -eq "$N" ]
