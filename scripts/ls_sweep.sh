for cfg in "384 8" "192 8" "192 16" "128 16" "256 16"; do
  set -- $cfg
  DEBWT_LS_BIN_ROWS=$1 DEBWT_LS_OVERSAMPLE=$2 DEBWT_TRACE_LARGE=1 python bench.py --workload real10x3G --steps 2 --warmup 1 --no-cpu-baseline --no-check --h2h-reps 0 > gpurun_out/ls_$1_$2.json 2> gpurun_out/ls_$1_$2.err
  echo "bin_rows $1 oversample $2: $(python -c "import json;j=json.loads(open('gpurun_out/ls_$1_$2.json').read().strip().splitlines()[-1]);print(j['ms_per_step'], j['stages_ms']['ms_blue'])")"
  grep "large blocks: round [0-3] " gpurun_out/ls_$1_$2.err | tail -4
done
