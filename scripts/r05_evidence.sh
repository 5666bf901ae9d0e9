#!/bin/bash
# Round-5 evidence with the committed code (results in gpurun_out/, copy to profiles/).  usage: scripts/r05_evidence.sh PART
#   1: the driver's command (30 Gbp, 20 steps)                       2: kernel stats + PMC passes at 30 Gbp (P)
#   3: distribution R at 30 Gbp: bench line + kernel stats            4: the other named workloads
PART=${1:-1}
case $PART in
1) python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_30G_driver_command.json 2> gpurun_out/r05_bench_30G_driver_command.err
   tail -c 400 gpurun_out/r05_bench_30G_driver_command.json ;;
2) bash scripts/prof_30g.sh r05_30G | head -40 && bash scripts/pmc_30g.sh r05_30G && ls gpurun_out/pmc_r05_30G.json ;;
3) python bench.py --workload real10x3G --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r05_bench_real10x3G.json 2> gpurun_out/r05_bench_real10x3G.err
   tail -c 300 gpurun_out/r05_bench_real10x3G.json; bash scripts/prof_30g.sh r05_real10x3G --workload real10x3G | head -45 ;;
4) for w in pan4x3.1G grch38_3.1G uniform_3.1G real_3.1G chr1_250M ecoli_4.6M; do
     python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r05_bench_$w.json 2> gpurun_out/r05_bench_$w.err || exit 1
     python -c "import json,sys; j=json.loads(open('gpurun_out/r05_bench_$w.json').read().strip().splitlines()[-1]); print('$w', j['ms_per_step'], j['value'], j['check']['inverse_bwt_ok'], j['host_to_host']['value'] if j.get('host_to_host') else None)"
   done ;;
esac
