#!/bin/bash
# Round-6 evidence with the committed code (results in gpurun_out/, copy to profiles/).  usage: scripts/r06_evidence.sh PART
#   1: the driver's command (30 Gbp, 20 steps)                       2: kernel stats + PMC passes at 30 Gbp (P)
#   3: distribution R at 30 Gbp: bench line + kernel stats            4: the other named workloads
#   5: the sharded code path in a process group of one (both key paths)   6: the GPU suite
PART=${1:-1}
mkdir -p gpurun_out
case $PART in
1) python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_30G_driver_command.json 2> gpurun_out/r06_bench_30G_driver_command.err
   tail -c 400 gpurun_out/r06_bench_30G_driver_command.json ;;
2) bash scripts/prof_30g.sh r06_30G | head -40 && bash scripts/pmc_30g.sh r06_30G && ls gpurun_out/pmc_r06_30G.json ;;
3) python bench.py --workload real10x3G --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r06_bench_real10x3G.json 2> gpurun_out/r06_bench_real10x3G.err
   tail -c 300 gpurun_out/r06_bench_real10x3G.json; bash scripts/prof_30g.sh r06_real10x3G --workload real10x3G | head -45 ;;
4) for w in pan4x3.1G grch38_3.1G uniform_3.1G real_3.1G chr1_250M ecoli_4.6M; do
     python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r06_bench_$w.json 2> gpurun_out/r06_bench_$w.err || exit 1
     python -c "import json,sys; j=json.loads(open('gpurun_out/r06_bench_$w.json').read().strip().splitlines()[-1]); print('$w', j['ms_per_step'], j['value'], j['check']['inverse_bwt_ok'], j['host_to_host']['value'] if j.get('host_to_host') else None)"
   done ;;
5) for m in rescan exchange; do
     python bench.py --force-sharded --mode $m --steps 3 --warmup 1 --h2h-reps 0 --no-cpu-baseline > gpurun_out/r06_bench_30G_keys_${m}_group_of_one.json 2> gpurun_out/r06_bench_30G_keys_${m}.err || exit 1
     python -c "import json; j=json.load(open('gpurun_out/r06_bench_30G_keys_${m}_group_of_one.json')); print('$m', j['ms_per_step'], j['stages_ms'], j['check']['inverse_bwt_ok'])"
   done ;;
6) python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r06_pytest_gpu.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06_pytest_gpu.txt; tail -n 4 gpurun_out/r06_pytest_gpu.txt
   python -c "import __graft_entry__ as g; g.smoke()" ;;
esac
