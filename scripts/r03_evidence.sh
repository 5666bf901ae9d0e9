#!/bin/bash
# Round-3 evidence with the final code (results in gpurun_out/, copy to profiles/).  usage: scripts/r03_evidence.sh PART TAG
PART=${1:-1}; TAG=${2:-r03}
case $PART in
1)  # driver-command bench, rocprofv3 kernel stats, PMC passes
    python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_30G_driver_cmd.json 2> gpurun_out/${TAG}_bench_30G_driver_cmd.err || exit 1
    echo "driver cmd done"; tail -c 600 gpurun_out/${TAG}_bench_30G_driver_cmd.json; echo
    bash scripts/prof_30g.sh $TAG > gpurun_out/${TAG}_prof.log 2>&1 || exit 1
    echo "prof done"
    bash scripts/pmc_30g.sh ${TAG}_30G > gpurun_out/${TAG}_pmc.log 2>&1 || exit 1
    echo "pmc done" ;;
2)  # the other workloads, both sharded key paths in a process group of one
    for w in pan4x3.1G grch38_3.1G uniform_3.1G real_3.1G chr1_250M ecoli_4.6M; do
      python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_$w.json 2> gpurun_out/${TAG}_bench_$w.err || exit 1
      echo "$w done"
    done
    for m in rescan exchange; do
      python bench.py --force-sharded --mode $m --steps 3 --warmup 1 --h2h-reps 0 --no-cpu-baseline > gpurun_out/${TAG}_bench_30G_keys_${m}_group_of_one.json 2> gpurun_out/${TAG}_bench_30G_keys_${m}.err || exit 1
      echo "$m done"
    done ;;
3)  # special-region module, the C program end to end at 3.1 Gbp, the reference on whole BASELINE configurations,
    # bench.py --gpus 2 started directly (two ranks sharing the GPU over gloo) at 3.1 Gbp
    python scripts/gpu_special.py 100 > gpurun_out/${TAG}_special_region_device.txt 2>&1 || exit 1
    echo "special done"
    python scripts/gpu_cli_3g.py grch38_3.1G > gpurun_out/${TAG}_cli_3.1G.txt 2>&1 || exit 1
    echo "cli done"
    python bench.py --workload chr1_250M --steps 5 --warmup 2 --cpu-configs > gpurun_out/${TAG}_bench_chr1_250M_cpu_configs.json 2> gpurun_out/${TAG}_cpu_configs.err || exit 1
    echo "cpu configs done"
    python bench.py --gpus 2 --backend gloo --workload grch38_3.1G --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_2_ranks_gloo_3.1G.json 2> gpurun_out/${TAG}_2ranks.err || exit 1
    echo "2 ranks done" ;;
esac
