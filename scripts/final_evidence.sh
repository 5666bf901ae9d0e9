#!/bin/bash
# end-of-round evidence with the final code: driver-command bench, the other workloads, both sharded key paths in a
# process group of one.  usage: scripts/final_evidence.sh TAG   (results in gpurun_out/, copy to profiles/)
TAG=${1:-r02_final}
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_30G_driver_cmd.json 2> gpurun_out/${TAG}_bench_30G_driver_cmd.err || exit 1
echo "driver cmd done: $(date)"
for w in pan4x3.1G grch38_3.1G uniform_3.1G real_3.1G chr1_250M ecoli_4.6M; do
  python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_$w.json 2> gpurun_out/${TAG}_bench_$w.err || exit 1
  echo "$w done"
done
for m in rescan exchange; do
  python bench.py --force-sharded --mode $m --steps 3 --warmup 1 --h2h-reps 0 --no-cpu-baseline > gpurun_out/${TAG}_bench_30G_keys_${m}_group_of_one.json 2> gpurun_out/${TAG}_bench_30G_keys_${m}.err || exit 1
  echo "$m done"
done
