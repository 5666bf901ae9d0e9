#!/bin/bash
# rehearsal of bench.py's N>1 path on the ONE GPU of a gpurun box: the ranks share the GPU, collectives over gloo
# (RCCL refuses two ranks on one device), plus the RCCL code path itself in a process group of one.
# usage: scripts/gpu_bench_multi.sh [workload=chr1_250M] [ranks=3]
set -e
W=${1:-chr1_250M}; R=${2:-3}
mkdir -p gpurun_out
python bench.py --gpus 1 --force-sharded --workload $W --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/multi_nccl1.json
tail -c 1200 gpurun_out/multi_nccl1.json; echo
for mode in auto exchange; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node $R --master-addr 127.0.0.1 --master-port 29577 \
    bench.py --gpus $R --backend gloo --mode $mode --workload $W --steps 2 --warmup 1 > gpurun_out/multi_gloo_$mode.json
tail -c 1500 gpurun_out/multi_gloo_$mode.json; echo
done
