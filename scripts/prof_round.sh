# Evidence of a round on one MI355X: bench line, rocprofv3 kernel stats of the same command, two PMC passes and their summary (copy into profiles/).
set -e
cd /root/repo
timeout -k 10 400 python bench.py > gpurun_out/v13_bench.json 2> gpurun_out/v13_bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_v13 -o v13 -- python3 /root/repo/bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline > /root/repo/gpurun_out/v13_under.json 2> /root/repo/gpurun_out/v13_under.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /root/repo/gpurun_out/pmc_v13_f -o f -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/pmc_f.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /root/repo/gpurun_out/pmc_v13_w -o w -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/pmc_w.log 2>&1
cd /root/repo
python scripts/pmc_summarize.py gpurun_out/pmc_v13_f/f_counter_collection.csv gpurun_out/pmc_v13_w/w_counter_collection.csv 249999970 gpurun_out/pmc_v13.json
cat gpurun_out/v13_bench.json
