ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | grep -E "LDS|WAIT_INST|ACTIVE_INST|INSTS_|BANK" | tr '\n' ' ' > $ROOT/gpurun_out/sq_counters.txt
timeout -k 10 400 rocprofv3 --kernel-trace --pmc ${SQ_SET:-SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE} --output-format csv -d $ROOT/gpurun_out/pmc_sq2 -o c -- python3 $ROOT/bench.py --workload grch38_3.1G --steps 1 --warmup 1 --no-cpu-baseline --no-check --h2h-reps 0 > $ROOT/gpurun_out/pmc_sq2.json 2> $ROOT/gpurun_out/pmc_sq2.err || echo failed
cd $ROOT
python - <<PY
import csv, glob, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc_sq2/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
        acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
for k, m in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]:
    wc = m.get("SQ_WAVE_CYCLES", 1) or 1
    print("%-40s" % k[:40], {c.replace("SQ_", ""): round(v / wc, 3) for c, v in m.items() if c != "SQ_WAVE_CYCLES"})
PY
find gpurun_out/pmc_sq2 -name "*.csv" -size +2M -delete
