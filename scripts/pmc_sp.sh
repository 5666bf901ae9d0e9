#!/bin/bash
# FETCH_SIZE and the read-request split (TCC_EA0_RDREQ, its 32-byte part) of the SP flags pass for the two node-table
# addressings (tune 0: by minimizer, tune 16384: by node hash): usage scripts/pmc_sp.sh TAG
TAG=${1:-r03}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "TCC_EA0_RDREQ[A-Za-z0-9_]*" | sort -u > $ROOT/gpurun_out/pmc_${TAG}_counters.txt
for T in 0 16384; do
for C in FETCH_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  N=$(echo $C | tr ' ' '+')
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}_t${T}_$N -o c -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-check --h2h-reps 0 --tune $T > $ROOT/gpurun_out/pmc_${TAG}_t${T}_$N.json 2> $ROOT/gpurun_out/pmc_${TAG}_t${T}_$N.err || echo "pass $T $C failed"
done
done
cd $ROOT
python - <<'PY'
import csv, glob, re, collections, json, sys
res = {}
for d in sorted(glob.glob("gpurun_out/pmc_*_t*_*")):
    if not d.endswith(("FETCH_SIZE", "_sum")): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
            if name.startswith(("k_sp_flags", "k_blue_refine", "rs_hist_kernel<0, 0", "rs_scatter_kernel<0, 0", "k_sp_emit")):
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
        res[d.split("/")[-1]] = {k: {c: {"launches": len(v), "mean_of_large": sum(x for x in v if x >= 0.5 * max(v)) / max(1, len([x for x in v if x >= 0.5 * max(v)]))}
                                     for c, v in cs.items()} for k, cs in acc.items()}
json.dump(res, open("gpurun_out/pmc_sp_summary.json", "w"), indent=1)
for d, t in res.items():
    print(d)
    for k, cs in t.items(): print("   ", k, {c: round(v["mean_of_large"]) for c, v in cs.items()})
PY
find gpurun_out -path "*pmc_${TAG}_t*" -name "*.csv" -size +2M -delete
