#!/bin/bash
# Where the waves of every kernel of a build spend their cycles (SQ counters, one pass): usage scripts/pmc_sq.sh TAG [bench args]
TAG=${1:-r03}; shift || true
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}_sq -o c -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-check --h2h-reps 0 "$@" > $ROOT/gpurun_out/pmc_${TAG}_sq.json 2> $ROOT/gpurun_out/pmc_${TAG}_sq.err || echo "pass failed"
cd $ROOT
python - <<PY
import csv, glob, re, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_${TAG}_sq/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, cs in acc.items():
    big = max(cs.get("SQ_WAVE_CYCLES", [0]))
    sel = [i for i, v in enumerate(cs.get("SQ_WAVE_CYCLES", [])) if v >= 0.5 * big]
    m = {c: sum(v[i] for i in sel) / max(1, len(sel)) for c, v in cs.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    out[k] = {"launches": len(cs.get("SQ_WAVE_CYCLES", [])), "wave_cycles": wc,
              "wait_any": round(m.get("SQ_WAIT_ANY", 0) / wc, 3), "wait_inst": round(m.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
              "active_inst": round(m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
              "valu_per_wave": round(m.get("SQ_INSTS_VALU", 0) / max(1, m.get("SQ_WAVES", 1)), 1),
              "lds_per_wave": round(m.get("SQ_INSTS_LDS", 0) / max(1, m.get("SQ_WAVES", 1)), 1), "waves": m.get("SQ_WAVES", 0)}
json.dump(out, open("gpurun_out/pmc_${TAG}_sq_summary.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["wave_cycles"])[:24]:
    print("%-46s wave_cycles %.3e wait_any %.2f wait_inst %.2f active %.2f valu/wave %9.0f lds/wave %8.0f waves %.3g" % (k[:46], v["wave_cycles"], v["wait_any"], v["wait_inst"], v["active_inst"], v["valu_per_wave"], v["lds_per_wave"], v["waves"]))
PY
find gpurun_out/pmc_${TAG}_sq -name "*.csv" -size +2M -delete
