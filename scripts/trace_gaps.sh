#!/bin/bash
# Kernel trace of one build: where the GPU waits for the host -- gaps between the end of a launch and the start of the next,
# largest first, with the kernels on both sides.   usage: scripts/trace_gaps.sh WORKLOAD
set -e
W=${1:-pan10x3G}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/ktr_$W -o t -- python3 $ROOT/bench.py --gpus 1 --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-check --h2h-reps 0 > $ROOT/gpurun_out/ktr_$W.json 2> $ROOT/gpurun_out/ktr_$W.err
cd $ROOT
F=$(find gpurun_out/ktr_$W -name "*kernel_trace.csv" | head -1)
python - "$F" > gpurun_out/gaps_$W.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# the last build: from the last rs_hist_ranges_words_kernel on
starts = [i for i, e in enumerate(ev) if "rs_hist_ranges_words" in e[2] or "rs_hist_ranges_kernel" in e[2]]
i0 = starts[-1] if starts else 0
ev = ev[i0:]
span = (max(e[1] for e in ev) - ev[0][0]) / 1e6
busy = 0; cur_end = ev[0][0]; gaps = []
for k, (s, e, n) in enumerate(ev):
    if s > cur_end:
        gaps.append(((s - cur_end) / 1e3, ev[k - 1][2][:50], n[:50]))
    if e > cur_end:
        busy += (e - max(s, cur_end)) / 1e6
        cur_end = e
print("last build: %d launches, span %.1f ms, GPU busy %.1f ms, idle %.1f ms in %d gaps" % (len(ev), span, busy, span - busy, len(gaps)))
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for g, a, b in gaps:
    agg[(a, b)][0] += 1; agg[(a, b)][1] += g
print("gaps by the kernels around them (count, total us), largest first")
for (a, b), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("  %4d %9.0f us   %-50s -> %s" % (c, t, a, b))
PY
find gpurun_out/ktr_$W -name "*.csv" -delete
cat gpurun_out/gaps_$W.txt
