#!/bin/bash
# kernel durations of tools/pmc_layers.py cases (one rocprofv3 kernel trace, no counters).  usage: tools/gpu_trace_layers.sh <tag> "<case substring>"
set -u
TAG=$1; ONLY=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/pmc_layers.py --manifest $OUT/manifest.json --only "$ONLY" > $OUT/trace.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, re
out = sys.argv[1]
man = json.load(open(out + "/manifest.json"))
rows = []
for p in glob.glob(out + "/trace/**/*_kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
i = 0
for c in man["cases"]:
    mine = []
    while i < len(rows) and len(mine) < c["launches"]:
        if c["kernel"] + "<" in rows[i]["Kernel_Name"] or c["kernel"] + "(" in rows[i]["Kernel_Name"]:
            mine.append(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]))
        i += 1
    mine = mine[c["warm"]:]
    us = sum(mine) / max(len(mine), 1) / 1e3
    print("%-50s %8.1f us %7.1f TF" % (c["case"][:50], us, c["direct_flops_per_launch"] / max(us, 1e-9) / 1e6))
PY
rm -rf $OUT/trace
