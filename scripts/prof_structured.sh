#!/bin/bash
# usage: scripts/prof_structured.sh <tag> <lg_n> "<kind>"   -- rocprofv3 kernel stats of scripts/time_structured.py for ONE scalar kind
set -e
tag=$1; lg=$2; kind=$3
cd /tmp && export TMPDIR=/tmp
export TS_KINDS="$kind"
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$GRAFT_REPO_ROOT/scripts/time_structured.py" "$lg" > "$out/run.log" 2>&1
f=$(find "$out" -name '*kernel_stats.csv' | head -1)
cp "$f" "$GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv"
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print("%-70s calls %5s avg %9.1f us  max %9.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
