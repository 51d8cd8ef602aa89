#!/bin/bash
# usage: scripts/prof_stats.sh <tag> <lg_n> [c] [reps]   -- rocprofv3 kernel stats of one MSM shape -> gpurun_out/<tag>_kernel_stats.csv (+ a short table on stdout)
set -e
tag=$1; lg=$2; c=${3:-0}; reps=${4:-20}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$GRAFT_REPO_ROOT/scripts/prof_one.py" "$lg" "$c" "$reps" > "$out/run.log" 2>&1
f=$(find "$out" -name '*kernel_stats.csv' | head -1)
cp "$f" "$GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv"
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:22]:
    print("%-70s calls %5s avg %9.1f us  total %6.2f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
