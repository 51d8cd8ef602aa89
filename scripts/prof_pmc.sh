#!/bin/bash
# usage: scripts/prof_pmc.sh <tag> <lg_n> "<counters>" [c] [reps]  -- rocprofv3 --pmc pass (counters only; no tracing domains) -> per-kernel averages
set -e
tag=$1; lg=$2; ctrs=$3; c=${4:-0}; reps=${5:-10}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --pmc $ctrs --output-format csv -d "$out" -- python3 "$GRAFT_REPO_ROOT/scripts/prof_one.py" "$lg" "$c" "$reps" > "$out/run.log" 2>&1
f=$(find "$out" -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "digit" in k or "accumulate" in k or "fine_place" in k:
        print(k)
        for c, v in d.items():
            print("   %-28s avg %14.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
