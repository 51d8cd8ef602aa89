#!/bin/bash
# usage (GPU box, repo root): scripts/prof_ipp.sh <tag> <curve> <lg_n> [table width | none]
# rocprofv3 kernel stats of scripts/time_ipp.py (3 create + verify, one round-API proof) -> gpurun_out/<tag>_kernel_stats.csv
set -e
tag=$1; curve=${2:-0}; lg=${3:-16}; tw=${4:-none}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
if [ "$tw" != "none" ]; then export TIME_IPP_TABLES=$tw; fi
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$root/scripts/time_ipp.py" "$curve" "$lg" > "$out/run.log" 2>&1
cat "$out/run.log" | grep -v amdgpu.ids
f=$(find "$out" -name '*kernel_stats.csv' | head -1)
cp "$f" "$root/gpurun_out/${tag}_kernel_stats.csv"
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:26]:
    print("%-64s calls %5s avg %9.1f us  total %6.2f%%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
