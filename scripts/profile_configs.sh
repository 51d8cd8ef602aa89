#!/bin/bash
# kernel-trace stats of the other BASELINE configs (IPP n = 64 / 2^16 with the one-launch small MSM, R1CS end to end, BN254):
#     scripts/profile_configs.sh r02  ->  gpurun_out/<tag>_bench_configs_kernel_stats.csv
set -e
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
d=$root/gpurun_out/prof_${tag}_configs; rm -rf "$d"; mkdir -p "$d"
rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 "$root/bench_configs.py" cfg1 cfg3 cfg3_e2e cfg5 > "$root/gpurun_out/${tag}_bench_configs_under_rocprof.json" 2> "$d/stderr.log"
cp "$(find "$d" -name '*kernel_stats.csv' | head -1)" "$root/gpurun_out/${tag}_bench_configs_kernel_stats.csv"
python3 - "$root/gpurun_out/${tag}_bench_configs_kernel_stats.csv" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:30]:
    print("%-72s calls %6s avg %9.1f us  total %6.2f%%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
