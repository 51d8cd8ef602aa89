#!/bin/bash
# Round profiles of the DEFAULT bench command (python3 bench.py ...): kernel-trace stats, then counter passes (each its own
# run, counters only -- never combined with tracing domains).  Run on the GPU box from the repo root:
#     scripts/profile_bench.sh r03
# writes gpurun_out/<tag>_bench_n1_{kernel_stats.csv,under_rocprof.json,pmc_hbm.json,pmc_sq.json}; copy them to profiles/.
set -e
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out
args="--steps 10 --warmup 2 --no-cpu-baseline --no-extras"
# clocks / power state of THIS box next to the profile (box-to-box spread of the dominant kernel was 11 % in round 2 and could not be
# attributed): rocm-smi before the run (idle) and right after it
smi() { { date -u +"%Y-%m-%d %H:%M:%S UTC  $1"; rocm-smi --showclocks --showpower --showperflevel --showtemp 2>&1 | grep -v "^$\|====\|WARNING"; } >> "$out/${tag}_bench_n1_rocm_smi.txt"; }
rm -f "$out/${tag}_bench_n1_rocm_smi.txt"
smi "before the kernel-trace run"
# 1. kernel trace + stats of the default command
d=$out/prof_${tag}_trace; rm -rf "$d"; mkdir -p "$d"
# sample the clocks WHILE the bench runs (after it, sclk has already dropped): a background loop on this shell, stopped by its PID
( while true; do { date -u +"%H:%M:%S.%N UTC  during the kernel-trace run"; rocm-smi --showclocks --showpower 2>&1 | grep "sclk\|mclk\|Power"; } >> "$out/${tag}_bench_n1_rocm_smi.txt"; sleep 0.3; done ) &
sampler=$!
trap 'kill $sampler 2>/dev/null || true' EXIT      # the script runs under set -e: a failing profiler run must not leave the sampler polling rocm-smi
rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 "$root/bench.py" $args --steps 60 > "$out/${tag}_bench_n1_under_rocprof.json" 2> "$d/stderr.log"
kill $sampler 2>/dev/null || true
wait $sampler 2>/dev/null || true
cp "$(find "$d" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_bench_n1_kernel_stats.csv"
smi "after the kernel-trace run"
# 2. counter passes
pass() {  # name, counters
    local dd=$out/prof_${tag}_$1; rm -rf "$dd"; mkdir -p "$dd"
    rocprofv3 --pmc $2 --output-format csv -d "$dd" -- python3 "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-extras > "$dd/stdout.log" 2> "$dd/stderr.log"
    echo "pass $1 done"
}
pass fetch "FETCH_SIZE"
pass write "WRITE_SIZE"
pass sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
pass sq2 "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE GRBM_COUNT"
pass tcc "TCC_HIT_sum TCC_MISS_sum"
python3 - "$out" "$tag" <<'PY'
import csv, glob, json, os, subprocess, sys, time, collections
out, tag = sys.argv[1], sys.argv[2]
def load(name):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out, "prof_%s_%s" % (tag, name), "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
rev = os.environ.get("BP_GIT_REV", "")
if not rev:
    try:
        rev = subprocess.check_output(["git", "-C", os.path.dirname(out), "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        rev = "worktree"
taken = time.strftime("%Y-%m-%d %H:%M UTC", time.gmtime()) + ", tree " + rev
hbm = {"command": "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify  (one pass per counter)",
       "units": "FETCH_SIZE / WRITE_SIZE in KiB as reported by rocprofv3, per-launch averages, RAW (the guide's x2 read-side correction is calibrated for wide "
                "streaming reads and is not applied: k_accumulate gathers 96-byte rows)", "taken": taken, "kernels": {}}
for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    for k, d in load(name).items():
        v = d.get(ctr, [])
        if v:
            e = hbm["kernels"].setdefault(k, {})
            e[ctr + "_KiB_avg"] = round(sum(v) / len(v), 1)
            e["launches"] = len(v)
json.dump(hbm, open(os.path.join(out, "%s_bench_n1_pmc_hbm.json" % tag), "w"), indent=1, sort_keys=True)
sq = {"command": "rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify  (separate passes: SQ wave "
                 "accounting; SQ instruction mix + GRBM; TCC)", "units": "per-launch averages; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles summed over waves", "taken": taken,
      "kernels": {}}
for name in ("sq1", "sq2", "tcc"):
    for k, d in load(name).items():
        e = sq["kernels"].setdefault(k, {})
        for c, v in d.items():
            e[c] = round(sum(v) / len(v), 1)
            e["launches"] = len(v)
acc = next((v for k, v in sq["kernels"].items() if k.startswith("void bp::k_accumulate<bp::Bls381") or k.startswith("bp::k_accumulate<bp::Bls381")), None)
if acc and acc.get("SQ_WAVE_CYCLES"):
    wc = acc["SQ_WAVE_CYCLES"]
    sq["derived_k_accumulate"] = {"issuing_fraction_of_wave_cycles": round(acc.get("SQ_ACTIVE_INST_ANY", 0) / wc, 4),
                                  "waiting_for_busy_pipe_fraction": round(acc.get("SQ_WAIT_INST_ANY", 0) / wc, 4),
                                  "waiting_on_memory_or_barrier_fraction": round(acc.get("SQ_WAIT_ANY", 0) / wc, 4),
                                  "L2_hit_rate": round(acc.get("TCC_HIT_sum", 0) / max(1.0, acc.get("TCC_HIT_sum", 0) + acc.get("TCC_MISS_sum", 0)), 4)}
json.dump(sq, open(os.path.join(out, "%s_bench_n1_pmc_sq.json" % tag), "w"), indent=1, sort_keys=True)
for k, v in sorted(hbm["kernels"].items()):
    if "accumulate" in k or "fine_place" in k or "reduce" in k:
        print(k[:70], v)
print(sq.get("derived_k_accumulate"))
PY
