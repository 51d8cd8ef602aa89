#!/bin/bash
# A/B on ONE box: the round-2 library (ab/libbpmsm_r2.so, built from the round-2 tag) against the tree's library.
# usage (on the GPU box): bash scripts/ab_msm.sh OUTDIR [sizes]
OUT=${1:-gpurun_out/ab}; SIZES=${2:-16,19,20}
mkdir -p "$OUT"
rocm-smi --showclocks --showpower > "$OUT/smi_before.txt" 2>&1
for round in 1 2; do
  BPMSM_SO=$PWD/ab/libbpmsm_r2.so python scripts/time_msm.py $SIZES 0 0 2>/dev/null | sed "s/^/r2  /" >> "$OUT/ab.log"
  python scripts/time_msm.py $SIZES 0 0 2>/dev/null | sed "s/^/new /" >> "$OUT/ab.log"
done
rocm-smi --showclocks --showpower > "$OUT/smi_after.txt" 2>&1
cat "$OUT/ab.log"
