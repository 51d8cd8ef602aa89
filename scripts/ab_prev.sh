#!/bin/bash
# A/B on ONE box: ab/libbpmsm_prev.so (a copy of the previous build) against the tree's library.  usage: bash scripts/ab_prev.sh [sizes] [curve]
SIZES=${1:-19,20}; CURVE=${2:-0}
for round in 1 2 3; do
  BPMSM_SO=$PWD/ab/libbpmsm_prev.so python scripts/time_msm.py $SIZES $CURVE 0 2>/dev/null | cut -c1-250 | sed "s/^/prev /"
  python scripts/time_msm.py $SIZES $CURVE 0 2>/dev/null | cut -c1-250 | sed "s/^/new  /"
done
