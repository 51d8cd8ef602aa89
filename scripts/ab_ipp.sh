#!/bin/bash
# A/B on ONE box for the inner-product prover: ab/libbpmsm_prev.so against the tree's library.  usage: bash scripts/ab_ipp.sh [lg sizes] [table width | none]
SIZES=${1:-13,16}; TW=${2:-16}
[ "$TW" != none ] && export TIME_IPP_TABLES=$TW
for round in 1 2 3; do
  BPMSM_SO=$PWD/ab/libbpmsm_prev.so python scripts/time_ipp.py ${CURVE:-0} $SIZES 2>/dev/null | grep "curve=" | sed "s/^/prev /"
  python scripts/time_ipp.py ${CURVE:-0} $SIZES 2>/dev/null | grep "curve=" | sed "s/^/new  /"
done
