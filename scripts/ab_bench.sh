#!/bin/bash
# A/B on ONE box of the headline MSM, mid-size MSMs and config 3: ab/libbpmsm_prev.so against the tree's library.  usage: bash scripts/ab_bench.sh [rounds]
for round in $(seq 1 ${1:-3}); do
  for tag in prev new; do
    if [ $tag = prev ]; then export BPMSM_SO=$PWD/ab/libbpmsm_prev.so; else unset BPMSM_SO; fi
    python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag headline ms_per_step %.4f  k_accumulate GB/s %.2f' % (d['ms_per_step'], d['roofline']['achieved']))"
    TIME_MSM_NOTIMING=1 python scripts/time_msm.py 16,17,18,20 2>/dev/null | grep -E "n=2" | sed "s/^/$tag /" | cut -c1-40
    python scripts/time_msm.py 20 2>/dev/null | grep -E "n=2" | sed "s/^/$tag /" | cut -c30-160
  done
done
