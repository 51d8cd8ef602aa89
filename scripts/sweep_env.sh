#!/bin/bash
# A/B of the bucket-reduce chain length on the 2^20 MSM (development aid; the knob is bp_ctx_set_tuning through scripts/time_msm.py)
for m in 1 2 4 8 16; do echo "TM_REDUCE_M=$m"; TM_REDUCE_M=$m python scripts/time_msm.py 20 0 2>&1 | grep "n=2"; done
