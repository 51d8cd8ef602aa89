#!/bin/bash
# A/B of tuning knobs on the 2^20 MSM (development aid)
for wps in 2 3 4; do echo "BP_ACC_WPS=$wps"; BP_ACC_WPS=$wps python scripts/time_msm.py 20 0 2>&1 | grep "n=2"; done
for m in 1 2 4 8 16; do echo "BP_REDUCE_M=$m"; BP_REDUCE_M=$m python scripts/time_msm.py 20 0 2>&1 | grep "n=2"; done
