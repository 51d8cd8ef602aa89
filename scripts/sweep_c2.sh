#!/bin/bash
for a in "16 0 12,13,14,15" "17 0 13,14,15,16" "18 0 13,14,15,16" "19 0 14,15,16" "20 0 14,15,16" "22 0 15,16"; do python scripts/time_msm.py $a 2>&1 | grep "n=2"; done
