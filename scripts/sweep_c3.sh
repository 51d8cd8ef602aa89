#!/bin/bash
for tgt in 262144 524288; do
  echo "TM_TASK_TARGET=$tgt"
  for a in "14 0 11,12,13" "16 0 12,13,14,15" "17 0 13,14,15,16" "18 0 14,15,16" "19 0 15,16" "20 0 15,16"; do TM_TASK_TARGET=$tgt python scripts/time_msm.py $a 2>&1 | grep "n=2"; done
done
