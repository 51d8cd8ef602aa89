#!/bin/bash
# A/B on ONE box of bp_r1cs_prove at BASELINE config 3, phase by phase (BP_PROFILE): ab/libbpmsm_prev.so against the tree's library
for round in 1 2 3; do for tag in prev new; do
  if [ $tag = prev ]; then export BPMSM_SO=$PWD/ab/libbpmsm_prev.so; else unset BPMSM_SO; fi
  BP_PROFILE=1 python scripts/prof_cfg3_phases.py 2>&1 | grep -E "r1cs prove|prove_ms" | tail -2 | sed "s/^/$tag /" | cut -c1-420
done; done
