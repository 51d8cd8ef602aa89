#!/bin/bash
# A/B on ONE box, everything the driver's line shows: ab/libbpmsm_prev.so against the tree's library.  usage: bash scripts/ab_all.sh [rounds]
for round in $(seq 1 ${1:-3}); do for tag in prev new; do
  if [ $tag = prev ]; then export BPMSM_SO=$PWD/ab/libbpmsm_prev.so; else unset BPMSM_SO; fi
  python bench.py --steps 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['configs']; sw=d['sweep']
print('$tag headline %.3f ms | 2^16 %.3f 2^18 %.3f | cfg1 %.2f/%.2f | cfg3 prove %.2f verify %.2f tables %.2f | cfg5 msm %.3f ipp %.2f/%.2f | h2d %.3f' % (d['ms_per_step'], sw['2^16']['ms'], sw['2^18']['ms'], c['cfg1']['create_ms'], c['cfg1']['verify_ms'], c['cfg3_e2e']['prove_ms'], c['cfg3_e2e']['verify_ms'], c['cfg3_e2e']['with_precomputed_generator_tables']['prove_ms'], c['cfg5']['msm_ms'], c['cfg5']['ipp_create_ms'], c['cfg5']['ipp_verify_ms'], d['with_scalar_h2d']['ms_per_step']))"
  python scripts/time_ipp.py 0 16 2>/dev/null | grep curve= | sed "s/^/$tag /" | cut -c1-60
done; done
