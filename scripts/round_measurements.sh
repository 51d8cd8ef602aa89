set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
echo "== same-box A/B of the driver's configs (prev = ab/libbpmsm_prev.so, the build named in the log's first line)"; echo "prev = ${AB_PREV:-unnamed}"
for round in 1 2; do for tag in prev new; do
  if [ $tag = prev ]; then export BPMSM_SO=$R/ab/libbpmsm_prev.so; else unset BPMSM_SO; fi
  python bench.py --steps 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['configs']
print('$tag headline %.3f ms | cfg1 %.2f/%.2f | cfg3 prove %.2f verify %.2f tables %.2f | cfg5 msm %.3f ipp %.2f/%.2f | h2d %.3f' % (d['ms_per_step'], c['cfg1']['create_ms'], c['cfg1']['verify_ms'], c['cfg3_e2e']['prove_ms'], c['cfg3_e2e']['verify_ms'], c['cfg3_e2e']['with_precomputed_generator_tables']['prove_ms'], c['cfg5']['msm_ms'], c['cfg5']['ipp_create_ms'], c['cfg5']['ipp_verify_ms'], d['with_scalar_h2d']['ms_per_step']))"
done; done 2>&1 | tee gpurun_out/${AB_LOG:-r04_ab_configs_same_box.log}
unset BPMSM_SO
echo "== ipp profiles"
bash scripts/prof_ipp.sh r04_ipp_2p16_plain 0 16 none 2>&1 | tail -14
bash scripts/prof_ipp.sh r04_ipp_2p16_tables 0 16 16 2>&1 | grep -E "curve=|tables:"
bash scripts/prof_ipp.sh r04_ipp_bn254_2p12 1 12 none 2>&1 | grep -E "curve=|k_small_msm"
bash scripts/prof_ipp.sh r04_ipp_n64 0 6 none 2>&1 | grep -E "curve=|k_small_msm"
cd $R
echo "== structured"
for lg in 16 18 20 22; do python scripts/time_structured.py $lg 2>&1 | grep "n=2"; done > gpurun_out/r04_structured_scalars.log; wc -l gpurun_out/r04_structured_scalars.log
echo "== sizes"
{ python scripts/time_msm.py 14,16,17,18,19,20,21,22 2>/dev/null | grep "n=2"; python scripts/time_msm.py 16,18,20 1 2>/dev/null | grep "n=2"; python scripts/time_ipp.py 0 6,8,10,12,13,14,16 2>/dev/null | grep curve=; TIME_IPP_TABLES=16 python scripts/time_ipp.py 0 12,13,14,16 2>/dev/null | grep -E "curve=|tables:" | sed "s/^/tables16 /"; python scripts/time_ipp.py 1 6,10,12,16 2>/dev/null | grep curve=; python scripts/time_verify.py 0 16 none; python scripts/time_verify.py 0 16 16; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_sizes_final_build.log; tail -12 gpurun_out/r04_sizes_final_build.log
echo "== fuzz"
python scripts/fuzz_msm.py 100 41 2>&1 | tail -1; python scripts/fuzz_ipp.py 100 42 2>&1 | tail -1; python scripts/fuzz_misc.py 80 43 2>&1 | tail -1
echo "== final bench line"
python bench.py --steps 20 > gpurun_out/r04_bench_n1_unprofiled.json 2>/dev/null; tail -c 300 gpurun_out/r04_bench_n1_unprofiled.json
