#!/bin/bash
# AddressSanitizer + UBSan over everything that runs on the CPU (GPU sanitizers are not available on this pool):
#   1. the C oracle (gcc) against every golden vector,
#   2. the HOST half of libbpmsm.so (transcript, host tail, argument checking; hipcc -Xarch_host) under the CPU tests.
# Builds go to $OUT (default /tmp/bp_san); nothing in the tree is touched.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${OUT:-/tmp/bp_san}
mkdir -p "$OUT"
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -march=x86-64-v3 -fPIC -std=gnu11 -shared \
    -o "$OUT/liboracle.so" "$ROOT/oracle/oracle.c" "$ROOT/oracle/orc_merlin.c" -lpthread
GCC_ASAN=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
# PYTHONMALLOC=malloc: ctypes buffers come from malloc, so overruns of CALLER buffers are seen too (checked with a deliberate one)
export PYTHONMALLOC=malloc ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
( cd "$ROOT" && LD_PRELOAD=$GCC_ASAN BP_ORACLE_SO="$OUT/liboracle.so" python -m pytest tests/test_oracle_golden.py -x -q )
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
for f in bp_capi bp_capi_ipp bp_capi_hash bp_capi_r1cs; do
    "$HIPCC" --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer \
        -c -o "$OUT/$f.o" "$ROOT/bulletproofs-amcl_amd/csrc/$f.hip" &
done
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -o "$OUT/libbpmsm.so" "$OUT"/bp_capi.o "$OUT"/bp_capi_ipp.o "$OUT"/bp_capi_hash.o "$OUT"/bp_capi_r1cs.o
CLANG_ASAN=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
( cd "$ROOT" && LD_PRELOAD=$CLANG_ASAN BPMSM_SO="$OUT/libbpmsm.so" python -m pytest tests/test_capi_cpu.py tests/test_host_cpu.py -x -q -m "not gpu" )
echo "sanitize_cpu: clean"
