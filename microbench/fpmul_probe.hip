#include <hip/hip_runtime.h>
#include "../bulletproofs-amcl_amd/csrc/bp_field.cuh"
using namespace bp;
using F = Fe<Bls381Fp>;
__global__ void k_mul(F* io, int iters) {
    F x = io[blockIdx.x * blockDim.x + threadIdx.x];
    F y = x;
    for (int i = 0; i < iters; i++) { x = fe_mul(x, y); y = fe_mul(y, x); }
    io[blockIdx.x * blockDim.x + threadIdx.x] = fe_add(x, y);
}
__global__ void k_addsub(F* io, int iters) {
    F x = io[blockIdx.x * blockDim.x + threadIdx.x];
    F y = x;
    for (int i = 0; i < iters; i++) { x = fe_add(x, y); y = fe_sub(y, x); }
    io[blockIdx.x * blockDim.x + threadIdx.x] = fe_add(x, y);
}
