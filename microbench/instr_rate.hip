// Instruction-throughput microbenchmark for gfx950: which multiplier path should the
// 381-bit Montgomery arithmetic be built on?  Each kernel issues ITER x 8 independent
// instances of one instruction per lane; the host reports wave-instructions/clk/CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITER = 4096;


#define K64(NAME, I0,I1,I2,I3,I4,I5,I6,I7)                                          \
__global__ void NAME(uint64_t* out, uint32_t a, uint32_t b) {                         \
    uint64_t c0,c1,c2,c3,c4,c5,c6,c7;                                                   \
    c0=(uint64_t)threadIdx.x+0;c1=(uint64_t)threadIdx.x+1;c2=(uint64_t)threadIdx.x+2;c3=(uint64_t)threadIdx.x+3;c4=(uint64_t)threadIdx.x+4;c5=(uint64_t)threadIdx.x+5;c6=(uint64_t)threadIdx.x+6;c7=(uint64_t)threadIdx.x+7; \
    uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;                                                                            \
    for (int it = 0; it < ITER; it++) {                                               \
        asm volatile(I0 "\n\t" I1 "\n\t" I2 "\n\t" I3 "\n\t" I4 "\n\t" I5 "\n\t" I6 "\n\t" I7       \
            : "+v"(c0),"+v"(c1),"+v"(c2),"+v"(c3),"+v"(c4),"+v"(c5),"+v"(c6),"+v"(c7)  \
            : "v"(x), "v"(y) : "vcc", "s20", "s21", "s22", "s23");                      \
    }                                                                                 \
    out[blockIdx.x*blockDim.x+threadIdx.x] = c0^c1^c2^c3^c4^c5^c6^c7;                                                                              \
}

#define K32(NAME, I0,I1,I2,I3,I4,I5,I6,I7)                                          \
__global__ void NAME(uint64_t* out, uint32_t a, uint32_t b) {                         \
    uint32_t c0,c1,c2,c3,c4,c5,c6,c7;                                                   \
    c0=threadIdx.x+0;c1=threadIdx.x+1;c2=threadIdx.x+2;c3=threadIdx.x+3;c4=threadIdx.x+4;c5=threadIdx.x+5;c6=threadIdx.x+6;c7=threadIdx.x+7; \
    uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;                                                                            \
    for (int it = 0; it < ITER; it++) {                                               \
        asm volatile(I0 "\n\t" I1 "\n\t" I2 "\n\t" I3 "\n\t" I4 "\n\t" I5 "\n\t" I6 "\n\t" I7       \
            : "+v"(c0),"+v"(c1),"+v"(c2),"+v"(c3),"+v"(c4),"+v"(c5),"+v"(c6),"+v"(c7)  \
            : "v"(x), "v"(y) : "vcc", "s20", "s21", "s22", "s23");                      \
    }                                                                                 \
    out[blockIdx.x*blockDim.x+threadIdx.x] = c0^c1^c2^c3^c4^c5^c6^c7;                                                                              \
}

#define KF64(NAME, I0,I1,I2,I3,I4,I5,I6,I7)                                          \
__global__ void NAME(uint64_t* out, uint32_t a, uint32_t b) {                         \
    double c0,c1,c2,c3,c4,c5,c6,c7;                                                   \
    c0=(double)threadIdx.x+0;c1=(double)threadIdx.x+1;c2=(double)threadIdx.x+2;c3=(double)threadIdx.x+3;c4=(double)threadIdx.x+4;c5=(double)threadIdx.x+5;c6=(double)threadIdx.x+6;c7=(double)threadIdx.x+7; \
    double x = 1.0 + a*1e-9, y = 1.0 - b*1e-9;                                                                            \
    for (int it = 0; it < ITER; it++) {                                               \
        asm volatile(I0 "\n\t" I1 "\n\t" I2 "\n\t" I3 "\n\t" I4 "\n\t" I5 "\n\t" I6 "\n\t" I7       \
            : "+v"(c0),"+v"(c1),"+v"(c2),"+v"(c3),"+v"(c4),"+v"(c5),"+v"(c6),"+v"(c7)  \
            : "v"(x), "v"(y) : "vcc", "s20", "s21", "s22", "s23");                      \
    }                                                                                 \
    out[blockIdx.x*blockDim.x+threadIdx.x] = (uint64_t)(c0+c1+c2+c3+c4+c5+c6+c7);                                                                              \
}
K64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %8, %9, %0","v_mad_u64_u32 %1, vcc, %8, %9, %1","v_mad_u64_u32 %2, vcc, %8, %9, %2","v_mad_u64_u32 %3, vcc, %8, %9, %3","v_mad_u64_u32 %4, vcc, %8, %9, %4","v_mad_u64_u32 %5, vcc, %8, %9, %5","v_mad_u64_u32 %6, vcc, %8, %9, %6","v_mad_u64_u32 %7, vcc, %8, %9, %7")
K64(k_mad_u64_u32_sdst, "v_mad_u64_u32 %0, s[20:21], %8, %9, %0","v_mad_u64_u32 %1, s[20:21], %8, %9, %1","v_mad_u64_u32 %2, s[20:21], %8, %9, %2","v_mad_u64_u32 %3, s[20:21], %8, %9, %3","v_mad_u64_u32 %4, s[20:21], %8, %9, %4","v_mad_u64_u32 %5, s[20:21], %8, %9, %5","v_mad_u64_u32 %6, s[20:21], %8, %9, %6","v_mad_u64_u32 %7, s[20:21], %8, %9, %7")
K64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %0","v_lshl_add_u64 %1, %1, 0, %1","v_lshl_add_u64 %2, %2, 0, %2","v_lshl_add_u64 %3, %3, 0, %3","v_lshl_add_u64 %4, %4, 0, %4","v_lshl_add_u64 %5, %5, 0, %5","v_lshl_add_u64 %6, %6, 0, %6","v_lshl_add_u64 %7, %7, 0, %7")
K64(k_lshrrev_b64, "v_lshrrev_b64 %0, 1, %0","v_lshrrev_b64 %1, 1, %1","v_lshrrev_b64 %2, 1, %2","v_lshrrev_b64 %3, 1, %3","v_lshrrev_b64 %4, 1, %4","v_lshrrev_b64 %5, 1, %5","v_lshrrev_b64 %6, 1, %6","v_lshrrev_b64 %7, 1, %7")
K32(k_mul_lo_u32, "v_mul_lo_u32 %0, %8, %0","v_mul_lo_u32 %1, %8, %1","v_mul_lo_u32 %2, %8, %2","v_mul_lo_u32 %3, %8, %3","v_mul_lo_u32 %4, %8, %4","v_mul_lo_u32 %5, %8, %5","v_mul_lo_u32 %6, %8, %6","v_mul_lo_u32 %7, %8, %7")
K32(k_mul_hi_u32, "v_mul_hi_u32 %0, %8, %0","v_mul_hi_u32 %1, %8, %1","v_mul_hi_u32 %2, %8, %2","v_mul_hi_u32 %3, %8, %3","v_mul_hi_u32 %4, %8, %4","v_mul_hi_u32 %5, %8, %5","v_mul_hi_u32 %6, %8, %6","v_mul_hi_u32 %7, %8, %7")
K32(k_mad_u32_u24, "v_mad_u32_u24 %0, %8, %9, %0","v_mad_u32_u24 %1, %8, %9, %1","v_mad_u32_u24 %2, %8, %9, %2","v_mad_u32_u24 %3, %8, %9, %3","v_mad_u32_u24 %4, %8, %9, %4","v_mad_u32_u24 %5, %8, %9, %5","v_mad_u32_u24 %6, %8, %9, %6","v_mad_u32_u24 %7, %8, %9, %7")
K32(k_mul_u32_u24, "v_mul_u32_u24 %0, %8, %0","v_mul_u32_u24 %1, %8, %1","v_mul_u32_u24 %2, %8, %2","v_mul_u32_u24 %3, %8, %3","v_mul_u32_u24 %4, %8, %4","v_mul_u32_u24 %5, %8, %5","v_mul_u32_u24 %6, %8, %6","v_mul_u32_u24 %7, %8, %7")
K32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %8, %0","v_mul_hi_u32_u24 %1, %8, %1","v_mul_hi_u32_u24 %2, %8, %2","v_mul_hi_u32_u24 %3, %8, %3","v_mul_hi_u32_u24 %4, %8, %4","v_mul_hi_u32_u24 %5, %8, %5","v_mul_hi_u32_u24 %6, %8, %6","v_mul_hi_u32_u24 %7, %8, %7")
K32(k_add_u32, "v_add_u32 %0, %8, %0","v_add_u32 %1, %8, %1","v_add_u32 %2, %8, %2","v_add_u32 %3, %8, %3","v_add_u32 %4, %8, %4","v_add_u32 %5, %8, %5","v_add_u32 %6, %8, %6","v_add_u32 %7, %8, %7")
K32(k_add_co_u32, "v_add_co_u32 %0, vcc, %8, %0","v_add_co_u32 %1, vcc, %8, %1","v_add_co_u32 %2, vcc, %8, %2","v_add_co_u32 %3, vcc, %8, %3","v_add_co_u32 %4, vcc, %8, %4","v_add_co_u32 %5, vcc, %8, %5","v_add_co_u32 %6, vcc, %8, %6","v_add_co_u32 %7, vcc, %8, %7")
K32(k_add_co_u32_sdst, "v_add_co_u32 %0, s[20:21], %8, %0","v_add_co_u32 %1, s[20:21], %8, %1","v_add_co_u32 %2, s[20:21], %8, %2","v_add_co_u32 %3, s[20:21], %8, %3","v_add_co_u32 %4, s[20:21], %8, %4","v_add_co_u32 %5, s[20:21], %8, %5","v_add_co_u32 %6, s[20:21], %8, %6","v_add_co_u32 %7, s[20:21], %8, %7")
K32(k_addc_co_u32, "v_addc_co_u32 %0, vcc, %8, %0, vcc","v_addc_co_u32 %1, vcc, %8, %1, vcc","v_addc_co_u32 %2, vcc, %8, %2, vcc","v_addc_co_u32 %3, vcc, %8, %3, vcc","v_addc_co_u32 %4, vcc, %8, %4, vcc","v_addc_co_u32 %5, vcc, %8, %5, vcc","v_addc_co_u32 %6, vcc, %8, %6, vcc","v_addc_co_u32 %7, vcc, %8, %7, vcc")
K32(k_add3_u32, "v_add3_u32 %0, %8, %9, %0","v_add3_u32 %1, %8, %9, %1","v_add3_u32 %2, %8, %9, %2","v_add3_u32 %3, %8, %9, %3","v_add3_u32 %4, %8, %9, %4","v_add3_u32 %5, %8, %9, %5","v_add3_u32 %6, %8, %9, %6","v_add3_u32 %7, %8, %9, %7")
K32(k_xor, "v_xor_b32 %0, %8, %0","v_xor_b32 %1, %8, %1","v_xor_b32 %2, %8, %2","v_xor_b32 %3, %8, %3","v_xor_b32 %4, %8, %4","v_xor_b32 %5, %8, %5","v_xor_b32 %6, %8, %6","v_xor_b32 %7, %8, %7")
K32(k_alignbit, "v_alignbit_b32 %0, %8, %0, 7","v_alignbit_b32 %1, %8, %1, 7","v_alignbit_b32 %2, %8, %2, 7","v_alignbit_b32 %3, %8, %3, 7","v_alignbit_b32 %4, %8, %4, 7","v_alignbit_b32 %5, %8, %5, 7","v_alignbit_b32 %6, %8, %6, 7","v_alignbit_b32 %7, %8, %7, 7")
K32(k_cndmask, "v_cndmask_b32 %0, %8, %0, vcc","v_cndmask_b32 %1, %8, %1, vcc","v_cndmask_b32 %2, %8, %2, vcc","v_cndmask_b32 %3, %8, %3, vcc","v_cndmask_b32 %4, %8, %4, vcc","v_cndmask_b32 %5, %8, %5, vcc","v_cndmask_b32 %6, %8, %6, vcc","v_cndmask_b32 %7, %8, %7, vcc")
K32(k_fma_f32, "v_fma_f32 %0, %8, %9, %0","v_fma_f32 %1, %8, %9, %1","v_fma_f32 %2, %8, %9, %2","v_fma_f32 %3, %8, %9, %3","v_fma_f32 %4, %8, %9, %4","v_fma_f32 %5, %8, %9, %5","v_fma_f32 %6, %8, %9, %6","v_fma_f32 %7, %8, %9, %7")
KF64(k_fma_f64, "v_fma_f64 %0, %8, %9, %0","v_fma_f64 %1, %8, %9, %1","v_fma_f64 %2, %8, %9, %2","v_fma_f64 %3, %8, %9, %3","v_fma_f64 %4, %8, %9, %4","v_fma_f64 %5, %8, %9, %5","v_fma_f64 %6, %8, %9, %6","v_fma_f64 %7, %8, %9, %7")
KF64(k_add_f64, "v_add_f64 %0, %8, %0","v_add_f64 %1, %8, %1","v_add_f64 %2, %8, %2","v_add_f64 %3, %8, %3","v_add_f64 %4, %8, %4","v_add_f64 %5, %8, %5","v_add_f64 %6, %8, %6","v_add_f64 %7, %8, %7")

typedef void (*kern_t)(uint64_t*, uint32_t, uint32_t);

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s CUs=%d clock=%d kHz\n", prop.name, cus, prop.clockRate);
    struct { const char* name; kern_t k; } ks[] = {
        {"mad_u64_u32", k_mad_u64_u32},
        {"mad_u64_u32_sdst", k_mad_u64_u32_sdst},
        {"lshl_add_u64", k_lshl_add_u64},
        {"lshrrev_b64", k_lshrrev_b64},
        {"mul_lo_u32", k_mul_lo_u32},
        {"mul_hi_u32", k_mul_hi_u32},
        {"mad_u32_u24", k_mad_u32_u24},
        {"mul_u32_u24", k_mul_u32_u24},
        {"mul_hi_u32_u24", k_mul_hi_u32_u24},
        {"add_u32", k_add_u32},
        {"add_co_u32", k_add_co_u32},
        {"add_co_u32_sdst", k_add_co_u32_sdst},
        {"addc_co_u32", k_addc_co_u32},
        {"add3_u32", k_add3_u32},
        {"xor", k_xor},
        {"alignbit", k_alignbit},
        {"cndmask", k_cndmask},
        {"fma_f32", k_fma_f32},
        {"fma_f64", k_fma_f64},
        {"add_f64", k_add_f64},
    };
    const int threads = 256;
    uint64_t* out; CHECK(hipMalloc(&out, sizeof(uint64_t) * (size_t)cus * 8 * threads));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps : {1, 2, 4, 8}) {   // waves per SIMD
        int blocks = cus * wps;      // 256 threads = 4 waves = 1 per SIMD per block
        printf("--- %d wave(s)/SIMD (%d blocks x %d threads)\n", wps, blocks, threads);
        for (auto& kk : ks) {
            hipLaunchKernelGGL(kk.k, dim3(blocks), dim3(threads), 0, 0, out, 3u, 5u);
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(kk.k, dim3(blocks), dim3(threads), 0, 0, out, 3u, 5u);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            double winstr = (double)blocks * (threads / 64) * ITER * 8;  // wave-instructions
            double per_s = winstr / (best * 1e-3);
            // cycles per wave-instruction per SIMD at 2.4 GHz nominal
            double cyc = 2.4e9 / (per_s / (cus * 4.0));
            printf("%-18s %8.3f ms  %.3e wave-instr/s  ~%.2f cyc/wave-instr/SIMD (@2.4GHz)  lane-ops/s %.3e\n",
                   kk.name, best, per_s, cyc, per_s * 64);
        }
    }
    return 0;
}
