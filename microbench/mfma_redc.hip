// Can the matrix pipe take the constant half of a Montgomery product?  ("Why the idle matrix pipe cannot take
// the reduction"; VERDICT r2 weak #2.)  q = m * p with p the BLS12-381 base-field modulus -- the only part of a field product with a
// shared operand -- computed for 64 values per wave three ways, in a dependent loop (m <- lo(q) + hi(q)), whole chip:
//   valu   13 x 13 limbs of 30 bits, 169 v_mad_u64_u32 in column chains (what the library's fused core spends on this half)
//   mfma   m as 56 digits of 7 bits (int8 MFMA is signed) x the 56 x 112 Toeplitz matrix of p's digits: 4 row blocks x 7 column
//          blocks of V_MFMA_I32_16X16X64_I8, WITH the glue a per-lane bignum needs: digit split, transposition of the A operand
//          through LDS (one value per lane -> one value per 4 lanes), the 112 column sums back through LDS, carry recombination.
//          The value is held in 14 limbs of 28 bits here (4 digits per limb: the cheapest possible split; the library's radix is 30).
//   mfma0  the 28 MFMAs alone on register operands: the matrix pipe's own time for the job.
// Every variant's first step is checked against the host's 128-bit schoolbook product.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma_redc mfma_redc.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define KEEP(acc) asm volatile("" ::"v"(acc))

static const uint8_t kPbe[48] = {0x1a, 0x01, 0x11, 0xea, 0x39, 0x7f, 0xe6, 0x9a, 0x4b, 0x1b, 0xa7, 0xb6, 0x43, 0x4b, 0xac, 0xd7,
                                 0x64, 0x77, 0x4b, 0x84, 0xf3, 0x85, 0x12, 0xbf, 0x67, 0x30, 0xd2, 0xa0, 0xf6, 0xb0, 0xf6, 0x24,
                                 0x1e, 0xab, 0xff, 0xfe, 0xb1, 0x53, 0xff, 0xff, 0xb9, 0xfe, 0xff, 0xff, 0xff, 0xff, 0xaa, 0xab};

// ---- host bignum helpers: little-endian bit strings -------------------------------------------------------------------------
static uint32_t bits_at(const uint8_t* le, int nbytes, int pos, int width) {
    uint64_t v = 0;
    for (int b = 0; b < 8; b++) { int i = pos / 8 + b; if (i < nbytes) v |= (uint64_t)le[i] << (8 * b); }
    return (uint32_t)((v >> (pos % 8)) & ((1ull << width) - 1));
}
static void put_bits(uint8_t* le, int nbytes, int pos, uint64_t v) {          // le += v << pos (with carry)
    int i = pos / 8;
    unsigned __int128 c = (unsigned __int128)v << (pos % 8);
    while (c && i < nbytes) { c += le[i]; le[i] = (uint8_t)c; c >>= 8; i++; }
}

constexpr int LB30 = 30, N30 = 13, LB28 = 28, N28 = 14, ND = 56, NCOL = 112;
constexpr uint32_t M30 = (1u << 30) - 1, M28 = (1u << 28) - 1;

struct Consts {
    uint32_t p30[N30];
    uint32_t bfrag[7][64][4];          // B operand of column block cb, lane l: 16 bytes B[k = 16 (l >> 4) + j][col = 16 cb + (l & 15)]
};
__constant__ uint32_t c_p30[N30];

// ---- valu: 169 mads in column chains; columns of more than 8 terms run as two chains (13 x 2^60 does not fit 64 bits) ------------
__device__ __forceinline__ void mp_valu(const uint32_t (&m)[N30], uint32_t (&q)[2 * N30]) {
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 2 * N30 - 1; k++) {
        const int lo = k < N30 ? 0 : k - N30 + 1, hi = k < N30 ? k : N30 - 1, nt = hi - lo + 1;
        uint64_t a = carry, b = 0;
#pragma unroll
        for (int i = lo; i <= hi; i++) {
            if (nt > 8 && i - lo >= 8) { b += (uint64_t)m[i] * c_p30[k - i]; KEEP(b); }
            else { a += (uint64_t)m[i] * c_p30[k - i]; KEEP(a); }
        }
        if (nt > 8) {
            uint64_t s = (a & M30) + (b & M30);
            q[k] = (uint32_t)s & M30;
            carry = (a >> LB30) + (b >> LB30) + (s >> LB30);
        } else {
            q[k] = (uint32_t)a & M30;
            carry = a >> LB30;
        }
    }
    q[2 * N30 - 1] = (uint32_t)carry;
}

__global__ void __launch_bounds__(64) k_valu(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int iters, int dump) {
    const int gid = blockIdx.x * 64 + threadIdx.x;
    uint32_t m[N30], q[2 * N30];
#pragma unroll
    for (int i = 0; i < N30; i++) m[i] = in[(size_t)gid * N30 + i];
    for (int it = 0; it < iters; it++) {
        mp_valu(m, q);
        if (dump) break;
#pragma unroll
        for (int i = 0; i < N30; i++) m[i] = (q[i] + q[N30 + i]) & M30;
    }
    if (dump) {
#pragma unroll
        for (int i = 0; i < 2 * N30; i++) out[(size_t)gid * 2 * N30 + i] = q[i];
    } else {
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < N30; i++) x ^= m[i];
        out[gid] = x;
    }
}

// ---- mfma: the same product through the matrix pipe, glue included -------------------------------------------------------------
constexpr int APITCH = 20;      // dwords per value in the A staging area (16 used: 64 digit bytes; 80 B keeps b128 accesses aligned)
constexpr int CPITCH = 116;     // dwords per value in the column-sum area (112 used; 4 * 116 = 16 mod 64 spreads the four row groups)

__global__ void __launch_bounds__(64) k_mfma(const uint32_t* __restrict__ in, const uint32_t* __restrict__ bfrag_g, uint32_t* __restrict__ out,
                                             int iters, int dump) {
    __shared__ __attribute__((aligned(16))) uint32_t lds_a[64 * APITCH];
    __shared__ __attribute__((aligned(16))) uint32_t lds_c[64 * CPITCH];
    const int l = threadIdx.x, gid = blockIdx.x * 64 + l;
    uint32_t m[N28], q[2 * N28];
#pragma unroll
    for (int i = 0; i < N28; i++) m[i] = in[(size_t)gid * N28 + i];
    v4i bf[7];
#pragma unroll
    for (int cb = 0; cb < 7; cb++) bf[cb] = *(const v4i*)&bfrag_g[(cb * 64 + l) * 4];
    lds_a[l * APITCH + 14] = 0;         // digits 56..63 are zero, once
    lds_a[l * APITCH + 15] = 0;
    v4i zero = {0, 0, 0, 0};
    asm volatile("" : "+v"(zero));      // one zero accumulator kept in registers (else every MFMA gets four v_mov of its own)
    for (int it = 0; it < iters; it++) {
        // (1) split: four 7-bit digits of a 28-bit limb -> the four bytes of a dword
#pragma unroll
        for (int i = 0; i < N28; i++) {
            const uint32_t x = m[i];
            lds_a[l * APITCH + i] = (x & 0x7f) | ((x & 0x3f80) << 1) | ((x & 0x1fc000) << 2) | ((x & 0xfe00000) << 3);
        }
        __syncthreads();
        // (2) the A operand of row block rb: lane l holds digits 16 (l >> 4) .. + 15 of value 16 rb + (l & 15)
        v4i a[4];
#pragma unroll
        for (int rb = 0; rb < 4; rb++) a[rb] = *(const v4i*)&lds_a[(16 * rb + (l & 15)) * APITCH + 4 * (l >> 4)];
        // (3) 28 MFMAs; (4) the column sums go back to one value per lane: C[row = 4 (l >> 4) + i][col = l & 15]
        v4i c[4][7];
#pragma unroll
        for (int cb = 0; cb < 7; cb++)
#pragma unroll
            for (int rb = 0; rb < 4; rb++) c[rb][cb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[rb], bf[cb], zero, 0, 0, 0);   // 28 independent results
#pragma unroll
        for (int rb = 0; rb < 4; rb++)
#pragma unroll
            for (int cb = 0; cb < 7; cb++)
#pragma unroll
                for (int i = 0; i < 4; i++) lds_c[(16 * rb + 4 * (l >> 4) + i) * CPITCH + 16 * cb + (l & 15)] = (uint32_t)c[rb][cb][i];
        __syncthreads();
        // (5) recombine: q = sum S_j 2^(7 j) in limbs of 28 bits
        // (all in 32 bits: S < 2^20, so S0 + (S1 << 7) + (lo14(S2) << 14) + (lo7(S3) << 21) + carry < 2^30; the high parts ride in the carry)
        uint32_t carry = 0;
#pragma unroll
        for (int i = 0; i < 2 * N28; i++) {
            const v4i s = *(const v4i*)&lds_c[l * CPITCH + 4 * i];
            const uint32_t s0 = (uint32_t)s[0], s1 = (uint32_t)s[1], s2 = (uint32_t)s[2], s3 = (uint32_t)s[3];
            const uint32_t t = carry + s0 + (s1 << 7) + ((s2 & 0x3fff) << 14) + ((s3 & 0x7f) << 21);
            q[i] = t & M28;
            carry = (t >> LB28) + (s2 >> 14) + (s3 >> 7);
        }
        __syncthreads();
        if (dump) break;
#pragma unroll
        for (int i = 0; i < N28; i++) m[i] = (q[i] + q[N28 + i]) & M28;
    }
    if (dump) {
#pragma unroll
        for (int i = 0; i < 2 * N28; i++) out[(size_t)gid * 2 * N28 + i] = q[i];
    } else {
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < N28; i++) x ^= m[i];
        out[gid] = x;
    }
}

// ---- mfma0: the 28 MFMAs alone (operands in registers, results folded back into the next A so the loop stays dependent) ---------
__global__ void __launch_bounds__(64) k_mfma0(const uint32_t* __restrict__ in, const uint32_t* __restrict__ bfrag_g, uint32_t* __restrict__ out, int iters) {
    const int l = threadIdx.x, gid = blockIdx.x * 64 + l;
    v4i a[4], bf[7];
#pragma unroll
    for (int rb = 0; rb < 4; rb++) a[rb] = *(const v4i*)&in[((size_t)gid * 4 + rb) * 4 % (64 * 14)];
#pragma unroll
    for (int cb = 0; cb < 7; cb++) bf[cb] = *(const v4i*)&bfrag_g[(cb * 64 + l) * 4];
    for (int it = 0; it < iters; it++) {
        v4i c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
        for (int cb = 0; cb < 7; cb++)
#pragma unroll
            for (int rb = 0; rb < 4; rb++) c[rb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[rb], bf[cb], c[rb], 0, 0, 0);   // four independent chains
#pragma unroll
        for (int rb = 0; rb < 4; rb++) a[rb] = (a[rb] ^ c[rb]) & 0x7f7f7f7f;
    }
    out[gid] = (uint32_t)(a[0][0] ^ a[1][1] ^ a[2][2] ^ a[3][3]);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, %d MHz\n", prop.name, cus, prop.clockRate / 1000);

    uint8_t p_le[48];
    for (int i = 0; i < 48; i++) p_le[i] = kPbe[47 - i];
    uint32_t p30[N30];
    for (int i = 0; i < N30; i++) p30[i] = bits_at(p_le, 48, 30 * i, 30);
    CK(hipMemcpyToSymbol(HIP_SYMBOL(c_p30), p30, sizeof(p30)));
    // Toeplitz B: B[k][j] = digit_{j - k}(p), 7-bit digits, k < 64, j < 112
    uint32_t pd[ND];
    for (int i = 0; i < ND; i++) pd[i] = bits_at(p_le, 48, 7 * i, 7);
    std::vector<uint32_t> bfrag(7 * 64 * 4, 0);
    for (int cb = 0; cb < 7; cb++)
        for (int l = 0; l < 64; l++)
            for (int j = 0; j < 16; j++) {
                const int k = 16 * (l >> 4) + j, col = 16 * cb + (l & 15), d = col - k;
                const uint32_t v = (k < ND && d >= 0 && d < ND) ? pd[d] : 0;
                bfrag[(cb * 64 + l) * 4 + j / 4] |= v << (8 * (j % 4));
            }

    const int max_blocks = cus * 8;
    const size_t nval = (size_t)max_blocks * 64;
    // inputs: 392-bit values below 2^381 (same integers in both radices)
    std::vector<uint8_t> vals(nval * 49, 0);
    uint64_t s = 0x9e3779b97f4a7c15ull;
    for (size_t v = 0; v < nval; v++)
        for (int b = 0; b < 48; b++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            vals[v * 49 + b] = (uint8_t)(s >> 24) & (b == 47 ? 0x1f : 0xff);
        }
    std::vector<uint32_t> in30(nval * N30), in28(nval * N28);
    for (size_t v = 0; v < nval; v++) {
        for (int i = 0; i < N30; i++) in30[v * N30 + i] = bits_at(&vals[v * 49], 49, 30 * i, 30);
        for (int i = 0; i < N28; i++) in28[v * N28 + i] = bits_at(&vals[v * 49], 49, 28 * i, 28);
    }
    uint32_t *d_in30, *d_in28, *d_b, *d_out;
    CK(hipMalloc(&d_in30, in30.size() * 4));
    CK(hipMalloc(&d_in28, in28.size() * 4));
    CK(hipMalloc(&d_b, bfrag.size() * 4));
    CK(hipMalloc(&d_out, nval * 2 * N28 * 4));
    CK(hipMemcpy(d_in30, in30.data(), in30.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_in28, in28.data(), in28.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_b, bfrag.data(), bfrag.size() * 4, hipMemcpyHostToDevice));

    // ---- check step one of both forms against the host product ------------------------------------------------------------------
    {
        const int nb = 4;
        std::vector<uint32_t> o30((size_t)nb * 64 * 2 * N30), o28((size_t)nb * 64 * 2 * N28);
        k_valu<<<nb, 64>>>(d_in30, d_out, 1, 1);
        CK(hipMemcpy(o30.data(), d_out, o30.size() * 4, hipMemcpyDeviceToHost));
        k_mfma<<<nb, 64>>>(d_in28, d_b, d_out, 1, 1);
        CK(hipMemcpy(o28.data(), d_out, o28.size() * 4, hipMemcpyDeviceToHost));
        int bad30 = 0, bad28 = 0;
        for (int v = 0; v < nb * 64; v++) {
            uint8_t want[100] = {0}, got30[100] = {0}, got28[100] = {0};
            for (int i = 0; i < 48; i++)
                for (int j = 0; j < 48; j++) put_bits(want, 100, 8 * (i + j), (uint64_t)vals[(size_t)v * 49 + i] * p_le[j]);
            for (int i = 0; i < 2 * N30; i++) put_bits(got30, 100, 30 * i, o30[(size_t)v * 2 * N30 + i]);
            for (int i = 0; i < 2 * N28; i++) put_bits(got28, 100, 28 * i, o28[(size_t)v * 2 * N28 + i]);
            bad30 += memcmp(want, got30, 100) != 0;
            bad28 += memcmp(want, got28, 100) != 0;
        }
        printf("check m*p against the host product over %d values: valu %s (%d bad), mfma %s (%d bad)\n", nb * 64, bad30 ? "FAIL" : "ok", bad30,
               bad28 ? "FAIL" : "ok", bad28);
        if (bad30 || bad28) return 2;
    }

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto report = [&](const char* name, int wps, float ms, int blocks) {
        const double prods = (double)blocks * 64 * iters;
        printf("%-6s %d wave(s)/SIMD: %8.3f ms for %d steps -> %7.1f G products/s, %6.0f clocks per wave-step at %d MHz\n", name, wps, ms, iters,
               prods / ms / 1e6, ms * 1e-3 * (prop.clockRate * 1e3) / iters / wps, prop.clockRate / 1000);
    };
    for (int wps = 1; wps <= 2; wps++) {
        const int blocks = cus * 4 * wps;
        float ms;
        k_valu<<<blocks, 64>>>(d_in30, d_out, 10, 0);
        CK(hipEventRecord(e0));
        k_valu<<<blocks, 64>>>(d_in30, d_out, iters, 0);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        report("valu", wps, ms, blocks);
    }
    {
        const int blocks = cus * 4;       // 34 KB of LDS per wave: one wave per SIMD is what fits
        float ms;
        k_mfma<<<blocks, 64>>>(d_in28, d_b, d_out, 10, 0);
        CK(hipEventRecord(e0));
        k_mfma<<<blocks, 64>>>(d_in28, d_b, d_out, iters, 0);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        report("mfma", 1, ms, blocks);
    }
    for (int wps = 1; wps <= 2; wps++) {
        const int blocks = cus * 4 * wps;
        float ms;
        k_mfma0<<<blocks, 64>>>(d_in28, d_b, d_out, 10);
        CK(hipEventRecord(e0));
        k_mfma0<<<blocks, 64>>>(d_in28, d_b, d_out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        report("mfma0", wps, ms, blocks);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
