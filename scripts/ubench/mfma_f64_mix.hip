// Micro-benchmark (diagnostic, not shipped): the emission scorer's matrix instruction mix per macro-step at 21..24 states -- 8 x
// v_mfma_f64_16x16x4_f64 (64 cycles each) + 16 x v_mfma_f64_4x4x4_4b_f64 (16 each) = 768 cycles nominal -- in the kernel's interleaved
// order, grouped by kind, and the 16x16x4 / 4x4x4 parts alone; 4 waves per SIMD (1 workgroup of 1024 threads per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(1024) k(double *out, unsigned long long *clk, int iters)
{
    d4 a0 = (d4){0, 0, 0, 0}, a1 = a0;
    double g00 = 0, g01 = 0, g10 = 0, g11 = 0;
    double a = out[threadIdx.x & 7] + threadIdx.x, b = a * 0.5, c = a * 0.25;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {                       // the kernel's order: per j: 2 big, 4 small
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, b, a1, 0, 0, 0);
                g00 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c, g00, 0, 0, 0);
                g10 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, c, g10, 0, 0, 0);
                g01 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, g01, 0, 0, 0);
                g11 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, b, g11, 0, 0, 0);
            }
        } else if (MODE == 1) {                // grouped: 8 big, then 16 small
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, b, a1, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                g00 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c, g00, 0, 0, 0);
                g10 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, c, g10, 0, 0, 0);
                g01 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, g01, 0, 0, 0);
                g11 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, b, g11, 0, 0, 0);
            }
        } else if (MODE == 2) {                // the 8 big only
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, b, a1, 0, 0, 0);
            }
        } else {                               // the 16 small only
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                g00 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c, g00, 0, 0, 0);
                g10 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, c, g10, 0, 0, 0);
                g01 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, g01, 0, 0, 0);
                g11 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, b, g11, 0, 0, 0);
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[8 + blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a0[1] + a0[2] + a0[3] + a1[0] + a1[1] + a1[2] + a1[3] + g00 + g01 + g10 + g11;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { clk[2 * (threadIdx.x >> 6)] = t0; clk[2 * (threadIdx.x >> 6) + 1] = t1; }
}

template <int MODE>
void run(const char *what, double *out, unsigned long long *clk, int nominal)
{
    const int iters = 4000;
    k<MODE><<<256, 1024>>>(out, clk, iters);
    (void)hipDeviceSynchronize();
    unsigned long long c[32]; (void)hipMemcpy(c, clk, 32 * 8, hipMemcpyDeviceToHost);
    unsigned long long lo = ~0ull, hi = 0;
    for (int q = 0; q < 16; ++q) { lo = c[2 * q] < lo ? c[2 * q] : lo; hi = c[2 * q + 1] > hi ? c[2 * q + 1] : hi; }
    printf("%-34s %.0f cycles per macro-step and SIMD (4 waves; nominal %d)\n", what, (hi - lo) / ((double)iters * 4), nominal);
}

int main()
{
    double *out; unsigned long long *clk;
    (void)hipMalloc(&out, 8 * (256 * 1024 + 8)); (void)hipMemset(out, 0, 8 * (256 * 1024 + 8)); (void)hipMalloc(&clk, 32 * 8);
    k<2><<<256, 1024>>>(out, clk, 2000); (void)hipDeviceSynchronize();
    run<0>("interleaved (2 big, 4 small) x 4", out, clk, 768);
    run<1>("8 big, then 16 small", out, clk, 768);
    run<2>("8 big", out, clk, 512);
    run<3>("16 small", out, clk, 256);
    return 0;
}
