// Micro-benchmark (diagnostic, not shipped): cycles per v_mfma_f64_16x16x4_f64 on gfx950, 1 / 2 / 4 accumulators, 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(1024) k(double *out, unsigned long long *clk, int iters)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = out[threadIdx.x & 7] + threadIdx.x, b = a * 0.5;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[8 + blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { clk[2 * (threadIdx.x >> 6)] = t0; clk[2 * (threadIdx.x >> 6) + 1] = t1; }
}

template <int NACC>
void run(double *out, unsigned long long *clk)
{
    const int iters = 2000;
    for (int w = 1; w <= 4; w *= 2) {
        k<NACC><<<256, 256 * w>>>(out, clk, iters);
        (void)hipDeviceSynchronize();
        unsigned long long c[32]; (void)hipMemcpy(c, clk, 32 * 8, hipMemcpyDeviceToHost);
        unsigned long long lo = ~0ull, hi = 0;
        for (int q = 0; q < 4 * w; ++q) { lo = std::min(lo, c[2 * q]); hi = std::max(hi, c[2 * q + 1]); }
        printf("accumulators=%d waves/SIMD=%d  cycles/MFMA/SIMD=%.2f  wave0 cycles/MFMA=%.2f\n", NACC, w,
               (hi - lo) / ((double)iters * 16 * w), (c[1] - c[0]) / ((double)iters * 16));
    }
}

int main()
{
    double *out; unsigned long long *clk;
    (void)hipMalloc(&out, 8 * (256 * 1024 + 8)); (void)hipMemset(out, 0, 8 * (256 * 1024 + 8)); (void)hipMalloc(&clk, 32 * 8);
    k<1><<<256, 1024>>>(out, clk, 5000); (void)hipDeviceSynchronize();
    run<1>(out, clk); run<2>(out, clk); run<4>(out, clk);
    return 0;
}
