// Micro-benchmark (diagnostic, not shipped): cycles per v_mfma_f64_4x4x4_4b_f64 on gfx950 (4 blocks of 4x4x4: 512 FLOP)
// against v_mfma_f64_16x16x4_f64 (2048 FLOP, 64 cycles): would a 4-state granularity of the emission tiles be cheaper
// than padding 17..23 states to 32?  Also prints the operand / result lane layout found by one-hot probing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>

template <int NACC>
__global__ void __launch_bounds__(1024) k(double *out, unsigned long long *clk, int iters)
{
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    double a = out[threadIdx.x & 7] + threadIdx.x, b = a * 0.5;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[8 + blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { clk[2 * (threadIdx.x >> 6)] = t0; clk[2 * (threadIdx.x >> 6) + 1] = t1; }
}

__global__ void probe(double *res)   // res[la][lb][lane]
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            res[(la * 64 + lb) * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        }
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
        printf("4x4x4_4b: accumulators=%d waves/SIMD=%d  cycles/MFMA/SIMD=%.2f  wave0 cycles/MFMA=%.2f\n", NACC, w,
               (hi - lo) / ((double)iters * 16 * w), (c[1] - c[0]) / ((double)iters * 16));
    }
}

int main()
{
    double *out; unsigned long long *clk;
    (void)hipMalloc(&out, 8 * (256 * 1024 + 8)); (void)hipMemset(out, 0, 8 * (256 * 1024 + 8)); (void)hipMalloc(&clk, 32 * 8);
    k<1><<<256, 1024>>>(out, clk, 5000); (void)hipDeviceSynchronize();
    run<1>(out, clk); run<4>(out, clk); run<8>(out, clk);
    double *res; (void)hipMalloc(&res, 8 * 64 * 64 * 64);
    probe<<<1, 64>>>(res); (void)hipDeviceSynchronize();
    static double h[64 * 64 * 64]; (void)hipMemcpy(h, res, sizeof(h), hipMemcpyDeviceToHost);
    // for A lane la: which B lanes pair with it, and where does the product land?
    for (int la = 0; la < 64; la += 1) {
        if (la % 16 >= 6 && la % 16 < 14) continue;
        printf("A lane %2d pairs with (B lane -> D lane):", la);
        for (int lb = 0; lb < 64; ++lb)
            for (int l = 0; l < 64; ++l)
                if (h[(la * 64 + lb) * 64 + l] != 0.0) printf(" %d->%d", lb, l);
        printf("\n");
    }
    return 0;
}
