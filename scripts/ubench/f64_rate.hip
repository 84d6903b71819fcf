// Micro-benchmark (diagnostic, not shipped): issue rate of fp64 add/max on gfx950 at 1..4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 f64_rate.hip -o f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

#define UNR 8
template <int MODE>
__global__ void __launch_bounds__(1024) k(double *out, unsigned long long *clk, int iters)
{
    double a[16], l[16];
    double h = out[threadIdx.x & 7];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x + i; l[i] = threadIdx.x * 0.5 + i; }
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(l[i])); }
                if (MODE == 1) { asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(l[i])); }
                if (MODE == 2) { double t; asm volatile("v_add_f64 %0, %1, %2" : "=v"(t) : "v"(h), "v"(l[i]));
                                 asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(t)); }
                if (MODE == 3) { asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(l[i])); }
                if (MODE == 4) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(((float *)&a[i])[0]) : "v"(((float *)&l[i])[0])); }
                if (MODE == 5) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(l[i])); }
            }
            if (MODE == 6) {   // groups of 4 adds then 4 maxes (what the pusher does)
#pragma unroll
                for (int i0 = 0; i0 < 16; i0 += 4) {
                    double t[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) asm volatile("v_add_f64 %0, %1, %2" : "=v"(t[q]) : "v"(h), "v"(l[i0 + q]));
#pragma unroll
                    for (int q = 0; q < 4; ++q) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i0 + q]) : "v"(t[q]));
                }
            }
            if (MODE == 7) {   // scalar h operand (SGPR pair) instead of VGPR
                unsigned long long hs = (unsigned)__builtin_amdgcn_readfirstlane(__double2loint(h)) |
                                        ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(__double2hiint(h)) << 32);
#pragma unroll
                for (int i0 = 0; i0 < 16; i0 += 4) {
                    double t[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) asm volatile("v_add_f64 %0, %1, %2" : "=v"(t[q]) : "s"(hs), "v"(l[i0 + q]));
#pragma unroll
                    for (int q = 0; q < 4; ++q) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i0 + q]) : "v"(t[q]));
                }
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x + 8] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { clk[2 * (threadIdx.x >> 6)] = t0; clk[2 * (threadIdx.x >> 6) + 1] = t1; }
}

template <int MODE>
void run(const char *name, int opsper, double *out, unsigned long long *clk)
{
    const int iters = 4000;
    for (int w = 1; w <= 4; ++w) {               // waves per SIMD
        const int wg = 256 * w;
        float best = 1e9; unsigned long long span = 0, own = 0;
        for (int rep = 0; rep < 4; ++rep) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            k<MODE><<<256, wg>>>(out, clk, iters);
            (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c[32]; (void)hipMemcpy(c, clk, 32 * 8, hipMemcpyDeviceToHost);
            unsigned long long lo = ~0ull, hi = 0;
            for (int q = 0; q < 4 * w; ++q) { lo = std::min(lo, c[2 * q]); hi = std::max(hi, c[2 * q + 1]); }
            if (ms < best) { best = ms; span = hi - lo; own = c[1] - c[0]; }
        }
        const double instr_per_simd = (double)iters * UNR * 16 * opsper * w;
        printf("%-16s waves/SIMD=%d  %.3f ms  WG-span cycles/instr/SIMD=%.2f   wave0 cycles/instr=%.2f  (event: %.3f ns/instr/SIMD)\n",
               name, w, best, span / instr_per_simd, own / ((double)iters * UNR * 16 * opsper), best * 1e6 / instr_per_simd);
    }
}

int main()
{
    double *out; unsigned long long *clk;
    (void)hipMalloc(&out, sizeof(double) * (256 * 1024 + 8)); (void)hipMemset(out, 0, sizeof(double) * (256 * 1024 + 8));
    (void)hipMalloc(&clk, 32 * 8);
    k<0><<<256, 1024>>>(out, clk, 20000); (void)hipDeviceSynchronize();   // clock ramp-up
    run<0>("v_add_f64", 1, out, clk);
    run<1>("v_max_f64", 1, out, clk);
    run<2>("add,max pairs", 2, out, clk);
    run<6>("4add,4max vgpr h", 2, out, clk);
    run<7>("4add,4max sgpr h", 2, out, clk);
    run<3>("v_fma_f64", 1, out, clk);
    run<4>("v_add_f32", 1, out, clk);
    run<5>("v_pk_add_f32", 1, out, clk);
    return 0;
}
