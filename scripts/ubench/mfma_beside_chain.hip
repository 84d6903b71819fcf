// Micro-benchmark (diagnostic, not shipped): does a wave that issues fp64 MFMAs slow a dependent fp64 VALU chain of the
// OTHER wave of its SIMD?  (The question behind scoring the features inside the DP workgroup: the only wave with room is the
// chain wave's SIMD partner.)  8 waves per workgroup, one workgroup per CU; wave 0: 256 dependent v_add/v_max_f64 per iteration
// at s_setprio 3; wave 4 (same SIMD): nothing / independent 16x16x4 + 4x4x4 fp64 MFMA chains / the same number of fp64 VALU FMAs;
// the other waves wait in the barrier.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_beside_chain.hip -o scripts/ubench/_bin/mfma_beside_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define R8(x) x x x x x x x x
#define R64(x) R8(R8(x))

template <int MODE>
__global__ void __launch_bounds__(512) k(unsigned long long *clk, double *out, int iters)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double a = 1.0 + lane, b = 0.5;
    d4 acc = {0, 0, 0, 0};
    double acc1 = 0.0, acc2 = 0.0, f0 = lane, f1 = 2.0 * lane, f2 = 3.0, f3 = 4.0;
    if (w == 0) __builtin_amdgcn_s_setprio(3);
    unsigned long long t0 = 0, t1 = 0, tw = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned long long s0 = __builtin_readcyclecounter();
        if (w == 0) {
            asm volatile(R64("v_add_f64 %0, %0, %1\n\t") R64("v_max_f64 %0, %0, %1\n\t") R64("v_add_f64 %0, %0, %1\n\t") R64("v_max_f64 %0, %0, %1\n\t") : "+v"(a) : "v"(b));
        } else if (w == 4) {
            if (MODE == 1) {
#pragma unroll
                for (int q = 0; q < 26; ++q) {                       // 26 x (16x16x4 + 2 x 4x4x4): half a 16-frame tile of 23 states
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f1, acc, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(f0, f2, acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f64_4x4x4f64(f0, f3, acc2, 0, 0, 0);
                }
            }
            if (MODE == 2) {
                asm volatile(R64("v_fma_f64 %0, %2, %3, %0\n\tv_fma_f64 %1, %2, %3, %1\n\t") R64("v_fma_f64 %0, %2, %3, %0\n\tv_fma_f64 %1, %2, %3, %1\n\t") : "+v"(f0), "+v"(f1) : "v"(f2), "v"(f3));
            }
        }
        const unsigned long long s1 = __builtin_readcyclecounter();
        if (w == 0) tw += s1 - s0;
        if (w == 4) t1 += s1 - s0;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        t0 += __builtin_readcyclecounter() - s0;
    }
    if (lane == 0 && blockIdx.x == 0 && (w == 0 || w == 4)) { clk[w] = (w == 0) ? tw : t1; clk[w + 1] = t0; }
    out[blockIdx.x * 512 + threadIdx.x] = a + acc[0] + acc[1] + acc[2] + acc[3] + acc1 + acc2 + f0 + f1;
}

template <int MODE>
void run(const char *name, unsigned long long *clk, double *out)
{
    const int iters = 5000;
    k<MODE><<<256, 512>>>(clk, out, 50);
    (void)hipDeviceSynchronize();
    k<MODE><<<256, 512>>>(clk, out, iters);
    (void)hipDeviceSynchronize();
    unsigned long long c[8]; (void)hipMemcpy(c, clk, 64, hipMemcpyDeviceToHost);
    printf("%-72s chain wave: %7.1f cycles per 256 dependent fp64 ops (%.2f each); its SIMD partner busy %7.1f cycles; iteration %7.1f\n",
           name, double(c[0]) / iters, double(c[0]) / iters / 256, double(c[4]) / iters, double(c[1]) / iters);
}

int main()
{
    unsigned long long *clk; double *out;
    (void)hipMalloc(&clk, 64); (void)hipMalloc(&out, 256 * 512 * 8);
    (void)hipMemset(clk, 0, 64);
    run<0>("partner idle", clk, out);
    run<1>("partner: 26 x (v_mfma_f64_16x16x4 + 2 v_mfma_f64_4x4x4), independent chains", clk, out);
    run<2>("partner: 256 v_fma_f64 (two independent chains)", clk, out);
    return 0;
}
