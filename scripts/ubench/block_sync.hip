// Micro-benchmark (diagnostic, not shipped): what the fixed parts of a hand-over block of the Viterbi kernels cost on one
// CU -- 8 waves of one workgroup (2 per SIMD), one workgroup per CU: the LDS-only block barrier, dependent scalar
// instructions, taken branches, dependent LDS round trips (8 B broadcast, 16 B per lane), s_memtime, v_readlane -> VALU.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/block_sync.hip -o scripts/ubench/_bin/block_sync
#include <hip/hip_runtime.h>
#include <cstdio>

#define R8(x) x x x x x x x x
#define R64(x) R8(R8(x))

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// MODE 0: barrier only   1: 64 dependent s_add_u32 + barrier (every wave)   2: 64 dependent v_mov (every wave) + barrier
//      3: wave 0 runs 256 dependent v_mov, the others nothing, + barrier   4: 64 taken s_branch + barrier
//      5: 16 dependent ds_read_b64 (broadcast address) + barrier   6: 16 dependent ds_read2_b64 (16 B per lane) + barrier
//      7: 16 s_memtime (each waited for) + barrier   8: 32 x (v_readlane -> v_add_f64 with the SGPR pair) + barrier
//      9: like 3, but wave 0 at s_setprio 3   10-12: fp64 chains (does the clock follow the instruction mix?)
template <int MODE>
__global__ void __launch_bounds__(512) k(unsigned long long *clk, double *out, int iters)
{
    __shared__ double sh[2048];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2048; i += 512) sh[i] = 0.0;
    __syncthreads();
    int s = iters, v = lane, idx = lane;
    double a = 1.0, b = 2.0;
    if (MODE == 9 && w == 0) __builtin_amdgcn_s_setprio(3);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) asm volatile(R64("s_add_u32 %0, %0, 1\n\t") : "+s"(s));
        if (MODE == 2) asm volatile(R64("v_mov_b32 %0, %0\n\t") : "+v"(v));
        if ((MODE == 3 || MODE == 9) && w == 0) asm volatile(R64("v_mov_b32 %0, %0\n\t") R64("v_mov_b32 %0, %0\n\t") R64("v_mov_b32 %0, %0\n\t") R64("v_mov_b32 %0, %0\n\t") : "+v"(v));
        if (MODE == 4) {
#pragma unroll
            for (int q = 0; q < 64; ++q) asm volatile("s_branch 1f\n\ts_nop 0\n1:\n\t" ::: "memory");
        }
        if (MODE == 5) {
#pragma unroll
            for (int q = 0; q < 16; ++q) { idx = (int)__double2loint(sh[idx & 63]) & 63; asm volatile("" : "+v"(idx)); }
        }
        if (MODE == 6) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const double2 x = *reinterpret_cast<const double2 *>(&sh[(2 * lane + 2 * idx) & 2047]);
                idx = ((int)__double2loint(x.x) + (int)__double2loint(x.y)) & 63;
                asm volatile("" : "+v"(idx));
            }
        }
        if (MODE == 7) {
#pragma unroll
            for (int q = 0; q < 16; ++q) { const unsigned long long t = __builtin_readcyclecounter(); s += (int)t; asm volatile("" : "+s"(s)); }
        }
        if (MODE == 8) {
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const int lo = __builtin_amdgcn_readlane(__double2loint(a), q), hi = __builtin_amdgcn_readlane(__double2hiint(a), q);
                b = b + __hiloint2double(hi, lo);
                asm volatile("" : "+v"(b));
            }
        }
        if (MODE == 10) asm volatile(R64("v_add_f64 %0, %0, %1\n\t") : "+v"(a) : "v"(b));
        if (MODE == 11 && w == 0) asm volatile(R64("v_add_f64 %0, %0, %1\n\t") R64("v_max_f64 %0, %0, %1\n\t") R64("v_add_f64 %0, %0, %1\n\t") R64("v_max_f64 %0, %0, %1\n\t") : "+v"(a) : "v"(b));
        if (MODE == 12) {
            if (w == 0) asm volatile(R64("v_add_f64 %0, %0, %1\n\t") R64("v_max_f64 %0, %0, %1\n\t") R64("v_add_f64 %0, %0, %1\n\t") R64("v_max_f64 %0, %0, %1\n\t") : "+v"(a) : "v"(b));
            else asm volatile(R64("v_add_f64 %0, %0, %1\n\t") R64("v_max_f64 %0, %0, %1\n\t") : "+v"(a) : "v"(b));
        }
        lds_barrier();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) { clk[2 * w] = t0; clk[2 * w + 1] = t1; }
    out[blockIdx.x * 512 + threadIdx.x] = a + b + s + v + idx;
}

template <int MODE>
void run(const char *name, double per, unsigned long long *clk, double *out)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<256, 512>>>(clk, out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE><<<256, 512>>>(clk, out, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[16]; (void)hipMemcpy(c, clk, 16 * 8, hipMemcpyDeviceToHost);
    const double cyc = double(c[1] - c[0]) / iters;
    printf("%-64s %8.1f cycles/iter  %8.1f ns/iter  (%.2f GHz)  per unit: %.1f cycles\n", name, cyc, ms * 1e6 / iters, cyc / (ms * 1e6 / iters), per > 0 ? cyc / per : 0.0);
}

int main()
{
    unsigned long long *clk; double *out;
    (void)hipMalloc(&clk, 16 * 8); (void)hipMalloc(&out, 256 * 512 * 8);
    run<0>("barrier only (8 waves)", 1, clk, out);
    run<1>("64 dependent s_add_u32 + barrier", 64, clk, out);
    run<2>("64 dependent v_mov_b32 (every wave) + barrier", 64, clk, out);
    run<3>("wave 0: 256 dependent v_mov_b32; the others wait in the barrier", 256, clk, out);
    run<9>("... wave 0 at s_setprio 3", 256, clk, out);
    run<4>("64 taken s_branch + barrier", 64, clk, out);
    run<5>("16 dependent ds_read_b64 (broadcast) + barrier", 16, clk, out);
    run<6>("16 dependent ds_read_b128 (16 B per lane) + barrier", 16, clk, out);
    run<7>("16 s_memtime, each waited for, + barrier", 16, clk, out);
    run<8>("32 x (2 v_readlane -> v_add_f64) + barrier", 32, clk, out);
    run<10>("64 dependent v_add_f64 (every wave) + barrier", 64, clk, out);
    run<11>("wave 0: 256 dependent v_add/max_f64; the others wait", 256, clk, out);
    run<12>("wave 0: 256 dependent f64, the others 128 each, + barrier", 256, clk, out);
    return 0;
}
