// Micro-benchmark (diagnostic, not shipped): issue cost of the cross-lane moves the DP kernels use, against v_add_f64:
// v_mov_b32_dpp wave_ror:1 (the length rings' rotation), row_ror:1, v_mov_b64_dpp row_newbcast (source broadcast),
// v_permlane32_swap, v_readlane_b32 -- independent instructions, 1, 2 and 3 waves per SIMD; and the pusher's actual
// step (2 add + 2 max + 2 wave_ror on dependent registers).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>

#define R8(x) x x x x x x x x
#define R64(x) R8(R8(x))
template <int MODE>
__global__ void __launch_bounds__(1024) k(unsigned long long *clk, double *out, int iters)
{
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, l0 = 0.5, l1 = 0.25, h = 1.0;
    int s0 = threadIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) asm volatile(R64("v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(h));
        if (MODE == 1) asm volatile(R64("v_mov_b32_dpp %0, %0 wave_ror:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %1 wave_ror:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %2 wave_ror:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %3 wave_ror:1 row_mask:0xf bank_mask:0xf\n\t") : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
        if (MODE == 2) asm volatile(R64("v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %3 row_ror:1 row_mask:0xf bank_mask:0xf\n\t") : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
        if (MODE == 3) asm volatile(R64("v_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %1, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %2, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %3, %3 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        if (MODE == 4) asm volatile(R64("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\t") : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
        // the pusher's step: two ring registers, one rotated per step (the compiler splits the 64-bit wave_ror into two b32 moves)
        if (MODE == 5) {
#pragma unroll
            for (int u = 0; u < 64; ++u) {
                double t0 = h + l0, t1 = h + l1;
                asm volatile("v_max_f64 %0, %0, %2\n\tv_max_f64 %1, %1, %3" : "+v"(a0), "+v"(a1) : "v"(t0), "v"(t1));
                if (u & 1) l1 = __builtin_amdgcn_update_dpp(l1, l1, 0x13C, 0xf, 0xf, false);
                else l0 = __builtin_amdgcn_update_dpp(l0, l0, 0x13C, 0xf, 0xf, false);
            }
        }
        if (MODE == 6) asm volatile(R64("v_max_f64 %0, %0, %4\n\tv_max_f64 %1, %1, %4\n\tv_max_f64 %2, %2, %4\n\tv_max_f64 %3, %3, %4\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(h));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { clk[2 * (threadIdx.x >> 6)] = t0; clk[2 * (threadIdx.x >> 6) + 1] = t1; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + l0 + l1 + s0 + s1 + s2 + s3;
}

template <int MODE>
void run(const char *name, int per_iter, unsigned long long *clk, double *out)
{
    const int iters = 2000;
    for (int w = 1; w <= 3; ++w) {
        k<MODE><<<256, 256 * w>>>(clk, out, iters);
        (void)hipDeviceSynchronize();
        unsigned long long c[32]; (void)hipMemcpy(c, clk, 32 * 8, hipMemcpyDeviceToHost);
        unsigned long long lo = ~0ull, hi = 0;
        for (int q = 0; q < 4 * w; ++q) { lo = std::min(lo, c[2 * q]); hi = std::max(hi, c[2 * q + 1]); }
        printf("%-44s waves/SIMD=%d  cycles/instr/SIMD=%.2f  wave0 cycles/instr=%.2f\n", name, w,
               (hi - lo) / ((double)iters * per_iter * w), (c[1] - c[0]) / ((double)iters * per_iter));
    }
}

int main()
{
    unsigned long long *clk; (void)hipMalloc(&clk, 32 * 8);
    double *out; (void)hipMalloc(&out, 256 * 1024 * 8);
    k<0><<<256, 1024>>>(clk, out, 5000); (void)hipDeviceSynchronize();
    run<0>("v_add_f64 (4 independent)", 256, clk, out);
    run<6>("v_max_f64 (4 independent)", 256, clk, out);
    run<1>("v_mov_b32_dpp wave_ror:1", 256, clk, out);
    run<2>("v_mov_b32_dpp row_ror:1", 256, clk, out);
    run<3>("v_mov_b64_dpp row_newbcast", 256, clk, out);
    run<4>("v_permlane32_swap_b32", 256, clk, out);
    run<5>("pusher step (2 add + 2 max + 2 wave_ror b32)", 64 * 6, clk, out);
    return 0;
}
