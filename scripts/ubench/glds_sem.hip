// Micro-benchmark / semantics check (diagnostic, not shipped): global_load_lds_dwordx4 from inline asm, as the BAND pushers use
// it for the delayed sources: per-lane source addresses that are only 8-byte aligned, sc1, landing at M0 base + lane * 16,
// a counted s_waitcnt vmcnt(N) by the issuing wave and then its own ds_reads -- and how long issue -> landed takes.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/glds_sem.hip -o scripts/ubench/_bin/glds_sem
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ void __launch_bounds__(128) k(const double *src, double *out, unsigned long long *clk)
{
    __shared__ __attribute__((aligned(16))) double land[2][4][128];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2 * 4 * 128; i += 128) (&land[0][0][0])[i] = -1.0;
    __syncthreads();
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)&land[w][0][0]);
    const double *mine = src + w * 4096 + lane * 37 + 1;          // 8-byte aligned, not 16
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16(mine + 2 * q, base + q * 1024);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    const double a0 = land[w][0][2 * lane], a1 = land[w][1][2 * lane + 1];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) { s[2 * q] = land[w][q][2 * lane]; s[2 * q + 1] = land[w][q][2 * lane + 1]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) out[(w * 64 + lane) * 10 + i] = s[i];
    out[(w * 64 + lane) * 10 + 8] = a0; out[(w * 64 + lane) * 10 + 9] = a1;
    if (lane == 0) clk[w] = t1 - t0;
}

int main()
{
    const int n = 2 * 4096;
    std::vector<double> h(n);
    for (int i = 0; i < n; ++i) h[i] = i;
    double *src, *out; unsigned long long *clk;
    (void)hipMalloc(&src, n * 8); (void)hipMalloc(&out, 128 * 10 * 8); (void)hipMalloc(&clk, 16);
    (void)hipMemcpy(src, h.data(), n * 8, hipMemcpyHostToDevice);
    k<<<1, 128>>>(src, out, clk);
    (void)hipDeviceSynchronize();
    std::vector<double> o(128 * 10); unsigned long long c[2];
    (void)hipMemcpy(o.data(), out, o.size() * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 128; ++t) {
        const int w = t >> 6, lane = t & 63;
        for (int i = 0; i < 8; ++i) bad += o[t * 10 + i] != (double)(w * 4096 + lane * 37 + 1 + i);
        bad += o[t * 10 + 8] != (double)(w * 4096 + lane * 37 + 1) || o[t * 10 + 9] != (double)(w * 4096 + lane * 37 + 1 + 3);
    }
    printf("global_load_lds_dwordx4 (asm, sc1, 8-byte aligned per-lane sources): %s (%d mismatches); 4 instructions issue -> all landed: %llu / %llu cycles (cold)\n",
           bad ? "WRONG" : "as expected", bad, c[0], c[1]);
    printf("lane 1 of wave 0 got: %g %g %g %g %g %g %g %g (expected 38..45)\n", o[10], o[11], o[12], o[13], o[14], o[15], o[16], o[17]);
    return bad != 0;
}
