// Micro-benchmark (diagnostic, not shipped): the latency of a chain of dependent v_add_f64 -- smm_cum_anchor_kernel's adder wave
// takes ~7.4 ns per addition (round 5) where round 4's chain-wave benchmark counted 5.3 cycles per dependent fp64 instruction.
// One wave per workgroup adds 32 x iters registers up in one dependent chain; grids of 1 workgroup (an otherwise idle GPU, as in
// cfg1's step) and of 1024 (every CU busy); wall time by HIP events around ONE launch, cycles by s_memtime, ns by s_memrealtime.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(64) k(double *out, const double *in, int iters, unsigned long long *stamps)
{
    double v[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) v[u] = in[(threadIdx.x + u) & 63];
    double cum = in[threadIdx.x & 63];
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) cum = cum + v[u];
        asm volatile("" : "+v"(cum));
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = cum;
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = r1 - r0; }
}

int main()
{
    double *out, *in; unsigned long long *st;
    (void)hipMalloc(&out, 8 * 64 * 4096); (void)hipMalloc(&in, 8 * 64); (void)hipMalloc(&st, 16);
    double h[64]; for (int i = 0; i < 64; ++i) h[i] = -280.0 - i;
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 231;                                      // 7 392 additions: cfg1's prefix sums
    for (int grid : {1, 8, 256, 1024, 4096}) {
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            k<<<grid, 64>>>(out, in, iters, st);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned long long s[2]; (void)hipMemcpy(s, st, 16, hipMemcpyDeviceToHost);
            if (rep == 2)
                printf("grid %4d: launch %.1f us; workgroup 0: %llu cycles (s_memtime) = %.2f per addition, %.1f us (s_memrealtime, 100 MHz) = %.2f ns per addition -> %.2f GHz\n",
                       grid, ms * 1e3, s[0], (double)s[0] / (32.0 * iters), s[1] / 100.0, s[1] * 10.0 / (32.0 * iters), (double)s[0] / (s[1] * 10.0));
        }
    }
    return 0;
}
