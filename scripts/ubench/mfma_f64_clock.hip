// Micro-benchmark (diagnostic, not shipped): what clock does the whole GPU sustain under back-to-back fp64 MFMAs (the emission
// scorer's matrix side)?  256 workgroups x 4 waves per SIMD of v_mfma_f64_16x16x4_f64, two accumulators per wave; wall time by
// HIP events against the cycle counter.  Also: the same with 2 KB of HBM reads per 8 MFMAs and wave (the scorer's ratio at 16 states).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <bool LOADS>
__global__ void __launch_bounds__(1024) k(double *out, const float4 *x, size_t nx4, int iters)
{
    d4 acc0 = (d4){0, 0, 0, 0}, acc1 = acc0;
    double a = out[threadIdx.x & 7] + threadIdx.x, b = a * 0.5;
    size_t p = ((size_t)blockIdx.x * 1024 + threadIdx.x) * 2;
    float4 v0 = make_float4(0, 0, 0, 0), v1 = v0;
    for (int it = 0; it < iters; ++it) {
        float4 n0, n1;
        if (LOADS) { n0 = x[p % nx4]; n1 = x[(p + 1) % nx4]; p += (size_t)gridDim.x * 2048; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a + v0.x, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a + v1.y, b, acc1, 0, 0, 0);
        }
        if (LOADS) { v0 = n0; v1 = n1; }
    }
    out[8 + blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
}

int main()
{
    double *out; float4 *x;
    const size_t nx4 = (size_t)1 << 27;     // 2 GiB of float4
    (void)hipMalloc(&out, 8 * (256 * 1024 + 8)); (void)hipMemset(out, 0, 8 * (256 * 1024 + 8));
    (void)hipMalloc(&x, nx4 * 16); (void)hipMemset(x, 0, nx4 * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int iters : {500, 2000, 8000, 32000}) {
        for (int loads = 0; loads < 2; ++loads) {
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                if (loads) k<true><<<256, 1024>>>(out, x, nx4, iters); else k<false><<<256, 1024>>>(out, x, nx4, iters);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                // per SIMD: 4 waves x iters x 8 MFMAs x 64 cycles
                const double cyc = 4.0 * iters * 8 * 64;
                printf("loads=%d iters=%5d  %.3f ms  -> %.2f GHz if the matrix pipe never idles%s\n", loads, iters, ms, cyc / (ms * 1e6),
                       loads ? "" : "");
                if (loads) printf("          HBM read %.2f TB/s\n", 256.0 * 1024 * iters * 32 / (ms * 1e9));
            }
        }
    }
    return 0;
}
