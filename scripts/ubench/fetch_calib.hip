// fetch_calib.hip -- what does rocprofv3's FETCH_SIZE report for the access SHAPES of this library's kernels?  (round 5)
// MI355X_MICROARCH.md: on gfx950 FETCH_SIZE is exactly 1/2 of the bytes of a wide coalesced streaming read (16 B per lane);
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  Every kernel below reads
// a 1 GiB buffer (4 x the Infinity Cache) exactly once; the factor of a shape = bytes read / (FETCH_SIZE x 1024).
//   calib_b16_coalesced   16 B per lane, a wave = 1 KB contiguous          (smm_class_sums_kernel's float4 rows)
//   calib_b16_rowpieces   16 B per lane, 4 lanes = 64 contiguous bytes of one 800-byte row, 16 rows per wave instruction
//                         (smm_emission_*'s A-operand loads at D = 200: lane (frame, k group))
//   calib_b8_coalesced    8 B per lane, a wave = 512 B contiguous          (smm_viterbi_kernel's elp rows and history reads)
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/fetch_calib.hip -o scripts/ubench/_bin/fetch_calib
// run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- scripts/ubench/_bin/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void calib_b16_coalesced(const float4 *p, size_t n, float *sink)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) sink[0] = acc;
}

// rows of 200 floats (800 B); a wave's instruction m of a 16-row tile: lane = (row fr = lane & 15, group kq = lane >> 4) reads
// floats [16 m + 4 kq, +4) of its row -- 13 instructions cover floats 0 .. 207 (the last one clamped as the kernel does)
__global__ void calib_b16_rowpieces(const float *p, size_t rows, float *sink)
{
    const int lane = threadIdx.x & 63, fr = lane & 15, kq = lane >> 4;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    float acc = 0.f;
    for (size_t t = wave; t * 16 + 15 < rows; t += nwaves) {
        const float *row = p + (t * 16 + fr) * 200;
#pragma unroll
        for (int m = 0; m < 13; ++m) {
            int db = 16 * m + 4 * kq;
            db = db + 3 < 200 ? db : 196;
            const float4 v = *reinterpret_cast<const float4 *>(row + db);
            acc += v.x + v.y + v.z + v.w;
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

__global__ void calib_b8_coalesced(const double *p, size_t n, double *sink)
{
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 123.456) sink[0] = acc;
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    void *buf = nullptr, *sink = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    const size_t rows = bytes / 800;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_b16_coalesced, dim3(2048), dim3(256), 0, 0, (const float4 *)buf, bytes / 16, (float *)sink);
        hipLaunchKernelGGL(calib_b16_rowpieces, dim3(2048), dim3(256), 0, 0, (const float *)buf, rows, (float *)sink);
        hipLaunchKernelGGL(calib_b8_coalesced, dim3(2048), dim3(256), 0, 0, (const double *)buf, bytes / 8, (double *)sink);
    }
    hipDeviceSynchronize();
    printf("calib_b16_coalesced bytes %zu\ncalib_b16_rowpieces bytes %zu\ncalib_b8_coalesced bytes %zu\n", bytes, (rows / 16) * 16 * 800, bytes);
    return 0;
}
