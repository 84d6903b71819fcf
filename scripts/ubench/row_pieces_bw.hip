// Micro-benchmark (diagnostic, not shipped): HBM read bandwidth of a wave-level 16-byte-per-lane load whose 64 pieces are laid out as
// R rows x P bytes of a [frames][D = 200] fp32 matrix (row stride 800 B) -- the emission scorer's A-operand loads are 16 rows x 64 B --
// against fully contiguous 1 KB instructions, by the number of loads a wave keeps in flight.  16 waves per CU, 256 CUs, 2 GiB read once.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// PIECE: bytes per row piece (16 * lanes per row); 0: contiguous.  DEPTH loads in flight per wave.
template <int PIECE, int DEPTH, bool PRIVATE = false>
__global__ void __launch_bounds__(1024) k(const char *__restrict__ x, size_t bytes, unsigned *out)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 16 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 16;
    constexpr int D4 = 800;
    // a wave's unit: contiguous: DEPTH KB; pieces: a block of RB rows x (800 B) walked in DEPTH-instruction groups
    unsigned acc = 0;
    if constexpr (PIECE == 0) {
        const size_t unit = (size_t)DEPTH * 1024;
        for (size_t u = wave * unit; u + unit <= bytes; u += nwaves * unit) {
            u4 v[DEPTH];
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) v[i] = *reinterpret_cast<const u4 *>(x + u + i * 1024 + lane * 16);
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
        }
    } else {
        constexpr int LPR = PIECE / 16;
        constexpr int ROWS = 64 / LPR;                          // lanes per row, rows per instruction
        constexpr int CPR = D4 / PIECE;                         // whole pieces per row (the rest of the row is skipped: 800 = 12.5 x 64)
        const int r = lane / LPR < ROWS ? lane / LPR : ROWS - 1, c = lane % LPR;
        const size_t blk = (size_t)ROWS * D4;                   // bytes of a block of rows
        // PRIVATE: every wave walks its own contiguous share of the matrix (the emission scorer's persistent workgroups) instead of
        // the grid-stride sweep in which all waves stay inside one moving window
        const size_t per = bytes / blk / nwaves;
        const size_t u_first = PRIVATE ? wave * per * blk : wave * blk, u_step = PRIVATE ? blk : nwaves * blk;
        const size_t u_end = PRIVATE ? (wave + 1) * per * blk : bytes - blk + 1;
        for (size_t u = u_first; u < u_end; u += u_step) {
            for (int p0 = 0; p0 < CPR; p0 += DEPTH) {
                u4 v[DEPTH];
#pragma unroll
                for (int i = 0; i < DEPTH; ++i) {
                    const int p = p0 + i < CPR ? p0 + i : CPR - 1;
                    v[i] = *reinterpret_cast<const u4 *>(x + u + (size_t)r * D4 + p * PIECE + c * 16);
                }
#pragma unroll
                for (int i = 0; i < DEPTH; ++i) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int PIECE, int DEPTH, bool PRIVATE = false>
void run(const char *x, size_t bytes, unsigned *out)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        k<PIECE, DEPTH, PRIVATE><<<256, 1024>>>(x, bytes, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    // useful bytes: contiguous all; pieces: CPR * PIECE of every 800
    const double useful = PIECE == 0 ? (double)bytes : (double)bytes / 800.0 * ((800 / PIECE) * PIECE);
    printf("piece %4d B  depth %d  %s  %.3f ms  %.2f TB/s useful\n", PIECE, DEPTH, PRIVATE ? "private ranges" : "grid-stride   ", best, useful / (best * 1e9));
}

int main()
{
    const size_t bytes = (size_t)2 << 30;
    char *x; unsigned *out;
    (void)hipMalloc(&x, bytes + 4096); (void)hipMemset(x, 1, bytes + 4096); (void)hipMalloc(&out, 64);
    run<0, 2>(x, bytes, out); run<0, 4>(x, bytes, out); run<0, 8>(x, bytes, out);
    run<64, 2>(x, bytes, out); run<64, 4>(x, bytes, out); run<64, 6>(x, bytes, out); run<64, 12>(x, bytes, out);
    run<128, 2>(x, bytes, out); run<128, 3>(x, bytes, out); run<128, 6>(x, bytes, out);
    run<256, 3>(x, bytes, out);
    run<400, 2>(x, bytes, out);
    run<64, 2, true>(x, bytes, out); run<64, 4, true>(x, bytes, out); run<64, 6, true>(x, bytes, out); run<64, 12, true>(x, bytes, out);
    return 0;
}
