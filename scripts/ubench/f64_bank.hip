// Micro-benchmark (diagnostic, not shipped): do VGPR bank conflicts slow v_add_f64 / v_max_f64 on gfx950?
// Same instruction mix with operands chosen so that src0/src1 pairs share banks (bank = reg & 3) or not.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>

// 16 ops: dst = v[2i:2i+1] (i = 0..15), src0 = dst, src1 = v[OFF+2i : OFF+2i+1]
#define OP(i, off) "v_max_f64 v[" #i ":" #i "+1], v[" #i ":" #i "+1], v[" #off "+" #i ":" #off "+" #i "+1]\n\t"
#define ROW(off) OP(0, off) OP(2, off) OP(4, off) OP(6, off) OP(8, off) OP(10, off) OP(12, off) OP(14, off) \
                 OP(16, off) OP(18, off) OP(20, off) OP(22, off) OP(24, off) OP(26, off) OP(28, off) OP(30, off)
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20", \
  "v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42", \
  "v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67"

template <int MODE>
__global__ void __launch_bounds__(1024) k(unsigned long long *clk, int iters)
{
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) asm volatile(ROW(32) ROW(32) ROW(32) ROW(32) ::: CLOB);   // src1 = v[32+2i]: same bank pair as dst
        if (MODE == 1) asm volatile(ROW(34) ROW(34) ROW(34) ROW(34) ::: CLOB);   // src1 = v[34+2i]: other bank pair
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { clk[2 * (threadIdx.x >> 6)] = t0; clk[2 * (threadIdx.x >> 6) + 1] = t1; }
}

template <int MODE>
void run(const char *name, unsigned long long *clk)
{
    const int iters = 20000;
    for (int w = 1; w <= 4; w *= 2) {
        k<MODE><<<256, 256 * w>>>(clk, iters);
        (void)hipDeviceSynchronize();
        unsigned long long c[32]; (void)hipMemcpy(c, clk, 32 * 8, hipMemcpyDeviceToHost);
        unsigned long long lo = ~0ull, hi = 0;
        for (int q = 0; q < 4 * w; ++q) { lo = std::min(lo, c[2 * q]); hi = std::max(hi, c[2 * q + 1]); }
        printf("%-28s waves/SIMD=%d  cycles/instr/SIMD=%.2f  wave0 cycles/instr=%.2f\n", name, w,
               (hi - lo) / ((double)iters * 64 * w), (c[1] - c[0]) / ((double)iters * 64));
    }
}

int main()
{
    unsigned long long *clk; (void)hipMalloc(&clk, 32 * 8);
    k<0><<<256, 1024>>>(clk, 50000); (void)hipDeviceSynchronize();
    run<0>("same bank pair", clk);
    run<1>("other bank pair", clk);
    return 0;
}
