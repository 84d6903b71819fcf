// smm_emission.hip -- diagonal-Gaussian emission scorer for gfx950 (fp64 FMA on the VALU, HBM-streaming).
//
// Replaces SemiMarkovModule.emission_log_probs / _emission_log_probs_with_means (reference
// semimarkov_modules.py:324-381): elp[t][c] = log N(x_t; mu_c, diag(sigma^2)) (+ constraints[t][c], :379-380),
// evaluated in the expanded form
//     elp[t][c] = cst[c] + sum_d x[t][d] * w[d][c] - 0.5 * sum_d x[t][d]^2 * inv_var[d]
// (w = mu/sigma^2, cst = -0.5 sum mu^2/sigma^2 - sum log sigma - D/2 log 2pi; host-side, fp64), which is one
// v_fma_f64 per (frame, d, state).  In fp64 the expansion costs nothing in accuracy (|terms| ~ 1e3, eps 1e-16).
//
// Mapping: see the comment above smm_emission_kernel (fp64 MFMA tiles, weights in LDS, 64-B row pieces of x).
#include "smm_launch.h"

// ---------------------------------------------------------------------------------------------------------------
// elp = x . w is a [frames x D] x [D x C] product: it runs on the fp64 matrix cores (v_mfma_f64_16x16x4_f64: a
// 16-frame x 16-state tile per instruction, 4 features deep).  fp64 MFMA has the fp64 VALU's FLOP rate on gfx950, but
// one instruction replaces 1024 FMAs, needs no scalar operand stream, and leaves the VALU free for the x^2 term.
//   A operand (lane l): x[frame f0 + (l & 15)][feature]  -- each lane loads 16 B = 4 consecutive features of its frame,
//       so a wave reads 16 rows x 64 contiguous bytes per macro-step of 16 features (whole 64-B sectors, once);
//       the 4 features of a lane feed 4 MFMAs (k index = l >> 4), i.e. MFMA j of a macro-step contracts features
//       {d0 + 4k + j}.  Any feature <-> (j, k) assignment is valid as long as B uses the same one.
//   B operand (lane l): w[d0 + 4 (l >> 4) + j][s0 + (l & 15)]  from an LDS copy of the group's weight table.
//   C/D (lane l, reg i): frame f0 + (l >> 4) + 4 i, state s0 + (l & 15)  -> 128-B contiguous row pieces on store.
// One workgroup = 4 waves on one video; every wave walks 16-frame tiles with a grid stride.
typedef double smm_d4 __attribute__((ext_vector_type(4)));

template <int NT>   // state tiles of 16 (1: C <= 16, 2: C <= 32)
__global__ void __launch_bounds__(256)
smm_emission_kernel(const SmmVideo *__restrict__ videos, const int32_t *__restrict__ n_states,
                    const float *__restrict__ xall, const double *__restrict__ wall, const double *__restrict__ cstall,
                    const double *__restrict__ iv, const float *__restrict__ cons, double *__restrict__ elp64,
                    float *__restrict__ elp32, int D, int cm)
{
    extern __shared__ __attribute__((aligned(16))) double wl[];      // [D16][16*NT + 1]: this group's weights (zero padded) | inv_var
    const int vid = blockIdx.y;
    const SmmVideo mv = videos[vid];
    const int T = mv.T, g = mv.group;
    const int C = n_states[g];
    const int ntiles = (T + 15) >> 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((int)(blockIdx.x * 4) >= ntiles) return;                     // whole block has nothing to do
    const int D16 = (D + 15) & ~15;
    constexpr int WS = 16 * NT + 1;                                  // LDS row stride (doubles): weights + inv_var
    const int nt = (NT == 2 && C > 16) ? 2 : 1;                      // state tiles this video really needs
    {
        const double *__restrict__ w = wall + (size_t)g * D * cm;
        for (int i = threadIdx.x; i < D16 * WS; i += 256) {
            const int d = i / WS, c = i - d * WS;
            wl[i] = (d >= D) ? 0.0 : ((c == 16 * NT) ? iv[d] : ((c < C) ? w[(size_t)d * cm + c] : 0.0));
        }
    }
    __syncthreads();
    const int fr = lane & 15, kq = lane >> 4;
    const float *__restrict__ xv = xall + (size_t)mv.frame_off * D;
    const double *__restrict__ cst = cstall + (size_t)g * cm;

    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int f0 = tile << 4;
        const int f = (f0 + fr < T) ? f0 + fr : T - 1;               // clamp: rows past the end are computed, not stored
        const float *__restrict__ xrow = xv + (size_t)f * D;
        smm_d4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (smm_d4){0.0, 0.0, 0.0, 0.0};
        double q = 0.0;
        auto load4 = [&](int d0) -> float4 {
            const int db = d0 + 4 * kq;                              // this lane's 4 features of the macro-step
            float4 r;
            if (db + 3 < D && (D & 3) == 0) {
                r = *reinterpret_cast<const float4 *>(xrow + db);
            } else {
                r.x = (db + 0 < D) ? xrow[db + 0] : 0.f;
                r.y = (db + 1 < D) ? xrow[db + 1] : 0.f;
                r.z = (db + 2 < D) ? xrow[db + 2] : 0.f;
                r.w = (db + 3 < D) ? xrow[db + 3] : 0.f;
            }
            return r;
        };
        float4 xn1 = load4(0), xn2 = load4(16);                      // two macro-steps of x in flight ahead of the MFMAs
        for (int d0 = 0; d0 < D16; d0 += 16) {
            const int db = d0 + 4 * kq;
            const float4 x4 = xn1;
            xn1 = xn2;
            xn2 = load4(d0 + 32);
            const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double a = (double)xs[j];
                const int d = db + j;
                q = fma(a * wl[(size_t)d * WS + 16 * NT], a, q);
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, wl[(size_t)d * WS + fr], acc[0], 0, 0, 0);
                if (NT == 2 && nt == 2)
                    acc[NT - 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, wl[(size_t)d * WS + 16 + fr], acc[NT - 1], 0, 0, 0);
            }
        }
        // q: this lane summed the features with k index kq of frame fr; add the four k groups (lanes fr + 16 k)
        q += __shfl_xor(q, 16);
        q += __shfl_xor(q, 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = kq + 4 * i;                              // frame of accumulator register i
            const double qr = __shfl(q, row);
            const int ff = f0 + row;
            if (ff < T) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int c = 16 * t + fr;
                    if (c < C) {
                        const size_t o = (size_t)(mv.frame_off + ff) * cm + c;
                        double v = (cst[c] + acc[t][i]) - 0.5 * qr;
                        if (cons) v += (double)cons[o];
                        if (elp64) elp64[o] = v;
                        if (elp32) elp32[o] = (float)v;
                    }
                }
            }
        }
    }
}

// fp32 -> fp64 widening of a [n] array (smm_viterbi_f32 boundary)
__global__ void smm_widen_kernel(const float *src, double *dst, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (double)src[i];
}

void smm_launch_emission(const SmmEmArgs &a, int ct, int t_max, hipStream_t stream)
{
    const int d16 = (a.d + 15) & ~15;
    const int tiles = (t_max + 15) / 16;
    int bx = (tiles + 4 * 8 - 1) / (4 * 8);                 // ~8 tiles per wave: amortises the LDS fill of the weights
    if (bx < 1) bx = 1;
    dim3 grid(bx, a.b), block(256);
    const size_t lds = sizeof(double) * d16 * (ct <= 16 ? 17 : 33);          // <= 160 KiB checked by the caller
    if (ct <= 16) {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(smm_emission_kernel<1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(smm_emission_kernel<1>, grid, block, lds, stream, a.videos, a.n_states, a.x, a.w, a.cst,
                           a.inv_var, a.cons, a.elp64, a.elp32, a.d, a.c_max);
    } else {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(smm_emission_kernel<2>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(smm_emission_kernel<2>, grid, block, lds, stream, a.videos, a.n_states, a.x, a.w, a.cst,
                           a.inv_var, a.cons, a.elp64, a.elp32, a.d, a.c_max);
    }
}

void smm_launch_widen(const float *src, double *dst, size_t n, hipStream_t stream)
{
    if (n == 0) return;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(smm_widen_kernel, dim3(blocks), dim3(256), 0, stream, src, dst, n);
}
