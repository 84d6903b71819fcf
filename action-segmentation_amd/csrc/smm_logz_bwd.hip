// smm_logz_bwd.hip -- gradient of the log-partition (posterior marginals) of the factored semi-Markov model.
//
// Replaces the autograd backward through torch_struct's LogSemiring DP and log_hsmm (reference
// src/models/semimarkov/semimarkov.py:286 `loss.backward()` through semimarkov_modules.py:416-523, 657).
// CPU statement: oracle/smm_oracle.c: smm_oracle_logz (want_grad branch).
//
// Inputs are the two histories left in the workspace:
//   forward  (smm_logz_kernel)            F_cum[n][c] = cumE,  F_h[s][c] = start[s][c] - cumE[s][c],  F_g[n][c] = gamma
//   backward (same kernel, time reversed,  B_cum[j][c] = cumE' (reversed prefix), B_h[j][c] = bend[T-j][c] - cumE'[j][c],
//             transposed transitions)      B_g[j][c]  = bstart[T-j][c]
// With them every marginal is local:
//   P(a span of c starts at s) = exp(F_h[s][c] + F_cum[s][c] + B_g[T-s][c] - logZ)
//   P(a span of c ends at n)   = exp(F_g[n][c] + B_h[T-n][c] + B_cum[T-n][c] - logZ)
//   d logZ / d elp[t][c]       = #spans of c covering t = sum_{s<=t} Pstart - sum_{n<=t} Pend          O(T C)
//   d logZ / d trans[to][from] = sum_n exp(F_g[n][from] + trans[to][from] + B_g[T-n][to] - logZ)         O(T C^2)
//   d logZ / d init[c]         = Pstart(0, c)
//   d logZ / d len[k][c]       = sum_s exp(F_h[s][c] + len[k][c] + B_h[T-s-k][c] + cumE[T][c] - logZ)    O(T K C)
// Only the last one is K-proportional; it has no serial dependence at all: lane = k (length scores and
// accumulators in registers), loop over s with one coalesced sliding-window load per cell, no reduction.
#include "smm_device.h"
#include "smm_launch.h"
#include "../../include/smmdp.h"

#define SMM_LOG2E_F 1.4426950408889634f

__device__ __forceinline__ double smm_expd(double x)     // exp of a non-positive (up to rounding) log-probability
{
    return (double)__builtin_amdgcn_exp2f((float)x * SMM_LOG2E_F);
}

__global__ void smm_transpose2d_kernel(const double *src, double *dst, int g, int cm)
{
    const int n = g * cm * cm;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int gi = i / (cm * cm), r = (i / cm) % cm, c = i % cm;
        dst[(size_t)gi * cm * cm + (size_t)c * cm + r] = src[i];
    }
}

// grid (videos, slabs of SMM_GTRANS_SLAB positions): every workgroup takes its slab of the transition sums (d); slab 0
// also does everything that needs the whole video in one workgroup (a, b, c, e)
#define SMM_GTRANS_SLAB 512
__global__ void __launch_bounds__(1024) smm_marginals_kernel(SmmBwdArgs a)
{
    const int vid = blockIdx.x;
    const SmmVideo mv = a.videos[vid];
    const int T = mv.T - a.no_eos, g = mv.group, cm = a.c_max;
    const int C = a.n_states[g];
    const size_t blk = (size_t)cm * (T + 1);
    const double *F_cum = a.hist + mv.hist_off, *F_h = F_cum + blk, *F_g = F_h + blk;
    const double *B_cum = F_g + blk, *B_h = B_cum + blk, *B_g = B_h + blk;
    double *hT0 = const_cast<double *>(B_g + blk), *hT1 = hT0 + blk;          // state-major copies for the length pass
    const double lz = a.logz[vid];
    const double up = a.grad_logz ? a.grad_logz[vid] : 1.0;
    const double *trans = a.trans + (size_t)g * cm * cm;
    const int tid = threadIdx.x, nth = blockDim.x;

    if ((int)blockIdx.y * SMM_GTRANS_SLAB >= T) return;
    // (d) d/d trans: threads = (pair, slice of this slab's n)
    {
        const int P = C * C;
        const int ng = nth / P;
        if (ng >= 1 && tid < ng * P) {
            const int pair = tid % P, sl = tid / P;
            const int to = pair / C, from = pair - to * C;
            const double tw = trans[(size_t)to * cm + from] - lz;
            const int n0 = 1 + (int)blockIdx.y * SMM_GTRANS_SLAB;
            const int n1 = (n0 + SMM_GTRANS_SLAB < T) ? n0 + SMM_GTRANS_SLAB : T;
            double acc = 0.0;
#pragma unroll 8
            for (int n = n0 + sl; n < n1; n += ng)
                acc += smm_expd(F_g[(size_t)n * cm + from] + tw + B_g[(size_t)(T - n) * cm + to]);
            atomicAdd(&a.g_trans[(size_t)g * cm * cm + (size_t)to * cm + from], up * acc);
        }
    }
    if (blockIdx.y != 0) return;
    // (a) state-major copies: hT0[c][s] = F_h[s][c], hT1[c][j] = B_h[j][c]
    for (size_t i = tid; i < blk; i += nth) {
        const int n = (int)(i / cm), c = (int)(i - (size_t)n * cm);
        hT0[(size_t)c * (T + 1) + n] = F_h[i];
        hT1[(size_t)c * (T + 1) + n] = B_h[i];
    }

    // (b) d/d elp: running (#starts - #ends) per state; threads = (state, chunk of frames), two passes
    __shared__ double part[32][33];
    const int c = tid & 31, j = tid >> 5, nj = nth >> 5;
    const int cs = (T + nj - 1) / nj;
    const int t0 = j * cs, t1 = (t0 + cs < T) ? t0 + cs : T;
    // Pass 1 leaves delta(t) in g_elp and the chunk sums in LDS; pass 2 turns them into running sums.  PB positions
    // per round, their 6 x PB loads issued before the first exp: one trip to the history per round instead of one per
    // position (the loop was latency-bound: 64 dependent trips per thread at T = 2048).  PB = 4: with eight, the 48 loads in
    // flight + the exps' temporaries overflowed the 128 registers a 1024-thread workgroup leaves (21 spilled, 88 B of scratch).
    constexpr int PB = 4;
    double sum = 0.0;
    if (c < C) {
        for (int t = t0; t < t1; t += PB) {
            double fs[PB], fe[PB];
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int tt = (t + u < t1) ? t + u : t1 - 1;
                fs[u] = F_h[(size_t)tt * cm + c] + F_cum[(size_t)tt * cm + c] + B_g[(size_t)(T - tt) * cm + c] - lz;
                fe[u] = F_g[(size_t)tt * cm + c] + B_h[(size_t)(T - tt) * cm + c] + B_cum[(size_t)(T - tt) * cm + c] - lz;
            }
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                if (t + u >= t1) break;
                const double d = (t + u == 0) ? exp(fs[u]) : exp(fs[u]) - exp(fe[u]);      // O(T C) terms: full fp64 exp
                sum += d;
                a.g_elp[(size_t)(mv.frame_off + t + u) * cm + c] = d;
            }
        }
    }
    if (j < 32) part[c][j] = sum;
    __syncthreads();
    if (c < C) {
        double run = 0.0;
        for (int q = 0; q < j && q < 32; ++q) run += part[c][q];
        for (int t = t0; t < t1; t += 8) {
            double d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) d[u] = a.g_elp[(size_t)(mv.frame_off + ((t + u < t1) ? t + u : t1 - 1)) * cm + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (t + u >= t1) break;
                run += d[u];
                a.g_elp[(size_t)(mv.frame_off + t + u) * cm + c] = up * run;
            }
        }
    }
    // (e) no EOS (add_eos=False): the closing transition into the label of frame T, which only emits:
    //     P(last label = to) = exp( LSE_c(F_g[T][c] + trans[to][c]) + elp[T][to] - logZ ) goes to elp[T][to], and its
    //     summands to trans[to][c].  (With EOS the closing weights are constants.)
    if (a.no_eos && tid < C * C) {
        const int to = tid / C, from = tid - to * C;
        const double e_last = a.elp[(size_t)(mv.frame_off + T) * cm + to];
        const double p = up * exp(F_g[(size_t)T * cm + from] + trans[(size_t)to * cm + from] + e_last - lz);
        atomicAdd(&a.g_trans[(size_t)g * cm * cm + (size_t)to * cm + from], p);
        atomicAdd(&a.g_elp[(size_t)(mv.frame_off + T) * cm + to], p);
    }
    // (c) d/d init
    if (tid < C)
        atomicAdd(&a.g_init[(size_t)g * cm + tid],
                  up * exp(F_h[tid] + F_cum[tid] + B_g[(size_t)T * cm + tid] - lz));
}

// d/d len[k][c]: grid (b * c_max, s tiles of SCH, k tiles of KT); thread = (k, slice of the s tile).  KT = the
// launch's largest length rounded up to a power of two (32..256): at K = 64 a lane-per-k layout would leave three
// quarters of the workgroup idle and a 2048-position loop in every thread; the slices are merged in LDS.
#define SMM_GLEN_SCH 512
__global__ void __launch_bounds__(256) smm_glen_kernel(SmmBwdArgs a, int kt)
{
    __shared__ double s_acc[256];
    const int cm = a.c_max;
    const int vid = blockIdx.x / cm, c = blockIdx.x - vid * cm;     // (x: the only grid dimension that goes past 65 535)
    const SmmVideo mv = a.videos[vid];
    const int T = mv.T - a.no_eos, g = mv.group;
    if (c >= a.n_states[g]) return;                                 // (uniform per workgroup)
    const int kk = threadIdx.x % kt, sl = threadIdx.x / kt, ns = 256 / kt;
    const int k = 1 + blockIdx.z * kt + kk;
    const int st = blockIdx.y * SMM_GLEN_SCH;
    if (st >= T) return;
    const int kmax = (mv.kp - 1 < T) ? mv.kp - 1 : T;
    const size_t blk = (size_t)cm * (T + 1);
    const double *F_cum = a.hist + mv.hist_off;
    const double *hT0 = F_cum + 6 * blk + (size_t)c * (T + 1);      // F_h[s][c] over s
    const double *hT1 = hT0 + blk;                                  // B_h[j][c] over j
    const double base = F_cum[(size_t)T * cm + c] - a.logz[vid];    // cumE[T][c] - logZ
    double acc = 0.0;
    if (k <= kmax) {
        const double lk = a.len[((size_t)g * a.k_rows + k) * cm + c] + base;
        const int per = SMM_GLEN_SCH / ns;
        const int s0 = st + sl * per;
        const int s1 = (s0 + per < T - k + 1) ? s0 + per : T - k + 1;   // s + k <= T
#pragma unroll 8
        for (int s = s0; s < s1; ++s) acc += smm_expd(hT0[s] + lk + hT1[T - s - k]);
    }
    s_acc[threadIdx.x] = acc;
    __syncthreads();
    if (sl == 0 && k <= kmax) {
        for (int q = 1; q < ns; ++q) acc += s_acc[q * kt + kk];
        if (acc != 0.0) {
            const double up = a.grad_logz ? a.grad_logz[vid] : 1.0;
            atomicAdd(&a.g_len[((size_t)g * a.k_rows + k) * cm + c], up * acc);
        }
    }
}

void smm_launch_transpose(const double *src, double *dst, int g, int cm, hipStream_t stream)
{
    hipLaunchKernelGGL(smm_transpose2d_kernel, dim3(8), dim3(256), 0, stream, src, dst, g, cm);
}

void smm_launch_marginals(const SmmBwdArgs &a, int t_max, int kp_max, hipStream_t stream)
{
    hipLaunchKernelGGL(smm_marginals_kernel, dim3(a.b, (t_max + SMM_GTRANS_SLAB - 1) / SMM_GTRANS_SLAB), dim3(1024), 0, stream, a);
    int kt = 32;
    while (kt < 256 && kt < kp_max - 1) kt *= 2;
    dim3 grid(a.b * a.c_max, (t_max + SMM_GLEN_SCH - 1) / SMM_GLEN_SCH, (kp_max - 1 + kt - 1) / kt);
    if (kp_max >= 2) hipLaunchKernelGGL(smm_glen_kernel, grid, dim3(256), 0, stream, a, kt);
}
