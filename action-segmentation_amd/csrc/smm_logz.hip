// smm_logz.hip -- log-partition (LogSemiring forward) of the factored semi-Markov model for gfx950.
//
// Replaces torch_struct SemiMarkovCRF(scores).partition (reference semimarkov_modules.py:657; the dense
// potentials of modules:416-523 are never built).  Same recurrence as smm_viterbi.hip with (logsumexp, +) for
// (max, +) -- see oracle/smm_oracle.c: smm_oracle_logz for the CPU statement:
//     A[n][c] = LSE_{k=1..min(kp-1,n)} ( h[n-k][c] + len[k][c] ),   gamma = cumE + A,
//     beta[n][to] = LSE_c ( gamma[n][c] + trans[to][c] ),           h = beta - cumE,
//     logZ = LSE over the last position's labels (EOS via endpen, real labels with the -1e9 of em+[T]).
//
// Same wave roles as the Viterbi kernel (one chain wave, lane = state; pusher waves owning SPW states with the
// K-proportional work in registers; one barrier per frame).  What changes is the accumulator: a ring slot keeps an
// ONLINE log-sum-exp  (m = running max in fp64, s = sum of exp(x - m) in fp32)  so that nothing can overflow although
// h[s][c] drifts by tens of nats per frame.  Per lattice cell: x = h + len (fp64), d = x - m (fp64 -> fp32),
// e = exp(-|d|) (one v_exp_f32), s = d > 0 ? s*e + 1 : s + e, m = max(m, x): one transcendental per cell.
// fp32 sums of <= 4096 terms in (0, 1] give log s to ~1e-6 absolute; the tolerance of the path is 1e-4 RELATIVE on
// logZ ~ 1e5..1e6.
#include "smm_device.h"
#include "smm_launch.h"
#include "../../include/smmdp.h"

#define SMM_LOG2E 1.4426950408889634
#define SMM_LN2 0.6931471805599453
#define SMM_MASKED (-1e300)      // "never": finite so that (-inf) - (-inf) cannot happen in x - m
#define SMM_MASKED_F (-3.0e38f)  // the same for the fp32 copy of the length table kept in the rings

__device__ __forceinline__ float smm_exp_neg_abs(float d)   // exp(-|d|)
{
    return __builtin_amdgcn_exp2f(-fabsf(d) * (float)SMM_LOG2E);
}

// online LSE update of (m, s) with x
__device__ __forceinline__ void smm_lse_push(double &m, float &s, double x)
{
    const float d = (float)(x - m);
    const float e = smm_exp_neg_abs(d);
    const bool gt = d > 0.f;
    s = fmaf(s, gt ? e : 1.f, gt ? 1.f : e);
    m = smm_fmax(m, x);
}

// value of an accumulator: m + log(s)   (s >= 1 whenever anything finite was pushed)
__device__ __forceinline__ double smm_lse_value(double m, float s)
{
    return m + (double)(__builtin_amdgcn_logf(s) * (float)SMM_LN2);
}

// log(exp(a) + exp(b)) for doubles of any magnitude, transcendental part in fp32
__device__ __forceinline__ double smm_lse2(double a, double b)
{
    const double mx = smm_fmax(a, b);
    const float d = (float)(a - b);
    const float t = __builtin_amdgcn_logf(1.f + smm_exp_neg_abs(d)) * (float)SMM_LN2;
    return (mx == SMM_NEG_INF) ? mx : mx + (double)t;
}

template <int R, int SPW, int NW, int HF>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(1, NW / 4)))
smm_logz_kernel(SmmDpArgs a, double *logz)
{
    constexpr int RING = 64 * R;
    constexpr int NP = NW - 1;
    const int vid = a.order[blockIdx.x];
    const SmmVideo mv = a.videos[vid];
    const int T = mv.T;
    const int g = mv.group;
    const int C = a.n_states[g];
    const int cm = a.c_max;
    const int kp = mv.kp;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;

    const double *trans = a.trans + (size_t)g * cm * cm;
    const double *init = a.init + (size_t)g * cm;
    const double *len = a.len + (size_t)g * a.k_rows * cm;
    const double *elp = a.elp + (size_t)mv.frame_off * cm;
    const double *endpen = a.endpen ? a.endpen + (size_t)vid * cm : nullptr;
    // bwd (a.flags bit 1): the same recursion on the time-reversed video with the transposed transition table gives
    // the backward messages (see smm_logz_bwd.hip); its history goes to the second half of the video's block.
    const bool bwd = (a.flags & 2) != 0;
    double *hcum = a.hist + mv.hist_off + (bwd ? (size_t)3 * cm * (T + 1) : 0);   // [T+1][cm]  cumE[n][c]
    double *hh = hcum + (size_t)cm * (T + 1);             // [T+1][cm]  h[n][c]   (log-weight of "a span of c starts at n" - cumE)
    double *hgam = hh + (size_t)cm * (T + 1);             // [T+1][cm]  gamma[n][c] (log-weight of "a span of c ends at n")

    __shared__ __attribute__((aligned(16))) double sh_am[2][SMM_MAX_STATES_DEV];    // A'[n][c] max part   pushers -> chain
    __shared__ float sh_as[2][SMM_MAX_STATES_DEV];                                  // A'[n][c] sum part
    __shared__ __attribute__((aligned(16))) double sh_h[2][SMM_MAX_STATES_DEV];     // h[n][c]  chain -> pushers
    __shared__ __attribute__((aligned(16))) double sh_gam[SMM_MAX_STATES_DEV];      // gamma[n][.] chain-private broadcast
    __shared__ __attribute__((aligned(16))) double sh_elp[2][64 * SMM_MAX_STATES_DEV];

    if (T <= 0) return;
    if (threadIdx.x < SMM_MAX_STATES_DEV) {
        const int c = threadIdx.x;
        // start weights: forward = init; backward = weight of "the video ends after a span of c":
        // LSE(endpen[c], LSE_to(trans[to][c]) - 1e9)  (a.trans is the transposed table in that mode: row c)
        double h0 = 0.0;
        if (c < C) {
            if (!bwd) {
                h0 = init[c];
            } else {
                double alt = SMM_NEG_INF;
                for (int t2 = 0; t2 < C; ++t2) alt = smm_lse2(alt, trans[(size_t)c * cm + t2]);
                h0 = smm_lse2(endpen ? endpen[c] : 0.0, alt + SMM_BIG_NEG);
            }
        }
        sh_h[0][c] = h0;
        sh_h[1][c] = 0.0;
        sh_am[0][c] = SMM_NEG_INF; sh_am[1][c] = SMM_NEG_INF;
        sh_as[0][c] = 0.f; sh_as[1][c] = 0.f;
        sh_gam[c] = SMM_NEG_INF;
        if (c < C) { hcum[c] = 0.0; hh[c] = h0; }
    }
    {
        const int nfr = (T < 64) ? T : 64;
        for (int i = threadIdx.x; i < nfr * cm; i += blockDim.x) {
            const int j = i / cm, c = i - j * cm;
            sh_elp[0][i] = elp[(size_t)(bwd ? T - 1 - j : j) * cm + c];
        }
    }
    __syncthreads();

    if (w == 0) {
        // ============================================================================ chain wave (lane = state)
        __builtin_amdgcn_s_setprio(3);
        const int to = lane & 31, half = lane >> 5;
        const bool live = to < C;
        double tr[HF];
#pragma unroll
        for (int i = 0; i < HF; ++i) {
            const int f = half * HF + i;
            tr[i] = (live && f < C) ? trans[(size_t)to * cm + f] : SMM_NEG_INF;
        }
        const double len1 = (live && kp >= 2) ? len[(size_t)cm + to] : SMM_NEG_INF;
        double cum = 0.0;
        double hcur = live ? sh_h[0][to] : SMM_NEG_INF;
        double enext = live ? sh_elp[0][to] : 0.0;
        for (int n = 0; n < T; ++n) {
            const double ecurv = enext;
            const int nn = n + 1;
            enext = (live && nn < T) ? sh_elp[(nn >> 6) & 1][(nn & 63) * cm + to] : 0.0;
            // A[nn] = LSE( sources <= n-1 (from the pushers), source n with k = 1 (own registers) )
            const double am = sh_am[nn & 1][to];
            const float as = sh_as[nn & 1][to];
            const double apart = (as > 0.f) ? smm_lse_value(am, as) : SMM_NEG_INF;
            const double acc = smm_lse2(apart, hcur + len1);
            cum = cum + ecurv;
            const double gm = cum + acc;
            if (half == 0 && live) {
                sh_gam[to] = gm;
                hgam[(size_t)nn * cm + to] = gm;
                hcum[(size_t)nn * cm + to] = cum;
            }
            if (nn < T) {
                // beta[to] = LSE_from (gamma[from] + trans[to][from]); this half folds sources half*HF ..
                const double2 *gp = reinterpret_cast<const double2 *>(&sh_gam[half * HF]);
                double v[HF];
                double mx = SMM_NEG_INF;
#pragma unroll
                for (int q = 0; q < HF / 2; ++q) {
                    const double2 gv = gp[q];
                    v[2 * q] = gv.x + tr[2 * q];
                    v[2 * q + 1] = gv.y + tr[2 * q + 1];
                    mx = smm_fmax(mx, smm_fmax(v[2 * q], v[2 * q + 1]));
                }
                mx = smm_max_halves(mx);                    // common reference of both halves
                float s = 0.f;
                const double ref = (mx == SMM_NEG_INF) ? 0.0 : mx;
#pragma unroll
                for (int i = 0; i < HF; ++i) s += __builtin_amdgcn_exp2f((float)(v[i] - ref) * (float)SMM_LOG2E);
                s += __shfl_xor(s, 32);
                const double beta = (mx == SMM_NEG_INF) ? mx : mx + (double)(__builtin_amdgcn_logf(s) * (float)SMM_LN2);
                hcur = beta - cum;
                if (half == 0 && live) {
                    sh_h[nn & 1][to] = hcur;
                    hh[(size_t)nn * cm + to] = hcur;
                }
            }
            __syncthreads();
        }
        // last position: LSE over fin[to], to = 0..C  (sh_gam holds gamma[T][.])
        double f = SMM_NEG_INF;
        if (!bwd) {
            if (lane <= C) {
                for (int c = 0; c < C; ++c) {
                    const double wgt = (lane == C) ? (endpen ? endpen[c] : 0.0) : trans[(size_t)lane * cm + c] + SMM_BIG_NEG;
                    f = smm_lse2(f, sh_gam[c] + wgt);
                }
            }
        } else if (lane < C) {
            f = sh_gam[lane] + init[lane];               // closes the recursion: must reproduce log Z
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) f = smm_lse2(f, __shfl_xor(f, off));
        if (lane == 0) logz[vid] = f;
    } else {
        // ============================================================================ pusher waves
        int rank = w - 1;
        if (NW == 8) rank = (w == 4) ? NP - 1 : ((w == NW - 1) ? 3 : w - 1);
        const int nv_all = (C - rank + NP - 1) / NP;
        const int nv = nv_all < 0 ? 0 : (nv_all > SPW ? SPW : nv_all);
        // A ring slot keeps m (fp64), s (fp32) and its length score as fp32: 4 registers, so that 21 states x 1024
        // slots fit the register file (with an fp64 copy they spill from 15 states on).  Rounding len to fp32 moves a
        // candidate by <= 6e-8 |len| nats -- 1e-6 where candidates carry weight, against a tolerance of 1e-4 relative
        // on log Z ~ 1e5 -- and the time-reversed run sees the same rounded table, so forward and backward agree.
        double M[SPW][R], hs[SPW];
        float S[SPW][R], L[SPW][R];
#pragma unroll
        for (int j = 0; j < SPW; ++j) {
            const int c = j * NP + rank;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int p = lane * R + r;
                M[j][r] = SMM_NEG_INF;
                S[j][r] = 0.f;
                L[j][r] = (j < nv && p >= 1 && p <= kp - 1) ? fmaxf((float)len[(size_t)p * cm + c], SMM_MASKED_F) : SMM_MASKED_F;
            }
            hs[j] = 0.0;
        }
        constexpr int QMAX = (SMM_MAX_STATES_DEV + NP - 1) / NP;
        double pre[QMAX];
#pragma unroll
        for (int q = 0; q < QMAX; ++q) pre[q] = 0.0;
        const int pidx = (w - 1) * 64 + lane;
        for (int n0 = 0; n0 < T; n0 += R) {
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int n = n0 + u;
                if (n >= T) break;
                const int r2 = (u + 2) % R;
                const bool clear = lane == (n & (RING - 1)) / R;
                const bool hand = lane == ((n + 2) & (RING - 1)) / R;
                if (u == 0 && (n & 31) == 0) {
                    const int nbase = (n & ~63) + 64;
                    const int nel = (T - nbase < 64 ? T - nbase : 64) * cm;
                    if ((n & 63) == 0) {
#pragma unroll
                        for (int q = 0; q < QMAX; ++q) {
                            const int e = pidx + q * NP * 64;
                            if (e < nel) {
                                const int j = e / cm, c = e - j * cm;
                                pre[q] = elp[(size_t)(bwd ? T - 1 - (nbase + j) : nbase + j) * cm + c];
                            }
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < QMAX; ++q) {
                            const int e = pidx + q * NP * 64;
                            if (e < nel) sh_elp[(nbase >> 6) & 1][e] = pre[q];
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < SPW; ++j) {
                    if (j >= nv) break;
                    hs[j] = smm_fmax(sh_h[n & 1][j * NP + rank], SMM_MASKED);   // h[n][c], never -inf (x - m must not be NaN)
                }
#pragma unroll
                for (int j = 0; j < SPW; ++j) {
                    if (j >= nv) break;
                    if (clear) { M[j][u] = SMM_NEG_INF; S[j][u] = 0.f; }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        smm_lse_push(M[j][r], S[j][r], hs[j] + (double)L[j][(r - u + R) % R]);
                        // four slots in flight are enough to cover the exp latency; without the fence the scheduler
                        // interleaves all R updates and their temporaries push the rings out of the register file
                        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                    }
                    if (hand) {
                        sh_am[n & 1][j * NP + rank] = M[j][r2];
                        sh_as[n & 1][j * NP + rank] = S[j][r2];
                    }
                    L[j][(R - 1 - u + R) % R] = smm_wave_ror1f(L[j][(R - 1 - u + R) % R]);
                }
                __syncthreads();
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ dispatch
template <int R, int SPW, int NW>
static int logz_launch_if(const SmmDpArgs &a, double *logz, int spw, int nw, int c_need, hipStream_t stream)
{
    if (spw != SPW || nw != NW) return 0;
    if (c_need <= 16) hipLaunchKernelGGL((smm_logz_kernel<R, SPW, NW, 8>), dim3(a.b), dim3(NW * 64), 0, stream, a, logz);
    else hipLaunchKernelGGL((smm_logz_kernel<R, SPW, NW, 16>), dim3(a.b), dim3(NW * 64), 0, stream, a, logz);
    return 1;
}

template <int R>
static int logz_launch_r(const SmmDpArgs &a, double *logz, int c_need, hipStream_t stream)
{
    // a pusher needs ~4*R*SPW + 50 VGPRs (m fp64, s fp32, len fp32 per slot): 8 waves -> R*SPW <= 50, 16 waves -> <= 19
    constexpr int SPW8 = (50 / R) > 5 ? 5 : (50 / R);
    int nw = 8;
    if ((c_need + 6) / 7 > SPW8) nw = 16;
    int spw = (c_need + nw - 2) / (nw - 1);
    if (nw == 16 && 4 * R * spw + 50 > 128) {
        // K > 512 with more than 15 states: the rings no longer fit the register file.  8 waves x 3..5 states per
        // pusher with the overflow in scratch: correct, a few times slower (a two-CU split as in the Viterbi kernel's
        // PAIR mode is the fast answer and is not built for the log semiring).
        nw = 8;
        spw = (c_need + 6) / 7;
    }
    int hit = 0;
    if constexpr (R <= 4) {
        hit = logz_launch_if<R, 1, 8>(a, logz, spw, nw, c_need, stream) || logz_launch_if<R, 2, 8>(a, logz, spw, nw, c_need, stream) ||
              logz_launch_if<R, 3, 8>(a, logz, spw, nw, c_need, stream) || logz_launch_if<R, 4, 8>(a, logz, spw, nw, c_need, stream) ||
              logz_launch_if<R, 5, 8>(a, logz, spw, nw, c_need, stream);
    } else if constexpr (R == 8) {
        hit = logz_launch_if<R, 1, 8>(a, logz, spw, nw, c_need, stream) || logz_launch_if<R, 2, 8>(a, logz, spw, nw, c_need, stream) ||
              logz_launch_if<R, 3, 8>(a, logz, spw, nw, c_need, stream) || logz_launch_if<R, 4, 8>(a, logz, spw, nw, c_need, stream) ||
              logz_launch_if<R, 5, 8>(a, logz, spw, nw, c_need, stream) ||
              logz_launch_if<R, 1, 16>(a, logz, spw, nw, c_need, stream);
    } else if constexpr (R == 16) {
        hit = logz_launch_if<R, 1, 8>(a, logz, spw, nw, c_need, stream) || logz_launch_if<R, 2, 8>(a, logz, spw, nw, c_need, stream) ||
              logz_launch_if<R, 1, 16>(a, logz, spw, nw, c_need, stream) ||
              logz_launch_if<R, 3, 8>(a, logz, spw, nw, c_need, stream) || logz_launch_if<R, 4, 8>(a, logz, spw, nw, c_need, stream) ||
              logz_launch_if<R, 5, 8>(a, logz, spw, nw, c_need, stream);
    }
    return hit ? SMM_OK : SMM_ERR_UNSUPPORTED;
}

int smm_launch_logz(const SmmDpArgs &a, double *logz, int r, int c_need, hipStream_t stream)
{
    switch (r) {
    case 1: return logz_launch_r<1>(a, logz, c_need, stream);
    case 2: return logz_launch_r<2>(a, logz, c_need, stream);
    case 4: return logz_launch_r<4>(a, logz, c_need, stream);
    case 8: return logz_launch_r<8>(a, logz, c_need, stream);
    case 16: return logz_launch_r<16>(a, logz, c_need, stream);
    default: return SMM_ERR_UNSUPPORTED;
    }
}
