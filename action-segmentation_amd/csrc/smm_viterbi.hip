// smm_viterbi.hip -- factored semi-Markov Viterbi for gfx950 (wave64, fp64 VALU, no MFMA).
//
// Replaces SemiMarkovModule.log_hsmm (reference semimarkov_modules.py:416-523) + torch_struct
// SemiMarkovCRF.argmax / from_parts (modules:677-679) + class un-mapping (modules:683-691) +
// spans_to_labels / trim (semimarkov_utils.py:51-63, modules:532-543).  The recurrence, its association
// order and the arg-max order are specified in oracle/smm_oracle.c (the CPU twin); they are repeated here
// only where the mapping to the hardware needs them.
//
// Mapping.  One workgroup per video, NW waves; wave w owns states c = j*NW + w (j < SPW): "one wavefront
// per (video, state) row".  The DP is run in PUSH form so that nothing has to be reduced across lanes in
// the O(K) part:
//   * ring slot p = n mod RING (RING = 64*R >= kp) holds the accumulator A[n][c] = max_k (h[n-k][c] + len[k][c])
//     for a future position n; slot p lives in lane p/R, register p%R of the state's wave -- all 64*R
//     accumulators of a state stay in VGPRs for the whole video;
//   * when h[s][c] is final it is a wave-uniform scalar; every slot does A = max(A, h[s] + len[k]) with its own
//     k = n - s.  k shrinks by one per step for every slot, so the length table is kept in registers too and
//     ROTATED by one slot per step: one register is renamed (the loop is unrolled R times so the renaming is
//     static) and one crosses to the next lane with a single DPP wave_ror:1 -- no LDS or memory traffic for
//     the K*C work at all;
//   * per frame the only cross-wave step is the C x C transition: gamma[n][c] goes through 8 bytes of LDS per
//     state, one barrier, and a 32-lane DPP max.
// The forward pass keeps VALUES only (2 fp64 VALU ops per lattice cell: v_add_f64 + v_max_f64).  The arg-max
// is recovered afterwards along the optimal path only (one K*C scan per SEGMENT instead of per frame) from
// the h / cumE history, re-evaluating exactly the expressions of the forward pass, so it is bit-identical
// to tracking back-pointers.
//
// HBM traffic per frame (c = states of the video): read elp 8c, write history 16c, write span 8 + label 8;
// back-trace reads 16 B per (k, state) candidate of each segment.
#include "smm_device.h"

template <int R>
struct SmmRing {
    static constexpr int RING = 64 * R;
};

// wave-level lexicographic arg-max: larger val first, then smaller k, then smaller c
__device__ __forceinline__ void smm_best3(double &v, int &k, int &c, double v2, int k2, int c2)
{
    bool take = (v2 > v) || (v2 == v && (k2 < k || (k2 == k && c2 < c)));
    if (take) { v = v2; k = k2; c = c2; }
}

__device__ __forceinline__ void smm_wave_best3(double &v, int &k, int &c)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        double v2 = __shfl_xor(v, off);
        int k2 = __shfl_xor(k, off);
        int c2 = __shfl_xor(c, off);
        smm_best3(v, k, c, v2, k2, c2);
    }
}

template <int R, int SPW, int NW>
__global__ void __launch_bounds__(NW * 64) smm_viterbi_kernel(SmmDpArgs a)
{
    constexpr int RING = 64 * R;
    const int vid = a.order[blockIdx.x];
    const SmmVideo mv = a.videos[vid];
    const int T = mv.T;
    const int g = mv.group;
    const int C = a.n_states[g];
    const int cm = a.c_max;
    const int kp = mv.kp;
    const int w = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;

    const double *trans = a.trans + (size_t)g * cm * cm;
    const double *init = a.init + (size_t)g * cm;
    const double *len = a.len + (size_t)g * a.k_rows * cm;
    const double *elp = a.elp + (size_t)mv.frame_off * cm;
    const double *endpen = a.endpen ? a.endpen + (size_t)vid * cm : nullptr;
    const int64_t *cmap = a.class_map ? a.class_map + (size_t)g * (cm + 1) : nullptr;
    double *hcum = a.hist + mv.hist_off;                  // [c][T+1]
    double *hh = hcum + (size_t)cm * (T + 1);             // [c][T+1]
    int64_t *spans = a.spans ? a.spans + (size_t)vid * (a.t_max + 1) : nullptr;
    int64_t *labels = a.labels ? a.labels + mv.frame_off : nullptr;

    __shared__ double gam[2][SMM_MAX_STATES_DEV];
    __shared__ double red_v[16];
    __shared__ int red_k[16], red_c[16];
    __shared__ int sh_k, sh_c;

    if (T <= 0) return;

    // -------------------------------------------------------------------------------- set-up
    if (spans)
        for (int i = threadIdx.x; i <= a.t_max; i += blockDim.x) spans[i] = -1;

    double A[SPW][R], L[SPW][R];
    double trn[SPW], cum[SPW], hs[SPW], ecur[SPW], enxt[SPW];
    bool valid[SPW];
#pragma unroll
    for (int j = 0; j < SPW; ++j) {
        const int c = j * NW + w;
        valid[j] = c < C;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int p = lane * R + r;
            A[j][r] = SMM_NEG_INF;
            L[j][r] = (valid[j] && p >= 1 && p <= kp - 1) ? len[(size_t)p * cm + c] : SMM_NEG_INF;
        }
        trn[j] = (valid[j] && lane < C) ? trans[(size_t)c * cm + lane] : SMM_NEG_INF;
        cum[j] = 0.0;
        hs[j] = valid[j] ? init[c] : 0.0;
        ecur[j] = 0.0;
        enxt[j] = (valid[j] && lane < T) ? elp[(size_t)lane * cm + c] : 0.0;
        if (valid[j] && lane == 0) { hcum[(size_t)c * (T + 1)] = 0.0; hh[(size_t)c * (T + 1)] = hs[j]; }
    }

    // -------------------------------------------------------------------------------- forward (values only)
    for (int n0 = 0; n0 <= T; n0 += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int n = n0 + u;
            if (n > T) break;
            if (n >= 1) {
                const int owner = (n & (RING - 1)) / R;   // slot n mod RING sits in (lane owner, register u)
                const int t = n - 1;                      // frame consumed by this step
                if ((t & 63) == 0) {
#pragma unroll
                    for (int j = 0; j < SPW; ++j) {
                        ecur[j] = enxt[j];
                        const int f = t + 64 + lane;
                        enxt[j] = (valid[j] && f < T) ? elp[(size_t)f * cm + (j * NW + w)] : 0.0;
                    }
                }
#pragma unroll
                for (int j = 0; j < SPW; ++j) {
                    if (!valid[j]) continue;
                    const int c = j * NW + w;
                    const double acc = smm_readlane(A[j][u], owner);
                    const double e = smm_readlane(ecur[j], t & 63);
                    cum[j] = cum[j] + e;
                    const double gm = cum[j] + acc;
                    if (lane == 0) {
                        gam[n & 1][c] = gm;
                        hcum[(size_t)c * (T + 1) + n] = cum[j];
                    }
                    if (lane == owner) A[j][u] = SMM_NEG_INF;   // the slot now accumulates position n + RING
                }
                __syncthreads();
                if (n < T) {
                    const double gv = (lane < C) ? gam[n & 1][lane] : 0.0;
#pragma unroll
                    for (int j = 0; j < SPW; ++j) {
                        if (!valid[j]) continue;
                        const int c = j * NW + w;
                        const double x = (lane < C) ? gv + trn[j] : SMM_NEG_INF;
                        const double bt = smm_wave_max32(x);
                        hs[j] = bt - cum[j];
                        if (lane == 0) hh[(size_t)c * (T + 1) + n] = hs[j];
                    }
                }
            }
            if (n < T) {
                // push h[n] into every open slot: A[p] = max(A[p], h[n] + len[k(p)]); then rotate the length ring
#pragma unroll
                for (int j = 0; j < SPW; ++j) {
                    if (!valid[j]) continue;
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        A[j][r] = fmax(A[j][r], hs[j] + L[j][(r - u + R) % R]);
                    L[j][(R - 1 - u + R) % R] = smm_wave_ror1(L[j][(R - 1 - u + R) % R]);
                }
            }
        }
    }

    // -------------------------------------------------------------------------------- last position
    // gam[T&1][.] holds gamma[T][.]; candidates fin[to], to = 0..C (C = EOS): first maximal entry wins.
    if (w == 0) {
        double f = SMM_NEG_INF;
        if (lane <= C) {
            for (int c = 0; c < C; ++c) {
                const double wgt = (lane == C) ? (endpen ? endpen[c] : 0.0) : trans[(size_t)lane * cm + c];
                f = fmax(f, gam[T & 1][c] + wgt);
            }
            if (lane < C) f = f + SMM_BIG_NEG;
        }
        int kk = 0, cc = (lane <= C) ? lane : 0x7fffffff;
        if (lane > C) f = SMM_NEG_INF;
        smm_wave_best3(f, kk, cc);
        if (lane == 0) {
            sh_c = cc;
            if (a.best) a.best[vid] = f;
            if (spans) spans[T] = cmap ? cmap[cc] : cc;
        }
    }
    __threadfence_block();
    __syncthreads();

    // -------------------------------------------------------------------------------- back-trace
    int n = T, to = sh_c, nseg = 0;
    while (n > 0) {
        const int kmax = (kp - 1 < n) ? kp - 1 : n;
        double bv = SMM_NEG_INF;
        int bk = 0x7fffffff, bc = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < SPW; ++j) {
            const int c = j * NW + w;
            if (c >= C) continue;
            const double wgt = (to == C) ? (endpen ? endpen[c] : 0.0) : trans[(size_t)to * cm + c];
            const double cn = hcum[(size_t)c * (T + 1) + n];
            const double *hrow = hh + (size_t)c * (T + 1);
            for (int k = 1 + lane; k <= kmax; k += 64) {
                const double val = (cn + (hrow[n - k] + len[(size_t)k * cm + c])) + wgt;
                smm_best3(bv, bk, bc, val, k, c);
            }
        }
        smm_wave_best3(bv, bk, bc);
        if (lane == 0) { red_v[w] = bv; red_k[w] = bk; red_c[w] = bc; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double v = red_v[0]; int k = red_k[0], c = red_c[0];
            for (int i = 1; i < NW; ++i) smm_best3(v, k, c, red_v[i], red_k[i], red_c[i]);
            sh_k = k; sh_c = c;
        }
        __syncthreads();
        const int k = sh_k, c = sh_c;
        if (k < 1 || k > kmax || c < 0 || c >= C) {           // NaN / inf-inf in the inputs: stop, flag, never spin
            if (threadIdx.x == 0) atomicExch(a.err, 1);
            break;
        }
        const int s = n - k;
        const int64_t gid = cmap ? cmap[c] : c;
        if (labels)
            for (int f = s + threadIdx.x; f < n; f += blockDim.x) labels[f] = gid;
        if (spans && threadIdx.x == 0) spans[s] = gid;
        ++nseg;
        n = s;
        to = c;
        __syncthreads();   // red_* / sh_* are rewritten by the next iteration
    }
    if (a.n_segs && threadIdx.x == 0) a.n_segs[vid] = nseg;
}

// ------------------------------------------------------------------------------------------------ dispatch
// (SPW, NW) per state count; VGPR budget = 512/(NW/4) per lane and the kernel needs ~4*R*SPW + 40.
#include "../../include/smmdp.h"
#include "smm_launch.h"

template <int R, int SPW, int NW>
static void launch_cfg(const SmmDpArgs &a, hipStream_t stream)
{
    hipLaunchKernelGGL((smm_viterbi_kernel<R, SPW, NW>), dim3(a.b), dim3(NW * 64), 0, stream, a);
}

template <int R>
static int launch_r(const SmmDpArgs &a, int c_need, hipStream_t stream)
{
    if constexpr (R <= 8) {
        if (c_need <= 16) launch_cfg<R, 1, 16>(a, stream);
        else launch_cfg<R, 2, 16>(a, stream);
        return SMM_OK;
    } else if constexpr (R == 16) {
        if (c_need <= 16) launch_cfg<R, 1, 16>(a, stream);
        else if (c_need <= 24) launch_cfg<R, 3, 8>(a, stream);
        else return SMM_ERR_UNSUPPORTED;
        return SMM_OK;
    } else {
        return SMM_ERR_UNSUPPORTED;
    }
}

int smm_launch_viterbi(const SmmDpArgs &a, int r, int c_need, hipStream_t stream)
{
    switch (r) {
    case 1: return launch_r<1>(a, c_need, stream);
    case 2: return launch_r<2>(a, c_need, stream);
    case 4: return launch_r<4>(a, c_need, stream);
    case 8: return launch_r<8>(a, c_need, stream);
    case 16: return launch_r<16>(a, c_need, stream);
    default: return SMM_ERR_UNSUPPORTED;
    }
}
