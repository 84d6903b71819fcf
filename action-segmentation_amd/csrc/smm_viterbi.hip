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

// Diagnostic build only (-DSMM_PROFILE, never shipped): s_memtime stamps around the phases of a frame step,
// summed per workgroup-0/wave-0 and written to the workspace's error block (+64 bytes).
#ifdef SMM_PROFILE
#define SMM_STAMP(var)                                                            \
    do {                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                        \
    } while (0)
#define SMM_ACC(slot, t_from, t_to) prof_acc[slot] += (t_to) - (t_from)
#else
#define SMM_STAMP(var) do { } while (0)
#define SMM_ACC(slot, a, b) do { } while (0)
#endif

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
    constexpr int NP = (SPW + 3) / 4;                 // transition passes: 4 target states (DPP rows) per pass
    const int vid = a.order[blockIdx.x];
    const SmmVideo mv = a.videos[vid];
    const int T = mv.T;
    const int g = mv.group;
    const int C = a.n_states[g];
    const int cm = a.c_max;
    const int kp = mv.kp;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform on purpose: scalar branches
    const int lane = threadIdx.x & 63;
    const int row = lane >> 4, col = lane & 15;
    // wave w owns states w, w+NW, w+2NW, ... (interleaved: waves w and w+4 share a SIMD, so SIMDs stay balanced)
    const int nv_all = (C - w + NW - 1) / NW;
    const int nv = nv_all < 0 ? 0 : (nv_all > SPW ? SPW : nv_all);
#define SMM_STATE(j) ((j) * NW + w)

    const double *trans = a.trans + (size_t)g * cm * cm;
    const double *init = a.init + (size_t)g * cm;
    const double *len = a.len + (size_t)g * a.k_rows * cm;
    const double *elp = a.elp + (size_t)mv.frame_off * cm;
    const double *endpen = a.endpen ? a.endpen + (size_t)vid * cm : nullptr;
    const int64_t *cmap = a.class_map ? a.class_map + (size_t)g * (cm + 1) : nullptr;
    double *hcum = a.hist + mv.hist_off;                  // [c][T+1]
    double *hh = hcum + (size_t)cm * (T + 1);             // [c][T+1]
    double *hgam = hh + (size_t)cm * (T + 1);             // [T+1][cm]  gamma[n][c], frame-major
    int64_t *spans = a.spans ? a.spans + (size_t)vid * (a.t_max + 1) : nullptr;
    int64_t *labels = a.labels ? a.labels + mv.frame_off : nullptr;

    __shared__ __attribute__((aligned(16))) double gam2[2][16][2];   // gamma[f + 16*half] at [parity][f][half]
    __shared__ int sh_c;
    __shared__ unsigned sh_kmin[3];

    if (T <= 0) return;

    // -------------------------------------------------------------------------------- set-up
    if (spans)
        for (int i = threadIdx.x; i <= a.t_max; i += blockDim.x) spans[i] = -1;
    if (threadIdx.x < 64) (&gam2[0][0][0])[threadIdx.x] = 0.0;

    double A[SPW][R], L[SPW][R];
    double cum[SPW], hs[SPW], ecur[SPW], enxt[SPW], hb_h[SPW], hb_c[SPW];
    double t0[NP], t1[NP];
#pragma unroll
    for (int j = 0; j < SPW; ++j) {
        const int c = SMM_STATE(j);
        const bool ok = j < nv;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int p = lane * R + r;
            A[j][r] = SMM_NEG_INF;
            L[j][r] = (ok && p >= 1 && p <= kp - 1) ? len[(size_t)p * cm + c] : SMM_NEG_INF;
        }
        cum[j] = 0.0;
        hs[j] = ok ? init[c] : 0.0;
        ecur[j] = 0.0;
        enxt[j] = (ok && lane < T) ? elp[(size_t)lane * cm + c] : 0.0;
        hb_h[j] = 0.0;
        hb_c[j] = 0.0;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int jt = 4 * p + row;                                        // target state handled by this DPP row
        const bool ok = jt < nv;
        t0[p] = (ok && col < C) ? trans[(size_t)SMM_STATE(jt) * cm + col] : SMM_NEG_INF;
        t1[p] = (ok && col + 16 < C) ? trans[(size_t)SMM_STATE(jt) * cm + col + 16] : SMM_NEG_INF;
    }
    __syncthreads();

    // -------------------------------------------------------------------------------- forward (values only)
    // Source step n (0 <= n < T) knows h[n] and cumE[n]; it pushes h[n] into the ring and finalises position n+1.
    // Program order inside a step puts the one push position n+1 depends on first, hands gamma[n+1] to LDS, and
    // only then issues the other R-1 pushes, so the K-proportional work sits in the shadow of the LDS round trip
    // and of the barrier.  Global memory is touched once per 64 frames (elp chunk in, history chunk out).
#ifdef SMM_PROFILE
    unsigned long long prof_acc[6] = {0, 0, 0, 0, 0, 0}, ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0;
#endif
    for (int base = 0; base < T; base += 64) {
        // elp of this wave's states: lane <-> frame base+lane; the next chunk is fetched one chunk ahead
#pragma unroll
        for (int j = 0; j < SPW; ++j) {
            ecur[j] = enxt[j];
            const int f = base + 64 + lane;
            enxt[j] = (j < nv && f < T) ? elp[(size_t)f * cm + SMM_STATE(j)] : 0.0;
        }
        const int stop = (T - base < 64) ? T - base : 64;
        for (int i0 = 0; i0 < stop; i0 += R) {
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int i = i0 + u;
                if (i >= stop) break;
                const int n = base + i;                      // n % R == u (64 % R == 0)
                const int nn = n + 1;
                const int rn = (u + 1) % R;                  // register of ring slot nn (static after unrolling)
                const int owner = (nn & (RING - 1)) / R;     // lane of ring slot nn
                const bool is_owner = lane == owner;
                const bool is_hist = lane == i;
                SMM_STAMP(ts0);
#pragma unroll
                for (int j = 0; j < SPW; ++j) {
                    if (j >= nv) break;
                    const int c = SMM_STATE(j);
                    const double e = smm_readlane(ecur[j], i);
                    if (is_hist) { hb_h[j] = hs[j]; hb_c[j] = cum[j]; }             // history of step n
                    A[j][rn] = smm_fmax(A[j][rn], hs[j] + L[j][(rn - u + R) % R]);
                    cum[j] = cum[j] + e;
                    const double gm = cum[j] + A[j][rn];                            // meaningful in the owner lane
                    if (is_owner) {
                        gam2[nn & 1][c & 15][c >> 4] = gm;
                        A[j][rn] = SMM_NEG_INF;                                     // slot now accumulates nn + RING
                    }
                }
                SMM_STAMP(ts1);
#pragma unroll
                for (int j = 0; j < SPW; ++j) {
                    if (j >= nv) break;
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (r != rn) A[j][r] = smm_fmax(A[j][r], hs[j] + L[j][(r - u + R) % R]);
                    L[j][(R - 1 - u + R) % R] = smm_wave_ror1(L[j][(R - 1 - u + R) % R]);
                }
                SMM_STAMP(ts2);
                __syncthreads();
                SMM_STAMP(ts3);
                const double2 gv = *reinterpret_cast<const double2 *>(&gam2[nn & 1][col][0]);
                if (w == (nn & (NW - 1)) && row == 0) {            // gamma[nn][.] -> history (128 B per store)
                    if (col < C) hgam[(size_t)nn * cm + col] = gv.x;
                    if (col + 16 < C) hgam[(size_t)nn * cm + col + 16] = gv.y;
                }
                if (nn < T) {
                    SMM_STAMP(ts4);
                    double beta[NP];
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        beta[p] = smm_row_max16(smm_fmax(gv.x + t0[p], gv.y + t1[p]));
#pragma unroll
                    for (int j = 0; j < SPW; ++j) {
                        if (j >= nv) break;
                        hs[j] = smm_readlane(beta[j / 4], 16 * (j % 4)) - cum[j];
                    }
                    SMM_STAMP(ts5);
                    SMM_ACC(0, ts0, ts1); SMM_ACC(1, ts1, ts2); SMM_ACC(2, ts2, ts3); SMM_ACC(3, ts3, ts4);
                    SMM_ACC(4, ts4, ts5); SMM_ACC(5, ts0, ts5);
                }
            }
        }
        // history chunk out: h[n], cumE[n] for n in [base, base + stop)
        if (lane < stop) {
#pragma unroll
            for (int j = 0; j < SPW; ++j) {
                if (j >= nv) break;
                const size_t o = (size_t)SMM_STATE(j) * (T + 1) + base + lane;
                hcum[o] = hb_c[j];
                hh[o] = hb_h[j];
            }
        }
    }
#ifdef SMM_PROFILE
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long *pp = reinterpret_cast<unsigned long long *>(a.err) + 8;
        for (int q = 0; q < 6; ++q) pp[q] = prof_acc[q];
        pp[6] = (unsigned long long)T;
    }
#endif
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < SPW; ++j)
            if (j < nv) hcum[(size_t)SMM_STATE(j) * (T + 1) + T] = cum[j];
    }

    // -------------------------------------------------------------------------------- last position
    // gam[T&1][.] holds gamma[T][.]; candidates fin[to], to = 0..C (C = EOS): first maximal entry wins.
    if (w == 0) {
        double f = SMM_NEG_INF;
        if (lane <= C) {
            for (int c = 0; c < C; ++c) {
                const double wgt = (lane == C) ? (endpen ? endpen[c] : 0.0) : trans[(size_t)lane * cm + c];
                f = fmax(f, gam2[T & 1][c & 15][c >> 4] + wgt);
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
    if (a.flags & 1) return;
    // At a span start (n, to) the predecessor is the FIRST (k ascending, then from ascending) whose
    //   (cumE[n][from] + (h[n-k][from] + len[k][from])) + w(to, from)  equals the maximum.
    // Adding is monotone, so a hit needs gamma[n][from] + w(to, from) == maximum: phase A (every wave, redundantly,
    // one lane per state) finds the maximum and the usually single state that attains it from 8*C bytes of the
    // gamma history; phase B scans only that state's row for the first k (16 B per candidate), all waves abreast.
    int n = T, to = sh_c, nseg = 0, round = 0;
    if (threadIdx.x == 0) { sh_kmin[0] = 0xffffffffu; sh_kmin[1] = 0xffffffffu; sh_kmin[2] = 0xffffffffu; }
    __syncthreads();
    while (n > 0) {
        const int kmax = (kp - 1 < n) ? kp - 1 : n;
        double wgt = 0.0, gmv = SMM_NEG_INF;
        if (lane < C) {
            wgt = (to == C) ? (endpen ? endpen[lane] : 0.0) : trans[(size_t)to * cm + lane];
            gmv = hgam[(size_t)n * cm + lane] + wgt;
        }
        const double rmax = smm_row_max16(gmv);
        const double best = fmax(smm_readlane(rmax, 0), smm_readlane(rmax, 16));
        unsigned long long fmask = __ballot(lane < C && gmv == best);
        int k = 0x7fffffff, c = 0x7fffffff;
        while (fmask) {
            const int f = __builtin_amdgcn_readfirstlane(__ffsll(fmask) - 1);
            fmask &= fmask - 1;
            const double cn = hcum[(size_t)f * (T + 1) + n];
            const double wf = smm_readlane(wgt, f);
            const double *hrow = hh + (size_t)f * (T + 1);
            const int lim = (kmax < k - 1) ? kmax : k - 1;      // an equal k with a larger state loses
            for (int kb = 0; kb < lim; kb += NW * 64) {
                const int kk = kb + w * 64 + lane + 1;
                bool hit = false;
                if (kk <= lim) hit = ((cn + (hrow[n - kk] + len[(size_t)kk * cm + f])) + wf) == best;
                const unsigned long long m = __ballot(hit);
                const int slot = round % 3;
                if (lane == 0 && m) atomicMin(&sh_kmin[slot], (unsigned)(kb + w * 64 + __ffsll(m)));
                if (threadIdx.x == 0) sh_kmin[(round + 1) % 3] = 0xffffffffu;
                __syncthreads();
                const unsigned kf = sh_kmin[slot];
                ++round;
                if (kf != 0xffffffffu) {
                    if ((int)kf < k) { k = (int)kf; c = f; }
                    break;
                }
            }
        }
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
    }
    if (a.n_segs && threadIdx.x == 0) a.n_segs[vid] = nseg;
}

// ------------------------------------------------------------------------------------------------ dispatch
// (SPW, NW) per state count; VGPR budget = 512/(NW/4) per lane and the kernel needs ~4*R*SPW + 40.
#include <cstdlib>
#include "../../include/smmdp.h"
#include "smm_launch.h"

template <int R, int SPW, int NW>
static void launch_cfg(const SmmDpArgs &a, hipStream_t stream)
{
    hipLaunchKernelGGL((smm_viterbi_kernel<R, SPW, NW>), dim3(a.b), dim3(NW * 64), 0, stream, a);
}

// Configuration: NW waves x SPW states per wave, NW * SPW >= states.  VALU code can only address the 256
// architected VGPRs (the other 256 of the unified file are AGPRs), and a wave needs ~4*R*SPW + 56 of them,
// so R*SPW <= 50; the smallest NW that fits is used (fewer waves = fewer copies of the per-frame transition).
template <int R, int SPW, int NW>
static int launch_if(const SmmDpArgs &a, int spw, int nw, hipStream_t stream)
{
    if (spw == SPW && nw == NW) { launch_cfg<R, SPW, NW>(a, stream); return 1; }
    return 0;
}

template <int R>
static int launch_r(const SmmDpArgs &a, int c_need, hipStream_t stream)
{
    constexpr int SPW_MAX = (50 / R) > 8 ? 8 : (50 / R);
    int nw = 4;
    if (const char *e = std::getenv("SMM_NW")) nw = std::atoi(e);   // tuning aid: minimum wave count
    if (nw != 4 && nw != 8 && nw != 16) nw = 4;
    while (nw <= 16 && (c_need + nw - 1) / nw > SPW_MAX) nw *= 2;
    if (nw > 16) return SMM_ERR_UNSUPPORTED;
    const int spw = (c_need + nw - 1) / nw;
    int hit = 0;
    if constexpr (R <= 4) {
        hit = launch_if<R, 1, 4>(a, spw, nw, stream) || launch_if<R, 2, 4>(a, spw, nw, stream) ||
              launch_if<R, 3, 4>(a, spw, nw, stream) || launch_if<R, 4, 4>(a, spw, nw, stream) ||
              launch_if<R, 5, 4>(a, spw, nw, stream) || launch_if<R, 6, 4>(a, spw, nw, stream) ||
              launch_if<R, 7, 4>(a, spw, nw, stream) || launch_if<R, 8, 4>(a, spw, nw, stream) ||
              launch_if<R, 1, 8>(a, spw, nw, stream) || launch_if<R, 2, 8>(a, spw, nw, stream) ||
              launch_if<R, 3, 8>(a, spw, nw, stream) || launch_if<R, 4, 8>(a, spw, nw, stream) ||
              launch_if<R, 1, 16>(a, spw, nw, stream) || launch_if<R, 2, 16>(a, spw, nw, stream);
    } else if constexpr (R == 8) {
        hit = launch_if<R, 1, 4>(a, spw, nw, stream) || launch_if<R, 2, 4>(a, spw, nw, stream) ||
              launch_if<R, 3, 4>(a, spw, nw, stream) || launch_if<R, 4, 4>(a, spw, nw, stream) ||
              launch_if<R, 5, 4>(a, spw, nw, stream) || launch_if<R, 6, 4>(a, spw, nw, stream) ||
              launch_if<R, 1, 8>(a, spw, nw, stream) || launch_if<R, 2, 8>(a, spw, nw, stream) ||
              launch_if<R, 3, 8>(a, spw, nw, stream) || launch_if<R, 4, 8>(a, spw, nw, stream) ||
              launch_if<R, 1, 16>(a, spw, nw, stream) || launch_if<R, 2, 16>(a, spw, nw, stream);
    } else if constexpr (R == 16) {
        hit = launch_if<R, 1, 4>(a, spw, nw, stream) || launch_if<R, 2, 4>(a, spw, nw, stream) ||
              launch_if<R, 3, 4>(a, spw, nw, stream) || launch_if<R, 2, 8>(a, spw, nw, stream) ||
              launch_if<R, 3, 8>(a, spw, nw, stream) || launch_if<R, 1, 8>(a, spw, nw, stream) ||
              launch_if<R, 1, 16>(a, spw, nw, stream) || launch_if<R, 2, 16>(a, spw, nw, stream);
    }
    return hit ? SMM_OK : SMM_ERR_UNSUPPORTED;
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
