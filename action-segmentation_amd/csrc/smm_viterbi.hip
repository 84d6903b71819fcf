// smm_viterbi.hip -- factored semi-Markov Viterbi for gfx950 (wave64, fp64 VALU, no MFMA).
//
// Replaces SemiMarkovModule.log_hsmm (reference semimarkov_modules.py:416-523) + torch_struct
// SemiMarkovCRF.argmax / from_parts (modules:677-679) + class un-mapping (modules:683-691) +
// spans_to_labels / trim (semimarkov_utils.py:51-63, modules:532-543).  The recurrence, its association
// order and the arg-max order are specified in oracle/smm_oracle.c (the CPU twin); they are repeated here
// only where the mapping to the hardware needs them.
//
// Mapping.  One workgroup per video with two kinds of waves:
//
//   PUSHER waves (waves 1..NW-1), each owning SPW states ("one wavefront per (video, state) row").  The DP runs in
//   PUSH form so that nothing K-proportional is ever reduced across lanes or leaves the register file:
//     * ring slot p = n mod RING (RING = 64*R >= kp) holds the accumulator A[n][c] = max_k (h[n-k][c] + len[k][c])
//       of a future position n, in lane p/R, register p%R of the state's wave;
//     * when h[s][c] is final it is a wave-uniform scalar; every slot does A = max(A, h[s] + len[k]), k = n - s:
//       2 fp64 VALU ops per lattice cell (v_add_f64, v_max_f64), nothing else;
//     * k shrinks by one per step for every slot, so the length table lives in registers too and is ROTATED one slot
//       per step: R-1 registers are renamed (the loop is unrolled R times, so statically) and one crosses to the next
//       lane with a single DPP wave_ror:1.
//
//   one CHAIN wave (wave 0) that owns the serial part for ALL states, one lane per state: cumE += elp, gamma = cumE + A,
//   the C x C transition, h = beta - cumE, and the history.  The transition is lane = target state: gamma[.] is
//   broadcast through LDS and every lane folds its own row of the transition table (registers) over it; the two
//   halves of the wave take half of the source states each and are merged with one v_permlane32_swap.
//
// Per frame: pushers do the ONE push the next position depends on, the owner lane hands A[n+1][c] to LDS, barrier 1,
// then the pushers issue the other R-1 pushes of that frame WHILE the chain wave turns A[n+1][.] into h[n+1][.];
// barrier 2; pushers read h[n+1][c].  The K-proportional work and the latency-bound serial chain overlap.
//
// The forward pass keeps VALUES only.  The arg-max is recovered afterwards along the optimal path only (one row
// scan per SEGMENT instead of arg-max tracking per cell) from the history, re-evaluating exactly the expressions of
// the forward pass, so it is bit-identical to tracking back-pointers.
//
// HBM traffic per frame (c states): read elp 8c, write history 24c (cumE, h, gamma; frame-major), label 8 B.
#include "smm_device.h"

// Diagnostic build only (-DSMM_PROFILE, never shipped or timed): s_memtime stamps around the phases of a frame,
// summed for workgroup 0 and written behind the workspace's error word (chain wave: +64 B, pusher wave 1: +192 B).
#ifdef SMM_PROFILE
#define SMM_STAMP(var)                                                            \
    do {                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                        \
    } while (0)
#define SMM_PROF_DECL unsigned long long pa[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0
#define SMM_ACC(slot, t_from, t_to) pa[slot] += (t_to) - (t_from)
#define SMM_PROF_OUT(off)                                                         \
    if (blockIdx.x == 0 && lane == 0) {                                           \
        unsigned long long *pp = reinterpret_cast<unsigned long long *>(a.err) + (off); \
        for (int q = 0; q < 8; ++q) pp[q] = pa[q];                                \
    }
#define SMM_PROF_FRAME                                                            \
    do {                                                                          \
        SMM_ACC(0, q0, q1); SMM_ACC(1, q1, q2); SMM_ACC(2, q2, q3); SMM_ACC(3, q0, q3); \
        if (q4) SMM_ACC(4, q4, q2);                                               \
        SMM_ACC(7, 0, 1); q4 = q3;                                                \
    } while (0)
#define SMM_PROF_WAVE(wv)                                                         \
    if (blockIdx.x == 0 && lane == 0) {                                           \
        unsigned long long *pp = reinterpret_cast<unsigned long long *>(a.err) + 40 + (wv); \
        pp[0] = pa[4];                                                            \
    }
#else
#define SMM_PROF_WAVE(wv) do { } while (0)
#define SMM_PROF_FRAME do { } while (0)
#define SMM_STAMP(var) do { } while (0)
#define SMM_PROF_DECL do { } while (0)
#define SMM_ACC(slot, a, b) do { } while (0)
#define SMM_PROF_OUT(off) do { } while (0)
#endif

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

// One frame of one state's ring (source step n, n % R == u after unrolling):
//   finish the R-1 pushes of h[n-1] that position n+1 did not depend on, rotate the length ring, clear slot n,
//   do the one push of h[n] that slot n+2 needs and hand A'[n+2] (sources <= n) to LDS.
template <int R>
__device__ __forceinline__ void smm_ring_frame(double (&A)[R], double (&L)[R], double &hs, double hn, int n, int u,
                                               int lane, double *apart_slot)
{
    constexpr int RING = 64 * R;
    const int r2 = (u + 2) % R;                                  // register of ring slot n+2 (static after unrolling)
    const bool clear = lane == (n & (RING - 1)) / R;             // lane of ring slot n (register u)
    const bool hand = lane == ((n + 2) & (RING - 1)) / R;        // lane of ring slot n+2
    if (n >= 1) {
        // adds and maxes in groups of four independent registers
#pragma unroll
        for (int r0 = 0; r0 < R; r0 += 4) {
            double tq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (r0 + q < R) tq[q] = hs + L[(r0 + q - u + 1 + R) % R];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (r0 + q < R && r0 + q != (u + 1) % R) A[r0 + q] = smm_fmax(A[r0 + q], tq[q]);
        }
        L[(R - u) % R] = smm_wave_ror1(L[(R - u) % R]);
    }
    hs = hn;
    if (clear) A[u] = SMM_NEG_INF;                               // slot n now accumulates position n + RING
    A[r2] = smm_fmax(A[r2], hs + L[(r2 - u + R) % R]);           // the push slot n+2 waits for
    if (hand) *apart_slot = A[r2];
}

// R   ring registers per lane (RING = 64 R >= kp)      SPW  states per pusher wave
// NW  waves per workgroup (1 chain + NW-1 pushers)       HF   source states per half of the chain wave (8 or 16)
// One workgroup per CU is all that fits (and all that is wanted): tell the register allocator it may use the whole
// architected VGPR budget of NW/4 waves per SIMD instead of spilling for an occupancy nobody asked for.
// CP  1: the chain wave also owns the ring of state (NW-1)*SPW (the 12-wave configuration for 22..23 states)
template <int R, int SPW, int NW, int HF, int CP>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(1, (NW + 3) / 4)))
smm_viterbi_kernel(SmmDpArgs a)
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
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform on purpose: scalar branches
    const int lane = threadIdx.x & 63;

    const double *trans = a.trans + (size_t)g * cm * cm;
    const double *init = a.init + (size_t)g * cm;
    const double *len = a.len + (size_t)g * a.k_rows * cm;
    const double *elp = a.elp + (size_t)mv.frame_off * cm;
    const double *endpen = a.endpen ? a.endpen + (size_t)vid * cm : nullptr;
    const int64_t *cmap = a.class_map ? a.class_map + (size_t)g * (cm + 1) : nullptr;
    double *hcum = a.hist + mv.hist_off;                  // [T+1][cm]  cumE[n][c]
    double *hh = hcum + (size_t)cm * (T + 1);             // [T+1][cm]  h[n][c]
    double *hgam = hh + (size_t)cm * (T + 1);             // [T+1][cm]  gamma[n][c]
    int64_t *spans = a.spans ? a.spans + (size_t)vid * (a.t_max + 1) : nullptr;
    int64_t *labels = a.labels ? a.labels + mv.frame_off : nullptr;

    __shared__ __attribute__((aligned(16))) double sh_apart[2][SMM_MAX_STATES_DEV];   // A'[n][c]   pushers -> chain
    __shared__ __attribute__((aligned(16))) double sh_h[2][SMM_MAX_STATES_DEV];       // h[n][c]    chain -> pushers
    __shared__ __attribute__((aligned(16))) double sh_gam[SMM_MAX_STATES_DEV];        // gamma[n][.] chain-private broadcast
    __shared__ __attribute__((aligned(16))) double sh_elp[2][64 * SMM_MAX_STATES_DEV];  // elp rows of 64 frames, x2
    __shared__ unsigned sh_kmin[3];
    __shared__ int sh_c;

    if (T <= 0) return;
    if (spans)
        for (int i = threadIdx.x; i <= a.t_max; i += blockDim.x) spans[i] = -1;
    if (threadIdx.x < SMM_MAX_STATES_DEV) {
        const int c = threadIdx.x;
        sh_h[0][c] = (c < C) ? init[c] : 0.0;
        sh_h[1][c] = 0.0;
        sh_apart[0][c] = SMM_NEG_INF;
        sh_apart[1][c] = SMM_NEG_INF;
        sh_gam[c] = SMM_NEG_INF;
        if (c < C) { hcum[c] = 0.0; hh[c] = init[c]; }                             // history of n = 0
    }
    // elp reaches the chain wave through LDS: the pusher waves copy 64-frame chunks (64*cm contiguous doubles) one
    // chunk ahead, so the chain wave itself never waits on a vector-memory counter (its history stores stay in flight).
    {
        const int nel = ((T < 64) ? T : 64) * cm;
        for (int i = threadIdx.x; i < nel; i += blockDim.x) sh_elp[0][i] = elp[i];
    }
    __syncthreads();

    // One barrier per frame.  During frame n (between barrier n and barrier n+1):
    //   pushers  start the LDS read of h[n] (parity n), finish the pushes of h[n-1] meanwhile, clear ring slot n, do the
    //            one push of h[n] that slot n+2 needs and hand A'[n+2][c] = max over sources <= n (slot n+2, owner
    //            lane) to LDS (parity n+2 = n); the other R-1 pushes of h[n] follow after the barrier;
    //   chain    reads A'[n+1][.] (handed over during frame n-1: sources <= n-1), adds the k = 1 term itself
    //            (h[n] + len[1], both in its registers), and turns it into gamma[n+1], beta[n+1], h[n+1] (LDS, parity n+1).
    // So the K-proportional pushes of frame n and the latency-bound serial chain of frame n+1 run side by side.
    if (w == 0) {
        // ============================================================================ chain wave
        // The serial chain is the critical path of every frame; its SIMD partner is a pusher wave with an endless
        // supply of independent fp64 work, so the chain wave takes priority in the issue arbitration.
        __builtin_amdgcn_s_setprio(3);
        const int to = lane & 31, half = lane >> 5;
        const bool live = to < C;
        double tr[HF];                                    // trans[to][half*HF + i]
#pragma unroll
        for (int i = 0; i < HF; ++i) {
            const int f = half * HF + i;
            tr[i] = (live && f < C) ? trans[(size_t)to * cm + f] : SMM_NEG_INF;
        }
        const double len1 = (live && kp >= 2) ? len[(size_t)cm + to] : SMM_NEG_INF;   // len[1][to]
        double cum = 0.0;
        double hcur = live ? init[to] : 0.0;              // h[n][to]
        double enext = live ? sh_elp[0][to] : 0.0;          // elp[n][to], read one frame ahead
        // own ring (CP): state cx, same code as a pusher with one state
        constexpr int cx = NP * SPW;
        const bool has1 = CP && cx < C;
        double A1[CP ? R : 1], L1[CP ? R : 1], hs1 = 0.0;
        if (CP) {
#pragma unroll
            for (int r = 0; r < (CP ? R : 1); ++r) {
                const int p = lane * R + r;
                A1[r] = SMM_NEG_INF;
                L1[r] = (has1 && p >= 1 && p <= kp - 1) ? len[(size_t)p * cm + cx] : SMM_NEG_INF;
            }
        }
        SMM_PROF_DECL;
        for (int n0 = 0; n0 < T; n0 += (CP ? R : 1)) {
#pragma unroll
            for (int u = 0; u < (CP ? R : 1); ++u) {
                const int n = n0 + u;
                if (n >= T) break;
                const double hn1 = (CP && has1) ? smm_readlane(hcur, cx) : 0.0;      // h[n][cx] before it is replaced
                const double ecurv = enext;
                const int n1 = n + 1;
                enext = (live && n1 < T) ? sh_elp[(n1 >> 6) & 1][(n1 & 63) * cm + to] : 0.0;
                const int nn = n + 1;
                SMM_STAMP(q0);
                const double acc = smm_fmax(sh_apart[nn & 1][to], hcur + len1);      // A[nn][to]
                cum = cum + ecurv;
                const double gm = cum + acc;
                if (half == 0 && live) {
                    sh_gam[to] = gm;
                    hgam[(size_t)nn * cm + to] = gm;
                    hcum[(size_t)nn * cm + to] = cum;
                }
                SMM_STAMP(q1);
                if (nn < T) {
                    // beta[to] = max_from (gamma[from] + trans[to][from]); this half folds sources half*HF ..
                    const double2 *gp = reinterpret_cast<const double2 *>(&sh_gam[half * HF]);
                    double bq[4] = {SMM_NEG_INF, SMM_NEG_INF, SMM_NEG_INF, SMM_NEG_INF};   // 4 independent max chains
#pragma unroll
                    for (int q = 0; q < HF / 2; ++q) {
                        const double2 gv = gp[q];
                        bq[(2 * q) & 3] = smm_fmax(bq[(2 * q) & 3], gv.x + tr[2 * q]);
                        bq[(2 * q + 1) & 3] = smm_fmax(bq[(2 * q + 1) & 3], gv.y + tr[2 * q + 1]);
                        // with its own ring the chain wave is register-bound: do not let the scheduler hoist all LDS reads
                        if (CP && (q & 1)) __builtin_amdgcn_sched_barrier(0);
                    }
                    const double beta = smm_max_halves(smm_fmax(smm_fmax(bq[0], bq[1]), smm_fmax(bq[2], bq[3])));
                    hcur = beta - cum;
                    if (half == 0 && live) {
                        sh_h[nn & 1][to] = hcur;
                        hh[(size_t)nn * cm + to] = hcur;
                    }
                }
                if constexpr (CP) {
                    if (has1) smm_ring_frame<R>(A1, L1, hs1, hn1, n, u, lane, &sh_apart[n & 1][cx]);
                }
                SMM_STAMP(q2);
                __syncthreads();                                           // barrier n+1
                SMM_STAMP(q3);
                SMM_PROF_FRAME;
            }
        }
        SMM_PROF_OUT(8);
        SMM_PROF_WAVE(0);
    } else {
        // ============================================================================ pusher waves
        // pusher rank: the wave that shares a SIMD with the chain wave (wave 4 when there are 8) goes last, so that it
        // owns the fewest states
        int rank = w - 1;
        if (NW == 8) rank = (w == 4) ? NP - 1 : ((w == NW - 1) ? 3 : w - 1);
        // (12 waves: every pusher owns SPW states, nothing to rebalance)
        const int nv_all = (C - rank + NP - 1) / NP;                       // states rank, rank+NP, ...
        const int nv = nv_all < 0 ? 0 : (nv_all > SPW ? SPW : nv_all);
        double A[SPW][R], L[SPW][R], hs[SPW];
#pragma unroll
        for (int j = 0; j < SPW; ++j) {
            const int c = j * NP + rank;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int p = lane * R + r;
                A[j][r] = SMM_NEG_INF;
                L[j][r] = (j < nv && p >= 1 && p <= kp - 1) ? len[(size_t)p * cm + c] : SMM_NEG_INF;
            }
            hs[j] = 0.0;
        }
        constexpr int QMAX = (SMM_MAX_STATES_DEV + NP - 1) / NP;   // chunk elements per pusher thread
        double pre[QMAX];
#pragma unroll
        for (int q = 0; q < QMAX; ++q) pre[q] = 0.0;
        const int pidx = (w - 1) * 64 + lane;
        SMM_PROF_DECL;
        for (int n0 = 0; n0 < T; n0 += R) {
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int n = n0 + u;                            // source step; n % R == u
                if (n >= T) break;
                if (u == 0 && (n & 31) == 0) {
                    // next 64-frame chunk of elp: global -> registers at the start of a chunk, -> LDS half a chunk later
                    const int nbase = (n & ~63) + 64;
                    const int nel = (T - nbase < 64 ? T - nbase : 64) * cm;        // <= 0 when there is no next chunk
                    if ((n & 63) == 0) {
#pragma unroll
                        for (int q = 0; q < QMAX; ++q) {
                            const int e = pidx + q * NP * 64;
                            if (e < nel) pre[q] = elp[(size_t)nbase * cm + e];
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < QMAX; ++q) {
                            const int e = pidx + q * NP * 64;
                            if (e < nel) sh_elp[(nbase >> 6) & 1][e] = pre[q];
                        }
                    }
                }
                SMM_STAMP(q0);
                double hn[SPW];
#pragma unroll
                for (int j = 0; j < SPW; ++j) {
                    if (j >= nv) break;
                    hn[j] = sh_h[n & 1][j * NP + rank];          // h[n][c]  (LDS broadcast read, consumed below)
                }
                // (smm_ring_frame overlaps that read with the R-1 left-over pushes of source n-1)
                SMM_STAMP(q1);
#pragma unroll
                for (int j = 0; j < SPW; ++j) {
                    if (j >= nv) break;
                    smm_ring_frame<R>(A[j], L[j], hs[j], hn[j], n, u, lane, &sh_apart[n & 1][j * NP + rank]);
                }
                SMM_STAMP(q2);
                __syncthreads();                                 // barrier n+1
                SMM_STAMP(q3);
                SMM_PROF_FRAME;
            }
        }
        if (w == 1) { SMM_PROF_OUT(24); }
        SMM_PROF_WAVE(w);
    }

    // -------------------------------------------------------------------------------- last position
    // sh_gam holds gamma[T][.]; candidates fin[to], to = 0..C (C = EOS): first maximal entry wins.
    __syncthreads();
    if (w == 0) {
        double f = SMM_NEG_INF;
        if (lane <= C) {
            for (int c = 0; c < C; ++c) {
                const double wgt = (lane == C) ? (endpen ? endpen[c] : 0.0) : trans[(size_t)lane * cm + c];
                f = fmax(f, sh_gam[c] + wgt);
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
            const double cn = hcum[(size_t)n * cm + f];
            const double wf = smm_readlane(wgt, f);
            const double *hcol = hh + f;
            const int lim = (kmax < k - 1) ? kmax : k - 1;      // an equal k with a larger state loses
            for (int kb = 0; kb < lim; kb += NW * 64) {
                const int kk = kb + w * 64 + lane + 1;
                bool hit = false;
                if (kk <= lim) hit = ((cn + (hcol[(size_t)(n - kk) * cm] + len[(size_t)kk * cm + f])) + wf) == best;
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
#include <cstdlib>
#include "../../include/smmdp.h"
#include "smm_launch.h"

// Configuration: 1 chain wave + NP pusher waves x SPW states, NP * SPW >= states.  VALU code can only address the
// 256 architected VGPRs (the other half of the unified file are AGPRs) and a pusher needs ~4*R*SPW + 40 of them, so
// R*SPW <= 52.  Fewer, fatter pushers are preferred (fewer waves per barrier), but at least one pusher per SIMD.
template <int R, int SPW, int NW>
static int launch_if(const SmmDpArgs &a, int spw, int nw, int c_need, hipStream_t stream)
{
    if (spw != SPW || nw != NW) return 0;
    if (c_need <= 16) hipLaunchKernelGGL((smm_viterbi_kernel<R, SPW, NW, 8, 0>), dim3(a.b), dim3(NW * 64), 0, stream, a);
    else hipLaunchKernelGGL((smm_viterbi_kernel<R, SPW, NW, 16, 0>), dim3(a.b), dim3(NW * 64), 0, stream, a);
    return 1;
}

template <int R>
static int launch_r(const SmmDpArgs &a, int c_need, hipStream_t stream)
{
    // 8 waves: 256 VGPRs per wave -> R*SPW <= 52;  16 waves: 128 VGPRs per wave -> R*SPW <= 22.
    // K > 512 with more than 21 states fits neither (the rings of 22+ states x 1024 slots x fp64 (A, len) exceed the
    // CU's register file); that shape still runs -- 16 waves x 2 states, spilling to scratch -- but slowly.
    constexpr int SPW8 = (52 / R) > 5 ? 5 : (52 / R);
    constexpr int SPW16 = (22 / R) > 3 ? 3 : ((22 / R) < 1 ? 1 : (22 / R));
    int nw = 8;
    if (const char *e = std::getenv("SMM_NW")) nw = std::atoi(e);   // tuning aid: minimum wave count
    if (nw != 4 && nw != 8 && nw != 16) nw = 8;
    if (nw == 8 && (c_need + 6) / 7 > SPW8) nw = 16;
    (void)SPW16;
    if constexpr (R == 16) {
        // 22..23 states at K > 512: 12 waves (170 VGPRs each) = 11 pushers x 2 states + the chain wave's own ring
        if (nw == 16 && c_need <= 23) {
            hipLaunchKernelGGL((smm_viterbi_kernel<16, 2, 12, 16, 1>), dim3(a.b), dim3(12 * 64), 0, stream, a);
            return SMM_OK;
        }
    }
    const int spw = (c_need + nw - 2) / (nw - 1);
    int hit = 0;
    if constexpr (R <= 4) {
        hit = launch_if<R, 1, 4>(a, spw, nw, c_need, stream) || launch_if<R, 2, 4>(a, spw, nw, c_need, stream) ||
              launch_if<R, 3, 4>(a, spw, nw, c_need, stream) || launch_if<R, 4, 4>(a, spw, nw, c_need, stream) ||
              launch_if<R, 6, 4>(a, spw, nw, c_need, stream) || launch_if<R, 8, 4>(a, spw, nw, c_need, stream) ||
              launch_if<R, 1, 8>(a, spw, nw, c_need, stream) || launch_if<R, 2, 8>(a, spw, nw, c_need, stream) ||
              launch_if<R, 3, 8>(a, spw, nw, c_need, stream) || launch_if<R, 4, 8>(a, spw, nw, c_need, stream) ||
              launch_if<R, 5, 8>(a, spw, nw, c_need, stream) ||
              launch_if<R, 1, 16>(a, spw, nw, c_need, stream) || launch_if<R, 2, 16>(a, spw, nw, c_need, stream) ||
              launch_if<R, 3, 16>(a, spw, nw, c_need, stream);
    } else if constexpr (R == 8) {
        hit = launch_if<R, 1, 8>(a, spw, nw, c_need, stream) || launch_if<R, 2, 8>(a, spw, nw, c_need, stream) ||
              launch_if<R, 3, 8>(a, spw, nw, c_need, stream) || launch_if<R, 4, 8>(a, spw, nw, c_need, stream) ||
              launch_if<R, 5, 8>(a, spw, nw, c_need, stream) ||
              launch_if<R, 1, 16>(a, spw, nw, c_need, stream) || launch_if<R, 2, 16>(a, spw, nw, c_need, stream) ||
              launch_if<R, 3, 16>(a, spw, nw, c_need, stream);
    } else if constexpr (R == 16) {
        hit = launch_if<R, 1, 8>(a, spw, nw, c_need, stream) || launch_if<R, 2, 8>(a, spw, nw, c_need, stream) ||
              launch_if<R, 3, 8>(a, spw, nw, c_need, stream) ||
              launch_if<R, 1, 16>(a, spw, nw, c_need, stream) || launch_if<R, 2, 16>(a, spw, nw, c_need, stream) ||
              launch_if<R, 3, 16>(a, spw, nw, c_need, stream);
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
