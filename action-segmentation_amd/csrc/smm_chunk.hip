// smm_chunk.hip -- time-split Viterbi decode of long videos (round 5): the kernels either side of the forward pass.
//
// Replaces nothing new in the reference: it is the same decode (semimarkov_modules.py:677-679 per video) with the forward
// pass of ONE video spread over several workgroups.  A video is one serial chain of T positions (smm_viterbi.hip) -- 1.6 ms
// for 10 000 frames on one of 256 CUs, whatever else the GPU is doing.  The recursion forgets its start within a segment
// or two (oracle/prune_probe.c: smm_conv_probe, profiles/round5_rank_convergence.txt: <= 407 positions on cfg1 / cfg3's
// longest videos), so the time axis can be cut into UNITS that run side by side, each warmed up on the positions in front
// of its own part -- and the result is still the unsplit decode's, bit for bit, because nothing is taken on trust:
//
//   units     unit 0 = positions (0, e_0] as in the unsplit decode.  Unit j >= 1 = positions (a_j, e_j], a_j = r_j - OV,
//             its OWN part is (r_j, e_j] with r_j = e_{j-1}; OV = warm-up (512) + kp - 1.  It starts from the guess
//             "h[a_j][c] = 0 for every state, nothing older" and from the SERIAL cumE[a_j][.] (smm_cum_anchor_kernel: the
//             unsplit decode's additions in the unsplit decode's order), so its cumE rows are the unsplit decode's bits.
//   certify   (smm_chunk_stitch_kernel, per cut) over the kp positions in front of r_j -- every source a target of the
//             unit's own part can reach -- the unit's h and the previous unit's h must differ by ONE constant (to 2^-32
//             of their magnitude), state by state and position by position; the previous unit is certified itself (unit 0
//             is exact), so by induction every value of the unit's own part is the unsplit value + a constant +- noise,
//             the noise being the rounding of <= 4 additions per position on values of that magnitude.
//   decide    the back-trace walks the units' histories from T down, re-evaluating the forward pass's expressions as
//             the unsplit kernel does, and asks MORE of every decision: the winner (state, then length) must beat every
//             other candidate by tau = 2^-30 of the magnitude in play -- 250 x the noise bound.  Then the unsplit
//             decode, whose values differ from these by a constant and less than tau / 2, decides the same, tie order
//             included (there is no tie).  The path is the unsplit decode's path.
//   score     the best score is re-evaluated along that path in the unsplit decode's association (cumE rows are its
//             bits; h along the path is add, add, add, sub per segment): the unsplit decode's number.
//   ties      two lengths within tau at the end of a run of ONE class that is decoded as two spans -- the (k2, k1) / (k1, k2)
//             orders of the same run, equal up to rounding -- are resolved, not repaired: both orders are verified, and the
//             score pass, which has the exact h in front of the run, evaluates both rounded candidates and takes the
//             one-piece decode's choice.
//   repair    a cut that does not certify, any other decision inside tau (an exact tie on an integer lattice, a boundary
//             that rounding decides), a NaN: the video's word in `redo` is set and the launch that follows decodes it again
//             in one piece with the ordinary kernel (one workgroup per split video, all but the flagged ones return at
//             once).  Correctness never rests on the split; only the time does.
//
// HBM traffic: a unit reads and writes what the unsplit video would for its positions (32C + 8 B per position), OV of
// them twice; the stitch reads (kp + 1) * C * 16 B per cut and what the back-trace reads anyway.
#include <algorithm>
#include "smm_device.h"
#include "../../include/smmdp.h"
#include "smm_launch.h"

// ------------------------------------------------------------------------------------------------ serial prefix sums
// cumE[a_j][c] for every unit j >= 1 of one split video: cum = 0; cum = cum + elp[n][c], n = 0, 1, ... in THIS order (the
// unsplit kernel's mover wave and chain wave do exactly these additions).  One workgroup per split video: wave 0 adds, one
// lane per state -- one dependent chain of fp64 additions, ~5 cycles each, the kernel's floor -- and waves 1..7 feed it: rounds
// of 256 rows, fetched THREE rounds ahead into registers (a round trip to HBM is ~2 us, a round of additions 0.6), written
// transposed into one of TWO LDS tiles while wave 0 adds the other up; one barrier per round.  (The first version of this
// kernel had one tile of 512 rows and every wave both fetching and waiting -- load, barrier, add, barrier, store per round:
// cfg1's 7 400 rows took 81 us; this one 60.  Ablation builds, same box: without the additions 37 us, without the loads 55,
// six rounds in flight instead of three 60: the two roles' times ADD UP although they run side by side -- the producers'
// transposed LDS writes and the adder's LDS reads share the CU's one LDS pipeline -- and the HBM latency is covered.)
#define SMM_ANCH_WAVES 8
#ifndef SMM_ANCH_ROWS
#define SMM_ANCH_ROWS 256                         // rows per round
#endif
#ifndef SMM_ANCH_PD
#define SMM_ANCH_PD 3                             // rounds in flight in the producers' registers
#endif
// NE: elements per producer lane and round, >= 256 c_max / 448 (10 / 14 / 19 for c_max <= 16 / 24 / 32); PD: rounds in flight in the
// producers' registers
template <int NE, int PD>
__global__ void __launch_bounds__(SMM_ANCH_WAVES * 64)
smm_cum_anchor_kernel(SmmDpArgs a, const SmmChunkVideo *cvs, double *anchors)
{
    extern __shared__ __attribute__((aligned(16))) double tile[];       // [2][cm][RB + 2], TRANSPOSED: a state's rows of a round are contiguous
    const SmmChunkVideo cv = cvs[blockIdx.x];
    const SmmVideo pv = a.videos[cv.vid];
    const int cm = a.c_max, C = a.n_states[pv.group];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const double *elp = a.elp + (size_t)pv.frame_off * cm;
    const int n_end = a.videos[cv.first_unit + cv.n_chunks - 1].pad >> 2;    // the last unit's first position: nothing is needed beyond
    constexpr int RB = SMM_ANCH_ROWS;
    constexpr int TS = RB + 2;                                                // doubles per state of a tile (16-byte rows, odd multiple of 16 B)
    constexpr int NP = (SMM_ANCH_WAVES - 1) * 64;                             // producer lanes
    const int n_rounds = (n_end + RB - 1) / RB;
    const size_t tile_doubles = (size_t)cm * TS;
    const int pl = (int)threadIdx.x - 64;                                     // producer lane id (waves 1..7)
    const int64_t e_max = (int64_t)n_end * cm - 1;
    double reg[PD][NE];
    auto fetch = [&](int round, double (&dst)[NE]) {
        const int64_t e0 = (int64_t)round * RB * cm;
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            const int e = pl + NP * q;
            const int64_t eg = e0 + e;
            dst[q] = elp[(e < RB * cm && eg <= e_max) ? eg : e_max];          // (clamped: rows beyond n_end are never added)
        }
    };
    auto park = [&](int round, const double (&src)[NE]) {
        double *t = tile + (size_t)(round & 1) * tile_doubles;
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            const int e = pl + NP * q;
            if (e < RB * cm) t[(size_t)(e % cm) * TS + e / cm] = src[q];
        }
    };
    double cum = 0.0;
    int next_unit = 1;
    int next_at = a.videos[cv.first_unit + 1].pad >> 2;
    // Two roles, two loops, the same n_rounds + 1 barriers in both (every wave is in one role for good: the producers' registers
    // and the adder's do not share a live range).  Iteration i: the producers park round i (fetched PD iterations ago) and fetch
    // round i + PD into the registers it leaves; wave 0 adds round i - 1 up from the other tile.  A tile is written again two
    // iterations after it was read.
    if (w > 0) {
#pragma unroll
        for (int q = 0; q < PD; ++q)
            if (q < n_rounds) fetch(q, reg[q]);
        for (int i0 = 0; i0 <= n_rounds; i0 += PD) {
            // (unrolled over the register sets: a set chosen by a run-time index would live in scratch)
#pragma unroll
            for (int q = 0; q < PD; ++q) {
                const int i = i0 + q;
                if (i > n_rounds) break;
                if (i < n_rounds) { park(i, reg[q]); if (i + PD < n_rounds) fetch(i + PD, reg[q]); }
                __syncthreads();
            }
        }
    } else {
        __builtin_amdgcn_s_setprio(3);              // (the adder's chain is the kernel's time: in front of the producer wave on its SIMD; 60.2 -> 57.2 us)
        for (int i = 0; i <= n_rounds; ++i) {
            if (i >= 1 && lane < C) {
                const int row0 = (i - 1) * RB;
                const int rows = (n_end - row0 < RB) ? n_end - row0 : RB;
                const double *col = tile + (size_t)((i - 1) & 1) * tile_doubles + (size_t)lane * TS;
                const double2 *col2 = reinterpret_cast<const double2 *>(col);
                int r = 0;
                while (r < rows) {
                    // cumE[n] = sum of the rows before n: the anchor of a unit that starts at n is taken BEFORE row n is added
                    if (row0 + r == next_at) {
                        anchors[(size_t)(cv.first_unit - a.b_videos + next_unit) * cm + lane] = cum;
                        ++next_unit;
                        next_at = next_unit < cv.n_chunks ? (a.videos[cv.first_unit + next_unit].pad >> 2) : 0x7fffffff;
                    }
                    // ... then straight on to the next anchor or the end of the round.  The additions are one dependent chain; the
                    // LDS reads run ahead of it, sixteen 16-byte reads (32 rows) in flight
                    const int stop = (next_at - row0 < rows) ? next_at - row0 : rows;
                    if (r & 1) { cum = cum + col[r]; ++r; if (r >= stop) continue; }
                    // (batches of 32 rows in TWO register sets: the reads of the next batch are issued in front of this batch's
                    // additions -- with one set, each read waits for the addition that frees its register and the chain then waits
                    // an LDS round trip per batch)
                    const int nb = (stop - r) >> 5;
                    if (nb >= 1) {
                        double2 va[16], vb[16];
                        const double2 *p = col2 + (r >> 1);
    #pragma unroll
                        for (int u = 0; u < 16; ++u) va[u] = p[u];
                        for (int b = 0; b < nb; b += 2) {
                            if (b + 1 < nb) {
    #pragma unroll
                                for (int u = 0; u < 16; ++u) vb[u] = p[16 * (b + 1) + u];
                            }
                            __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                            for (int u = 0; u < 16; ++u) { cum = cum + va[u].x; cum = cum + va[u].y; }
                            if (b + 1 < nb) {
                                if (b + 2 < nb) {
    #pragma unroll
                                    for (int u = 0; u < 16; ++u) va[u] = p[16 * (b + 2) + u];
                                }
                                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                                for (int u = 0; u < 16; ++u) { cum = cum + vb[u].x; cum = cum + vb[u].y; }
                            }
                        }
                        r += 32 * nb;
                    }
                    for (; r + 2 <= stop; r += 2) {
                        const double2 v = col2[r >> 1];
                        cum = cum + v.x;
                        cum = cum + v.y;
                    }
                    if (r < stop) { cum = cum + col[r]; ++r; }
                }
            }
            __syncthreads();
        }
    }
    // (a unit that starts exactly at n_end: the loop above ends before row n_end)
    if (w == 0 && lane < C && next_unit < cv.n_chunks && next_at == n_end)
        anchors[(size_t)(cv.first_unit - a.b_videos + next_unit) * cm + lane] = cum;
}

// ------------------------------------------------------------------------------------------------ stitch
#define SMM_STITCH_THREADS 1024
#define SMM_STITCH_MAXSEG 1024

__device__ __forceinline__ bool smm_finite_bits(double x)
{
    int hi = __double2hiint(x);
    asm volatile("" : "+v"(hi));
    return (hi & 0x7ff00000) != 0x7ff00000;
}

// one workgroup per split video (see the head of this file).  Error block word 4 counts the split videos, word 5 the ones
// handed to the repair launch, word 6 ORs the reasons, word 7 counts the one-class-run ties resolved (ops.error_words).
__global__ void __launch_bounds__(SMM_STITCH_THREADS)
smm_chunk_stitch_kernel(SmmDpArgs a, const SmmChunkVideo *cvs, int32_t *redo)
{
    const SmmChunkVideo cv = cvs[blockIdx.x];
    const int vid = cv.vid;
    const SmmVideo pv = a.videos[vid];
    const int T = pv.T, g = pv.group, C = a.n_states[g], cm = a.c_max, kp = pv.kp;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const double *trans = a.trans + (size_t)g * cm * cm;
    const double *init = a.init + (size_t)g * cm;
    const double *len = a.len + (size_t)g * a.k_rows * cm;
    const double *endpen = a.endpen ? a.endpen + (size_t)vid * cm : nullptr;
    const int64_t *cmap = a.class_map ? a.class_map + (size_t)g * (cm + 1) : nullptr;
    int64_t *spans = a.spans ? a.spans + (size_t)vid * (a.t_max + 1) : nullptr;
    int64_t *labels = a.labels ? a.labels + pv.frame_off : nullptr;
    const SmmVideo *units = a.videos + cv.first_unit;
    const int nu = cv.n_chunks;

    __shared__ int sh_bad;
    __shared__ double sh_dref[SMM_CHUNK_MAX_UNITS];
    __shared__ unsigned sh_kmin[2], sh_near[2], sh_nlo[2], sh_nhi[2];   // (two sets, used in turn: see decide)
    __shared__ int sh_tie_s2[SMM_STITCH_MAXSEG];
    __shared__ double sh_tie_c2[SMM_STITCH_MAXSEG];
    __shared__ int sh_guess[SMM_MAX_STATES_DEV + 1];
    __shared__ int sh_ua[SMM_CHUNK_MAX_UNITS], sh_ut[SMM_CHUNK_MAX_UNITS];
    __shared__ long long sh_uoff[SMM_CHUNK_MAX_UNITS];
    __shared__ int sh_seg_s[SMM_STITCH_MAXSEG];
    __shared__ int sh_seg_c[SMM_STITCH_MAXSEG];
    __shared__ double sh_cn[SMM_STITCH_MAXSEG], sh_ct[SMM_STITCH_MAXSEG];     // cumE[n][c], cumE[n][to] at the segments' ends
    __shared__ double sh_ln[SMM_STITCH_MAXSEG], sh_tr[SMM_STITCH_MAXSEG];   // ... their length scores and the transitions behind them
    __shared__ double sh_h0;

    if (threadIdx.x == 0) { sh_bad = 0; sh_kmin[0] = sh_kmin[1] = 0xffffffffu; sh_near[0] = sh_near[1] = 0; sh_nlo[0] = sh_nlo[1] = 0xffffffffu; sh_nhi[0] = sh_nhi[1] = 0; }
    // (sh_bad: WHY the video goes to the repair launch -- 1 a cut does not certify, 2 the closing step, 4 two states within tau,
    // 8 two lengths within tau / none attains the maximum, 16 NaN or too many segments; OR-ed into error block word 6)
    if (spans)
        for (int i = threadIdx.x; i <= a.t_max; i += blockDim.x) spans[i] = -1;
    __syncthreads();

    // unit j: positions a_j .. a_j + T_j, history rows by LOCAL position n - a_j (the units' table in LDS: the back-trace asks
    // for it once per segment)
    for (int j = threadIdx.x; j < nu; j += blockDim.x) {
        sh_ua[j] = units[j].pad >> 2;
        sh_ut[j] = units[j].T;
        sh_uoff[j] = units[j].hist_off;
    }
    __syncthreads();
    auto u_a = [&](int j) { return sh_ua[j]; };
    auto u_t = [&](int j) { return sh_ut[j]; };
    auto u_cum = [&](int j) { return a.hist + sh_uoff[j]; };
    auto u_h = [&](int j) { return a.hist + sh_uoff[j] + (size_t)C * (sh_ut[j] + 1); };
    auto u_gam = [&](int j) { return a.hist + sh_uoff[j] + (size_t)2 * C * (sh_ut[j] + 1); };

    // ---------------------------------------------------------------------------------------------- certify the cuts
    // Two passes with ONE barrier between them (the cuts are independent of each other; one after the other, each behind two
    // barriers and two dependent trips to the histories, they took ~3 us a cut).  Pass 1: wave w takes the cuts w + 1, w + 1 + waves,
    // ...: the reference difference of a cut = h(unit j) - h(unit j - 1) at r for the state that leads h there in unit j.
    const int n_waves = (int)blockDim.x >> 6;
    for (int j = 1 + w; j < nu; j += n_waves) {
        const int r = u_a(j) + cv.ov;                             // unit j's own part begins behind r = the end of unit j-1
        const int a1 = u_a(j), a0 = u_a(j - 1), t1 = u_t(j), t0 = u_t(j - 1);
        const double *h1 = u_h(j), *h0 = u_h(j - 1);
        if (r != a0 + t0 || r - (kp - 1) - a1 < 1) { if (lane == 0) sh_bad = 16; continue; }   // (the host's layout: never)
        double v = (lane < C) ? h1[(size_t)lane * (t1 + 1) + (r - a1)] : SMM_NEG_INF;
        const double v0 = (lane < C) ? h0[(size_t)lane * (t0 + 1) + (r - a0)] : SMM_NEG_INF;
        if (!(smm_finite_bits(v) && smm_finite_bits(v0))) v = SMM_NEG_INF;
        double m = v;
        int mc = lane;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double m2 = __shfl_xor(m, off);
            const int c2 = __shfl_xor(mc, off);
            if (m2 > m || (m2 == m && c2 < mc)) { m = m2; mc = c2; }
        }
        const double d = smm_readlane(v, mc) - smm_readlane(v0, mc);
        if (lane == 0) { sh_dref[j] = d; if (!smm_finite_bits(d) || m == SMM_NEG_INF) sh_bad = 1; }
    }
    __syncthreads();
    // Pass 2: every thread checks ITS position of every cut (r - (kp - 1) .. r: one per thread, kp <= the workgroup), the states in
    // batches of eight with all sixteen loads of a batch in flight (h is state-major: consecutive threads read consecutive
    // doubles); nothing waits between the cuts.
    if (!sh_bad) {
        int bad = 0;
        for (int j = 1; j < nu; ++j) {
            const int r = u_a(j) + cv.ov;
            const int a1 = u_a(j), a0 = u_a(j - 1), t1 = u_t(j), t0 = u_t(j - 1);
            const double *h1 = u_h(j), *h0 = u_h(j - 1);
            const double *c1 = u_cum(j), *c0 = u_cum(j - 1);
            const double dref = sh_dref[j];
            const double cscale = fabs(c1[(size_t)(r - a1) * C]);     // the magnitude of the prefix sums there
            const int s = r - (kp - 1) + (int)threadIdx.x;
            if ((int)threadIdx.x < kp) {
                for (int cb = 0; cb < C; cb += 8) {
                    double x1[8], x0[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int c = cb + q < C ? cb + q : C - 1;
                        x1[q] = h1[(size_t)c * (t1 + 1) + (s - a1)];
                        x0[q] = h0[(size_t)c * (t0 + 1) + (s - a0)];
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const bool f1 = smm_finite_bits(x1[q]), f0 = smm_finite_bits(x0[q]);
                        if (f1 != f0) { bad = 1; continue; }              // -inf on one side only (or a NaN)
                        if (!f1) { if (smm_nan_bits(x1[q]) || smm_nan_bits(x0[q]) || x1[q] != x0[q]) bad = 1; continue; }
                        const double tol = 0x1p-32 * (fabs(x1[q]) + fabs(x0[q]) + cscale + 1.0);
                        if (!(fabs((x1[q] - x0[q]) - dref) <= tol)) bad = 1;
                    }
                }
            }
            // the prefix sums of the two units are the same additions: the same bits
            for (int e = threadIdx.x; e < C; e += blockDim.x)
                if (__double_as_longlong(c1[(size_t)(r - a1) * C + e]) != __double_as_longlong(c0[(size_t)(r - a0) * C + e])) bad = 1;
        }
        if (bad) sh_bad = 1;
    }
    __syncthreads();

    // ---------------------------------------------------------------------------------------------- closing step (n = T)
    // as in smm_viterbi_kernel: candidates fin[to], to = 0..C (C = EOS); here the winner must win by tau
    int n = T, to = 0, nseg = 0;
    int ju = nu - 1;                                              // the unit whose own part holds n
    if (!sh_bad) {
        const double *gT = u_gam(ju) + (size_t)(T - u_a(ju)) * C;
        double f = SMM_NEG_INF;
        if (lane <= C) {
            for (int c = 0; c < C; ++c) {
                const double wgt = (lane == C) ? (endpen ? endpen[c] : 0.0) : trans[(size_t)lane * cm + c];
                f = fmax(f, gT[c] + wgt);
            }
            if (lane < C) f = f + SMM_BIG_NEG;
        }
        double m = f;
        int mc = (lane <= C) ? lane : 0x7fffffff;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double m2 = __shfl_xor(m, off);
            const int c2 = __shfl_xor(mc, off);
            if (m2 > m || (m2 == m && c2 < mc)) { m = m2; mc = c2; }
        }
        const double tau = 0x1p-30 * (fabs(m) + 1.0);
        const bool near = lane <= C && lane != mc && !(f < m - tau);
        // (a winner other than EOS -- every end penalised -- goes to the repair launch as well: the score below assumes EOS)
        if ((__ballot(near) != 0 || !smm_finite_bits(m) || mc != C) && threadIdx.x == 0) sh_bad = 2;
        to = mc;
    }
    __syncthreads();

    // ---------------------------------------------------------------------------------------------- back-trace
    // (n, to): the segment that ends at n in front of label `to`.  One thread per candidate length (kp - 1 <= 1023 of
    // them); lane c of every wave holds state c's row entries.  As in smm_viterbi_kernel a segment is ONE trip to the
    // history when the predecessor state is the one seen the last time (sh_guess; at first the arg-max of the transition
    // row): that state's column travels with the row, and the next segment's trip is issued in front of this one's stores.
    if (threadIdx.x <= (unsigned)C) {
        const int t2 = threadIdx.x;
        int bi = C - 1;
        if (t2 < C) {
            double bv = SMM_NEG_INF;
            bi = 0;
            for (int c2 = 0; c2 < C; ++c2) {
                const double v2 = trans[(size_t)t2 * cm + c2];
                if (v2 > bv) { bv = v2; bi = c2; }
            }
        } else if (endpen) {
            for (int c2 = 0; c2 < C; ++c2)
                if (endpen[c2] == 0.0) bi = c2;
        }
        sh_guess[t2] = bi;
    }
    __syncthreads();
    // BAND-mode launches carry the state-major length table ([c][k + 1] = len[k][c]: a state's lengths contiguous); the ring
    // kernels' launches (span limits up to 512) read the [k][c] table, one cache line per candidate
    const bool band = (a.flags & 128) != 0;
    const double *lcol0 = band ? a.len_t + (size_t)g * cm * SMM_BAND_ROW + 1 : nullptr;
    auto len_of = [&](int c_, int k_) { return band ? lcol0[(size_t)c_ * SMM_BAND_ROW + k_] : len[(size_t)k_ * cm + c_]; };
    int fg = 0, kmax = 0, aj = 0, tj = 0;
    const double *hc = nullptr, *hh = nullptr;
    double g0 = SMM_NEG_INF, cnl = 0.0, wgt = 0.0, sp_h = 0.0, sp_l = 0.0;
    auto trip = [&](int n_, int to_) {
        // the unit whose own part holds n_: (a_j + OV, a_j + T_j], unit 0: (0, T_0]  (the walk goes down; the look-aside of a
        // tie, below, may have left ju too low)
        while (ju < nu - 1 && n_ > u_a(ju) + u_t(ju)) ++ju;
        while (ju > 0 && n_ <= u_a(ju) + cv.ov) --ju;
        aj = u_a(ju); tj = u_t(ju);
        hc = u_cum(ju); hh = u_h(ju);
        const double *hg = u_gam(ju);
        fg = sh_guess[to_];
        kmax = (kp - 1 < n_) ? kp - 1 : n_;
        if (lane < C) {
            g0 = hg[(size_t)(n_ - aj) * C + lane];
            cnl = hc[(size_t)(n_ - aj) * C + lane];
            wgt = (to_ == C) ? (endpen ? endpen[lane] : 0.0) : trans[(size_t)to_ * cm + lane];
        }
        const int kk0 = threadIdx.x + 1, kc = kk0 <= kmax ? kk0 : kmax;
        sp_h = hh[(size_t)fg * (tj + 1) + (n_ - aj - kc)];
        sp_l = len_of(fg, kc);
    };
    // One decision from what the last trip brought: the state and the length(s) within tau of the maximum.
    //   status 1  one state, ONE length: the one-piece decode decides the same
    //   status 2  one state, TWO lengths k1 < k2 within tau (one of them equals the maximum): a tie candidate, see below
    //   status 0  anything else (reason in sh_bad)
    int d_c = 0, d_k1 = 0, d_k2 = 0, d_par = 0;
    double d_cn = 0.0;
    // (the four reduction words exist twice and decisions use them in turn: a decision's words are cleared by thread 0 behind the
    // barrier of the NEXT decision -- by then every thread has read them -- and written again one decision later, behind another
    // barrier: one barrier per decision instead of three)
    auto decide = [&](int n_) -> int {
        const int par = d_par;
        d_par ^= 1;
        const double gmv = (lane < C) ? g0 + wgt : SMM_NEG_INF;
        const bool nan_row = __ballot(lane < C && smm_nan_bits(gmv)) != 0;
        const double rmax = smm_row_max16(gmv);
        const double best = fmax(smm_readlane(rmax, 0), smm_readlane(rmax, 16));
        double cmag = (lane < C) ? fabs(cnl) : 0.0;               // the magnitude of the prefix sums at n_
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) cmag = fmax(cmag, __shfl_xor(cmag, off));
        cmag = fmax(smm_readlane(cmag, 0), smm_readlane(cmag, 16));
        const double tau = 0x1p-30 * (fabs(best) + cmag + 1.0);
        // ONE state within tau of the maximum
        const unsigned long long nearm = __ballot(lane < C && !(gmv < best - tau));
        if (nan_row || __builtin_popcountll(nearm) != 1 || !smm_finite_bits(best)) {
            if (threadIdx.x == 0) sh_bad = (nan_row || !smm_finite_bits(best)) ? 16 : 4;
            __syncthreads();
            return 0;
        }
        const int c = __ffsll(nearm) - 1;
        const double cn = smm_readlane(cnl, c), wf = smm_readlane(wgt, c);
        // ... and the lengths: the candidate that EQUALS the maximum, and every candidate within tau of it
        const int kk = threadIdx.x + 1;
        if (kk <= kmax) {
            const double hv = (c == fg) ? sp_h : hh[(size_t)c * (tj + 1) + (n_ - aj - kk)];
            const double lv = (c == fg) ? sp_l : len_of(c, kk);
            const double cand = (cn + (hv + lv)) + wf;
            if (cand == best) atomicMin(&sh_kmin[par], (unsigned)kk);
            if (!(cand < best - tau)) { atomicAdd(&sh_near[par], 1u); atomicMin(&sh_nlo[par], (unsigned)kk); atomicMax(&sh_nhi[par], (unsigned)kk); }
        }
        __syncthreads();
        const unsigned kf = sh_kmin[par], nn = sh_near[par], nlo = sh_nlo[par], nhi = sh_nhi[par];
        // the OTHER set: read for the last time in front of the barrier above (by the previous decision), written next by the next
        // decision's atomics, which every thread issues behind the barrier of the walk's loop (or the tie's look-aside)
        if (threadIdx.x == 0) { sh_kmin[par ^ 1] = 0xffffffffu; sh_near[par ^ 1] = 0; sh_nlo[par ^ 1] = 0xffffffffu; sh_nhi[par ^ 1] = 0; }
        d_c = c; d_cn = cn;
        if (kf == 0xffffffffu || nn < 1 || nn > 2) { if (threadIdx.x == 0) sh_bad = 8; __syncthreads(); return 0; }
        d_k1 = (int)nlo; d_k2 = (int)nhi;
        return (int)nn;
    };
    // (lane c keeps the global id of state c: a load of cmap[c] behind a trip's loads would wait for all of them)
    const int64_t gid_l = cmap ? cmap[lane < C ? lane : C] : (int64_t)lane;
    // TIES.  A run of one class that the model prefers to decode as TWO spans (its length an outlier of the class's Poisson)
    // can be cut (k2, k1) or (k1, k2): the two orders share everything in front of the run and tie to within rounding (DESIGN
    // 2), so at the run's end two lengths sit within tau, time and again on real corpora.  That tie is RESOLVED instead of
    // repaired: both orders are verified to be what they seem -- behind the cut at n - k1 the decision must be (c, k2), behind
    // the cut at n - k2 it must be (c, k1), both clear of tau -- and when the path is re-scored from the video's start the
    // exact h in front of the run is at hand: the two rounded candidates are evaluated in the one-piece association, compared,
    // and the one-piece decode's choice (the larger; the shorter last span when equal) is taken.  Per segment i (as recorded,
    // last segment first): sh_tie_s2[i] = the other order's cut n - k2, or -1; sh_tie_c2[i] = cumE[n - k2][c].
    int expect_c = -1, expect_k = 0;                              // the decision the previous segment's tie asked for
    if (!sh_bad && n > 0) trip(n, to);
    while (!sh_bad && n > 0) {
        if (nseg >= SMM_STITCH_MAXSEG) { if (threadIdx.x == 0) sh_bad = 16; break; }
        int st = decide(n);
        if (st == 0) break;
        int c = d_c, k = d_k1;
        const double cn = d_cn;
        int tie_s2 = -1;
        double tie_c2 = 0.0;
        if (expect_c >= 0) {
            // the second half of a tie's first order: must be exactly (expect_c, expect_k), clear
            if (st != 1 || c != expect_c || k != expect_k) { if (threadIdx.x == 0) sh_bad = 8; break; }
            expect_c = -1;
        } else if (st == 2) {
            const int k1 = d_k1, k2 = d_k2, s2 = n - k2;
            if (n - k1 - k2 < 0) { if (threadIdx.x == 0) sh_bad = 8; break; }
            // look aside: the other order's cut -- behind n - k2 the decision must be (c, k1), clear
            __syncthreads();
            trip(s2, c);
            const int st2 = decide(s2);
            if (st2 != 1 || d_c != c || d_k1 != k1) { if (threadIdx.x == 0 && !sh_bad) sh_bad = 8; break; }
            tie_s2 = s2;
            tie_c2 = d_cn;                                        // cumE[s2][c]
            expect_c = c; expect_k = k2;                          // ... and behind n - k1 it must be (c, k2): the next iteration
            k = k1;                                               // tentatively the shorter last span; the score pass decides
            // (this segment's own row: cumE[n][to] below needs hc of n's unit again)
            trip(n, to);
        }
        const int s = n - k;
        if (threadIdx.x == 0) {
            sh_seg_s[nseg] = s;
            sh_seg_c[nseg] = c;
            sh_tie_s2[nseg] = tie_s2;
            sh_tie_c2[nseg] = tie_c2;
            sh_guess[to] = c;
        }
        // what the score along the path needs of position n: cumE[n][c] (this segment's state) and cumE[n][to] (the next one's)
        if (threadIdx.x == 1) { sh_cn[nseg] = cn; sh_ct[nseg] = (to < C) ? hc[(size_t)(n - aj) * C + to] : 0.0; }
        const int n0 = n;
        ++nseg;
        n = s;
        to = c;
        __syncthreads();                                          // (the guess table: written by thread 0, read by every thread's trip)
        if (n > 0) trip(n, to);                                   // the next segment's trip goes out in front of the stores
        const int64_t gid = ((int64_t)__builtin_amdgcn_readlane((int)(gid_l >> 32), c) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)gid_l, c);
        if (labels)
            for (int f = s + threadIdx.x; f < n0; f += blockDim.x) labels[f] = gid;
        if (threadIdx.x == 0 && spans) spans[s] = gid;
    }
    if (expect_c >= 0 && threadIdx.x == 0 && !sh_bad) sh_bad = 8;   // (a tie whose first order ran into the video's start)
    __syncthreads();
    const bool bad = sh_bad != 0;

    // ---------------------------------------------------------------------------------------------- the score, along the path
    // segments were recorded last to first: segment i = (sh_seg_s[i], end = the start of segment i-1 or T, state sh_seg_c[i]).
    // gamma[n][c] = cumE[n][c] + (h[s][c] + len[n-s][c]);  beta[n][to] = gamma[n][c] + trans[to][c];  h[n][to] = beta - cumE[n][to]
    // (the table entries of every segment are fetched by as many threads at once; the chain itself is one thread's ~4 nseg
    // dependent additions on LDS operands)
    if (!bad) {
        for (int i = threadIdx.x; i < nseg; i += blockDim.x) {
            const int s = sh_seg_s[i], c = sh_seg_c[i];
            const int ne = (i == 0) ? T : sh_seg_s[i - 1];
            sh_ln[i] = len[(size_t)(ne - s) * cm + c];
            sh_tr[i] = (i > 0) ? trans[(size_t)sh_seg_c[i - 1] * cm + c] : (endpen ? endpen[c] : 0.0);
        }
        if (threadIdx.x == 0) sh_h0 = init[sh_seg_c[nseg - 1]];
    }
    __syncthreads();
    if (!bad && threadIdx.x == 0) {
        double h = sh_h0;
        double gam = 0.0;
        for (int i = nseg - 1; i >= 0; --i) {
            if (i >= 1 && sh_tie_s2[i - 1] >= 0) {
                // segments i (earlier) and i - 1 (later) are one run of class c cut at s1 = n - k1; the other order cuts it at
                // s2 = n - k2.  h = the exact h in front of the run.  Both candidates of the decision at n, in the one-piece
                // decode's association; it takes the larger, and the shorter last span (k1: as recorded) when they are equal
                const double ts = sh_tr[i], wf = sh_tr[i - 1], ln2 = sh_ln[i], ln1 = sh_ln[i - 1];   // trans[c][c], w(to_n, c), len[k2], len[k1]
                const double c_s1 = sh_cn[i], c_n = sh_cn[i - 1], c_s2 = sh_tie_c2[i - 1];
                const double g1 = c_s1 + (h + ln2), h1 = (g1 + ts) - c_s1, cand1 = (c_n + (h1 + ln1)) + wf;
                const double g2 = c_s2 + (h + ln1), h2 = (g2 + ts) - c_s2, cand2 = (c_n + (h2 + ln2)) + wf;
                if (cand2 > cand1) {
                    // the other order: [p, s2) of length k1, then [s2, n) of length k2
                    const int s1 = sh_seg_s[i - 1], s2 = sh_tie_s2[i - 1];
                    sh_seg_s[i - 1] = s2;
                    sh_ln[i] = ln1; sh_ln[i - 1] = ln2;
                    sh_cn[i] = c_s2; sh_ct[i] = c_s2;
                    if (spans) { spans[s1] = -1; spans[s2] = cmap ? cmap[sh_seg_c[i]] : (int64_t)sh_seg_c[i]; }
                }
            }
            gam = sh_cn[i] + (h + sh_ln[i]);
            if (i > 0) h = (gam + sh_tr[i]) - sh_ct[i];
        }
        const double f = gam + sh_tr[0];
        if (a.best) a.best[vid] = f;
        if (spans) spans[T] = cmap ? cmap[C] : (int64_t)C;
        if (a.n_segs) a.n_segs[vid] = nseg;
    }
    if (threadIdx.x == 0) {
        redo[blockIdx.x] = bad ? 1 : 0;
        atomicAdd(a.err + 4, 1);
        if (bad) { atomicAdd(a.err + 5, 1); atomicOr(a.err + 6, sh_bad); }
        else {
            int nt = 0;
            for (int i = 0; i < nseg; ++i) nt += sh_tie_s2[i] >= 0;
            if (nt) atomicAdd(a.err + 7, nt);                     // one-class-run ties resolved
        }
    }
}

void smm_launch_cum_anchors(const SmmDpArgs &a, const SmmChunkVideo *cvs, int n_split, double *anchors, hipStream_t stream)
{
    const size_t lds = sizeof(double) * 2 * (SMM_ANCH_ROWS + 2) * (size_t)a.c_max;                  // two tiles
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    auto go = [&](auto kern, bool *raised) {
        if (!raised[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
            raised[dev] = true;
        }
        hipLaunchKernelGGL(kern, dim3(n_split), dim3(SMM_ANCH_WAVES * 64), lds, stream, a, cvs, anchors);
    };
    static bool r10[64] = {}, r14[64] = {}, r19[64] = {};
    if (a.c_max <= 16) go(smm_cum_anchor_kernel<(SMM_ANCH_ROWS * 16 + 447) / 448 + 0, SMM_ANCH_PD>, r10);
    else if (a.c_max <= 24) go(smm_cum_anchor_kernel<(SMM_ANCH_ROWS * 24 + 447) / 448, SMM_ANCH_PD>, r14);
    else go(smm_cum_anchor_kernel<(SMM_ANCH_ROWS * 32 + 447) / 448, SMM_ANCH_PD>, r19);
}

void smm_launch_chunk_stitch(const SmmDpArgs &a, const SmmChunkVideo *cvs, int n_split, int32_t *redo, hipStream_t stream)
{
    // one thread per candidate length / certified position: kp <= k_rows of them (cfg2, K = 256: four waves at the walk's barriers
    // instead of sixteen, 98.5 -> 91.8 us; a segment's time is its trip to the units' histories, not its barriers)
    int threads = (std::min(a.k_rows, SMM_STITCH_THREADS) + 63) & ~63;
    if (threads < 64) threads = 64;
    hipLaunchKernelGGL(smm_chunk_stitch_kernel, dim3(n_split), dim3(threads), 0, stream, a, cvs, redo);
}
