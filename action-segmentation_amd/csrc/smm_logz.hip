// smm_logz.hip -- log-partition (LogSemiring forward) of the factored semi-Markov model for gfx950.
//
// Replaces torch_struct SemiMarkovCRF(scores).partition (reference semimarkov_modules.py:657; the dense
// potentials of modules:416-523 are never built).  Same recurrence as smm_viterbi.hip with (logsumexp, +) for
// (max, +) -- see oracle/smm_oracle.c: smm_oracle_logz for the CPU statement:
//     A[n][c] = LSE_{k=1..min(kp-1,n)} ( h[n-k][c] + len[k][c] ),   gamma = cumE + A,
//     beta[n][to] = LSE_c ( gamma[n][c] + trans[to][c] ),           h = beta - cumE,
//     logZ = LSE over the last position's labels (EOS via endpen, real labels with the -1e9 of em+[T]).
//
// Kernel v2: the structure of the Viterbi kernel's generation 4 (smm_viterbi.hip) -- one chain wave (lane = state)
// that owns the serial part and evaluates the K0 = 2B shortest segment lengths itself, pusher waves that own the
// K-proportional work in register-resident rings, hand-over in BLOCKS of B positions with one barrier per block, HBM
// traffic (elp prefetch, history stores) moved block-wise by one pusher wave -- with a ring slot made for the log
// semiring:
//
//   A slot keeps its running sum as  S * 2^M  relative to a per-state reference:  M an INTEGER-valued fp32 exponent,
//   S an fp32 sum, the length score L of the slot as fp32 in log2 units (3 registers per slot; Viterbi: 4).  Per state
//   the wave keeps ref = ceil(running max of h * log2 e), an integer-valued double.  A block of B sources is folded into
//   a slot at once (the "online softmax" of B candidates):
//       t_i = c0_i + L_i           c0_i = (float)(h_i * log2 e - ref)  (wave-uniform, <= 0 up to rounding)
//       M'  = max(M, ceil(max_i t_i));     S' = S * 2^(M - M') + sum_i 2^(t_i - M')
//   = B adds, B/2 max3, B subs, B+1 v_exp_f32, B adds and 4 more per slot and block: ~33 issue cycles per lattice
//   cell at B = 4 against ~55 for the per-cell fp64 online log-sum-exp of kernel v1, and nothing spills at 21..28
//   states x 1024 slots.  When ref moves up by D (an integer) every slot's M moves down by D: integer arithmetic in
//   fp32, exact, so exponents never drift however long a slot lives.
//
//   Accuracy.  S' is a sum of at most 1024 terms in (0, 1] plus rescalings by powers of two (exact): relative error
//   <= 1024 * 2^-24 worst case, ~2^-21 typically (1e-6 in log space).  A candidate's exponent t_i is rounded to fp32:
//   absolute error 2^-24 |t_i| log2-units, i.e. 4e-8 x (how far the candidate lies below the state's running
//   maximum + |its length score|) nats -- 2e-5 nats for a candidate 500 nats down, which is also what kernel v1's fp32
//   copy of the length table cost.  Candidates that carry posterior mass sit within a few hundred nats of the
//   reference; candidates thousands of nats down are rounded coarsely and weigh e^-1000.  The
//   tolerance of the path is 1e-4 RELATIVE on logZ ~ 1e5 and 1e-4 on posteriors; the tests hold 1e-6 / 2e-5.
#include "smm_device.h"
#include "smm_launch.h"
#include "../../include/smmdp.h"

#define SMM_LOG2E 1.4426950408889634
#define SMM_LN2 0.6931471805599453
// Round 5: INSIDE the workgroup every log-weight is kept in log2 units (x log2 e): what the chain wave and the pushers exchange through
// LDS (h, A', cumE, gamma, elp) and the chain wave's own tables (trans, the short lengths).  v_exp_f32 / v_log_f32 are base 2, so the
// chain wave's 14 exponentials and 2 logarithms per position lose their conversion multiplies -- 16 of the ~140 instructions of the
// position's serial stream.  The mover wave converts at the boundary: elp x log2 e on the way into LDS, the histories x ln 2 on the way
// to HBM (two roundings of 2^-53 relative: nothing at the path's 1e-6).
#define SMM_M_EMPTY (-1e30f)     // exponent of an empty slot (finite: M - M' must never be inf - inf)

#ifndef SMM_LZ_B
#define SMM_LZ_B 4               // positions per hand-over block (long rings)
#endif
#ifndef SMM_LZ_B2_MAX_R
#define SMM_LZ_B2_MAX_R 2        // rings of up to 64 * this many slots hand over in blocks of 2
#endif

__device__ __forceinline__ float smm_exp2f(float x) { return __builtin_amdgcn_exp2f(x); }

// log(exp(a) + exp(b)) for doubles of any magnitude, transcendental part in fp32
__device__ __forceinline__ double smm_lse2(double a, double b)
{
    const double mx = smm_fmax(a, b);
    const float d = (float)(a - b);
    const float t = __builtin_amdgcn_logf(1.f + smm_exp2f(-fabsf(d) * (float)SMM_LOG2E)) * (float)SMM_LN2;
    return (mx == SMM_NEG_INF) ? mx : mx + (double)t;
}

// A wave-uniform double, told to the compiler: it then lives in a scalar register pair.
__device__ __forceinline__ double smm_uniform(double x)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    return __hiloint2double(hi, lo);
}

// One block (B sources) of one state's ring.  Same slot / register / rotation scheme as smm_ring_block of the Viterbi
// kernel: push step t = s + B - 1, u = t mod R; logical length register r lives in physical register (r - u) mod R and
// one register crosses lanes per step.  The B candidates of a slot are gathered first (register aliases, no copies:
// the loops are unrolled) and folded in together.
template <int R, int B>
__device__ __forceinline__ void smm_lse_ring_block(float (&M)[R], float (&S)[R], float (&L)[R], double &ref, double &hd,
                                                   const double *h_blk, double *a_blk, int j, int jj, int lane, const double lm)
{
    constexpr int RING = 64 * R;
    double src[B];
    {
        double hv[B];
#pragma unroll
        for (int i = 0; i < B; ++i) hv[i] = h_blk[i * SMM_MAX_STATES_DEV];
        src[0] = hd;                                       // D = 1: the last row of the block before
#pragma unroll
        for (int i = 1; i < B; ++i) src[i] = hv[i - 1];
        hd = hv[B - 1];
    }
    // Reference of the state (of h + lm, lm = the state's largest ring length score: the ring's L are kept relative to it, see the
    // pusher waves' set-up): integer-valued, >= the block's own sources (every c0 <= 0).  It moves UP with every new maximum at once.
    // Round 5: it also moves DOWN -- h is the path's advantage over emitting c for ever and FALLS wherever segments pay for their
    // lengths (a state whose likely lengths lie beyond the span limit: hundreds of nats per segment), and a reference that only moved
    // up stood thousands of units above every live candidate after a few hundred frames: an fp32 c0 of -3500 resolves 2.4e-4, and
    // scripts/soak_logz.py found log Z 4e-4 and posteriors up to 8e-4 off the twin's on such lattices (a 200-frame video of three
    // states: the forward message 4.6e-4 off at its end; 1e-6 with this and lm).  Down by at most 2^15 a block, towards the block's own
    // maximum: the slots' exponents move by the same exact integer, and over a slot's life of at most RING / B <= 256 blocks they
    // stay below 2^24.  (A drop of 1e9 behind a mask is followed at that pace, i.e. not at all -- as before.)
    double hm = src[0];
#pragma unroll
    for (int i = 1; i < B; ++i) hm = smm_fmax(hm, src[i]);
    const double ch = __builtin_ceil(hm + lm);                 // (h_blk is in log2 units)
    const double nr = smm_fmax(ch, ref - 32768.0);             // (no finite source in the block: ch = -inf and the reference just drifts)
    const float dlt = (float)(nr - ref);
    ref = nr;
    const double nrl = nr - lm;
    float c0[B];
#pragma unroll
    for (int i = 0; i < B; ++i) c0[i] = (float)(src[i] - nrl);
    // the length score every slot sees at each of the B steps, and the rotation of the ring
    float Ls[B][R];
#pragma unroll
    for (int i = 0; i < B; ++i) {
        const int u = (jj * B + i) % R;
#pragma unroll
        for (int r = 0; r < R; ++r) Ls[i][r] = L[(r - u + R) % R];
        L[(2 * R - 1 - u) % R] = smm_wave_ror1f(L[(2 * R - 1 - u) % R]);
    }
#pragma unroll
    for (int r = R - 1; r >= 0; --r) {
        const float mr = M[r] - dlt;                       // (an empty slot stays at -1e30)
        float t[B];
#pragma unroll
        for (int i = 0; i < B; ++i) t[i] = c0[i] + Ls[i][r];
        float tm = t[0];
#pragma unroll
        for (int i = 1; i < B; ++i) tm = fmaxf(tm, t[i]);
        const float mn = fmaxf(mr, __builtin_ceilf(tm));
        float acc = S[r] * smm_exp2f(mr - mn);
#pragma unroll
        for (int i = 0; i < B; ++i) acc += smm_exp2f(t[i] - mn);
        M[r] = mn;
        S[r] = acc;
    }
    // hand A' of block j+1 to the chain wave (log2 units, fp64) and clear those slots
    auto hand = [&](int r, double *dst) {
        // (M + log2 S summed in fp64: in fp32 the sum of an exponent of magnitude ~8 and a fraction is rounded to 5e-7 --
        // once per POSITION and state, a random walk that reached 5e-5 in log Z at T = 8192 (round 4:
        // tests/test_gpu_fullsize.py::test_logz_gradient_error_does_not_grow_with_the_lattice); M is integer-valued,
        // log2 S in [0, 10] carries 6e-8)
        const float lg = __builtin_amdgcn_logf(S[r]);      // log2; -inf for an empty slot
        *dst = (S[r] > 0.f) ? (ref + (double)M[r]) + (double)lg : SMM_NEG_INF;       // (log2 units)
        M[r] = SMM_M_EMPTY;
        S[r] = 0.f;
    };
    if constexpr (R % B == 0) {
        if (lane == (((j + 1) * B) & (RING - 1)) / R) {
#pragma unroll
            for (int i = 0; i < B; ++i) hand(((jj + 1) * B + i) % R, &a_blk[i * SMM_MAX_STATES_DEV]);
        }
    } else if constexpr (B % R == 0) {
        const int d = lane - (((j + 1) * B) & (RING - 1)) / R;
        if (d >= 0 && d < B / R) {
#pragma unroll
            for (int r = 0; r < R; ++r) hand(r, &a_blk[(d * R + r) * SMM_MAX_STATES_DEV]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < B; ++i) {
            if (lane == (((j + 1) * B + i) & (RING - 1)) / R) hand(((jj + 1) * B + i) % R, &a_blk[i * SMM_MAX_STATES_DEV]);
        }
    }
}

// R   ring registers per lane (RING = 64 R >= kp)      SPW  states per pusher wave
// NW  waves per workgroup (1 chain + NW-1 pushers)       HF   source states per lane group of the chain wave: 16 (two groups of
//                                                             32 lanes, up to 32 states) or 4 (four groups of 16 lanes, up to 16 states)
// B   positions per hand-over block (K0 = 2 B lengths stay with the chain wave): 4 where the pushers bound the frame time
//     (long rings: the per-block rescaling of a slot amortises over more candidates), 2 where the chain wave does
//     (short rings: half the candidates folded serially per position)
template <int R, int SPW, int NW, int HF, int B>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(1, (NW + 3) / 4)))
smm_logz_kernel(SmmDpArgs a, double *logz)
{
    constexpr int K0 = 2 * B;                              // segment lengths the chain wave evaluates itself (D = 1)
    constexpr int NP = NW - 1;
    constexpr int UB = (R / B) > 2 ? (R / B) : 2;          // blocks per unrolled pusher iteration (UB*B % R == 0, UB even)
    constexpr int MQ = 4 * B;                              // chain wave: h[n] of the last MQ > K0 positions, slot n mod MQ
    constexpr int MW = (NW >= 8) ? 4 : 1;                  // the wave that moves HBM traffic (shares the chain wave's SIMD)
    // flags bit 6: two workgroups per video, the second one runs the time-reversed recursion (independent of the first)
    const bool both = (a.flags & 64) != 0;
    const int vid = a.order[both ? blockIdx.x >> 1 : blockIdx.x];
    const SmmVideo mv = a.videos[vid];
    const int T = mv.T - ((a.flags & 8) ? 1 : 0);   // no EOS: the DP covers the frames before the last one (smmdp.h)
    const int g = mv.group;
    const int C = a.n_states[g];
    const int cm = a.c_max;
    const int kp = mv.kp;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;

    const bool bwd = (a.flags & 2) != 0 || (both && (blockIdx.x & 1));
    const double *trans = ((both && bwd) ? a.trans_t : a.trans) + (size_t)g * cm * cm;
    const double *init = a.init + (size_t)g * cm;
    const double *len = a.len + (size_t)g * a.k_rows * cm;
    const double *elp = a.elp + (size_t)mv.frame_off * cm;
    const double *endpen = a.endpen ? a.endpen + (size_t)vid * cm : nullptr;
    if (both && bwd) logz = a.logz_b;
    // bwd (a.flags bit 1, or the odd workgroups of a two-direction launch): the same recursion on the time-reversed video with the transposed transition table gives
    // the backward messages (see smm_logz_bwd.hip); its history goes to the second half of the video's block.
    // no_eos (a.flags bit 3): add_eos=False of the reference (modules:494-505): T counts the frames BEFORE the last one;
    // the video closes with a transition into the label of frame T, which only emits (no length score, no EOS).
    const bool no_eos = (a.flags & 8) != 0;
    double *hcum = a.hist + mv.hist_off + (bwd ? (size_t)3 * cm * (T + 1) : 0);   // [T+1][cm]  cumE[n][c]
    double *hh = hcum + (size_t)cm * (T + 1);             // [T+1][cm]  h[n][c]   (log-weight of "a span of c starts at n" - cumE)
    double *hgam = hh + (size_t)cm * (T + 1);             // [T+1][cm]  gamma[n][c] (log-weight of "a span of c ends at n")

    // block q = positions qB+1 .. (q+1)B, buffer q & 1
    __shared__ __attribute__((aligned(16))) double sh_apart[2][B][SMM_MAX_STATES_DEV];   // A'[n][c]   pushers -> chain
    __shared__ __attribute__((aligned(16))) double sh_h[2][B][SMM_MAX_STATES_DEV];       // h[n][c]    chain -> pushers, HBM
    __shared__ __attribute__((aligned(16))) double sh_cum[2][B][SMM_MAX_STATES_DEV];     // cumE[n][c] chain -> HBM
    __shared__ __attribute__((aligned(16))) double sh_g[2][B][SMM_MAX_STATES_DEV];       // gamma[n][c] chain -> HBM
    __shared__ __attribute__((aligned(16))) double sh_e[2][B][SMM_MAX_STATES_DEV];       // elp[n-1][c] HBM -> chain
    __shared__ __attribute__((aligned(16))) double sh_gam[SMM_MAX_STATES_DEV];           // gamma[n][.] chain-private broadcast
    __shared__ double sh_gfin[SMM_MAX_STATES_DEV];                                        // gamma[T][.] for the closing step
    __shared__ __attribute__((aligned(16))) double sh_junk[2][B][SMM_MAX_STATES_DEV];    // where the chain wave's upper half stores
    __shared__ double sh_h0[SMM_MAX_STATES_DEV];

    if (T <= 0) return;
    // frame of the (possibly time-reversed) video behind position n-1 .. : row i of the DP <-> frame fr(i)
    auto frame_of = [&](int i) { return bwd ? T - 1 - i : i; };
    if (threadIdx.x < SMM_MAX_STATES_DEV) {
        const int c = threadIdx.x;
        // start weights: forward = init; backward = weight of "the video ends after a span of c":
        //   EOS:    LSE(endpen[c], LSE_to(trans[to][c]) - 1e9)      (a.trans is the transposed table in that mode: row c)
        //   no EOS: LSE_to(trans[to][c] + elp[T][to])
        double h0 = SMM_NEG_INF;
        if (c < C) {
            if (!bwd) {
                h0 = init[c];
            } else if (no_eos) {
                for (int t2 = 0; t2 < C; ++t2) h0 = smm_lse2(h0, trans[(size_t)c * cm + t2] + elp[(size_t)T * cm + t2]);
            } else {
                double alt = SMM_NEG_INF;
                for (int t2 = 0; t2 < C; ++t2) alt = smm_lse2(alt, trans[(size_t)c * cm + t2]);
                h0 = smm_lse2(endpen ? endpen[c] : 0.0, alt + SMM_BIG_NEG);
            }
        }
        sh_h0[c] = h0;
#pragma unroll
        for (int i = 0; i < B; ++i) {
            sh_apart[0][i][c] = SMM_NEG_INF;              // block 0 needs no pusher source
            sh_apart[1][i][c] = SMM_NEG_INF;
            sh_h[0][i][c] = SMM_NEG_INF;
            sh_h[1][i][c] = (i == B - 1) ? h0 * SMM_LOG2E : SMM_NEG_INF;     // "block -1": only position 0 exists (log2 units, as everything in LDS)
            sh_e[0][i][c] = (c < C && i < T) ? elp[(size_t)frame_of(i) * cm + c] * SMM_LOG2E : 0.0;    // block 0
            sh_e[1][i][c] = 0.0;   // (columns >= c_max are never written again: the chain wave's dead lanes must not read LDS garbage)
        }
        sh_gam[c] = SMM_NEG_INF;
        sh_gfin[c] = SMM_NEG_INF;
        if (c < C) { hcum[c] = 0.0; hh[c] = h0; }
    }
    __syncthreads();

    constexpr int NE = (B * SMM_MAX_STATES_DEV + 63) / 64;   // block elements per lane of the mover
    const int J = (T + B - 1) / B;                         // blocks; one barrier each, in every wave
    if (w == 0) {
        // ============================================================================ chain wave (lane = state)
        __builtin_amdgcn_s_setprio(3);
        // lane = (target state `to`, group `half` of source states): two groups of 32 lanes x 16 sources, or -- up to 16 states --
        // four groups of 16 lanes x 4 sources: a quarter of the transition's exponentials per lane, one more swap in its two reductions
        constexpr int LPG = (HF == 4) ? 16 : 32;          // lanes per group
        static_assert(HF == 4 || HF == 16, "lane groups of the chain wave");
        const int to = lane & (LPG - 1), half = lane / LPG;
        const bool live = to < C;
        double tr[HF];                                    // trans[to][half*HF + i], log2 units
#pragma unroll
        for (int i = 0; i < HF; ++i) {
            const int f = half * HF + i;
            tr[i] = (live && f < C) ? trans[(size_t)to * cm + f] * SMM_LOG2E : SMM_NEG_INF;
        }
        double lk[K0 + 1];                                // len[k][to], k = 1..K0
#pragma unroll
        for (int k = 1; k <= K0; ++k) lk[k] = (live && k <= kp - 1) ? len[(size_t)k * cm + to] * SMM_LOG2E : SMM_NEG_INF;
        double hq[MQ];                                    // h[n][to], slot n mod MQ
#pragma unroll
        for (int i = 0; i < MQ; ++i) hq[i] = SMM_NEG_INF;
        hq[0] = live ? sh_h0[to] * SMM_LOG2E : SMM_NEG_INF;
        double cum = 0.0;
        // every group computes every position; all but the first store to a junk array (no exec juggling on the serial path)
        double *const st_gam = half ? &sh_junk[0][0][to] : &sh_gam[to];
        double *const st_fin = half ? &sh_junk[0][0][to] : &sh_gfin[to];
        constexpr bool TAILFREE = R < 16;                  // no bounds tests inside a block (smm_viterbi.hip, the same loop)
        double *const st_g = half ? &sh_junk[0][0][to] : &sh_g[0][0][to];
        double *const st_cum = half ? &sh_junk[0][0][to] : &sh_cum[0][0][to];
        double *const st_h = half ? &sh_junk[0][0][to] : &sh_h[0][0][to];
        constexpr int UC = MQ / B;                        // blocks per unrolled chain iteration (UC*B % MQ == 0, UC even)
        __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): the tables have arrived; the loop is LDS-only
        for (int j0 = 0; j0 < J; j0 += UC) {
#pragma unroll
            for (int jj = 0; jj < UC; ++jj) {
                const int j = j0 + jj;
                if (j >= J) break;
                double ap[B], ev[B];
#pragma unroll
                for (int i = 0; i < B; ++i) {
                    ap[i] = sh_apart[jj & 1][i][to];
                    ev[i] = live ? sh_e[jj & 1][i][to] : 0.0;   // dead lanes stay at (cum 0, everything else -inf): no NaN can form
                }
                // Everything of a position that does not depend on h[n-1] -- the pushers' A' and the candidates
                // k = 2..K0 -- is folded into (pm, ps) ahead of the serial path: A = LSE(that, h[n-1] + len[1]).
                auto partial = [&](int i, double &pm, float &ps) {
                    double x[K0 + 1];
#pragma unroll
                    for (int k = 2; k <= K0; ++k) x[k] = hq[(jj * B + 1 + i - k + 4 * MQ) % MQ] + lk[k];
                    pm = ap[i];
#pragma unroll
                    for (int k = 2; k <= K0; ++k) pm = smm_fmax(pm, x[k]);
                    // (reference of the exponentials: the maximum, or a huge finite number when everything is -inf -- then every
                    // difference is -inf, every exp2 is 0 and log2(0) = -inf carries on; one v_max where a compare and two
                    // selects stood, five times per position)
                    const double rf = smm_fmax(pm, -1e300);
                    ps = smm_exp2f((float)(ap[i] - rf));
#pragma unroll
                    for (int k = 2; k <= K0; ++k) ps += smm_exp2f((float)(x[k] - rf));
                };
                double pm;
                float ps;
                partial(0, pm, ps);
                double cumn = cum + ev[0];
#pragma unroll
                for (int i = 0; i < B; ++i) {
                    const int n = j * B + 1 + i;           // position; n mod MQ == (jj*B + 1 + i) mod MQ
                    if constexpr (!TAILFREE) { if (n > T) break; }
                    // A[n] = LSE( (pm, ps), h[n-1] + len[1] )
                    const double x1 = hq[(jj * B + i + 4 * MQ) % MQ] + lk[1];
                    const double mx = smm_fmax(pm, x1);
                    const double rf = smm_fmax(mx, -1e300);
                    const float s = ps * smm_exp2f((float)(pm - rf)) + smm_exp2f((float)(x1 - rf));
                    const double acc = mx + (double)__builtin_amdgcn_logf(s);          // (-inf + log2(0) = -inf)
                    cum = cumn;
                    const double gm = cum + acc;
                    st_gam[0] = gm;
                    st_g[((jj & 1) * B + i) * SMM_MAX_STATES_DEV] = gm;
                    st_cum[((jj & 1) * B + i) * SMM_MAX_STATES_DEV] = cum;
                    if (n == T) st_fin[0] = gm;                                  // (wave-uniform, once per video)
                    if (TAILFREE || n < T) {
                        // beta[to] = LSE_from (gamma[from] + trans[to][from]); this half folds sources half*HF ..
                        const double2 *gp = reinterpret_cast<const double2 *>(&sh_gam[half * HF]);
                        double2 gv[HF / 2];
#pragma unroll
                        for (int q = 0; q < HF / 2; ++q) gv[q] = gp[q];
                        __builtin_amdgcn_sched_barrier(0);
                        if (i + 1 < B) {                   // the next position's h-independent part, in the shadow of the LDS round trip
                            partial(i + 1 < B ? i + 1 : 0, pm, ps);
                            cumn = cum + ev[i + 1 < B ? i + 1 : 0];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        double v[HF];
                        double vm = SMM_NEG_INF;
#pragma unroll
                        for (int q = 0; q < HF / 2; ++q) {
                            v[2 * q] = gv[q].x + tr[2 * q];
                            v[2 * q + 1] = gv[q].y + tr[2 * q + 1];
                            vm = smm_fmax(vm, smm_fmax(v[2 * q], v[2 * q + 1]));
                        }
                        if constexpr (HF == 4) vm = smm_max_rows16(vm);
                        vm = smm_max_halves(vm);                    // common reference of all groups
                        const double vr = smm_fmax(vm, -1e300);
                        float sv = 0.f;
#pragma unroll
                        for (int q = 0; q < HF; ++q) sv += smm_exp2f((float)(v[q] - vr));
                        if constexpr (HF == 4) sv = smm_sum_rows16f(sv);
                        sv = smm_sum_halvesf(sv);                   // (v_permlane32_swap: __shfl_xor(.., 32) is a trip through LDS)
                        const double beta = vm + (double)__builtin_amdgcn_logf(sv);
                        const double hcur = beta - cum;
                        hq[(jj * B + 1 + i) % MQ] = hcur;
                        st_h[((jj & 1) * B + i) * SMM_MAX_STATES_DEV] = hcur;
                    }
                }
                __syncthreads();                                           // end of block j
            }
        }
    } else {
        // ============================================================================ pusher waves
        int rank = w - 1;
        if (NW == 8) rank = (w == 4) ? NP - 1 : ((w == NW - 1) ? 3 : w - 1);   // the chain wave's SIMD partner owns the fewest states
        const int nv_all = (C - rank + NP - 1) / NP;                       // states rank, rank+NP, ...
        const int nv = nv_all < 0 ? 0 : (nv_all > SPW ? SPW : nv_all);
        // Round 5: a state's length scores ride in the ring RELATIVE to the largest of them, lmx (which joins h in the state's reference):
        // a slot's exponent t = c0 + L is an fp32 number, and with the scores themselves in it every candidate carried 2^-24 |score| of
        // rounding -- 2e-5 nats where a state's likely lengths lie beyond the span limit (scores of -300 .. -500 for EVERY allowed
        // length).  Relative to lmx the lengths that carry the mass have |L| of a few units; the ones hundreds of nats below weigh
        // e^-100 as before.
        float M[SPW][R], S[SPW][R], L[SPW][R];
        double ref[SPW], hd[SPW], lmx[SPW];
#pragma unroll
        for (int js = 0; js < SPW; ++js) {
            const int c = js * NP + rank;
            const bool on = js < nv;
            double lv[R];
            double lm = SMM_NEG_INF;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                // block protocol (source position -B at push step 0): slot p waits for k = (p + B + 1) mod RING; lengths
                // up to K0 belong to the chain wave
                const int k = (lane * R + r + B + 1) & (64 * R - 1);
                M[js][r] = SMM_M_EMPTY;
                S[js][r] = 0.f;
                lv[r] = (on && k > K0 && k <= kp - 1) ? len[(size_t)k * cm + c] * SMM_LOG2E : SMM_NEG_INF;
                lm = smm_fmax(lm, lv[r]);
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) lm = smm_fmax(lm, __shfl_xor(lm, off));
            if (!(lm > -1e300)) lm = 0.0;                                  // (no length beyond K0 in reach: every L is -inf anyway)
            lm = smm_uniform(lm);                                           // (a scalar register pair per state)
#pragma unroll
            for (int r = 0; r < R; ++r) L[js][r] = (lv[r] > -1e300) ? (float)(lv[r] - lm) : -__builtin_huge_valf();
            lmx[js] = lm;
            const double h0 = on ? sh_h0[c] : 0.0;
            ref[js] = (h0 > -1e300 && h0 < 1e300) ? __builtin_ceil(h0 * SMM_LOG2E + lm) : 0.0;
            hd[js] = SMM_NEG_INF;
        }
        // mover role of this wave: block-relative element e = lane + 64 q  <->  (row e / cm, column e % cm)
        int lo[NE], row[NE], col[NE];
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            const int e = lane + 64 * q;
            row[q] = e / cm;
            col[q] = e - row[q] * cm;
            lo[q] = (e < B * cm) ? row[q] * SMM_MAX_STATES_DEV + col[q] : -1;
        }
        // elp rows of block q (positions qB+1.., i.e. DP rows qB..): unconditional loads from clamped rows
        auto load_block = [&](double (&pre)[NE], int q) {
#pragma unroll
            for (int x = 0; x < NE; ++x) {
                int i = q * B + row[x];
                i = i < T ? i : T - 1;
                pre[x] = elp[(size_t)frame_of(i) * cm + (lo[x] >= 0 ? col[x] : 0)];
            }
        };
        double pre[NE];
        if (w == MW) load_block(pre, 1);
        __builtin_amdgcn_s_waitcnt(0x0F70);                                // vmcnt(0), see smm_viterbi.hip
        auto store_block = [&](const double *src, double *dst, int q) {
#pragma unroll
            for (int x = 0; x < NE; ++x) {
                const int e = lane + 64 * x;
                if (lo[x] >= 0 && q * B + 1 + row[x] <= T) dst[(size_t)(q * B + 1) * cm + e] = src[lo[x]] * SMM_LN2;   // (LDS: log2 units)
            }
        };
        for (int j0 = 0; j0 < J; j0 += UB) {
#pragma unroll
            for (int jj = 0; jj < UB; ++jj) {
                const int j = j0 + jj;
                if (j >= J) break;
                if (w == MW) {
                    // block j+1 (fetched a block ago) -> LDS, then fetch block j+2; history of block j-1 -> HBM
                    double *dst = &sh_e[(jj + 1) & 1][0][0];
#pragma unroll
                    for (int q = 0; q < NE; ++q)
                        if (lo[q] >= 0) dst[lo[q]] = pre[q] * SMM_LOG2E;
                    load_block(pre, j + 2);
                    if (j >= 1) {
                        store_block(&sh_cum[(jj + 1) & 1][0][0], hcum, j - 1);
                        store_block(&sh_h[(jj + 1) & 1][0][0], hh, j - 1);
                        store_block(&sh_g[(jj + 1) & 1][0][0], hgam, j - 1);
                    }
                }
#pragma unroll
                for (int js = 0; js < SPW; ++js) {
                    if (js >= nv) break;
                    const int c = js * NP + rank;
                    smm_lse_ring_block<R, B>(M[js], S[js], L[js], ref[js], hd[js], &sh_h[(jj + 1) & 1][0][c],
                                             &sh_apart[(jj + 1) & 1][0][c], j, jj % UB, lane, lmx[js]);
                }
                __syncthreads();                             // end of block j
            }
        }
        if (w == MW) {                                       // the last block's history
            store_block(&sh_cum[(J - 1) & 1][0][0], hcum, J - 1);
            store_block(&sh_h[(J - 1) & 1][0][0], hh, J - 1);
            store_block(&sh_g[(J - 1) & 1][0][0], hgam, J - 1);
        }
    }

    // -------------------------------------------------------------------------------- last position
    // sh_gfin holds gamma[T][.] (log2 units)
    __syncthreads();
    if (threadIdx.x < SMM_MAX_STATES_DEV) sh_gfin[threadIdx.x] *= SMM_LN2;
    __syncthreads();
    if (w == 0) {
        double f = SMM_NEG_INF;
        if (!bwd) {
            if (no_eos) {
                if (lane < C) {
                    for (int c = 0; c < C; ++c) f = smm_lse2(f, sh_gfin[c] + trans[(size_t)lane * cm + c]);
                    f = f + elp[(size_t)T * cm + lane];   // the closing label only emits frame T
                }
            } else if (lane <= C) {
                for (int c = 0; c < C; ++c) {
                    const double wgt = (lane == C) ? (endpen ? endpen[c] : 0.0) : trans[(size_t)lane * cm + c] + SMM_BIG_NEG;
                    f = smm_lse2(f, sh_gfin[c] + wgt);
                }
            }
        } else if (lane < C) {
            f = sh_gfin[lane] + init[lane];               // closes the recursion: must reproduce log Z
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) f = smm_lse2(f, __shfl_xor(f, off));
        if (lane == 0) logz[vid] = f;
    }
}

// ------------------------------------------------------------------------------------------------ dispatch
template <int R, int SPW>
static int logz_launch_if(const SmmDpArgs &a, double *logz, int spw, int c_need, hipStream_t stream)
{
    if (spw != SPW) return 0;
    constexpr int B = (R <= SMM_LZ_B2_MAX_R) ? 2 : SMM_LZ_B;
    const dim3 grid((a.flags & 64) ? 2 * a.b : a.b);
    if (c_need <= 16) hipLaunchKernelGGL((smm_logz_kernel<R, SPW, 8, 4, B>), grid, dim3(512), 0, stream, a, logz);
    else hipLaunchKernelGGL((smm_logz_kernel<R, SPW, 8, 16, B>), grid, dim3(512), 0, stream, a, logz);
    return 1;
}

template <int R>
static int logz_launch_r(const SmmDpArgs &a, double *logz, int c_need, hipStream_t stream)
{
    // 8 waves: 1 chain + 7 pushers x SPW states; a pusher keeps 3*R*SPW ring registers (+ ~50): everything up to
    // 28 states x 1024 slots stays in the register file (32 states x 1024: the 5-state configuration spills a little)
    const int spw = (c_need + 6) / 7;
    const int hit = logz_launch_if<R, 1>(a, logz, spw, c_need, stream) || logz_launch_if<R, 2>(a, logz, spw, c_need, stream) ||
                    logz_launch_if<R, 3>(a, logz, spw, c_need, stream) || logz_launch_if<R, 4>(a, logz, spw, c_need, stream) ||
                    logz_launch_if<R, 5>(a, logz, spw, c_need, stream);
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
