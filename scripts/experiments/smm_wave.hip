// smm_wave.hip -- EXPERIMENT, not part of libsmmdp (round 5; profiles/round5_wave_kernel.txt): semi-Markov Viterbi for SHORT span
// limits (kp - 1 <= 32) with one WAVE per video.  Built into the library for one measurement (SOURCES of _build.py + a dispatch
// line in smm_api.hip: run_viterbi), it passed the whole GPU suite and ran refdef's decode in 0.44 ms where the eight-wave kernel
// with its window back-trace needs 0.25 -- kept here as the record of what was measured, not compiled by build().
//
// lane = state (<= 32).  h of the last KR positions and len[1 .. KR-1] of the lane's state live in registers (the position loop is
// unrolled KR times, so the ring index is static); a position is  cumE += elp,  A = max_k (h[n-k] + len[k]),  gamma = cumE + A,  the
// C x C transition (gamma broadcast through 256 B of the wave's LDS row, every lane folds its own row of the table from registers),
// h = beta - cumE.  elp rows are fetched four positions ahead; cumE / gamma / h rows go to a frame-major history.  Values only: the
// arg-max is recovered per SEGMENT by re-evaluating the forward pass's expressions (lane = state for the maximum, lane = length for
// the first length that attains it; the predecessor seen the last time travels with the row).
#include "smm_device.h"
#include "../../include/smmdp.h"
#include "smm_launch.h"

#define SMM_WAVE_WPB 4            // videos (waves) per workgroup
#define SMM_WAVE_PF 4             // elp rows in flight per wave

template <int KR, int CH>
__global__ void __launch_bounds__(SMM_WAVE_WPB * 64)
smm_viterbi_wave_kernel(SmmDpArgs a)
{
    static_assert(KR % SMM_WAVE_PF == 0 && KR >= 4 && KR <= 32, "ring size");
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int slot = blockIdx.x * SMM_WAVE_WPB + w;
    __shared__ __attribute__((aligned(16))) double sh_g[SMM_WAVE_WPB][SMM_MAX_STATES_DEV];
    __shared__ int sh_guess[SMM_WAVE_WPB][SMM_MAX_STATES_DEV + 1];
    if (slot >= a.b) return;
    const int vid = a.order[slot];
    SmmVideo mv = a.videos[vid];
    const bool no_eos = (a.flags & 8) != 0;
    if (no_eos) mv.T -= 1;
    const int T = mv.T, g = mv.group, C = a.n_states[g], cm = a.c_max, kp = mv.kp;
    if (T <= 0) return;
    const double *trans = a.trans + (size_t)g * cm * cm;
    const double *init = a.init + (size_t)g * cm;
    const double *len = a.len + (size_t)g * a.k_rows * cm;
    const double *elp = a.elp + (size_t)mv.frame_off * cm;
    const double *endpen = a.endpen ? a.endpen + (size_t)vid * cm : nullptr;
    const int64_t *cmap = a.class_map ? a.class_map + (size_t)g * (cm + 1) : nullptr;
    double *hcum = a.hist + mv.hist_off;
    double *hgam = hcum + (size_t)C * (T + 1);
    double *hh = hgam + (size_t)C * (T + 1);
    int64_t *spans = a.spans ? a.spans + (size_t)vid * (a.t_max + 1) : nullptr;
    int64_t *labels = a.labels ? a.labels + mv.frame_off : nullptr;
    double *gbc = &sh_g[w][0];
    const int c = lane & (SMM_MAX_STATES_DEV - 1);
    const bool live = c < C, mine = lane < C;
    if (spans)
        for (int i = lane; i <= a.t_max; i += 64) spans[i] = -1;
    double tr[CH], lk[KR], hq[KR], ev[SMM_WAVE_PF];
#pragma unroll
    for (int f = 0; f < CH; ++f) tr[f] = (live && f < C) ? trans[(size_t)c * cm + f] : SMM_NEG_INF;
#pragma unroll
    for (int k = 1; k < KR; ++k) lk[k] = (live && k <= kp - 1) ? len[(size_t)k * cm + c] : SMM_NEG_INF;
    const double lk_top = (live && KR <= kp - 1) ? len[(size_t)KR * cm + c] : SMM_NEG_INF;
#pragma unroll
    for (int i = 0; i < KR; ++i) hq[i] = SMM_NEG_INF;
    hq[0] = live ? init[c] : SMM_NEG_INF;
    const int64_t e_last = (int64_t)T * cm - 1;
    auto row_at = [&](int r) { const int64_t e = (int64_t)r * cm + c; return elp[e < e_last ? e : e_last]; };
#pragma unroll
    for (int i = 0; i < SMM_WAVE_PF; ++i) ev[i] = live ? row_at(i) : 0.0;
    if (mine) { hcum[lane] = 0.0; hh[lane] = init[lane]; }
    double cum = 0.0, gam = SMM_NEG_INF, gam_T = SMM_NEG_INF;
    for (int n0 = 1; n0 <= T; n0 += KR) {
#pragma unroll
        for (int i = 0; i < KR; ++i) {
            // (no exit from inside the round: it would keep it from unrolling and the ring would be a scratch array; positions past T
            // are computed like the others, never stored)
            const int n = n0 + i;
            cum = cum + ev[i % SMM_WAVE_PF];
            ev[i % SMM_WAVE_PF] = live ? row_at(n - 1 + SMM_WAVE_PF) : 0.0;
            double acc = hq[(1 + i) % KR] + lk_top;
#pragma unroll
            for (int k0 = 1; k0 < KR; k0 += 4) {
                double sq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (k0 + q < KR) sq[q] = hq[(1 + i - (k0 + q) + 2 * KR) % KR] + lk[k0 + q];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (k0 + q < KR) acc = fmax(acc, sq[q]);
            }
            gam = cum + acc;
            gbc[c] = live ? gam : SMM_NEG_INF;
            double bq[4] = {SMM_NEG_INF, SMM_NEG_INF, SMM_NEG_INF, SMM_NEG_INF};
#pragma unroll
            for (int f = 0; f < CH; f += 2) {
                const double2 g2 = *reinterpret_cast<const double2 *>(gbc + f);
                bq[(f >> 1) & 1] = fmax(bq[(f >> 1) & 1], g2.x + tr[f]);
                bq[2 + ((f >> 1) & 1)] = fmax(bq[2 + ((f >> 1) & 1)], g2.y + tr[f + 1]);
            }
            const double beta = fmax(fmax(bq[0], bq[1]), fmax(bq[2], bq[3]));
            const double hn = beta - cum;
            hq[(1 + i) % KR] = hn;
            if (n == T) gam_T = gam;
            if (mine && n <= T) {
                hcum[(size_t)n * C + lane] = cum;
                hgam[(size_t)n * C + lane] = gam;
                hh[(size_t)n * C + lane] = hn;
            }
        }
    }
    gbc[c] = live ? gam_T : SMM_NEG_INF;
    int to;
    {
        double f = SMM_NEG_INF;
        const int last = no_eos ? C - 1 : C;
        if (lane <= last) {
            for (int c2 = 0; c2 < C; ++c2) {
                const double wgt = (lane == C) ? (endpen ? endpen[c2] : 0.0) : trans[(size_t)lane * cm + c2];
                f = fmax(f, gbc[c2] + wgt);
            }
            if (no_eos) f = f + elp[(size_t)T * cm + lane];
            else if (lane < C) f = f + SMM_BIG_NEG;
        }
        int cc = (lane <= last) ? lane : 0x7fffffff;
        if (lane > last) f = SMM_NEG_INF;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double f2 = __shfl_xor(f, off);
            const int c2 = __shfl_xor(cc, off);
            if (f2 > f || (f2 == f && c2 < cc)) { f = f2; cc = c2; }
        }
        to = cc;
        if (lane == 0) {
            if (a.best) a.best[vid] = f;
            if (spans) spans[T] = cmap ? cmap[cc] : cc;
            if (no_eos && labels) labels[T] = cmap ? cmap[cc] : cc;
        }
    }
    int *guess = &sh_guess[w][0];
    if (lane <= C) {
        int bi = C - 1;
        if (lane < C) {
            double bv = SMM_NEG_INF;
            bi = 0;
            for (int c2 = 0; c2 < C; ++c2) {
                const double v2 = trans[(size_t)lane * cm + c2];
                if (v2 > bv) { bv = v2; bi = c2; }
            }
        } else if (endpen) {
            for (int c2 = 0; c2 < C; ++c2)
                if (endpen[c2] == 0.0) bi = c2;
        }
        guess[lane] = bi;
    }
    const int64_t gid_l = cmap ? cmap[lane < C ? lane : C] : (int64_t)lane;
    int n = T, nseg = 0, fg = 0, kmax = 0;
    double g0 = SMM_NEG_INF, cnl = 0.0, wgt = 0.0, sp_h = 0.0, sp_l = 0.0;
    auto trip = [&](int n_, int to_) {
        fg = guess[to_];
        kmax = (kp - 1 < n_) ? kp - 1 : n_;
        if (mine) {
            g0 = hgam[(size_t)n_ * C + lane];
            cnl = hcum[(size_t)n_ * C + lane];
            wgt = (to_ == C) ? (endpen ? endpen[lane] : 0.0) : trans[(size_t)to_ * cm + lane];
        }
        const int kk0 = lane + 1, kc = kk0 <= kmax ? kk0 : kmax;
        sp_h = hh[(size_t)(n_ - kc) * C + fg];
        sp_l = len[(size_t)kc * cm + fg];
    };
    bool bad = false;
    if (n > 0) trip(n, to);
    while (n > 0) {
        const double gmv = mine ? g0 + wgt : SMM_NEG_INF;
        if (__ballot(mine && smm_nan_bits(gmv)) != 0) { bad = true; break; }
        const double rmax = smm_row_max16(gmv);
        const double best = fmax(smm_readlane(rmax, 0), smm_readlane(rmax, 16));
        unsigned long long fmask = __ballot(mine && gmv == best);
        int k = 0x7fffffff, cs = 0x7fffffff;
        while (fmask) {
            const int f = __builtin_amdgcn_readfirstlane(__ffsll(fmask) - 1);
            fmask &= fmask - 1;
            const double cn = smm_readlane(cnl, f), wf = smm_readlane(wgt, f);
            const int lim = (kmax < k - 1) ? kmax : k - 1;
            const int kk = lane + 1;
            bool hit = false;
            if (kk <= lim) {
                const double hv = (f == fg) ? sp_h : hh[(size_t)(n - kk) * C + f];
                const double lv = (f == fg) ? sp_l : len[(size_t)kk * cm + f];
                hit = ((cn + (hv + lv)) + wf) == best;
            }
            const unsigned long long m = __ballot(hit);
            if (m) { k = __ffsll(m); cs = f; }
        }
        if (k < 1 || k > kmax || cs < 0 || cs >= C) { bad = true; break; }
        const int s = n - k;
        if (lane == 0) guess[to] = cs;
        n = s;
        to = cs;
        if (n > 0) trip(n, to);
        const int64_t gid = ((int64_t)__builtin_amdgcn_readlane((int)(gid_l >> 32), cs) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)gid_l, cs);
        if (labels && lane < k) labels[s + lane] = gid;
        if (spans && lane == 0) spans[s] = gid;
        ++nseg;
    }
    if (bad && lane == 0) atomicExch(a.err, 1);
    if (a.n_segs && lane == 0) a.n_segs[vid] = nseg + (no_eos ? 1 : 0);
}

template <int KR>
static void launch_kr(const SmmDpArgs &a, int c_need, hipStream_t stream)
{
    const dim3 grid((a.b + SMM_WAVE_WPB - 1) / SMM_WAVE_WPB), block(SMM_WAVE_WPB * 64);
    if (c_need <= 16) hipLaunchKernelGGL((smm_viterbi_wave_kernel<KR, 16>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((smm_viterbi_wave_kernel<KR, 32>), grid, block, 0, stream, a);
}

// kp_max - 1 <= 32 usable lengths, c_need <= 32 states
int smm_launch_viterbi_wave(const SmmDpArgs &a, int kp_max, int c_need, hipStream_t stream)
{
    const int need = kp_max - 1;
    if (need > 32 || c_need > SMM_MAX_STATES_DEV) return SMM_ERR_UNSUPPORTED;
    if (need <= 4) launch_kr<4>(a, c_need, stream);
    else if (need <= 8) launch_kr<8>(a, c_need, stream);
    else if (need <= 12) launch_kr<12>(a, c_need, stream);
    else if (need <= 16) launch_kr<16>(a, c_need, stream);
    else if (need <= 20) launch_kr<20>(a, c_need, stream);
    else if (need <= 24) launch_kr<24>(a, c_need, stream);
    else if (need <= 28) launch_kr<28>(a, c_need, stream);
    else launch_kr<32>(a, c_need, stream);
    return SMM_OK;
}
