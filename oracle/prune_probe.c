/*
 * oracle/prune_probe.c -- EXPERIMENT (test infrastructure, never shipped): how much of the long segment-length range
 * of the factored Viterbi DP can be skipped EXACTLY with a bound test?
 *
 * The DP (oracle/smm_oracle.c) needs  A[n][c] = max_{k=1..kmax} ( h[n-k][c] + len[k][c] ).  A candidate block
 * (targets n0 .. n0+TB-1) x (lengths kb .. kb+KB-1) of one state cannot change any A[n][c] of the block when
 *     max_{sources s of the block} h[s][c]  +  max_{k in the block} len[k][c]   <=   min_{targets n} acc[n]
 * where acc[n] is the running maximum over the candidates already evaluated for target n (a lower bound of A[n][c]).
 * max is exact, so skipping such a block changes no bit of the result; the back-trace re-evaluates the candidates of
 * the optimal path only.  Lengths are walked in ascending order (the most recent sources first: h[s][c] = beta - cumE
 * grows while c is not the best explanation of the frames).
 *
 * This file measures the surviving fraction of blocks on real lattices (scripts/probe_prune.py feeds it the cfg3
 * corpus).  Source maxima are taken over 64-ALIGNED source blocks (what a kernel would keep), not over the exact span.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double dmin(double a, double b) { return a < b ? a : b; }

/*
 * elp t x c, trans c x c [to][from], init c, len kp x c.  Long range: k in [kl, kp-1], target blocks of tb positions,
 * length blocks of kb lengths.  out[0] = blocks in all (state x target block x length block with at least one valid
 * candidate), out[1] = blocks evaluated, out[2] = lattice cells in all (long range), out[3] = cells evaluated,
 * out[4] = cells of the whole lattice (all k), out[5] = blocks evaluated if only the FIRST block's minimum is used as
 * the bound (no update of the bound after later blocks).
 * Returns 0 / -1 (allocation).
 */
int smm_prune_probe(const double *elp, int t, int c, const double *trans, const double *init, const double *len, int kp,
                    int kl, int tb, int kb, double *out)
{
    double *cum = (double *)calloc((size_t)(t + 1) * c, sizeof(double));
    double *h = (double *)malloc(sizeof(double) * (size_t)(t + 1) * c);
    double *gam = (double *)malloc(sizeof(double) * c);
    const int nsb = t / 64 + 2;
    double *hmax = (double *)malloc(sizeof(double) * (size_t)nsb * c);      /* max of h over aligned 64-source blocks */
    const int nkb = (kp + kb - 1) / kb + 1;
    double *lmax = (double *)malloc(sizeof(double) * (size_t)nkb * c);      /* max of len over the length blocks */
    double *acc = (double *)malloc(sizeof(double) * tb);
    if (!cum || !h || !gam || !hmax || !lmax || !acc) return -1;
    memset(out, 0, sizeof(double) * 8);
    for (int j = 0; j < c; ++j) h[j] = init[j];
    for (int n = 1; n <= t; ++n) {
        const int kmax = (kp - 1 < n) ? kp - 1 : n;
        for (int j = 0; j < c; ++j) {
            cum[(size_t)n * c + j] = cum[(size_t)(n - 1) * c + j] + elp[(size_t)(n - 1) * c + j];
            double a = -INFINITY;
            for (int k = 1; k <= kmax; ++k) a = dmax(a, h[(size_t)(n - k) * c + j] + len[(size_t)k * c + j]);
            gam[j] = cum[(size_t)n * c + j] + a;
        }
        out[4] += (double)kmax * c;
        for (int to = 0; to < c; ++to) {
            double bt = -INFINITY;
            for (int j = 0; j < c; ++j) bt = dmax(bt, gam[j] + trans[(size_t)to * c + j]);
            h[(size_t)n * c + to] = bt - cum[(size_t)n * c + to];
        }
    }
    for (int sb = 0; sb < nsb; ++sb)
        for (int j = 0; j < c; ++j) {
            double m = -INFINITY;
            for (int s = sb * 64; s < sb * 64 + 64 && s < t; ++s) m = dmax(m, h[(size_t)s * c + j]);
            hmax[(size_t)sb * c + j] = m;
        }
    for (int q = 0; q < nkb; ++q)
        for (int j = 0; j < c; ++j) {
            double m = -INFINITY;
            for (int k = kl + q * kb; k < kl + (q + 1) * kb && k <= kp - 1; ++k) m = dmax(m, len[(size_t)k * c + j]);
            lmax[(size_t)q * c + j] = m;
        }
    for (int j = 0; j < c; ++j) {
        for (int n0 = 1; n0 <= t; n0 += tb) {
            const int n1 = (n0 + tb - 1 < t) ? n0 + tb - 1 : t;            /* targets n0 .. n1 */
            for (int i = 0; i < tb; ++i) acc[i] = -INFINITY;
            double bound_first = INFINITY;
            int first_done = 0;
            for (int q = 0; kl + q * kb <= kp - 1; ++q) {
                const int k0 = kl + q * kb, k1 = (k0 + kb - 1 < kp - 1) ? k0 + kb - 1 : kp - 1;
                /* sources n - k, n0 <= n <= n1, k0 <= k <= k1, >= 0 */
                const int s_hi = n1 - k0, s_lo = (n0 - k1 > 0) ? n0 - k1 : 0;
                if (s_hi < 0) break;                                          /* no candidate here or in any later block */
                double cells = 0.0;
                for (int n = n0; n <= n1; ++n) {
                    const int ka = k0, kz = (k1 < n) ? k1 : n;
                    if (kz >= ka) cells += kz - ka + 1;
                }
                out[0] += 1.0;
                out[2] += cells;
                double hm = -INFINITY;
                for (int sb = s_lo / 64; sb <= s_hi / 64; ++sb) hm = dmax(hm, hmax[(size_t)sb * c + j]);
                const double ub = hm + lmax[(size_t)q * c + j];
                /* lower bound: min over the targets that have a long-range candidate at all (n >= kl) */
                double lb = INFINITY;
                for (int n = (n0 > kl ? n0 : kl); n <= n1; ++n) lb = dmin(lb, acc[n - n0]);
                if (first_done && ub <= bound_first) { /* skipped by the static bound */ } else out[5] += 1.0;
                if (ub <= lb) continue;                                       /* skipped: cannot change any acc */
                out[1] += 1.0;
                out[3] += cells;
                for (int n = n0; n <= n1; ++n) {
                    const int kz = (k1 < n) ? k1 : n;
                    for (int k = k0; k <= kz; ++k)
                        acc[n - n0] = dmax(acc[n - n0], h[(size_t)(n - k) * c + j] + len[(size_t)k * c + j]);
                }
                if (!first_done) {
                    first_done = 1;
                    bound_first = INFINITY;
                    for (int n = (n0 > kl ? n0 : kl); n <= n1; ++n) bound_first = dmin(bound_first, acc[n - n0]);
                }
            }
        }
    }
    free(cum); free(h); free(gam); free(hmax); free(lmax); free(acc);
    return 0;
}

/*
 * Second design: BANDED PUSH.  The lengths are cut into bands of `bw` (band m = lengths m*bw .. m*bw+bw-1); band m
 * pushes source s - m*bw when the chain is at s, into the one ring of bw upcoming targets all bands share.  Per state,
 * per group of 64 sources and band m >= 1 the whole group is skipped when
 *     max_{s in group} h[s][c] + max_{k in band m} len[k][c]  <=  max_{s in the newest complete group} h[s][c] + min_{1<=k<=gap} len[k][c]
 * (right side: a real candidate of every target the group can reach, so a lower bound of their A[n][c]; gap = 64 + 63
 * + bw - 1).  Band 0 is always evaluated.  out[0] = (state, group, band) triples with sources, out[1] = evaluated,
 * out[2] = band-0 triples (always evaluated), out[3] = cells of the whole lattice, out[4] = cells evaluated.
 */
int smm_band_probe(const double *elp, int t, int c, const double *trans, const double *init, const double *len, int kp,
                   int bw, double *out)
{
    double *cum = (double *)calloc((size_t)(t + 1) * c, sizeof(double));
    double *h = (double *)malloc(sizeof(double) * (size_t)(t + 1) * c);
    double *gam = (double *)malloc(sizeof(double) * c);
    const int ng = t / 64 + 2, nb = (kp + bw - 1) / bw;
    double *hmax = (double *)malloc(sizeof(double) * (size_t)ng * c);
    double *lmax = (double *)malloc(sizeof(double) * (size_t)(nb + 1) * c);
    if (!cum || !h || !gam || !hmax || !lmax) return -1;
    memset(out, 0, sizeof(double) * 8);
    for (int j = 0; j < c; ++j) h[j] = init[j];
    for (int n = 1; n <= t; ++n) {
        const int kmax = (kp - 1 < n) ? kp - 1 : n;
        for (int j = 0; j < c; ++j) {
            cum[(size_t)n * c + j] = cum[(size_t)(n - 1) * c + j] + elp[(size_t)(n - 1) * c + j];
            double a = -INFINITY;
            for (int k = 1; k <= kmax; ++k) a = dmax(a, h[(size_t)(n - k) * c + j] + len[(size_t)k * c + j]);
            gam[j] = cum[(size_t)n * c + j] + a;
        }
        out[3] += (double)kmax * c;
        for (int to = 0; to < c; ++to) {
            double bt = -INFINITY;
            for (int j = 0; j < c; ++j) bt = dmax(bt, gam[j] + trans[(size_t)to * c + j]);
            h[(size_t)n * c + to] = bt - cum[(size_t)n * c + to];
        }
    }
    for (int g = 0; g < ng; ++g)
        for (int j = 0; j < c; ++j) {
            double m = -INFINITY;
            for (int s = g * 64; s < g * 64 + 64 && s < t; ++s) m = dmax(m, h[(size_t)s * c + j]);
            hmax[(size_t)g * c + j] = m;
        }
    const int gap = 64 + 63 + bw - 1;
    for (int j = 0; j < c; ++j) {
        double lmin = INFINITY;
        for (int k = 1; k <= gap && k <= kp - 1; ++k) lmin = dmin(lmin, len[(size_t)k * c + j]);
        if (gap > kp - 1) lmin = -INFINITY;
        for (int m = 0; m < nb; ++m) {
            double mm = -INFINITY;
            for (int k = m * bw; k < (m + 1) * bw && k <= kp - 1; ++k)
                if (k >= 1) mm = dmax(mm, len[(size_t)k * c + j]);
            lmax[(size_t)m * c + j] = mm;
        }
        for (int g = 0; g * 64 < t; ++g) {
            const double lb = (g >= 1) ? hmax[(size_t)(g - 1) * c + j] + lmin : -INFINITY;
            for (int m = 0; m < nb; ++m) {
                const int gs = g - m * bw / 64;                 /* the source group of band m */
                if (gs < 0) break;
                /* cells: sources s in the group, lengths k in the band, target s + k <= t */
                double cells = 0.0;
                for (int s = gs * 64; s < gs * 64 + 64 && s < t; ++s) {
                    int k0 = m * bw < 1 ? 1 : m * bw, k1 = (m + 1) * bw - 1;
                    if (k1 > kp - 1) k1 = kp - 1;
                    if (k1 > t - s) k1 = t - s;
                    if (k1 >= k0) cells += k1 - k0 + 1;
                }
                out[0] += 1.0;
                if (m == 0) out[2] += 1.0;
                const double ub = hmax[(size_t)gs * c + j] + lmax[(size_t)m * c + j];
                if (m > 0 && ub <= lb) continue;
                out[1] += 1.0;
                out[4] += cells;
            }
        }
    }
    free(cum); free(h); free(gam); free(hmax); free(lmax);
    return 0;
}

/*
 * Third probe: the banded push AS BUILT (smm_viterbi.hip, BAND mode).  Band 0 = lengths 9..127, always evaluated.
 * Band m = 1..8 covers lengths 16+112m .. 127+112m with sources delayed by 112m; tests per group of `grp` sources; the
 * witness of the lower bound is the group before the last one (the last one is still being pushed when the decision
 * is due):
 *   skip (state c, band m, undelayed group g)  iff  hmax[g - 112m/grp] + lenmax[m]  <=  hmax[g-2] + min_{2 grp + 1 <= k <= 3 grp + 126} len[k]
 * out[0] = band-groups with sources (m >= 1), out[1] = evaluated, out[2] = activation edges (inactive -> active),
 * out[3] = cells of the lattice, out[4] = cells evaluated (bands) + band 0 + chain lengths.
 */
int smm_band_probe2(const double *elp, int t, int c, const double *trans, const double *init, const double *len, int kp,
                    int grp, double *out)
{
    double *cum = (double *)calloc((size_t)(t + 1) * c, sizeof(double));
    double *h = (double *)malloc(sizeof(double) * (size_t)(t + 1) * c);
    double *gam = (double *)malloc(sizeof(double) * c);
    const int ng = t / grp + 2;
    double *hmax = (double *)malloc(sizeof(double) * (size_t)ng * c);
    if (!cum || !h || !gam || !hmax) return -1;
    memset(out, 0, sizeof(double) * 8);
    for (int j = 0; j < c; ++j) h[j] = init[j];
    for (int n = 1; n <= t; ++n) {
        const int kmax = (kp - 1 < n) ? kp - 1 : n;
        for (int j = 0; j < c; ++j) {
            cum[(size_t)n * c + j] = cum[(size_t)(n - 1) * c + j] + elp[(size_t)(n - 1) * c + j];
            double a = -INFINITY;
            for (int k = 1; k <= kmax; ++k) a = dmax(a, h[(size_t)(n - k) * c + j] + len[(size_t)k * c + j]);
            gam[j] = cum[(size_t)n * c + j] + a;
        }
        out[3] += (double)kmax * c;
        out[4] += (double)(kmax < 127 ? kmax : 127) * c;
        for (int to = 0; to < c; ++to) {
            double bt = -INFINITY;
            for (int j = 0; j < c; ++j) bt = dmax(bt, gam[j] + trans[(size_t)to * c + j]);
            h[(size_t)n * c + to] = bt - cum[(size_t)n * c + to];
        }
    }
    for (int g = 0; g < ng; ++g)
        for (int j = 0; j < c; ++j) {
            double m = -INFINITY;
            for (int s = g * grp; s < g * grp + grp && s < t; ++s) m = dmax(m, h[(size_t)s * c + j]);
            hmax[(size_t)g * c + j] = m;
        }
    const int reach = 2 * grp + grp - 1 + 127;    /* n - witness <= 2 grp + (grp - 1) + 127 */
    for (int j = 0; j < c; ++j) {
        double lmin = INFINITY;
        for (int k = 2 * grp + 1; k <= reach && k <= kp - 1; ++k) lmin = dmin(lmin, len[(size_t)k * c + j]);
        if (reach > kp - 1) lmin = -INFINITY;
        for (int m = 1; m <= 8; ++m) {
            double lm = -INFINITY;
            for (int k = 16 + 112 * m; k <= 127 + 112 * m && k <= kp - 1; ++k) lm = dmax(lm, len[(size_t)k * c + j]);
            int was = 0;
            for (int g = 0; g * grp < t; ++g) {
                const int s0 = g * grp - 112 * m;         /* first source of the band's group (may straddle two aligned groups) */
                if (s0 + grp - 1 < 0) continue;
                const int ga = (s0 < 0 ? 0 : s0) / grp, gb = (s0 + grp - 1) / grp;
                double hm = dmax(hmax[(size_t)ga * c + j], hmax[(size_t)gb * c + j]);
                const double lb = (g >= 2) ? hmax[(size_t)(g - 2) * c + j] + lmin : -INFINITY;
                out[0] += 1.0;
                const int act = hm + lm > lb;
                if (act) {
                    out[1] += 1.0;
                    out[4] += 112.0 * grp;
                    if (!was) out[2] += 1.0;
                }
                was = act;
            }
        }
    }
    free(cum); free(h); free(gam); free(hmax);
    return 0;
}

/* Round 4: SOURCE DOMINANCE inside band 0.  The pushers push the 8 sources of a hand-over block into the ring with the
 * lengths 9..127.  Candidate (s, k) and candidate (s + 1, k - 1) aim at the same target, so source s need not be pushed
 * for state c when  h[s+1][c] - h[s][c] >= X_c = max_{9 <= k <= 127} (len[k][c] - len[k-1][c])  (its successor beats it
 * at every target, and the successor's candidate is evaluated by a pusher or -- k - 1 = 8 -- by the chain wave); the
 * relation is transitive along the block and the block's last source is always pushed.
 * out[0] = (state, block) pairs, out[1] = sources pushed with the successor test, out[2] = sources in all,
 * out[3] = (state, block) pairs that push only the last source, out[4] = blocks, out[5] = sum over blocks of the
 * largest per-state push count (the leader), out[6] = pushed with the test against EVERY later source of the block
 * (X_c(d) = max_k len[k] - len[k-d]).
 */
int smm_dom_probe(const double *elp, int t, int c, const double *trans, const double *init, const double *len, int kp,
                  double *out)
{
    double *cum = (double *)calloc((size_t)(t + 1) * c, sizeof(double));
    double *h = (double *)malloc(sizeof(double) * (size_t)(t + 1) * c);
    double *gam = (double *)malloc(sizeof(double) * c);
    if (!cum || !h || !gam) return -1;
    memset(out, 0, sizeof(double) * 8);
    for (int j = 0; j < c; ++j) h[j] = init[j];
    for (int n = 1; n <= t; ++n) {
        const int kmax = (kp - 1 < n) ? kp - 1 : n;
        for (int j = 0; j < c; ++j) {
            cum[(size_t)n * c + j] = cum[(size_t)(n - 1) * c + j] + elp[(size_t)(n - 1) * c + j];
            double a = -INFINITY;
            for (int k = 1; k <= kmax; ++k) a = dmax(a, h[(size_t)(n - k) * c + j] + len[(size_t)k * c + j]);
            gam[j] = cum[(size_t)n * c + j] + a;
        }
        for (int to = 0; to < c; ++to) {
            double bt = -INFINITY;
            for (int j = 0; j < c; ++j) bt = dmax(bt, gam[j] + trans[(size_t)to * c + j]);
            h[(size_t)n * c + to] = bt - cum[(size_t)n * c + to];
        }
    }
    const int khi = (kp - 1 < 127) ? kp - 1 : 127;
    for (int j0 = 0; j0 * 8 + 1 <= t; ++j0) {
        int worst = 0;
        out[4] += 1.0;
        for (int j = 0; j < c; ++j) {
            double x[8];
            for (int d = 1; d < 8; ++d) {
                x[d] = -INFINITY;
                for (int k = 9; k <= khi; ++k) {
                    if (len[(size_t)k * c + j] == -INFINITY) continue;
                    x[d] = dmax(x[d], len[(size_t)k * c + j] - len[(size_t)(k - d) * c + j]);
                }
                if (x[d] < 0) x[d] = 0;
            }
            int pushed = 0, pushed_any = 0, nsrc = 0;
            for (int i = 0; i < 8; ++i) {
                const int s = j0 * 8 + 1 + i;
                if (s > t) break;
                ++nsrc;
                int dom = 0, dom_any = 0;
                if (i < 7 && s + 1 <= t) dom = h[(size_t)(s + 1) * c + j] - h[(size_t)s * c + j] > x[1];
                for (int d = 1; i + d < 8 && s + d <= t; ++d)
                    if (h[(size_t)(s + d) * c + j] - h[(size_t)s * c + j] > x[d]) dom_any = 1;
                pushed += !dom;
                pushed_any += !dom_any;
            }
            out[0] += 1.0; out[1] += pushed; out[2] += nsrc; out[6] += pushed_any;
            if (pushed <= 1) out[3] += 1.0;
            if (pushed > worst) worst = pushed;
        }
        out[5] += worst;
    }
    free(cum); free(h); free(gam);
    return 0;
}

/* Round 4: the band skip test with EVERY complete group as a witness.  smm_band_probe2 (the kernel of round 3) bounds the
 * final A[n] of the targets a band-group can reach from below with ONE witness, the best source of group g - 2; here the
 * lower bound is the best of all groups g - delta, delta = 2..55 (the best source s* of group g - delta is a real candidate
 * of every target n the band-group reaches, with 16 delta + 1 <= n - s* <= 16 delta + 142), and a band-group is skipped only
 * when its upper bound is STRICTLY below it.  Groups of 16 sources, bands as built (16 + 112 m .. 127 + 112 m, sources
 * delayed by 112 m).  out[0] = band-groups with sources (m >= 1), out[1] = evaluated with the round-3 test,
 * out[2] = evaluated with every witness, out[3] = frames.
 */
int smm_band_probe3(const double *elp, int t, int c, const double *trans, const double *init, const double *len, int kp,
                    double *out)
{
    const int grp = 16;
    double *cum = (double *)calloc((size_t)(t + 1) * c, sizeof(double));
    double *h = (double *)malloc(sizeof(double) * (size_t)(t + 1) * c);
    double *gam = (double *)malloc(sizeof(double) * c);
    const int ng = t / grp + 2;
    double *hmax = (double *)malloc(sizeof(double) * (size_t)ng * c);
    if (!cum || !h || !gam || !hmax) return -1;
    memset(out, 0, sizeof(double) * 8);
    for (int j = 0; j < c; ++j) h[j] = init[j];
    for (int n = 1; n <= t; ++n) {
        const int kmax = (kp - 1 < n) ? kp - 1 : n;
        for (int j = 0; j < c; ++j) {
            cum[(size_t)n * c + j] = cum[(size_t)(n - 1) * c + j] + elp[(size_t)(n - 1) * c + j];
            double a = -INFINITY;
            for (int k = 1; k <= kmax; ++k) a = dmax(a, h[(size_t)(n - k) * c + j] + len[(size_t)k * c + j]);
            gam[j] = cum[(size_t)n * c + j] + a;
        }
        for (int to = 0; to < c; ++to) {
            double bt = -INFINITY;
            for (int j = 0; j < c; ++j) bt = dmax(bt, gam[j] + trans[(size_t)to * c + j]);
            h[(size_t)n * c + to] = bt - cum[(size_t)n * c + to];
        }
    }
    /* group g = sources 16 g + 1 .. 16 g + 16 (the kernel's D = 0 convention); position 0 counts into "group -1" */
    for (int g = 0; g < ng; ++g)
        for (int j = 0; j < c; ++j) {
            double m = -INFINITY;
            for (int s = g * grp + 1; s <= g * grp + grp && s <= t; ++s) m = dmax(m, h[(size_t)s * c + j]);
            hmax[(size_t)g * c + j] = m;
        }
    out[3] = t;
    for (int j = 0; j < c; ++j) {
        double mlw[56];
        for (int d = 2; d <= 55; ++d) {
            double mn = INFINITY;
            const int k0 = 16 * d + 1, k1 = 16 * d + 142;
            if (k1 > kp - 1) mn = -INFINITY;
            else for (int k = k0; k <= k1; ++k) mn = dmin(mn, len[(size_t)k * c + j]);
            mlw[d] = mn;
        }
        for (int m = 1; m <= 8; ++m) {
            if (16 + 112 * m > kp - 1) continue;
            double lm = -INFINITY;
            for (int k = 16 + 112 * m; k <= 127 + 112 * m && k <= kp - 1; ++k) lm = dmax(lm, len[(size_t)k * c + j]);
            for (int g = 0; g * grp < t; ++g) {
                const int gs = g - 7 * m;                      /* the band's source group */
                if (gs < -1) continue;
                const double hm = (gs == -1) ? h[j] : hmax[(size_t)gs * c + j];
                const double ub = hm + lm;
                out[0] += 1.0;
                const double lb2 = (g >= 2) ? hmax[(size_t)(g - 2) * c + j] + mlw[2] : ((g == 1) ? h[j] + mlw[2] : -INFINITY);
                const int on3 = ub > lb2;
                double lb = -INFINITY;
                for (int d = 2; d <= 55; ++d) {
                    const int gw = g - d;
                    if (gw < -1) break;
                    const double hw = (gw == -1) ? h[j] : hmax[(size_t)gw * c + j];
                    lb = dmax(lb, hw + mlw[d]);
                }
                const int on4 = on3 && !(ub < lb);
                out[1] += on3;
                out[2] += on4;
            }
        }
    }
    free(cum); free(h); free(gam); free(hmax);
    return 0;
}

/* ------------------------------------------------------------------------------------------------ round 5: rank convergence
 * Go / no-go probe for parallelism INSIDE one video.  A (max,+) recurrence forgets its start: run the
 * forward recursion from position a with a GUESSED ring -- beta[s][c] = 0 for every ring position s <= a and state c --
 * and the values it produces differ from the true ones by one constant (over states and positions) from some n0 on; a
 * chunk of the time axis could then be decoded from a guess after n0 - a positions of warm-up.
 * true_beta: beta[n][c] of the real forward pass, n = 0 .. T-1 (row 0 unused), computed here first.
 * For each of the n_starts positions a: first n0 >= a + 1 such that for every n in [n0, T-1] and every state c
 *   | (beta_g[n][c] - beta[n][c]) - delta | <= tol,   delta = the difference at (T-1, state 0).
 * out[i] = n0 - a  (T - a if it never converges before the end).  Same expressions as smm_oracle.c's forward pass. */
#ifdef _OPENMP
#include <omp.h>
#endif
static void conv_forward(const double *elp, int t, int c, const double *trans, const double *len, int kp, const double *cum,
                         double *h, double *beta, int n_from)
{
    /* h[s][c] valid for s < n_from (at least the last kp-1 of them); fills h, beta for n = n_from .. t-1 */
    double *gam = (double *)malloc(sizeof(double) * c);
    (void)elp;
    for (int n = n_from; n < t; ++n) {
        const int kmax = (kp - 1 < n) ? kp - 1 : n;
        for (int j = 0; j < c; ++j) {
            double a = -INFINITY;
            for (int k = 1; k <= kmax; ++k) {
                const double v = h[(size_t)(n - k) * c + j] + len[(size_t)k * c + j];
                a = v > a ? v : a;
            }
            gam[j] = cum[(size_t)n * c + j] + a;
        }
        for (int to = 0; to < c; ++to) {
            double bt = -INFINITY;
            for (int j = 0; j < c; ++j) {
                const double v = gam[j] + trans[(size_t)to * c + j];
                bt = v > bt ? v : bt;
            }
            beta[(size_t)n * c + to] = bt;
            h[(size_t)n * c + to] = bt - cum[(size_t)n * c + to];
        }
    }
    free(gam);
}

/* mode 0: the flat ring (beta = 0 at every ring position up to a); mode 1: a BOUNDARY forced at a with a uniform start --
 * h[a][c] = 0 for every state, nothing older in the ring (what a chunk of a time-split decode starts from: "the video
 * begins at a") */
int smm_conv_probe_mode(const double *elp, int t, int c, const double *trans, const double *init, const double *len, int kp,
                        const int32_t *starts, int n_starts, double tol, int mode, int32_t *out);
int smm_conv_probe(const double *elp, int t, int c, const double *trans, const double *init, const double *len, int kp,
                   const int32_t *starts, int n_starts, double tol, int32_t *out)
{
    return smm_conv_probe_mode(elp, t, c, trans, init, len, kp, starts, n_starts, tol, 0, out);
}
int smm_conv_probe_mode(const double *elp, int t, int c, const double *trans, const double *init, const double *len, int kp,
                        const int32_t *starts, int n_starts, double tol, int mode, int32_t *out)
{
    double *cum = (double *)malloc(sizeof(double) * (size_t)(t + 1) * c);
    double *h = (double *)malloc(sizeof(double) * (size_t)t * c);
    double *beta = (double *)calloc((size_t)t * c, sizeof(double));
    if (!cum || !h || !beta) return -1;
    for (int j = 0; j < c; ++j) { cum[j] = 0.0; h[j] = init[j]; }
    for (int n = 1; n <= t; ++n)
        for (int j = 0; j < c; ++j) cum[(size_t)n * c + j] = cum[(size_t)(n - 1) * c + j] + elp[(size_t)(n - 1) * c + j];
    conv_forward(elp, t, c, trans, len, kp, cum, h, beta, 1);
    #pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < n_starts; ++i) {
        const int a = starts[i];
        double *hg = (double *)malloc(sizeof(double) * (size_t)t * c);
        double *bg = (double *)calloc((size_t)t * c, sizeof(double));
        /* the guessed ring: beta = 0 at every position up to a (positions before a - kp + 1 are never read) */
        for (int s = (a - kp + 1 > 0 ? a - kp + 1 : 0); s <= a; ++s)
            for (int j = 0; j < c; ++j) hg[(size_t)s * c + j] = mode == 0 ? 0.0 - cum[(size_t)s * c + j] : (s == a ? 0.0 : -INFINITY);
        conv_forward(elp, t, c, trans, len, kp, cum, hg, bg, a + 1);
        const double delta = bg[(size_t)(t - 1) * c] - beta[(size_t)(t - 1) * c];
        int n0 = t;
        for (int n = t - 1; n > a; --n) {
            int ok = 1;
            for (int j = 0; j < c; ++j) {
                const double d = (bg[(size_t)n * c + j] - beta[(size_t)n * c + j]) - delta;
                if (!(d <= tol && d >= -tol)) { ok = 0; break; }
            }
            if (!ok) break;
            n0 = n;
        }
        out[i] = n0 - a;
        free(hg); free(bg);
    }
    free(cum); free(h); free(beta);
    return 0;
}

/* Round 5: ANCHOR dominance in band 0, on top of the successor test of smm_dom_probe.  A source s of state c is also left
 * out when an OLDER source a ("anchor": the last source that was not itself dominated this way; position 0 at first)
 * beats it STRICTLY at every target band 0 reaches: candidate (a, k + s - a) exists for every band-0 length k (k + s - a
 * <= kp - 1) and  h[s][c] - h[a][c] < D_c[s - a] = min_{9 <= k <= min(127, kp-1)} (len[k + s - a][c] - len[k][c])
 * (deflated by 2^-49 relative).  Inside a long segment the segment's START is such an anchor for every later source.
 * The block's last source is still always pushed (the successor test's chain ends there).
 * out[0] = (state, block) pairs, out[1] = pushed with the successor test only, out[2] = pushed with both tests,
 * out[3] = blocks, out[4] = sum over blocks of the largest per-state push count, successor test only, out[5] = the same
 * with both tests, out[6] = anchor changes. */
int smm_anchor_probe(const double *elp, int t, int c, const double *trans, const double *init, const double *len, int kp,
                     double *out)
{
    double *cum = (double *)calloc((size_t)(t + 1) * c, sizeof(double));
    double *h = (double *)malloc(sizeof(double) * (size_t)(t + 1) * c);
    double *gam = (double *)malloc(sizeof(double) * c);
    double *dmn = (double *)malloc(sizeof(double) * (size_t)kp * c);
    if (!cum || !h || !gam || !dmn) return -1;
    memset(out, 0, sizeof(double) * 8);
    for (int j = 0; j < c; ++j) h[j] = init[j];
    for (int n = 1; n <= t; ++n) {
        const int kmax = (kp - 1 < n) ? kp - 1 : n;
        for (int j = 0; j < c; ++j) {
            cum[(size_t)n * c + j] = cum[(size_t)(n - 1) * c + j] + elp[(size_t)(n - 1) * c + j];
            double a = -INFINITY;
            for (int k = 1; k <= kmax; ++k) a = dmax(a, h[(size_t)(n - k) * c + j] + len[(size_t)k * c + j]);
            gam[j] = cum[(size_t)n * c + j] + a;
        }
        for (int to = 0; to < c; ++to) {
            double bt = -INFINITY;
            for (int j = 0; j < c; ++j) bt = dmax(bt, gam[j] + trans[(size_t)to * c + j]);
            h[(size_t)n * c + to] = bt - cum[(size_t)n * c + to];
        }
    }
    const int khi = (kp - 1 < 127) ? kp - 1 : 127;
    for (int j = 0; j < c; ++j)
        for (int d = 1; d < kp; ++d) {
            double m = INFINITY;
            if (khi + d > kp - 1) m = -INFINITY;
            else
                for (int k = 9; k <= khi; ++k) {
                    if (len[(size_t)k * c + j] == -INFINITY) continue;
                    m = dmin(m, len[(size_t)(k + d) * c + j] - len[(size_t)k * c + j]);
                }
            dmn[(size_t)d * c + j] = m > 0 ? m * (1.0 - 0x1p-49) : m * (1.0 + 0x1p-49);
        }
    int *anc = (int *)calloc(c, sizeof(int));
    for (int j0 = 0; j0 * 8 + 1 <= t; ++j0) {
        int worst1 = 0, worst2 = 0;
        out[3] += 1.0;
        for (int j = 0; j < c; ++j) {
            double x1 = -INFINITY;
            for (int k = 9; k <= khi; ++k) {
                if (len[(size_t)k * c + j] == -INFINITY) continue;
                x1 = dmax(x1, len[(size_t)k * c + j] - len[(size_t)(k - 1) * c + j]);
            }
            if (x1 < 0) x1 = 0;
            int p1 = 0, p2 = 0;
            for (int i = 0; i < 8; ++i) {
                const int s = j0 * 8 + 1 + i;
                if (s > t) break;
                const int last = (i == 7 || s == t);
                int dom = 0;
                if (!last) dom = h[(size_t)(s + 1) * c + j] - h[(size_t)s * c + j] > x1;
                const int a = anc[j], d = s - a;
                int adom = 0;
                if (d >= 1 && d < kp) adom = h[(size_t)s * c + j] - h[(size_t)a * c + j] < dmn[(size_t)d * c + j];
                if (!adom) { anc[j] = s; out[6] += 1.0; }
                p1 += last || !dom;
                p2 += last || (!dom && !adom);
            }
            out[0] += 1.0; out[1] += p1; out[2] += p2;
            if (p1 > worst1) worst1 = p1;
            if (p2 > worst2) worst2 = p2;
        }
        out[4] += worst1; out[5] += worst2;
    }
    free(cum); free(h); free(gam); free(dmn); free(anc);
    return 0;
}
