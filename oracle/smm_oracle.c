/*
 * oracle/smm_oracle.c -- plain-C fp64 restatement of the FACTORED semi-Markov DP.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the checker / the
 * reported CPU number.  The product never links or loads it.
 *
 * What it restates.  The reference builds dense potentials
 *   scores[n,k,to,from] = trans[to,from] + [n==0]*init[from] + len[k,from] + sum_{j=n}^{n+k-1} em[j,from] (+EOS terms)
 * (/root/reference/src/models/semimarkov/semimarkov_modules.py:416-523) and scans them with
 * pytorch-struct's SemiMarkov._dp (pinned commit 1c9b038a; call sites modules:624,657,677-679):
 *   beta[n][to] = plus_k plus_from ( beta[n-k][from] times scores[n-k,k,to,from] ).
 * Because scores factor as trans[to,from] + f(n,k,from), the same optimum is
 *   cumE[n][c] = sum_{j<n} elp[j][c]                       (sequential fp64 prefix sum)
 *   h[0][c]    = init[c];    h[s][c] = beta[s][c] - cumE[s][c]           (s >= 1)
 *   A[n][c]    = plus_{k=1..min(Kp-1,n)} ( h[n-k][c] + len[k][c] )
 *   gamma[n][c]= cumE[n][c] + A[n][c]
 *   beta[n][to]= plus_c ( gamma[n][c] + trans[to][c] )                   (1 <= n < T)
 *   fin[EOS]   = plus_c ( gamma[T][c] + endpen[c] ),  endpen = 0 / -1e9  (modules:462-471)
 *   fin[to]    = plus_c ( gamma[T][c] + trans[to][c] ) + (-1e9)          (em+[T][to], modules:485-489)
 * with Kp = min(K, Tmax) (modules:450-452).  Every "+" above is ONE IEEE fp64 add in exactly this
 * association (compile with -ffp-contract=off); the HIP kernels use the same expressions, so the
 * Viterbi outputs of the two are comparable bit for bit.
 *
 * Arg-max order (torch.max returns the first maximal index; _dp stacks k ascending): at a span start
 * (n,to) the chosen predecessor is the first (k ascending, then from ascending) whose
 *   (cumE[n][from] + (h[n-k][from] + len[k][from])) + w(to,from)   equals the maximum.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SMM_BIG_NEG (-1e9)

static inline double dmax(double a, double b) { return a > b ? a : b; }

/* OpenMP threads the batch loops below run on (bench.py's cpu_factored leg reports the number it used). */
#ifdef _OPENMP
#include <omp.h>
int smm_oracle_set_threads(int n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }
#else
int smm_oracle_set_threads(int n) { (void)n; return 1; }
#endif

/* ---------------------------------------------------------------- emission */
/* elp[i,t,c] = lognorm - 0.5 * sum_d (x-mu)^2 * inv_var + cons ; semimarkov_modules.py:324-381 */
void smm_oracle_emission(const float *x, const int64_t *lengths, const double *mu, const double *inv_var,
                         double lognorm, const double *cons, double *elp, int b, int tmax, int d, int c)
{
    #pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < b; ++i) {
        for (int t = 0; t < tmax; ++t) {
            const float *xr = x + ((size_t)i * tmax + t) * d;
            for (int j = 0; j < c; ++j) {
                double q = 0.0;
                for (int e = 0; e < d; ++e) {
                    double z = (double)xr[e] - mu[(size_t)j * d + e];
                    q += z * z * inv_var[e];
                }
                double v = lognorm - 0.5 * q;
                if (cons) v += cons[((size_t)i * tmax + t) * c + j];
                elp[((size_t)i * tmax + t) * c + j] = v;
            }
        }
    }
}

/* ---------------------------------------------------------------- Viterbi */
/*
 * elp b x tmax x c (fp64), lengths b, trans c x c [to][from], init c, len kp x c, endpen nullable b x c.
 * spans out: b x (tmax+1) int64, local ids, EOS = c, -1 = continuation / beyond the end.
 * v out: b.  Returns 0, or -1 on allocation failure.
 */
/* no_eos != 0: add_eos=False of the reference (semimarkov_modules.py:494-505, positions = frames): the segments cover
 * frames 0 .. T-2 and the video closes with a transition into the label of frame T-1, which only emits:
 *   fin[to] = max_c ( gamma[T-1][c] + trans[to][c] ) + elp[T-1][to],   no EOS label, no end penalties. */
int smm_oracle_viterbi_ex(const double *elp, const int64_t *lengths, const double *trans, const double *init,
                          const double *len, const double *endpen, int b, int tmax, int c, int kp, int no_eos,
                          int64_t *spans, double *v)
{
    int rc = 0;
    #pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < b; ++i) {
        const int t_i = (int)lengths[i] - (no_eos ? 1 : 0);
        const double *e = elp + (size_t)i * tmax * c;
        int64_t *sp = spans + (size_t)i * (tmax + 1);
        for (int n = 0; n <= tmax; ++n) sp[n] = -1;
        double *cum = (double *)malloc(sizeof(double) * (size_t)(t_i + 1) * c);
        double *h = (double *)malloc(sizeof(double) * (size_t)(t_i + 1) * c);
        double *gam = (double *)malloc(sizeof(double) * c);
        if (!cum || !h || !gam) { rc = -1; free(cum); free(h); free(gam); continue; }
        for (int j = 0; j < c; ++j) { cum[j] = 0.0; h[j] = init[j]; }
        for (int n = 1; n <= t_i; ++n) {
            const int kmax = (kp - 1 < n) ? kp - 1 : n;
            for (int j = 0; j < c; ++j) {
                cum[(size_t)n * c + j] = cum[(size_t)(n - 1) * c + j] + e[(size_t)(n - 1) * c + j];
                double a = -INFINITY;
                for (int k = 1; k <= kmax; ++k)
                    a = dmax(a, h[(size_t)(n - k) * c + j] + len[(size_t)k * c + j]);
                gam[j] = cum[(size_t)n * c + j] + a;
            }
            if (n < t_i) {
                for (int to = 0; to < c; ++to) {
                    double bt = -INFINITY;
                    for (int j = 0; j < c; ++j) bt = dmax(bt, gam[j] + trans[(size_t)to * c + j]);
                    h[(size_t)n * c + to] = bt - cum[(size_t)n * c + to];
                }
            }
        }
        /* last position: first maximal entry of [fin[0..c-1], fin[EOS]] */
        int best_to = -1;
        double best = -INFINITY;
        for (int to = 0; to <= c - (no_eos ? 1 : 0); ++to) {
            double f = -INFINITY;
            for (int j = 0; j < c; ++j) {
                double w = (to == c) ? (endpen ? endpen[(size_t)i * c + j] : 0.0) : trans[(size_t)to * c + j];
                f = dmax(f, gam[j] + w);
            }
            if (no_eos) f = f + e[(size_t)t_i * c + to];
            else if (to < c) f = f + SMM_BIG_NEG;
            if (best_to < 0 || f > best) { best = f; best_to = to; }
        }
        v[i] = best;
        /* back-trace */
        int n = t_i, to = best_to;
        sp[t_i] = to;
        while (n > 0) {
            const int kmax = (kp - 1 < n) ? kp - 1 : n;
            double m = -INFINITY;
            for (int k = 1; k <= kmax; ++k)
                for (int j = 0; j < c; ++j) {
                    double w = (to == c) ? (endpen ? endpen[(size_t)i * c + j] : 0.0) : trans[(size_t)to * c + j];
                    double val = (cum[(size_t)n * c + j] + (h[(size_t)(n - k) * c + j] + len[(size_t)k * c + j])) + w;
                    m = dmax(m, val);
                }
            int bk = -1, bj = -1;
            for (int k = 1; k <= kmax && bk < 0; ++k)
                for (int j = 0; j < c; ++j) {
                    double w = (to == c) ? (endpen ? endpen[(size_t)i * c + j] : 0.0) : trans[(size_t)to * c + j];
                    double val = (cum[(size_t)n * c + j] + (h[(size_t)(n - k) * c + j] + len[(size_t)k * c + j])) + w;
                    if (val == m) { bk = k; bj = j; break; }
                }
            if (bk < 0) { rc = -2; break; }   /* NaN in the inputs */
            n -= bk;
            to = bj;
            sp[n] = bj;
        }
        free(cum); free(h); free(gam);
    }
    return rc;
}

int smm_oracle_viterbi(const double *elp, const int64_t *lengths, const double *trans, const double *init,
                       const double *len, const double *endpen, int b, int tmax, int c, int kp,
                       int64_t *spans, double *v)
{
    return smm_oracle_viterbi_ex(elp, lengths, trans, init, len, endpen, b, tmax, c, kp, 0, spans, v);
}

/* ---------------------------------------------------------------- log-partition + posteriors */
static inline double lse2(double a, double b)
{
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    double m = a > b ? a : b;
    return m + log(exp(a - m) + exp(b - m));
}

/*
 * Forward (LogSemiring) and, when any gradient pointer is non-NULL, the exact backward:
 *   g_elp  b x tmax x c   d sum_i gl[i]*logZ_i / d elp      (posterior state occupancy of each frame)
 *   g_trans c x c, g_init c, g_len kp x c                    (summed over the batch)
 * gl (nullable) = upstream gradient per instance (default 1).
 */
int smm_oracle_logz(const double *elp, const int64_t *lengths, const double *trans, const double *init,
                    const double *len, const double *endpen, const double *gl, int b, int tmax, int c, int kp,
                    double *logz, double *g_elp, double *g_trans, double *g_init, double *g_len)
{
    const int want_grad = g_elp || g_trans || g_init || g_len;
    if (g_elp) memset(g_elp, 0, sizeof(double) * (size_t)b * tmax * c);
    if (g_trans) memset(g_trans, 0, sizeof(double) * (size_t)c * c);
    if (g_init) memset(g_init, 0, sizeof(double) * (size_t)c);
    if (g_len) memset(g_len, 0, sizeof(double) * (size_t)kp * c);
    for (int i = 0; i < b; ++i) {
        const int t_i = (int)lengths[i];
        const double *e = elp + (size_t)i * tmax * c;
        const double *ep = endpen ? endpen + (size_t)i * c : NULL;
        size_t sz = (size_t)(t_i + 1) * c;
        double *cum = (double *)calloc(sz, sizeof(double));
        double *start = (double *)calloc(sz, sizeof(double)); /* log-weight of "a span of c starts at s" (incl. init at 0) */
        double *gam = (double *)calloc(sz, sizeof(double));   /* log-weight of "a span of c ends at n" */
        if (!cum || !start || !gam) { free(cum); free(start); free(gam); return -1; }
        for (int j = 0; j < c; ++j) start[j] = init[j];
        for (int n = 1; n <= t_i; ++n) {
            const int kmax = (kp - 1 < n) ? kp - 1 : n;
            for (int j = 0; j < c; ++j) {
                cum[(size_t)n * c + j] = cum[(size_t)(n - 1) * c + j] + e[(size_t)(n - 1) * c + j];
                double a = -INFINITY;
                for (int k = 1; k <= kmax; ++k)
                    a = lse2(a, start[(size_t)(n - k) * c + j] + len[(size_t)k * c + j]
                                + (cum[(size_t)n * c + j] - cum[(size_t)(n - k) * c + j]));
                gam[(size_t)n * c + j] = a;
            }
            if (n < t_i)
                for (int to = 0; to < c; ++to) {
                    double bt = -INFINITY;
                    for (int j = 0; j < c; ++j) bt = lse2(bt, gam[(size_t)n * c + j] + trans[(size_t)to * c + j]);
                    start[(size_t)n * c + to] = bt;
                }
        }
        /* v = logsumexp over ALL last-position labels (torch_struct sums beta[len-1] over C incl. EOS) */
        double z = -INFINITY;
        for (int to = 0; to <= c; ++to) {
            double f = -INFINITY;
            for (int j = 0; j < c; ++j) {
                double w = (to == c) ? (ep ? ep[j] : 0.0) : trans[(size_t)to * c + j] + SMM_BIG_NEG;
                f = lse2(f, gam[(size_t)t_i * c + j] + w);
            }
            z = lse2(z, f);
        }
        logz[i] = z;
        if (want_grad) {
            /* bwd_end[n][j]: log-weight of everything after a span of j ended at n */
            double *bend = (double *)calloc(sz, sizeof(double));
            double *bstart = (double *)calloc(sz, sizeof(double));
            double *occ = (double *)calloc((size_t)(t_i + 2) * c, sizeof(double)); /* difference array over frames */
            const double up = gl ? gl[i] : 1.0;
            for (int j = 0; j < c; ++j) {
                double w = ep ? ep[j] : 0.0;
                double f = w;
                for (int to = 0; to < c; ++to) f = lse2(f, trans[(size_t)to * c + j] + SMM_BIG_NEG);
                bend[(size_t)t_i * c + j] = f;
            }
            for (int s = t_i - 1; s >= 0; --s) {
                for (int j = 0; j < c; ++j) {
                    double a = -INFINITY;
                    for (int k = 1; k <= kp - 1 && s + k <= t_i; ++k)
                        a = lse2(a, len[(size_t)k * c + j] + (cum[(size_t)(s + k) * c + j] - cum[(size_t)s * c + j])
                                    + bend[(size_t)(s + k) * c + j]);
                    bstart[(size_t)s * c + j] = a;
                }
                if (s > 0)
                    for (int j = 0; j < c; ++j) {
                        double a = -INFINITY;
                        for (int to = 0; to < c; ++to) a = lse2(a, trans[(size_t)to * c + j] + bstart[(size_t)s * c + to]);
                        bend[(size_t)s * c + j] = a;
                    }
            }
            for (int s = 0; s < t_i; ++s)
                for (int j = 0; j < c; ++j) {
                    if (s == 0 && g_init) g_init[j] += up * exp(start[j] + bstart[j] - z);
                    for (int k = 1; k <= kp - 1 && s + k <= t_i; ++k) {
                        double p = up * exp(start[(size_t)s * c + j] + len[(size_t)k * c + j]
                                            + (cum[(size_t)(s + k) * c + j] - cum[(size_t)s * c + j])
                                            + bend[(size_t)(s + k) * c + j] - z);
                        if (g_len) g_len[(size_t)k * c + j] += p;
                        occ[(size_t)s * c + j] += p;
                        occ[(size_t)(s + k) * c + j] -= p;
                    }
                }
            if (g_trans)
                for (int n = 1; n < t_i; ++n)
                    for (int to = 0; to < c; ++to)
                        for (int j = 0; j < c; ++j)
                            g_trans[(size_t)to * c + j] += up * exp(gam[(size_t)n * c + j] + trans[(size_t)to * c + j]
                                                                    + bstart[(size_t)n * c + to] - z);
            if (g_elp)
                for (int j = 0; j < c; ++j) {
                    double run = 0.0;
                    for (int t = 0; t < t_i; ++t) {
                        run += occ[(size_t)t * c + j];
                        g_elp[((size_t)i * tmax + t) * c + j] = run;
                    }
                }
            free(bend); free(bstart); free(occ);
        }
        free(cum); free(start); free(gam);
    }
    return 0;
}
