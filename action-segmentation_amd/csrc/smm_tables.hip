// smm_tables.hip -- the factor tables of every parameter group from the model parameters, and their chain rule.
//
// Replaces, for training steps, the ~45 small differentiable torch ops (and as many in backward) that build
//   initial_log_probs    (reference src/models/semimarkov/semimarkov_modules.py:284-296)
//   transition_log_probs (:298-322)   masks before the softmax, every column normalised over the valid `to`
//   _length_log_probs_with_rates (:383-414)   Poisson(rate).log_prob(k), row == length
//   emission_log_probs in expanded form (:324-381)   w = mu / var, cst = -0.5 sum mu^2/var - 0.5 sum log var - D/2 log 2pi
// for every class set of a launch: a training step over many tasks spent most of its host time dispatching them
// (cfg4: 1.5 of 3.8 ms).  Two launches instead: this is launch-latency-bound work on a few KB of parameters.
//
// Layouts as the DP / emission kernels read them (include/smmdp.h): trans [g][c_max][c_max] ([to][from]),
// init [g][c_max], len [g][k_rows][c_max], w [g][d][c_max], cst [g][c_max], inv_var [d]; columns past a group's state
// count are 0.  Parameters fp32 (the reference's nn.Parameters), tables fp64.
#include "../../include/smmdp.h"
#include "smm_device.h"
#include "smm_launch.h"

#define SMM_TAB_LEN_ROWS 64      // length-table rows per workgroup

struct SmmTabArgs {
    const float *init_logits;        // [n]
    const float *trans_logits;       // [n][n]  [to][from]
    const float *log_rates;          // [n]
    const float *means;              // [n][d]
    const float *cov;                // [d][d]  (diagonal used: tied diagonal covariance)
    const uint8_t *init_cons;        // [n] or null; 1 = forbidden
    const uint8_t *trans_cons;       // [n][n] or null
    const int64_t *classes;          // [g][c_max] class id of each local state
    const int64_t *merged;           // [g][c_max] parameter row of each local state (merge_classes applied)
    const int32_t *n_states;         // [g]
    double *trans, *init, *len, *w, *cst, *inv_var;
    // backward (null in the forward launch)
    const double *g_trans, *g_init, *g_len, *g_w_cm, *g_cst;    // g_w_cm: CLASS-major [g][c_max][d]
    double *g_init_logits, *g_trans_logits, *g_log_rates, *g_means;
    int32_t n, d, g, cm, k_rows, allow_self;
};

__device__ __forceinline__ double smm_wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

__device__ __forceinline__ double smm_wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ void smm_tab_atomic(double *p, double v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// grid (groups, 2 + ceil(k_rows / 64)): y = 0 initial + transitions, y = 1 emission factors, y >= 2 a slab of length rows
template <bool BWD>
__global__ void __launch_bounds__(256) smm_tables_kernel(SmmTabArgs a)
{
    const int g = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
    const int N = a.n, D = a.d, cm = a.cm;
    const int C = a.n_states[g];
    const int64_t *vc = a.classes + (size_t)g * cm;
    const int64_t *mv = a.merged + (size_t)g * cm;
    if (part == 0) {
        // ---- initial: log_softmax over the valid states (one wave; c_max <= 32)
        if (tid < 64) {
            const bool on = tid < C;
            const int cls = on ? (int)vc[tid] : 0;
            const bool masked = on && a.init_cons && a.init_cons[cls];
            if (!BWD) {
                const double x = on ? (masked ? SMM_BIG_NEG : (double)a.init_logits[cls]) : SMM_NEG_INF;
                const double m = smm_wave_max(x);
                const double s = smm_wave_sum(on ? exp(x - m) : 0.0);
                if (tid < cm) a.init[(size_t)g * cm + tid] = on ? (x - m) - log(s) : 0.0;
            } else if (a.g_init) {
                const double gi = on ? a.g_init[(size_t)g * cm + tid] : 0.0;
                const double tot = smm_wave_sum(gi);
                if (on && !masked) smm_tab_atomic(a.g_init_logits + cls, gi - exp(a.init[(size_t)g * cm + tid]) * tot);
            }
        }
        // ---- transitions [to][from]: thread = source state, each column normalised over the valid targets
        if (tid < cm) {
            const int from = tid;
            const bool on = from < C;
            const int cf = on ? (int)vc[from] : 0;
            auto is_masked = [&](int to) {
                return (a.trans_cons && a.trans_cons[(size_t)vc[to] * N + cf]) || (!a.allow_self && to == from);
            };
            auto logit = [&](int to, bool *masked) {               // (forward only: the logits are not passed to backward)
                *masked = is_masked(to);
                return *masked ? SMM_BIG_NEG : (double)a.trans_logits[(size_t)vc[to] * N + cf];
            };
            if (!BWD) {
                double m = SMM_NEG_INF, s = 0.0;
                bool mk;
                if (on) {
                    for (int to = 0; to < C; ++to) m = fmax(m, logit(to, &mk));
                    for (int to = 0; to < C; ++to) s += exp(logit(to, &mk) - m);
                }
                const double ls = on ? log(s) : 0.0;
                for (int to = 0; to < cm; ++to)
                    a.trans[((size_t)g * cm + to) * cm + from] = (on && to < C) ? (logit(to, &mk) - m) - ls : 0.0;
            } else if (a.g_trans && on) {
                double tot = 0.0;
                for (int to = 0; to < C; ++to) tot += a.g_trans[((size_t)g * cm + to) * cm + from];
                for (int to = 0; to < C; ++to) {
                    if (is_masked(to)) continue;                    // masked_fill: no gradient into a masked logit
                    const size_t o = ((size_t)g * cm + to) * cm + from;
                    smm_tab_atomic(a.g_trans_logits + (size_t)vc[to] * N + cf, a.g_trans[o] - exp(a.trans[o]) * tot);
                }
            }
        }
    } else if (part == 1) {
        // ---- emission factors: w[d][c] = mu[c][d] / var[d];  cst[c] = -0.5 sum_d mu^2/var - 0.5 sum_d log var - D/2 log 2 pi
        if (!BWD) {
            for (int i = tid; i < D * cm; i += 256) {
                const int d = i / cm, c = i - d * cm;
                a.w[(size_t)g * D * cm + i] = c < C ? (double)a.means[(size_t)mv[c] * D + d] / (double)a.cov[(size_t)d * (D + 1)] : 0.0;
            }
            // cst: 8 slices of the feature axis per state, merged in LDS; sum_d log var over the whole workgroup
            __shared__ double s_part[8][32], s_lv[4];
            {
                double lv = 0.0;
                for (int d = tid; d < D; d += 256) lv += log((double)a.cov[(size_t)d * (D + 1)]);
                lv = smm_wave_sum(lv);
                if ((tid & 63) == 0) s_lv[tid >> 6] = lv;
                const int c = tid & 31, r = tid >> 5;
                double s = 0.0;
                if (c < C)
                    for (int d = r; d < D; d += 8) {
                        const double var = (double)a.cov[(size_t)d * (D + 1)], mu = (double)a.means[(size_t)mv[c] * D + d];
                        s += mu * mu / var;
                    }
                s_part[r][c] = s;
            }
            __syncthreads();
            if (tid < cm) {
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < 8; ++q) s += s_part[q][tid & 31];
                const double lv = (s_lv[0] + s_lv[1]) + (s_lv[2] + s_lv[3]);
                a.cst[(size_t)g * cm + tid] = tid < C ? -0.5 * s - 0.5 * lv - 0.5 * D * 1.8378770664093453 : 0.0;   // log(2 pi)
            }
            if (g == 0)
                for (int d = tid; d < D; d += 256) a.inv_var[d] = 1.0 / (double)a.cov[(size_t)d * (D + 1)];
        } else {
            for (int i = tid; i < C * D; i += 256) {                // consecutive threads = consecutive d of one class row
                const int c = i / D, d = i - c * D;
                const double var = (double)a.cov[(size_t)d * (D + 1)], mu = (double)a.means[(size_t)mv[c] * D + d];
                double v = 0.0;
                if (a.g_w_cm) v += a.g_w_cm[((size_t)g * cm + c) * D + d] / var;
                if (a.g_cst) v -= a.g_cst[(size_t)g * cm + c] * mu / var;
                smm_tab_atomic(a.g_means + (size_t)mv[c] * D + d, v);
            }
        }
    } else {
        // ---- lengths: len[k][c] = k log(rate) - rate - lgamma(k + 1), rate = exp(log_rate)   (xlogy: 0 at k == 0)
        const int k0 = (part - 2) * SMM_TAB_LEN_ROWS, k1 = min(a.k_rows, k0 + SMM_TAB_LEN_ROWS);
        if (!BWD) {
            for (int i = tid; i < (k1 - k0) * cm; i += 256) {
                const int k = k0 + i / cm, c = i % cm;
                double v = 0.0;
                if (c < C) {
                    const double rate = exp((double)a.log_rates[mv[c]]);
                    v = (k == 0 ? 0.0 : (double)k * log(rate)) - rate - lgamma((double)k + 1.0);
                }
                a.len[((size_t)g * a.k_rows + k) * cm + c] = v;
            }
        } else if (a.g_len) {
            __shared__ double s_glen[8][32];
            const int c = tid & 31, r = tid >> 5;
            double acc = 0.0;
            if (c < C) {
                const double rate = exp((double)a.log_rates[mv[c]]);
                for (int k = k0 + r; k < k1; k += 8) acc += a.g_len[((size_t)g * a.k_rows + k) * cm + c] * ((double)k - rate);
            }
            s_glen[r][c] = acc;
            __syncthreads();
            if (tid < C) {
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < 8; ++q) s += s_glen[q][tid];
                smm_tab_atomic(a.g_log_rates + mv[tid], s);
            }
        }
    }
}

static int tables_check(const smm_tables_shape *s)
{
    if (!s || s->n_classes < 1 || s->d < 1 || s->n_groups < 1 || s->c_max < 1 || s->k_rows < 2) return SMM_ERR_ARG;
    if (s->c_max > SMM_MAX_STATES_DEV) return SMM_ERR_UNSUPPORTED;
    return SMM_OK;
}

extern "C" int smm_factor_tables_f64(const smm_tables_shape *s, const float *init_logits, const float *transition_logits,
                                     const float *poisson_log_rates, const float *gaussian_means, const float *gaussian_cov,
                                     const uint8_t *init_constraints, const uint8_t *transition_constraints,
                                     const int64_t *classes, const int64_t *merged, const int32_t *n_states,
                                     double *trans, double *init, double *len_scores, double *w, double *cst,
                                     double *inv_var, void *stream)
{
    int rc = tables_check(s);
    if (rc != SMM_OK) return rc;
    if (!init_logits || !transition_logits || !poisson_log_rates || !gaussian_means || !gaussian_cov || !classes || !merged ||
        !n_states || !trans || !init || !len_scores || !w || !cst || !inv_var)
        return SMM_ERR_ARG;
    SmmTabArgs a{};
    a.init_logits = init_logits; a.trans_logits = transition_logits; a.log_rates = poisson_log_rates; a.means = gaussian_means;
    a.cov = gaussian_cov; a.init_cons = init_constraints; a.trans_cons = transition_constraints;
    a.classes = classes; a.merged = merged; a.n_states = n_states;
    a.trans = trans; a.init = init; a.len = len_scores; a.w = w; a.cst = cst; a.inv_var = inv_var;
    a.n = s->n_classes; a.d = s->d; a.g = s->n_groups; a.cm = s->c_max; a.k_rows = s->k_rows; a.allow_self = s->allow_self_transitions;
    const int parts = 2 + (s->k_rows + SMM_TAB_LEN_ROWS - 1) / SMM_TAB_LEN_ROWS;
    hipLaunchKernelGGL(smm_tables_kernel<false>, dim3(s->n_groups, parts), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SMM_OK : SMM_ERR_HIP;
}

extern "C" int smm_factor_tables_bwd_f64(const smm_tables_shape *s, const float *poisson_log_rates, const float *gaussian_means,
                                         const float *gaussian_cov, const uint8_t *init_constraints,
                                         const uint8_t *transition_constraints, const int64_t *classes, const int64_t *merged,
                                         const int32_t *n_states, const double *trans, const double *init,
                                         const double *g_trans, const double *g_init, const double *g_len,
                                         const double *g_w_class_major, const double *g_cst, double *g_init_logits,
                                         double *g_transition_logits, double *g_poisson_log_rates, double *g_gaussian_means,
                                         void *stream)
{
    int rc = tables_check(s);
    if (rc != SMM_OK) return rc;
    if (!poisson_log_rates || !gaussian_means || !gaussian_cov || !classes || !merged || !n_states || !trans || !init ||
        !g_init_logits || !g_transition_logits || !g_poisson_log_rates || !g_gaussian_means)
        return SMM_ERR_ARG;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    const size_t n = s->n_classes;
    {
        void *const zp[4] = {g_init_logits, g_transition_logits, g_poisson_log_rates, g_gaussian_means};
        const size_t zb[4] = {sizeof(double) * n, sizeof(double) * n * n, sizeof(double) * n, sizeof(double) * n * s->d};
        if (smm_zero_multi_async(zp, zb, 4, hs) != (int)hipSuccess) return SMM_ERR_HIP;
    }
    SmmTabArgs a{};
    a.log_rates = poisson_log_rates; a.means = gaussian_means; a.cov = gaussian_cov;
    a.init_cons = init_constraints; a.trans_cons = transition_constraints;
    a.classes = classes; a.merged = merged; a.n_states = n_states;
    a.trans = const_cast<double *>(trans); a.init = const_cast<double *>(init);
    a.g_trans = g_trans; a.g_init = g_init; a.g_len = g_len; a.g_w_cm = g_w_class_major; a.g_cst = g_cst;
    a.g_init_logits = g_init_logits; a.g_trans_logits = g_transition_logits; a.g_log_rates = g_poisson_log_rates;
    a.g_means = g_gaussian_means;
    a.n = s->n_classes; a.d = s->d; a.g = s->n_groups; a.cm = s->c_max; a.k_rows = s->k_rows; a.allow_self = s->allow_self_transitions;
    const int parts = 2 + (s->k_rows + SMM_TAB_LEN_ROWS - 1) / SMM_TAB_LEN_ROWS;
    hipLaunchKernelGGL(smm_tables_kernel<true>, dim3(s->n_groups, parts), dim3(256), 0, hs, a);
    return hipGetLastError() == hipSuccess ? SMM_OK : SMM_ERR_HIP;
}
