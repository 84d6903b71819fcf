// smm_dense.hip -- the reference's INNER boundary as it stands: a semiring DP over dense semi-Markov potentials.
//
// Drop-in for torch_struct.SemiMarkovCRF(log_potentials, lengths).argmax / .partition (pinned
// harvardnlp/pytorch-struct@1c9b038a, SemiMarkov._dp; reference call sites semimarkov_modules.py:624, 657, 677-679
// and src/models/test_semimarkov.py:312-314).  The factored kernels (smm_viterbi.hip, smm_logz.hip) are the fast
// path; this one exists so that code written against the dense interface (scores b x (N-1) x K x C x C, fp32) keeps
// working on the device for the small lattices that interface can hold (reference defaults: K = 20, C <= 24).
//
//   alpha[n-1][k][to] = plus_from ( beta[n-1][from] times edge[n-1, k, to, from] )
//   beta[n][to]       = plus_{k=1..min(K-1,n)} alpha[n-k][k][to]          beta[0] = one
//   v                 = plus_to beta[len-1][to]
// Max semiring: first maximal `from`, then smallest k, then smallest final `to` (torch.max order); the arg-max is
// kept as back-pointers (1 byte per (n, k, to) + 2 bytes per (n, to)) and emitted in from_parts' span encoding.
// Accumulation in fp64 (edge values are widened on load): HBM-bound, one pass over the potentials.
#include "smm_device.h"
#include "smm_launch.h"
#include "../../include/smmdp.h"

__device__ __forceinline__ double smm_lse2d(double a, double b)
{
    if (a == SMM_NEG_INF) return b;
    if (b == SMM_NEG_INF) return a;
    const double m = a > b ? a : b;
    return m + log(exp(a - m) + exp(b - m));
}

// one workgroup per instance; threads = (k, to) pairs for alpha, `to` for beta
template <bool LOG>
__global__ void __launch_bounds__(1024) smm_dense_kernel(SmmDenseArgs a)
{
    const int i = blockIdx.x;
    const int N1 = a.n1, K = a.k, C = a.c;                      // N1 = N - 1 edge positions
    const int L = (int)a.lengths[i];                            // positions of this instance (<= N1 + 1)
    const float *edge = a.edge + (size_t)i * N1 * K * C * C;
    double *alpha = a.alpha + (size_t)i * K * K * C;            // ring over n mod K: [K][K][C]
    double *beta = a.beta + (size_t)i * (N1 + 1) * C;           // [N][C]
    uint8_t *bpf = a.bp_from ? a.bp_from + (size_t)i * N1 * K * C : nullptr;
    uint16_t *bpk = a.bp_k ? a.bp_k + (size_t)i * (N1 + 1) * C : nullptr;
    const int tid = threadIdx.x, nth = blockDim.x;
    for (int c = tid; c < C; c += nth) beta[c] = 0.0;
    __syncthreads();
    for (int n = 1; n < L; ++n) {
        const float *e = edge + (size_t)(n - 1) * K * C * C;
        const double *bprev = beta + (size_t)(n - 1) * C;
        double *arow = alpha + (size_t)((n - 1) % K) * K * C;
        for (int p = tid; p < K * C; p += nth) {                // p = k * C + to
            const float *er = e + (size_t)p * C;
            double best = SMM_NEG_INF;
            int arg = 0;
            for (int f = 0; f < C; ++f) {
                const double v = bprev[f] + (double)er[f];
                if (LOG) best = smm_lse2d(best, v);
                else if (v > best) { best = v; arg = f; }
            }
            arow[p] = best;
            if (!LOG && bpf) bpf[(size_t)(n - 1) * K * C + p] = (uint8_t)arg;
        }
        __syncthreads();
        const int kmax = (K - 1 < n) ? K - 1 : n;
        for (int to = tid; to < C; to += nth) {
            double best = SMM_NEG_INF;
            int arg = 1;
            for (int k = 1; k <= kmax; ++k) {
                const double v = alpha[(size_t)((n - k) % K) * K * C + (size_t)k * C + to];
                if (LOG) best = smm_lse2d(best, v);
                else if (v > best) { best = v; arg = k; }
            }
            beta[(size_t)n * C + to] = best;
            if (!LOG && bpk) bpk[(size_t)n * C + to] = (uint16_t)arg;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double *bl = beta + (size_t)(L - 1) * C;
        double best = SMM_NEG_INF;
        int cur = 0;
        for (int c = 0; c < C; ++c) {
            if (LOG) best = smm_lse2d(best, bl[c]);
            else if (bl[c] > best) { best = bl[c]; cur = c; }
        }
        a.v[i] = best;
        if (!LOG && a.spans) {
            int64_t *sp = a.spans + (size_t)i * (N1 + 1);
            for (int n = 0; n <= N1; ++n) sp[n] = -1;
            int n = L - 1;
            sp[n] = cur;
            while (n > 0) {
                const int k = bpk[(size_t)n * C + cur];
                const int frm = bpf[(size_t)(n - k) * K * C + (size_t)k * C + cur];
                n -= k;
                cur = frm;
                sp[n] = cur;      // start of the span [n, n+k) labelled `frm`; at n == 0 this is seq[0]
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ posterior marginals
// d logZ / d edge[n, k, to, from] = exp( beta[n][from] + edge[n, k, to, from] + R[n + k][to] - logZ ): what autograd
// through torch_struct's LogSemiring DP gives the reference (semimarkov.py:286 through modules:624-657), for code that
// trains through the dense interface.  beta is what the forward kernel left in the workspace;
//   R[m][to] = log-weight of everything behind "a span of `to` starts at position m":   R[L-1][.] = 0,
//   R[n][from] = LSE_{k >= 1, n + k <= L - 1, to} ( edge[n, k, to, from] + R[n + k][to] ).
// One workgroup per instance walks n downwards (thread = (from, slice of the (k, to) pairs), slices merged in LDS);
// a second, fully parallel kernel writes the marginals.
__global__ void __launch_bounds__(1024) smm_dense_backward_kernel(SmmDenseArgs a, double *rmsg)
{
    const int i = blockIdx.x;
    const int N1 = a.n1, K = a.k, C = a.c;
    const int L = (int)a.lengths[i];
    const float *edge = a.edge + (size_t)i * N1 * K * C * C;
    double *R = rmsg + (size_t)i * (N1 + 1) * C;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int parts = nth / C > 0 ? nth / C : 1;               // slices of the (k, to) pairs per `from`
    extern __shared__ double sh[];                              // [parts][C] partial log-sums
    for (int c = tid; c < (N1 + 1) * C; c += nth) R[c] = (c / C == L - 1) ? 0.0 : SMM_NEG_INF;
    __syncthreads();
    const int from = tid % C, part = tid / C;
    for (int n = L - 2; n >= 0; --n) {
        double acc = SMM_NEG_INF;
        if (part < parts) {
            const int kmax = (K - 1 < L - 1 - n) ? K - 1 : L - 1 - n;
            const int terms = kmax * C;                         // (k - 1) * C + to
            for (int q = part; q < terms; q += parts) {
                const int k = 1 + q / C, to = q - (k - 1) * C;
                const double v = (double)edge[(((size_t)n * K + k) * C + to) * C + from] + R[(size_t)(n + k) * C + to];
                acc = smm_lse2d(acc, v);
            }
            sh[part * C + from] = acc;
        }
        __syncthreads();
        if (tid < C) {
            double r = SMM_NEG_INF;
            for (int p2 = 0; p2 < parts; ++p2) r = smm_lse2d(r, sh[p2 * C + tid]);
            R[(size_t)n * C + tid] = r;
        }
        __syncthreads();
    }
}

// grid (n, instance); threads stride over (k, to, from)
__global__ void __launch_bounds__(256) smm_dense_marginals_kernel(SmmDenseArgs a, const double *rmsg, const double *grad_v, float *out)
{
    const int n = blockIdx.x, i = blockIdx.y;
    const int N1 = a.n1, K = a.k, C = a.c;
    const int L = (int)a.lengths[i];
    const size_t base = ((size_t)i * N1 + n) * K * C * C;
    const double *beta = a.beta + (size_t)i * (N1 + 1) * C + (size_t)n * C;
    const double *R = rmsg + (size_t)i * (N1 + 1) * C;
    const double lz = a.v[i];
    const double up = grad_v ? grad_v[i] : 1.0;
    for (int q = threadIdx.x; q < K * C * C; q += blockDim.x) {
        const int k = q / (C * C), to = (q / C) % C, from = q % C;
        double m = 0.0;
        if (k >= 1 && n + k <= L - 1)
            m = up * exp(beta[from] + (double)a.edge[base + q] + R[(size_t)(n + k) * C + to] - lz);
        out[base + q] = (float)m;
    }
}

void smm_launch_dense_marginals(const SmmDenseArgs &a, double *rmsg, const double *grad_v, float *out, hipStream_t stream)
{
    int threads = a.k * a.c;
    threads = threads < 64 ? 64 : (threads > 1024 ? 1024 : ((threads + 63) / 64) * 64);
    const int parts = threads / a.c > 0 ? threads / a.c : 1;
    hipLaunchKernelGGL(smm_dense_backward_kernel, dim3(a.b), dim3(threads), sizeof(double) * parts * a.c, stream, a, rmsg);
    hipLaunchKernelGGL(smm_dense_marginals_kernel, dim3(a.n1, a.b), dim3(256), 0, stream, a, rmsg, grad_v, out);
}

void smm_launch_dense(const SmmDenseArgs &a, bool log_semiring, hipStream_t stream)
{
    int threads = a.k * a.c;
    threads = threads < 64 ? 64 : (threads > 1024 ? 1024 : ((threads + 63) / 64) * 64);
    if (log_semiring) hipLaunchKernelGGL(smm_dense_kernel<true>, dim3(a.b), dim3(threads), 0, stream, a);
    else hipLaunchKernelGGL(smm_dense_kernel<false>, dim3(a.b), dim3(threads), 0, stream, a);
}
