// smm_fit.hip -- sufficient statistics of the closed-form supervised fit (HBM-bound single pass over the features).
//
// Replaces semimarkov_utils.semimarkov_sufficient_stats (reference src/models/semimarkov/semimarkov_utils.py:74-126:
// sklearn GaussianMixture._initialize on one-hot responsibilities + the span counting loop) as consumed by
// SemiMarkovModule.fit_supervised (semimarkov_modules.py:195-256).  CPU statement: oracle/dense_ref.py: sufficient_stats.
//
//   smm_class_sums_kernel   sum_x[c][d] = sum of x[t][d] over frames labelled c;  sum_x2[d] = sum of x[t][d]^2
//                           (class means = sum_x / count; tied diagonal variance = sum_x2/n - (sum_c sum_x / n)^2)
//   smm_span_stats_kernel   frames per class, spans per class, first-span class, span transitions [to][from], where a
//                           span ends at a label change or after max_k - 1 frames (labels_to_spans, utils.py:6-23)
//
// Algorithmic bytes per frame: 4 D (features, read once) + 8 (label, read by each kernel) -> 4 D + 16.
// Sums are fp64 and leave the workgroup through atomics: the ORDER of the additions (and with it the last bits of the
// result) is not fixed from run to run; the reference's BLAS-threaded sklearn sums are not either.
#include <vector>

#include "../../include/smmdp.h"
#include "smm_device.h"
#include "smm_launch.h"

#define SMM_FIT_ROWS 1024        // frames per workgroup of the class-sum kernel (256 per wave)
#define SMM_FIT_UNROLL 8

struct SmmFitVideo {
    int64_t frame_off;
    int32_t T;
    int32_t pad;
};

struct SmmFitArgs {
    const SmmFitVideo *videos;
    const float *x;              // [total_frames][d]
    const int64_t *labels;       // [total_frames]
    double *sum_x;               // [n_classes][d]
    double *sum_x2;              // [d]
    unsigned long long *frames;  // [n_classes]
    unsigned long long *spans;   // [n_classes]
    unsigned long long *starts;  // [n_classes]
    unsigned long long *trans;   // [n_classes][n_classes]  [to][from]
    int32_t *err;                // label outside [0, n_classes)
    int32_t d, n_classes, max_k, b;
};

__device__ __forceinline__ void smm_atomic_add(double *p, double v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// grid (b, ceil(t_max / ROWS)); each wave streams its 256 rows; lane = 4 consecutive feature columns
__global__ void __launch_bounds__(256) smm_class_sums_kernel(SmmFitArgs a)
{
    const SmmFitVideo mv = a.videos[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r0 = blockIdx.y * SMM_FIT_ROWS + wv * (SMM_FIT_ROWS / 4);
    if (r0 >= mv.T) return;
    const int r1 = min(mv.T, r0 + SMM_FIT_ROWS / 4);
    const int d = a.d;
    const float *x = a.x + (size_t)mv.frame_off * d;
    const int64_t *y = a.labels + mv.frame_off;
    for (int c0 = 0; c0 < d; c0 += 256) {                       // column pass (one pass for D <= 256)
        const int col = c0 + lane * 4;
        const int ncol = min(4, d - col);                       // <= 0: this lane has no columns in this pass
        const bool vec = ncol == 4 && (d & 3) == 0;
        double acc[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
        int64_t cur = y[r0];
        for (int r = r0; r < r1; r += SMM_FIT_UNROLL) {
            float v[SMM_FIT_UNROLL][4];
            int64_t lab[SMM_FIT_UNROLL];
#pragma unroll
            for (int u = 0; u < SMM_FIT_UNROLL; ++u) {
                const int rr = min(r + u, r1 - 1);              // clamped: the tail re-reads the last row, then skips it
                lab[u] = y[rr];
                if (vec) {
                    const float4 q = *reinterpret_cast<const float4 *>(x + (size_t)rr * d + col);
                    v[u][0] = q.x; v[u][1] = q.y; v[u][2] = q.z; v[u][3] = q.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[u][j] = j < ncol ? x[(size_t)rr * d + col + j] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < SMM_FIT_UNROLL; ++u) {
                if (r + u >= r1) break;
                if (lab[u] != cur) {                             // wave-uniform
                    if (cur >= 0 && cur < a.n_classes)
                        for (int j = 0; j < 4; ++j)
                            if (j < ncol) smm_atomic_add(a.sum_x + (size_t)cur * d + col + j, acc[j]);
                    acc[0] = acc[1] = acc[2] = acc[3] = 0;
                    cur = lab[u];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double xv = (double)v[u][j];
                    acc[j] += xv;
                    sq[j] += xv * xv;
                }
            }
        }
        if (cur >= 0 && cur < a.n_classes)
            for (int j = 0; j < 4; ++j)
                if (j < ncol) smm_atomic_add(a.sum_x + (size_t)cur * d + col + j, acc[j]);
        for (int j = 0; j < 4; ++j)
            if (j < ncol) smm_atomic_add(a.sum_x2 + col + j, sq[j]);
    }
}

// one workgroup per video: ordered tiles of 256 frames; run starts by a max-scan of the change positions
__global__ void __launch_bounds__(256) smm_span_stats_kernel(SmmFitArgs a)
{
    const SmmFitVideo mv = a.videos[blockIdx.x];
    const int T = mv.T, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t *y = a.labels + mv.frame_off;
    const int cut = a.max_k > 1 ? a.max_k - 1 : 1;              // a run is cut every `cut` frames
    const bool do_cut = a.max_k > 0;
    extern __shared__ unsigned int s_hist[];                    // frames per class of this video
    __shared__ int s_wmax[4];
    for (int i = tid; i < a.n_classes; i += blockDim.x) s_hist[i] = 0;
    __syncthreads();
    int carry = 0;                                              // start of the run that is open at the tile boundary
    for (int t0 = 0; t0 < T; t0 += 256) {
        const int t = t0 + tid;
        const bool live = t < T;
        int64_t lab = -1, before = -1;
        if (live) {
            lab = y[t];
            before = t > 0 ? y[t - 1] : -1;
        }
        const bool bad = live && (lab < 0 || lab >= a.n_classes);
        if (bad) atomicOr(a.err, 1);
        const bool change = live && (t == 0 || lab != before);
        int rs = change ? t : -1;                               // inclusive max-scan -> start of the run containing t
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(rs, off);
            if (lane >= off) rs = max(rs, o);
        }
        if (lane == 63) s_wmax[wv] = rs;
        __syncthreads();
        int pre = carry;
        for (int q = 0; q < wv; ++q) pre = max(pre, s_wmax[q]);
        rs = max(rs, pre);
        if (live && !bad) {
            atomicAdd(&s_hist[lab], 1u);
            const bool start = change || (do_cut && (t - rs) % cut == 0);
            if (start) {
                atomicAdd(a.spans + lab, 1ull);
                if (t == 0) atomicAdd(a.starts + lab, 1ull);
                else if (before >= 0 && before < a.n_classes)
                    atomicAdd(a.trans + (size_t)lab * a.n_classes + before, 1ull);
            }
        }
        int nxt = carry;
        for (int q = 0; q < 4; ++q) nxt = max(nxt, s_wmax[q]);
        __syncthreads();
        carry = nxt;
    }
    for (int i = tid; i < a.n_classes; i += blockDim.x)
        if (s_hist[i]) atomicAdd(a.frames + i, (unsigned long long)s_hist[i]);
}

extern "C" size_t smm_fit_workspace_bytes(int32_t b) { return b > 0 ? sizeof(SmmFitVideo) * (size_t)b + 256 : 0; }

extern "C" int smm_fit_stats_f64(int32_t b, const int64_t *lengths, const int64_t *frame_off, int64_t total_frames,
                                 int32_t d, int32_t n_classes, int32_t max_k, const float *x, const int64_t *labels,
                                 double *sum_x, double *sum_x2, int64_t *frame_counts, int64_t *span_counts,
                                 int64_t *span_start_counts, int64_t *span_transition_counts,
                                 void *ws, size_t ws_bytes, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (b <= 0 || !lengths || !frame_off || d <= 0 || n_classes <= 0 || n_classes > 16000 || !x || !labels || !sum_x ||
        !sum_x2 || !frame_counts || !span_counts || !span_start_counts || !span_transition_counts || !ws)
        return SMM_ERR_ARG;
    if (ws_bytes < smm_fit_workspace_bytes(b)) return SMM_ERR_WORKSPACE;
    std::vector<SmmFitVideo> hv(b);
    int64_t t_max = 0;
    for (int i = 0; i < b; ++i) {
        if (lengths[i] < 1 || frame_off[i] < 0 || frame_off[i] + lengths[i] > total_frames || lengths[i] > 0x7FFFFFFF)
            return SMM_ERR_ARG;
        hv[i].frame_off = frame_off[i];
        hv[i].T = (int32_t)lengths[i];
        hv[i].pad = 0;
        t_max = lengths[i] > t_max ? lengths[i] : t_max;
    }
    char *base = static_cast<char *>(ws);
    const size_t o_err = (sizeof(SmmFitVideo) * (size_t)b + 63) / 64 * 64;
#define SMM_FIT_HIP(call) do { if ((call) != hipSuccess) return SMM_ERR_HIP; } while (0)
    SMM_FIT_HIP((hipError_t)smm_upload_meta(base, hv.data(), sizeof(SmmFitVideo) * b, stream));
    SMM_FIT_HIP(hipMemsetAsync(base + o_err, 0, 64, stream));
    const size_t n = (size_t)n_classes;
    SMM_FIT_HIP(hipMemsetAsync(sum_x, 0, sizeof(double) * n * d, stream));
    SMM_FIT_HIP(hipMemsetAsync(sum_x2, 0, sizeof(double) * d, stream));
    SMM_FIT_HIP(hipMemsetAsync(frame_counts, 0, sizeof(int64_t) * n, stream));
    SMM_FIT_HIP(hipMemsetAsync(span_counts, 0, sizeof(int64_t) * n, stream));
    SMM_FIT_HIP(hipMemsetAsync(span_start_counts, 0, sizeof(int64_t) * n, stream));
    SMM_FIT_HIP(hipMemsetAsync(span_transition_counts, 0, sizeof(int64_t) * n * n, stream));
    SmmFitArgs a{};
    a.videos = reinterpret_cast<const SmmFitVideo *>(base);
    a.x = x;
    a.labels = labels;
    a.sum_x = sum_x;
    a.sum_x2 = sum_x2;
    a.frames = reinterpret_cast<unsigned long long *>(frame_counts);
    a.spans = reinterpret_cast<unsigned long long *>(span_counts);
    a.starts = reinterpret_cast<unsigned long long *>(span_start_counts);
    a.trans = reinterpret_cast<unsigned long long *>(span_transition_counts);
    a.err = reinterpret_cast<int32_t *>(base + o_err);
    a.d = d;
    a.n_classes = n_classes;
    a.max_k = max_k;
    a.b = b;
    dim3 grid(b, (unsigned)((t_max + SMM_FIT_ROWS - 1) / SMM_FIT_ROWS));
    hipLaunchKernelGGL(smm_class_sums_kernel, grid, dim3(256), 0, stream, a);
    hipLaunchKernelGGL(smm_span_stats_kernel, dim3(b), dim3(256), sizeof(unsigned int) * n, stream, a);
    return hipGetLastError() == hipSuccess ? SMM_OK : SMM_ERR_HIP;
}

extern "C" size_t smm_fit_error_word_offset(int32_t b)
{
    return b > 0 ? (sizeof(SmmFitVideo) * (size_t)b + 63) / 64 * 64 : 0;
}
