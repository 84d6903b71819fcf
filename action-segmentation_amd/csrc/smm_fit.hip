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
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "../../include/smmdp.h"
#include "smm_device.h"
#include "smm_launch.h"

#define SMM_FIT_CHUNK 2048       // frames per work item of both kernels (one table of chunks per video)
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
    const int32_t *cum;          // [b + 1] chunks of SMM_FIT_CHUNK frames before each video
    int32_t d, n_classes, max_k, b, n_chunks;
};

// largest i in [0, n) with cum[i] <= c
__device__ __forceinline__ int smm_fit_find_video(const int32_t *__restrict__ cum, int n, int c)
{
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cum[mid] <= c) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ void smm_atomic_add(double *p, double v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Persistent grid (a few workgroups per CU) over the flat list of 2048-frame chunks; a wave streams 512 rows of its
// chunk, lane = 4 consecutive feature columns (one 800-byte row per load instruction at D = 200, 8 rows in flight).
// Class sums leave the wave at every label change (one fp64 atomic per column into the class's row: ~300 different rows
// on CrossTask, so the memory-side atomic units do not queue); the sum of squares -- ONE row for everybody, where
// per-chunk atomics serialise at ~0.09 TB/s (MI355X_MICROARCH.md, Global float atomics: contention) -- stays in
// registers across all chunks of the wave, is reduced over the workgroup's waves in LDS and leaves once per workgroup.
// VEC (D % 4 == 0): every lane loads its 4 columns as ONE 16-byte piece, unconditionally (lanes past the last column
// re-read column 0 and are ignored) -- with the element-wise fallback in the same kernel the compiler if-converts both
// paths into per-element conditional loads.
template <bool VEC>
__global__ void __launch_bounds__(256) smm_class_sums_kernel(SmmFitArgs a)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int d = a.d;
    __shared__ double s_sq[4][256];
    // A lane owns 4 consecutive columns, so its 4 sums are 32 bytes apart from its neighbour's: an atomic instruction
    // straight from those registers would touch 64 separate 8-byte words.  The sums go through the wave's row of s_sq
    // first, so that every atomic instruction adds 64 CONSECUTIVE doubles (512 contiguous bytes, the shape the
    // memory-side atomic units run at full rate for).  Wave-private LDS, wave-uniform control flow: no barrier needed.
    auto flush = [&](double (&acc)[4], double *dst_row, int c0, int ncol_total) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s_sq[wv][lane * 4 + j] = acc[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cc = 64 * j + lane;
            if (cc < ncol_total) smm_atomic_add(dst_row + c0 + cc, s_sq[wv][cc]);
        }
    };
    for (int c0 = 0; c0 < d; c0 += 256) {                       // column pass (one pass for D <= 256)
        const int ncol_total = min(256, d - c0);
        const int col = c0 + lane * 4;
        const int ncol = min(4, d - col);                       // <= 0: this lane has no columns in this pass
        const int colc = ncol > 0 ? col : 0;                    // (VEC: where an idle lane reads)
        double sq[4] = {0, 0, 0, 0};
        // work items of 512 rows (a quarter of a chunk; 128 rows per wave), dealt round-robin: ~5 per workgroup, so the
        // last round costs a tenth of the launch instead of a third
        for (int item = blockIdx.x; item < 4 * a.n_chunks; item += gridDim.x) {
            const int ch = item >> 2;
            const int vi = smm_fit_find_video(a.cum, a.b, ch);
            const SmmFitVideo mv = a.videos[vi];
            const int r0 = (ch - a.cum[vi]) * SMM_FIT_CHUNK + (item & 3) * (SMM_FIT_CHUNK / 4) + wv * (SMM_FIT_CHUNK / 16);
            if (r0 >= mv.T) continue;
            const int r1 = min(mv.T, r0 + SMM_FIT_CHUNK / 16);
            const float *x = a.x + (size_t)mv.frame_off * d;
            const int64_t *y = a.labels + mv.frame_off;
            double acc[4] = {0, 0, 0, 0};
            int64_t cur = y[r0];
            // two row buffers: the next 8 rows are in flight while these 8 are added up
            float v[2][SMM_FIT_UNROLL][4];
            int64_t lab[2][SMM_FIT_UNROLL];
            auto fetch = [&](int r, float (&vv)[SMM_FIT_UNROLL][4], int64_t (&ll)[SMM_FIT_UNROLL]) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < SMM_FIT_UNROLL; ++u) {
                    const int rr = min(r + u, r1 - 1);          // clamped: the tail re-reads the last row, then skips it
                    ll[u] = y[rr];
                    if constexpr (VEC) {
                        const float4 q = *reinterpret_cast<const float4 *>(x + (size_t)rr * d + colc);
                        vv[u][0] = q.x; vv[u][1] = q.y; vv[u][2] = q.z; vv[u][3] = q.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) vv[u][j] = j < ncol ? x[(size_t)rr * d + col + j] : 0.f;
                    }
                }
            };
            auto consume = [&](int r, const float (&vv)[SMM_FIT_UNROLL][4], const int64_t (&ll)[SMM_FIT_UNROLL]) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < SMM_FIT_UNROLL; ++u) {
                    if (r + u >= r1) break;
                    if (ll[u] != cur) {                          // wave-uniform
                        if (cur >= 0 && cur < a.n_classes) flush(acc, a.sum_x + (size_t)cur * d, c0, ncol_total);
                        acc[0] = acc[1] = acc[2] = acc[3] = 0;
                        cur = ll[u];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const double xv = (VEC && ncol <= 0) ? 0.0 : (double)vv[u][j];
                        acc[j] += xv;
                        sq[j] += xv * xv;
                    }
                }
            };
            fetch(r0, v[0], lab[0]);
            for (int r = r0; r < r1; r += 2 * SMM_FIT_UNROLL) {
                fetch(r + SMM_FIT_UNROLL, v[1], lab[1]);
                consume(r, v[0], lab[0]);
                if (r + SMM_FIT_UNROLL >= r1) break;
                fetch(r + 2 * SMM_FIT_UNROLL, v[0], lab[0]);
                consume(r + SMM_FIT_UNROLL, v[1], lab[1]);
            }
            if (cur >= 0 && cur < a.n_classes) flush(acc, a.sum_x + (size_t)cur * d, c0, ncol_total);
        }
        // sum of squares: waves -> LDS -> one atomic per column and workgroup
        __syncthreads();                                        // (every wave is done with its row of s_sq)
#pragma unroll
        for (int j = 0; j < 4; ++j) s_sq[wv][lane * 4 + j] = sq[j];
        __syncthreads();
        {
            const int cc = c0 + (int)threadIdx.x;
            if (cc < d) smm_atomic_add(a.sum_x2 + cc, s_sq[0][threadIdx.x] + s_sq[1][threadIdx.x] + s_sq[2][threadIdx.x] + s_sq[3][threadIdx.x]);
        }
        __syncthreads();
    }
}

// One workgroup per 2048-frame chunk (tiles of 256 frames in order); run starts by a max-scan of the change positions.
// The run that is open where the chunk begins may have started long before it: its start is found by scanning the labels
// backwards from the chunk's first frame, 256 at a time (one iteration for runs shorter than 256 frames).
__global__ void __launch_bounds__(256) smm_span_stats_kernel(SmmFitArgs a)
{
    const int vi = smm_fit_find_video(a.cum, a.b, blockIdx.x);
    const SmmFitVideo mv = a.videos[vi];
    const int T = mv.T, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t *y = a.labels + mv.frame_off;
    const int c_begin = (blockIdx.x - a.cum[vi]) * SMM_FIT_CHUNK;
    const int c_end = min(T, c_begin + SMM_FIT_CHUNK);
    const int cut = a.max_k > 1 ? a.max_k - 1 : 1;              // a run is cut every `cut` frames
    const bool do_cut = a.max_k > 0;
    extern __shared__ unsigned int s_hist[];                    // frames per class of this chunk
    __shared__ int s_wmax[4];
    __shared__ int s_carry;
    for (int i = tid; i < a.n_classes; i += blockDim.x) s_hist[i] = 0;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    // start of the run containing frame c_begin - 1 (0 when the chunk opens the video): the largest p < c_begin with
    // p == 0 or y[p] != y[p-1]
    if (c_begin > 0) {
        for (int hi = c_begin - 1; hi >= 0; hi -= 256) {
            const int p = hi - tid;
            const bool chg = p >= 0 && (p == 0 || y[p] != y[p - 1]);
            if (chg) atomicMax(&s_carry, p);
            __syncthreads();
            if (s_carry > 0 || hi - 255 <= 0) break;            // (uniform: read after the barrier; p == 0 always counts)
            __syncthreads();
        }
        __syncthreads();
    }
    int carry = s_carry;
    for (int t0 = c_begin; t0 < c_end; t0 += 256) {
        const int t = t0 + tid;
        const bool live = t < c_end;
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

static size_t fit_off_err(int32_t b) { return (sizeof(SmmFitVideo) * (size_t)b + 63) / 64 * 64; }
static size_t fit_off_cum(int32_t b) { return fit_off_err(b) + 64; }

extern "C" size_t smm_fit_workspace_bytes(int32_t b)
{
    return b > 0 ? fit_off_cum(b) + sizeof(int32_t) * ((size_t)b + 1) + 256 : 0;
}

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
    std::vector<int32_t> cum((size_t)b + 1);
    int64_t n_chunks = 0;
    for (int i = 0; i < b; ++i) {
        if (lengths[i] < 1 || frame_off[i] < 0 || frame_off[i] + lengths[i] > total_frames || lengths[i] > 0x7FFFFFFF)
            return SMM_ERR_ARG;
        hv[i].frame_off = frame_off[i];
        hv[i].T = (int32_t)lengths[i];
        hv[i].pad = 0;
        cum[i] = (int32_t)n_chunks;
        n_chunks += (lengths[i] + SMM_FIT_CHUNK - 1) / SMM_FIT_CHUNK;
    }
    if (n_chunks > 0x7fffffff) return SMM_ERR_UNSUPPORTED;
    cum[b] = (int32_t)n_chunks;
    char *base = static_cast<char *>(ws);
    const size_t o_err = fit_off_err(b);
#define SMM_FIT_HIP(call) do { if ((call) != hipSuccess) return SMM_ERR_HIP; } while (0)
    SMM_FIT_HIP((hipError_t)smm_upload_meta(base, hv.data(), sizeof(SmmFitVideo) * b, stream));
    SMM_FIT_HIP((hipError_t)smm_upload_meta(base + fit_off_cum(b), cum.data(), sizeof(int32_t) * ((size_t)b + 1), stream));
    const size_t n = (size_t)n_classes;
    {
        // (one launch for the seven of them: smm_zero_multi_async)
        void *const zp[7] = {base + o_err, sum_x, sum_x2, frame_counts, span_counts, span_start_counts, span_transition_counts};
        const size_t zb[7] = {64, sizeof(double) * n * d, sizeof(double) * d, sizeof(int64_t) * n, sizeof(int64_t) * n, sizeof(int64_t) * n,
                              sizeof(int64_t) * n * n};
        SMM_FIT_HIP((hipError_t)smm_zero_multi_async(zp, zb, 7, stream));
    }
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
    a.cum = reinterpret_cast<const int32_t *>(base + fit_off_cum(b));
    a.d = d;
    a.n_classes = n_classes;
    a.max_k = max_k;
    a.b = b;
    a.n_chunks = (int32_t)n_chunks;
    // class sums: a persistent grid of 3 workgroups per CU (what the kernel's 167 VGPRs admit: one resident round, and
    // few flushes of the one sum-of-squares row)
    int g_sum = 768;
    if (smm_env_fit_grid() > 0) g_sum = smm_env_fit_grid();                    // (SMM_FIT_GRID: tuning aid)
    g_sum = (int)std::max<int64_t>(1, std::min<int64_t>(4 * n_chunks, g_sum));
    if ((d & 3) == 0) hipLaunchKernelGGL(smm_class_sums_kernel<true>, dim3(g_sum), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(smm_class_sums_kernel<false>, dim3(g_sum), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(smm_span_stats_kernel, dim3((unsigned)n_chunks), dim3(256), sizeof(unsigned int) * n, stream, a);
    return hipGetLastError() == hipSuccess ? SMM_OK : SMM_ERR_HIP;
}

extern "C" size_t smm_fit_error_word_offset(int32_t b) { return b > 0 ? fit_off_err(b) : 0; }
