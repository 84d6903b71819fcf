// smm_eval.hip -- evaluation counters of decoded frame labels against ground truth (integer work, HBM-bound).
//
// Replaces the per-frame Python loops of the reference's src/evaluation/accuracy.py as driven by
// Datasplit.accuracy_corpus (src/data/corpus.py:486-565):
//   smm_eval_confusion_i64  <- Accuracy._create_voting_table (:232-283) and the per-class masks of mof (:500-521)
//   smm_eval_videos_i64     <- the frame loop of mof (:538-576), run_length_encode (:21-37) + levenshtein (:364-408,
//                              editdistance.eval), single_step_recall (:410-472)
// The label assignment itself (identity or Hungarian on a <= 64 x 64 table) and the final ratios stay on the host
// (action_segmentation_amd/evaluation.py).  CPU statement: oracle/eval_ref.py.
//
// Labels are int64 global class ids on the packed frame axis (what smm_decode_f32 writes).  Each task ("group") has a
// table local_of[g][n_labels] -> local id in [0, c_max) or -1; ids without a local id are counted in the "other"
// bucket (c_max) and reported, never dropped silently.
//
// Algorithmic bytes per frame: kernel 1 reads pred 8 + gt 8*gt_width; kernel 2 reads them twice (counters + centre
// pass) and writes at most 8 B of run-length sequence: 16 + 8*gt_width ... 40 + 16*gt_width B/frame in all.
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/smmdp.h"
#include "smm_device.h"
#include "smm_launch.h"

#define SMM_EVAL_THREADS 256
#define SMM_EVAL_CHUNK 4096          // frames per workgroup of the confusion kernel

struct SmmEvalVideo {
    int64_t frame_off;
    int64_t scratch_off;             // offset (int32 units) of this video's 4*(T+1) scratch words
    int32_t T;
    int32_t group;
    int32_t key;                     // index of the video inside its task (seeds the frame hash)
    int32_t pad;
};

struct SmmEvalArgs {
    const SmmEvalVideo *videos;
    const int64_t *pred;             // [total_frames]
    const int64_t *gt;               // [total_frames][gt_width], -1 padded
    const int32_t *local_of;         // [g][n_labels]
    const int32_t *cluster_of;       // [g][c_max+1]  local gt id -> extended local pred id (< 2*(c_max+1)) or -1
    const uint8_t *gt_is_bg;         // [g][c_max+1]
    const uint8_t *pred_is_bg;       // [g][2*(c_max+1)]
    int64_t *confusion;              // [g][c_max+1][c_max+1]
    int64_t *counters;               // [b][SMM_EVAL_COUNTERS]
    int32_t *scratch;
    int32_t c_max, n_labels, gt_width, b;
    uint32_t seed;
};

__device__ __forceinline__ int smm_local(const int32_t *tab, int n_labels, int cm, int64_t v)
{
    if (v < 0 || v >= n_labels) return cm;
    const int l = tab[v];
    return l < 0 ? cm : l;
}

// identical arithmetic in oracle/eval_ref.py: frame_hash
__device__ __forceinline__ uint32_t smm_frame_hash(uint32_t seed, uint32_t video, uint32_t t)
{
    uint32_t x = seed ^ (video * 0x9E3779B1u) ^ (t * 0x85EBCA77u);
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

// grid (b, ceil(t_max / CHUNK)): LDS histogram over (first gt label, predicted label), flushed with global atomics
__global__ void __launch_bounds__(SMM_EVAL_THREADS) smm_eval_confusion_kernel(SmmEvalArgs a)
{
    const SmmEvalVideo mv = a.videos[blockIdx.x];
    const int t0 = blockIdx.y * SMM_EVAL_CHUNK;
    if (t0 >= mv.T) return;
    const int t1 = min(mv.T, t0 + SMM_EVAL_CHUNK);
    const int cm = a.c_max, w = cm + 1;
    extern __shared__ unsigned int hist[];
    for (int i = threadIdx.x; i < w * w; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const int32_t *tab = a.local_of + (size_t)mv.group * a.n_labels;
    for (int t = t0 + threadIdx.x; t < t1; t += blockDim.x) {
        const int64_t f = mv.frame_off + t;
        const int p = smm_local(tab, a.n_labels, cm, a.pred[f]);
        const int g = smm_local(tab, a.n_labels, cm, a.gt[f * a.gt_width]);
        atomicAdd(&hist[g * w + p], 1u);
    }
    __syncthreads();
    int64_t *out = a.confusion + (size_t)mv.group * w * w;
    for (int i = threadIdx.x; i < w * w; i += blockDim.x)
        if (hist[i]) atomicAdd(reinterpret_cast<unsigned long long *>(out + i), (unsigned long long)hist[i]);
}

__device__ __forceinline__ int64_t smm_block_sum(int64_t v, int64_t *red)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    int64_t s = 0;
    for (int i = 0; i < SMM_EVAL_THREADS / 64; ++i) s += red[i];
    return s;
}

// one workgroup per video
__global__ void __launch_bounds__(SMM_EVAL_THREADS) smm_eval_video_kernel(SmmEvalArgs a)
{
    const int vid = blockIdx.x;
    const SmmEvalVideo mv = a.videos[vid];
    const int T = mv.T, cm = a.c_max, w = cm + 1, w2 = 2 * w, gw = a.gt_width;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int32_t *tab = a.local_of + (size_t)mv.group * a.n_labels;
    const int64_t *pred = a.pred + mv.frame_off;
    const int64_t *gt = a.gt + mv.frame_off * gw;
    int32_t *seq_a = a.scratch + mv.scratch_off, *seq_b = seq_a + (T + 1);
    int32_t *row0 = seq_b + (T + 1), *row1 = row0 + (T + 1);

    constexpr int W2 = 2 * (SMM_EVAL_MAX_LABELS + 1);
    __shared__ int32_t s_cluster[SMM_EVAL_MAX_LABELS + 1];
    __shared__ uint8_t s_gbg[SMM_EVAL_MAX_LABELS + 1], s_pbg[W2];
    __shared__ unsigned int s_first[W2], s_last[W2], s_pred_seen[W2], s_gt_seen[W2];
    __shared__ unsigned long long s_draw[W2], s_mid[W2];
    __shared__ int s_wtot[SMM_EVAL_THREADS / 64][3];
    __shared__ int64_t s_red[SMM_EVAL_THREADS / 64];

    for (int i = tid; i < w; i += blockDim.x) {
        s_cluster[i] = a.cluster_of[(size_t)mv.group * w + i];
        s_gbg[i] = a.gt_is_bg[(size_t)mv.group * w + i];
    }
    for (int i = tid; i < w2; i += blockDim.x) {
        s_pbg[i] = a.pred_is_bg[(size_t)mv.group * w2 + i];
        s_first[i] = 0xFFFFFFFFu;
        s_last[i] = 0;
        s_pred_seen[i] = 0;
        s_gt_seen[i] = 0;
        s_draw[i] = ~0ull;
        s_mid[i] = ~0ull;
    }
    __syncthreads();

    // ---- pass 1: frame counters, per-label first/last/draw, ordered run-length compaction --------------------
    int64_t c_multi = 0, c_reclen = 0, c_tp = 0, c_pbg = 0, c_tbg = 0, c_ioud = 0, c_ioun = 0, c_recnb = 0,
            c_precnb = 0, c_tpnb = 0, c_other = 0;
    int base_a = 0, base_b = 0, base_bnb = 0;                         // running segment counts (uniform)
    for (int t0 = 0; t0 < T; t0 += SMM_EVAL_THREADS) {
        const int t = t0 + tid;
        const bool live = t < T;
        bool start_a = false, start_b = false, start_bnb = false;
        int p = cm, ra = -1;
        if (live) {
            const int64_t pv = pred[t];
            p = smm_local(tab, a.n_labels, cm, pv);
            c_other += p == cm;
            int n_lab = 0;
            bool tp = false, is_bg = false;
            for (int j = 0; j < gw; ++j) {
                const int64_t gv = gt[(size_t)t * gw + j];
                if (gv < 0) continue;
                const int g = smm_local(tab, a.n_labels, cm, gv);
                if (j == 0) {
                    c_other += g == cm;
                    ra = s_cluster[g];
                    if (ra >= 0) s_gt_seen[ra] = 1;                       // benign race: everyone writes 1
                }
                ++n_lab;
                tp |= s_cluster[g] == p && p != cm;
                is_bg |= s_gbg[g] != 0;
            }
            const bool p_bg = s_pbg[p] != 0;
            c_multi += n_lab > 1;
            c_reclen += n_lab;
            c_tp += tp;
            c_pbg += p_bg;
            c_tbg += is_bg;
            if (!(is_bg && p_bg)) { ++c_ioud; c_ioun += tp; }
            if (!is_bg) { c_recnb += n_lab; ++c_precnb; c_tpnb += tp; }
            s_pred_seen[p] = 1;
            atomicMin(&s_first[p], (unsigned)t);
            atomicMax(&s_last[p], (unsigned)t);
            atomicMin(&s_draw[p], ((unsigned long long)smm_frame_hash(a.seed, (uint32_t)mv.key, (uint32_t)t) << 32) | (unsigned)t);
            start_b = t == 0 || pred[t - 1] != pv;
            start_bnb = start_b && !p_bg;
            start_a = t == 0 || gt[(size_t)(t - 1) * gw] != gt[(size_t)t * gw];
        }
        const unsigned long long ba = __ballot(start_a), bb = __ballot(start_b), bn = __ballot(start_bnb);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (lane == 0) { s_wtot[wv][0] = __popcll(ba); s_wtot[wv][1] = __popcll(bb); s_wtot[wv][2] = __popcll(bn); }
        __syncthreads();
        int pre_a = base_a, pre_b = base_b;
        for (int q = 0; q < wv; ++q) { pre_a += s_wtot[q][0]; pre_b += s_wtot[q][1]; }
        if (start_a) seq_a[pre_a + __popcll(ba & below)] = ra;
        if (start_b) seq_b[pre_b + __popcll(bb & below)] = p;
        for (int q = 0; q < SMM_EVAL_THREADS / 64; ++q) { base_a += s_wtot[q][0]; base_b += s_wtot[q][1]; base_bnb += s_wtot[q][2]; }
        __syncthreads();
    }

    // ---- pass 2: the frame of each predicted label closest to the middle of its extent (ties: the earlier one) ----
    for (int t = tid; t < T; t += blockDim.x) {
        const int p = smm_local(tab, a.n_labels, cm, pred[t]);
        const long long d = 2ll * t - ((long long)s_first[p] + (long long)s_last[p]);
        atomicMin(&s_mid[p], ((unsigned long long)(d < 0 ? -d : d) << 32) | (unsigned)t);
    }
    __syncthreads();

    // ---- per-label step statistics (thread = extended local label) ----
    int64_t tot = 0, tot_nb = 0, hit = 0, hit_nb = 0, mid = 0, mid_nb = 0, types = 0, types_nb = 0;
    for (int l = tid; l < w2; l += blockDim.x) {
        const bool nb = s_pbg[l] == 0;
        if (l < w && l != cm && s_pred_seen[l]) { ++types; types_nb += nb; }
        if (!s_gt_seen[l]) continue;
        ++tot;
        tot_nb += nb;
        if (l >= w || !s_pred_seen[l]) continue;
        const int td = (int)(s_draw[l] & 0xFFFFFFFFull), tc = (int)(s_mid[l] & 0xFFFFFFFFull);
        const int gd = smm_local(tab, a.n_labels, cm, gt[(size_t)td * gw]);
        const int gc = smm_local(tab, a.n_labels, cm, gt[(size_t)tc * gw]);
        if (s_cluster[gd] == l) { ++hit; hit_nb += nb; }
        if (s_cluster[gc] == l) { ++mid; mid_nb += nb; }
    }

    // ---- Levenshtein distance of the two run-length label sequences (wave 0; row i from row i-1 by a prefix min) ----
    __threadfence_block();
    __syncthreads();
    int lev = 0;
    if (wv == 0) {
        const int n = base_a, m = base_b;
        int32_t *prev = row0, *cur = row1;
        for (int j = lane; j <= m; j += 64) prev[j] = j;
        __threadfence_block();
        for (int i = 1; i <= n; ++i) {
            const int ai = seq_a[i - 1];
            int carry = 0x3FFFFFFF;
            for (int j0 = 0; j0 <= m; j0 += 64) {
                const int j = j0 + lane;
                int x = 0x3FFFFFFF;
                if (j <= m) {
                    const int tcell = j == 0 ? i : min(prev[j] + 1, prev[j - 1] + (ai != seq_b[j - 1] ? 1 : 0));
                    x = tcell - j;
                }
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int y = __shfl_up(x, off);
                    if (lane >= off) x = min(x, y);
                }
                x = min(x, carry);
                if (j <= m) cur[j] = x + j;
                carry = __shfl(x, 63);
            }
            __threadfence_block();
            int32_t *sw = prev; prev = cur; cur = sw;
        }
        lev = prev[m];            // every lane reads the same word
    }

    const int64_t sums[] = {c_multi, c_reclen, c_tp, c_pbg, c_tbg, c_ioud, c_ioun, c_recnb, c_precnb, c_tpnb,
                            tot, tot_nb, hit, hit_nb, mid, mid_nb, types, types_nb, c_other};
    int64_t *out = a.counters + (size_t)vid * SMM_EVAL_COUNTERS;
    constexpr int NS = sizeof(sums) / sizeof(sums[0]);
    for (int q = 0; q < NS; ++q) {
        const int64_t s = smm_block_sum(sums[q], s_red);
        if (tid == 0) out[4 + q] = s;
    }
    if (tid == 0) {
        out[0] = T;
        out[1] = base_a;
        out[2] = base_b;
        out[3] = base_bnb;
        out[4 + NS] = lev;
        for (int q = 5 + NS; q < SMM_EVAL_COUNTERS; ++q) out[q] = 0;
    }
}

// ------------------------------------------------------------------------------------------------ C ABI
static inline size_t ev_align(size_t x, size_t al) { return (x + al - 1) / al * al; }

static bool ev_shape_ok(const smm_eval_shape *s)
{
    return s && s->b > 0 && s->n_groups > 0 && s->c_max > 0 && s->c_max <= SMM_EVAL_MAX_LABELS && s->n_labels > 0 &&
           s->gt_width > 0 && s->t_max > 0 && s->total_frames > 0;
}

extern "C" size_t smm_eval_workspace_bytes(const smm_eval_shape *s, const int64_t *lengths)
{
    if (!ev_shape_ok(s) || !lengths) return 0;
    size_t words = 0;
    for (int i = 0; i < s->b; ++i) {
        if (lengths[i] < 1 || lengths[i] > s->t_max) return 0;
        words += 4 * (size_t)(lengths[i] + 1);
    }
    return ev_align(sizeof(SmmEvalVideo) * s->b, 256) + 4 * words + 256;
}

static int ev_stage(const smm_eval_shape *s, const int64_t *lengths, const int64_t *frame_off, const int32_t *group,
                    const int32_t *video_key, void *ws, size_t ws_bytes, hipStream_t stream, SmmEvalArgs *a)
{
    if (!ev_shape_ok(s) || !lengths || !frame_off || !ws) return SMM_ERR_ARG;
    const size_t need = smm_eval_workspace_bytes(s, lengths);
    if (need == 0) return SMM_ERR_ARG;
    if (ws_bytes < need) return SMM_ERR_WORKSPACE;
    std::vector<SmmEvalVideo> hv(s->b);
    size_t off = 0;
    for (int i = 0; i < s->b; ++i) {
        if (frame_off[i] < 0 || frame_off[i] + lengths[i] > s->total_frames) return SMM_ERR_ARG;
        const int g = group ? group[i] : 0;
        if (g < 0 || g >= s->n_groups) return SMM_ERR_ARG;
        hv[i].frame_off = frame_off[i];
        hv[i].scratch_off = (int64_t)off;
        hv[i].T = (int32_t)lengths[i];
        hv[i].group = g;
        hv[i].key = video_key ? video_key[i] : i;
        hv[i].pad = 0;
        off += 4 * (size_t)(lengths[i] + 1);
    }
    char *base = static_cast<char *>(ws);
    // through kernel arguments: no pageable host-to-device copy, the host never waits for the stream (smm_launch.h)
    if (smm_upload_meta(base, hv.data(), sizeof(SmmEvalVideo) * s->b, stream) != (int)hipSuccess)
        return SMM_ERR_HIP;
    std::memset(a, 0, sizeof(*a));
    a->videos = reinterpret_cast<const SmmEvalVideo *>(base);
    a->scratch = reinterpret_cast<int32_t *>(base + ev_align(sizeof(SmmEvalVideo) * s->b, 256));
    a->c_max = s->c_max;
    a->n_labels = s->n_labels;
    a->gt_width = s->gt_width;
    a->b = s->b;
    return SMM_OK;
}

extern "C" int smm_eval_confusion_i64(const smm_eval_shape *s, const int64_t *lengths, const int64_t *frame_off,
                                      const int32_t *group, const int64_t *pred, const int64_t *gt,
                                      const int32_t *local_of, int64_t *confusion, void *ws, size_t ws_bytes,
                                      void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!pred || !gt || !local_of || !confusion) return SMM_ERR_ARG;
    SmmEvalArgs a;
    const int rc = ev_stage(s, lengths, frame_off, group, nullptr, ws, ws_bytes, stream, &a);
    if (rc != SMM_OK) return rc;
    a.pred = pred;
    a.gt = gt;
    a.local_of = local_of;
    a.confusion = confusion;
    const size_t w = (size_t)s->c_max + 1;
    if (smm_zero_async(confusion, sizeof(int64_t) * s->n_groups * w * w, stream) != (int)hipSuccess) return SMM_ERR_HIP;
    dim3 grid(s->b, (s->t_max + SMM_EVAL_CHUNK - 1) / SMM_EVAL_CHUNK);
    hipLaunchKernelGGL(smm_eval_confusion_kernel, grid, dim3(SMM_EVAL_THREADS), sizeof(unsigned int) * w * w, stream, a);
    return hipGetLastError() == hipSuccess ? SMM_OK : SMM_ERR_HIP;
}

extern "C" int smm_eval_videos_i64(const smm_eval_shape *s, const int64_t *lengths, const int64_t *frame_off,
                                   const int32_t *group, const int32_t *video_key, const int64_t *pred,
                                   const int64_t *gt, const int32_t *local_of, const int32_t *cluster_of,
                                   const uint8_t *gt_is_bg, const uint8_t *pred_is_bg, uint32_t seed,
                                   int64_t *counters, void *ws, size_t ws_bytes, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!pred || !gt || !local_of || !cluster_of || !gt_is_bg || !pred_is_bg || !counters) return SMM_ERR_ARG;
    SmmEvalArgs a;
    const int rc = ev_stage(s, lengths, frame_off, group, video_key, ws, ws_bytes, stream, &a);
    if (rc != SMM_OK) return rc;
    a.pred = pred;
    a.gt = gt;
    a.local_of = local_of;
    a.cluster_of = cluster_of;
    a.gt_is_bg = gt_is_bg;
    a.pred_is_bg = pred_is_bg;
    a.counters = counters;
    a.seed = seed;
    hipLaunchKernelGGL(smm_eval_video_kernel, dim3(s->b), dim3(SMM_EVAL_THREADS), 0, stream, a);
    return hipGetLastError() == hipSuccess ? SMM_OK : SMM_ERR_HIP;
}
