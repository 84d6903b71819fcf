// smm_api.hip -- host side of libsmmdp.so: argument checks, launch planning, C ABI (include/smmdp.h).
#include <algorithm>
#include <atomic>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>
#include <unordered_map>
#include <mutex>
#include <utility>
#include <string>

#include "../../include/smmdp.h"
#include "smm_device.h"
#include "smm_launch.h"

static thread_local int g_last_hip = 0;

// ------------------------------------------------------------------------------------------------ tuning switches
// Read ONCE, when the library is first used, and again whenever smm_env_reload() is called (tests and A/B scripts that
// flip a switch inside one process call it; nothing on the per-call path touches the environment).  Release builds know
// the switches of the first block only; the ones that change RESULTS (profiling stops, half-run splits) exist in
// -DSMM_DEV builds alone.
namespace {
struct SmmEnv {
    int spec = 1;             // SMM_SPEC=0: Viterbi without the speculative transition (A/B aid; same results)
    int no_split = 0;         // SMM_NO_SPLIT=1: smm_decode_f32 never splits a launch over two streams (same results)
    double split_min_us = 100.0, split_ns = 0.0;     // SMM_SPLIT_MIN_US, SMM_SPLIT_NS (0: from smm_band_frame_ns), SMM_SPLIT_MARGIN: choose_split's model
    int split_margin = 400;
    int plan_cache = 1;       // SMM_PLAN_CACHE=0: no resident plans
    int small_wg = 1;         // SMM_SMALL_WG=0: no four-wave workgroups for <= 16-state videos; 2: wherever they apply, whatever the part's size (tests; same results)
    int chunk = 1;            // SMM_CHUNK=0: no time-split decode of long videos (same results)
    int chunk_p = 0;          // SMM_CHUNK_P: positions per unit of a time-split decode (0: from the launch's CU-time; tests force small ones)
    int chunk_lmin = 0;       // SMM_CHUNK_LMIN: least own part of a unit (0: SMM_CHUNK_LMIN below; never less than kp - 1 or 130)
    int chunk_wc = 512;       // SMM_CHUNK_WC: warm-up positions of a unit in front of the kp - 1 it is certified on (at most kp)
    int no_bt_window = 0;     // SMM_NO_BT_WINDOW=1: the general back-trace also for kp <= 64 (same results)
    int fit_grid = 0;         // SMM_FIT_GRID: workgroups of the class-sums kernel (tuning aid)
    int verbose = 0;          // SMM_VERBOSE
#ifdef SMM_DEV
    int debug_flags = 0;      // SMM_DEBUG_FLAGS: SmmDpArgs::flags bits (bit 0: stop after the forward pass -- outputs UNDEFINED)
    int split_debug = 0;      // SMM_SPLIT_DEBUG: leave parts of a split decode out (results INCOMPLETE)
    int upload_memcpy = 0;    // SMM_UPLOAD_MEMCPY: metadata by hipMemcpyAsync instead of kernel arguments
    int emission_v2 = 0;      // SMM_EMISSION_V2: the LDS-staged emission variant (slower)
#endif
    std::string key;          // what the planning functions depend on, for the resident plans' keys
};
// The current set of switches.  Other threads hold `const SmmEnv &` from env() across a whole call (plan_key reads
// env().key), so a reload never writes into a struct that has been published: it publishes a NEW one, and the old ones
// stay allocated (a few hundred bytes per smm_env_reload(), which only tests and A/B scripts call).
std::atomic<const SmmEnv *> g_env{nullptr};
std::once_flag g_env_once;

void env_read()
{
    SmmEnv e;
    struct Item { const char *name; int *i; double *d; };
    const Item items[] = {
        {"SMM_SPEC", &e.spec, nullptr}, {"SMM_NO_SPLIT", &e.no_split, nullptr}, {"SMM_SPLIT_MIN_US", nullptr, &e.split_min_us},
        {"SMM_SPLIT_NS", nullptr, &e.split_ns}, {"SMM_SPLIT_MARGIN", &e.split_margin, nullptr},
        {"SMM_PLAN_CACHE", &e.plan_cache, nullptr}, {"SMM_SMALL_WG", &e.small_wg, nullptr}, {"SMM_CHUNK", &e.chunk, nullptr}, {"SMM_CHUNK_P", &e.chunk_p, nullptr},
        {"SMM_CHUNK_WC", &e.chunk_wc, nullptr}, {"SMM_CHUNK_LMIN", &e.chunk_lmin, nullptr}, {"SMM_NO_BT_WINDOW", &e.no_bt_window, nullptr},
        {"SMM_FIT_GRID", &e.fit_grid, nullptr}, {"SMM_VERBOSE", &e.verbose, nullptr},
#ifdef SMM_DEV
        {"SMM_DEBUG_FLAGS", &e.debug_flags, nullptr}, {"SMM_SPLIT_DEBUG", &e.split_debug, nullptr},
        {"SMM_UPLOAD_MEMCPY", &e.upload_memcpy, nullptr}, {"SMM_EMISSION_V2", &e.emission_v2, nullptr},
#endif
    };
    for (const Item &it : items) {
        const char *v = std::getenv(it.name);
        if (!v) continue;
        if (it.i) *it.i = (*v == 0) ? 1 : std::atoi(v);        // (a switch set to the empty string counts as set)
        if (it.d) *it.d = std::atof(v);
        e.key += it.name; e.key += '='; e.key += v; e.key += ';';
    }
    g_env.store(new SmmEnv(e), std::memory_order_release);
}
const SmmEnv &env()
{
    std::call_once(g_env_once, env_read);
    return *g_env.load(std::memory_order_acquire);
}
}  // namespace

extern "C" void smm_env_reload(void)
{
    (void)env();
    env_read();
}
int smm_env_fit_grid() { return env().fit_grid; }
int smm_env_emission_v2()
{
#ifdef SMM_DEV
    return env().emission_v2;
#else
    return 0;
#endif
}

#define SMM_HIP(call)                                                         \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) { g_last_hip = (int)e_; return SMM_ERR_HIP; }   \
    } while (0)

extern "C" const char *smm_strerror(int status)
{
    switch (status) {
    case SMM_OK: return "ok";
    case SMM_ERR_ARG: return "invalid argument";
    case SMM_ERR_UNSUPPORTED: return "shape not supported by the compiled kernels";
    case SMM_ERR_WORKSPACE: return "workspace too small";
    case SMM_ERR_HIP: return "HIP runtime error";
    case SMM_ERR_NO_DEVICE: return "no gfx950 device";
    default: return "unknown smm_status";
    }
}

extern "C" int smm_last_hip_error(void) { return g_last_hip; }
extern "C" const char *smm_version(void) { return "smmdp 0.1 (gfx950)"; }

extern "C" int smm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------------------------------ metadata upload
struct SmmMetaChunk { uint32_t w[512]; };

__global__ void __launch_bounds__(512) smm_meta_upload_kernel(SmmMetaChunk c, uint32_t *dst, int n)
{
    // the chunk is the first kernel argument: read it from the kernel-argument segment with a per-thread offset (indexing
    // the by-value struct would make the compiler copy all of it into scratch first)
    typedef const __attribute__((address_space(4))) uint32_t *smm_kernarg_ptr;
    (void)c;
#if defined(__HIP_DEVICE_COMPILE__)
    smm_kernarg_ptr ka = (smm_kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    if ((int)threadIdx.x < n) dst[threadIdx.x] = ka[threadIdx.x];
#else
    (void)dst; (void)n;
#endif
}

int smm_upload_meta(void *dst_dev, const void *src_host, size_t bytes, hipStream_t stream)
{
#ifdef SMM_DEV
    if (env().upload_memcpy) return (int)hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, stream);
#endif
    const size_t words = (bytes + 3) / 4;                    // (every destination is padded to 256 B by the planners)
    const unsigned char *src = static_cast<const unsigned char *>(src_host);
    for (size_t off = 0; off < words; off += 512) {
        SmmMetaChunk c;
        const size_t n = std::min<size_t>(512, words - off);
        const size_t nb = std::min<size_t>(n * 4, bytes - off * 4);
        std::memset(&c, 0, sizeof(c));
        std::memcpy(c.w, src + off * 4, nb);
        hipLaunchKernelGGL(smm_meta_upload_kernel, dim3(1), dim3(512), 0, stream, c, static_cast<uint32_t *>(dst_dev) + off, (int)n);
    }
    return (int)hipGetLastError();
}

// Zero fill by a kernel of ours instead of hipMemsetAsync: a memset node captured into a hipGraph came back from the
// SECOND replay on with a 16-byte pattern of stale kernel arguments instead of zeros (ROCm 7.2, found by
// tests/test_gpu_graph.py: error words and gang counters full of pointers) -- kernel nodes replay correctly.
__global__ void __launch_bounds__(256) smm_zero_kernel(uint32_t *p, size_t nwords)
{
    const size_t head = ((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) >> 2;    // words before 16-byte alignment
    const size_t h = head < nwords ? head : nwords;
    const size_t nq = (nwords - h) >> 2;                                               // 16-byte pieces
    uint4 *q = reinterpret_cast<uint4 *>(p + h);
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    for (size_t i = tid; i < nq; i += nth) q[i] = make_uint4(0, 0, 0, 0);
    if (tid < h) p[tid] = 0;
    const size_t tail0 = h + 4 * nq;
    if (tid < nwords - tail0) p[tail0 + tid] = 0;
}

// Several regions in ONE launch (grid.y = region): the gradient buffers of a backward entry point were zeroed by three to six
// launches of ~4 us each back to back -- 15 per training step on cfg4, a quarter of everything that is not the DP there.
struct SmmZeroRegions { uint32_t *p[8]; size_t nwords[8]; };
__global__ void __launch_bounds__(256) smm_zero_multi_kernel(SmmZeroRegions r)
{
    uint32_t *p = r.p[blockIdx.y];
    const size_t nwords = r.nwords[blockIdx.y];
    const size_t head = ((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) >> 2;
    const size_t h = head < nwords ? head : nwords;
    const size_t nq = (nwords - h) >> 2;
    uint4 *q = reinterpret_cast<uint4 *>(p + h);
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    for (size_t i = tid; i < nq; i += nth) q[i] = make_uint4(0, 0, 0, 0);
    if (tid < h) p[tid] = 0;
    const size_t tail0 = h + 4 * nq;
    if (tid < nwords - tail0) p[tail0 + tid] = 0;
}

int smm_zero_multi_async(void *const *dst_dev, const size_t *bytes, int n, hipStream_t stream)
{
    SmmZeroRegions r{};
    int m = 0;
    size_t most = 0;
    for (int i = 0; i < n; ++i) {
        if (bytes[i] == 0) continue;
        if (m == 8 || (bytes[i] & 3) || (reinterpret_cast<uintptr_t>(dst_dev[i]) & 3)) {       // (never: at most six word-made buffers here)
            const int rc = smm_zero_async(dst_dev[i], bytes[i], stream);
            if (rc != (int)hipSuccess) return rc;
            continue;
        }
        r.p[m] = static_cast<uint32_t *>(dst_dev[i]);
        r.nwords[m] = bytes[i] >> 2;
        most = r.nwords[m] > most ? r.nwords[m] : most;
        ++m;
    }
    if (m == 0) return (int)hipSuccess;
    size_t blocks = (most / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(smm_zero_multi_kernel, dim3((unsigned)blocks, (unsigned)m), dim3(256), 0, stream, r);
    return (int)hipGetLastError();
}

int smm_zero_async(void *dst_dev, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return (int)hipSuccess;
    if ((bytes & 3) || (reinterpret_cast<uintptr_t>(dst_dev) & 3)) return (int)hipMemsetAsync(dst_dev, 0, bytes, stream);   // (never: every buffer here is made of 4- or 8-byte words)
    const size_t nwords = bytes >> 2;
    size_t blocks = (nwords / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(smm_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<uint32_t *>(dst_dev), nwords);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ planning
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct SmmPlan {
    size_t meta_bytes;     // SmmVideo[b] | order[b] | n_states[g] | emission block table[b+1] | err
    size_t o_order, o_nstates, o_emcum, o_err;
    size_t hist_doubles;   // sum over videos of 8*c_max*(T+1): forward cumE/h/gamma, backward cumE/h/gamma, 2 transposes
    size_t elp_doubles;    // total_frames*c_max  (smm_decode_f32 / smm_viterbi_f32)
    size_t tab_doubles;    // widened tables       (smm_viterbi_f32)
    size_t band_doubles;   // Viterbi BAND mode: state-major length table + skip-test bounds per (group, state)
    size_t b;              // videos
    size_t chunk_bytes;    // Viterbi, time-split videos: redo words | anchors | (not resident: the units' metadata); an upper bound
    size_t o_chunk;        // ... its offset in the workspace
    size_t total;
};

// Upper bounds of a time-split plan (the workspace is sized before anything is planned): a unit's own part is never shorter
// than SMM_CHUNK_LMIN positions, so a video of T frames has at most T / SMM_CHUNK_LMIN + 1 units
constexpr int SMM_CHUNK_LMIN = 512;                 // ... or 2 kp where that is less (kp > 64: at least 130)
constexpr int SMM_CHUNK_LMIN_FINE = 384;            // (plan_chunks: where one more unit per video still fits one round of workgroups)
static size_t chunk_units_max(const smm_shape *s) { return (size_t)(s->total_frames / 128) + 2 * (size_t)s->b; }
static size_t chunk_ext_meta_bytes(size_t b, size_t units, size_t n_cv)
{
    return align_up(sizeof(SmmVideo) * (b + units), 256) + align_up(sizeof(int32_t) * (b + units), 256) +
           align_up(sizeof(SmmChunkVideo) * n_cv, 256) + align_up(sizeof(int32_t) * n_cv, 256);
}

static bool shape_ok(const smm_shape *s)
{
    return s && s->b > 0 && s->n_groups > 0 && s->c_max > 0 && s->k_rows >= 2 && s->t_max > 0 && s->total_frames > 0;
}

static SmmPlan make_plan(const smm_shape *s, const int64_t *lengths)
{
    SmmPlan p{};
    p.b = (size_t)s->b;
    p.o_order = align_up(sizeof(SmmVideo) * s->b, 256);
    p.o_nstates = p.o_order + align_up(sizeof(int32_t) * s->b, 256);
    p.o_emcum = p.o_nstates + align_up(sizeof(int32_t) * s->n_groups, 256);
    p.o_err = p.o_emcum + align_up(sizeof(int32_t) * ((size_t)s->b + 1), 256);
    p.meta_bytes = p.o_err + 512;   // error word + diagnostic counters
    size_t h = 0;
    for (int i = 0; i < s->b; ++i) h += 8 * (size_t)s->c_max * (size_t)(lengths[i] + 1);
    p.hist_doubles = h;
    p.elp_doubles = (size_t)s->total_frames * s->c_max;
    p.tab_doubles = (size_t)s->n_groups * s->c_max * ((size_t)s->c_max + 1 + s->k_rows) + (size_t)s->b * s->c_max;
    p.band_doubles = (size_t)s->n_groups * s->c_max * ((size_t)SMM_BAND_ROW + SMM_BAND_TAB + 64);   // len_t | band_tab | dmin_t
    {
        const size_t um = chunk_units_max(s);
        p.chunk_bytes = align_up(sizeof(int32_t) * s->b, 256) + align_up(sizeof(double) * um * s->c_max, 256) +
                        chunk_ext_meta_bytes(s->b, um, s->b);
    }
    p.o_chunk = align_up(p.meta_bytes + 8 * (p.hist_doubles + p.elp_doubles + p.tab_doubles + p.band_doubles), 256);
    p.total = p.o_chunk + p.chunk_bytes + 1024;
    return p;
}

extern "C" size_t smm_workspace_bytes(const smm_shape *shape, const int64_t *lengths_host)
{
    if (!shape_ok(shape) || !lengths_host) return 0;
    for (int i = 0; i < shape->b; ++i)
        if (lengths_host[i] < 1 || lengths_host[i] > shape->t_max) return 0;
    return make_plan(shape, lengths_host).total;
}

extern "C" size_t smm_error_word_offset(const smm_shape *shape)
{
    if (!shape_ok(shape)) return 0;
    return align_up(sizeof(SmmVideo) * shape->b, 256) + align_up(sizeof(int32_t) * shape->b, 256) +
           align_up(sizeof(int32_t) * shape->n_groups, 256) + align_up(sizeof(int32_t) * ((size_t)shape->b + 1), 256);
}

struct Staged {
    SmmVideo *videos;
    int32_t *order;
    int32_t *n_states;
    int32_t *err;
    double *hist;
    double *elp;
    double *tabs;
    double *band;          // [g][c_max][k_rows] len_t | [g][c_max][16] band bounds
    bool band_mode;        // Viterbi at K > 512: BAND mode (smm_viterbi.hip)
    int32_t *em_cum;       // emission: workgroups before each video of `order` ([b + 1])
    std::vector<int32_t> em_cum_host;   // (decode split: the same table on the host)
    int em_tpw, em_blocks;
    int kp_max, c_need;
    int n_split;           // decode only: the first n_split videos of `order` are the launch's critical path (0: no split)
    // time-split videos (Viterbi; smm_chunk.hip).  n_cv = 0: none.  uvideos = [the b videos | the units]; uorder = the DP
    // launch's order over [unsplit videos, units] -- the units first (u_part1 of them: with a stream split they are its first part)
    SmmVideo *uvideos;
    int32_t *uorder, *porder;     // porder: the split videos, for the repair launch
    SmmChunkVideo *cvs;
    int32_t *redo;
    double *anchors;
    int n_cv, n_units, u_part1;
    // the LAST part of the DP launch order (the rest of a stream split, or the whole launch) ends with small_count units of <= 16
    // states that run in four-wave workgroups, two per CU, beside the others (smm_viterbi.hip: smm_launch_viterbi_small); 0: none
    int small_first, small_count;
};

static bool band_mode(int kp_max, int c_need);

// Validates the metadata, builds SmmVideo[] (+ longest-first block order) and stages it into the workspace.
// for_viterbi: the call launches the Viterbi kernel (part of a resident plan's key; rounds 1-3 planned gangs here: only
// the entry points that launch that kernel ask for it)
// Split of a decode (smm_decode_f32) into [critical videos | the rest].  The DP kernel's time is the time of the launch's
// longest videos (one workgroup each, ~0.3 us per frame), while the emission scorer in front of it streams EVERY video's
// features (0.6 ms on cfg3).  With the few longest videos scored first, their DP can start at once and the rest of the
// corpus is scored -- and then decoded -- on a second stream beside it: the second part must be through before the first
// is, i.e. its longest video has to be shorter than the launch's longest by the time the emission scorer takes.
// Returns how many videos of `order` (reordered: critical ones first, both parts keep their order) form the first part;
// 0: no split (few videos, a flat length distribution, or nothing to hide).
static int device_cus()
{
    static std::mutex mu;
    static int n_cu[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> lock(mu);
    if (!n_cu[dev] && hipDeviceGetAttribute(&n_cu[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu[dev] = 0;
    return n_cu[dev] > 0 ? n_cu[dev] : 0;
}

// ns per frame of one BAND-mode video on its CU (include/smmdp.h).  Measured, round 4's final kernels, CrossTask-like
// lattices of 64 x 4096 frames (profiles/round4_band_stamps.txt): 167 / 172 / 186 / 191 ns at 11 / 16 / 20 / 23 states.
// Re-fit these two numbers when the kernel changes; everything that needs the kernel's speed reads them from here.
constexpr double SMM_BAND_NS_BASE = 145.0, SMM_BAND_NS_PER_STATE = 2.0;
extern "C" double smm_band_frame_ns(int n_states)
{
    if (n_states < 1 || n_states > SMM_MAX_STATES) return 0.0;
    return SMM_BAND_NS_BASE + SMM_BAND_NS_PER_STATE * n_states;
}

static int choose_split(const SmmVideo *hv, int32_t *order, int b, int d, int c_max, int64_t total_frames)
{
    const SmmEnv &ev = env();
    if (b < 24 || ev.no_split) return 0;
    const int n_cu = device_cus();
    if (n_cu <= 0) return 0;
    const double em_us = (double)total_frames * (4.0 * d + 8.0 * c_max) / 4.0e6;      // ~4 TB/s of algorithmic bytes
    if (em_us < ev.split_min_us) return 0;                                             // (SMM_SPLIT_MIN_US: test hook, split small launches too)
    int tmax = 0;
    for (int i = 0; i < b; ++i) tmax = std::max(tmax, hv[i].T);
    // a video of the second part starts em_us later than the critical ones and must not outlast them: the critical set is
    // the videos within em_us / split_ns + margin frames of the longest.  Same-box scan on cfg3 (scripts/gpu_ab_cfg3.sh,
    // BAND kernel of round 3, ~250 ns per frame beside a full GPU): 230 / 250 ns 3.72-3.74 ms per step, 300 3.58-3.62,
    // 400 3.59-3.62, 600 3.65-3.71, 1000+ 3.68-3.70 -- the em_us estimate below (4 TB/s) is already on the long side.
    // SMM_SPLIT_NS / SMM_SPLIT_MARGIN: tuning aids
    // (round 5: the slack per microsecond of emission follows the kernel's own speed -- split_slack x the modelled ns per
    // frame of the launch's largest class set, 1.75 x 191 = 334 at 23 states, the value the scans above settled on when it
    // was a literal -- instead of being re-fitted by hand each time the DP kernel gets faster; SMM_SPLIT_NS overrides)
    const double split_ns = ev.split_ns > 0.0 ? ev.split_ns : 1.75 * smm_band_frame_ns(std::min(c_max, (int)SMM_MAX_STATES));
    const int thr = tmax - (int)(em_us * 1000.0 / split_ns) - ev.split_margin;
    int n1 = 0;
    for (int i = 0; i < b; ++i) n1 += hv[i].T >= thr;
    // ... and only a launch that is bound by its longest videos: with more than two rounds of workgroups the DP is bound by
    // CU-time and the split only adds a second launch (scripts/sweep_split.py, round 4: 1000 videos of ~2000 frames lost
    // 3.8 % with it; every latency-bound distribution won 5..10 % or stayed level)
    if (thr <= 0 || n1 < 1 || n1 > b / 3 || n1 > n_cu / 2 || b - n1 < 16 || b > 2 * n_cu) return 0;
    std::stable_partition(order, order + b, [&](int32_t v) { return hv[v].T >= thr; });
    return n1;
}

// ---------------------------------------------------------------------------------------------------------------
// Resident plans.  Staging a call is host work (ordering the videos, the split, the emission grid) plus the upload of
// the metadata in 2 KB kernel-argument chunks -- about 0.2 ms in front of the first real kernel of a 360-video
// launch, every call, although a training loop decodes the SAME batches every epoch.  A staged call is therefore kept:
// the immutable part of the metadata (videos | order | n_states | emission block table) lives in a device buffer the
// LIBRARY owns -- nobody else can write to it, so a later call with bit-identical inputs (compared in full, not by hash)
// and the same planning environment points its kernels at that buffer and skips planning and upload.  The mutable
// part (the error words) stays in the caller's workspace and is cleared per call as before.
// Lifetime: a plan is ADMITTED the second time its inputs are seen (a training loop that packs shuffled batches shows a new
// key every step: those are staged the old way and leave nothing behind); plans live in slabs of the device they were
// made on, 64 MB at most per process, and stay until smm_release_cached_plans() frees them (the caller's promise: no
// call in flight and no captured graph that was captured from one of these calls still to be replayed).  With
// SMM_PLAN_CACHE=0, past the cap, or when a call with new inputs happens under stream capture (no allocation, no event
// query there), a call is staged into its workspace as before.
namespace {
struct PlanEntry {
    std::vector<char> key;
    int device = 0;
    char *dev_meta = nullptr;
    SmmPlan plan{};
    Staged st{};
    hipEvent_t ready = nullptr;
    hipStream_t stream = nullptr;
    bool settled = false;
};
struct PlanSlabs {                                                      // per device: metadata buffers are cut from 1 MB slabs
    char *cur = nullptr;
    size_t left = 0;
    std::vector<std::pair<char *, size_t>> all;                         // (slab, its real size: a plan larger than 1 MB gets its own)
};
struct PlanCache {
    std::mutex mu;
    std::unordered_map<uint64_t, std::vector<PlanEntry *>> entries;     // by FNV-1a of the key; compared in full on a hit
    std::unordered_map<uint64_t, uint32_t> seen_once;                   // keys seen once (hash only): admitted at the second sighting
    size_t n = 0, bytes = 0, key_bytes = 0;                            // plans, device bytes, host bytes of their keys
    PlanSlabs slabs[64];
} g_plans;
constexpr size_t SMM_PLAN_MAX_BYTES = (size_t)64 << 20, SMM_PLAN_MAX_ENTRIES = 8192, SMM_PLAN_SLAB = (size_t)1 << 20;

uint64_t plan_hash(const std::vector<char> &k)
{
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : k) { h ^= c; h *= 1099511628211ull; }
    return h;
}

// (called with the cache locked; `dev` is the current device: a plan's buffer lives on the device its kernels run on)
char *plan_alloc(size_t bytes, int dev)
{
    if (dev < 0 || dev >= 64) return nullptr;
    PlanSlabs &sl = g_plans.slabs[dev];
    bytes = align_up(bytes, 256);
    if (bytes > sl.left) {
        const size_t want = std::max(bytes, SMM_PLAN_SLAB);
        char *p = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&p), want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        sl.all.emplace_back(p, want);      // (the rest of the previous slab is given up)
        sl.cur = p;
        sl.left = want;
    }
    char *r = sl.cur;
    sl.cur += bytes;
    sl.left -= bytes;
    return r;
}

void plan_key_append(std::vector<char> &k, const void *p, size_t n)
{
    const char *c = static_cast<const char *>(p);
    k.insert(k.end(), c, c + n);
}

std::vector<char> plan_key(const smm_shape *s, const int64_t *lengths, const int64_t *frame_off, const int32_t *group,
                           const int32_t *kp, const int32_t *n_states, bool for_viterbi, int cum_chunk, bool want_split)
{
    std::vector<char> k;
    k.reserve(sizeof(*s) + (size_t)s->b * 24 + (size_t)s->n_groups * 4 + 128);
    plan_key_append(k, s, sizeof(*s));
    plan_key_append(k, lengths, sizeof(int64_t) * s->b);
    plan_key_append(k, frame_off, sizeof(int64_t) * s->b);
    const char has[2] = {(char)(group != nullptr), (char)(kp != nullptr)};
    plan_key_append(k, has, 2);
    if (group) plan_key_append(k, group, sizeof(int32_t) * s->b);
    if (kp) plan_key_append(k, kp, sizeof(int32_t) * s->b);
    plan_key_append(k, n_states, sizeof(int32_t) * s->n_groups);
    const int32_t f[3] = {for_viterbi, cum_chunk, want_split};
    plan_key_append(k, f, sizeof(f));
    // the switches the planning functions read (smm_env_reload() may change them between calls)
    const std::string &ek = env().key;
    plan_key_append(k, ek.data(), ek.size() + 1);
    return k;
}

void plan_point(const SmmPlan &p, void *ws, Staged *out)
{
    char *base = static_cast<char *>(ws);
    out->err = reinterpret_cast<int32_t *>(base + p.o_err);
    out->hist = reinterpret_cast<double *>(base + p.meta_bytes);
    out->elp = out->hist + p.hist_doubles;
    out->tabs = out->elp + p.elp_doubles;
    out->band = out->tabs + p.tab_doubles;
    out->redo = reinterpret_cast<int32_t *>(base + p.o_chunk);
    out->anchors = reinterpret_cast<double *>(base + p.o_chunk + align_up(sizeof(int32_t) * p.b, 256));
}
}  // namespace

// meta_alloc(bytes): where the IMMUTABLE metadata of this call goes when it is to be kept as a resident plan (called once,
// with the exact size, after the host-side planning); nullptr result / no function: into the workspace
static int stage_uncached(const smm_shape *s, const int64_t *lengths, const int64_t *frame_off, const int32_t *group,
                          const int32_t *kp, const int32_t *n_states, void *ws, size_t ws_bytes, hipStream_t stream, Staged *out,
                          bool for_viterbi, int cum_chunk, bool want_split, const std::function<char *(size_t)> *meta_alloc,
                          SmmPlan *plan_out);

static int stage(const smm_shape *s, const int64_t *lengths, const int64_t *frame_off, const int32_t *group,
                 const int32_t *kp, const int32_t *n_states, void *ws, size_t ws_bytes, hipStream_t stream, Staged *out,
                 bool for_viterbi = false, int cum_chunk = 0, bool want_split = false)
{
    if (!shape_ok(s) || !lengths || !frame_off || !n_states || !ws) return SMM_ERR_ARG;
    if (!env().plan_cache) return stage_uncached(s, lengths, frame_off, group, kp, n_states, ws, ws_bytes, stream, out, for_viterbi, cum_chunk, want_split, nullptr, nullptr);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
    const bool capturing = cap != hipStreamCaptureStatusNone;
    const std::vector<char> key = plan_key(s, lengths, frame_off, group, kp, n_states, for_viterbi, cum_chunk, want_split);
    const uint64_t hk = plan_hash(key) ^ (uint64_t)dev;
    PlanEntry *hit = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_plans.mu);
        auto it = g_plans.entries.find(hk);
        if (it != g_plans.entries.end())
            for (PlanEntry *e : it->second)
                if (e->device == dev && e->key.size() == key.size() && std::memcmp(e->key.data(), key.data(), key.size()) == 0) { hit = e; break; }
        if (hit && !hit->settled) {
            // the upload was queued on another stream, maybe: usable once it is known to be through (or on that very
            // stream).  Under capture nothing is queried or waited for (an event query can invalidate a global-mode capture):
            // an unsettled plan is then simply not used.
            if (capturing) { if (hit->stream != stream) hit = nullptr; }
            else if (hipEventQuery(hit->ready) == hipSuccess) hit->settled = true;
            else if (hit->stream != stream && hipStreamWaitEvent(stream, hit->ready, 0) != hipSuccess) hit = nullptr;
        }
    }
    if (hit) {
        if (ws_bytes < hit->plan.total) return SMM_ERR_WORKSPACE;
        *out = hit->st;
        plan_point(hit->plan, ws, out);
        char *base = static_cast<char *>(ws);
        SMM_HIP((hipError_t)smm_zero_async(base + hit->plan.o_err, hit->plan.meta_bytes - hit->plan.o_err, stream));
        return SMM_OK;
    }
    // miss: stage, and keep the plan unless this is a capture (no allocation there) or the cache is full
    char *dev_meta = nullptr;
    bool keep = !capturing;
    if (keep) {
        std::lock_guard<std::mutex> lock(g_plans.mu);
        // admitted at the second sighting: one-off batches (shuffled training steps) never take a slot
        auto seen = g_plans.seen_once.find(hk);
        if (seen == g_plans.seen_once.end()) {
            if (g_plans.seen_once.size() >= 4 * SMM_PLAN_MAX_ENTRIES) g_plans.seen_once.clear();
            g_plans.seen_once.emplace(hk, 1u);
            keep = false;
        }
    }
    // (called by stage_uncached once the size of the immutable metadata is known: the videos, and the units of a time-split plan)
    const std::function<char *(size_t)> alloc = [&](size_t bytes) -> char * {
        std::lock_guard<std::mutex> lock(g_plans.mu);
        if (!(g_plans.bytes + bytes <= SMM_PLAN_MAX_BYTES && g_plans.n < SMM_PLAN_MAX_ENTRIES &&
              g_plans.key_bytes + key.size() <= SMM_PLAN_MAX_BYTES)) return nullptr;
        dev_meta = plan_alloc(bytes, dev);
        if (dev_meta) g_plans.bytes += align_up(bytes, 256);
        return dev_meta;
    };
    SmmPlan plan{};
    const int rc = stage_uncached(s, lengths, frame_off, group, kp, n_states, ws, ws_bytes, stream, out, for_viterbi, cum_chunk, want_split,
                                  keep ? &alloc : nullptr, &plan);
    keep = keep && dev_meta != nullptr;
    if (rc != SMM_OK || !keep) return rc;          // (a buffer cut for a call that failed stays cut: 64 MB bound the total)
    PlanEntry *e = new PlanEntry;
    e->key = key;
    e->device = dev;
    e->dev_meta = dev_meta;
    e->plan = plan;
    e->st = *out;
    e->stream = stream;
    if (hipEventCreateWithFlags(&e->ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(e->ready, stream) != hipSuccess) {
        // (cannot tell later whether the upload is through: this call is fine -- same stream -- but the plan is not kept)
        if (e->ready) (void)hipEventDestroy(e->ready);
        delete e;
        return SMM_OK;     // dev_meta stays allocated: this call's kernels read it
    }
    std::lock_guard<std::mutex> lock(g_plans.mu);
    g_plans.entries[hk].push_back(e);
    g_plans.seen_once.erase(hk);
    g_plans.n += 1;
    g_plans.key_bytes += key.size();
    return SMM_OK;
}

// Frees every resident plan and its device memory (all devices), the decode's second streams and the pooled events.
// The caller's promise: no libsmmdp call is in flight on any stream, and no hipGraph captured from one of these calls will
// be replayed again (a captured call points at its plan's buffer).  Returns the device bytes given back.
static void release_aux();
extern "C" size_t smm_release_cached_plans(void)
{
    release_aux();
    std::lock_guard<std::mutex> lock(g_plans.mu);
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    size_t freed = 0;
    for (auto &kv : g_plans.entries)
        for (PlanEntry *e : kv.second) {
            if (e->ready) (void)hipEventDestroy(e->ready);
            delete e;
        }
    g_plans.entries.clear();
    g_plans.seen_once.clear();
    for (int d = 0; d < 64; ++d) {
        PlanSlabs &sl = g_plans.slabs[d];
        if (sl.all.empty()) continue;
        (void)hipSetDevice(d);
        for (auto &ps : sl.all) if (hipFree(ps.first) == hipSuccess) freed += ps.second;
        sl = PlanSlabs{};
    }
    if (have_cur) (void)hipSetDevice(cur);
    g_plans.n = g_plans.bytes = g_plans.key_bytes = 0;
    return freed;
}

extern "C" size_t smm_cached_plan_bytes(void)
{
    std::lock_guard<std::mutex> lock(g_plans.mu);
    size_t n = 0;
    for (int d = 0; d < 64; ++d)
        for (auto &ps : g_plans.slabs[d].all) n += ps.second;
    return n;
}

// Time-split plan of a Viterbi launch (smm_chunk.hip).  A launch lasts as long as its longest video -- one serial chain -- while
// its CU-time may be a fraction of that: videos longer than P positions are cut into units of about P positions each (unit
// 0: the video's first P'; the others: OV = warm-up + kp - 1 positions in front of an own part of L' = P' - OV), P from the
// launch's CU-time (SMM_CHUNK_P overrides).  An own part is at least kp - 1 positions (the window a unit is certified on
// must lie inside the previous unit's own part) and at least SMM_CHUNK_LMIN; the units' histories live in the video's own
// history block (8 c_max (T + 1) doubles: a unit needs 3 C (T_u + 1)).
struct ChunkPlan {
    std::vector<SmmVideo> units;
    std::vector<SmmChunkVideo> cvs;
};

static void plan_chunks_lmin(const smm_shape *s, const SmmVideo *hv, const int32_t *n_states, int kp_max, bool band, ChunkPlan &out, int n_cu_given, int lmin_cfg)
{
    const SmmEnv &ev = env();
    if (!ev.chunk || (s->flags & (SMM_SHAPE_NO_EOS | SMM_SHAPE_NO_TIME_SPLIT)) || kp_max <= 64) return;      // (kp <= 64: the window back-trace's launches are left alone)
    const int n_cu = n_cu_given > 0 ? n_cu_given : device_cus();
    if (n_cu <= 0) return;
    // (the ring kernels, span limits up to 512: cfg2's 16 states at K = 256 take 199 ns per frame; BAND mode: the library's model)
    auto ns = [&](int c) { return band ? smm_band_frame_ns(c) : 150.0 + 3.0 * c; };
    double t_cu = 0.0, t_long = 0.0;
    for (int i = 0; i < s->b; ++i) {
        const double t = (double)hv[i].T * ns(n_states[hv[i].group]);
        t_cu += t;
        t_long = std::max(t_long, t);
    }
    // the launch's CU-time: the DP's share per CU + what the emission pass in front of it (or beside it) takes of the chip
    t_cu = t_cu / n_cu + (s->d > 0 ? (double)s->total_frames * (4.0 * s->d + 8.0 * s->c_max) / 4.0e3 : 0.0);
    // Only a launch that is bound by its longest video is split.  Measured on cfg3 (360 videos, CU-time 2.3 ms against 2.3 ms
    // for its longest video; profiles/round5_time_split.txt): cutting its 46 longest videos shortens the critical launch from
    // 2.3 to 1.5 ms and the step not at all -- the other 314 videos keep 256 CUs busy for 2.1 ms either way -- while every
    // video that has to be repaired (2..4 per launch there: runs of one class decoded as two spans, whose (a, b) / (b, a) orders
    // tie to within rounding) costs a whole one-piece decode on top.  One long video on an idle GPU (cfg1) is the other end:
    // 1.70 -> 0.92 ms.  SMM_CHUNK_P forces the split (tests).
    if (ev.chunk_p <= 0 && t_long < 1.5 * t_cu) return;
    const size_t units_cap = chunk_units_max(s);
    for (int i = 0; i < s->b; ++i) {
        const int T = hv[i].T, C = n_states[hv[i].group], kpv = hv[i].kp;
        // warm-up: the recursion forgets its start behind the next segment boundary, and a segment is shorter than kp
        const int ov = std::max(std::min(ev.chunk_wc, kpv), 16) + kpv - 1;
        const int lmin = std::max(kpv - 1, std::min(lmin_cfg, 2 * kpv));
        const int pmin = ov + lmin;
        // positions per unit: what the launch's CU-time lasts on this video's state count (+ 10 %), at least pmin
        int P = ev.chunk_p > 0 ? ev.chunk_p : (int)(1.1 * t_cu / ns(C));
        P = std::max(P, pmin);
        if ((double)T < 1.05 * P || T < pmin + lmin) continue;                   // (not worth a second unit)
        int n_rest = (T - P + (P - ov) - 1) / (P - ov);
        int Pq = 0;
        for (; n_rest >= 1; --n_rest) {                                          // equal shares: P' + n_rest (P' - OV) = T
            Pq = (int)(((int64_t)T + (int64_t)n_rest * ov + n_rest) / (n_rest + 1));
            if (Pq - ov >= lmin) break;
        }
        if (n_rest < 1 || n_rest + 1 > SMM_CHUNK_MAX_UNITS) continue;
        const int L = Pq - ov;
        // unit j >= 1: own part (r_j, e_j], r_j = Pq + (j - 1) L, first position a_j = r_j - OV; the last one ends at T
        size_t need = (size_t)3 * C * ((size_t)Pq + 1);
        for (int j = 1; j <= n_rest; ++j) {
            const int r = Pq + (j - 1) * L, e = (j == n_rest) ? T : r + L;
            need += (size_t)3 * C * ((size_t)(e - (r - ov)) + 1);
        }
        if (need > (size_t)8 * s->c_max * ((size_t)T + 1) || Pq + (n_rest - 1) * L >= T) continue;
        if (out.units.size() + n_rest + 1 > units_cap) break;
        SmmChunkVideo cv{i, (int32_t)out.units.size(), n_rest + 1, ov};           // (first_unit: made absolute by the caller)
        size_t hoff = (size_t)hv[i].hist_off;
        for (int j = 0; j <= n_rest; ++j) {
            const int r = j ? Pq + (j - 1) * L : 0, a0 = j ? r - ov : 0, e = (j == n_rest) ? T : Pq + j * L;
            SmmVideo u = hv[i];
            u.frame_off = hv[i].frame_off + a0;
            u.hist_off = (int64_t)hoff;
            u.T = e - a0;
            u.pad = 1 | (j ? 2 : 0) | (a0 << 2);
            hoff += (size_t)3 * C * ((size_t)u.T + 1);
            out.units.push_back(u);
        }
        out.cvs.push_back(cv);
    }
}

// The least own part of a unit: SMM_CHUNK_LMIN, or the finer SMM_CHUNK_LMIN_FINE where the launch then still runs in ONE round of
// workgroups (units + unsplit videos <= CUs).  cfg2 (64 videos x 2048 frames, K = 256): four units per video on 256 CUs instead of
// three on 192, 0.378 -> 0.369 ms per step, two alternating runs on one box; five or six units per video (320 / 384 workgroups, a
// second round): 0.49-0.50 ms.  SMM_CHUNK_LMIN (environment) overrides both.
static void plan_chunks(const smm_shape *s, const SmmVideo *hv, const int32_t *n_states, int kp_max, bool band, ChunkPlan &out, int n_cu_given = 0)
{
    const SmmEnv &ev = env();
    if (ev.chunk_lmin > 0) { plan_chunks_lmin(s, hv, n_states, kp_max, band, out, n_cu_given, std::max(ev.chunk_lmin, 130)); return; }
    const int n_cu = n_cu_given > 0 ? n_cu_given : device_cus();
    ChunkPlan fine;
    plan_chunks_lmin(s, hv, n_states, kp_max, band, fine, n_cu_given, SMM_CHUNK_LMIN_FINE);
    if (!fine.cvs.empty() && (int)fine.units.size() + (s->b - (int)fine.cvs.size()) <= n_cu) { out = std::move(fine); return; }
    plan_chunks_lmin(s, hv, n_states, kp_max, band, out, n_cu_given, SMM_CHUNK_LMIN);
}

static int stage_uncached(const smm_shape *s, const int64_t *lengths, const int64_t *frame_off, const int32_t *group,
                          const int32_t *kp, const int32_t *n_states, void *ws, size_t ws_bytes, hipStream_t stream, Staged *out,
                          bool for_viterbi, int cum_chunk, bool want_split, const std::function<char *(size_t)> *meta_alloc,
                          SmmPlan *plan_out)
{
    if (!shape_ok(s) || !lengths || !frame_off || !n_states || !ws) return SMM_ERR_ARG;
    if (s->c_max > SMM_MAX_STATES || s->k_rows > SMM_MAX_K_ROWS) return SMM_ERR_UNSUPPORTED;
    int c_need = 0;
    for (int g = 0; g < s->n_groups; ++g) {
        if (n_states[g] < 1 || n_states[g] > s->c_max) return SMM_ERR_ARG;
        c_need = std::max(c_need, n_states[g]);
    }
    const bool no_eos = (s->flags & SMM_SHAPE_NO_EOS) != 0;
    for (int i = 0; i < s->b; ++i)
        if (lengths[i] < (no_eos ? 2 : 1) || lengths[i] > s->t_max) return SMM_ERR_ARG;
    const SmmPlan p = make_plan(s, lengths);
    if (ws_bytes < p.total) return SMM_ERR_WORKSPACE;

    std::vector<char> host(p.meta_bytes, 0);
    SmmVideo *hv = reinterpret_cast<SmmVideo *>(host.data());
    int32_t *ho = reinterpret_cast<int32_t *>(host.data() + p.o_order);
    int32_t *hn = reinterpret_cast<int32_t *>(host.data() + p.o_nstates);

    size_t hoff = 0;
    int kp_max = 2;
    for (int i = 0; i < s->b; ++i) {
        const int64_t t = lengths[i];
        if (frame_off[i] < 0 || frame_off[i] + t > s->total_frames) return SMM_ERR_ARG;
        const int g = group ? group[i] : 0;
        if (g < 0 || g >= s->n_groups) return SMM_ERR_ARG;
        const int k = kp ? kp[i] : std::min<int>(s->k_rows, s->t_max);
        if (k < 1 || k > s->k_rows) return SMM_ERR_ARG;
        hv[i].frame_off = frame_off[i];
        hv[i].hist_off = (int64_t)hoff;
        hv[i].T = (int32_t)t;                      // (no EOS: the DP kernels take T - 1, the emission kernel every frame)
        hv[i].group = g;
        hv[i].kp = k;
        hv[i].pad = 0;
        hoff += 8 * (size_t)s->c_max * (size_t)(t + 1);
        kp_max = std::max(kp_max, k);
    }
    std::iota(ho, ho + s->b, 0);
    // most work first, so the tail of the grid is made of short videos
    std::stable_sort(ho, ho + s->b, [&](int a, int b) {
        return (int64_t)hv[a].T * n_states[hv[a].group] > (int64_t)hv[b].T * n_states[hv[b].group];
    });
    // (chunk table for the emission chain rule: by group, so that a workgroup's run of chunks rarely changes group)
    if (cum_chunk > 0) std::stable_sort(ho, ho + s->b, [&](int a, int b) { return hv[a].group < hv[b].group; });
    std::memcpy(hn, n_states, sizeof(int32_t) * s->n_groups);
    out->band_mode = band_mode(kp_max, c_need);
    // time-split plan (Viterbi launches only): the split videos come FIRST in `order` -- with a stream split they are its
    // first part (their emission, the prefix sums at their units' starts and their units' DP on the caller's stream, the
    // rest beside them)
    ChunkPlan cp;
    if (for_viterbi && cum_chunk == 0) plan_chunks(s, hv, n_states, kp_max, out->band_mode, cp);
    const int n_cv = (int)cp.cvs.size(), n_un = (int)cp.units.size();
    out->n_cv = n_cv;
    out->n_units = s->b - n_cv + n_un;
    out->u_part1 = n_un;
    out->n_split = 0;
    if (n_cv > 0) {
        std::vector<char> is_cv(s->b, 0);
        for (const SmmChunkVideo &cv : cp.cvs) is_cv[cv.vid] = 1;
        std::stable_partition(ho, ho + s->b, [&](int32_t v) { return is_cv[v] != 0; });
        // the stream split of a decode: the split videos are its critical part, if there is enough beside them to be worth a second stream
        const SmmEnv &ev = env();
        const double em_us = (double)s->total_frames * (4.0 * s->d + 8.0 * s->c_max) / 4.0e6;
        if (want_split && !ev.no_split && s->b - n_cv >= 16 && em_us >= ev.split_min_us) out->n_split = n_cv;
    } else {
        out->n_split = want_split ? choose_split(hv, ho, s->b, s->d, s->c_max, s->total_frames) : 0;
    }
    // SMALL workgroups: the last part of the launch -- the rest of a stream split, or everything that is not a unit of a
    // time-split video -- when it is bound by CU-time through and through: at least three videos per CU (cfg3's rest, 340 videos
    // on 256 CUs, lost 15 % to them: its <= 16-state videos of up to 11 800 frames run 25..60 % longer in four waves and became
    // the part's critical path; cfg5's 2754 videos gained 11 %), and only videos that stay well inside the part's CU-time at the
    // four-wave speed.  Those go to the part's END (work order kept on both sides).
    out->small_first = out->small_count = 0;
    if (for_viterbi && cum_chunk == 0 && out->band_mode && env().small_wg && !(s->flags & SMM_SHAPE_NO_EOS)) {
        const bool force = env().small_wg >= 2;
        const int n_cu = device_cus();
        const int t0 = out->n_split > 0 ? out->n_split : n_cv;              // first video of the last part, in `order`
        const int n_tail = s->b - t0;
        if (n_cu > 0 && (force || n_tail >= 3 * n_cu)) {
            double t_cu = 0.0;                                              // the part's CU-time at the eight-wave speed, ns
            for (int i = t0; i < s->b; ++i) t_cu += (double)hv[ho[i]].T * smm_band_frame_ns(n_states[hv[ho[i]].group]);
            t_cu /= n_cu;
            auto small = [&](int32_t v) {
                return n_states[hv[v].group] <= 16 && (force || (double)hv[v].T * 1.6 * smm_band_frame_ns(n_states[hv[v].group]) <= 0.5 * t_cu);
            };
            const int n_small = (int)std::count_if(ho + t0, ho + s->b, small);
            if (n_small >= (force ? 1 : 64)) {
                std::stable_partition(ho + t0, ho + s->b, [&](int32_t v) { return !small(v); });
                out->small_count = n_small;
            }
        }
    }
    {
        // emission grid (flat): video order[i] gets smm_emission_blocks(T) workgroups; in the DP's final order
        int32_t *hc = reinterpret_cast<int32_t *>(host.data() + p.o_emcum);
        out->em_tpw = smm_emission_tiles_per_wave(s->total_frames, s->b);
        int64_t cum = 0;
        for (int i = 0; i < s->b; ++i) {
            hc[i] = (int32_t)cum;
            // (cum_chunk: the table counts chunks of that many frames instead -- smm_emission_bwd_f64)
            cum += cum_chunk > 0 ? (hv[ho[i]].T + cum_chunk - 1) / cum_chunk : smm_emission_blocks(hv[ho[i]].T, out->em_tpw);
        }
        if (cum > 0x7fffffff) return SMM_ERR_UNSUPPORTED;
        hc[s->b] = (int32_t)cum;
        out->em_blocks = (int)cum;
        if (out->n_split > 0) out->em_cum_host.assign(hc, hc + s->b + 1);
    }
    // the units' metadata: [videos | units], the DP launch's order over [units (most work first) | unsplit videos (as in `order`)],
    // the split videos' table and -- for the repair launch -- their list
    std::vector<char> ext;
    size_t x_order = 0, x_cvs = 0, x_porder = 0;
    if (n_cv > 0) {
        x_order = align_up(sizeof(SmmVideo) * ((size_t)s->b + n_un), 256);
        x_cvs = x_order + align_up(sizeof(int32_t) * ((size_t)s->b + n_un), 256);
        x_porder = x_cvs + align_up(sizeof(SmmChunkVideo) * n_cv, 256);
        ext.assign(x_porder + align_up(sizeof(int32_t) * n_cv, 256), 0);
        SmmVideo *xv = reinterpret_cast<SmmVideo *>(ext.data());
        int32_t *xo = reinterpret_cast<int32_t *>(ext.data() + x_order);
        SmmChunkVideo *xc = reinterpret_cast<SmmChunkVideo *>(ext.data() + x_cvs);
        int32_t *xp = reinterpret_cast<int32_t *>(ext.data() + x_porder);
        std::memcpy(xv, hv, sizeof(SmmVideo) * s->b);
        std::memcpy(xv + s->b, cp.units.data(), sizeof(SmmVideo) * n_un);
        for (int i = 0; i < n_un; ++i) xo[i] = s->b + i;
        std::stable_sort(xo, xo + n_un, [&](int a, int b) {
            return (int64_t)xv[a].T * n_states[xv[a].group] > (int64_t)xv[b].T * n_states[xv[b].group];
        });
        for (int i = n_cv; i < s->b; ++i) xo[n_un + i - n_cv] = ho[i];
        for (int i = 0; i < n_cv; ++i) {
            xc[i] = cp.cvs[i];
            xc[i].first_unit += s->b;
            xp[i] = cp.cvs[i].vid;
        }
    }

    // (in the DP launch's order -- `order`, or with time-split videos [units | unsplit videos in `order`'s order] -- the small videos
    // are the last small_count entries)
    if (out->small_count > 0) out->small_first = out->n_units - out->small_count;

    char *base = static_cast<char *>(ws);
    // videos | order | n_states travel as kernel arguments (no pageable copy: the host never waits for the stream) -- into
    // the workspace, or into the resident plan's own buffer; the error words behind them (always in the workspace) start at zero
    const size_t o_ext = align_up(p.o_err, 256);
    char *meta_dst = meta_alloc ? (*meta_alloc)(o_ext + ext.size()) : nullptr;
    char *meta = meta_dst ? meta_dst : base;
    SMM_HIP((hipError_t)smm_upload_meta(meta, host.data(), p.o_err, stream));
    SMM_HIP((hipError_t)smm_zero_async(base + p.o_err, p.meta_bytes - p.o_err, stream));
    out->videos = reinterpret_cast<SmmVideo *>(meta);
    out->order = reinterpret_cast<int32_t *>(meta + p.o_order);
    out->n_states = reinterpret_cast<int32_t *>(meta + p.o_nstates);
    out->em_cum = reinterpret_cast<int32_t *>(meta + p.o_emcum);
    plan_point(p, ws, out);
    out->uvideos = nullptr; out->uorder = nullptr; out->porder = nullptr; out->cvs = nullptr;
    if (n_cv > 0) {
        // (workspace: behind the redo words and the anchors of the chunk region, whose size bounds every plan of this shape)
        char *xdst = meta_dst ? meta_dst + o_ext
                              : base + p.o_chunk + align_up(sizeof(int32_t) * s->b, 256) + align_up(sizeof(double) * chunk_units_max(s) * s->c_max, 256);
        SMM_HIP((hipError_t)smm_upload_meta(xdst, ext.data(), ext.size(), stream));
        out->uvideos = reinterpret_cast<SmmVideo *>(xdst);
        out->uorder = reinterpret_cast<int32_t *>(xdst + x_order);
        out->cvs = reinterpret_cast<SmmChunkVideo *>(xdst + x_cvs);
        out->porder = reinterpret_cast<int32_t *>(xdst + x_porder);
    }
    out->kp_max = kp_max;
    out->c_need = c_need;
    if (plan_out) *plan_out = p;
    return SMM_OK;
}

// The time-split plan of a Viterbi launch as the library would make it on a GPU of n_cu compute units (include/smmdp.h): host
// logic only, no device needed -- what the CPU tests look at.
extern "C" int smm_time_split_plan(const smm_shape *s, const int64_t *lengths, const int32_t *group, const int32_t *kp,
                                   const int32_t *n_states, int n_cu, int32_t *unit_video, int32_t *unit_first, int32_t *unit_len,
                                   int32_t *unit_overlap, int cap)
{
    if (!shape_ok(s) || !lengths || !n_states || n_cu < 1) return SMM_ERR_ARG;
    std::vector<SmmVideo> hv(s->b);
    size_t hoff = 0;
    int kp_max = 2, c_need = 0;
    for (int g = 0; g < s->n_groups; ++g) c_need = std::max(c_need, n_states[g]);
    for (int i = 0; i < s->b; ++i) {
        const int g = group ? group[i] : 0;
        if (g < 0 || g >= s->n_groups || lengths[i] < 1) return SMM_ERR_ARG;
        hv[i] = SmmVideo{0, (int64_t)hoff, (int32_t)lengths[i], g, kp ? kp[i] : std::min<int>(s->k_rows, s->t_max), 0};
        hoff += 8 * (size_t)s->c_max * (size_t)(lengths[i] + 1);
        kp_max = std::max(kp_max, hv[i].kp);
    }
    ChunkPlan cp;
    plan_chunks(s, hv.data(), n_states, kp_max, band_mode(kp_max, c_need), cp, n_cu);
    int n = 0;
    for (const SmmChunkVideo &cv : cp.cvs)
        for (int j = 0; j < cv.n_chunks; ++j, ++n) {
            if (n >= cap) continue;
            const SmmVideo &u = cp.units[cv.first_unit + j];
            if (unit_video) unit_video[n] = cv.vid;
            if (unit_first) unit_first[n] = u.pad >> 2;
            if (unit_len) unit_len[n] = u.T;
            if (unit_overlap) unit_overlap[n] = j ? cv.ov : 0;
        }
    return n;
}

// Viterbi at K > 512: BAND mode (smm_viterbi.hip), up to SMM_MAX_STATES states on one CU; also 29..32 states at K > 256
// (the 512-slot rings of five states per pusher wave do not fit the registers)
static bool band_mode(int kp_max, int c_need)
{
    return kp_max > 512 || (kp_max > 256 && c_need > 28);
}

static int ring_regs(int kp_max)
{
    int r = 1;
    while (64 * r < kp_max) r *= 2;
    return r;
}

// ------------------------------------------------------------------------------------------------ pieces
static int run_emission(const smm_shape *s, const Staged &st, const float *x, const double *w, const double *cst,
                        const double *inv_var, const float *cons, double *elp64, float *elp32, hipStream_t stream,
                        int first = 0, int count = -1)
{
    if (!x || !w || !cst || !inv_var || s->d < 1 || (!elp64 && !elp32)) return SMM_ERR_ARG;
    if (smm_emission_lds_bytes(s->d, st.c_need) > 160 * 1024)
        return SMM_ERR_UNSUPPORTED;   // the group's weight table must fit the CU's LDS (D <= 544 at 32 states)
    SmmEmArgs a{st.videos, st.order, st.n_states, x, w, cst, inv_var, cons, elp64, elp32, s->d, s->c_max, s->b};
    if (count < 0) smm_launch_emission(a, st.c_need, st.em_tpw, st.em_blocks, st.em_cum, s->total_frames, stream);
    else smm_launch_emission(a, st.c_need, st.em_tpw, st.em_cum_host[first + count] - st.em_cum_host[first], st.em_cum,
                             s->total_frames, stream, st.em_cum_host[first], first, count);
    SMM_HIP(hipGetLastError());
    return SMM_OK;
}

// first / count: only the videos order[first .. first + count) (count < 0: all); prep: launch the band tables kernel (a
// split decode launches it once, in front of both parts)
// smm_dp_timing_*: event pairs around the DP kernel launches (measurement aid, smmdp.h)
namespace {
struct DpTiming {
    std::mutex mu;
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> rec, pool;
    std::vector<int> tag;          // per record: 0 the only DP launch of its call, 1 the critical videos of a split decode
                                   // (caller's stream), 2 the rest of a split decode (second stream)
} g_dp_timing;

bool dp_timing_begin(hipStream_t stream, std::pair<hipEvent_t, hipEvent_t> &ev)
{
    std::lock_guard<std::mutex> lock(g_dp_timing.mu);
    if (!g_dp_timing.on) return false;
    if (!g_dp_timing.pool.empty()) { ev = g_dp_timing.pool.back(); g_dp_timing.pool.pop_back(); }
    else if (hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess) return false;
    return hipEventRecord(ev.first, stream) == hipSuccess;
}

void dp_timing_end(hipStream_t stream, const std::pair<hipEvent_t, hipEvent_t> &ev, int tag)
{
    (void)hipEventRecord(ev.second, stream);
    std::lock_guard<std::mutex> lock(g_dp_timing.mu);
    g_dp_timing.rec.push_back(ev);
    g_dp_timing.tag.push_back(tag);
}
}  // namespace

extern "C" void smm_dp_timing_enable(int on)
{
    std::lock_guard<std::mutex> lock(g_dp_timing.mu);
    g_dp_timing.on = on != 0;
}

extern "C" int smm_dp_timing_read_tagged(float *ms, int32_t *tags, int cap)
{
    std::vector<std::pair<hipEvent_t, hipEvent_t>> rec;
    std::vector<int> tag;
    {
        std::lock_guard<std::mutex> lock(g_dp_timing.mu);
        rec.swap(g_dp_timing.rec);
        tag.swap(g_dp_timing.tag);
    }
    int n = 0;
    for (auto &ev : rec) {
        float t = 0.f;
        if (hipEventSynchronize(ev.second) == hipSuccess && hipEventElapsedTime(&t, ev.first, ev.second) == hipSuccess && n < cap) {
            if (ms) ms[n] = t;
            if (tags) tags[n] = tag[n];
        }
        ++n;
    }
    std::lock_guard<std::mutex> lock(g_dp_timing.mu);
    for (auto &ev : rec) g_dp_timing.pool.push_back(ev);
    return n;
}

extern "C" int smm_dp_timing_read(float *ms, int cap) { return smm_dp_timing_read_tagged(ms, nullptr, cap); }

namespace {
// (defined with the split decode below) per device: second streams and pooled event pairs
hipStream_t aux_stream_n(int dev, int which);
bool aux_events_get(int dev, std::pair<hipEvent_t, hipEvent_t> &ev);
void aux_events_put(int dev, const std::pair<hipEvent_t, hipEvent_t> &ev);
}  // namespace

static int run_viterbi(const smm_shape *s, const Staged &st, const double *elp, const double *trans, const double *init,
                       const double *len_scores, const double *endpen, const int64_t *class_map, int64_t *spans,
                       int64_t *labels, double *best, int32_t *n_segs, hipStream_t stream, int first = 0, int count = -1,
                       bool prep = true, bool launch = true, int timing_tag = 0)
{
    if (!elp || !trans || !init || !len_scores) return SMM_ERR_ARG;
    SmmDpArgs a{};
    a.videos = st.videos; a.order = st.order; a.n_states = st.n_states;
    const bool no_eos = (s->flags & SMM_SHAPE_NO_EOS) != 0;
    a.elp = elp; a.trans = trans; a.init = init; a.len = len_scores; a.endpen = no_eos ? nullptr : endpen; a.class_map = class_map;
    a.hist = st.hist; a.spans = spans; a.labels = labels; a.best = best; a.n_segs = n_segs; a.err = st.err;
    a.c_max = s->c_max; a.k_rows = s->k_rows; a.t_max = s->t_max; a.b = s->b;
    if (count >= 0) { a.order = st.order + first; a.b = count; }
    // time-split videos (smm_chunk.hip): the launch runs over UNITS -- [the split videos' units | the unsplit videos]; a stream
    // split's first part is exactly the units
    const bool with_units = st.n_cv > 0 && (count < 0 || first == 0);
    if (st.n_cv > 0) {
        a.videos = st.uvideos;
        a.b_videos = s->b;
        a.chunk_anchor = st.anchors;
        if (count < 0) { a.order = st.uorder; a.b = st.n_units; }
        else if (first == 0) { a.order = st.uorder; a.b = st.u_part1; }
        else { a.order = st.uorder + st.u_part1; a.b = st.n_units - st.u_part1; }
    }
    a.flags = 0;
#ifdef SMM_DEV
    a.flags = env().debug_flags;                              // (SmmDpArgs::flags bit 0: profiling, outputs undefined)
#endif
    if (no_eos) a.flags |= 8;
    if (!env().spec) a.flags |= 256;                          // A/B aid: no speculative transition
    if (st.band_mode) {
        double *len_t = st.band, *band_tab = st.band + (size_t)s->n_groups * s->c_max * SMM_BAND_ROW;
        double *dmin_t = band_tab + (size_t)s->n_groups * s->c_max * SMM_BAND_TAB;
        if (prep) smm_launch_band_tables(len_scores, st.n_states, len_t, band_tab, dmin_t, s->n_groups, s->c_max, s->k_rows, stream);
        a.len_t = len_t;
        a.band_tab = band_tab;
        a.dmin_t = dmin_t;
        a.flags |= 128;
    }
    if (ring_regs(st.kp_max) == 1 && !env().no_bt_window) {
        // short segments: window back-trace (smm_viterbi.hip); W >= 2 kp keeps a window good for many segments
        // about 48 KB of window (three workgroups per CU stay possible), at least 2 kp positions, at most 512
        int w = (int)(48 * 1024 / (24 * (size_t)st.c_need)) & ~7;
        w = std::min(512, std::max(w, (2 * st.kp_max + 7) & ~7));
        const size_t bytes = sizeof(double) * ((size_t)3 * w + st.c_need + st.kp_max) * st.c_need;
        if (bytes <= 126 * 1024) { a.bt_window = w; a.bt_dyn_bytes = (int32_t)bytes; }
    }
    if (!launch) { SMM_HIP(hipGetLastError()); return SMM_OK; }      // (prep only)
    if (with_units) smm_launch_cum_anchors(a, st.cvs, st.n_cv, st.anchors, stream);   // cumE at the units' first positions, serially
    // SMALL workgroups (smm_viterbi.hip: smm_launch_viterbi_small): when this range of the launch order ends with the part's
    // <= 16-state videos (stage_uncached put them there), those run in four-wave workgroups, two per CU, on a side stream
    // BESIDE the eight-wave launch of the others -- forked behind everything this stream has queued so far (band tables,
    // this part's emission) and joined before the call goes on.  Not under stream capture (the side stream is shared).
    const int lo = (int)(a.order - (st.n_cv > 0 ? st.uorder : st.order));
    int n_small = (st.small_count > 0 && lo <= st.small_first && lo + a.b == st.small_first + st.small_count) ? st.small_count : 0;
    hipStream_t side = nullptr;
    std::pair<hipEvent_t, hipEvent_t> fj{nullptr, nullptr};
    int dev = 0;
    if (n_small > 0) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && hipStreamIsCapturing(stream, &cap) == hipSuccess &&
            cap == hipStreamCaptureStatusNone)
            side = aux_stream_n(dev, 1);
        if (side && !aux_events_get(dev, fj)) side = nullptr;
        if (!side) n_small = 0;
    }
    int rc_small = SMM_OK;
    if (n_small > 0) {
        SmmDpArgs as = a;
        as.order = a.order + (a.b - n_small);
        as.b = n_small;
        a.b -= n_small;
        if (hipEventRecord(fj.first, stream) != hipSuccess || hipStreamWaitEvent(side, fj.first, 0) != hipSuccess) rc_small = SMM_ERR_HIP;
        if (rc_small == SMM_OK) {
            std::pair<hipEvent_t, hipEvent_t> tev2;
            const bool timed2 = dp_timing_begin(side, tev2);
            rc_small = smm_launch_viterbi_small(as, side);
            if (timed2) dp_timing_end(side, tev2, 3);
        }
    }
    std::pair<hipEvent_t, hipEvent_t> tev;
    int rc = SMM_OK;
    if (a.b > 0) {
        const bool timed = dp_timing_begin(stream, tev);
        rc = smm_launch_viterbi(a, st.band_mode ? 16 : ring_regs(st.kp_max), st.c_need, stream);
        if (timed) dp_timing_end(stream, tev, timing_tag);
    }
    if (side) {
        // the join is made even after an error on the way, so that this stream never runs ahead of the side stream
        if (hipEventRecord(fj.second, side) != hipSuccess || hipStreamWaitEvent(stream, fj.second, 0) != hipSuccess) rc_small = SMM_ERR_HIP;
        aux_events_put(dev, fj);
    }
    if (rc != SMM_OK) return rc;
    if (rc_small != SMM_OK) return rc_small;
    if (with_units) {
        // certify the cuts, walk the path, write the split videos' outputs -- and decode again, in one piece, the ones that
        // could not be certified (one workgroup per split video; all but the flagged ones return at once)
        smm_launch_chunk_stitch(a, st.cvs, st.n_cv, st.redo, stream);
        SmmDpArgs ar = a;
        ar.order = st.porder;
        ar.b = st.n_cv;
        ar.redo = st.redo;
        // (BAND mode: the launch's own kernel under its second name, so that per-kernel statistics of the DP launch are not
        // diluted by a launch that nearly always returns at once; the ring kernels repair with themselves)
        const int rc2 = st.band_mode ? smm_launch_viterbi_repair(ar, st.c_need, stream)
                                     : smm_launch_viterbi(ar, ring_regs(st.kp_max), st.c_need, stream);
        if (rc2 != SMM_OK) return rc2;
    }
    SMM_HIP(hipGetLastError());
    return SMM_OK;
}

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" int smm_emission_f64(const smm_shape *shape, const int64_t *lengths_host, const int64_t *frame_offset_host,
                                const int32_t *group_host, const int32_t *n_states_host, const float *x, const double *w,
                                const double *cst, const double *inv_var, const float *cons, double *elp64, float *elp32,
                                void *workspace, size_t workspace_bytes, void *stream)
{
    Staged st;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    int rc = stage(shape, lengths_host, frame_offset_host, group_host, nullptr, n_states_host, workspace, workspace_bytes,
                   hs, &st);
    if (rc != SMM_OK) return rc;
    return run_emission(shape, st, x, w, cst, inv_var, cons, elp64, elp32, hs);
}

extern "C" int smm_emission_bwd_f64(const smm_shape *shape, const int64_t *lengths_host, const int64_t *frame_offset_host,
                                    const int32_t *group_host, const int32_t *n_states_host, const float *x,
                                    const double *g_elp, double *g_w, double *g_cst, double *g_inv_var,
                                    void *workspace, size_t workspace_bytes, void *stream)
{
    Staged st;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    if (!x || !g_elp || !g_w || !g_cst || !g_inv_var || !shape || shape->d < 1) return SMM_ERR_ARG;
    int rc = stage(shape, lengths_host, frame_offset_host, group_host, nullptr, n_states_host, workspace, workspace_bytes,
                   hs, &st, false, smm_emission_bwd_chunk());
    if (rc != SMM_OK) return rc;
    const size_t g = shape->n_groups, cm = shape->c_max, d = shape->d;
    {
        void *const zp[3] = {g_w, g_cst, g_inv_var};
        const size_t zb[3] = {sizeof(double) * g * cm * d, sizeof(double) * g * cm, sizeof(double) * d};
        SMM_HIP((hipError_t)smm_zero_multi_async(zp, zb, 3, hs));
    }
    SmmEmBwdArgs a{st.videos, st.order, st.n_states, st.em_cum, x, g_elp, g_w, g_cst, g_inv_var,
                   shape->d, shape->c_max, shape->b, st.em_blocks};
    smm_launch_emission_bwd(a, st.c_need, hs);
    SMM_HIP(hipGetLastError());
    return SMM_OK;
}

extern "C" int smm_viterbi_f64(const smm_shape *shape, const int64_t *lengths_host, const int64_t *frame_offset_host,
                               const int32_t *group_host, const int32_t *kp_host, const int32_t *n_states_host,
                               const double *elp, const double *trans, const double *init, const double *len_scores,
                               const double *endpen, const int64_t *class_map, int64_t *spans, int64_t *labels,
                               double *best, int32_t *n_segs, void *workspace, size_t workspace_bytes, void *stream)
{
    Staged st;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    int rc = stage(shape, lengths_host, frame_offset_host, group_host, kp_host, n_states_host, workspace, workspace_bytes,
                   hs, &st, true);
    if (rc != SMM_OK) return rc;
    return run_viterbi(shape, st, elp, trans, init, len_scores, endpen, class_map, spans, labels, best, n_segs, hs);
}

extern "C" int smm_viterbi_f32(const smm_shape *shape, const int64_t *lengths_host, const int64_t *frame_offset_host,
                               const int32_t *group_host, const int32_t *kp_host, const int32_t *n_states_host,
                               const float *elp, const float *trans, const float *init, const float *len_scores,
                               const float *endpen, const int64_t *class_map, int64_t *spans, int64_t *labels,
                               double *best, int32_t *n_segs, void *workspace, size_t workspace_bytes, void *stream)
{
    Staged st;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    int rc = stage(shape, lengths_host, frame_offset_host, group_host, kp_host, n_states_host, workspace, workspace_bytes,
                   hs, &st, true);
    if (rc != SMM_OK) return rc;
    if (!elp || !trans || !init || !len_scores) return SMM_ERR_ARG;
    const size_t g = shape->n_groups, cm = shape->c_max;
    double *t64 = st.tabs, *i64 = t64 + g * cm * cm, *l64 = i64 + g * cm, *e64 = l64 + g * shape->k_rows * cm;
    smm_launch_widen(trans, t64, g * cm * cm, hs);
    smm_launch_widen(init, i64, g * cm, hs);
    smm_launch_widen(len_scores, l64, g * shape->k_rows * cm, hs);
    if (endpen) smm_launch_widen(endpen, e64, (size_t)shape->b * cm, hs);
    smm_launch_widen(elp, st.elp, (size_t)shape->total_frames * cm, hs);
    SMM_HIP(hipGetLastError());
    return run_viterbi(shape, st, st.elp, t64, i64, l64, endpen ? e64 : nullptr, class_map, spans, labels, best, n_segs, hs);
}

// The second stream of a split decode: one per device, created on first use with the lowest priority (the critical videos'
// DP on the caller's stream goes first); it lives until smm_release_cached_plans().  Every use is bracketed by a pooled
// event pair on the caller's stream, so from the caller's point of view the call is still ordered on ITS stream.  A call
// made under stream capture does NOT split (the one shared second stream would be pulled into the capture and a
// concurrent call from another thread would then enqueue onto a capturing stream): it runs as one launch pair.
namespace {
struct AuxDev {                       // per device: the second stream of a split decode, the side stream of the small workgroups,
    hipStream_t stream = nullptr, stream1 = nullptr;             // and the event pairs that bracket them
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;     // idle (fork, join) pairs
};
std::mutex g_aux_mu;
AuxDev g_aux[64];

hipStream_t aux_stream_n(int dev, int which)
{
    std::lock_guard<std::mutex> lock(g_aux_mu);
    AuxDev &d = g_aux[dev];
    hipStream_t &st = which ? d.stream1 : d.stream;
    if (!st) {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, least) != hipSuccess) st = nullptr;
    }
    return st;
}
hipStream_t aux_stream(int dev) { return aux_stream_n(dev, 0); }

bool aux_events_get(int dev, std::pair<hipEvent_t, hipEvent_t> &ev)
{
    {
        std::lock_guard<std::mutex> lock(g_aux_mu);
        AuxDev &d = g_aux[dev];
        if (!d.events.empty()) { ev = d.events.back(); d.events.pop_back(); return true; }
    }
    ev = {nullptr, nullptr};
    if (hipEventCreateWithFlags(&ev.first, hipEventDisableTiming) != hipSuccess) return false;
    if (hipEventCreateWithFlags(&ev.second, hipEventDisableTiming) != hipSuccess) { (void)hipEventDestroy(ev.first); return false; }
    return true;
}

void aux_events_put(int dev, const std::pair<hipEvent_t, hipEvent_t> &ev)
{
    // (an event that is re-recorded while an earlier wait on it is still queued keeps that wait's meaning: a wait
    // captures the record it follows)
    std::lock_guard<std::mutex> lock(g_aux_mu);
    g_aux[dev].events.push_back(ev);
}
}  // namespace

static void release_aux()
{
    std::lock_guard<std::mutex> lock(g_aux_mu);
    for (int d = 0; d < 64; ++d) {
        for (auto &ev : g_aux[d].events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
        g_aux[d].events.clear();
        if (g_aux[d].stream) { (void)hipStreamDestroy(g_aux[d].stream); g_aux[d].stream = nullptr; }
        if (g_aux[d].stream1) { (void)hipStreamDestroy(g_aux[d].stream1); g_aux[d].stream1 = nullptr; }
    }
    std::lock_guard<std::mutex> lock2(g_dp_timing.mu);
    for (auto &ev : g_dp_timing.pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    g_dp_timing.pool.clear();
}

extern "C" int smm_decode_f32(const smm_shape *shape, const int64_t *lengths_host, const int64_t *frame_offset_host,
                              const int32_t *group_host, const int32_t *kp_host, const int32_t *n_states_host,
                              const float *x, const double *w, const double *cst, const double *inv_var, const float *cons,
                              const double *trans, const double *init, const double *len_scores, const double *endpen,
                              const int64_t *class_map, int64_t *spans, int64_t *labels, double *best, int32_t *n_segs,
                              float *elp32, void *workspace, size_t workspace_bytes, void *stream)
{
    Staged st;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    int rc = stage(shape, lengths_host, frame_offset_host, group_host, kp_host, n_states_host, workspace, workspace_bytes,
                   hs, &st, true, 0, true);
    if (rc != SMM_OK) return rc;
    int dev = 0;
    hipStream_t aux = nullptr;
    if (st.n_split > 0 && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(hs, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) aux = aux_stream(dev);
    }
    std::pair<hipEvent_t, hipEvent_t> fj{nullptr, nullptr};
    if (aux && !aux_events_get(dev, fj)) aux = nullptr;
    if (!aux) {
        rc = run_emission(shape, st, x, w, cst, inv_var, cons, st.elp, elp32, hs);
        if (rc != SMM_OK) return rc;
        return run_viterbi(shape, st, st.elp, trans, init, len_scores, endpen, class_map, spans, labels, best, n_segs, hs);
    }
    // split decode (choose_split): [band tables, emission of the critical videos] on the caller's stream, then their DP
    // there, while the second stream scores and decodes the rest; the caller's stream waits for it at the end
    const int n1 = st.n_split, n2 = shape->b - st.n_split;
    rc = run_viterbi(shape, st, st.elp, trans, init, len_scores, endpen, class_map, spans, labels, best, n_segs, hs, 0, n1, true, false);
    if (rc != SMM_OK) { aux_events_put(dev, fj); return rc; }
    const hipEvent_t fork = fj.first, join = fj.second;
    int rc2 = SMM_OK;
    rc = run_emission(shape, st, x, w, cst, inv_var, cons, st.elp, elp32, hs, 0, n1);
    if (rc != SMM_OK) { aux_events_put(dev, fj); return rc; }
    // (the fork sits BEHIND the critical videos' emission on purpose: forked in front of it, the rest's emission workgroups fill
    // the CUs the critical videos' DP workgroups need -- one per CU, 245 VGPRs -- and that launch starts late: cfg3 2.81 -> 2.89 ms
    // per step, critical launch 2.31 -> 2.65 ms, three alternating runs on one box, round 5)
    if (hipEventRecord(fork, hs) != hipSuccess || hipStreamWaitEvent(aux, fork, 0) != hipSuccess) rc2 = SMM_ERR_HIP;
#ifdef SMM_DEV
    const int dbg_split = env().split_debug;                  // (timing experiments: results INCOMPLETE)
#else
    const int dbg_split = 0;
#endif
    if (rc2 == SMM_OK && !(dbg_split & 2)) rc2 = run_emission(shape, st, x, w, cst, inv_var, cons, st.elp, elp32, aux, n1, n2);
    if (rc2 == SMM_OK && !(dbg_split & 1))
        rc2 = run_viterbi(shape, st, st.elp, trans, init, len_scores, endpen, class_map, spans, labels, best, n_segs, aux, n1, n2, false, true, 2);
    rc = run_viterbi(shape, st, st.elp, trans, init, len_scores, endpen, class_map, spans, labels, best, n_segs, hs, 0, n1, false, true, 1);
    // the join is made even after an error on the way, so that the caller's stream never runs ahead of the second one
    if (hipEventRecord(join, aux) != hipSuccess || hipStreamWaitEvent(hs, join, 0) != hipSuccess) rc2 = SMM_ERR_HIP;
    aux_events_put(dev, fj);
    return rc != SMM_OK ? rc : rc2;
}

extern "C" int smm_logz_f64(const smm_shape *shape, const int64_t *lengths_host, const int64_t *frame_offset_host,
                            const int32_t *group_host, const int32_t *kp_host, const int32_t *n_states_host,
                            const double *elp, const double *trans, const double *init, const double *len_scores,
                            const double *endpen, double *logz, void *workspace, size_t workspace_bytes, void *stream)
{
    Staged st;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    int rc = stage(shape, lengths_host, frame_offset_host, group_host, kp_host, n_states_host, workspace, workspace_bytes,
                   hs, &st);
    if (rc != SMM_OK) return rc;
    if (!elp || !trans || !init || !len_scores || !logz) return SMM_ERR_ARG;
    SmmDpArgs a{};
    a.videos = st.videos; a.order = st.order; a.n_states = st.n_states;
    a.elp = elp; a.trans = trans; a.init = init; a.len = len_scores; a.endpen = endpen;
    a.hist = st.hist; a.err = st.err;
    a.c_max = shape->c_max; a.k_rows = shape->k_rows; a.t_max = shape->t_max; a.b = shape->b;
    if (shape->flags & SMM_SHAPE_NO_EOS) { a.flags |= 8; a.endpen = nullptr; }
    if (shape->flags & SMM_SHAPE_LOGZ_BOTH) {
        // forward and time-reversed recursion in one launch: the transposed tables and the reversed runs' closing values
        // live where smm_logz_bwd_f64 keeps them
        const size_t g = shape->n_groups, cm = shape->c_max;
        double *trans_t = st.tabs;
        smm_launch_transpose(trans, trans_t, (int)g, (int)cm, hs);
        a.trans_t = trans_t;
        a.logz_b = trans_t + g * cm * cm;
        a.flags |= 64;
    }
    rc = smm_launch_logz(a, logz, ring_regs(st.kp_max), st.c_need, hs);
    if (rc != SMM_OK) return rc;
    SMM_HIP(hipGetLastError());
    return SMM_OK;
}

extern "C" int smm_logz_bwd_f64(const smm_shape *shape, const int64_t *lengths_host, const int64_t *frame_offset_host,
                                const int32_t *group_host, const int32_t *kp_host, const int32_t *n_states_host,
                                const double *elp, const double *trans, const double *init, const double *len_scores,
                                const double *endpen, const double *logz, const double *grad_logz, double *g_elp,
                                double *g_trans, double *g_init, double *g_len, void *workspace, size_t workspace_bytes,
                                void *stream)
{
    Staged st;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    int rc = stage(shape, lengths_host, frame_offset_host, group_host, kp_host, n_states_host, workspace, workspace_bytes,
                   hs, &st);
    if (rc != SMM_OK) return rc;
    if (!elp || !trans || !init || !len_scores || !logz || !g_elp || !g_trans || !g_init || !g_len) return SMM_ERR_ARG;
    const size_t g = shape->n_groups, cm = shape->c_max;
    double *trans_t = st.tabs;                 // [g][cm][cm] transposed
    double *logz_b = trans_t + g * cm * cm;    // [b] log Z as closed by the backward recursion (consistency value)
    const bool no_eos = (shape->flags & SMM_SHAPE_NO_EOS) != 0;
    if (!(shape->flags & SMM_SHAPE_LOGZ_BOTH)) {   // (else: smm_logz_f64 already ran the reversed recursion)
        smm_launch_transpose(trans, trans_t, (int)g, (int)cm, hs);
        SmmDpArgs a{};
        a.videos = st.videos; a.order = st.order; a.n_states = st.n_states;
        a.elp = elp; a.trans = trans_t; a.init = init; a.len = len_scores; a.endpen = endpen;
        a.hist = st.hist; a.err = st.err;
        a.c_max = shape->c_max; a.k_rows = shape->k_rows; a.t_max = shape->t_max; a.b = shape->b;
        a.flags = 2;                               // time-reversed run -> backward messages in the second history half
        if (no_eos) { a.flags |= 8; a.endpen = nullptr; }
        rc = smm_launch_logz(a, logz_b, ring_regs(st.kp_max), st.c_need, hs);
        if (rc != SMM_OK) return rc;
    }
    {
        void *const zp[4] = {g_trans, g_init, g_len, g_elp};
        const size_t zb[4] = {sizeof(double) * g * cm * cm, sizeof(double) * g * cm, sizeof(double) * g * shape->k_rows * cm,
                              sizeof(double) * (size_t)shape->total_frames * cm};
        SMM_HIP((hipError_t)smm_zero_multi_async(zp, zb, 4, hs));
    }
    SmmBwdArgs m{st.videos, st.n_states, trans, len_scores, st.hist, logz, grad_logz, g_elp, g_trans, g_init, g_len,
                 shape->c_max, shape->k_rows, shape->b, elp, no_eos ? 1 : 0};
    smm_launch_marginals(m, shape->t_max, st.kp_max, hs);
    SMM_HIP(hipGetLastError());
    return SMM_OK;
}

// ------------------------------------------------------------------------------------------------ dense boundary
static size_t dense_off(size_t &cur, size_t bytes)
{
    const size_t o = cur;
    cur += align_up(bytes, 256);
    return o;
}

extern "C" size_t smm_dense_workspace_bytes(int32_t b, int32_t n1, int32_t k, int32_t c)
{
    if (b < 1 || n1 < 1 || k < 1 || c < 1) return 0;
    size_t cur = 0;
    dense_off(cur, sizeof(int64_t) * b);
    dense_off(cur, sizeof(double) * (size_t)b * k * k * c);
    dense_off(cur, sizeof(double) * (size_t)b * (n1 + 1) * c);
    dense_off(cur, (size_t)b * n1 * k * c);
    dense_off(cur, sizeof(uint16_t) * (size_t)b * (n1 + 1) * c);
    dense_off(cur, sizeof(double) * (size_t)b * (n1 + 1) * c);      // backward messages (smm_dense_marginals_f32)
    return cur;
}

extern "C" int smm_dense_dp_f32(const float *scores, const int64_t *lengths_host, int32_t b, int32_t n1, int32_t k,
                                int32_t c, int32_t semiring, double *v, int64_t *spans, void *workspace,
                                size_t workspace_bytes, void *stream)
{
    if (!scores || !lengths_host || !v || !workspace || b < 1 || n1 < 1 || k < 1 || c < 1) return SMM_ERR_ARG;
    if (c > 255 || k > 65535) return SMM_ERR_UNSUPPORTED;
    if (workspace_bytes < smm_dense_workspace_bytes(b, n1, k, c)) return SMM_ERR_WORKSPACE;
    for (int i = 0; i < b; ++i)
        if (lengths_host[i] < 1 || lengths_host[i] > n1 + 1) return SMM_ERR_ARG;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    char *base = static_cast<char *>(workspace);
    size_t cur = 0;
    SmmDenseArgs a{};
    int64_t *dlen = reinterpret_cast<int64_t *>(base + dense_off(cur, sizeof(int64_t) * b));
    a.alpha = reinterpret_cast<double *>(base + dense_off(cur, sizeof(double) * (size_t)b * k * k * c));
    a.beta = reinterpret_cast<double *>(base + dense_off(cur, sizeof(double) * (size_t)b * (n1 + 1) * c));
    a.bp_from = reinterpret_cast<uint8_t *>(base + dense_off(cur, (size_t)b * n1 * k * c));
    a.bp_k = reinterpret_cast<uint16_t *>(base + dense_off(cur, sizeof(uint16_t) * (size_t)b * (n1 + 1) * c));
    SMM_HIP((hipError_t)smm_upload_meta(dlen, lengths_host, sizeof(int64_t) * b, hs));
    a.edge = scores; a.lengths = dlen; a.v = v; a.spans = spans;
    a.b = b; a.n1 = n1; a.k = k; a.c = c;
    smm_launch_dense(a, semiring != 0, hs);
    SMM_HIP(hipGetLastError());
    return SMM_OK;
}

extern "C" int smm_dense_marginals_f32(const float *scores, const int64_t *lengths_host, int32_t b, int32_t n1, int32_t k,
                                       int32_t c, const double *v, const double *grad_v, float *marginals, void *workspace,
                                       size_t workspace_bytes, void *stream)
{
    if (!scores || !lengths_host || !v || !marginals || !workspace || b < 1 || n1 < 1 || k < 1 || c < 1) return SMM_ERR_ARG;
    if (c > 255 || k > 65535) return SMM_ERR_UNSUPPORTED;
    if (workspace_bytes < smm_dense_workspace_bytes(b, n1, k, c)) return SMM_ERR_WORKSPACE;
    for (int i = 0; i < b; ++i)
        if (lengths_host[i] < 1 || lengths_host[i] > n1 + 1) return SMM_ERR_ARG;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    char *base = static_cast<char *>(workspace);
    size_t cur = 0;
    SmmDenseArgs a{};
    int64_t *dlen = reinterpret_cast<int64_t *>(base + dense_off(cur, sizeof(int64_t) * b));
    a.alpha = reinterpret_cast<double *>(base + dense_off(cur, sizeof(double) * (size_t)b * k * k * c));
    a.beta = reinterpret_cast<double *>(base + dense_off(cur, sizeof(double) * (size_t)b * (n1 + 1) * c));
    dense_off(cur, (size_t)b * n1 * k * c);
    dense_off(cur, sizeof(uint16_t) * (size_t)b * (n1 + 1) * c);
    double *rmsg = reinterpret_cast<double *>(base + dense_off(cur, sizeof(double) * (size_t)b * (n1 + 1) * c));
    a.edge = scores; a.lengths = dlen; a.v = const_cast<double *>(v);
    a.b = b; a.n1 = n1; a.k = k; a.c = c;
    smm_launch_dense_marginals(a, rmsg, grad_v, marginals, hs);     // (dlen, beta: left by smm_dense_dp_f32, semiring 1)
    SMM_HIP(hipGetLastError());
    return SMM_OK;
}
