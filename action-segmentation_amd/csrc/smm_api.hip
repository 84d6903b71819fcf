// smm_api.hip -- host side of libsmmdp.so: argument checks, launch planning, C ABI (include/smmdp.h).
#include <algorithm>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>
#include <unordered_map>
#include <mutex>
#include <utility>

#include "../../include/smmdp.h"
#include "smm_device.h"
#include "smm_launch.h"

static thread_local int g_last_hip = 0;

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
    if (std::getenv("SMM_UPLOAD_MEMCPY")) return (int)hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, stream);   // (debug)
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

#define SMM_MAX_PAIRS 256

struct SmmPlan {
    size_t meta_bytes;     // SmmVideo[b] | order[b] | n_states[g] | emission block table[b+1] | err | gang counters
    size_t o_order, o_nstates, o_emcum, o_err, o_pflags;
    size_t hist_doubles;   // sum over videos of 8*c_max*(T+1): forward cumE/h/gamma, backward cumE/h/gamma, 2 transposes
    size_t elp_doubles;    // total_frames*c_max  (smm_decode_f32 / smm_viterbi_f32)
    size_t tab_doubles;    // widened tables       (smm_viterbi_f32)
    size_t band_doubles;   // Viterbi BAND mode: state-major length table + skip-test bounds per (group, state)
    size_t total;
};

static bool shape_ok(const smm_shape *s)
{
    return s && s->b > 0 && s->n_groups > 0 && s->c_max > 0 && s->k_rows >= 2 && s->t_max > 0 && s->total_frames > 0;
}

static SmmPlan make_plan(const smm_shape *s, const int64_t *lengths)
{
    SmmPlan p{};
    p.o_order = align_up(sizeof(SmmVideo) * s->b, 256);
    p.o_nstates = p.o_order + align_up(sizeof(int32_t) * s->b, 256);
    p.o_emcum = p.o_nstates + align_up(sizeof(int32_t) * s->n_groups, 256);
    p.o_err = p.o_emcum + align_up(sizeof(int32_t) * ((size_t)s->b + 1), 256);
    p.o_pflags = p.o_err + 512;     // error word + diagnostic counters, then 4 counters per leader / follower gang
    p.meta_bytes = p.o_pflags + align_up(sizeof(int32_t) * 4 * (size_t)std::min(s->b, SMM_MAX_PAIRS), 256);
    size_t h = 0;
    for (int i = 0; i < s->b; ++i) h += 8 * (size_t)s->c_max * (size_t)(lengths[i] + 1);
    p.hist_doubles = h;
    p.elp_doubles = (size_t)s->total_frames * s->c_max;
    p.tab_doubles = (size_t)s->n_groups * s->c_max * ((size_t)s->c_max + 1 + s->k_rows) + (size_t)s->b * s->c_max;
    p.band_doubles = (size_t)s->n_groups * s->c_max * ((size_t)SMM_BAND_ROW + SMM_BAND_TAB);
    p.total = p.meta_bytes + 8 * (p.hist_doubles + p.elp_doubles + p.tab_doubles + p.band_doubles) + 1024;
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
    bool band_mode;        // Viterbi: BAND mode (one workgroup per video, exact band skipping) instead of 1024-slot rings / gangs
    int32_t *pair_flags;
    int32_t *em_cum;       // emission: workgroups before each video of `order` ([b + 1])
    std::vector<int32_t> em_cum_host;   // (decode split: the same table on the host)
    int em_tpw, em_blocks;
    int kp_max, c_need;
    int n_pairs;           // Viterbi only: the first n_pairs videos of `order` may run on two CUs each
    bool pairs_cover_big;  // every video with more than 21 states is among them
    int n_split;           // decode only: the first n_split videos of `order` are the launch's critical path (0: no split)
};

// Gangs of two or three CUs for the most expensive videos (smm_viterbi.hip, PAIR mode).  The DP kernel's time is the
// time of its longest videos while other CUs idle; choose_pairs() below picks which videos ride in gangs by simulating a
// longest-first list schedule with a measured cost model (ns per frame, see there).  frame_ns_single: one CU.
static double frame_ns_single(int c)
{
    int nv[7];
    for (int r = 0; r < 7; ++r) nv[r] = c > r ? (c - r + 6) / 7 : 0;
    const int sw = smm_rebalanced_rank(c);                  // (smm_device.h: the kernel's wave <-> rank mapping)
    if (sw >= 0) std::swap(nv[sw], nv[6]);
    // SIMDs: ranks (0,4), (1,5), (2,3) and chain wave (+ its partner's mover duty: ~2.5 states) + rank 6
    const double load = std::max(std::max<double>(std::max(nv[0] + nv[4], nv[1] + nv[5]), nv[2] + nv[3]), 2.5 + nv[6]);
    return std::max(310.0, 73.0 * load) + 15.0;
}

// Reorders `order` (most work first on entry) into [gang videos | single videos], both most work first, sets
// SmmVideo::nfol of the gang videos (1: leader + follower on two CUs, 2: leader + two followers on three) and returns
// the number of gangs.  Videos with more than 21 states MUST ride in a gang (a single 8-wave workgroup holds 21
// rings): *big_ok says whether all of them do (else the caller falls back to the 12-wave configuration, without
// gangs).  Optional gangs: the n most expensive eligible videos, the first n3 of them (above 16 states) as triples;
// (n, n3) chosen by simulating a list schedule in grid order.  Cost model, ns per frame including the back-trace
// (measured at K = 1024, T = 4096, 64 videos at a time): one CU: the most loaded SIMD's states x 73, at least 310,
// + 15; pair: 252 (<= 15 states), 300 (16), 330 (17..23); triple: 235 (round 2: per-video finish times on cfg3 seed 2).
static int choose_pairs(SmmVideo *hv, int32_t *order, const int32_t *n_states, int b, int kp_max, int c_need, bool *big_ok)
{
    *big_ok = false;
    int forced = -1, forced3 = -1;
    if (const char *e = std::getenv("SMM_PAIRS")) forced = std::atoi(e);
    if (const char *e = std::getenv("SMM_TRIPLES")) forced3 = std::atoi(e);
    // gangs exist for 1024-slot rings, 8 waves; a leader's short rings hold every state (24..32: always a triple, whose
    // two followers split the long rings, 16 each at most)
    if (kp_max <= 512 || c_need > 32 || std::getenv("SMM_NW")) return 0;
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return 0;
    std::vector<int32_t> must, opt, rest;
    for (int i = 0; i < b; ++i) {
        const SmmVideo &v = hv[order[i]];
        const int c = n_states[v.group];
        if (c > 21) must.push_back(order[i]);
        else if ((forced > 0 || (v.T >= 1024 && v.kp >= 256 && c >= 4)) && opt.size() == (size_t)(i - (int)must.size()))
            opt.push_back(order[i]);
        else rest.push_back(order[i]);                                      // (optional gangs: a prefix of the <= 21-state order)
    }
    if ((int)must.size() > SMM_MAX_PAIRS) return 0;
    if (c_need > 21 && forced == 0) return 0;                               // gangs switched off: 12-wave configuration
    const int cap = std::max(0, std::min(SMM_MAX_PAIRS, n_cu / 2) - (int)must.size());
    const int eligible = std::min((int)opt.size(), cap);
    auto states = [&](int32_t v) { return n_states[hv[v].group]; };
    auto gang_ns = [&](int32_t v, int nfol) {
        const int c = states(v);
        return hv[v].T * (nfol == 2 ? 235.0 : (c > 16 ? 330.0 : (c > 15 ? 300.0 : 252.0)));   // (measured, round 2: profiles/round2_gang_probe.txt)
    };
    auto single_ns = [&](int32_t v) { return hv[v].T * frame_ns_single(states(v)); };
    // the gang list for (n optional gangs, n3 triples): must + opt[0..n), most expensive first; the first n3 of its
    // videos above 16 states become triples
    std::vector<int32_t> gangs;
    std::vector<int> nf;
    auto build = [&](int n, int n3) {
        gangs.assign(must.begin(), must.end());
        gangs.insert(gangs.end(), opt.begin(), opt.begin() + n);
        std::stable_sort(gangs.begin(), gangs.end(), [&](int32_t x, int32_t y) { return gang_ns(x, 1) > gang_ns(y, 1); });
        nf.assign(gangs.size(), 1);
        for (size_t i = 0; i < gangs.size(); ++i)
            if (states(gangs[i]) > 23) nf[i] = 2;                             // (one follower holds 16 long rings)
        for (size_t i = 0; i < gangs.size() && n3 > 0; ++i)
            if (states(gangs[i]) > 16 && nf[i] == 1) { nf[i] = 2; --n3; }
    };
    std::vector<double> cu(n_cu);
    auto simulate = [&](int n) {                                             // list schedule in grid order
        std::fill(cu.begin(), cu.end(), 0.0);
        std::make_heap(cu.begin(), cu.end(), std::greater<double>());
        for (size_t i = 0; i < gangs.size(); ++i) {                           // the 2 or 3 earliest-free CUs
            const int m = 1 + nf[i];
            double t0 = 0.0;
            for (int q = 0; q < m; ++q) {
                std::pop_heap(cu.begin(), cu.end(), std::greater<double>());
                t0 = std::max(t0, cu.back());
                cu.pop_back();
            }
            for (int q = 0; q < m; ++q) {
                cu.push_back(t0 + gang_ns(gangs[i], nf[i]));
                std::push_heap(cu.begin(), cu.end(), std::greater<double>());
            }
        }
        auto run_single = [&](int32_t v) {
            std::pop_heap(cu.begin(), cu.end(), std::greater<double>());
            cu.back() += single_ns(v);
            std::push_heap(cu.begin(), cu.end(), std::greater<double>());
        };
        for (size_t i = n; i < opt.size(); ++i) run_single(opt[i]);
        for (int32_t v : rest) run_single(v);
        return *std::max_element(cu.begin(), cu.end());
    };
    int best_n = 0, best_n3 = 0;
    if (forced >= 0) {
        best_n = std::min(std::min(forced, (int)opt.size()), SMM_MAX_PAIRS - (int)must.size());
        best_n3 = forced3 >= 0 ? forced3 : 0;
    } else {
        double best_t = 1e300;
        // (host time is on the caller's critical path: a handful of candidates, ~b heap operations each)
        static const int n3s[] = {0, 1, 2, 4, 8, 16, 32, 64};
        for (int n = 0; n <= eligible; n += (eligible <= 8 ? 1 : (n < 32 ? 8 : (n < 64 ? 16 : 32)))) {
            for (int n3 : n3s) {
                // triples only when every workgroup of the launch fits the GPU at once (a latency-bound launch: few
                // videos, CUs to spare); on a full GPU their third CU costs the one-CU videos more than it gains
                // (measured on cfg3: 32 pairs 5.0 ms, 32 triples 5.4 ms)
                if (n3 > n + (int)must.size() || (n3 > 0 && b + n + (int)must.size() + n3 > n_cu)) continue;
                build(n, n3);
                const double t = simulate(n);
                if (std::getenv("SMM_VERBOSE"))
                    std::fprintf(stderr, "libsmmdp:   %d optional gangs, %d triples -> %.3f ms predicted\n", n, n3, t * 1e-6);
                if (t < best_t * 0.98) { best_t = t; best_n = n; best_n3 = n3; }   // more gangs only for a clear gain
            }
        }
    }
    build(best_n, best_n3);
    if (std::getenv("SMM_VERBOSE"))
        std::fprintf(stderr, "libsmmdp: %d videos, %zu with > 21 states (always in a gang), %d optional gangs of %d eligible, %d triples, %d CUs\n",
                     b, must.size(), best_n, eligible, best_n3, n_cu);
    // new order: gangs (most expensive first), then the singles in their old relative order
    std::vector<int32_t> singles;
    {
        std::vector<char> in_gang(b, 0);
        for (size_t i = 0; i < gangs.size(); ++i) { in_gang[gangs[i]] = 1; hv[gangs[i]].nfol = nf[i]; }
        for (int i = 0; i < b; ++i)
            if (!in_gang[order[i]]) singles.push_back(order[i]);
    }
    std::copy(gangs.begin(), gangs.end(), order);
    std::copy(singles.begin(), singles.end(), order + gangs.size());
    *big_ok = true;
    return (int)gangs.size();
}

static bool band_mode(int kp_max, int c_need);

// Validates the metadata, builds SmmVideo[] (+ longest-first block order) and stages it into the workspace.
// want_gangs: plan gangs for the Viterbi kernel (a list-schedule simulation, ~0.3 ms of host time at 360 videos: only
// the entry points that launch that kernel ask for it)
// Split of a decode (smm_decode_f32) into [critical videos | the rest].  The DP kernel's time is the time of the launch's
// longest videos (one workgroup each, ~0.3 us per frame), while the emission scorer in front of it streams EVERY video's
// features (0.6 ms on cfg3).  With the few longest videos scored first, their DP can start at once and the rest of the
// corpus is scored -- and then decoded -- on a second stream beside it: the second part must be through before the first
// is, i.e. its longest video has to be shorter than the launch's longest by the time the emission scorer takes.
// Returns how many videos of `order` (reordered: critical ones first, both parts keep their order) form the first part;
// 0: no split (few videos, a flat length distribution, or nothing to hide).
static int choose_split(const SmmVideo *hv, int32_t *order, int b, int d, int c_max, int64_t total_frames)
{
    if (b < 24 || std::getenv("SMM_NO_SPLIT")) return 0;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            n_cu = 0;
        if (n_cu <= 0) { n_cu = 0; return 0; }
    }
    const double em_us = (double)total_frames * (4.0 * d + 8.0 * c_max) / 4.0e6;      // ~4 TB/s of algorithmic bytes
    const char *mn = std::getenv("SMM_SPLIT_MIN_US");                                  // (test hook: split small launches too)
    if (em_us < (mn ? std::atof(mn) : 100.0)) return 0;
    int tmax = 0;
    for (int i = 0; i < b; ++i) tmax = std::max(tmax, hv[i].T);
    // a video of the second part starts em_us later than the critical ones and must not outlast them: the critical set is
    // the videos within em_us / split_ns + margin frames of the longest.  Same-box scan on cfg3 (scripts/gpu_ab_cfg3.sh,
    // BAND kernel of round 3, ~250 ns per frame beside a full GPU): 230 / 250 ns 3.72-3.74 ms per step, 300 3.58-3.62,
    // 400 3.59-3.62, 600 3.65-3.71, 1000+ 3.68-3.70 -- the em_us estimate below (4 TB/s) is already on the long side.
    // SMM_SPLIT_NS / SMM_SPLIT_MARGIN: tuning aids
    static const double split_ns = [] { const char *e = std::getenv("SMM_SPLIT_NS"); return e ? std::atof(e) : 330.0; }();
    static const int split_margin = [] { const char *e = std::getenv("SMM_SPLIT_MARGIN"); return e ? std::atoi(e) : 400; }();
    const int thr = tmax - (int)(em_us * 1000.0 / split_ns) - split_margin;
    int n1 = 0;
    for (int i = 0; i < b; ++i) n1 += hv[i].T >= thr;
    if (thr <= 0 || n1 < 1 || n1 > b / 3 || n1 > n_cu / 2 || b - n1 < 16) return 0;
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
// part (error words, gang counters) stays in the caller's workspace and is cleared per call as before.  Plans are never
// freed (a captured graph may still point at one); past 64 MB of them, or with SMM_PLAN_CACHE=0, or when the first call
// with some inputs happens under stream capture (no allocation there), a call is staged the old way, into its workspace.
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
struct PlanCache {
    std::mutex mu;
    std::unordered_map<uint64_t, std::vector<PlanEntry *>> entries;     // by FNV-1a of the key; compared in full on a hit
    size_t n = 0, bytes = 0, key_bytes = 0;                            // plans, device bytes, host bytes of their keys
    char *slab = nullptr;                                               // metadata buffers are cut from 1 MB device slabs
    size_t slab_left = 0;
} g_plans;
constexpr size_t SMM_PLAN_MAX_BYTES = (size_t)64 << 20, SMM_PLAN_MAX_ENTRIES = 8192, SMM_PLAN_SLAB = (size_t)1 << 20;

uint64_t plan_hash(const std::vector<char> &k)
{
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : k) { h ^= c; h *= 1099511628211ull; }
    return h;
}

// (called with the cache locked)
char *plan_alloc(size_t bytes)
{
    bytes = align_up(bytes, 256);
    if (bytes > g_plans.slab_left) {
        const size_t want = std::max(bytes, SMM_PLAN_SLAB);
        char *p = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&p), want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        g_plans.slab = p;                  // (the rest of the previous slab is given up: plans are never freed)
        g_plans.slab_left = want;
    }
    char *r = g_plans.slab;
    g_plans.slab += bytes;
    g_plans.slab_left -= bytes;
    return r;
}

void plan_key_append(std::vector<char> &k, const void *p, size_t n)
{
    const char *c = static_cast<const char *>(p);
    k.insert(k.end(), c, c + n);
}

std::vector<char> plan_key(const smm_shape *s, const int64_t *lengths, const int64_t *frame_off, const int32_t *group,
                           const int32_t *kp, const int32_t *n_states, bool want_gangs, int cum_chunk, bool want_split)
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
    const int32_t f[3] = {want_gangs, cum_chunk, want_split};
    plan_key_append(k, f, sizeof(f));
    // the environment the planning functions read (tests and A/B runs flip these between calls)
    for (const char *name : {"SMM_BAND", "SMM_PAIRS", "SMM_TRIPLES", "SMM_NW", "SMM_NO_SPLIT", "SMM_SPLIT_MIN_US"}) {
        const char *e = std::getenv(name);
        const char sep = e ? 1 : 0;
        plan_key_append(k, &sep, 1);
        if (e) plan_key_append(k, e, std::strlen(e) + 1);
    }
    return k;
}

void plan_point(const SmmPlan &p, void *ws, Staged *out)
{
    char *base = static_cast<char *>(ws);
    out->err = reinterpret_cast<int32_t *>(base + p.o_err);
    out->pair_flags = reinterpret_cast<int32_t *>(base + p.o_pflags);
    out->hist = reinterpret_cast<double *>(base + p.meta_bytes);
    out->elp = out->hist + p.hist_doubles;
    out->tabs = out->elp + p.elp_doubles;
    out->band = out->tabs + p.tab_doubles;
}
}  // namespace

static int stage_uncached(const smm_shape *s, const int64_t *lengths, const int64_t *frame_off, const int32_t *group,
                          const int32_t *kp, const int32_t *n_states, void *ws, size_t ws_bytes, hipStream_t stream, Staged *out,
                          bool want_gangs, int cum_chunk, bool want_split, char *meta_dst, SmmPlan *plan_out);

static int stage(const smm_shape *s, const int64_t *lengths, const int64_t *frame_off, const int32_t *group,
                 const int32_t *kp, const int32_t *n_states, void *ws, size_t ws_bytes, hipStream_t stream, Staged *out,
                 bool want_gangs = false, int cum_chunk = 0, bool want_split = false)
{
    if (!shape_ok(s) || !lengths || !frame_off || !n_states || !ws) return SMM_ERR_ARG;
    static const bool enabled = [] { const char *e = std::getenv("SMM_PLAN_CACHE"); return !(e && std::atoi(e) == 0); }();
    if (!enabled) return stage_uncached(s, lengths, frame_off, group, kp, n_states, ws, ws_bytes, stream, out, want_gangs, cum_chunk, want_split, nullptr, nullptr);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
    const bool capturing = cap != hipStreamCaptureStatusNone;
    const std::vector<char> key = plan_key(s, lengths, frame_off, group, kp, n_states, want_gangs, cum_chunk, want_split);
    const uint64_t hk = plan_hash(key) ^ (uint64_t)dev;
    PlanEntry *hit = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_plans.mu);
        auto it = g_plans.entries.find(hk);
        if (it != g_plans.entries.end())
            for (PlanEntry *e : it->second)
                if (e->device == dev && e->key.size() == key.size() && std::memcmp(e->key.data(), key.data(), key.size()) == 0) { hit = e; break; }
        if (hit && !hit->settled) {
            // the upload was queued on another stream, maybe: usable once it is known to be through (or on that very stream)
            if (hipEventQuery(hit->ready) == hipSuccess) hit->settled = true;
            else if (hit->stream != stream) {
                if (capturing || hipStreamWaitEvent(stream, hit->ready, 0) != hipSuccess) hit = nullptr;
            }
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
    const SmmPlan p0 = make_plan(s, lengths);
    bool keep = !capturing;
    if (keep) {
        std::lock_guard<std::mutex> lock(g_plans.mu);
        keep = g_plans.bytes + p0.o_err <= SMM_PLAN_MAX_BYTES && g_plans.n < SMM_PLAN_MAX_ENTRIES &&
               g_plans.key_bytes + key.size() <= SMM_PLAN_MAX_BYTES;
        if (keep) {
            dev_meta = plan_alloc(p0.o_err);
            keep = dev_meta != nullptr;
            if (keep) g_plans.bytes += align_up(p0.o_err, 256);
        }
    }
    SmmPlan plan{};
    const int rc = stage_uncached(s, lengths, frame_off, group, kp, n_states, ws, ws_bytes, stream, out, want_gangs, cum_chunk, want_split,
                                  keep ? dev_meta : nullptr, &plan);
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
    g_plans.n += 1;
    g_plans.key_bytes += key.size();
    return SMM_OK;
}

static int stage_uncached(const smm_shape *s, const int64_t *lengths, const int64_t *frame_off, const int32_t *group,
                          const int32_t *kp, const int32_t *n_states, void *ws, size_t ws_bytes, hipStream_t stream, Staged *out,
                          bool want_gangs, int cum_chunk, bool want_split, char *meta_dst, SmmPlan *plan_out)
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
        hv[i].nfol = 0;
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
    out->n_pairs = 0;
    out->pairs_cover_big = false;
    out->band_mode = want_gangs && band_mode(kp_max, c_need);
    if (want_gangs && !out->band_mode) out->n_pairs = choose_pairs(hv, ho, n_states, s->b, kp_max, c_need, &out->pairs_cover_big);
    out->n_split = (want_split && out->n_pairs == 0) ? choose_split(hv, ho, s->b, s->d, s->c_max, s->total_frames) : 0;
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

    char *base = static_cast<char *>(ws);
    // videos | order | n_states travel as kernel arguments (no pageable copy: the host never waits for the stream) -- into
    // the workspace, or into the resident plan's own buffer (meta_dst); the error word and the gang counters behind them
    // (always in the workspace) start at zero
    char *meta = meta_dst ? meta_dst : base;
    SMM_HIP((hipError_t)smm_upload_meta(meta, host.data(), p.o_err, stream));
    SMM_HIP((hipError_t)smm_zero_async(base + p.o_err, p.meta_bytes - p.o_err, stream));
    out->videos = reinterpret_cast<SmmVideo *>(meta);
    out->order = reinterpret_cast<int32_t *>(meta + p.o_order);
    out->n_states = reinterpret_cast<int32_t *>(meta + p.o_nstates);
    out->em_cum = reinterpret_cast<int32_t *>(meta + p.o_emcum);
    plan_point(p, ws, out);
    out->kp_max = kp_max;
    out->c_need = c_need;
    if (plan_out) *plan_out = p;
    return SMM_OK;
}

// Viterbi at K > 512: BAND mode (smm_viterbi.hip) for up to 28 states; SMM_BAND=0 brings the 1024-slot rings and the gangs
// back (A/B measurements, and the shapes above 28 states)
static bool band_mode(int kp_max, int c_need)
{
    if (kp_max <= 512 || c_need > 28 || std::getenv("SMM_NW")) return false;
    if (std::getenv("SMM_PAIRS") || std::getenv("SMM_TRIPLES")) return false;     // an explicit gang configuration is asked for
    const char *e = std::getenv("SMM_BAND");
    return !(e && std::atoi(e) == 0);
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
    if ((size_t)((s->d + 15) & ~15) * (st.c_need <= 16 ? 21 : 37) * sizeof(double) > 160 * 1024)
        return SMM_ERR_UNSUPPORTED;   // the group's weight table must fit the CU's LDS (D <= 640 at 32 states)
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
} g_dp_timing;

bool dp_timing_begin(hipStream_t stream, std::pair<hipEvent_t, hipEvent_t> &ev)
{
    std::lock_guard<std::mutex> lock(g_dp_timing.mu);
    if (!g_dp_timing.on) return false;
    if (!g_dp_timing.pool.empty()) { ev = g_dp_timing.pool.back(); g_dp_timing.pool.pop_back(); }
    else if (hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess) return false;
    return hipEventRecord(ev.first, stream) == hipSuccess;
}

void dp_timing_end(hipStream_t stream, const std::pair<hipEvent_t, hipEvent_t> &ev)
{
    (void)hipEventRecord(ev.second, stream);
    std::lock_guard<std::mutex> lock(g_dp_timing.mu);
    g_dp_timing.rec.push_back(ev);
}
}  // namespace

extern "C" void smm_dp_timing_enable(int on)
{
    std::lock_guard<std::mutex> lock(g_dp_timing.mu);
    g_dp_timing.on = on != 0;
}

extern "C" int smm_dp_timing_read(float *ms, int cap)
{
    std::vector<std::pair<hipEvent_t, hipEvent_t>> rec;
    {
        std::lock_guard<std::mutex> lock(g_dp_timing.mu);
        rec.swap(g_dp_timing.rec);
    }
    int n = 0;
    for (auto &ev : rec) {
        float t = 0.f;
        if (hipEventSynchronize(ev.second) == hipSuccess && hipEventElapsedTime(&t, ev.first, ev.second) == hipSuccess && ms && n < cap)
            ms[n] = t;
        ++n;
    }
    std::lock_guard<std::mutex> lock(g_dp_timing.mu);
    for (auto &ev : rec) g_dp_timing.pool.push_back(ev);
    return n;
}

static int run_viterbi(const smm_shape *s, const Staged &st, const double *elp, const double *trans, const double *init,
                       const double *len_scores, const double *endpen, const int64_t *class_map, int64_t *spans,
                       int64_t *labels, double *best, int32_t *n_segs, hipStream_t stream, int first = 0, int count = -1,
                       bool prep = true, bool launch = true)
{
    if (!elp || !trans || !init || !len_scores) return SMM_ERR_ARG;
    SmmDpArgs a{};
    a.videos = st.videos; a.order = st.order; a.n_states = st.n_states;
    const bool no_eos = (s->flags & SMM_SHAPE_NO_EOS) != 0;
    a.elp = elp; a.trans = trans; a.init = init; a.len = len_scores; a.endpen = no_eos ? nullptr : endpen; a.class_map = class_map;
    a.hist = st.hist; a.spans = spans; a.labels = labels; a.best = best; a.n_segs = n_segs; a.err = st.err;
    a.c_max = s->c_max; a.k_rows = s->k_rows; a.t_max = s->t_max; a.b = s->b;
    if (count >= 0) { a.order = st.order + first; a.b = count; }
    {
        const char *dbg = std::getenv("SMM_DEBUG_FLAGS");   // profiling / test aid, see SmmDpArgs::flags
        a.flags = dbg ? std::atoi(dbg) : 0;
    }
    if (no_eos) a.flags |= 8;
    if (const char *e = std::getenv("SMM_SPEC")) { if (std::atoi(e) == 0) a.flags |= 256; }   // A/B aid: no speculative transition
    a.n_pairs = st.n_pairs;
    a.pair_flags = st.pair_flags;
    if (st.pairs_cover_big) a.flags |= 4;
    if (st.band_mode) {
        double *len_t = st.band, *band_tab = st.band + (size_t)s->n_groups * s->c_max * SMM_BAND_ROW;
        if (prep) smm_launch_band_tables(len_scores, st.n_states, len_t, band_tab, s->n_groups, s->c_max, s->k_rows, stream);
        a.len_t = len_t;
        a.band_tab = band_tab;
        a.flags |= 128;
    }
    if (ring_regs(st.kp_max) == 1 && !std::getenv("SMM_NO_BT_WINDOW")) {
        // short segments: window back-trace (smm_viterbi.hip); W >= 2 kp keeps a window good for many segments
        // about 48 KB of window (three workgroups per CU stay possible), at least 2 kp positions, at most 512
        int w = (int)(48 * 1024 / (24 * (size_t)st.c_need)) & ~7;
        w = std::min(512, std::max(w, (2 * st.kp_max + 7) & ~7));
        const size_t bytes = sizeof(double) * ((size_t)3 * w + st.c_need + st.kp_max) * st.c_need;
        if (bytes <= 126 * 1024) { a.bt_window = w; a.bt_dyn_bytes = (int32_t)bytes; }
    }
    if (!launch) { SMM_HIP(hipGetLastError()); return SMM_OK; }      // (prep only)
    std::pair<hipEvent_t, hipEvent_t> tev;
    const bool timed = dp_timing_begin(stream, tev);
    const int rc = smm_launch_viterbi(a, ring_regs(st.kp_max), st.c_need, stream);
    if (timed) dp_timing_end(stream, tev);
    if (rc != SMM_OK) return rc;
    // Gangs depend on their workgroups being resident together, which HIP does not promise (another stream or tenant
    // may hold the CUs): a gang that gives up flags itself and its video is decoded again here, on one CU, behind the
    // main kernel on the same stream.  Without a time-out these launches cost two empty grids.
    if (st.n_pairs > 0 && !(a.flags & 1)) smm_launch_viterbi_recovery(a, st.c_need, stream);
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
    SMM_HIP((hipError_t)smm_zero_async(g_w, sizeof(double) * g * cm * d, hs));
    SMM_HIP((hipError_t)smm_zero_async(g_cst, sizeof(double) * g * cm, hs));
    SMM_HIP((hipError_t)smm_zero_async(g_inv_var, sizeof(double) * d, hs));
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
// DP on the caller's stream goes first), never destroyed.  It is the one piece of state the library keeps; every use is
// bracketed by events on the caller's stream, so from the caller's point of view the call is still ordered on ITS stream
// (and captures into a hipGraph like before: the event wait pulls the second stream into the capture).
#include <mutex>
static hipStream_t aux_stream()
{
    static std::mutex mu;
    static hipStream_t streams[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!streams[dev]) {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&streams[dev], hipStreamNonBlocking, least) != hipSuccess) streams[dev] = nullptr;
    }
    return streams[dev];
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
    hipStream_t aux = st.n_split > 0 ? aux_stream() : nullptr;
    if (!aux) {
        rc = run_emission(shape, st, x, w, cst, inv_var, cons, st.elp, elp32, hs);
        if (rc != SMM_OK) return rc;
        return run_viterbi(shape, st, st.elp, trans, init, len_scores, endpen, class_map, spans, labels, best, n_segs, hs);
    }
    // split decode (choose_split): [band tables, emission of the critical videos] on the caller's stream, then their DP
    // there, while the second stream scores and decodes the rest; the caller's stream waits for it at the end
    const int n1 = st.n_split, n2 = shape->b - st.n_split;
    rc = run_viterbi(shape, st, st.elp, trans, init, len_scores, endpen, class_map, spans, labels, best, n_segs, hs, 0, n1, true, false);
    if (rc != SMM_OK) return rc;
    rc = run_emission(shape, st, x, w, cst, inv_var, cons, st.elp, elp32, hs, 0, n1);
    if (rc != SMM_OK) return rc;
    hipEvent_t fork = nullptr, join = nullptr;
    SMM_HIP(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    SMM_HIP(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    int rc2 = SMM_OK;
    if (hipEventRecord(fork, hs) != hipSuccess || hipStreamWaitEvent(aux, fork, 0) != hipSuccess) rc2 = SMM_ERR_HIP;
    static const int dbg_split = [] { const char *e = std::getenv("SMM_SPLIT_DEBUG"); return e ? std::atoi(e) : 0; }();   // (timing experiments: results INCOMPLETE)
    if (rc2 == SMM_OK && !(dbg_split & 2)) rc2 = run_emission(shape, st, x, w, cst, inv_var, cons, st.elp, elp32, aux, n1, n2);
    if (rc2 == SMM_OK && !(dbg_split & 1))
        rc2 = run_viterbi(shape, st, st.elp, trans, init, len_scores, endpen, class_map, spans, labels, best, n_segs, aux, n1, n2, false);
    rc = run_viterbi(shape, st, st.elp, trans, init, len_scores, endpen, class_map, spans, labels, best, n_segs, hs, 0, n1, false);
    // the join is made even after an error on the way, so that the caller's stream never runs ahead of the second one
    if (hipEventRecord(join, aux) != hipSuccess || hipStreamWaitEvent(hs, join, 0) != hipSuccess) rc2 = SMM_ERR_HIP;
    (void)hipEventDestroy(fork);
    (void)hipEventDestroy(join);
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
    SMM_HIP((hipError_t)smm_zero_async(g_trans, sizeof(double) * g * cm * cm, hs));
    SMM_HIP((hipError_t)smm_zero_async(g_init, sizeof(double) * g * cm, hs));
    SMM_HIP((hipError_t)smm_zero_async(g_len, sizeof(double) * g * shape->k_rows * cm, hs));
    SMM_HIP((hipError_t)smm_zero_async(g_elp, sizeof(double) * (size_t)shape->total_frames * cm, hs));
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
