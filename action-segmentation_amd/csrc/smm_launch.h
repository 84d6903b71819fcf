// smm_launch.h -- launch wrappers shared between the kernel translation units and smm_api.hip.
#pragma once
#include "smm_device.h"

struct SmmEmArgs {
    const SmmVideo *videos;
    const int32_t *order;    // [b] block -> video (most work first)
    const int32_t *n_states;
    const float *x;          // [total_frames][d]
    const double *w;         // [g][d][c_max]
    const double *cst;       // [g][c_max]
    const double *inv_var;   // [d]
    const float *cons;       // [total_frames][c_max] or null
    double *elp64;           // [total_frames][c_max] or null
    float *elp32;            // [total_frames][c_max] or null
    int32_t d, c_max, b;
};

// Host metadata -> device, stream-ordered, WITHOUT a host-to-device copy: the bytes travel in the kernel-argument
// segment of tiny copy kernels (2 KB per launch).  A hipMemcpyAsync from pageable host memory makes the host wait until
// the stream has drained (and cannot be captured into a hipGraph); this does neither.  Returns a hipError_t as int.
int smm_upload_meta(void *dst_dev, const void *src_host, size_t bytes, hipStream_t stream);
// zero fill by a kernel (a hipMemsetAsync captured into a hipGraph does not replay reliably: smm_api.hip); hipError_t as int
int smm_zero_async(void *dst_dev, size_t bytes, hipStream_t stream);
// ... of up to eight regions in one launch
int smm_zero_multi_async(void *const *dst_dev, const size_t *bytes, int n, hipStream_t stream);

// flat grid: blk_cum[i] (device, [b + 1]) = workgroups of the videos order[0..i), built by the host from
// smm_emission_tiles_per_wave / smm_emission_blocks
int smm_emission_tiles_per_wave(int64_t total_frames, int b);
int smm_emission_blocks(int t, int tpw);
size_t smm_emission_lds_bytes(int d, int c_need);   // LDS of one workgroup: the largest class set's weight table + inv_var
void smm_launch_emission(const SmmEmArgs &a, int c_need, int tpw, int n_blocks, const int32_t *blk_cum, int64_t total_frames,
                         hipStream_t stream, int blk_base = 0, int vid0 = 0, int nvid = -1);
void smm_launch_widen(const float *src, double *dst, size_t n, hipStream_t stream);

// chain rule through the emission scorer (smm_emission.hip): outputs must be zero at launch
struct SmmEmBwdArgs {
    const SmmVideo *videos;
    const int32_t *order;    // [b]
    const int32_t *n_states;
    const int32_t *cum;      // [b + 1] chunks of smm_emission_bwd_chunk() frames before each video of `order`
    const float *x;          // [total_frames][d]
    const double *g_elp;     // [total_frames][c_max]
    double *g_w;             // [g][c_max][d]   (class-major)
    double *g_cst;           // [g][c_max]
    double *g_iv;            // [d]
    int32_t d, c_max, b, n_chunks;
};
int smm_emission_bwd_chunk();
void smm_launch_emission_bwd(const SmmEmBwdArgs &a, int c_need, hipStream_t stream);
// returns an smm_status; r = ring registers per lane (1,2,4,..,64), c_need = max states of any group
int smm_launch_viterbi(const SmmDpArgs &a, int r, int c_need, hipStream_t stream);
int smm_launch_viterbi_small(const SmmDpArgs &a, hipStream_t stream);   // BAND mode, <= 16 states, four-wave workgroups (two per CU)
int smm_launch_viterbi_repair(const SmmDpArgs &a, int c_need, hipStream_t stream);   // BAND mode (time-split decode: smm_chunk.hip)
// Viterbi BAND mode: the state-major length table and the skip-test bounds of every (group, state) (smm_viterbi.hip)
void smm_launch_band_tables(const double *len, const int32_t *n_states, double *len_t, double *band_tab, double *dmin_t,
                            int n_groups, int cm, int k_rows, hipStream_t stream);
// time-split Viterbi decode (smm_chunk.hip): serial prefix sums at the units' first positions; certification, back-trace and
// outputs of the split videos (redo[i] = 1: video i of `cvs` has to be decoded again in one piece)
void smm_launch_cum_anchors(const SmmDpArgs &a, const SmmChunkVideo *cvs, int n_split, double *anchors, hipStream_t stream);
void smm_launch_chunk_stitch(const SmmDpArgs &a, const SmmChunkVideo *cvs, int n_split, int32_t *redo, hipStream_t stream);
// tuning switches other translation units read (smm_api.hip: SmmEnv; read once, see smm_env_reload)
int smm_env_fit_grid();        // SMM_FIT_GRID (0: default)
int smm_env_emission_v2();     // SMM_EMISSION_V2 (-DSMM_DEV builds only)
// LogSemiring forward: logz[b]; same arguments as the Viterbi launch
int smm_launch_logz(const SmmDpArgs &a, double *logz, int r, int c_need, hipStream_t stream);

struct SmmBwdArgs {
    const SmmVideo *videos;
    const int32_t *n_states;
    const double *trans;       // forward tables [g][c_max][c_max], [g][k_rows][c_max]
    const double *len;
    const double *hist;        // per video: F_cum, F_h, F_g, B_cum, B_h, B_g, (scratch) hT0, hT1; each [T+1][c_max]
    const double *logz;        // [b]
    const double *grad_logz;   // [b] or null (= 1)
    double *g_elp;             // [total_frames][c_max]
    double *g_trans;           // [g][c_max][c_max]
    double *g_init;            // [g][c_max]
    double *g_len;             // [g][k_rows][c_max]
    int32_t c_max, k_rows, b;
    const double *elp;         // [total_frames][c_max]   (no_eos only: the closing label's emission)
    int32_t no_eos;            // add_eos=False (smmdp.h: SMM_SHAPE_NO_EOS)
};
void smm_launch_transpose(const double *src, double *dst, int g, int cm, hipStream_t stream);
void smm_launch_marginals(const SmmBwdArgs &a, int t_max, int kp_max, hipStream_t stream);

struct SmmDenseArgs {
    const float *edge;        // [b][n1][k][c][c]  (c_to, c_from)
    const int64_t *lengths;   // [b] positions (device)
    double *alpha;            // scratch [b][k][k][c]
    double *beta;             // scratch [b][n1+1][c]
    uint8_t *bp_from;         // scratch [b][n1][k][c]   (max semiring)
    uint16_t *bp_k;           // scratch [b][n1+1][c]
    double *v;                // [b]
    int64_t *spans;           // [b][n1+1] or null
    int32_t b, n1, k, c;
};
void smm_launch_dense(const SmmDenseArgs &a, bool log_semiring, hipStream_t stream);
// posterior edge marginals (x upstream gradient) from the beta a LogSemiring smm_launch_dense left in a.beta and a.v
void smm_launch_dense_marginals(const SmmDenseArgs &a, double *rmsg, const double *grad_v, float *out, hipStream_t stream);
