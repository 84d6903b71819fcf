/*
 * smmdp.h -- C ABI of libsmmdp.so: MI355X (gfx950) semi-Markov decode path.
 *
 * This library replaces, for the `--classifier semimarkov` path of dpfried/action-segmentation,
 * everything between "features are on the device" and "span encoding / frame labels are back":
 *
 *   smm_emission_f64        <- SemiMarkovModule.emission_log_probs / _emission_log_probs_with_means
 *                              (reference src/models/semimarkov/semimarkov_modules.py:324-381)
 *   smm_viterbi_f64/_f32    <- SemiMarkovModule.log_hsmm (modules:416-523) + torch_struct
 *                              SemiMarkovCRF(...).argmax + .struct.from_parts (modules:677-679) + class
 *                              un-mapping (modules:683-691) + semimarkov_utils.spans_to_labels
 *                              (semimarkov_utils.py:51-63) + SemiMarkovModule.trim (modules:532-543)
 *   smm_decode_f32          <- SemiMarkovModule.viterbi end to end (modules:660-696)
 *   smm_logz_f64 / _bwd     <- SemiMarkovCRF(...).partition (modules:657) and its autograd backward
 *                              (reference src/models/semimarkov/semimarkov.py:286)
 *
 * The reference has no FFI: its boundary is the Python call SemiMarkovCRF(scores, lengths) on a dense
 * b x N x K x C x C tensor.  These entry points take the FACTORS of that tensor instead,
 * which is what a maintainer's binding passes (INTEGRATION.md shows the ctypes stub).
 *
 * Conventions
 *   - Plain C, no exceptions; every function returns SMM_OK (0) or a negative smm_status.
 *   - "dev" pointers are HIP device pointers owned by the caller; "host" pointers are small per-video /
 *     per-group metadata arrays in ordinary host memory (the library stages them itself; they may be reused
 *     as soon as the call returns).  Results never depend on earlier calls.  What the library keeps between calls is
 *     listed under "State" below; nothing else is allocated or retained.
 *   - All work is enqueued on `stream` (a hipStream_t passed as void*, NULL = default stream); no call
 *     synchronises.  Distinct streams may be used from distinct threads.
 *   - A "group" is a parameter set (one CrossTask task: its valid classes, transition/init/length tables).
 *     A reference-style batch (corpus.py:613-644: one task, padded) is n_groups = 1, group = NULL,
 *     frame_offset[i] = i * t_max.
 *   - Frames of all videos live on one packed frame axis; video i occupies frames
 *     [frame_offset[i], frame_offset[i] + lengths[i]).
 *   - Tables are fp64, padded to c_max columns: trans[g][to][from] (c_max x c_max), init[g][c_max],
 *     len_scores[g][k_rows][c_max] (row index == segment length, rows 1..k_rows-1 usable; modules:383-398),
 *     class_map[g][c_max + 1] int64: local state -> global class id, entry n_states[g] = EOS id (n_classes).
 *   - Per-video kp[i] = min(K, Tmax of the video's reference batch) reproduces modules:450-452 (NULL: min(k_rows, t_max)).
 *   - endpen[i][c_max] fp64 (dev, nullable): 0 for allowed end states, -1e9 otherwise (modules:462-471).
 *
 * State (all of it released by smm_release_cached_plans(); none of it changes a result)
 *   - resident plans: the staged, immutable metadata of a call whose inputs have been seen twice, in library-owned device
 *     memory (at most 64 MB per process; the entry points that take lengths_host / frame_offset_host stage through it);
 *   - one low-priority stream per device for smm_decode_f32's split decode, and pooled events around it;
 *   - the SMM_* tuning switches, read from the environment once, at first use (smm_env_reload() reads them again);
 *   - smm_dp_timing_*: the event pairs of the measurement aid while it is enabled.
 */
#ifndef SMMDP_H
#define SMMDP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum smm_status {
    SMM_OK = 0,
    SMM_ERR_ARG = -1,          /* null pointer / non-positive size / inconsistent metadata */
    SMM_ERR_UNSUPPORTED = -2,  /* shape outside the compiled kernels (c_max > 32, k_rows > 1024) */
    SMM_ERR_WORKSPACE = -3,    /* workspace too small */
    SMM_ERR_HIP = -4,          /* a HIP runtime call failed (smm_last_hip_error() has the code) */
    SMM_ERR_NO_DEVICE = -5     /* no gfx950 device visible */
} smm_status;

#define SMM_MAX_STATES 32
#define SMM_MAX_K_ROWS 1024

/* Shape of one decode call (plain data; passed by pointer). */
typedef struct smm_shape {
    int32_t b;         /* videos */
    int32_t d;         /* feature dim (emission only) */
    int32_t n_groups;  /* parameter groups */
    int32_t c_max;     /* table column stride, >= every n_states[g], <= SMM_MAX_STATES */
    int32_t k_rows;    /* rows of the length table (= --sm_max_span_length, or 2 for the K==1 HMM table) */
    int32_t t_max;     /* max lengths[i] */
    int32_t flags;     /* SMM_SHAPE_* bits, 0 = the reference's default (add_eos=True) */
    int64_t total_frames; /* extent of the packed frame axis (>= every frame_offset[i] + lengths[i]) */
} smm_shape;

/* add_eos=False of the reference (semimarkov_modules.py:494-505, :660): no EOS label is appended.  The DP positions
 * are the frames themselves: segments cover frames 0 .. T-2 and the video closes with a transition into the label of
 * frame T-1, which contributes its emission only (no length score); end penalties do not apply (endpen is ignored).
 * Viterbi: spans[i][T-1] holds that label and no EOS id is written; log Z and its gradient likewise.  lengths[i] >= 2. */
#define SMM_SHAPE_NO_EOS 1
/* smm_logz_f64 only: also run the time-reversed recursion (the backward messages smm_logz_bwd_f64 needs) in the SAME
 * launch, one extra workgroup per video.  The two directions are independent, so a batch that does not fill the GPU
 * gets its gradient's DP for free; pass the same flag to smm_logz_bwd_f64, which then skips its own reversed run. */
#define SMM_SHAPE_LOGZ_BOTH 2
/* Viterbi entry points: never split a video along the time axis ("Long videos" below).  For callers that know their tables
 * carry hard masks (ordering constraints: transitions / initial states / ends at -1e9): a unit that starts in the middle of
 * such a video from "every state equally good" reaches states the masks forbid there, its cuts do not certify, and every
 * split video would be decoded a second time in one piece -- correct as ever, and slower than not splitting. */
#define SMM_SHAPE_NO_TIME_SPLIT 4

const char *smm_strerror(int status);
int smm_last_hip_error(void);
const char *smm_version(void);
/* number of visible gfx950 devices (0 when none; never initialises a context on failure) */
int smm_device_count(void);

/* Bytes of device workspace any entry point below needs for this shape (lengths: host array [b]).
 * Returns 0 on invalid arguments.  The workspace is scratch: its contents are undefined after a call,
 * except between smm_logz_f64 and smm_logz_bwd_f64. */
size_t smm_workspace_bytes(const smm_shape *shape, const int64_t *lengths_host);

/* Byte offset, inside the workspace, of the int32 error word the kernels set.  1: a NaN / inf-inf reached the DP of
 * some video and its decode stopped early.  It is cleared at the start of every call.  Returns 0 on invalid shape.
 * (The int32 word behind it is always 0: rounds 1-3 counted the time-outs of multi-workgroup "gangs" there, which no
 * longer exist; the third and fourth words are diagnostics of the Viterbi kernel's BAND mode: sources pushed into
 * band 0, delayed band-blocks evaluated; the fifth and sixth count the videos a Viterbi call decoded as several units
 * along the time axis and, of those, the ones it decoded again in one piece because a cut could not be certified or a
 * decision was closer than rounding can tell -- the outputs are the one-piece decode's either way, see "Long videos"
 * below.) */
size_t smm_error_word_offset(const smm_shape *shape);

/*
 * Long videos (Viterbi entry points, EOS mode, span limit > 64).  The decode of one video is one serial chain over its
 * frames; a launch whose CU-time is shorter than its longest video cuts that video along the TIME axis into units that
 * run on different CUs, each warmed up on the frames in front of its own part, and stitches them: the cuts are certified
 * against each other, every decision of the back-trace has to be clear of rounding, the best score is re-evaluated along
 * the path in the one-piece decode's association -- and a video that fails any of it is decoded again in one piece by the
 * same call (csrc/smm_chunk.hip).  spans / labels / best / n_segs are those of the one-piece decode, bit for bit; the
 * workspace bound of smm_workspace_bytes covers it.  SMM_CHUNK=0 in the environment switches the splitting off.
 */

/* The plan "Long videos" would make for this launch on a GPU of n_cu compute units -- host logic only, no device needed (tests;
 * capacity planning): for every unit of every video that would be split, in time order per video: its video, its first
 * position, its length in positions, and how many of those it runs in front of its own part (0 for a video's first unit).
 * Arrays of `cap` entries (any may be NULL); returns the number of units (may exceed cap), 0 when nothing would be split,
 * or a negative smm_status.  group / kp may be NULL as in the decode entry points. */
int smm_time_split_plan(const smm_shape *shape, const int64_t *lengths_host, const int32_t *group_host, const int32_t *kp_host,
                        const int32_t *n_states_host, int n_cu, int32_t *unit_video, int32_t *unit_first, int32_t *unit_len,
                        int32_t *unit_overlap, int cap);

/*
 * Measurement aid (bench.py's roofline; not part of the reference's interface): while enabled, every launch of the
 * Viterbi DP kernel made by smm_viterbi_* / smm_decode_f32 is bracketed by a pair of HIP events on the stream it is
 * launched on (smm_decode_f32 may launch the kernel twice per call, on two streams: smmdp.h / DESIGN.md "split decode").
 * smm_dp_timing_read waits for the recorded launches, writes their durations in milliseconds (launch order, at most
 * `cap`) and forgets them; it returns the number of launches recorded since the last read (which may exceed cap).
 * Not to be enabled around a stream capture.  The two event records cost a few microseconds per launch.
 */
void smm_dp_timing_enable(int on);
int smm_dp_timing_read(float *ms, int cap);
/* as smm_dp_timing_read, plus which launch each one was: tags[i] = 0 the only DP launch of its call, 1 the launch of the
 * critical (longest) videos of a split smm_decode_f32 on the caller's stream, 2 the rest of that call on the library's
 * second stream, 3 the <= 16-state videos of a launch part that holds more videos than the GPU has CUs, decoded in four-wave
 * workgroups (two per CU) on a side stream beside the launch of tag 0 / 2 (either array may be NULL) */
int smm_dp_timing_read_tagged(float *ms, int32_t *tags, int cap);

/*
 * Cost model of the shipped Viterbi kernel (not part of the reference's interface): nanoseconds per frame of ONE video
 * of `n_states` states decoded with segment lengths beyond 512 (BAND mode: the time of its serial chain, which hardly
 * depends on the span limit), as measured on an MI355X with this library's kernels (the constants sit next
 * to the kernel's dispatch and move with it).  ONE place for the number that two host-side decisions need: the split of
 * smm_decode_f32 (how much shorter than the launch's longest video a video must be to start behind the emission pass
 * of the whole corpus) and the balancing of video shards over ranks (batching.batch_cost).  Returns 0 for n_states
 * outside 1..32.
 */
double smm_band_frame_ns(int n_states);

/*
 * Library state (see "State" above).  smm_release_cached_plans frees every resident plan (all devices), the split
 * decode's second streams and all pooled events, and returns the device bytes it gave back.  The caller's promise: no
 * libsmmdp call is in flight on any stream, and no hipGraph captured from a call will be replayed afterwards (a captured
 * call points at its plan's buffer).  smm_cached_plan_bytes: device bytes currently held by resident plans.
 * smm_env_reload: read the SMM_* tuning switches from the environment again (they are read once, at first use:
 * SMM_SPEC, SMM_NO_SPLIT, SMM_SPLIT_MIN_US / _NS / _MARGIN, SMM_PLAN_CACHE, SMM_NO_BT_WINDOW, SMM_FIT_GRID, SMM_SMALL_WG,
 * SMM_CHUNK, SMM_CHUNK_P / _WC / _LMIN, SMM_VERBOSE --
 * none of them changes a result; switches that do exist only in -DSMM_DEV builds of the library).
 */
size_t smm_release_cached_plans(void);
size_t smm_cached_plan_bytes(void);
void smm_env_reload(void);

/*
 * Emission scorer.  elp[t][c] = cst[g][c] + sum_d x[t][d]*w[g][c][d] - 0.5*sum_d x[t][d]^2*inv_var[d] (+ cons[t][c])
 * which is the diagonal-Gaussian log density of modules:324-381 with w = mu/sigma^2,
 * cst = -0.5*sum mu^2/sigma^2 - sum log sigma - D/2 log 2pi.
 *   x        dev fp32 [total_frames][d]
 *   w        dev fp64 [n_groups][d][c_max]  (feature-major, so one frame step reads one contiguous row);
 *   cst      dev fp64 [n_groups][c_max];  inv_var dev fp64 [d]
 *   cons     dev fp32 [total_frames][c_max] or NULL (narration constraints, modules:379-380)
 *   elp64    dev fp64 [total_frames][c_max] or NULL   (frame-major, reference layout)
 *   elp32    dev fp32 [total_frames][c_max] or NULL   (what `return_elp=True` hands back)
 */
int smm_emission_f64(const smm_shape *shape, const int64_t *lengths_host, const int64_t *frame_offset_host,
                     const int32_t *group_host, const int32_t *n_states_host,
                     const float *x, const double *w, const double *cst, const double *inv_var, const float *cons,
                     double *elp64, float *elp32, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Chain rule through the emission scorer (training): with g_elp = dL/d elp (smm_logz_bwd_f64's output),
 *   g_w[g][c][d] = sum_t x[t][d] g_elp[t][c]      g_cst[g][c] = sum_t g_elp[t][c]
 *   g_inv_var[d] = -0.5 sum_t x[t][d]^2 sum_c g_elp[t][c]              (t over the frames of the videos of group g)
 * -- what autograd does behind emission_log_probs (modules:324-381) in the reference's loss.backward()
 * (src/models/semimarkov/semimarkov.py:286).  Outputs are overwritten.
 *   g_w        dev fp64 [n_groups][c_max][d]   CLASS-major (the layout of the reference's gaussian_means; the
 *              transpose of smm_emission_f64's w)
 *   g_cst      dev fp64 [n_groups][c_max];   g_inv_var  dev fp64 [d]
 * Sums leave the workgroups through fp64 atomics: the last bits depend on the order of arrival.
 */
int smm_emission_bwd_f64(const smm_shape *shape, const int64_t *lengths_host, const int64_t *frame_offset_host,
                         const int32_t *group_host, const int32_t *n_states_host,
                         const float *x, const double *g_elp, double *g_w, double *g_cst, double *g_inv_var,
                         void *workspace, size_t workspace_bytes, void *stream);

/*
 * Viterbi decode on emission scores (fp64 path).
 *   elp       dev fp64 [total_frames][c_max]
 *   spans     dev int64 [b][t_max + 1]  span encoding of modules:679-691: global class id at each span start,
 *             -1 continuation, EOS id at position lengths[i], -1 after it           (nullable)
 *   labels    dev int64 [total_frames]  per-frame global class ids (spans_to_labels + trim)   (nullable)
 *   best      dev fp64 [b] Viterbi score (nullable);  n_segs dev int32 [b] (nullable)
 */
int smm_viterbi_f64(const smm_shape *shape, const int64_t *lengths_host, const int64_t *frame_offset_host,
                    const int32_t *group_host, const int32_t *kp_host, const int32_t *n_states_host,
                    const double *elp, const double *trans, const double *init, const double *len_scores,
                    const double *endpen, const int64_t *class_map,
                    int64_t *spans, int64_t *labels, double *best, int32_t *n_segs,
                    void *workspace, size_t workspace_bytes, void *stream);

/* Same with the reference's dtypes at the boundary: fp32 elp [total_frames][c_max] and fp32 tables
 * (log_hsmm's inputs, modules:416-417); converted to fp64 on load, then the same DP. */
int smm_viterbi_f32(const smm_shape *shape, const int64_t *lengths_host, const int64_t *frame_offset_host,
                    const int32_t *group_host, const int32_t *kp_host, const int32_t *n_states_host,
                    const float *elp, const float *trans, const float *init, const float *len_scores,
                    const float *endpen, const int64_t *class_map,
                    int64_t *spans, int64_t *labels, double *best, int32_t *n_segs,
                    void *workspace, size_t workspace_bytes, void *stream);

/* Features -> decode in one call (emission kernel + DP kernel on `stream`); elp32 nullable. */
int smm_decode_f32(const smm_shape *shape, const int64_t *lengths_host, const int64_t *frame_offset_host,
                   const int32_t *group_host, const int32_t *kp_host, const int32_t *n_states_host,
                   const float *x, const double *w, const double *cst, const double *inv_var, const float *cons,
                   const double *trans, const double *init, const double *len_scores,
                   const double *endpen, const int64_t *class_map,
                   int64_t *spans, int64_t *labels, double *best, int32_t *n_segs, float *elp32,
                   void *workspace, size_t workspace_bytes, void *stream);

/*
 * Log-partition (LogSemiring forward) and its backward.
 *   logz   dev fp64 [b]
 *   bwd:   grad_logz dev fp64 [b] (upstream, NULL = ones), outputs g_elp dev fp64 [total_frames][c_max],
 *          g_trans dev fp64 [n_groups][c_max][c_max], g_init [n_groups][c_max], g_len [n_groups][k_rows][c_max]
 *          (all overwritten).  The workspace written by smm_logz_f64 must be passed unchanged to smm_logz_bwd_f64.
 */
int smm_logz_f64(const smm_shape *shape, const int64_t *lengths_host, const int64_t *frame_offset_host,
                 const int32_t *group_host, const int32_t *kp_host, const int32_t *n_states_host,
                 const double *elp, const double *trans, const double *init, const double *len_scores,
                 const double *endpen, double *logz, void *workspace, size_t workspace_bytes, void *stream);

int smm_logz_bwd_f64(const smm_shape *shape, const int64_t *lengths_host, const int64_t *frame_offset_host,
                     const int32_t *group_host, const int32_t *kp_host, const int32_t *n_states_host,
                     const double *elp, const double *trans, const double *init, const double *len_scores,
                     const double *endpen, const double *logz, const double *grad_logz,
                     double *g_elp, double *g_trans, double *g_init, double *g_len,
                     void *workspace, size_t workspace_bytes, void *stream);

/*
 * Factor tables of every parameter group from the model parameters (training steps), and their chain rule.
 * One launch each instead of the differentiable torch ops behind initial_log_probs (modules:284-296),
 * transition_log_probs (:298-322), _length_log_probs_with_rates (:383-414) and the expanded emission_log_probs
 * (:324-381); layouts as the entry points above read them, columns past a group's state count are 0.
 * Every pointer is a DEVICE pointer; nothing is staged and the host never waits.
 *   parameters: init_logits [n], transition_logits [n][n] ([to][from]), poisson_log_rates [n], gaussian_means [n][d],
 *               gaussian_cov [d][d] (its diagonal: tied diagonal covariance), fp32 like the reference's nn.Parameters
 *   init_constraints [n] / transition_constraints [n][n]: bytes, 1 = forbidden (set_transition_constraints,
 *               modules:160-193), or NULL
 *   classes [g][c_max]: class id of each local state (valid_classes);  merged [g][c_max]: its parameter row after
 *               merge_classes;  n_states [g] int32
 * smm_factor_tables_bwd_f64: g_* of the tables in (NULL = no gradient through that table; g_w_class_major is
 * [g][c_max][d], smm_emission_bwd_f64's layout), fp64 gradients of the parameters out (overwritten; sums over groups
 * leave through fp64 atomics).  trans / init: the forward call's outputs.
 */
typedef struct smm_tables_shape {
    int32_t n_classes, d, n_groups, c_max, k_rows;
    int32_t allow_self_transitions;
} smm_tables_shape;
int smm_factor_tables_f64(const smm_tables_shape *shape, const float *init_logits, const float *transition_logits,
                          const float *poisson_log_rates, const float *gaussian_means, const float *gaussian_cov,
                          const uint8_t *init_constraints, const uint8_t *transition_constraints,
                          const int64_t *classes, const int64_t *merged, const int32_t *n_states,
                          double *trans, double *init, double *len_scores, double *w, double *cst, double *inv_var,
                          void *stream);
int smm_factor_tables_bwd_f64(const smm_tables_shape *shape, const float *poisson_log_rates, const float *gaussian_means,
                              const float *gaussian_cov, const uint8_t *init_constraints,
                              const uint8_t *transition_constraints, const int64_t *classes, const int64_t *merged,
                              const int32_t *n_states, const double *trans, const double *init,
                              const double *g_trans, const double *g_init, const double *g_len,
                              const double *g_w_class_major, const double *g_cst,
                              double *g_init_logits, double *g_transition_logits, double *g_poisson_log_rates,
                              double *g_gaussian_means, void *stream);

/*
 * The reference's inner boundary as it stands: semiring DP over DENSE potentials, for lattices small enough to be
 * materialised (reference defaults).  Replaces torch_struct.SemiMarkovCRF(scores, lengths).argmax + from_parts
 * (semiring = 0, max) and .partition (semiring = 1, log) -- call sites semimarkov_modules.py:624, 657, 677-679,
 * src/models/test_semimarkov.py:312-314.
 *   scores   dev fp32 [b][n1][k][c][c] indexed [n][k][c_to][c_from];  lengths_host[b] = positions (max == n1 + 1)
 *   v        dev fp64 [b];  spans dev int64 [b][n1 + 1] (max semiring only, nullable)
 */
size_t smm_dense_workspace_bytes(int32_t b, int32_t n1, int32_t k, int32_t c);
int smm_dense_dp_f32(const float *scores, const int64_t *lengths_host, int32_t b, int32_t n1, int32_t k, int32_t c,
                     int32_t semiring, double *v, int64_t *spans, void *workspace, size_t workspace_bytes, void *stream);
/* Posterior edge marginals of the dense lattice = d sum_i grad_v[i] * logZ_i / d scores -- the gradient the reference
 * obtains by autograd through torch_struct's LogSemiring DP (src/models/semimarkov/semimarkov.py:286 through
 * semimarkov_modules.py:624-657).  Must follow smm_dense_dp_f32(semiring = 1) on the same scores with the SAME
 * workspace (its forward messages are read from there); v = that call's output.
 *   grad_v     dev fp64 [b] upstream gradient (NULL = ones)
 *   marginals  dev fp32 [b][n1][k][c][c], overwritten (0 outside each instance's lattice) */
int smm_dense_marginals_f32(const float *scores, const int64_t *lengths_host, int32_t b, int32_t n1, int32_t k, int32_t c,
                            const double *v, const double *grad_v, float *marginals,
                            void *workspace, size_t workspace_bytes, void *stream);

/*
 * Evaluation counters of decoded frame labels against ground truth -- the per-frame loops of the reference's
 * src/evaluation/accuracy.py as driven by Datasplit.accuracy_corpus (src/data/corpus.py:486-565).  Integer work only;
 * the label assignment (identity / Hungarian on the confusion table) and the final ratios are the caller's.
 *   pred        dev int64 [total_frames]            global class ids (what smm_decode_f32 writes to `labels`)
 *   gt          dev int64 [total_frames][gt_width]  ground-truth ids, first column = "the" label, -1 = no further label
 *   local_of    dev int32 [n_groups][n_labels]      global id -> local id in [0, c_max) of the video's task, -1 = not
 *                                                   in the task (such frames are counted under local id c_max)
 * smm_eval_confusion_i64  (accuracy.py:232-283 voting table, :500-521 per-class masks)
 *   confusion   dev int64 [n_groups][c_max+1][c_max+1]  frames with (first gt label, predicted label); overwritten
 * smm_eval_videos_i64     (accuracy.py:538-576 frame loop, :21-37 + :364-408 run lengths and edit distance,
 *                          :410-472 step recall)
 *   video_key   host int32 [b] index of the video inside its task (seeds the random frame draw; NULL: i)
 *   cluster_of  dev int32 [n_groups][c_max+1]      local gt id -> predicted label that it owns after assignment, as a
 *                                                  local id, or c_max+1+j for an invented label j, or -1 (none)
 *   gt_is_bg    dev uint8 [n_groups][c_max+1]      local gt id is a background class
 *   pred_is_bg  dev uint8 [n_groups][2*(c_max+1)]  (extended) predicted id is owned by a background class
 *   seed        the frame drawn for `single_step_recall` is the candidate with the smallest hash(seed, key, t)
 *   counters    dev int64 [b][SMM_EVAL_COUNTERS]   per video, indexed by smm_eval_counter; overwritten
 */
#define SMM_EVAL_MAX_LABELS 63
#define SMM_EVAL_COUNTERS 32
typedef enum smm_eval_counter {
    SMM_EV_FRAMES = 0, SMM_EV_SEGS_GT = 1, SMM_EV_SEGS_PRED = 2, SMM_EV_SEGS_PRED_NON_BG = 3,
    SMM_EV_MULTI = 4,            /* frames with more than one gt label */
    SMM_EV_GT_LABELS = 5,        /* sum of gt labels per frame (recall denominator) */
    SMM_EV_TP = 6,               /* prediction owned by one of the frame's gt labels */
    SMM_EV_PRED_BG = 7, SMM_EV_TRUE_BG = 8,
    SMM_EV_IOU_DEN = 9, SMM_EV_IOU_NUM = 10,          /* frames not (gt bg and pred bg); of those, true positives */
    SMM_EV_GT_LABELS_NON_BG = 11, SMM_EV_FRAMES_NON_BG = 12, SMM_EV_TP_NON_BG = 13,
    SMM_EV_STEPS = 14, SMM_EV_STEPS_NON_BG = 15,      /* distinct (remapped) gt labels of the video */
    SMM_EV_DRAW_HIT = 16, SMM_EV_DRAW_HIT_NON_BG = 17, SMM_EV_MID_HIT = 18, SMM_EV_MID_HIT_NON_BG = 19,
    SMM_EV_TYPES = 20, SMM_EV_TYPES_NON_BG = 21,      /* distinct predicted labels */
    SMM_EV_OTHER = 22,           /* labels outside the task's table (must be 0 for the statistics to be meaningful) */
    SMM_EV_LEVENSHTEIN = 23
} smm_eval_counter;

typedef struct smm_eval_shape {
    int32_t b;          /* videos */
    int32_t n_groups;   /* tasks */
    int32_t c_max;      /* local label ids per task, <= SMM_EVAL_MAX_LABELS */
    int32_t n_labels;   /* size of the global label space */
    int32_t gt_width;   /* ground-truth labels per frame (>= 1) */
    int32_t t_max;      /* max lengths[i] */
    int64_t total_frames;
} smm_eval_shape;

size_t smm_eval_workspace_bytes(const smm_eval_shape *shape, const int64_t *lengths_host);
int smm_eval_confusion_i64(const smm_eval_shape *shape, const int64_t *lengths_host, const int64_t *frame_offset_host,
                           const int32_t *group_host, const int64_t *pred, const int64_t *gt, const int32_t *local_of,
                           int64_t *confusion, void *workspace, size_t workspace_bytes, void *stream);
int smm_eval_videos_i64(const smm_eval_shape *shape, const int64_t *lengths_host, const int64_t *frame_offset_host,
                        const int32_t *group_host, const int32_t *video_key_host, const int64_t *pred, const int64_t *gt,
                        const int32_t *local_of, const int32_t *cluster_of, const uint8_t *gt_is_bg,
                        const uint8_t *pred_is_bg, uint32_t seed, int64_t *counters,
                        void *workspace, size_t workspace_bytes, void *stream);

/*
 * Sufficient statistics of the closed-form supervised fit -- semimarkov_utils.semimarkov_sufficient_stats
 * (reference src/models/semimarkov/semimarkov_utils.py:74-126) as consumed by SemiMarkovModule.fit_supervised
 * (semimarkov_modules.py:195-256).  One pass over the features (HBM-bound).
 *   x        dev fp32 [total_frames][d];  labels dev int64 [total_frames] global class ids in [0, n_classes)
 *   max_k    --sm_max_span_length: a run of one label counts as spans of at most max_k - 1 frames
 *            (labels_to_spans, utils.py:6-23); <= 0: runs are never cut
 *   sum_x    dev fp64 [n_classes][d]  per-class feature sums;   sum_x2 dev fp64 [d]  sum of squares over all frames
 *   frame_counts / span_counts / span_start_counts dev int64 [n_classes];
 *   span_transition_counts dev int64 [n_classes][n_classes] indexed [to][from]            (all outputs overwritten)
 * The int32 at smm_fit_error_word_offset(b) in the workspace is non-zero when a label was outside [0, n_classes).
 */
size_t smm_fit_workspace_bytes(int32_t b);
size_t smm_fit_error_word_offset(int32_t b);
int smm_fit_stats_f64(int32_t b, const int64_t *lengths_host, const int64_t *frame_offset_host, int64_t total_frames,
                      int32_t d, int32_t n_classes, int32_t max_k, const float *x, const int64_t *labels,
                      double *sum_x, double *sum_x2, int64_t *frame_counts, int64_t *span_counts,
                      int64_t *span_start_counts, int64_t *span_transition_counts,
                      void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SMMDP_H */
