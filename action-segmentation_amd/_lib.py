"""ctypes binding of libsmmdp.so (include/smmdp.h).  No fallback: a missing library is an error."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMM_LIB_PATH") or os.path.join(HERE, "libsmmdp.so")   # (override: diagnostic builds only)

SYMBOLS = [
    "smm_strerror", "smm_last_hip_error", "smm_version", "smm_device_count", "smm_workspace_bytes",
    "smm_error_word_offset", "smm_dp_timing_enable", "smm_dp_timing_read", "smm_dp_timing_read_tagged", "smm_band_frame_ns", "smm_time_split_plan",
    "smm_env_reload", "smm_release_cached_plans", "smm_cached_plan_bytes",
    "smm_emission_f64", "smm_emission_bwd_f64", "smm_viterbi_f64", "smm_viterbi_f32", "smm_decode_f32", "smm_logz_f64", "smm_logz_bwd_f64",
    "smm_factor_tables_f64", "smm_factor_tables_bwd_f64",
    "smm_dense_workspace_bytes", "smm_dense_dp_f32", "smm_dense_marginals_f32",
    "smm_eval_workspace_bytes", "smm_eval_confusion_i64", "smm_eval_videos_i64",
    "smm_fit_workspace_bytes", "smm_fit_error_word_offset", "smm_fit_stats_f64",
]


class SmmShape(ctypes.Structure):
    _fields_ = [("b", ctypes.c_int32), ("d", ctypes.c_int32), ("n_groups", ctypes.c_int32),
                ("c_max", ctypes.c_int32), ("k_rows", ctypes.c_int32), ("t_max", ctypes.c_int32),
                ("flags", ctypes.c_int32), ("total_frames", ctypes.c_int64)]


class SmmTablesShape(ctypes.Structure):
    _fields_ = [("n_classes", ctypes.c_int32), ("d", ctypes.c_int32), ("n_groups", ctypes.c_int32),
                ("c_max", ctypes.c_int32), ("k_rows", ctypes.c_int32), ("allow_self_transitions", ctypes.c_int32)]


SHAPE_NO_EOS = 1
SHAPE_LOGZ_BOTH = 2
SHAPE_NO_TIME_SPLIT = 4


class SmmEvalShape(ctypes.Structure):
    _fields_ = [("b", ctypes.c_int32), ("n_groups", ctypes.c_int32), ("c_max", ctypes.c_int32),
                ("n_labels", ctypes.c_int32), ("gt_width", ctypes.c_int32), ("t_max", ctypes.c_int32),
                ("total_frames", ctypes.c_int64)]


EVAL_MAX_LABELS = 63
EVAL_COUNTERS = 32
EVAL_COUNTER_NAMES = ['frames', 'segs_gt', 'segs_pred', 'segs_pred_non_bg', 'multi', 'gt_labels', 'tp', 'pred_bg',
                      'true_bg', 'iou_den', 'iou_num', 'gt_labels_non_bg', 'frames_non_bg', 'tp_non_bg', 'steps',
                      'steps_non_bg', 'draw_hit', 'draw_hit_non_bg', 'mid_hit', 'mid_hit_non_bg', 'types',
                      'types_non_bg', 'other', 'levenshtein']


class SmmError(RuntimeError):
    pass


_lib = None


def load():
    """Load (once) and return the CDLL.  Raises SmmError when the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SmmError("libsmmdp.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'`; "
                       "there is no CPU fallback for the semi-Markov decode path" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name in SYMBOLS:
        getattr(lib, name)   # AttributeError if the ABI and the header drifted apart
    lib.smm_strerror.restype = ctypes.c_char_p
    lib.smm_strerror.argtypes = [ctypes.c_int]
    lib.smm_version.restype = ctypes.c_char_p
    lib.smm_workspace_bytes.restype = ctypes.c_size_t
    lib.smm_workspace_bytes.argtypes = [ctypes.POINTER(SmmShape), ctypes.c_void_p]
    lib.smm_dense_workspace_bytes.restype = ctypes.c_size_t
    lib.smm_dense_workspace_bytes.argtypes = [ctypes.c_int32] * 4
    lib.smm_eval_workspace_bytes.restype = ctypes.c_size_t
    lib.smm_eval_workspace_bytes.argtypes = [ctypes.POINTER(SmmEvalShape), ctypes.c_void_p]
    lib.smm_fit_workspace_bytes.restype = ctypes.c_size_t
    lib.smm_fit_workspace_bytes.argtypes = [ctypes.c_int32]
    lib.smm_fit_error_word_offset.restype = ctypes.c_size_t
    lib.smm_fit_error_word_offset.argtypes = [ctypes.c_int32]
    lib.smm_error_word_offset.restype = ctypes.c_size_t
    lib.smm_error_word_offset.argtypes = [ctypes.POINTER(SmmShape)]
    lib.smm_dp_timing_enable.restype = None
    lib.smm_dp_timing_enable.argtypes = [ctypes.c_int]
    lib.smm_dp_timing_read.restype = ctypes.c_int
    lib.smm_dp_timing_read.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int]
    lib.smm_dp_timing_read_tagged.restype = ctypes.c_int
    lib.smm_dp_timing_read_tagged.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32), ctypes.c_int]
    lib.smm_band_frame_ns.restype = ctypes.c_double
    lib.smm_band_frame_ns.argtypes = [ctypes.c_int]
    lib.smm_time_split_plan.restype = ctypes.c_int
    lib.smm_time_split_plan.argtypes = [ctypes.POINTER(SmmShape)] + [ctypes.c_void_p] * 4 + [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int]
    lib.smm_env_reload.restype = None
    lib.smm_env_reload.argtypes = []
    lib.smm_release_cached_plans.restype = ctypes.c_size_t
    lib.smm_release_cached_plans.argtypes = []
    lib.smm_cached_plan_bytes.restype = ctypes.c_size_t
    lib.smm_cached_plan_bytes.argtypes = []
    _lib = lib
    return lib


def reload_env():
    """The library reads its SMM_* tuning switches once, when it is first used; a process that changes one afterwards
    (tests, A/B scripts) calls this.  No-op while the library is not loaded."""
    if _lib is not None:
        _lib.smm_env_reload()


def check(status):
    if status != 0:
        lib = load()
        msg = lib.smm_strerror(status).decode()
        if status == -4:
            msg += " (hipError %d)" % lib.smm_last_hip_error()
        raise SmmError("libsmmdp: %s" % msg)


def host_cores():
    """CPU cores this process is entitled to: the cgroup's CFS quota when there is one, else the affinity mask
    (SMM_HOST_CORES overrides).  A container that SEES 256 logical CPUs but owns 16 is throttled for the rest of every
    100 ms scheduling period once default-sized thread pools have spun the quota away: ~90 ms stalls in the host glue
    of a 1 ms decode.  The CLI sizes torch's intra-op pool with this."""
    import os
    env = os.environ.get('SMM_HOST_CORES')
    if env:
        return max(1, int(env))
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    for quota_file, period_file in (('/sys/fs/cgroup/cpu.max', None),
                                    ('/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us')):
        try:
            if period_file is None:
                q, p = open(quota_file).read().split()[:2]
                if q == 'max':
                    continue
            else:
                q, p = open(quota_file).read(), open(period_file).read()
            if float(q) > 0:
                return max(1, min(n, int(float(q) / float(p) + 0.5)))
        except (OSError, ValueError):
            continue
    return n
