"""Builds libsmmdp.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

One object per translation unit (compiled in parallel, only the stale ones), then one link."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libsmmdp.so")
SOURCES = ["smm_api.hip", "smm_emission.hip", "smm_viterbi.hip", "smm_logz.hip", "smm_logz_bwd.hip", "smm_dense.hip",
           "smm_eval.hip", "smm_fit.hip", "smm_tables.hip"]
HEADERS = ["smm_device.h", "smm_launch.h", os.path.join("..", "..", "include", "smmdp.h")]
# -ffp-contract=off: every a+b in the DP must be ONE IEEE add (bit-exact twin of oracle/smm_oracle.c)
# unroll thresholds: the frame loops must unroll completely, or the register-resident rings become scratch arrays
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
         "-mllvm", "-pragma-unroll-threshold=1048576", "-mllvm", "-unroll-threshold=1048576"]


# per translation unit: the Viterbi kernels take fmax() as a bare v_max_f64 (smm_device.h: smm_fmax) -- no NaN ever enters
# the DP (inputs are finite or -inf and nothing subtracts infinities), and -fno-honor-nans says so to the compiler
UNIT_FLAGS = {"smm_viterbi.hip": ["-fno-honor-nans", "-DSMM_FMAX_BUILTIN"]}


def _extra_flags():
    # development aid only (SMM_DEV_FLAGS="-DSMM_DEV_R=16"): never set by build() callers in the repo
    return os.environ.get("SMM_DEV_FLAGS", "").split()


def _obj(src):
    return os.path.join(OBJ, os.path.splitext(src)[0] + ".o")


def _stale_obj(src, force):
    o = _obj(src)
    if force or not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in [src] + HEADERS)


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJ, exist_ok=True)
    todo = [s for s in SOURCES if _stale_obj(s, force)]

    def compile_one(src):
        cmd = [hipcc] + FLAGS + UNIT_FLAGS.get(src, []) + _extra_flags() + ["-c", "-o", _obj(src), os.path.join(CSRC, src)]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(todo)))) as pool:
        list(pool.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [_obj(s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB
