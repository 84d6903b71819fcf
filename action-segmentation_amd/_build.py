"""Builds libsmmdp.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsmmdp.so")
SOURCES = ["smm_api.hip", "smm_emission.hip", "smm_viterbi.hip", "smm_logz.hip", "smm_logz_bwd.hip", "smm_dense.hip",
           "smm_eval.hip", "smm_fit.hip"]
HEADERS = ["smm_device.h", "smm_launch.h", os.path.join("..", "..", "include", "smmdp.h")]
# -ffp-contract=off: every a+b in the DP must be ONE IEEE add (bit-exact twin of oracle/smm_oracle.c)
# unroll thresholds: the frame loops must unroll completely, or the register-resident rings become scratch arrays
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-mllvm", "-pragma-unroll-threshold=1048576", "-mllvm", "-unroll-threshold=1048576"]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB
