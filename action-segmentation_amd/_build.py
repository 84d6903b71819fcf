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
SOURCES = ["smm_api.hip", "smm_emission.hip", "smm_viterbi.hip", "smm_chunk.hip", "smm_logz.hip", "smm_logz_bwd.hip", "smm_dense.hip",
           "smm_eval.hip", "smm_fit.hip", "smm_tables.hip"]
HEADERS = ["smm_device.h", "smm_launch.h", os.path.join("..", "..", "include", "smmdp.h")]
# -ffp-contract=off: every a+b in the DP must be ONE IEEE add (bit-exact twin of oracle/smm_oracle.c)
# unroll thresholds: the frame loops must unroll completely, or the register-resident rings become scratch arrays
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
         "-mllvm", "-pragma-unroll-threshold=1048576", "-mllvm", "-unroll-threshold=1048576"]


# per translation unit: the Viterbi kernels take fmax() as a bare v_max_f64 (smm_device.h: smm_fmax): inputs are finite or
# -inf by contract and -fno-honor-nans says so to the compiler.  What still has to work when the contract is broken -- the
# error word for a NaN in the inputs -- is done on the bits (smm_nan_bits), never by a float compare that is meant to fail;
# the two one-sided tests that can subtract -inf from -inf (DOM, SPEC) take either compare result (smm_device.h)
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
    # objects of development variants (scripts/build_variants.sh: <unit>_<tag>.o) are debris once the library is rebuilt --
    # and everything under the package travels to the GPU box with every lease
    keep = {os.path.basename(_obj(s)) for s in SOURCES}
    for f in os.listdir(OBJ):
        if f.endswith(".o") and f not in keep:
            os.remove(os.path.join(OBJ, f))
    return LIB


def kernel_resources(lib=LIB):
    """{kernel name: {'private_segment_fixed_size', 'vgpr_count', 'vgpr_spill_count', 'group_segment_fixed_size'}} of every gfx950
    kernel inside the shared library: the code objects are cut out of the clang offload bundles of its .hip_fatbin section
    and their metadata notes read with llvm-readelf.  Used by the no-scratch check (tests/test_module_host.py): a DP kernel
    with a private segment -- a single spilled register -- is dispatched measurably slower beside another kernel
    (round 4: +0.3 ms on the critical launch of a split decode), so none may have one."""
    import re
    import struct
    import tempfile
    readelf = shutil.which("llvm-readelf") or "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(lib) or not os.path.exists(readelf):
        return {}                      # (nothing built / no tool: the caller skips)
    blob = open(lib, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out = {}                           # (stays empty for a compressed bundle -- magic CCOB, --offload-compress: the caller skips)
    pos = blob.find(magic)
    while pos >= 0:
        n = struct.unpack_from("<Q", blob, pos + len(magic))[0]
        q = pos + len(magic) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "amdgcn" not in triple or size == 0:
                continue
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(blob[pos + off:pos + off + size])
                f.flush()
                notes = subprocess.run([readelf, "--notes", f.name], capture_output=True, text=True).stdout
            for block in notes.split("- .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", block)
                if not name:
                    continue
                out[name.group(1)] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, block).group(1))
                                      for k in ("private_segment_fixed_size", "vgpr_count", "vgpr_spill_count", "group_segment_fixed_size")
                                      if re.search(r"\.%s:\s+(\d+)" % k, block)}
        pos = blob.find(magic, pos + len(magic))
    return out
