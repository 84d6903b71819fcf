"""ctypes front-end of oracle/smm_oracle.c (TEST INFRASTRUCTURE ONLY; see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libsmm_oracle.so')
_lib = None


def build(force=False):
    src = os.path.join(_HERE, 'smm_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE] + (['-B'] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.smm_oracle_set_threads(host_cores())      # (not OpenMP's default of one spinning thread per visible CPU)
    return _lib


def host_cores():
    """CPU cores this process may really use: the cgroup's CFS quota when there is one (a container that sees 256
    logical CPUs but is entitled to 16 gets THROTTLED for the rest of every 100 ms period once spinning OpenMP threads
    have burnt the quota -- stalls of ~90 ms in whatever runs next), else the affinity mask.  SMM_HOST_CORES overrides."""
    env = os.environ.get('SMM_HOST_CORES')
    if env:
        return max(1, int(env))
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    quota = None
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()[:2]             # cgroup v2: "max 100000" or "1600000 100000"
        if q != 'max':
            quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())    # cgroup v1
            p = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    return n


def set_threads(n=0):
    """OpenMP threads of the batch loops (n <= 0: as many as host_cores()).  Returns the number in use."""
    if n <= 0:
        n = host_cores()
    return int(lib().smm_oracle_set_threads(int(n)))


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ct))


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def emission(x, lengths, mu, inv_var, lognorm, cons=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    b, tmax, d = x.shape
    mu, inv_var, cons = _f64(mu), _f64(inv_var), _f64(cons)
    c = mu.shape[0]
    lengths = np.ascontiguousarray(lengths, dtype=np.int64)
    elp = np.empty((b, tmax, c), np.float64)
    lib().smm_oracle_emission(_p(x, ctypes.c_float), _p(lengths, ctypes.c_int64), _p(mu, ctypes.c_double),
                              _p(inv_var, ctypes.c_double), ctypes.c_double(lognorm), _p(cons, ctypes.c_double),
                              _p(elp, ctypes.c_double), b, tmax, d, c)
    return elp


def viterbi(elp, lengths, trans, init, len_scores, endpen=None, no_eos=False):
    """-> (spans b x (Tmax+1) int64 local ids with EOS = C, v b).  len_scores: K x C, clipped to Tmax rows here.
    ``no_eos``: add_eos=False of the reference (no EOS label; the last frame's label only emits)."""
    elp = _f64(elp)
    b, tmax, c = elp.shape
    len_scores = _f64(len_scores)[:tmax]
    kp = len_scores.shape[0]
    lengths = np.ascontiguousarray(lengths, dtype=np.int64)
    trans, init, endpen = _f64(trans), _f64(init), _f64(endpen)
    spans = np.empty((b, tmax + 1), np.int64)
    v = np.empty(b, np.float64)
    rc = lib().smm_oracle_viterbi_ex(_p(elp, ctypes.c_double), _p(lengths, ctypes.c_int64), _p(trans, ctypes.c_double),
                                     _p(init, ctypes.c_double), _p(len_scores, ctypes.c_double),
                                     _p(endpen, ctypes.c_double), b, tmax, c, kp, int(bool(no_eos)),
                                     _p(spans, ctypes.c_int64), _p(v, ctypes.c_double))
    assert rc == 0, rc
    return spans, v


def logz(elp, lengths, trans, init, len_scores, endpen=None, grad=False, upstream=None):
    elp = _f64(elp)
    b, tmax, c = elp.shape
    len_scores = _f64(len_scores)[:tmax]
    kp = len_scores.shape[0]
    lengths = np.ascontiguousarray(lengths, dtype=np.int64)
    trans, init, endpen, upstream = _f64(trans), _f64(init), _f64(endpen), _f64(upstream)
    z = np.empty(b, np.float64)
    g = dict(elp=np.empty((b, tmax, c)), trans=np.empty((c, c)), init=np.empty(c), len=np.empty((kp, c))) if grad else {}
    rc = lib().smm_oracle_logz(_p(elp, ctypes.c_double), _p(lengths, ctypes.c_int64), _p(trans, ctypes.c_double),
                               _p(init, ctypes.c_double), _p(len_scores, ctypes.c_double), _p(endpen, ctypes.c_double),
                               _p(upstream, ctypes.c_double), b, tmax, c, kp, _p(z, ctypes.c_double),
                               _p(g.get('elp'), ctypes.c_double), _p(g.get('trans'), ctypes.c_double),
                               _p(g.get('init'), ctypes.c_double), _p(g.get('len'), ctypes.c_double))
    assert rc == 0, rc
    return (z, g) if grad else z


def endpen_from_allowed_ends(allowed_ends_per_instance, b, c):
    """0 for allowed end states, -1e9 otherwise (semimarkov_modules.py:462-471); None = unrestricted."""
    if allowed_ends_per_instance is None:
        return None
    ep = np.full((b, c), -1e9)
    for i, ends in enumerate(allowed_ends_per_instance):
        ep[i, list(ends)] = 0.0
    return ep
