"""Tensor-level wrappers of the libsmmdp C ABI (include/smmdp.h).

Every function takes CUDA(=HIP) tensors for bulk data and host sequences for the per-video / per-group
metadata, enqueues on torch's current stream and returns CUDA tensors.  Nothing here computes on the CPU.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import SmmShape


class Batch:
    """Metadata of one ragged decode batch (host side).

    lengths        frames per video
    frame_offset   first frame of each video on the packed frame axis (default: padded, i * t_max)
    group          parameter group per video (default 0)
    kp             per-video min(K, Tmax of its reference batch) (reference semimarkov_modules.py:450-452)
    n_states       states per group
    no_eos         add_eos=False of the reference: no EOS label, the last frame's label only emits
    no_time_split  the tables carry hard masks (ordering constraints): long videos are decoded in one piece
                   (include/smmdp.h: SMM_SHAPE_NO_TIME_SPLIT)
    """

    def __init__(self, lengths, n_states, k_rows, c_max=None, frame_offset=None, group=None, kp=None, d=0,
                 t_max=None, total_frames=None, no_eos=False, no_time_split=False):
        self.lengths = np.ascontiguousarray(np.asarray(lengths, dtype=np.int64).reshape(-1))
        self.b = int(self.lengths.shape[0])
        self.n_states = np.ascontiguousarray(np.asarray(n_states, dtype=np.int32).reshape(-1))
        self.n_groups = int(self.n_states.shape[0])
        self.c_max = int(c_max if c_max is not None else self.n_states.max())
        self.k_rows = int(k_rows)
        self.t_max = int(t_max if t_max is not None else self.lengths.max())
        if frame_offset is None:
            frame_offset = np.arange(self.b, dtype=np.int64) * self.t_max
        self.frame_offset = np.ascontiguousarray(np.asarray(frame_offset, dtype=np.int64).reshape(-1))
        self.group = None if group is None else np.ascontiguousarray(np.asarray(group, dtype=np.int32).reshape(-1))
        self.kp = None if kp is None else np.ascontiguousarray(np.asarray(kp, dtype=np.int32).reshape(-1))
        if total_frames is None:
            total_frames = int((self.frame_offset + self.lengths).max())
        self.total_frames = int(total_frames)
        self.d = int(d)
        self.no_eos = bool(no_eos)     # add_eos=False of the reference (include/smmdp.h: SMM_SHAPE_NO_EOS)
        self.shape = SmmShape(self.b, self.d, self.n_groups, self.c_max, self.k_rows, self.t_max,
                              (_lib.SHAPE_NO_EOS if self.no_eos else 0) | (_lib.SHAPE_NO_TIME_SPLIT if no_time_split else 0),
                              self.total_frames)

    def workspace_bytes(self):
        n = _lib.load().smm_workspace_bytes(ctypes.byref(self.shape), self.lengths.ctypes.data)
        if n == 0:
            raise _lib.SmmError("libsmmdp: invalid batch shape")
        return n

    def host_ptrs(self):
        return (self.lengths.ctypes.data, self.frame_offset.ctypes.data,
                None if self.group is None else self.group.ctypes.data,
                None if self.kp is None else self.kp.ctypes.data, self.n_states.ctypes.data)


def _dev(t, dtype, name):
    if t is None:
        return None
    if not t.is_cuda and not (name in ('labels', 'spans') and t.is_pinned()):
        raise _lib.SmmError("libsmmdp: %s must be a CUDA/HIP tensor (there is no CPU path)" % name)
    if t.dtype != dtype:
        raise TypeError("%s: expected %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return ctypes.c_void_p(t.data_ptr())


def _raw_stream():
    """Handle of torch's current stream on the current device, by the C-level getters (torch.cuda.current_stream() builds a
    Stream object and resolves the device through three python layers: ~10 us, twice per call, of a 150 us viterbi())."""
    get = getattr(torch._C, '_cuda_getCurrentRawStream', None)
    if get is None:                                    # (a torch build without the C-level getter)
        return torch.cuda.current_stream().cuda_stream
    return get(torch._C._cuda_getDevice())


def _stream():
    return ctypes.c_void_p(_raw_stream())


_ws_cache = {}


def workspace(nbytes, device):
    """Grow-only scratch buffer per (device, stream).  (Caller-owned from the library's point of view.)"""
    key = (device.index, _raw_stream())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


_host_labels = {}


def _labels_on_host(batch, device):
    """Pinned host buffer the DP kernel writes the frame labels into directly (host-pinned memory is mapped into the
    GPU's address space on ROCm; the stores go over PCIe while the kernel is still decoding other videos, so there is
    no separate device -> host copy at the end).  Valid after the stream has been synchronised, until the next call."""
    buf = _host_labels.get(device.index)
    if buf is None or buf.numel() < batch.total_frames:
        buf = torch.empty(int(batch.total_frames * 1.25) + 16, dtype=torch.int64, pin_memory=True)
        _host_labels[device.index] = buf
    out = buf[:batch.total_frames]
    if int(batch.lengths.sum()) != batch.total_frames:
        out.fill_(-1)                                  # frames no video covers (padded layouts) keep the -1 filler
    return out


_host_small = {}


def _pinned_small(kind, device, numel, dtype):
    """Small pinned host buffers the kernels (spans) or an async copy (error words) write into: the reference's call
    pattern -- one viterbi() per batch of five videos -- is host latency, and every blocking device -> host copy is ~50 us
    of it.  Valid after the stream has been synchronised, until the next call."""
    key = (kind, device.index, dtype)
    buf = _host_small.get(key)
    if buf is None or buf.numel() < numel:
        buf = torch.empty(int(numel * 1.5) + 16, dtype=dtype, pin_memory=True)
        _host_small[key] = buf
    return buf[:numel]


def pinned_labels(kind, device, numel):
    """A pinned int64 label buffer per (kind, device): the DP kernel's label stores land in it (see _labels_on_host); valid after
    the launch's stream has been synchronised, until the next launch that asks for the same kind."""
    return _pinned_small(('labels',) + tuple(kind), device, numel, torch.int64)


_label_leases = {}
LABEL_LEASES = 3      # pinned label buffers per device that can be out on lease at once


def _storage_users(t):
    """How many tensors (and numpy arrays made from them) share ``t``'s storage, ``t`` included; None if this torch build
    cannot tell (then nothing is ever leased)."""
    fn = getattr(torch._C, '_storage_Use_Count', None)
    return None if fn is None else fn(t.untyped_storage()._cdata)


def lease_host_labels(batch, device):
    """A pinned int64 [total_frames] label buffer that belongs to the CALLER for as long as the returned tensor -- or
    anything that shares its storage, such as the numpy arrays of ``SemiMarkovModel.predict`` -- is alive; pass it as
    ``labels_out``.  The buffer is free again when the last of them has died (the storage's use count tells: a numpy array
    made from a tensor keeps a NEW tensor object alive, not the one it was made from, so a weak reference would not), i.e. a
    caller that drops its result before the next decode keeps reusing one buffer and never copies 8 bytes per frame out of a
    staging buffer (cfg3: 20 MB).  At most LABEL_LEASES buffers per device are out at once: None when all are (the caller
    then decodes into the shared staging buffer and copies, as before)."""
    pool = _label_leases.setdefault(device.index, [])
    n = batch.total_frames
    entry = None
    for e in pool:
        if _storage_users(e[0]) == e[1]:               # nobody but the pool holds it
            entry = e
            if e[0].numel() >= n:
                break
    if entry is None or entry[0].numel() < n:
        if entry is None and len(pool) >= LABEL_LEASES:
            return None
        buf = torch.empty(int(n * 1.25) + 16, dtype=torch.int64, pin_memory=True)
        users = _storage_users(buf)
        if users is None:
            return None
        if entry is None:
            entry = [buf, users]
            pool.append(entry)
        else:
            entry[0], entry[1] = buf, users
    out = entry[0][:n]
    if int(batch.lengths.sum()) != n:
        out.fill_(-1)                                  # frames no video covers (padded layouts) keep the -1 filler
    return out


def _outputs(batch, device, want_spans, want_labels, labels_on_host=False, labels_out=None, spans_on_host=False, host_slot=0):
    if want_spans and spans_on_host:
        spans = _pinned_small(('spans', host_slot), device, batch.b * (batch.t_max + 1), torch.int64).view(batch.b, batch.t_max + 1)
    else:
        spans = torch.empty((batch.b, batch.t_max + 1), dtype=torch.int64, device=device) if want_spans else None
    if labels_out is not None:
        labels = labels_out
    elif want_labels and labels_on_host:
        labels = _labels_on_host(batch, device)
    else:
        labels = torch.full((batch.total_frames,), -1, dtype=torch.int64, device=device) if want_labels else None
    best = torch.empty(batch.b, dtype=torch.float64, device=device)
    n_segs = torch.empty(batch.b, dtype=torch.int32, device=device)
    return spans, labels, best, n_segs


def emission(batch, x, w, cst, inv_var, cons=None, want64=True, want32=False, out64=None):
    """x fp32 [total_frames, d] -> elp fp64 and/or fp32 [total_frames, c_max].  (smm_emission_f64)
    ``out64``: write into this [total_frames, c_max] tensor (sub-batches of one packed frame axis share it)."""
    lib = _lib.load()
    dev = x.device
    elp64 = out64 if out64 is not None else (
        torch.empty((batch.total_frames, batch.c_max), dtype=torch.float64, device=dev) if want64 else None)
    elp32 = torch.zeros((batch.total_frames, batch.c_max), dtype=torch.float32, device=dev) if want32 else None
    ws = workspace(batch.workspace_bytes(), dev)
    ln, fo, gr, _, ns = batch.host_ptrs()
    _lib.check(lib.smm_emission_f64(
        ctypes.byref(batch.shape), ctypes.c_void_p(ln), ctypes.c_void_p(fo), ctypes.c_void_p(gr), ctypes.c_void_p(ns),
        _dev(x, torch.float32, 'x'), _dev(w, torch.float64, 'w'), _dev(cst, torch.float64, 'cst'),
        _dev(inv_var, torch.float64, 'inv_var'), _dev(cons, torch.float32, 'cons'),
        _dev(elp64, torch.float64, 'elp64'), _dev(elp32, torch.float32, 'elp32'),
        ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()), _stream()))
    return elp64, elp32


def emission_bwd(batch, x, g_elp, ws=None):
    """Chain rule through the emission scorer (smm_emission_bwd_f64): g_elp fp64 [total_frames, c_max] ->
    (g_w [n_groups, d, c_max] (a transposed view of the kernel's class-major output), g_cst [n_groups, c_max],
    g_inv_var [d]), all fp64."""
    lib = _lib.load()
    dev = x.device
    f64 = torch.float64
    g, cm, d = batch.n_groups, batch.c_max, int(x.size(1))
    if d != batch.d or x.size(0) < batch.total_frames or tuple(g_elp.shape) != (batch.total_frames, cm):
        raise ValueError("emission_bwd: x [>= total_frames, batch.d] and g_elp [total_frames, c_max] expected")
    g_w = torch.empty((g, cm, d), dtype=f64, device=dev)
    g_cst = torch.empty((g, cm), dtype=f64, device=dev)
    g_iv = torch.empty(d, dtype=f64, device=dev)
    if ws is None:
        ws = workspace(batch.workspace_bytes(), dev)
    ln, fo, gr, _, ns = batch.host_ptrs()
    _lib.check(lib.smm_emission_bwd_f64(
        ctypes.byref(batch.shape), ctypes.c_void_p(ln), ctypes.c_void_p(fo), ctypes.c_void_p(gr), ctypes.c_void_p(ns),
        _dev(x, torch.float32, 'x'), _dev(g_elp, f64, 'g_elp'), _dev(g_w, f64, 'g_w'), _dev(g_cst, f64, 'g_cst'),
        _dev(g_iv, f64, 'g_inv_var'), ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()), _stream()))
    return g_w.transpose(1, 2), g_cst, g_iv


class TablesMeta:
    """Which class sets a launch's parameter groups are (device index tensors, built once per PackedCorpus):
    classes / merged int64 [g, c_max], n_states int32 [g]; plus the scalars of smm_tables_shape."""

    def __init__(self, classes, merged, n_states, n_classes, d, k_rows, allow_self_transitions):
        self.classes, self.merged, self.n_states = classes, merged, n_states
        g, cm = classes.shape
        self.g, self.cm, self.n, self.d, self.k_rows = int(g), int(cm), int(n_classes), int(d), int(k_rows)
        self.shape = _lib.SmmTablesShape(self.n, self.d, self.g, self.cm, self.k_rows, 1 if allow_self_transitions else 0)


def _u8(t, name):
    if t is None:
        return None
    if not t.is_cuda or not t.is_contiguous() or t.element_size() != 1:
        raise ValueError("%s: a contiguous 1-byte device tensor is expected" % name)
    return ctypes.c_void_p(t.data_ptr())


def factor_tables(meta, init_logits, transition_logits, poisson_log_rates, gaussian_means, gaussian_cov,
                  init_constraints=None, transition_constraints=None):
    """fp32 parameters -> fp64 tables of every group (smm_factor_tables_f64):
    dict(trans [g,cm,cm], init [g,cm], len [g,K,cm], w [g,d,cm], cst [g,cm], inv_var [d])."""
    lib = _lib.load()
    dev, f64, f32 = init_logits.device, torch.float64, torch.float32
    g, cm, d, k = meta.g, meta.cm, meta.d, meta.k_rows
    if tuple(gaussian_means.shape) != (meta.n, d) or tuple(gaussian_cov.shape) != (d, d) or \
            tuple(transition_logits.shape) != (meta.n, meta.n):
        raise ValueError("factor_tables: parameter shapes do not match the TablesMeta")
    t = dict(trans=torch.empty((g, cm, cm), dtype=f64, device=dev), init=torch.empty((g, cm), dtype=f64, device=dev),
             len=torch.empty((g, k, cm), dtype=f64, device=dev), w=torch.empty((g, d, cm), dtype=f64, device=dev),
             cst=torch.empty((g, cm), dtype=f64, device=dev), inv_var=torch.empty(d, dtype=f64, device=dev))
    _lib.check(lib.smm_factor_tables_f64(
        ctypes.byref(meta.shape), _dev(init_logits, f32, 'init_logits'), _dev(transition_logits, f32, 'transition_logits'),
        _dev(poisson_log_rates, f32, 'poisson_log_rates'), _dev(gaussian_means, f32, 'gaussian_means'),
        _dev(gaussian_cov, f32, 'gaussian_cov'), _u8(init_constraints, 'init_constraints'),
        _u8(transition_constraints, 'transition_constraints'), _dev(meta.classes, torch.int64, 'classes'),
        _dev(meta.merged, torch.int64, 'merged'), _dev(meta.n_states, torch.int32, 'n_states'),
        _dev(t['trans'], f64, 'trans'), _dev(t['init'], f64, 'init'), _dev(t['len'], f64, 'len'), _dev(t['w'], f64, 'w'),
        _dev(t['cst'], f64, 'cst'), _dev(t['inv_var'], f64, 'inv_var'), _stream()))
    return t


def factor_tables_bwd(meta, poisson_log_rates, gaussian_means, gaussian_cov, trans, init, g_trans, g_init, g_len,
                      g_w_class_major, g_cst, init_constraints=None, transition_constraints=None):
    """Gradients of the tables -> fp64 gradients of (init_logits [n], transition_logits [n,n], poisson_log_rates [n],
    gaussian_means [n,d])  (smm_factor_tables_bwd_f64).  g_w_class_major: [g, c_max, d]."""
    lib = _lib.load()
    dev, f64, f32 = trans.device, torch.float64, torch.float32
    n, d = meta.n, meta.d
    # (one flat buffer, four views: the caller converts the gradients to the parameters' dtype in ONE launch -- flat_of)
    flat = torch.empty(2 * n + n * n + n * d, dtype=f64, device=dev)
    out = (flat[:n], flat[n:n + n * n].view(n, n), flat[n + n * n:2 * n + n * n], flat[2 * n + n * n:].view(n, d))
    _lib.check(lib.smm_factor_tables_bwd_f64(
        ctypes.byref(meta.shape), _dev(poisson_log_rates, f32, 'poisson_log_rates'), _dev(gaussian_means, f32, 'gaussian_means'),
        _dev(gaussian_cov, f32, 'gaussian_cov'), _u8(init_constraints, 'init_constraints'),
        _u8(transition_constraints, 'transition_constraints'), _dev(meta.classes, torch.int64, 'classes'),
        _dev(meta.merged, torch.int64, 'merged'), _dev(meta.n_states, torch.int32, 'n_states'),
        _dev(trans, f64, 'trans'), _dev(init, f64, 'init'), _dev(g_trans, f64, 'g_trans'), _dev(g_init, f64, 'g_init'),
        _dev(g_len, f64, 'g_len'), _dev(g_w_class_major, f64, 'g_w'), _dev(g_cst, f64, 'g_cst'),
        _dev(out[0], f64, 'g_init_logits'), _dev(out[1], f64, 'g_transition_logits'), _dev(out[2], f64, 'g_poisson_log_rates'),
        _dev(out[3], f64, 'g_gaussian_means'), _stream()))
    return out + (flat,)


def viterbi(batch, elp, trans, init, len_scores, endpen=None, class_map=None, want_spans=True, want_labels=True,
            labels_on_host=False, labels_out=None):
    """Viterbi on emission scores.  elp fp64 (smm_viterbi_f64) or fp32 (smm_viterbi_f32, tables fp32 too).
    ``labels_on_host``: the kernel writes the frame labels straight into pinned host memory (see _labels_on_host);
    synchronise the stream before reading them."""
    lib = _lib.load()
    dev = elp.device
    dt = elp.dtype
    fn = lib.smm_viterbi_f64 if dt == torch.float64 else lib.smm_viterbi_f32
    spans, labels, best, n_segs = _outputs(batch, dev, want_spans, want_labels, labels_on_host, labels_out)
    ws = workspace(batch.workspace_bytes(), dev)
    ln, fo, gr, kp, ns = batch.host_ptrs()
    _lib.check(fn(
        ctypes.byref(batch.shape), ctypes.c_void_p(ln), ctypes.c_void_p(fo), ctypes.c_void_p(gr), ctypes.c_void_p(kp),
        ctypes.c_void_p(ns), _dev(elp, dt, 'elp'), _dev(trans, dt, 'trans'), _dev(init, dt, 'init'),
        _dev(len_scores, dt, 'len_scores'), _dev(endpen, dt, 'endpen'), _dev(class_map, torch.int64, 'class_map'),
        _dev(spans, torch.int64, 'spans'), _dev(labels, torch.int64, 'labels'), _dev(best, torch.float64, 'best'),
        _dev(n_segs, torch.int32, 'n_segs'), ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()), _stream()))
    return dict(spans=spans, labels=labels, best=best, n_segs=n_segs, _err=_err_copy(batch, ws))


def decode(batch, x, w, cst, inv_var, trans, init, len_scores, cons=None, endpen=None, class_map=None,
           want_spans=True, want_labels=True, want_elp=False, labels_on_host=False, labels_out=None, spans_on_host=False,
           host_slot=0):
    """Features -> spans / labels in one call (smm_decode_f32): emission kernel + DP kernel on the current stream.
    ``labels_on_host`` / ``labels_out``: see ``viterbi``.  ``spans_on_host``: the kernel writes the span encoding into
    pinned host memory and the error words follow by an asynchronous copy (small batches: the reference's per-batch call
    pattern); both are valid once the stream has been synchronised, until the next such call with the same ``host_slot``
    (a caller that keeps two decodes in flight alternates two slots)."""
    lib = _lib.load()
    dev = x.device
    spans, labels, best, n_segs = _outputs(batch, dev, want_spans, want_labels, labels_on_host, labels_out, spans_on_host,
                                           host_slot)
    elp32 = torch.zeros((batch.total_frames, batch.c_max), dtype=torch.float32, device=dev) if want_elp else None
    ws = workspace(batch.workspace_bytes(), dev)
    ln, fo, gr, kp, ns = batch.host_ptrs()
    f64 = torch.float64
    _lib.check(lib.smm_decode_f32(
        ctypes.byref(batch.shape), ctypes.c_void_p(ln), ctypes.c_void_p(fo), ctypes.c_void_p(gr), ctypes.c_void_p(kp),
        ctypes.c_void_p(ns), _dev(x, torch.float32, 'x'), _dev(w, f64, 'w'), _dev(cst, f64, 'cst'),
        _dev(inv_var, f64, 'inv_var'), _dev(cons, torch.float32, 'cons'), _dev(trans, f64, 'trans'),
        _dev(init, f64, 'init'), _dev(len_scores, f64, 'len_scores'), _dev(endpen, f64, 'endpen'),
        _dev(class_map, torch.int64, 'class_map'), _dev(spans, torch.int64, 'spans'),
        _dev(labels, torch.int64, 'labels'), _dev(best, f64, 'best'), _dev(n_segs, torch.int32, 'n_segs'),
        _dev(elp32, torch.float32, 'elp32'), ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()), _stream()))
    if spans_on_host:
        err = _pinned_small(('err', host_slot), dev, 8, torch.int32)
        err.copy_(_err_view(batch, ws), non_blocking=True)
    else:
        err = _err_copy(batch, ws)
    return dict(spans=spans, labels=labels, best=best, n_segs=n_segs, elp=elp32, _err=err)


class ResidentDecode:
    """``decode`` for a packed corpus that is decoded again and again (the training loop's per-epoch decode, bench.py's step):
    the call's FIXED arguments -- shape, the five metadata arrays, features, tables, constraints, end penalties, class map --
    are validated and turned into ctypes values once; a call then marshals three pointers (labels, workspace, stream) instead
    of eighteen (round 4 measured 25-35 us of python per ``decode`` call in front of the library's own 10-36 us).  Frame
    labels only (no spans, no emission copy); ``best`` / ``n_segs`` are the object's own buffers, overwritten by every call.
    The tensors are kept referenced; ``key`` lets the owner notice that the tables were rebuilt."""

    def __init__(self, batch, x, w, cst, inv_var, trans, init, len_scores, cons=None, endpen=None, class_map=None):
        self.lib = _lib.load()
        self.batch = batch
        dev = x.device
        f64 = torch.float64
        self.keep = (x, w, cst, inv_var, trans, init, len_scores, cons, endpen, class_map)
        self.key = tuple(None if t is None else (t.data_ptr(), t._version) for t in self.keep)
        self.best = torch.empty(batch.b, dtype=f64, device=dev)
        self.n_segs = torch.empty(batch.b, dtype=torch.int32, device=dev)
        ln, fo, gr, kp, ns = batch.host_ptrs()
        self.head = (ctypes.byref(batch.shape), ctypes.c_void_p(ln), ctypes.c_void_p(fo), ctypes.c_void_p(gr), ctypes.c_void_p(kp),
                     ctypes.c_void_p(ns), _dev(x, torch.float32, 'x'), _dev(w, f64, 'w'), _dev(cst, f64, 'cst'),
                     _dev(inv_var, f64, 'inv_var'), _dev(cons, torch.float32, 'cons'), _dev(trans, f64, 'trans'),
                     _dev(init, f64, 'init'), _dev(len_scores, f64, 'len_scores'), _dev(endpen, f64, 'endpen'),
                     _dev(class_map, torch.int64, 'class_map'), None)
        self.tail = (_dev(self.best, f64, 'best'), _dev(self.n_segs, torch.int32, 'n_segs'), None)
        self.ws_bytes = batch.workspace_bytes()
        self.dev = dev

    def __call__(self, labels_on_host=False, labels_out=None):
        batch = self.batch
        if labels_out is not None:
            labels = labels_out
        elif labels_on_host:
            labels = _labels_on_host(batch, self.dev)
        else:
            labels = torch.full((batch.total_frames,), -1, dtype=torch.int64, device=self.dev)
        ws = workspace(self.ws_bytes, self.dev)
        _lib.check(self.lib.smm_decode_f32(*self.head, _dev(labels, torch.int64, 'labels'), *self.tail,
                                           ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()), _stream()))
        return dict(spans=None, labels=labels, best=self.best, n_segs=self.n_segs, elp=None, _err=_err_copy(batch, ws))


def _shape_with(batch, extra_flags):
    """The batch's smm_shape with extra SMM_SHAPE_* bits for one call."""
    if not extra_flags:
        return batch.shape
    s = batch.shape
    return SmmShape(s.b, s.d, s.n_groups, s.c_max, s.k_rows, s.t_max, s.flags | extra_flags, s.total_frames)


def logz(batch, elp, trans, init, len_scores, endpen=None, ws=None, with_backward=False):
    """Log-partition per video (smm_logz_f64).  elp fp64 [total_frames, c_max] -> logZ fp64 [b].
    ``ws``: a private uint8 workspace tensor (keep it for ``logz_bwd``); default: the shared per-stream one.
    ``with_backward``: run the time-reversed recursion in the same launch (SMM_SHAPE_LOGZ_BOTH); pass the same to
    ``logz_bwd``."""
    lib = _lib.load()
    shape = _shape_with(batch, _lib.SHAPE_LOGZ_BOTH if with_backward else 0)
    dev = elp.device
    f64 = torch.float64
    out = torch.empty(batch.b, dtype=f64, device=dev)
    if ws is None:
        ws = workspace(batch.workspace_bytes(), dev)
    ln, fo, gr, kp, ns = batch.host_ptrs()
    _lib.check(lib.smm_logz_f64(
        ctypes.byref(shape), ctypes.c_void_p(ln), ctypes.c_void_p(fo), ctypes.c_void_p(gr), ctypes.c_void_p(kp),
        ctypes.c_void_p(ns), _dev(elp, f64, 'elp'), _dev(trans, f64, 'trans'), _dev(init, f64, 'init'),
        _dev(len_scores, f64, 'len_scores'), _dev(endpen, f64, 'endpen'), _dev(out, f64, 'logz'),
        ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()), _stream()))
    return out


def logz_bwd(batch, elp, trans, init, len_scores, logz_val, grad_logz=None, endpen=None, ws=None, with_backward=False):
    """Gradient of sum_i grad_logz[i] * logZ_i (smm_logz_bwd_f64).  Must follow ``logz`` for the same batch with the
    same workspace (same device + stream).  -> dict(elp [total_frames, c_max], trans, init, len) fp64.
    ``with_backward``: ``logz`` was called with it (the backward messages are already in the workspace)."""
    lib = _lib.load()
    shape = _shape_with(batch, _lib.SHAPE_LOGZ_BOTH if with_backward else 0)
    dev = elp.device
    f64 = torch.float64
    g = dict(elp=torch.empty_like(elp), trans=torch.empty_like(trans), init=torch.empty_like(init),
             len=torch.empty_like(len_scores))
    if ws is None:
        ws = workspace(batch.workspace_bytes(), dev)
    ln, fo, gr, kp, ns = batch.host_ptrs()
    _lib.check(lib.smm_logz_bwd_f64(
        ctypes.byref(shape), ctypes.c_void_p(ln), ctypes.c_void_p(fo), ctypes.c_void_p(gr), ctypes.c_void_p(kp),
        ctypes.c_void_p(ns), _dev(elp, f64, 'elp'), _dev(trans, f64, 'trans'), _dev(init, f64, 'init'),
        _dev(len_scores, f64, 'len_scores'), _dev(endpen, f64, 'endpen'), _dev(logz_val, f64, 'logz'),
        _dev(grad_logz, f64, 'grad_logz'), _dev(g['elp'], f64, 'g_elp'), _dev(g['trans'], f64, 'g_trans'),
        _dev(g['init'], f64, 'g_init'), _dev(g['len'], f64, 'g_len'),
        ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()), _stream()))
    return g


_pinned = {}


def to_host(t):
    """Device -> host copy through a cached pinned staging buffer (pageable copies are staged by the runtime and
    run at a fraction of the PCIe rate).  Returns a CPU tensor that is valid until the next call for the same dtype."""
    key = (t.dtype, t.device.index)
    buf = _pinned.get(key)
    if buf is None or buf.numel() < t.numel():
        buf = torch.empty(int(t.numel() * 1.25) + 16, dtype=t.dtype, pin_memory=True)
        _pinned[key] = buf
    out = buf[:t.numel()].view(t.shape)
    out.copy_(t, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return out


def _err_view(batch, ws):
    """int32 view of the error word inside the workspace a launch was given (the kernels of that launch write it)."""
    off = _lib.load().smm_error_word_offset(ctypes.byref(batch.shape))
    return ws[off:off + 32].view(torch.int32)      # [error, 0, band-0 sources pushed, band-blocks evaluated, videos split in time, of those decoded again in one piece, why (bits), one-class-run ties resolved]


def _err_copy(batch, ws):
    """The launch's error words COPIED out of the workspace (32 bytes, stream-ordered behind the kernels, graph-capturable):
    the per-stream workspace is shared, and the next launch on the stream re-stages it and zeroes these words."""
    # (an elementwise kernel, not clone(): torch copies device -> device with hipMemcpyAsync, and memset / memcpy nodes are
    # what replayed wrongly from a captured hipGraph on ROCm 7.2 -- tests/test_gpu_graph.py; kernel nodes replay correctly)
    return torch.add(_err_view(batch, ws), 0)


def error_flag(batch, out=None, ws=None):
    """The kernels' error word of a decode (synchronises): non-zero means a NaN (or inf - inf) reached the DP and the
    decode of that video stopped early.  ``out``: the dict the launch returned (its ``_err`` entry views the workspace
    that launch wrote to -- the per-stream cache may have been regrown or switched since); without it the current
    (device, stream) workspace is read, which is only right directly after the launch."""
    return error_words(batch, out, ws)[0]


def error_words(batch, out=None, ws=None):
    """[error word, 0, sources pushed into band 0, delayed band-blocks evaluated (words 2 and 3: diagnostics of the Viterbi
    kernel's BAND mode, smm_viterbi.hip: DOM and the band skip test), videos decoded as several units along the time axis, of
    those the ones that were decoded again in one piece, why (a bit mask), one-class-run ties the stitch resolved (words 4..7:
    csrc/smm_chunk.hip)] of a decode (synchronises).
    (Words 1 and 2 counted gang time-outs in rounds 1-3.)"""
    if out is not None and out.get('_err') is not None:
        return [int(v) for v in out['_err'].tolist()]
    if ws is None:
        ws = workspace(batch.workspace_bytes(), torch.device('cuda', torch.cuda.current_device()))
    return [int(v) for v in _err_view(batch, ws).tolist()]


def check_decoded(batch, out=None):
    """Call after the decode's outputs have reached the host (the stream is idle then): raises SmmError when the DP
    kernel flagged the run: a NaN / inf - inf reached the DP of some video and its decode stopped early."""
    flag = error_words(batch, out)[0]
    if flag != 0:
        raise _lib.SmmError("libsmmdp: NaN (or inf - inf) in the DP inputs; decode stopped early (error word %d)" % flag)


# ------------------------------------------------------------------------------------------------ evaluation counters
class EvalBatch:
    """Host metadata of one evaluation call: videos on the packed frame axis, their task and index inside the task."""

    def __init__(self, lengths, frame_offset, group, n_groups, c_max, n_labels, gt_width=1, video_key=None,
                 total_frames=None):
        self.lengths = np.ascontiguousarray(np.asarray(lengths, dtype=np.int64).reshape(-1))
        self.b = int(self.lengths.shape[0])
        self.frame_offset = np.ascontiguousarray(np.asarray(frame_offset, dtype=np.int64).reshape(-1))
        self.group = np.ascontiguousarray(np.asarray(group, dtype=np.int32).reshape(-1))
        self.video_key = None if video_key is None else np.ascontiguousarray(np.asarray(video_key, dtype=np.int32).reshape(-1))
        self.n_groups, self.c_max, self.n_labels, self.gt_width = int(n_groups), int(c_max), int(n_labels), int(gt_width)
        if self.c_max > _lib.EVAL_MAX_LABELS:
            raise _lib.SmmError("libsmmdp: a task with %d labels exceeds SMM_EVAL_MAX_LABELS" % self.c_max)
        if total_frames is None:
            total_frames = int((self.frame_offset + self.lengths).max())
        self.total_frames = int(total_frames)
        self.shape = _lib.SmmEvalShape(self.b, self.n_groups, self.c_max, self.n_labels, self.gt_width,
                                       int(self.lengths.max()), self.total_frames)

    def workspace_bytes(self):
        n = _lib.load().smm_eval_workspace_bytes(ctypes.byref(self.shape), self.lengths.ctypes.data)
        if n == 0:
            raise _lib.SmmError("libsmmdp: invalid evaluation batch shape")
        return n


def _eval_check(eb, pred, gt):
    if pred.numel() < eb.total_frames or gt.numel() < eb.total_frames * eb.gt_width:
        raise ValueError("pred / gt shorter than the packed frame axis")


def eval_confusion(eb, pred, gt, local_of):
    """(first gt label, predicted label) frame counts per task: int64 [n_groups, c_max+1, c_max+1].  (smm_eval_confusion_i64)"""
    lib = _lib.load()
    _eval_check(eb, pred, gt)
    dev = pred.device
    conf = torch.empty((eb.n_groups, eb.c_max + 1, eb.c_max + 1), dtype=torch.int64, device=dev)
    ws = workspace(eb.workspace_bytes(), dev)
    _lib.check(lib.smm_eval_confusion_i64(
        ctypes.byref(eb.shape), ctypes.c_void_p(eb.lengths.ctypes.data), ctypes.c_void_p(eb.frame_offset.ctypes.data),
        ctypes.c_void_p(eb.group.ctypes.data), _dev(pred, torch.int64, 'pred'), _dev(gt, torch.int64, 'gt'),
        _dev(local_of, torch.int32, 'local_of'), _dev(conf, torch.int64, 'confusion'),
        ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()), _stream()))
    return conf


def eval_videos(eb, pred, gt, local_of, cluster_of, gt_is_bg, pred_is_bg, seed=0):
    """Per-video counters int64 [b, EVAL_COUNTERS] (columns: _lib.EVAL_COUNTER_NAMES).  (smm_eval_videos_i64)"""
    lib = _lib.load()
    _eval_check(eb, pred, gt)
    dev = pred.device
    out = torch.empty((eb.b, _lib.EVAL_COUNTERS), dtype=torch.int64, device=dev)
    ws = workspace(eb.workspace_bytes(), dev)
    _lib.check(lib.smm_eval_videos_i64(
        ctypes.byref(eb.shape), ctypes.c_void_p(eb.lengths.ctypes.data), ctypes.c_void_p(eb.frame_offset.ctypes.data),
        ctypes.c_void_p(eb.group.ctypes.data),
        ctypes.c_void_p(None if eb.video_key is None else eb.video_key.ctypes.data),
        _dev(pred, torch.int64, 'pred'), _dev(gt, torch.int64, 'gt'), _dev(local_of, torch.int32, 'local_of'),
        _dev(cluster_of, torch.int32, 'cluster_of'), _dev(gt_is_bg, torch.uint8, 'gt_is_bg'),
        _dev(pred_is_bg, torch.uint8, 'pred_is_bg'), ctypes.c_uint32(int(seed) & 0xFFFFFFFF),
        _dev(out, torch.int64, 'counters'), ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()), _stream()))
    return out


# ------------------------------------------------------------------------------------------------ closed-form fit
def fit_stats(x, labels, lengths, frame_offset, n_classes, max_k):
    """Sufficient statistics of the supervised fit for videos on the packed frame axis (smm_fit_stats_f64).

    x cuda fp32 [F, D], labels cuda int64 [F] -> dict of cuda tensors: sum_x fp64 [n_classes, D], sum_x2 fp64 [D],
    frame_counts / span_counts / span_start_counts int64 [n_classes], span_transition_counts int64 [to, from].
    """
    lib = _lib.load()
    dev = x.device
    ln = np.ascontiguousarray(np.asarray(lengths, dtype=np.int64).reshape(-1))
    fo = np.ascontiguousarray(np.asarray(frame_offset, dtype=np.int64).reshape(-1))
    b, d = int(ln.shape[0]), int(x.shape[1])
    n = int(n_classes)
    out = dict(sum_x=torch.empty((n, d), dtype=torch.float64, device=dev),
               sum_x2=torch.empty(d, dtype=torch.float64, device=dev),
               frame_counts=torch.empty(n, dtype=torch.int64, device=dev),
               span_counts=torch.empty(n, dtype=torch.int64, device=dev),
               span_start_counts=torch.empty(n, dtype=torch.int64, device=dev),
               span_transition_counts=torch.empty((n, n), dtype=torch.int64, device=dev))
    ws = workspace(lib.smm_fit_workspace_bytes(b), dev)
    _lib.check(lib.smm_fit_stats_f64(
        ctypes.c_int32(b), ctypes.c_void_p(ln.ctypes.data), ctypes.c_void_p(fo.ctypes.data),
        ctypes.c_int64(int(x.shape[0])), ctypes.c_int32(d), ctypes.c_int32(n),
        ctypes.c_int32(0 if max_k is None else int(max_k)), _dev(x, torch.float32, 'x'),
        _dev(labels, torch.int64, 'labels'), _dev(out['sum_x'], torch.float64, 'sum_x'),
        _dev(out['sum_x2'], torch.float64, 'sum_x2'), _dev(out['frame_counts'], torch.int64, 'frame_counts'),
        _dev(out['span_counts'], torch.int64, 'span_counts'),
        _dev(out['span_start_counts'], torch.int64, 'span_start_counts'),
        _dev(out['span_transition_counts'], torch.int64, 'span_transition_counts'),
        ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()), _stream()))
    off = lib.smm_fit_error_word_offset(b)
    out['_err'] = ws[off:off + 4].view(torch.int32).clone()     # (a copy, stream-ordered behind the kernels: the shared
    return out                                                   #  workspace may be reused by the next call)


def dp_timing(on):
    """Measurement aid (smmdp.h: smm_dp_timing_enable): HIP event pairs around every DP kernel launch the library makes."""
    _lib.load().smm_dp_timing_enable(1 if on else 0)


def dp_timing_read(cap=4096, tagged=False):
    """Durations (ms, launch order) of the DP kernel launches recorded since the last read; waits for them.
    ``tagged``: (ms, tag) pairs -- tag 0: the only DP launch of its call, 1: the critical videos of a split decode
    (caller's stream), 2: the rest of a split decode (the library's second stream), 3: the <= 16-state videos of a CU-time-bound
    part in four-wave workgroups, two per CU, on the library's side stream (beside a launch of tag 0 or 2)."""
    buf = (ctypes.c_float * cap)()
    tags = (ctypes.c_int32 * cap)()
    n = _lib.load().smm_dp_timing_read_tagged(buf, tags, cap)
    if tagged:
        return [(float(buf[i]), int(tags[i])) for i in range(min(n, cap))]
    return [float(buf[i]) for i in range(min(n, cap))]


def time_split_plan(batch, n_cu=256):
    """The time-split plan the library would make for this batch's Viterbi launch on a GPU of ``n_cu`` compute units (host logic
    only: include/smmdp.h, smm_time_split_plan) -> list of (video, first position, positions, positions in front of its own part)."""
    import numpy as np
    lib = _lib.load()
    ln, fo, gr, kp, ns = batch.host_ptrs()
    cap = int(batch.total_frames // 128 + 2 * batch.b + 8)
    arrs = [np.zeros(cap, dtype=np.int32) for _ in range(4)]
    n = lib.smm_time_split_plan(ctypes.byref(batch.shape), ctypes.c_void_p(ln), ctypes.c_void_p(gr), ctypes.c_void_p(kp), ctypes.c_void_p(ns),
                                int(n_cu), *[ctypes.c_void_p(a.ctypes.data) for a in arrs], cap)
    if n < 0:
        raise _lib.SmmError("smm_time_split_plan: %s" % lib.smm_strerror(n).decode())
    return [tuple(int(a[i]) for a in arrs) for i in range(min(n, cap))]


def reload_env():
    """Make the library read its SMM_* tuning switches again (it reads them once, at first use)."""
    _lib.reload_env()


def release_cached_plans():
    """smm_release_cached_plans(): frees the library's resident plans, second streams and pooled events; returns the
    device bytes given back.  Only when no call is in flight and no captured graph of a call will be replayed."""
    torch.cuda.synchronize()
    return int(_lib.load().smm_release_cached_plans())


def cached_plan_bytes():
    return int(_lib.load().smm_cached_plan_bytes())
