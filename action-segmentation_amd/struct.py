"""The slice of ``torch_struct`` the reference uses, on the MI355X (dense potentials kept as they are).

Drop-in for ``from torch_struct import SemiMarkovCRF, SemiMarkov, MaxSemiring`` (reference
``src/models/semimarkov/semimarkov_modules.py:11, 624-657, 677-679``; ``src/models/test_semimarkov.py:7, 14, 312-314``):

    dist = SemiMarkovCRF(scores, lengths=eos_lengths)        # scores b x (N-1) x K x C x C, device fp32
    pred_spans, extra = dist.struct.from_parts(dist.argmax)
    dist.partition;  dist.log_prob(parts);  dist.struct.to_parts(seq, (C, K), lengths);  dist.struct().score(...)
    SemiMarkov(MaxSemiring).marginals(scores, lengths=lengths)

The DP runs in ``smm_dense_dp_f32`` (csrc/smm_dense.hip).  This is the compatibility boundary for lattices small
enough to materialise; ``SemiMarkovModule.viterbi / log_likelihood`` never build the dense tensor.
``partition`` is differentiable w.r.t. the potentials (``smm_dense_marginals_f32``: the posterior edge marginals are
its gradient, as autograd through torch_struct gives the reference, ``semimarkov.py:286``), and
``SemiMarkov(LogSemiring).marginals`` returns them.
"""
import ctypes

import numpy as np
import torch

from . import _lib, ops


class MaxSemiring:
    pass


class LogSemiring:
    pass


def _host_lengths(scores, lengths):
    b, n1 = scores.shape[:2]
    if lengths is None:
        lengths = torch.full((b,), n1 + 1, dtype=torch.long)
    ln = np.ascontiguousarray(lengths.detach().cpu().numpy(), dtype=np.int64)
    assert int(ln.max()) == n1 + 1, "one instance must span the whole lattice (torch_struct's _check_potentials)"
    return ln


def _dense_dp(scores, lengths, log_semiring, want_spans, ws=None):
    if not scores.is_cuda:
        raise _lib.SmmError("struct.SemiMarkovCRF runs on the MI355X only (there is no CPU path)")
    lib = _lib.load()
    scores = scores.detach().to(torch.float32).contiguous()
    b, n1, k, c, c2 = scores.shape
    assert c == c2
    ln = _host_lengths(scores, lengths)
    dev = scores.device
    v = torch.empty(b, dtype=torch.float64, device=dev)
    spans = torch.empty((b, n1 + 1), dtype=torch.int64, device=dev) if want_spans else None
    nbytes = lib.smm_dense_workspace_bytes(b, n1, k, c)
    if ws is None:
        ws = ops.workspace(nbytes, dev)
    _lib.check(lib.smm_dense_dp_f32(
        ctypes.c_void_p(scores.data_ptr()), ctypes.c_void_p(ln.ctypes.data), b, n1, k, c, 1 if log_semiring else 0,
        ctypes.c_void_p(v.data_ptr()), None if spans is None else ctypes.c_void_p(spans.data_ptr()),
        ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()),
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return v, spans


def _dense_marginals(scores32, ln, v, grad_v, ws):
    lib = _lib.load()
    b, n1, k, c, _ = scores32.shape
    out = torch.empty_like(scores32)
    _lib.check(lib.smm_dense_marginals_f32(
        ctypes.c_void_p(scores32.data_ptr()), ctypes.c_void_p(ln.ctypes.data), b, n1, k, c, ctypes.c_void_p(v.data_ptr()),
        None if grad_v is None else ctypes.c_void_p(grad_v.data_ptr()), ctypes.c_void_p(out.data_ptr()),
        ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()),
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


class _DensePartition(torch.autograd.Function):
    """log Z of dense potentials, differentiable: backward = posterior edge marginals x upstream (smm_dense_marginals_f32)."""

    @staticmethod
    def forward(ctx, scores, lengths):
        s32 = scores.detach().to(torch.float32).contiguous()
        b, n1, k, c, _ = s32.shape
        ws = torch.empty(_lib.load().smm_dense_workspace_bytes(b, n1, k, c), dtype=torch.uint8, device=s32.device)
        v, _ = _dense_dp(s32, lengths, True, False, ws=ws)
        ctx.ln, ctx.ws, ctx.dtype = _host_lengths(s32, lengths), ws, scores.dtype
        ctx.save_for_backward(s32, v)
        return v.to(scores.dtype)

    @staticmethod
    def backward(ctx, gv):
        s32, v = ctx.saved_tensors
        m = _dense_marginals(s32, ctx.ln, v, gv.to(torch.float64).contiguous(), ctx.ws)
        return m.to(ctx.dtype), None


class SemiMarkov:
    """``torch_struct.SemiMarkov``: ``marginals`` (one-hot arg-max under MaxSemiring), ``to_parts`` / ``from_parts`` / ``score``."""

    def __init__(self, semiring=LogSemiring):
        self.semiring = semiring

    def marginals(self, edge, lengths=None):
        if self.semiring is not MaxSemiring:
            # posterior edge marginals (what autograd through torch_struct's LogSemiring sum returns)
            s32 = edge.detach().to(torch.float32).contiguous()
            b, n1, k, c, _ = s32.shape
            ws = torch.empty(_lib.load().smm_dense_workspace_bytes(b, n1, k, c), dtype=torch.uint8, device=s32.device)
            v, _ = _dense_dp(s32, lengths, True, False, ws=ws)
            return _dense_marginals(s32, _host_lengths(s32, lengths), v, None, ws).to(edge.dtype)
        _, spans = _dense_dp(edge, lengths, False, True)
        b, n1, k, c, _ = edge.shape
        return self.to_parts(spans.cpu(), (c, k), lengths).to(device=edge.device, dtype=edge.dtype)

    def sum(self, edge, lengths=None):
        if self.semiring is not MaxSemiring and edge.requires_grad:
            return _DensePartition.apply(edge, lengths)
        v, _ = _dense_dp(edge, lengths, self.semiring is not MaxSemiring, False)
        return v.to(edge.dtype)

    @staticmethod
    def to_parts(sequence, extra, lengths=None):
        """span encoding b x N (-1 = continuation) -> 0/1 b x (N-1) x K x C x C"""
        c, k = extra
        seq = sequence.detach().cpu()
        b, n = seq.shape
        parts = torch.zeros(b, n - 1, k, c, c, dtype=torch.long)
        for i in range(b):
            row = seq[i].tolist()
            starts = [p for p, val in enumerate(row) if val != -1]
            for s0, s1 in zip(starts[:-1], starts[1:]):
                parts[i, s0, s1 - s0, row[s1], row[s0]] = 1
        return parts

    @staticmethod
    def from_parts(edge):
        """one-hot edges -> (span encoding b x N with -1 for continuations, (C, K))"""
        b, n_1, k, c, _ = edge.shape
        seq = torch.full((b, n_1 + 1), -1, dtype=torch.long)
        for (i, n, kk, c_to, c_from) in edge.detach().cpu().nonzero().tolist():
            if n == 0:
                seq[i, 0] = c_from
            seq[i, n + kk] = c_to
        return seq, (c, k)

    def score(self, potentials, parts, batch_dims=(0,)):
        nb = len(list(batch_dims))
        return (potentials * parts.to(potentials)).flatten(nb).sum(-1)


class SemiMarkovCRF:
    struct = SemiMarkov

    def __init__(self, log_potentials, lengths=None):
        self.log_potentials = log_potentials
        self.lengths = lengths
        self.event_shape = log_potentials.shape[1:]

    @property
    def argmax(self):
        return SemiMarkov(MaxSemiring).marginals(self.log_potentials, self.lengths)

    @property
    def max(self):
        return SemiMarkov(MaxSemiring).sum(self.log_potentials, self.lengths)

    @property
    def partition(self):
        return SemiMarkov(LogSemiring).sum(self.log_potentials, self.lengths)

    def log_prob(self, parts):
        d = parts.dim()
        batch_dims = range(d - len(self.event_shape))
        return SemiMarkov().score(self.log_potentials, parts, batch_dims) - self.partition
