"""The decode path enqueues on the caller's stream and never waits for it (metadata travels in kernel arguments): a
whole decode -- metadata uploads, band tables, emission kernel, DP kernel -- must capture into a
hipGraph and replay bit-identically, also after the features changed in place (include/smmdp.h: stream-async,
caller-owned buffers)."""
import numpy as np
import pytest
import torch

from test_gpu_fullsize import make_corpus

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('shape', [((1500, 1100, 700), 23, 1024), ((400, 380, 300, 250, 90), 9, 64),
                                   ((2600, 2500) + (900,) * 30, 7, 1024)])
def test_decode_captures_into_a_hip_graph_and_replays_bit_exactly(shape, monkeypatch):
    from action_segmentation_amd import ops
    lengths, c, k = shape
    if len(lengths) > 24:
        # eager: a SPLIT decode (smm_api.hip: choose_split): the two longest videos on the caller's stream, the other thirty
        # scored and decoded on the library's second stream beside them.  Under capture the same call does NOT split (the
        # shared second stream must not be pulled into somebody's capture): one launch pair, same bits
        monkeypatch.setenv('SMM_SPLIT_MIN_US', '0')
    cp = make_corpus(31, lengths, c, k, d=64, rate=(10, 120))
    dev = torch.device('cuda:0')
    ln = np.asarray(lengths, dtype=np.int64)
    b, tmax, d = len(ln), int(ln.max()), cp['d']
    off = np.concatenate([[0], np.cumsum(ln)[:-1]])
    x = torch.from_numpy(np.concatenate(cp['xs'], 0)).to(dev)
    mu, var = cp['mu'], cp['var']
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev).contiguous()
    tabs = (t((mu / var).T[None]), t((-0.5 * (mu * mu / var).sum(1) - 0.5 * np.log(var).sum() - 0.5 * d * np.log(2 * np.pi))[None]),
            t(1.0 / var), t(cp['trans'][None]), t(cp['init'][None]), t(cp['lens'][None]))
    batch = ops.Batch(ln, [c], k, c_max=c, frame_offset=off, kp=[min(k, tmax)] * b, d=d, t_max=tmax,
                      total_frames=int(ln.sum()))

    def decode():
        return ops.decode(batch, x, *tabs, want_spans=True, want_labels=True)

    eager = decode()
    torch.cuda.synchronize()
    ops.check_decoded(batch, eager)
    ref = (eager['labels'].clone(), eager['spans'].clone(), eager['best'].clone())
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        decode()                                        # warm-up on the side stream (workspace of that stream)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = decode()
    for rep in range(2):
        out['labels'].fill_(-9); out['spans'].fill_(-9)
        graph.replay()
        torch.cuda.synchronize()
        assert out['_err'].tolist()[0] == 0
        assert torch.equal(out['labels'], ref[0]) and torch.equal(out['spans'], ref[1]) and torch.equal(out['best'], ref[2])
    # new features in the same buffer: replay == eager decode of the new features
    x.copy_(x.flip(0))
    eager2 = decode()
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out['labels'], eager2['labels']) and torch.equal(out['spans'], eager2['spans'])
    assert not torch.equal(eager2['labels'], ref[0])


def test_training_kernels_capture_into_a_hip_graph():
    """emission -> log-partition (both directions) -> marginals -> emission chain rule, captured once, replayed three
    times: every replay must reproduce the eager gradients (the accumulators are zeroed by kernels inside the graph)."""
    from action_segmentation_amd import ops
    lengths, c, k = (700, 512, 300, 64), 11, 64
    cp = make_corpus(5, lengths, c, k, d=40, rate=(5, 40))
    dev = torch.device('cuda:0')
    ln = np.asarray(lengths, dtype=np.int64)
    b, tmax, d = len(ln), int(ln.max()), cp['d']
    off = np.concatenate([[0], np.cumsum(ln)[:-1]])
    x = torch.from_numpy(np.concatenate(cp['xs'], 0)).to(dev)
    mu, var = cp['mu'], cp['var']
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev).contiguous()
    w, cst, iv = t((mu / var).T[None]), t((-0.5 * (mu * mu / var).sum(1) - 0.5 * np.log(var).sum() - 0.5 * d * np.log(2 * np.pi))[None]), t(1.0 / var)
    trans, init, lens = t(cp['trans'][None]), t(cp['init'][None]), t(cp['lens'][None])
    batch = ops.Batch(ln, [c], k, c_max=c, frame_offset=off, kp=[min(k, tmax)] * b, d=d, t_max=tmax,
                      total_frames=int(ln.sum()))
    ws = torch.empty(batch.workspace_bytes(), dtype=torch.uint8, device=dev)

    def step():
        elp, _ = ops.emission(batch, x, w, cst, iv)
        z = ops.logz(batch, elp, trans, init, lens, ws=ws, with_backward=True)
        g = ops.logz_bwd(batch, elp, trans, init, lens, z, ws=ws, with_backward=True)
        g_w, g_cst, g_iv = ops.emission_bwd(batch, x, g['elp'], ws=ws)
        return z, g['trans'], g['init'], g['len'], g_w, g_cst, g_iv

    eager = [v.clone() for v in step()]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = step()
    for rep in range(3):
        graph.replay()
        torch.cuda.synchronize()
        for name, a, e in zip(('logz', 'g_trans', 'g_init', 'g_len', 'g_w', 'g_cst', 'g_inv_var'), out, eager):
            # (sums that leave through fp64 atomics are not bit-reproducible from run to run)
            np.testing.assert_allclose(a.cpu().numpy(), e.cpu().numpy(), rtol=1e-9, atol=1e-9 * float(e.abs().max()),
                                       err_msg='%s, replay %d' % (name, rep))
