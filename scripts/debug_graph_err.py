import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from test_gpu_fullsize import make_corpus
from action_segmentation_amd import ops
lengths, c, k = (1500, 1100, 700), 23, 1024
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
batch = ops.Batch(ln, [c], k, c_max=c, frame_offset=off, kp=[min(k, tmax)] * b, d=d, t_max=tmax, total_frames=int(ln.sum()))
def decode():
    return ops.decode(batch, x, *tabs, want_spans=True, want_labels=True)
eager = decode(); torch.cuda.synchronize()
print('eager err', eager['_err'].tolist(), 'ptr', hex(eager['_err'].data_ptr()))
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    o = decode()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
print('side err', o['_err'].tolist(), 'ptr', hex(o['_err'].data_ptr()), 'cache keys', list(ops._ws_cache.keys()))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = decode()
print('capture: ptr', hex(out['_err'].data_ptr()), 'cache keys', list(ops._ws_cache.keys()), 'ws sizes', [v.numel() for v in ops._ws_cache.values()])
g.replay(); torch.cuda.synchronize()
print('after replay err', out['_err'].tolist(), 'labels ok', torch.equal(out['labels'], eager['labels']))
g.replay(); torch.cuda.synchronize()
print('after replay 2 err', out['_err'].tolist())
for r in range(4):
    g.replay(); torch.cuda.synchronize()
    ws = ops._ws_cache[list(ops._ws_cache.keys())[-1]]
    off = out['_err'].data_ptr() - ws.data_ptr()
    blk = ws[off - 64:off + 64].view(torch.int32).tolist()
    print('replay', r + 3, 'err', out['_err'].tolist(), 'labels ok', torch.equal(out['labels'], eager['labels']), 'spans ok', torch.equal(out['spans'], eager['spans']))
    print('   around err:', blk)
