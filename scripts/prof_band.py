"""Block stamps of the BAND kernel (diagnostic -DSMM_PROFILE build: scripts/gpu_r3_prof.sh): busy / in-barrier cycles
per hand-over block of every wave of workgroup 0, BAND mode vs the 1024-slot rings, on CrossTask-like lattices (the true
state's emission beats the others by ~18 nats per frame; Poisson length tables with rates 20..400)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from scipy.special import gammaln
from action_segmentation_amd import _lib, ops


def structured(seed, b, t, c, k, margin=18.0, rate=(20, 400)):
    g = np.random.default_rng(seed)
    rates = g.uniform(rate[0], rate[1], size=c)
    elp = np.zeros((b, t, c))
    for i in range(b):
        lab, cur, tot = [], int(g.integers(0, c)), 0
        while tot < t:
            ln = int(np.clip(g.poisson(rates[cur]), 1, k - 1))
            lab.append(np.full(ln, cur)); tot += ln; cur = (cur + 1) % c
        lab = np.concatenate(lab)[:t]
        e = -290.0 - margin + 6.0 * g.standard_normal((t, c))
        e[np.arange(t), lab] += margin
        elp[i] = e
    kk = np.arange(k)[:, None]
    lens = kk * np.log(rates) - rates - gammaln(kk + 1)
    trans = np.log(g.dirichlet(np.ones(c) * 0.5, size=c).T + 1e-3)
    trans -= np.log(np.exp(trans).sum(0, keepdims=True))
    init = np.log(g.dirichlet(np.ones(c)))
    return elp, trans, init, lens


def run(b, T, C, K, lib='libsmmdp_prof16.so'):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), lib)
    _lib._lib = None
    dev = torch.device('cuda:0')
    elp, trans, init, lens = structured(1, b, T, C, K)
    tt = lambda a: torch.tensor(a, dtype=torch.float64, device=dev).contiguous()
    batch = ops.Batch([T] * b, [C], K, t_max=T, total_frames=b * T)
    args = (tt(elp.reshape(b * T, C)), tt(trans[None]), tt(init[None]), tt(lens[None]))
    ops._ws_cache.clear()
    ops.viterbi(batch, *args); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = ops.viterbi(batch, *args); e1.record(); torch.cuda.synchronize()
    ws = list(ops._ws_cache.values())[0]
    o_err = _lib.load().smm_error_word_offset(ctypes.byref(batch.shape))
    raw = ws[o_err: o_err + 512].cpu().numpy()
    pp = raw.view(np.uint64).astype(np.float64)
    nblk = pp[7]
    print(f"b={b} T={T} C={C} K={K}: {e0.elapsed_time(e1):.3f} ms = {e0.elapsed_time(e1) * 1e6 / T:.0f} ns/frame; band-blocks evaluated "
          f"{raw.view(np.int32)[3]} of {b * (T // (4 if os.environ.get('SMM_BAND_B') == '4' else 8)) * C * 8}; cycles per block (busy / in barrier) by wave:")
    if nblk:
        if pp[45]:
            print('   workgroup 0, cycles: prologue %d | forward %d (%.0f per block) | closing + back-trace %d (%d segments; phase A %d, phase B %d, labels %d)'
                  % (pp[44], pp[45], pp[45] / nblk, pp[46], pp[43], pp[40], pp[41], pp[42]))
        if pp[34]:
            print("   chain wave of workgroup 0: %d of %d positions took the fast (speculated) transition" % (pp[34], T))
        for w in range(16):
            if pp[8 + w] or pp[24 + w]:
                extra = ''
                if w < 8 and pp[32 + 4 * w] and os.environ.get('SMM_PROF_LAST'):
                    # -DSMM_PROFILE=2: blocks this wave reached the barrier last (waited < 150 cycles), by j mod 4, and its mean busy time in those
                    raw4 = [int(pp[32 + 4 * w + ph]) for ph in range(4)]
                    cnt = [raw4[0] & 0xfffff] + raw4[1:]
                    tot = raw4[0] >> 20
                    extra = '  last in %5d blocks (by j mod 4: %s), mean busy then %5.0f' % (pp[16 + w], ' '.join('%4d' % c for c in cnt), tot / max(1, pp[16 + w]))
                elif w < 8 and pp[32 + 4 * w] and os.environ.get('SMM_PROF_SEG'):
                    # -DSMM_PROFILE=3: the BAND pushers' block in segments: rows + dominance test | pushes | hand-over | decisions + loads
                    extra = '  segments: %s' % ' '.join('%5.0f' % (pp[32 + 4 * w + ph] / nblk) for ph in range(4))
                elif w < 8 and pp[32 + 4 * w]:
                    extra = '  longest %6.0f  busy by j mod 4: %s' % (pp[16 + w], ' '.join('%5.0f' % (pp[32 + 4 * w + ph] / (nblk / 4)) for ph in range(4)))
                print("   wave %2d  busy %6.0f  barrier %6.0f%s" % (w, pp[8 + w] / nblk, pp[24 + w] / nblk, extra))
    sys.stdout.flush()


if __name__ == '__main__':
    which = sys.argv[1:] or ['prof']
    for band in (('1',) if os.environ.get('SMM_ONLY_BAND') else ('1', '0')):
        os.environ['SMM_BAND'] = band
        if band == '0':
            os.environ['SMM_PAIRS'] = '0'
        for lib in (['libsmmdp_prof16.so'] if 'prof' in which else []) + (['libsmmdp.so'] if 'plain' in which else []):
            print('SMM_BAND', band, lib)
            for b, t, c in ((64, 4096, 23), (64, 4096, 20), (64, 4096, 16), (64, 4096, 11), (1, 10000, 20), (250, 4096, 16)):
                if band == '0' and c > 21:
                    continue
                run(b, t, c, 1024, lib)
