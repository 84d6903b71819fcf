"""Development aid: the DP kernel of libsmmdp_<tag>.so variants (scripts/build_variants.sh) on the bench's cfg3 corpus:
kernel time (HIP events), delayed band-blocks evaluated, and -- profile builds -- the block stamps of workgroup 0 (the
launch's most expensive video).  usage: python scripts/prof_cfg3.py base profbase ...   (SMM_PROF_LAST=1 with -DSMM_PROFILE=2)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from action_segmentation_amd import _lib, ops, synth

a = bench.parse(['--workload', os.environ.get('SMM_WORKLOAD', 'cfg3')])
dev = torch.device('cuda:0')
cfg = synth.CONFIGS[a.workload]
data = synth.SynthDatasplit(a.workload, seed=a.seed, device=dev, scale=a.scale)
_, model = bench.fit_model(a, cfg, data, dev, None, 1)
pc = model.prepare(data)
t = pc.tables
elp64, _ = ops.emission(pc.batch, pc.x, t['w'], t['cst'], t['inv_var'], cons=pc.cons)
elp64 = elp64.clone()
ref = None
for tag in sys.argv[1:]:
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libsmmdp_%s.so' % tag if tag != 'shipped' else 'libsmmdp.so')
    _lib._lib = None
    ops._ws_cache.clear()
    lab = torch.empty(pc.batch.total_frames, dtype=torch.int64, device=dev)
    kw = dict(endpen=pc.endpen, class_map=t['class_map'], want_spans=False, want_labels=True, labels_out=lab)
    ops.viterbi(pc.batch, elp64, t['trans'], t['init'], t['len'], **kw)
    torch.cuda.synchronize()
    ms = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.viterbi(pc.batch, elp64, t['trans'], t['init'], t['len'], **kw); e1.record()
        torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
    ws = list(ops._ws_cache.values())[0]
    o_err = _lib.load().smm_error_word_offset(ctypes.byref(pc.batch.shape))
    raw = ws[o_err: o_err + 512].cpu().numpy()
    pp = raw.view(np.uint64).astype(np.float64)
    tot = sum((ln // (4 if os.environ.get('SMM_BAND_B') == '4' else 8)) * pc.n_states[g] * 8 for ln, g in zip(pc.lengths, pc.group))
    same = '' if ref is None else '  labels equal to the first variant: %s' % bool((lab == ref).all())
    ref = lab.clone() if ref is None else ref
    print('%s: DP kernel %.3f ms (min %.3f); delayed band-blocks evaluated %d of %d (%.2f %%); sources pushed into band 0 per (state, block): %.3f%s'
          % (tag, float(np.mean(ms)), min(ms), raw.view(np.int32)[3], tot, 100.0 * raw.view(np.int32)[3] / tot, raw.view(np.int32)[2] / (tot / 8.0), same))
    nblk = pp[7]
    if nblk and pp[45]:
        print('   workgroup 0, cycles: prologue %d | forward %d (%.0f per block) | closing + back-trace %d (%d segments; phase A %d, phase B %d, labels %d)'
              % (pp[44], pp[45], pp[45] / nblk, pp[46], pp[43], pp[40], pp[41], pp[42]))
    if nblk:
        print('   workgroup 0: %d blocks; delayed band-blocks of its waves 1 2 3 5 6: %s' % (nblk, ' '.join('%d' % pp[k] for k in range(2, 7))))
        if pp[34]:
            print('   chain wave of workgroup 0: %d positions took the fast (speculated) transition' % pp[34])
        for w in range(8):
            extra = ''
            if os.environ.get('SMM_PROF_CLS'):
                # -DSMM_PROFILE=4: a pusher's busy cycles by class of block
                r4 = [pp[32 + 4 * w + ph] for ph in range(4)]
                mx = int(pp[16 + w])
                extra = '  blocks without a delayed band: %5d, busy %5.0f | with: %5d, busy %5.0f | with a state that pushed most sources: %5d, busy %5.0f' % (
                    r4[1], r4[0] / max(1, r4[1]), r4[3], r4[2] / max(1, r4[3]), mx & 0xfffff, (mx >> 20) / max(1, mx & 0xfffff))
            elif os.environ.get('SMM_PROF_LAST'):
                raw4 = [int(pp[32 + 4 * w + ph]) for ph in range(4)]
                cnt = [raw4[0] & 0xfffff] + raw4[1:]
                extra = '  last in %5d blocks (by j mod 4: %s), mean busy then %5.0f' % (pp[16 + w], ' '.join('%4d' % c for c in cnt), (raw4[0] >> 20) / max(1, pp[16 + w]))
            else:
                extra = '  longest %6.0f  busy by j mod 4: %s' % (pp[16 + w], ' '.join('%5.0f' % (pp[32 + 4 * w + ph] / (nblk / 4)) for ph in range(4)))
            print('   wave %2d  busy %6.0f  barrier %6.0f%s' % (w, pp[8 + w] / nblk, pp[24 + w] / nblk, extra))
    sys.stdout.flush()
