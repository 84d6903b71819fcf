"""Where do SemiMarkovModel.predict's 4 s (first call / per-batch) on cfg4 go?  (What's weak #9)"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(8)
from action_segmentation_amd import synth
from action_segmentation_amd.semimarkov import SemiMarkovModel

wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg4'
cfg = synth.CONFIGS[wl]
dev = torch.device('cuda:0')
data = synth.SynthDatasplit(wl, seed=2, device=dev)
fitted = SemiMarkovModel.from_args(synth.make_args(cfg['max_k'], cuda=True, batch_size=cfg['batch_size']), data)
fitted.fit(data.subset(6), use_labels=True)
args = synth.make_args(cfg['max_k'], cuda=True, batch_size=cfg['batch_size'], sm_constrain_transitions=bool(cfg.get('narration')),
                       sm_constrain_with_narration=['test'] if cfg.get('narration') else [])
model = SemiMarkovModel.from_args(args, data)
model.model.load_state_dict(fitted.model.state_dict(), strict=False)
model.model.to(dev)


def timed(label, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    print('%-34s %9.2f ms' % (label, (time.perf_counter() - t0) * 1e3), flush=True)


timed('warm-up predict(subset(1))', lambda: model.predict(data.subset(1)))
timed('fused, first call', lambda: model.predict(data))
for i in range(3):
    timed('fused, resident #%d' % i, lambda: model.predict(data))
timed('per batch #0', lambda: model.predict(data, fused=False))
timed('per batch #1', lambda: model.predict(data, fused=False))
for label, fn in (('per batch', lambda: model.predict(data, fused=False)),
                  ('fused first call', lambda: (model.clear_prepared(), model.predict(data)))):
    pr = cProfile.Profile()
    pr.enable()
    fn()
    torch.cuda.synchronize()
    pr.disable()
    print('=' * 30, label)
    pstats.Stats(pr).sort_stats('cumulative').print_stats(28)

# ---- steady-state breakdown of the per-batch pattern (no profiler): where do ~8 ms per batch go?
import numpy as np
from action_segmentation_amd import ops, semimarkov_utils
from action_segmentation_amd.batching import make_data_loader
m = model.model
acc = {}


def tick(name, t0):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t1 - t0)
    return t1


for rep in range(2):
    acc.clear()
    t0 = time.perf_counter()
    loader = make_data_loader(args, data, shuffle=False, batch_by_task=True, batch_size=args.batch_size)
    cons_fn = model._test_constraints(data)
    t0 = tick('loader construction', t0)
    it = iter(loader)
    while True:
        try:
            batch = next(it)
        except StopIteration:
            break
        t0 = tick('collate (DataLoader next)', t0)
        features, lengths = batch['features'].to(model.device), batch['lengths']
        cons = cons_fn(batch) if cons_fn else None
        addl = model.make_additional_allowed_ends(batch['task_name'], lengths)
        t0 = tick('constraints + ends (host)', t0)
        vc = m._check_valid_classes(batch['task_indices'])
        t0 = tick('_check_valid_classes', t0)
        out = m._decode(features, lengths, vc, addl, cons, want_labels=False)
        t0 = tick('_decode (tables, endpen, launches)', t0)
        sp = out['spans'].cpu()
        t0 = tick('spans.cpu()', t0)
        ops.check_decoded(out['_batch'], out)
        t0 = tick('check_decoded', t0)
        lab = semimarkov_utils.spans_to_labels(sp)
        t0 = tick('spans_to_labels', t0)
        tr = m.trim(lab, lengths, check_eos=True)
        res = [s.numpy() for s in tr]
        t0 = tick('trim + numpy', t0)
    print('--- pass %d (ms per epoch of %d batches)' % (rep, len(loader)))
    for k, v in acc.items():
        print('%-40s %8.2f' % (k, v * 1e3))
