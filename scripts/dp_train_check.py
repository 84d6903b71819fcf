"""Unsupervised gradient training of the tiny corpus for a few optimiser steps; dumps the trained parameters and the
per-epoch losses.  One process, or one rank of a torchrun job (gloo reductions, every rank on GPU 0): data-parallel
training must reproduce the single-process parameters (tests/test_gpu_sharded.py)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from action_segmentation_amd import synth, distributed
from action_segmentation_amd.semimarkov import SemiMarkovModel

out, accum = sys.argv[1], int(sys.argv[2])
rank, world = distributed.init(backend='gloo')
torch.cuda.set_device(0)
data = synth.SynthDatasplit('tiny', seed=9)
args = synth.make_args(data.max_k, cuda=True, batch_size=2, epochs=2, batch_accumulation=accum, lr=0.05,
                       sm_constrain_transitions=True, sm_constrain_with_narration=['train'])
torch.manual_seed(0)
model = SemiMarkovModel.from_args(args, data)
logs = []
model.fit(data, use_labels=False, callback_fn=lambda epoch, stats: logs.append((epoch, stats.get('train_loss'))))
if rank == 0:
    sd = {k: v.detach().cpu().double().tolist() for k, v in model.model.state_dict().items() if v.dtype.is_floating_point}
    json.dump({'state': sd, 'logs': logs, 'world': world}, open(out, 'w'))
if world > 1:
    # every rank must hold the same parameters after training
    import torch.distributed as dist
    for k, v in sorted(model.model.state_dict().items()):
        if v.dtype.is_floating_point:
            a = v.detach().cpu().double().contiguous(); b = a.clone()
            dist.broadcast(b, 0)
            assert torch.equal(a, b), 'rank %d diverged from rank 0 in %s' % (rank, k)
