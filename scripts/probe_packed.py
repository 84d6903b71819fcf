"""Where does the packed log-likelihood step (cfg4, all batches in one launch) spend its wall time?"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np, torch
from action_segmentation_amd import ops, synth
from action_segmentation_amd.batching import make_data_loader, pack_batches
from action_segmentation_amd.semimarkov import SemiMarkovModel
cfg = synth.CONFIGS['cfg4']
dev = torch.device('cuda:0')
data = synth.SynthDatasplit('cfg4', seed=2, device=dev)
args = synth.make_args(cfg['max_k'], cuda=True, batch_size=cfg['batch_size'], sm_constrain_transitions=True,
                       sm_constrain_with_narration=['train'])
model = SemiMarkovModel.from_args(args, data)
m = model.model
m.train()
batches = list(make_data_loader(args, data, shuffle=False, batch_by_task=True, batch_size=args.batch_size))
cons_fn = model._train_constraints(data)
pc = pack_batches(batches, model.device, m.max_k, constraints_fn=cons_fn,
                  additional_ends_fn=lambda b: model.make_additional_allowed_ends(b['task_name'], b['lengths']))
def step():
    m.zero_grad()
    ll = m.log_likelihood_packed(pc)
    (-ll.mean()).backward()
    torch.cuda.synchronize()
for i in range(4):
    t0 = time.perf_counter(); step(); print('packed step %d: %.2f ms' % (i, (time.perf_counter() - t0) * 1e3), flush=True)
pr = cProfile.Profile(); pr.enable(); step(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=60))
ev = prof.key_averages()
print('ops', sum(e.count for e in ev), 'cuda kernels', sum(e.count for e in ev if e.device_type.name == 'CUDA'))
