"""Where the packed training step of cfg4 (log_likelihood_packed + backward: bench.py train_step_rate 'packed') spends its wall time:
the step timed with the host in the loop, then with torch.profiler's kernel table (GPU time per kernel) -- what is DP, what is glue."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from action_segmentation_amd import synth
from action_segmentation_amd.batching import make_data_loader, pack_batches
a = bench.parse(['--workload', 'cfg4'])
dev = torch.device('cuda:0')
cfg = synth.CONFIGS['cfg4']
data = synth.SynthDatasplit('cfg4', seed=a.seed, device=dev)
args, model = bench.fit_model(a, cfg, data, dev, None, 1)
m = model.model
m.train()
batches = list(make_data_loader(args, data, shuffle=False, batch_by_task=True, batch_size=args.batch_size))
cons_fn = model._test_constraints(data)
ends_fn = lambda b: model.make_additional_allowed_ends(b['task_name'], b['lengths'])
pc = pack_batches(batches, model.device, m.max_k, constraints_fn=cons_fn, additional_ends_fn=ends_fn)


def packed():
    m.zero_grad()
    ll = m.log_likelihood_packed(pc)
    (-ll.mean()).backward()


for _ in range(3):
    packed()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    t0 = time.perf_counter(); packed(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print('packed step: min %.3f median %.3f ms' % (min(ts), sorted(ts)[5]))
# host time alone (no sync): how long the host needs to queue the step
ts = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); packed(); ts.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
print('host time to queue a step: min %.3f median %.3f ms' % (min(ts), sorted(ts)[5]))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        packed()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=28, max_name_column_width=60)[:9000])
