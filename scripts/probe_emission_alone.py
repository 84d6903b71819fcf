"""ops.emission alone on the bench's cfg3 corpus (x = 1.96 GB = 7.7 x the Infinity Cache): the known-bytes run that calibrates
rocprofv3's FETCH_SIZE for the emission kernel's access shape (scripts/gpu_r5_final.sh folds it into
profiles/pmc_summary.json).  Prints the bytes of x one launch reads: 4 D frames."""
import sys
sys.path.insert(0, '.')
import torch
import bench
from action_segmentation_amd import ops, synth

a = bench.parse(['--workload', 'cfg3'])
dev = torch.device('cuda:0')
cfg = synth.CONFIGS[a.workload]
data = synth.SynthDatasplit(a.workload, seed=a.seed, device=dev)
_, model = bench.fit_model(a, cfg, data, dev, None, 1)
pc = model.prepare(data)
t = pc.tables
for _ in range(5):
    elp64, _ = ops.emission(pc.batch, pc.x, t['w'], t['cst'], t['inv_var'], cons=pc.cons)
torch.cuda.synchronize()
print('emission_alone frames %d x_bytes %d elp_bytes %d launches 5' % (pc.n_frames, pc.n_frames * 4 * cfg['d'], pc.n_frames * 8 * pc.c_max))
