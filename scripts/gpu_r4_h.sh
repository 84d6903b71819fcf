# diagnostic (-DSMM_DEV library, results INCOMPLETE with SMM_SPLIT_DEBUG): what slows the critical launch of a split decode
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for dbg in 0 1 2 3; do
SMM_SPLIT_DEBUG=$dbg SMM_LIB_PATH=$PWD/action-segmentation_amd/libsmmdp_dev.so timeout -k 10 300 python - <<'PY'
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from action_segmentation_amd import ops, synth
a = bench.parse(['--workload', 'cfg3'])
dev = torch.device('cuda:0')
cfg = synth.CONFIGS['cfg3']
data = synth.SynthDatasplit('cfg3', seed=2, device=dev)
_, model = bench.fit_model(a, cfg, data, dev, None, 1)
pc = model.prepare(data)
t = pc.tables
def step():
    return ops.decode(pc.batch, pc.x, t['w'], t['cst'], t['inv_var'], t['trans'], t['init'], t['len'], cons=pc.cons, endpen=pc.endpen, class_map=t['class_map'], want_spans=False, want_labels=True)
for _ in range(3): step()
torch.cuda.synchronize()
ops.dp_timing_read(); ops.dp_timing(True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): step()
e1.record(); torch.cuda.synchronize()
ops.dp_timing(False)
rec = ops.dp_timing_read(tagged=True)
crit = [m for m, tg in rec if tg == 1]; rest = [m for m, tg in rec if tg == 2]
print('SMM_SPLIT_DEBUG=%s: step %.3f ms; critical launch %.3f ms; rest launch %s' % (os.environ['SMM_SPLIT_DEBUG'], e0.elapsed_time(e1) / 10, np.mean(crit) if crit else -1, ('%.3f ms' % np.mean(rest)) if rest else 'none'))
PY
done 2>&1 | grep -v amdgpu.ids
