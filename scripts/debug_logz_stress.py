import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from action_segmentation_amd import synth, ops, semimarkov_modules as SM
from action_segmentation_amd.batching import make_data_loader
from action_segmentation_amd.semimarkov import SemiMarkovModel
from oracle import factored as F
data = synth.SynthDatasplit('tiny', seed=12)
args = synth.make_args(data.max_k, cuda=True, batch_size=2)
torch.manual_seed(0)
model = SemiMarkovModel.from_args(args, data)
m = model.model
with torch.no_grad():
    m.gaussian_means.normal_(0, 0.3); m.poisson_log_rates.uniform_(1.0, 2.0); m.transition_logits.normal_()
batches = list(make_data_loader(args, data, shuffle=False, batch_by_task=True, batch_size=2))
feats = [b['features'].to(model.device) for b in batches]
orig = ops.logz_bwd
def spy(batch, elp, trans, init, len_scores, z, grad_logz=None, endpen=None, ws=None):
    torch.cuda.synchronize()
    b, tmax, c = batch.b, batch.t_max, batch.c_max
    e = elp.view(b, tmax, c).cpu().numpy()
    for i, t in enumerate(batch.lengths):
        e[i, t:] = 0
    ref_z, ref_g = F.logz(e, batch.lengths, trans[0].cpu().numpy(), init[0].cpu().numpy(), len_scores[0].cpu().numpy(), None, grad=True,
                          upstream=grad_logz.cpu().numpy())
    g = orig(batch, elp, trans, init, len_scores, z, grad_logz=grad_logz, endpen=endpen, ws=ws)
    torch.cuda.synchronize()
    ge = g['elp'].view(b, tmax, c).cpu().numpy()
    print('bwd C', c, 'T', batch.lengths, 'z ok', np.allclose(z.cpu().numpy(), ref_z), 'elp finite', bool(np.isfinite(e).all()),
          'nan:', {k: int(torch.isnan(v).sum()) for k, v in g.items()},
          'max err elp', float(np.nanmax(np.abs(ge - ref_g['elp']))), 'len', float(np.nanmax(np.abs(g['len'][0].cpu().numpy() - ref_g['len']))), flush=True)
    if torch.isnan(g['elp']).any():
        bad = np.argwhere(np.isnan(ge))
        print('   nan positions (video, t, c) first/last:', bad[:3].tolist(), bad[-3:].tolist(), 'count', len(bad))
        # look at the histories in the workspace
    return g
ops.logz_bwd = spy
m.zero_grad()
lls = []
for b, f in zip(batches, feats):
    ll, _ = m.log_likelihood(f, b['lengths'], b['task_indices'], spans=None)
    lls.append(ll)
(-(sum(lls) / len(lls))).backward()
