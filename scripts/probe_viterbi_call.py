"""Host time of one SemiMarkovModule.viterbi() call at reference-default shapes (5 videos, T ~ 300, K = 20)."""
import sys, time, cProfile, pstats
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from module_util import make_args
from action_segmentation_amd.semimarkov_modules import SemiMarkovModule
torch.manual_seed(0)
n_classes, d, k = 30, 200, 20
m = SemiMarkovModule(make_args(k), n_classes, d, allow_self_transitions=False).cuda()
with torch.no_grad():
    m.gaussian_means.normal_(); m.poisson_log_rates.uniform_(1.0, 3.0); m.transition_logits.normal_()
b, T = 5, 300
lengths = torch.tensor([300, 250, 280, 190, 220])
x = torch.randn(b, T, d, device='cuda')
vc = torch.tensor([0, 3, 4, 7, 9, 12, 15, 16, 20, 22, 25, 29])
vcs = [vc] * b
def call():
    return m.viterbi(x, lengths, vcs, add_eos=True)
for _ in range(5):
    call()
torch.cuda.synchronize()
ts = []
for _ in range(30):
    t0 = time.perf_counter(); s = call(); ts.append((time.perf_counter() - t0) * 1e3)
print('viterbi() wall: min %.3f ms, median %.3f ms' % (min(ts), float(np.median(ts))))
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    call()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(25)
