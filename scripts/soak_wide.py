"""Randomised soak: 22..32-state videos at K = 1024 (gangs: pairs / triples) mixed with narrower tasks, ragged lengths
incl. very short ones, against the C twin (bit-exact best score, spans, labels)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from action_segmentation_amd import ops
import test_gpu_viterbi as TV
from oracle import factored as F
dev = torch.device('cuda:0')
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
for case in range(n_cases):
    k = 1024
    n_groups = int(rng.integers(1, 4))
    cs = [int(rng.integers(22, 33)) if g == 0 else int(rng.integers(3, 33)) for g in range(n_groups)]
    cm = max(cs)
    b = int(rng.integers(1, 7))
    group = rng.integers(0, n_groups, size=b).astype(np.int32); group[0] = 0
    lengths = np.array([int(rng.choice([rng.integers(1, 140), rng.integers(140, 1400), rng.integers(1024, 2600)])) for _ in range(b)], dtype=np.int64)
    tmax = int(lengths.max())
    ends = bool(rng.integers(0, 2))
    probs = [TV.make_problem(1000 * case + i, 1, tmax, cs[g], k, c_max=cm, ends=ends, integer=bool(rng.integers(0, 3) == 0)) for i, g in enumerate(group)]
    tabs = [TV.make_problem(5000 + 10 * case + g, 1, 8, c, k, c_max=cm, integer=bool(rng.integers(0, 3) == 0)) for g, c in enumerate(cs)]
    elp = np.stack([p['elp'][0] for p in probs])
    endpen = None
    if ends:
        endpen = np.stack([p['endpen'][0] for p in probs])
        for i, g in enumerate(group):
            endpen[i, cs[g]:] = -1e9
    batch = ops.Batch(lengths, cs, k, c_max=cm, t_max=tmax, total_frames=b * tmax, group=group)
    t = lambda a: None if a is None else torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    out = ops.viterbi(batch, t(elp.reshape(b * tmax, cm)), t(np.stack([x['trans'] for x in tabs])), t(np.stack([x['init'] for x in tabs])),
                      t(np.stack([x['lens'] for x in tabs])), t(endpen))
    torch.cuda.synchronize()
    ops.check_decoded(batch, out)
    got = {kk: v.cpu().numpy() for kk, v in out.items() if kk in ('best', 'spans', 'labels', 'n_segs')}
    for i, g in enumerate(group):
        c = cs[g]
        ep = None if endpen is None else endpen[i:i + 1, :c]
        spans, v = F.viterbi(elp[i:i + 1, :, :c], lengths[i:i + 1], tabs[g]['trans'][:c, :c], tabs[g]['init'][:c], tabs[g]['lens'][:, :c], ep)
        ok = got['best'][i] == v[0]
        sp = got['spans'][i, :lengths[i] + 1]
        ok = ok and np.array_equal(sp, spans[0][:lengths[i] + 1])
        if not ok:
            bad += 1
            print('MISMATCH case %d video %d: C=%d T=%d' % (case, i, c, lengths[i]), flush=True)
    print('case %d: groups %s, lengths %s, ends %s: ok' % (case, cs, lengths.tolist(), ends), flush=True)
print('soak done: %d mismatches' % bad)
sys.exit(1 if bad else 0)
