#!/usr/bin/env python
"""Benchmark of the semi-Markov decode path on MI355X (contract: see the task statement / DESIGN.md §Measurement).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg3c|cfg2|cfg1|cfg4] [--scale S]

One "step" = one decode pass (emission scorer + Viterbi DP + back-trace + label expansion + labels on the host: the
DP kernel writes them into pinned host memory over PCIe while it decodes) over this rank's synthetic corpus, features
already resident in HBM.  Default workload: cfg3, the shape
BASELINE.json's metric is quoted on (CrossTask-shaped: 18 tasks x 20 videos, T ~ 6k (500..14k), 11..23 states per
task, max span L = K-1 = 1023, D = 200).  With N > 1 (torchrun, one rank per GPU) every rank decodes its own
corpus of that size (weak scaling; videos are independent, so there is no data-path collective); RCCL carries the
MAX of the step time and the SUM of the metric counters.

Rank 0 prints ONE JSON line.  ``roofline`` is for the dominant kernel (the DP kernel), timed live with HIP events
on the stream it is launched on; ``cpu_baseline`` is the oracle's dense reference-path restatement on a bounded
sample (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK = 256 * 4 * 16 * 2.4e9   # fp64 lane-ops/s: 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=5)
    p.add_argument('--warmup', type=int, default=2)
    p.add_argument('--workload', default='cfg3', choices=['cfg1', 'cfg2', 'cfg3', 'cfg3c', 'cfg4'])
    p.add_argument('--scale', type=float, default=1.0, help='videos per task multiplier')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--seed', type=int, default=1000, help='corpus seed of rank 0 (rank r uses seed + r)')
    p.add_argument('--fit-videos', type=int, default=6, help='videos per task the closed-form fit sees (untimed)')
    p.add_argument('--labels-via-copy', action='store_true',
                   help='labels to a device tensor + D->H copy through a pinned buffer instead of kernel stores to pinned host memory')
    return p.parse_args()


def dist_setup(n):
    """One process per GPU (torchrun env).  Collectives run over RCCL (backend "nccl" on ROCm); if RCCL cannot come up
    on this box the tiny metric reductions fall back to gloo on host tensors rather than losing the measurement."""
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    backend = None
    if n > 1 or world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        try:
            dist.init_process_group('nccl', rank=rank, world_size=world)
            probe = torch.ones(1, device='cuda')
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            assert int(probe.item()) == world
            backend = 'nccl'
        except Exception as e:                                    # pragma: no cover (needs a broken RCCL setup)
            sys.stderr.write('bench.py: RCCL unavailable (%s); metric reductions fall back to gloo\n' % e)
            if dist.is_initialized():
                dist.destroy_process_group()
            dist.init_process_group('gloo', rank=rank, world_size=world)
            backend = 'gloo'
    else:
        torch.cuda.set_device(0)
    return rank, world, local, backend


def cpu_baseline(data, model, pc):
    """Dense reference-path restatement (oracle/dense_ref.py: log_hsmm potentials + sequential max-DP with
    back-pointers, fp32 like the reference) on a bounded sample: the first frames of the first video."""
    from oracle import dense_ref as O
    m = model.model
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    p = O.RefParams(m.n_classes, sd['poisson_log_rates'], sd['gaussian_means'], torch.diagonal(sd['gaussian_cov']).clone(),
                    sd['transition_logits'], sd['init_logits'], m.max_k, True)
    (task, name) = sorted(data._videos)[0]
    smp = data._videos[(task, name)]
    c = len(smp['task_indices'])
    k_all = min(m.max_k, 4096)
    torch.set_num_threads(min(8, os.cpu_count() or 1))     # small per-step tensors: more threads only add overhead
    feats_all = smp['features'].cpu().float()

    def run(t):
        t0 = time.perf_counter()
        with torch.no_grad():
            O.viterbi(p, feats_all[:t].unsqueeze(0), torch.tensor([t]), smp['task_indices'])   # (current smp / feats_all)
        return time.perf_counter() - t0

    # bounded sample: cost ~ t * min(K, t) * C^2; calibrate on 192 frames, then aim at ~15 s and <= 2 GB of potentials
    t_cal = min(192, feats_all.shape[0])
    dt_cal = run(t_cal)
    per_cell = dt_cal / (t_cal * min(k_all, t_cal))
    t = int(min(feats_all.shape[0], 2.0e9 / (4.0 * k_all * (c + 1) ** 2), max(k_all + 64, 15.0 / (per_cell * k_all))))
    while t > 64 and per_cell * t * min(k_all, t) > 40.0:
        t //= 2
    # about 12 s of CPU work: the same prefix length on successive videos (one video's potentials at a time in memory)
    keys = sorted(data._videos)
    frames_done, dt, n_done = 0, 0.0, 0
    for (tk, nm) in keys:
        smp = data._videos[(tk, nm)]
        feats_all = smp['features'].cpu().float()
        cv = len(smp['task_indices'])
        tt = int(min(t, feats_all.shape[0], 2.0e9 / (4.0 * k_all * (cv + 1) ** 2)))
        dt += run(tt)
        frames_done += tt
        n_done += 1
        if dt > 12.0:
            break
    return {"value": frames_done / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "first <= %d frames of each of %d videos (C=%d.., K=%d, D=%d): dense b x N x K x C x C fp32 "
                      "potentials + sequential max-DP with back-pointers (oracle/dense_ref.py), %.1f s in all"
                      % (t, n_done, c, m.max_k, feats_all.shape[-1], dt)}


def cpu_factored(pc, model, max_videos=8):
    """The plain-C factored oracle (OpenMP over videos) on a few videos: the 'fair' CPU number."""
    from oracle import factored as F
    t = pc.tables
    n = min(max_videos, pc.n_videos)
    frames, dt = 0, 0.0
    for i in range(n):
        g = pc.group[i]
        c = pc.n_states[g]
        off, ln = pc.frame_offset[i], pc.lengths[i]
        x = pc.x[off:off + ln].cpu().numpy()
        t0 = time.perf_counter()
        w = t['w'][g, :, :c].cpu().numpy()
        xd = x.astype(np.float64)
        elp = t['cst'][g, :c].cpu().numpy() + xd @ w - 0.5 * (xd * xd) @ t['inv_var'].cpu().numpy()[:, None]
        F.viterbi(elp[None], [ln], t['trans'][g, :c, :c].cpu().numpy(), t['init'][g, :c].cpu().numpy(),
                  t['len'][g, :pc.kp[i], :c].cpu().numpy())
        dt += time.perf_counter() - t0
        frames += ln
    return {"value": frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d videos, oracle/smm_oracle.c factored fp64 DP, one video at a time" % n}


def train_step_rate(args, data, model):
    """Secondary figure for config 4: frames/s of the unsupervised objective's forward + backward (emission, log Z
    forward kernel, time-reversed backward kernel, marginal kernels, chain rule into the parameters), batch by batch
    as SemiMarkovModel.fit does it (single-task batches of --batch_size videos, narration constraints on)."""
    from action_segmentation_amd.batching import make_data_loader
    m = model.model
    m.train()
    batches = list(make_data_loader(args, data, shuffle=False, batch_by_task=True, batch_size=args.batch_size))
    cons_fn = model._test_constraints(data)

    def one_pass():
        frames = 0
        for b in batches:
            m.zero_grad()
            cons = cons_fn(b) if cons_fn else None
            addl = model.make_additional_allowed_ends(b['task_name'], b['lengths'])
            ll, _ = m.log_likelihood(b['features'].to(model.device), b['lengths'], b['task_indices'], spans=None,
                                     additional_allowed_ends_per_instance=addl, constraints=cons)
            (-ll).backward()
            frames += int(b['lengths'].sum())
        torch.cuda.synchronize()
        return frames
    one_pass()
    t0 = time.perf_counter()
    frames = one_pass()
    dt = time.perf_counter() - t0
    return {"value": frames / dt, "unit": "frames/s", "batches": len(batches), "ms_per_batch": dt / len(batches) * 1e3}


def main():
    a = parse()
    torch.set_num_threads(min(8, os.cpu_count() or 1))   # host-side torch ops are tiny: a 256-thread pool only adds latency
    rank, world, local, backend = dist_setup(a.gpus)
    dev = torch.device('cuda', torch.cuda.current_device())
    red_dev = dev if backend != 'gloo' else torch.device('cpu')
    from action_segmentation_amd import ops, synth
    from action_segmentation_amd.semimarkov import SemiMarkovModel

    cfg = synth.CONFIGS[a.workload]
    data = synth.SynthDatasplit(a.workload, seed=a.seed + rank, device=dev, scale=a.scale)
    fit_args = synth.make_args(cfg['max_k'], cuda=True, batch_size=cfg['batch_size'])
    fitted = SemiMarkovModel.from_args(fit_args, data)
    fitted.fit(data.subset(a.fit_videos), use_labels=True)   # closed-form fit (device statistics) on a few videos per task
    args = synth.make_args(cfg['max_k'], cuda=True, batch_size=cfg['batch_size'],
                           sm_constrain_transitions=bool(cfg.get('narration')),
                           sm_constrain_with_narration=['test'] if cfg.get('narration') else [])
    model = SemiMarkovModel.from_args(args, data)
    model.model.load_state_dict(fitted.model.state_dict(), strict=False)
    model.model.to(dev)
    pc = model.prepare(data)                                         # inputs resident in HBM from here on
    frames = pc.n_frames
    t = pc.tables
    stream = torch.cuda.current_stream()

    def step(events=None):
        """emission -> DP -> labels on the host.  (Same two launches as smm_decode_f32; split only so that HIP
        events can bracket the DP kernel on the stream it runs on.)  Returns the int64 labels as a CPU tensor."""
        elp64, _ = ops.emission(pc.batch, pc.x, t['w'], t['cst'], t['inv_var'], cons=pc.cons)
        if events:
            events[0].record(stream)
        out = ops.viterbi(pc.batch, elp64, t['trans'], t['init'], t['len'], endpen=pc.endpen,
                          class_map=t['class_map'], want_spans=False, want_labels=True, labels_on_host=not a.labels_via_copy)
        if events:
            events[1].record(stream)
        if a.labels_via_copy:
            return ops.to_host(out['labels'])
        stream.synchronize()          # the kernel wrote the labels into pinned host memory: they are on the host now
        return out['labels']

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()

    for _ in range(a.warmup):
        labels = step()
    sync()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    import gc
    gc.collect()
    gc.disable()                      # a collection inside a 0.5 ms step would be a quarter of it
    t0 = time.perf_counter()
    for i in range(a.steps):
        labels = step(evs[i])
    sync()
    dt = time.perf_counter() - t0
    gc.enable()
    ops.check_decoded(pc.batch)
    labels_dev = labels.to(dev)      # the evaluation kernels below (outside the timed region) read device labels
    dp_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in evs]))

    # evaluation, outside the timed region: the reference's per-task statistics from device counters.
    # Every rank holds its own synthetic corpus (weak scaling), so the statistics are per rank; the job's collectives
    # are the all-reduce of the MoF counters (SUM) and of the wall time (MAX).
    from action_segmentation_amd import evaluation

    space = evaluation.LabelSpace.from_corpus(data.corpus, list(data._videos_by_task))
    gt_dev = torch.empty(pc.batch.total_frames, dtype=torch.int64, device=dev)
    for nm, tk, o, n in zip(pc.video_names, pc.task_names, pc.frame_offset, pc.lengths):
        gt_dev[o:o + n] = data._videos[(tk, nm)]['gt_single'].to(dev)
    eval_ms = []
    for _ in range(3):
        sync()
        e0 = time.perf_counter()
        stats_by_task = evaluation.evaluate_labels(labels_dev, gt_dev, pc.lengths, pc.frame_offset, pc.task_names, space,
                                                   optimal_assignment=False, seed=0)
        eval_ms.append((time.perf_counter() - e0) * 1e3)
    summary = evaluation.summarise(stats_by_task, evaluation.STAT_KEYS)
    # closed-form fit statistics over the whole resident corpus: one HBM pass over the features
    fit_ms = []
    for _ in range(4):
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record(stream)
        ops.fit_stats(pc.x, gt_dev, pc.lengths, pc.frame_offset, data.corpus.n_classes, cfg['max_k'])
        f1.record(stream)
        torch.cuda.synchronize()
        fit_ms.append(f0.elapsed_time(f1))
    fit_bytes = frames * (4 * cfg['d'] + 16)
    lab = labels.numpy()
    correct = sum(int((lab[o:o + n] == data._videos[(tk, nm)]['gt_single'].cpu().numpy()).sum())
                  for nm, tk, o, n in zip(pc.video_names, pc.task_names, pc.frame_offset, pc.lengths))
    counters = torch.tensor([float(correct), float(frames), float(frames)], dtype=torch.float64, device=red_dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    if world > 1:
        torch.distributed.all_reduce(counters, op=torch.distributed.ReduceOp.SUM)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    assert abs(summary['mof'] - correct / frames) < 1e-12, "device MoF != host MoF"
    total_frames = float(counters[2])
    dt = float(tmax[0])

    if rank == 0:
        c_avg = float(np.mean([pc.n_states[g] for g in pc.group]))
        cells = sum(ln * ((min(kp, ln + 1) - 1) * pc.n_states[g] + pc.n_states[g] ** 2)
                    for ln, kp, g in zip(pc.lengths, pc.kp, pc.group))
        # algorithmic HBM bytes of the DP kernel per frame: elp in 8C, history out 24C, label out 8
        dp_bytes = sum(ln * (32 * pc.n_states[g] + 8) for ln, g in zip(pc.lengths, pc.group))
        achieved = dp_bytes / (dp_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, 'profiles', 'pmc_summary.json')
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(a.workload, {}).get('smm_viterbi_kernel_hbm_bytes_per_launch')
            except Exception:
                traffic = None
        res = {
            "metric": "frames/sec semi-Markov decode, CrossTask T~10k K~20 L=1024, 1/2/4/8 GPUs",
            "value": total_frames * a.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d tasks x %d videos per GPU, %d frames per GPU (T %d..%d), %d..%d states per "
                                   "task (mean %.1f), max span length %d, D=%d; closed-form-fitted HSMM parameters"
                       % (a.workload, cfg['n_tasks'], len(pc.lengths) // cfg['n_tasks'], frames, min(pc.lengths),
                          max(pc.lengths), min(pc.n_states), max(pc.n_states), c_avg, cfg['max_k'] - 1, cfg['d']),
                       "parallelism": "videos sharded across %d GPU(s), no data-path collective; metric counters all-reduced over %s"
                                      % (world, {None: "nothing (1 rank)", "nccl": "RCCL", "gloo": "gloo (RCCL unavailable)"}[backend])},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "smm_viterbi_kernel", "kernel_ms": dp_ms,
                         "algorithmic_bytes_per_launch": dp_bytes,
                         "note": "the DP is fp64-VALU-bound, not HBM-bound: %.3g lattice cells/launch = %.2f T cell/s "
                                 "= %.3f of the 2-op-per-cell fp64 VALU peak" % (
                                     cells, cells / (dp_ms * 1e-3) / 1e12,
                                     2 * cells / (dp_ms * 1e-3) / FP64_VALU_PEAK)},
            "mof": float(counters[0] / counters[1]),
            "evaluation": {"ms": min(eval_ms), "frames_per_s": frames / (min(eval_ms) * 1e-3),
                           "what": "accuracy_corpus statistics (confusion + per-video counters on the device, "
                                   "assignment and ratios on the host), all tasks, outside the timed decode",
                           "stats": {k: round(v, 6) for k, v in summary.items()}},
            "fit_stats": {"ms": min(fit_ms[1:]), "frames_per_s": frames / (min(fit_ms[1:]) * 1e-3),
                          "roofline": {"bound": "hbm", "achieved": fit_bytes / (min(fit_ms[1:]) * 1e-3) / 1e9,
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": fit_bytes / (min(fit_ms[1:]) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "algorithmic_bytes_per_launch": fit_bytes},
                          "what": "smm_fit_stats_f64 (class sums + span statistics) over every frame of the workload"},
        }
        if a.workload == 'cfg4':
            res["logz_fwd_bwd"] = train_step_rate(args, data, model)
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(data, model, pc)
            try:
                res["cpu_factored"] = cpu_factored(pc, model)
            except Exception as e:                              # the C oracle needs gcc on the box; report, don't fail
                res["cpu_factored"] = {"error": str(e)}
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
