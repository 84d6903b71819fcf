#!/usr/bin/env python
"""Benchmark of the semi-Markov decode path on MI355X (contract: see the task statement / DESIGN.md §Measurement).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg5|cfg3c|cfg2|cfg1|cfg4] [--scaling weak|strong]

One "step" = one decode pass (emission scorer + Viterbi DP + back-trace + label expansion + labels on the host: the
DP kernel writes them into pinned host memory over PCIe while it decodes) over this rank's share of a synthetic corpus,
features already resident in HBM.  Default workload: cfg3, the shape BASELINE.json's metric is quoted on
(CrossTask-shaped: 18 tasks x 20 videos, T ~ 6k (500..14k), 11..23 states per task, max span L = K-1 = 1023, D = 200;
seed 2 as SURVEY.md 8(d) fixes it: the draw contains two 23-state tasks, one with a 14 000-frame video).

N > 1: one process per GPU.  Under ``torchrun`` (RANK / WORLD_SIZE in the environment) this process is one rank; started
plainly as ``python bench.py --gpus N`` it spawns the N ranks itself as child processes BEFORE touching the GPU and
relays rank 0's line.  Either way the job fails (non-zero exit) unless the group comes up with exactly N ranks on RCCL
(backend "nccl"); ``--backend gloo`` is an explicit opt-in for rehearsals without one GPU per rank.
  * headline, at EVERY N (``scaling: weak``; the same ``config.workload`` string for N = 1, 2, 4, 8): every rank decodes
    its own cfg3-sized corpus (seed + rank); videos are independent, so there is no data-path collective; RCCL carries
    the MAX of the step time and the SUM of the frame counters;
  * ``strong_scaling`` (second object in the same line, at every N including 1; --no-strong-leg skips it): ONE corpus
    (default cfg5 = the full CrossTask primary set, 2754 videos) sharded by video over the ranks
    (distributed.shard_batches), every rank decodes its share, and the reference's evaluation counters
    (Datasplit.accuracy_corpus, src/data/corpus.py:405-604, summed as src/main.py:486-532 does) are all-reduced over
    RCCL before they are finalised: the statistics printed are those of the whole corpus and do not depend on N.
    ``--scaling strong`` makes this leg the headline instead -- again at every N, so a curve never mixes the two.

Rank 0 prints ONE JSON line.  ``roofline`` is for the dominant kernel (the DP kernel), timed live with HIP events
on the stream it is launched on; ``cpu_baseline`` is the oracle's dense reference-path restatement on a bounded
sample (N = 1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK = 256 * 4 * 16 * 2.4e9   # fp64 lane-ops/s: 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz
METRIC = "frames/sec semi-Markov decode, CrossTask T~10k K~20 L=1024, 1/2/4/8 GPUs"


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=5)
    p.add_argument('--warmup', type=int, default=2)
    p.add_argument('--workload', default='cfg3', choices=['cfg1', 'cfg2', 'cfg3', 'cfg3c', 'cfg4', 'cfg5', 'refdef', 'tiny'])
    p.add_argument('--scale', type=float, default=1.0, help='videos per task multiplier')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--seed', type=int, default=2, help='corpus seed of rank 0 (weak scaling: rank r uses seed + r)')
    p.add_argument('--fit-videos', type=int, default=6, help='videos per task the closed-form fit sees (untimed)')
    p.add_argument('--labels-via-copy', action='store_true',
                   help='labels to a device tensor + D->H copy through a pinned buffer instead of kernel stores to pinned host memory')
    p.add_argument('--scaling', default=None, choices=['weak', 'strong'],
                   help='which leg is the headline value (the other one is reported beside it), the same for every N.  Default: '
                        'weak (every rank its own cfg3-sized corpus: the workload BASELINE.json quotes the metric on); strong = '
                        'BASELINE config 5, ONE corpus sharded by video over the ranks')
    p.add_argument('--strong-workload', default='cfg5', choices=['cfg3', 'cfg5', 'cfg4', 'tiny'],
                   help='corpus of the strong-scaling leg (one corpus, seed --seed, sharded by video over the ranks)')
    p.add_argument('--strong-leg', action='store_true', help='(kept for old command lines: the strong-scaling leg runs at every N since round 5)')
    p.add_argument('--no-strong-leg', action='store_true')
    p.add_argument('--no-predict-e2e', action='store_true', help='skip the SemiMarkovModel.predict wall-time figures')
    p.add_argument('--second-seed', type=int, default=1000,
                   help='N = 1, cfg3 only: also time the decode of the corpus of this seed (round 1\'s default draw, 11..21 '
                        'states) and report it beside the headline; negative: skip')
    p.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                   help="collectives: 'nccl' = RCCL (required for a measurement); 'gloo' only for rehearsals")
    p.add_argument('--share-gpus', action='store_true',
                   help='rehearsal only: ranks may share a GPU (LOCAL_RANK %% device_count); needs --backend gloo')
    p.add_argument('--dry-run', action='store_true',
                   help='host-only rehearsal of the N-rank path: rendezvous, shard the corpus structure, reduce the '
                        'frame counters; nothing is decoded and no GPU is touched (implies --backend gloo)')
    a = p.parse_args(argv)
    if a.dry_run:
        a.backend = 'gloo'
    if a.share_gpus and a.backend != 'gloo':
        p.error('--share-gpus needs --backend gloo (RCCL refuses two ranks on one GPU)')
    if a.gpus < 1:
        p.error('--gpus must be >= 1')
    if a.scaling is None:
        a.scaling = 'weak'                  # ONE headline leg for every N (a 1 -> 8 curve must not change workload)
    return a


# ------------------------------------------------------------------------------------------------ N-rank launch
def free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def spawn_ranks(a, argv):
    """``python bench.py --gpus N`` without torchrun: start the N ranks as fresh child processes (this parent never
    touches the GPU runtime at all -- a rank without a GPU of its own fails in dist_setup), wait for them, exit with
    their status.  Rank 0's stdout is the parent's, so its JSON line is the job's line."""
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(a.gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY='0', LOCAL_WORLD_SIZE=str(a.gpus))
    procs = []
    for r in range(a.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            c = p.poll()
            if c is None:
                continue
            alive.remove(p)
            if c != 0 and rc == 0:
                rc = c
                for q in alive:                      # a rank died: the others would wait for it until the timeout
                    q.terminate()
        time.sleep(0.05)
    return rc


def dist_setup(a):
    """One process per GPU.  Fails loudly: the world size must equal --gpus and the backend must come up as asked."""
    import torch.distributed as dist
    from action_segmentation_amd import distributed as D
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torchrun --nproc-per-node %d, or plainly "
                         "as `python bench.py --gpus %d`, which spawns the ranks)" % (a.gpus, world, a.gpus, a.gpus))
    if a.dry_run:
        if world > 1:
            D.init('gloo')
        return rank, world, local, ('gloo' if world > 1 else None)
    n_dev = torch.cuda.device_count()
    if a.share_gpus:
        torch.cuda.set_device(local % max(1, n_dev))
    else:
        if local >= n_dev:
            raise SystemExit("bench.py: rank %d (LOCAL_RANK %d) has no GPU of its own (%d visible)" % (rank, local, n_dev))
        torch.cuda.set_device(local)
    if world == 1 and os.environ.get('SMM_DIST_SINGLE_RANK') != '1':
        return rank, world, local, None
    # (SMM_DIST_SINGLE_RANK=1: a one-rank group, so that every collective of the N-rank path runs through RCCL on the
    # one GPU of a test box -- tests/test_gpu_rccl.py)
    D.init(a.backend)
    probe = torch.ones(1, device=D.reduce_device())
    dist.all_reduce(probe)
    if a.backend == 'nccl':
        torch.cuda.synchronize()
    if int(probe.item()) != world or dist.get_world_size() != world:
        raise SystemExit("bench.py: the %s group did not come up with %d ranks" % (a.backend, world))
    return rank, world, local, a.backend


# ------------------------------------------------------------------------------------------------ CPU baselines
def cpu_baseline(data, model, pc):
    """Dense reference-path restatement (oracle/dense_ref.py: log_hsmm potentials + sequential max-DP with
    back-pointers, fp32 like the reference) on a bounded sample: the first frames of the first videos."""
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


def spans_to_frame_labels(row, t):
    """span encoding (label at every span start, -1 = continuation) -> one label per frame (forward fill)."""
    row = np.asarray(row[:t])
    idx = np.maximum.accumulate(np.where(row != -1, np.arange(t), 0))
    return row[idx]


def cpu_factored(pc, model, gpu_labels=None, budget_s=12.0):
    """The plain-C factored oracle, OpenMP over the videos of one task at a time (the 'fair' CPU
    number): emission + Viterbi for whole tasks until ~budget_s of wall time are spent.  ``gpu_labels`` (host int64
    [total_frames], the labels the TIMED decode produced): the twin's spans are turned into frame labels and compared
    with them for every video the twin covered -- the checker's answer was computed anyway (second return value)."""
    from oracle import factored as F
    t = pc.tables
    checked = mismatches = vids_checked = 0
    cores = F.set_threads(F.host_cores())
    frames, dt, n_vid, n_task = 0, 0.0, 0, 0
    by_group = {}
    for i in range(pc.n_videos):
        by_group.setdefault(pc.group[i], []).append(i)
    inv_var = t['inv_var'].cpu().numpy()
    for g, vids in sorted(by_group.items()):
        c = pc.n_states[g]
        tmax = max(pc.lengths[i] for i in vids)
        kp = max(pc.kp[i] for i in vids)
        w = t['w'][g, :, :c].cpu().numpy()
        cst = t['cst'][g, :c].cpu().numpy()
        xs = [pc.x[pc.frame_offset[i]:pc.frame_offset[i] + pc.lengths[i]].cpu().numpy() for i in vids]
        t0 = time.perf_counter()
        elp = np.zeros((len(vids), tmax, c))
        for j, x in enumerate(xs):                           # (the emission GEMM runs on numpy's BLAS threads)
            xd = x.astype(np.float64)
            elp[j, :x.shape[0]] = cst + xd @ w - 0.5 * (xd * xd) @ inv_var[:, None]
        ep = None if pc.endpen is None else pc.endpen[vids][:, :c].cpu().numpy()
        if pc.cons is not None:
            for j, i in enumerate(vids):
                elp[j, :pc.lengths[i]] += pc.cons[pc.frame_offset[i]:pc.frame_offset[i] + pc.lengths[i], :c].cpu().numpy()
        spans, _ = F.viterbi(elp, [pc.lengths[i] for i in vids], t['trans'][g, :c, :c].cpu().numpy(),
                             t['init'][g, :c].cpu().numpy(), t['len'][g, :kp, :c].cpu().numpy(), ep)
        dt += time.perf_counter() - t0
        if gpu_labels is not None:                               # (outside the CPU timing)
            cmap = t['class_map'][g].cpu().numpy()
            for j, i in enumerate(vids):
                n, o = pc.lengths[i], pc.frame_offset[i]
                ref = cmap[spans_to_frame_labels(spans[j], n)]
                mismatches += int((ref != gpu_labels[o:o + n]).sum())
                checked += n
                vids_checked += 1
        frames += sum(pc.lengths[i] for i in vids)
        n_vid += len(vids)
        n_task += 1
        if dt > budget_s:
            break
    res = {"value": frames / dt, "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": "%d videos of %d task(s), oracle/smm_oracle.c factored fp64 DP, OpenMP over the videos of a task "
                     "(%d host threads), %.1f s" % (n_vid, n_task, cores, dt)}
    par = None if gpu_labels is None else {"frames_checked": checked, "videos_checked": vids_checked,
                                            "label_mismatches": mismatches}
    return res, par


# ------------------------------------------------------------------------------------------------ legs
def fit_model(a, cfg, data, dev, D, world):
    """Closed-form fit on a few videos per task (untimed), the decode model around it.  N > 1: rank 0's parameters are
    broadcast, like every rank loading the same pickle."""
    from action_segmentation_amd import synth
    from action_segmentation_amd.semimarkov import SemiMarkovModel
    fit_args = synth.make_args(cfg['max_k'], cuda=True, batch_size=cfg['batch_size'])
    fitted = SemiMarkovModel.from_args(fit_args, data)
    fitted.fit(data.subset(a.fit_videos), use_labels=True)
    args = synth.make_args(cfg['max_k'], cuda=True, batch_size=cfg['batch_size'],
                           sm_constrain_transitions=bool(cfg.get('narration')),
                           sm_constrain_with_narration=['test'] if cfg.get('narration') else [])
    model = SemiMarkovModel.from_args(args, data)
    model.model.load_state_dict(fitted.model.state_dict(), strict=False)
    model.model.to(dev)
    return args, model


def timed_decode(a, pc, world, want_events=True):
    """W warm-up + exactly K timed decode steps of this rank's packed corpus, bracketed by barrier + synchronize.
    Returns (wall seconds of the K steps, DP kernel launch statistics of those steps or None, last labels (host int64))."""
    from action_segmentation_amd import ops
    empty = pc is None or pc.n_videos == 0
    t = None if empty else pc.tables
    stream = torch.cuda.current_stream()

    def step():
        """One decode pass through the product's entry point (smm_decode_f32): emission -> DP -> labels on the host."""
        if empty:
            return torch.zeros(0, dtype=torch.int64)
        # (smm_decode_f32 through ops.ResidentDecode, as SemiMarkovModule.decode_packed calls it: the fixed arguments of the
        # call are marshalled once per corpus)
        call = last.get('call')
        if call is None:
            call = last['call'] = ops.ResidentDecode(pc.batch, pc.x, t['w'], t['cst'], t['inv_var'], t['trans'], t['init'], t['len'],
                                                     cons=pc.cons, endpen=pc.endpen, class_map=t['class_map'])
        out = call(labels_on_host=not a.labels_via_copy)
        last['out'] = out
        if a.labels_via_copy:
            return ops.to_host(out['labels'])
        stream.synchronize()          # the kernel wrote the labels into pinned host memory: they are on the host now
        return out['labels']

    def sync():
        torch.cuda.synchronize()
        if torch.distributed.is_initialized():
            torch.distributed.barrier()

    labels = None
    last = {}
    for _ in range(a.warmup):
        labels = step()
    sync()
    import gc
    gc.collect()
    gc.disable()                      # a collection inside a 0.5 ms step would be a quarter of it
    # HIP events around every DP kernel launch of the timed region, recorded by the library on the streams it launches on
    # (smm_dp_timing_*: smm_decode_f32 launches the kernel once, or twice on two streams -- the launch's critical videos
    # first, the rest beside them); two event records per launch, a few microseconds of the step
    timing = want_events and not empty
    if timing:
        ops.dp_timing_read()
        ops.dp_timing(True)
    t0 = time.perf_counter()
    for i in range(a.steps):
        labels = step()
    sync()
    dt = time.perf_counter() - t0
    gc.enable()
    dp = None
    if timing:
        ops.dp_timing(False)
        rec = ops.dp_timing_read(tagged=True)
        if rec:
            ms = [m for m, _ in rec]
            # tag 1: the launch of the launch's critical (longest) videos on the caller's stream, tag 2: the rest of a split
            # decode on the library's second stream, tag 0: the only DP launch of an unsplit call
            crit = [m for m, t in rec if t in (0, 1)]
            rest = [m for m, t in rec if t == 2]
            small = [m for m, t in rec if t == 3]             # <= 16-state videos in four-wave workgroups (two per CU), beside tag 0 / 2
            dp = {"small_wg_launch_ms": float(np.mean(small)) if small else None, "launch_ms": float(np.mean(ms)), "launches_per_step": len(ms) / a.steps, "per_step_sum_ms": float(np.sum(ms)) / a.steps,
                  "max_launch_ms": float(np.max(ms)), "critical_launch_ms": float(np.mean(crit)) if crit else None,
                  "rest_launch_ms": float(np.mean(rest)) if rest else None}
    if not empty:
        ops.check_decoded(pc.batch, last.get('out'))
        labels = labels.clone()       # (the pinned staging buffer is reused by the next decode)
        # long videos decoded as several units along the time axis (csrc/smm_chunk.hip), and how many of them had to be decoded
        # again in one piece, in the LAST timed step: error block words 4 and 5
        words = ops.error_words(pc.batch, last.get('out'))
        if dp is not None and len(words) > 7:
            dp["time_split"] = {"videos_split": words[4], "of_those_decoded_again_in_one_piece": words[5], "one_class_run_ties_resolved": words[7],
                                "why": [n for bit, n in ((1, "a cut did not certify"), (2, "closing step"), (4, "two states within the margin"),
                                                         (8, "two lengths within the margin"), (16, "NaN / too many segments")) if words[6] & bit]}
    return dt, dp, labels


def gt_on_device(pc, data, dev):
    gt_dev = torch.empty(max(1, pc.batch.total_frames), dtype=torch.int64, device=dev)
    for nm, tk, o, n in zip(pc.video_names, pc.task_names, pc.frame_offset, pc.lengths):
        gt_dev[o:o + n] = data._videos[(tk, nm)]['gt_single'].to(dev)
    return gt_dev


def video_keys(pc, data):
    """Index of each packed video inside its task's (sorted) video list: seeds the step-recall draw, so that a sharded
    evaluation draws exactly the frames the one-process evaluation draws."""
    pos = {(t, n): i for t, names in data._videos_by_task.items() for i, n in enumerate(names)}
    return [pos[(t, n)] for t, n in zip(pc.task_names, pc.video_names)]


def strong_leg(a, rank, world, dev, D):
    """ONE corpus (seed --seed) sharded by video over the ranks; evaluation counters reduced over the backend."""
    from action_segmentation_amd import evaluation, synth
    from action_segmentation_amd.batching import batch_cost
    cfg = synth.CONFIGS[a.strong_workload]
    # structure first (no features): the shard is a function of lengths and state counts only
    dry = synth.SynthDatasplit(a.strong_workload, seed=a.seed, keep=set())
    batches = dry.batch_sampler(cfg['batch_size'], True, False).batches
    costs = [batch_cost(dry, keys, cfg['max_k']) for keys in batches]
    mine = D.shard_batches(batches, costs, rank, world)
    keep = {name for i in mine for (_, name) in batches[i]}
    keep |= {n for names in dry._videos_by_task.values() for n in names[:a.fit_videos]}     # the fit's videos
    data = synth.SynthDatasplit(a.strong_workload, seed=a.seed, device=dev, keep=keep)
    args, model = fit_model(a, cfg, data, dev, D, world)
    D.broadcast_parameters(model.model, src=0)
    pc = model.prepare(data, shard=(rank, world))
    assert sorted(pc.video_names) == sorted(n for i in mine for (_, n) in batches[i])
    dt, dp, labels = timed_decode(a, pc, world, want_events=True)
    dp_ms = dp["per_step_sum_ms"] if dp else None                    # DP kernel time of a step, all its launches
    space = evaluation.LabelSpace.from_corpus(data.corpus, list(data._videos_by_task))
    if pc.n_videos:
        stats_by_task = evaluation.evaluate_labels(labels.to(dev), gt_on_device(pc, data, dev), pc.lengths, pc.frame_offset,
                                                   pc.task_names, space, optimal_assignment=False, seed=0,
                                                   video_key=video_keys(pc, data), reduce=D.all_reduce_tensor)
    else:
        z = torch.zeros(1, dtype=torch.int64, device=dev)
        stats_by_task = evaluation.evaluate_labels(z, z, [], [], [], space, optimal_assignment=False, seed=0,
                                                   video_key=[], reduce=D.all_reduce_tensor)
    summary = evaluation.summarise(stats_by_task, evaluation.STAT_KEYS)
    rd = D.reduce_device()
    tot = D.all_reduce_tensor(torch.tensor([float(pc.n_frames), float(pc.n_videos), float(sum(costs[i] for i in mine))],
                                           dtype=torch.float64, device=rd))
    tmax = D.all_reduce_tensor(torch.tensor([dt, float(pc.n_frames), dp_ms or 0.0], dtype=torch.float64, device=rd),
                               op=torch.distributed.ReduceOp.MAX)
    dt_all = float(tmax[0])
    roof = None
    if pc.n_videos and dp_ms:
        dp_bytes = sum(ln * (32 * pc.n_states[g] + 8) for ln, g in zip(pc.lengths, pc.group))
        nl = dp["launches_per_step"]
        roof = {"bound": "hbm", "achieved": dp_bytes / (dp_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": dp_bytes / (dp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel": "smm_viterbi_kernel",
                "kernel_ms": dp["launch_ms"], "launches_per_step": nl, "algorithmic_bytes_per_launch": dp_bytes / nl,
                "note": "rank 0's shard of the sharded corpus; the DP is latency-bound (one serial chain per video), not HBM-bound"}
    return {"scaling": "strong", "roofline_rank0": roof, "workload": "%s seed %d: %d videos, %d frames, sharded by video (whole single-task "
            "batches of %d, greedy LPT on the DP work)" % (a.strong_workload, a.seed, int(tot[1]), int(tot[0]), cfg['batch_size']),
            "value": float(tot[0]) * a.steps / dt_all, "unit": "frames/s", "n_gpus": world, "ms_per_step": dt_all / a.steps * 1e3,
            "frames": int(tot[0]), "videos": int(tot[1]), "max_frames_on_a_rank": int(tmax[1]),
            "dp_kernel_ms_max_over_ranks": float(tmax[2]),
            "stats_reduced_over": ("%s all-reduce%s" % ("RCCL" if torch.distributed.get_backend() == 'nccl' else "gloo",
                                                         " (one-rank group)" if world == 1 else "")) if D.active() else "1 rank",
            "stats": {k: round(v, 9) for k, v in summary.items()}}


def predict_end_to_end(model, data):
    """Wall time of SemiMarkovModel.predict over the same corpus, host work included (collate, pack_batches, table
    stacking, constraint scatter, upload of nothing: the synthetic features already live on the device): the fused ragged
    launch and the reference's per-batch call pattern (fused=False)."""
    out = {}

    def timed(fused, reps=1):
        import gc
        best = None
        gc.collect()                  # (once, in front: a full collection walks the whole heap -- 40 ms here -- and the call
        for _ in range(reps):         # right behind it runs 0.6 ms slower on cold caches: scripts/probe_predict_fused.py)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            preds = model.predict(data, fused=fused)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        frames = sum(len(v) for v in preds.values())
        return {"ms": best * 1e3, "frames_per_s": frames / best, "label_path": getattr(model, 'last_predict_path', None)}

    model.__dict__.pop('_prepared', None)
    model.predict(data.subset(1))                        # warm-up of workspaces / table cache on another datasplit
    out['fused_first_call'] = timed(True)                # collate + pack + upload + decode
    out['fused'] = timed(True, reps=5)                   # the datasplit is resident now (training loop: every epoch)
    model.predict(data, fused=False)
    out['per_batch'] = timed(False)
    out["what"] = ("SemiMarkovModel.predict(test_data) wall time, all host work included: 'fused_first_call' collates, "
                   "packs and uploads the datasplit, 'fused' finds it resident (the per-epoch decode of the training "
                   "loop, main.py:207-244); 'per_batch' = the reference's call pattern, one decode per single-task "
                   "batch (semimarkov.py:318-410); label_path: which way the labels took (fused: a leased pinned buffer or "
                   "the shared staging buffer + a copy; per batch: the ragged launch's kernel labels, or -- with narration "
                   "constraints -- the padded batch through viterbi() and spans_to_labels on the host)")
    return out


def host_features_leg(a, cfg, model, data, labels_ref, pc_ref, n_slabs=6):
    """The same decode with the features in HOST memory (the reference's flow: features loaded from disk
    into host memory, crosstask.py:95-112, every batch moved to the device, semimarkov.py:349-354): the corpus packed
    into pinned slabs once (untimed, the loader's job), then K timed passes of SemiMarkovModel.decode_host -- upload on a
    copy stream into two device buffers, overlapped with the decode of the previous slab, labels back through the DP
    kernel's stores into pinned memory.  PCIe-inclusive: never the headline value."""
    host = data.subset(10 ** 9)                              # (the same videos, features on the host)
    t0 = time.perf_counter()
    slabs = model.prepare_host(host, n_slabs)
    prep = time.perf_counter() - t0
    frames = sum(pc.n_frames for pc in slabs)
    nbytes = sum(pc.x.numel() * 4 for pc in slabs)
    labels, _ = model.decode_host(slabs)                     # warm-up (buffers, workspaces)
    # per video: the slabs hold the batches longest-first (prepare_host), the resident corpus in the loader's order
    ref_of = {(t, n): labels_ref[o:o + ln] for t, n, o, ln in zip(pc_ref.task_names, pc_ref.video_names, pc_ref.frame_offset,
                                                                  pc_ref.lengths)}
    got, off, n_vid = labels.numpy(), 0, 0
    same = True
    for s in slabs:
        for t, n, o, ln in zip(s.task_names, s.video_names, s.frame_offset, s.lengths):
            same = same and bool(np.array_equal(got[off + o:off + o + ln], ref_of[(t, n)]))
            n_vid += 1
        off += s.x.size(0)
    same = same and n_vid == len(ref_of)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        model.decode_host(slabs)
    dt = (time.perf_counter() - t0) / a.steps
    # what this box's link delivers for the same bytes as ONE pinned copy with nothing beside it: the practical bound
    raw = None
    try:
        big = max(slabs, key=lambda pc: pc.x.numel())
        dst = torch.empty(big.x.shape, dtype=big.x.dtype, device='cuda')
        dst.copy_(big.x, non_blocking=True)
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        for _ in range(3):
            dst.copy_(big.x, non_blocking=True)
        torch.cuda.synchronize()
        raw = 3 * big.x.numel() * 4 / (time.perf_counter() - c0) / 1e9
        del dst
    except Exception:
        pass
    return {"value": frames / dt, "unit": "frames/s", "ms_per_step": dt * 1e3, "slabs": len(slabs),
            "h2d_GBps": nbytes / dt / 1e9, "h2d_GBps_plain_pinned_copy": raw,
            "pcie_bound_frames_per_s": 63e9 / (4.0 * cfg['d']),
            "labels_equal_resident_decode": same, "prepare_host_s": prep,
            "what": "features in pinned host memory, streamed over PCIe every pass (%d slabs, two device buffers, copy "
                    "stream) while the previous slab decodes; PCIe Gen5 x16 bounds it at 63 GB/s / (4 D B/frame)" % len(slabs)}


def reference_default_leg(a, dev, D, with_cpu):
    """The reference's own default shapes (synth 'refdef': --sm_max_span_length 20, batches of 5 videos of one task,
    T ~ 300, 11..23 states, D = 200): SemiMarkovModel.predict in the reference's call pattern (one viterbi() per batch:
    wall time per call, everything included) and as one fused launch; with its own cpu_baseline (the dense port, which
    IS runnable at these sizes: 14 MB of potentials per video)."""
    from action_segmentation_amd import synth
    cfg = synth.CONFIGS['refdef']
    data = synth.SynthDatasplit('refdef', seed=a.seed, device=dev)
    aa = argparse.Namespace(**vars(a))
    aa.fit_videos = 5
    args, model = fit_model(aa, cfg, data, dev, D, 1)
    frames = data.n_frames
    n_batches = cfg['n_tasks']
    out = {"workload": "refdef seed %d: %d tasks x %d videos, %d frames, max span length %d, D=%d"
                       % (a.seed, cfg['n_tasks'], cfg['videos_per_task'], frames, cfg['max_k'] - 1, cfg['d'])}
    for name, fused in (('per_batch', False), ('fused', True)):
        model.predict(data, fused=fused)
        model.predict(data, fused=fused)
        best = None
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.predict(data, fused=fused)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out[name] = {"ms": best * 1e3, "frames_per_s": frames / best}
    out["viterbi_call_ms"] = out['per_batch']['ms'] / n_batches
    if with_cpu:
        from oracle import dense_ref as O
        m = model.model
        sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        p = O.RefParams(m.n_classes, sd['poisson_log_rates'], sd['gaussian_means'], torch.diagonal(sd['gaussian_cov']).clone(),
                        sd['transition_logits'], sd['init_logits'], m.max_k, True)
        torch.set_num_threads(min(8, os.cpu_count() or 1))
        done, dt = 0, 0.0
        for (tk, nm) in sorted(data._videos):
            smp = data._videos[(tk, nm)]
            x = smp['features'].cpu().float()
            t0 = time.perf_counter()
            with torch.no_grad():
                O.viterbi(p, x.unsqueeze(0), torch.tensor([x.shape[0]]), smp['task_indices'])
            dt += time.perf_counter() - t0
            done += x.shape[0]
            if dt > 8.0:
                break
        out["cpu_baseline"] = {"value": done / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "%d frames (whole videos, one at a time): dense potentials + sequential max-DP with "
                                         "back-pointers (oracle/dense_ref.py), %.1f s" % (done, dt)}
    return out


def train_step_rate(args, data, model):
    """Secondary figures for config 4: frames/s of the unsupervised objective's forward + backward (emission, log Z
    forward kernel, time-reversed backward kernel, marginal kernels, chain rule into the parameters):
      * ``packed``: every batch of the corpus through ONE launch of each kernel (log_likelihood_packed) -- the E-step of
        EM / a --batch_accumulation step over the whole corpus;
      * ``per_batch``: batch by batch as SemiMarkovModel.fit steps with --batch_accumulation 1 (single-task batches of
        --batch_size videos, narration constraints on): a latency-bound launch pair per 5 videos."""
    from action_segmentation_amd.batching import make_data_loader, pack_batches
    m = model.model
    m.train()
    batches = list(make_data_loader(args, data, shuffle=False, batch_by_task=True, batch_size=args.batch_size))
    cons_fn = model._test_constraints(data)
    ends_fn = lambda b: model.make_additional_allowed_ends(b['task_name'], b['lengths'])
    frames = sum(int(b['lengths'].sum()) for b in batches)

    def per_batch():
        for b in batches:
            m.zero_grad()
            cons = cons_fn(b) if cons_fn else None
            ll, _ = m.log_likelihood(b['features'].to(model.device), b['lengths'], b['task_indices'], spans=None,
                                     additional_allowed_ends_per_instance=ends_fn(b), constraints=cons)
            (-ll).backward()
        torch.cuda.synchronize()

    pc = pack_batches(batches, model.device, m.max_k, constraints_fn=cons_fn, additional_ends_fn=ends_fn)

    def packed():
        m.zero_grad()
        ll = m.log_likelihood_packed(pc)
        (-ll.mean()).backward()
        torch.cuda.synchronize()

    def kernels_only():
        """the two DP launches + marginal kernels alone, HIP events on the stream (no torch glue)"""
        from action_segmentation_amd import ops
        t = pc.tables
        tt = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in t.items()}
        ws = torch.empty(pc.batch.workspace_bytes(), dtype=torch.uint8, device=model.device)
        elp64, _ = ops.emission(pc.batch, pc.x, tt['w'], tt['cst'], tt['inv_var'], cons=pc.cons)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        st = torch.cuda.current_stream()
        e[0].record(st)
        # (as the autograd function launches them: forward and time-reversed recursion in ONE launch, then the marginals)
        z = ops.logz(pc.batch, elp64, tt['trans'], tt['init'], tt['len'], endpen=pc.endpen, ws=ws, with_backward=True)
        e[1].record(st)
        ops.logz_bwd(pc.batch, elp64, tt['trans'], tt['init'], tt['len'], z, endpen=pc.endpen, ws=ws, with_backward=True)
        e[2].record(st)
        torch.cuda.synchronize()
        return e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])

    out = {}
    for name, fn in (('packed', packed), ('per_batch', per_batch)):
        fn()
        dts = []
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            dts.append(time.perf_counter() - t0)
        dt = sorted(dts)[1]                                   # median of three passes
        out[name] = {"value": frames / dt, "unit": "frames/s", "ms": dt * 1e3, "passes_ms": [round(d * 1e3, 3) for d in dts]}
    kernels_only()
    fwd_ms, bwd_ms = kernels_only()
    # algorithmic HBM bytes of the forward + backward DP: elp 8C read twice, two histories 24C each written,
    # both read by the marginals, g_elp 8C written
    c_sum = sum(ln * pc.n_states[g] for ln, g in zip(pc.lengths, pc.group))
    lz_bytes = c_sum * (8 + 24) * 2 + c_sum * (48 + 8)
    cells = sum(ln * ((min(kp, ln + 1) - 1) * pc.n_states[g] + pc.n_states[g] ** 2)
                for ln, kp, g in zip(pc.lengths, pc.kp, pc.group))
    out.update(value=out['packed']['value'], unit="frames/s", batches=len(batches),
               ms_per_batch=out['per_batch']['ms'] / len(batches),
               kernels={"logz_fwd_ms": fwd_ms, "logz_bwd_ms": bwd_ms,
                        "roofline": {"bound": "hbm", "achieved": lz_bytes / ((fwd_ms + bwd_ms) * 1e-3) / 1e9,
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": lz_bytes / ((fwd_ms + bwd_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "algorithmic_bytes": lz_bytes,
                                     "note": "latency/VALU-bound like the Viterbi kernel: %.3g lattice cells x 2 passes" % cells}})
    return out


def logz_cpu_baseline(pc, budget_s=10.0):
    """cpu_baseline of the log-partition leg (kind 'port'): oracle/smm_oracle.c forward + exact backward (posteriors),
    OpenMP over the videos of a task, on whole tasks until ~budget_s are spent.  The twin's logZ and gradients are
    compared with smm_logz_f64 / smm_logz_bwd_f64 on the same packed corpus (second return value: max relative error
    of logZ, max absolute error of the four gradients; tolerances of the path: 1e-6 / 2e-5)."""
    from oracle import factored as F
    from action_segmentation_amd import ops
    t = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in pc.tables.items()}
    elp_dev, _ = ops.emission(pc.batch, pc.x, t['w'], t['cst'], t['inv_var'], cons=pc.cons)
    ws = torch.empty(pc.batch.workspace_bytes(), dtype=torch.uint8, device=pc.x.device)
    z_dev = ops.logz(pc.batch, elp_dev, t['trans'], t['init'], t['len'], endpen=pc.endpen, ws=ws, with_backward=True)
    g_dev = ops.logz_bwd(pc.batch, elp_dev, t['trans'], t['init'], t['len'], z_dev, endpen=pc.endpen, ws=ws, with_backward=True)
    torch.cuda.synchronize()
    z_gpu = z_dev.cpu().numpy()
    g_gpu = {k: v.cpu().numpy() for k, v in g_dev.items()}
    z_rel, g_abs, g_tol, g_rel, n_checked = 0.0, 0.0, 0.0, 0.0, 0

    def cmp(got, ref):
        """max |got - ref|, max of |got - ref| / (2e-5 + 2e-5 |ref|) (<= 1: inside the tests' rtol = atol = 2e-5) and
        max of |got - ref| / max(1, |ref|) (the path's tolerance, SURVEY 8c(3): 1e-4 on posteriors; a gradient entry is a
        sum of posteriors, an expected count of up to ~1e3)"""
        err = np.abs(got - ref)
        return (float(err.max()), float((err / (2e-5 + 2e-5 * np.abs(ref))).max()),
                float((err / np.maximum(1.0, np.abs(ref))).max()))
    cores = F.set_threads(F.host_cores())
    by_group = {}
    for i in range(pc.n_videos):
        by_group.setdefault(pc.group[i], []).append(i)
    inv_var = t['inv_var'].cpu().numpy()
    frames, dt, n_vid = 0, 0.0, 0
    for g, vids in sorted(by_group.items()):
        c = pc.n_states[g]
        tmax = max(pc.lengths[i] for i in vids)
        kp = max(pc.kp[i] for i in vids)
        w, cst = t['w'][g, :, :c].cpu().numpy(), t['cst'][g, :c].cpu().numpy()
        xs = [pc.x[pc.frame_offset[i]:pc.frame_offset[i] + pc.lengths[i]].cpu().numpy() for i in vids]
        cons = None if pc.cons is None else [pc.cons[pc.frame_offset[i]:pc.frame_offset[i] + pc.lengths[i], :c].cpu().numpy() for i in vids]
        ep = None if pc.endpen is None else pc.endpen[vids][:, :c].cpu().numpy()
        t0 = time.perf_counter()
        elp = np.zeros((len(vids), tmax, c))
        for j, x in enumerate(xs):
            xd = x.astype(np.float64)
            elp[j, :x.shape[0]] = cst + xd @ w - 0.5 * (xd * xd) @ inv_var[:, None]
            if cons is not None:
                elp[j, :x.shape[0]] += cons[j]
        z_ref, g_ref = F.logz(elp, [pc.lengths[i] for i in vids], t['trans'][g, :c, :c].cpu().numpy(),
                              t['init'][g, :c].cpu().numpy(), t['len'][g, :kp, :c].cpu().numpy(), ep, grad=True)
        dt += time.perf_counter() - t0
        # parity (outside the CPU timing): one group = one task here, so the twin's table gradients of this call are the
        # GPU's rows of group g
        z_rel = max(z_rel, float(np.max(np.abs(z_gpu[vids] - z_ref) / np.abs(z_ref))))
        pairs = [(g_gpu['elp'][pc.frame_offset[i]:pc.frame_offset[i] + pc.lengths[i], :c], g_ref['elp'][j, :pc.lengths[i]])
                 for j, i in enumerate(vids)]
        if sum(1 for q in range(pc.n_videos) if pc.group[q] == g) == len(vids):
            pairs += [(g_gpu['trans'][g, :c, :c], g_ref['trans']), (g_gpu['init'][g, :c], g_ref['init']),
                      (g_gpu['len'][g, :kp, :c], g_ref['len'])]
        for got, ref in pairs:
            ea, et, er = cmp(got, ref)
            g_abs, g_tol, g_rel = max(g_abs, ea), max(g_tol, et), max(g_rel, er)
        n_checked += len(vids)
        frames += sum(pc.lengths[i] for i in vids)
        n_vid += len(vids)
        if dt > budget_s:
            break
    res = {"value": frames / dt, "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": "%d videos, oracle/smm_oracle.c log-partition forward + exact backward, OpenMP over the videos of a "
                     "task (%d host threads), %.1f s" % (n_vid, cores, dt)}
    return res, {"logz_videos_checked": n_checked, "logz_max_rel": z_rel, "grad_max_abs": g_abs,
                 "grad_max_rel": g_rel, "grad_max_err_over_tol": g_tol,
                 "grad_tolerance": "grad_max_rel = max |got - ref| / max(1, |ref|): the path's tolerance is 1e-4 on posteriors "
                                   "(SURVEY 8c(3); a gradient entry is a sum of posteriors, an expected count of up to ~1e3); "
                                   "grad_max_err_over_tol measures against the unit tests' tighter rtol = atol = 2e-5"}


def pmc_traffic(workload):
    """HBM bytes per launch of the DP kernel from the committed rocprofv3 PMC summary -- a STORED value (PMC passes
    cannot run inside this process), returned with its provenance; None when there is none for this workload."""
    path = os.path.join(ROOT, 'profiles', 'pmc_summary.json')
    try:
        rec = json.load(open(path)).get(workload, {})
        val = rec.get('smm_viterbi_kernel_hbm_bytes_per_launch')
        if val is None:
            return None, None
        return val, "stored: profiles/pmc_summary.json[%s] (%s)" % (workload, rec.get('_source', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes'))
    except Exception:
        return None, None


def weak_workload_text(a, cfg, lengths, n_states):
    """``config.workload`` of the weak leg: a function of the workload name, the seed and rank 0's corpus STRUCTURE only, so
    that the N = 1 line, the N > 1 lines and the host-only rehearsal all print the same string (tests/test_bench_spawn.py)."""
    lengths = [int(t) for t in lengths]
    return ("%s seed %d: %d tasks x %d videos per GPU, %d frames per GPU (T %d..%d), %d..%d states per "
            "task (mean over videos %.1f; per task: %s), max span length %d, D=%d; closed-form-fitted HSMM parameters"
            % (a.workload, a.seed, cfg['n_tasks'], len(lengths) // cfg['n_tasks'], sum(lengths), min(lengths),
               max(lengths), min(n_states), max(n_states), float(np.mean(n_states)),
               ' '.join(str(c) for c in sorted(set_per_task(n_states, cfg))), cfg['max_k'] - 1, cfg['d']))


def set_per_task(n_states_per_video, cfg):
    """states per task from states per video (videos of a task are consecutive and equally many)."""
    per = len(n_states_per_video) // cfg['n_tasks']
    return [n_states_per_video[i * per] for i in range(cfg['n_tasks'])]


def dry_run(a, rank, world, D):
    """Host-only rehearsal of the N-rank path (tests/test_bench_spawn.py): shard the corpus STRUCTURE, reduce the frame
    counters over gloo, print the line with value null."""
    from action_segmentation_amd import synth
    from action_segmentation_amd.batching import batch_cost
    wl = a.strong_workload if a.strong_workload in synth.CONFIGS else 'tiny'
    cfg = synth.CONFIGS[wl]
    dry = synth.SynthDatasplit(wl, seed=a.seed, keep=set())
    batches = dry.batch_sampler(cfg['batch_size'], True, False).batches
    costs = [batch_cost(dry, keys, cfg['max_k']) for keys in batches]
    mine = D.shard_batches(batches, costs, rank, world)
    frames = sum(int(dry[key]['features'].shape[0]) for i in mine for key in batches[i])
    red = D.all_reduce_counters({'frames': [frames, len(mine)], 'ranks': [1, rank]})
    if rank == 0:
        # the headline leg's workload string, from rank 0's corpus structure alone (no features, no GPU)
        wcfg = synth.CONFIGS[a.workload]
        wdry = synth.SynthDatasplit(a.workload, seed=a.seed, keep=set(), scale=a.scale)
        w_len, w_c = [], []
        for task, names in wdry._videos_by_task.items():
            for nm in names:
                w_len.append(int(wdry[(task, nm)]['features'].shape[0]))
                w_c.append(len(wdry.corpus._indices_by_task[task]))
        weak_text = weak_workload_text(a, wcfg, w_len, w_c)
        strong_text = "%s seed %d: %d videos, %d frames, sharded by video (whole single-task batches of %d, greedy LPT on the DP work)" % (
            wl, a.seed, len(dry), dry.n_frames, cfg['batch_size'])
        print(json.dumps({"metric": METRIC, "value": None, "unit": "frames/s", "n_gpus": world, "dry_run": True,
                          "scaling": a.scaling,
                          "config": {"workload": weak_text if a.scaling == 'weak' else strong_text},
                          "strong_scaling": {"workload": strong_text, "scaling": "strong", "n_gpus": world},
                          "frames": int(red['frames'][0]), "batches": int(red['frames'][1]), "ranks_seen": int(red['ranks'][0]),
                          "corpus_frames": dry.n_frames, "n_batches": len(batches), "backend": "gloo" if world > 1 else None}),
              flush=True)


def main():
    argv = sys.argv[1:]
    a = parse(argv)
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(a, argv))                      # before any GPU call in this process
    torch.set_num_threads(min(8, os.cpu_count() or 1))   # host-side torch ops are tiny: a 256-thread pool only adds latency
    os.environ.setdefault('OMP_NUM_THREADS', '16')          # (any OpenMP runtime loaded later: not one spinning thread per visible CPU)
    rank, world, local, backend = dist_setup(a)
    from action_segmentation_amd import distributed as D
    if a.dry_run:
        dry_run(a, rank, world, D)
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    dev = torch.device('cuda', torch.cuda.current_device())
    from action_segmentation_amd import evaluation, ops, synth

    # ---------------------------------------------------------------- weak leg: every rank its own corpus
    cfg = synth.CONFIGS[a.workload]
    data = synth.SynthDatasplit(a.workload, seed=a.seed + rank, device=dev, scale=a.scale)
    args, model = fit_model(a, cfg, data, dev, D, world)
    pc = model.prepare(data)                                         # inputs resident in HBM from here on
    frames = pc.n_frames
    stream = torch.cuda.current_stream()
    dt, dp, labels = timed_decode(a, pc, world)
    dp_ms = dp["per_step_sum_ms"]                                    # DP kernel time of a step, all its launches
    labels_dev = labels.to(dev)      # the evaluation kernels below (outside the timed region) read device labels

    # evaluation, outside the timed region: the reference's per-task statistics from device counters
    # of THIS rank's corpus (weak scaling: the corpora differ per rank; the strong leg reduces one corpus's counters)
    space = evaluation.LabelSpace.from_corpus(data.corpus, list(data._videos_by_task))
    gt_dev = gt_on_device(pc, data, dev)
    eval_ms = []
    for _ in range(3):
        torch.cuda.synchronize()
        e0 = time.perf_counter()
        stats_by_task = evaluation.evaluate_labels(labels_dev, gt_dev, pc.lengths, pc.frame_offset, pc.task_names, space,
                                                   optimal_assignment=False, seed=0, video_key=video_keys(pc, data))
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
    assert abs(summary['mof'] - correct / frames) < 1e-12, "device MoF != host MoF"
    rd = D.reduce_device()
    counters = D.all_reduce_tensor(torch.tensor([float(correct), float(frames), float(frames)], dtype=torch.float64, device=rd))
    tmax = D.all_reduce_tensor(torch.tensor([dt], dtype=torch.float64, device=rd),
                               op=torch.distributed.ReduceOp.MAX if D.active() else None)
    total_frames = float(counters[2])
    dt = float(tmax[0])

    strong = None
    if not a.no_strong_leg:
        del labels_dev, gt_dev
        strong = strong_leg(a, rank, world, dev, D)

    if rank == 0:
        c_avg = float(np.mean([pc.n_states[g] for g in pc.group]))
        cells = sum(ln * ((min(kp, ln + 1) - 1) * pc.n_states[g] + pc.n_states[g] ** 2)
                    for ln, kp, g in zip(pc.lengths, pc.kp, pc.group))
        # algorithmic HBM bytes of the DP kernel per frame: elp in 8C, history out 24C, label out 8
        dp_bytes = sum(ln * (32 * pc.n_states[g] + 8) for ln, g in zip(pc.lengths, pc.group))
        achieved = dp_bytes / (dp_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(a.workload)
        par = {None: "1 rank, no collective", "nccl": "RCCL", "gloo": "gloo (rehearsal: NOT a measurement of the RCCL path)"}[backend]
        weak = {"value": total_frames * a.steps / dt, "ms_per_step": dt / a.steps * 1e3}
        head = weak if (a.scaling == 'weak' or strong is None) else strong
        by_name = {nm: (ln, pc.n_states[g]) for nm, ln, g in zip(pc.video_names, pc.lengths, pc.group)}
        in_order = [by_name[nm] for names in data._videos_by_task.values() for nm in names]
        wl_text = weak_workload_text(a, cfg, [t for t, _ in in_order], [c for _, c in in_order])
        if head is not weak:
            wl_text = head["workload"]
        res = {
            "metric": METRIC,
            "value": head["value"], "unit": "frames/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
            "scaling": "weak" if head is weak else "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic", "backend": backend,
            "config": {"workload": wl_text,
                       "parallelism": "videos sharded across %d GPU(s), no data-path collective; step time MAX and frame counters "
                                      "SUM all-reduced over: %s" % (world, par)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "smm_viterbi_kernel", "kernel_ms": dp["launch_ms"], "launches_per_step": dp["launches_per_step"],
                         "kernel_ms_per_step": dp_ms, "kernel_ms_longest_launch": dp["max_launch_ms"],
                         "critical_launch_ms": dp["critical_launch_ms"], "rest_launch_ms": dp["rest_launch_ms"],
                         "small_wg_launch_ms": dp.get("small_wg_launch_ms"),
                         "frac_wall": dp_bytes / (weak["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "kernel_ms_source": "HIP events recorded by the library (smm_dp_timing_*) around every DP kernel launch "
                                             "of the K timed steps, on the stream each launch runs on: smm_decode_f32 launches "
                                             "the kernel twice per step on a corpus whose longest videos set the DP's time (those "
                                             "videos first, the others on a second stream beside them: smm_api.hip choose_split); "
                                             "kernel_ms is the mean launch (what rocprofv3 --stats averages), achieved = the "
                                             "step's algorithmic bytes / the step's summed launch time; critical_launch_ms / "
                                             "rest_launch_ms: the means per stream (the critical launch is what must fit into "
                                             "ms_per_step, and bench.py asserts that it does); frac_wall = the same bytes / the "
                                             "step's wall time / peak",
                         "algorithmic_bytes_per_launch": dp_bytes / dp["launches_per_step"],
                         "note": "the DP is latency-bound, not HBM-bound: one serial chain per video (a few hundred cycles per "
                                 "position), and the launch lasts as long as its longest video; %.3g lattice cells per step, "
                                 "of which the BAND kernel evaluates the part its bound tests cannot exclude" % cells},
            "time_split": dp.get("time_split"),
            "mof": float(counters[0] / counters[1]),
            "weak_scaling": dict(weak, scaling="weak", n_gpus=world,
                                 note="every rank decodes its own corpus of this size (seed + rank)"),
            "evaluation": {"ms": min(eval_ms), "frames_per_s": frames / (min(eval_ms) * 1e-3),
                           "what": "accuracy_corpus statistics of rank 0's corpus (confusion + per-video counters on the "
                                   "device, assignment and ratios on the host), all tasks, outside the timed decode",
                           "stats": {k: round(v, 9) for k, v in summary.items()}},
            "fit_stats": {"ms": min(fit_ms[1:]), "frames_per_s": frames / (min(fit_ms[1:]) * 1e-3),
                          "roofline": {"bound": "hbm", "achieved": fit_bytes / (min(fit_ms[1:]) * 1e-3) / 1e9,
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": fit_bytes / (min(fit_ms[1:]) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "algorithmic_bytes_per_launch": fit_bytes},
                          "what": "smm_fit_stats_f64 (class sums + span statistics) over every frame of the workload"},
        }
        # the launches of a split decode overlap: their sum may exceed the step, the critical one may not
        assert dp["critical_launch_ms"] is None or dp["critical_launch_ms"] <= weak["ms_per_step"] * 1.001, \
            (dp["critical_launch_ms"], weak["ms_per_step"])
        if strong is not None:
            res["strong_scaling"] = strong
            if head is strong and strong.get("roofline_rank0"):
                res["roofline_weak_leg"] = res["roofline"]
                res["roofline"] = strong["roofline_rank0"]
        # (before any CPU-baseline leg: the oracle's OpenMP pool -- one spinning thread per host core -- and torch's own
        # intra-op pool would fight over the cores, and predict()'s host work would be timed under that contention:
        # round 2 reported 4 s for a 25 ms call this way)
        if world == 1 and not a.no_predict_e2e:
            res["predict_end_to_end"] = predict_end_to_end(model, data)
            try:
                res["host_features"] = host_features_leg(a, cfg, model, data, lab, pc)
            except Exception as e:
                res["host_features"] = {"error": repr(e)}
            if a.workload == 'cfg3':
                try:
                    res["reference_default"] = reference_default_leg(a, dev, D, not a.no_cpu_baseline)
                except Exception as e:
                    res["reference_default"] = {"error": repr(e)}
        parity = {"frames_checked": 0, "label_mismatches": None, "logz_max_rel": None, "grad_max_abs": None,
                  "what": "the C twin (oracle/smm_oracle.c) against the GPU on this workload, checker side only: frame "
                          "labels of the TIMED decode for every video the cpu_factored leg covered; cfg4: logZ and the "
                          "four gradients of smm_logz_f64 / smm_logz_bwd_f64 for the videos of the logZ baseline"}
        if a.workload == 'cfg4':
            res["logz_fwd_bwd"] = train_step_rate(args, data, model)
            if not a.no_cpu_baseline:
                try:
                    res["logz_fwd_bwd"]["cpu_baseline"], zp = logz_cpu_baseline(pc)
                    parity.update(zp)
                except Exception as e:
                    res["logz_fwd_bwd"]["cpu_baseline"] = {"error": str(e)}
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(data, model, pc)
            try:
                res["cpu_factored"], lp = cpu_factored(pc, model, gpu_labels=lab)
                parity.update(lp)
            except Exception as e:                              # the C oracle needs gcc on the box; report, don't fail
                res["cpu_factored"] = {"error": str(e)}
        res["parity"] = parity
        if world == 1 and a.workload == 'cfg3' and a.second_seed >= 0 and a.second_seed != a.seed and a.scale == 1.0:
            # the other draw of states per task (not the headline: same shapes, friendlier state counts)
            data2 = synth.SynthDatasplit(a.workload, seed=a.second_seed, device=dev)
            _, model2 = fit_model(a, cfg, data2, dev, D, world)
            pc2 = model2.prepare(data2)
            dt2, dp2, _ = timed_decode(a, pc2, world)
            dp2 = dp2["per_step_sum_ms"] if dp2 else None
            res["other_draw"] = {"seed": a.second_seed, "value": pc2.n_frames * a.steps / dt2, "unit": "frames/s",
                                 "ms_per_step": dt2 / a.steps * 1e3, "dp_kernel_ms": dp2, "frames": pc2.n_frames,
                                 "states_per_task": ' '.join(str(c) for c in sorted(pc2.n_states))}
        print(json.dumps(res), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
