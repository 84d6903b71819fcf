"""Minimal driver with the reference's flag surface for the semi-Markov path (reference ``src/main.py``).

    python -m action_segmentation_amd.cli --classifier semimarkov --training supervised --cuda \\
        --dataset synthetic:cfg4 --sm_max_span_length 64 [--model_output_path DIR | --model_input_path DIR] ...

Every flag the reference's launchers pass (``run_crosstask_*.sh``, ``decode*.sh``, README rows S6 / U7) parses here;
the dataset readers are out of scope, so ``--dataset`` takes ``synthetic:<config>`` (``synth.CONFIGS``)
-- with ``crosstask`` / ``breakfast`` the driver stops with a message, and a maintainer instead points the reference's own
``main.py`` at this package.  Like the reference it prints the command line first (``decode*.sh``
greps it back out of ``log.txt``), trains, evaluates MoF, and pickles / unpickles the whole model object.
"""
import argparse
import os
import pickle
import sys

import numpy as np
import torch

from . import synth
from .batching import add_training_args
from .evaluation import STAT_KEYS, accuracy_corpus, summarise
from .semimarkov import SemiMarkovModel

CLASSIFIERS = {'semimarkov': SemiMarkovModel}


def build_parser():
    p = argparse.ArgumentParser(fromfile_prefix_chars='@')
    g = p.add_argument_group('serialization')
    g.add_argument('--model_output_path')
    g.add_argument('--model_input_path')
    g.add_argument('--prediction_output_path')
    g = p.add_argument_group('data')
    g.add_argument('--dataset', default='synthetic:tiny')
    g.add_argument('--features', choices=['raw', 'pca'], default='pca')
    g.add_argument('--feature_downscale', type=float, default=1.0)
    g.add_argument('--feature_permutation_seed', type=int)
    g.add_argument('--batch_size', type=int, default=5)
    g.add_argument('--remove_background', action='store_true')
    g.add_argument('--pca_components_per_group', type=int, default=100)
    g.add_argument('--pca_no_background', action='store_true')
    g.add_argument('--mix_tasks', action='store_true')
    g.add_argument('--frame_subsample', type=int, default=1)
    g.add_argument('--task_specific_steps', action='store_true')
    g.add_argument('--annotate_background_with_previous', action='store_true')
    g.add_argument('--no_merge_classes', action='store_true')
    g.add_argument('--force_optimal_assignment', action='store_true')
    g.add_argument('--no_cache_features', action='store_true')
    g.add_argument('--crosstask_feature_groups', choices=['i3d', 'resnet', 'audio', 'narration'], nargs='+',
                   default=['i3d', 'resnet', 'audio'])
    g.add_argument('--crosstask_training_data', choices=['primary', 'related'], nargs='+', default=['primary'])
    g.add_argument('--crosstask_cross_validation', action='store_true')
    g.add_argument('--crosstask_cross_validation_seed', type=int)
    g = p.add_argument_group('classifier')
    g.add_argument('--classifier', choices=sorted(CLASSIFIERS), required=True)
    g.add_argument('--training', choices=['supervised', 'unsupervised'], default='supervised')
    g.add_argument('--cuda', action='store_true')
    g.add_argument('--seed', type=int, default=0)
    for cls in CLASSIFIERS.values():
        cls.add_args(p)
    add_training_args(p)
    return p


def optimal_assignment_for(args):
    """main.py:126-135: Hungarian re-assignment only for unsupervised training without ordering constraints."""
    if args.force_optimal_assignment:
        return True
    if args.training == 'supervised':
        return False
    if args.sm_constrain_transitions:
        return False
    return not ('train' in args.sm_constrain_with_narration or 'test' in args.sm_constrain_with_narration)


def _reduce():
    """Summation over the ranks of a torchrun job for the evaluation counters (None for one process)."""
    from . import distributed
    return distributed.all_reduce_tensor if distributed.active() else None


def evaluate(model, data, name, args=None, verbose=True):
    """main.py:125-160 ``test``: decode, then the per-task statistics of ``accuracy_corpus`` summed over tasks.
    Under torchrun every rank decodes its shard of the videos and the counters are all-reduced (RCCL) first."""
    preds = model.predict(data)
    by_task = accuracy_corpus(data, preds, optimal_assignment_for(args) if args is not None else False,
                              seed=getattr(args, 'seed', 0) if args is not None else 0, reduce=_reduce())
    stats = summarise(by_task, STAT_KEYS, prefix=name + '_')
    if verbose and int(os.environ.get('RANK', 0)) == 0:
        print(', '.join(STAT_KEYS))
        print(', '.join('%.4f' % stats[name + '_' + k] for k in STAT_KEYS))
    return preds, stats


def train(args, train_data, dev_data, split_name, verbose=False, train_sub_data=None):
    """main.py:163-264: fit with a per-epoch callback that decodes train (or a train subset) and -- every
    ``--dev_decode_frequency`` epochs -- dev, keeps a ``pickle.dumps`` snapshot of the model per epoch, writes
    ``<out>/<split>_epoch-N.pkl`` every 5 epochs, and finally selects the best snapshot: best dev MoF for supervised
    training (early stopping on dev), lowest training loss for unsupervised training.  The closed-form supervised fit
    has no epochs and never calls back (semimarkov.py:165-171); its model is returned as is."""
    model = CLASSIFIERS[args.classifier].from_args(args, train_data)
    use_labels = args.training == 'supervised'
    early_stopping_on_dev = use_labels
    models_by_epoch, dev_mof_by_epoch, stats_by_epoch = {}, {}, {}
    rank0 = int(os.environ.get('RANK', 0)) == 0

    def callback_fn(epoch, stats):
        stats_by_epoch[epoch] = stats
        name = 'train_subset' if train_sub_data is not None else 'train'
        split_stats = [evaluate(model, train_sub_data if train_sub_data is not None else train_data, name, args,
                                verbose=verbose)[1]]
        dev_stats = None
        if epoch == -1 or epoch % args.dev_decode_frequency == 0:
            dev_stats = evaluate(model, dev_data, 'dev', args, verbose=verbose)[1]
            split_stats.append(dev_stats)
        log = '%s\tepoch %2d' % (split_name, epoch)
        for stat, value in stats.items():
            log += ('\t%s %.4f' % (stat, value)) if isinstance(value, float) else ('\t%s %s' % (stat, value))
        for st in split_stats:
            log += '\n' + ''.join(' %s %.4f' % (k, v) for k, v in sorted(st.items()))
        if rank0:
            print(log)
        models_by_epoch[epoch] = pickle.dumps(model)
        if dev_stats is not None:
            dev_mof_by_epoch[epoch] = dev_stats['dev_mof']
        if args.model_output_path and epoch % 5 == 0 and rank0:
            os.makedirs(args.model_output_path, exist_ok=True)
            fname = os.path.join(args.model_output_path, '%s_epoch-%d.pkl' % (split_name, epoch))
            print("writing model to %s" % fname)
            with open(fname, 'wb') as f:
                pickle.dump(model, f)

    model.fit(train_data, use_labels=use_labels, callback_fn=callback_fn)
    if early_stopping_on_dev and dev_mof_by_epoch:
        best_epoch, best_mof = max(dev_mof_by_epoch.items(), key=lambda t: t[1])
        if rank0:
            print("best dev mof %.4f in epoch %d" % (best_mof, best_epoch))
        best = pickle.loads(models_by_epoch[best_epoch])
    elif stats_by_epoch and 'train_loss' in next(iter(stats_by_epoch.values())):
        best_epoch, best_stats = min(stats_by_epoch.items(), key=lambda t: t[1]['train_loss'])
        if rank0:
            print("best train loss %.4f in epoch %d" % (best_stats['train_loss'], best_epoch))
        best = pickle.loads(models_by_epoch[best_epoch])
    else:
        best = model
    train.last_history = dict(stats_by_epoch=stats_by_epoch, dev_mof_by_epoch=dev_mof_by_epoch)
    if best is not model and args.cuda:
        best.model.cuda()
    if args.model_output_path and rank0:
        os.makedirs(args.model_output_path, exist_ok=True)
        fname = make_model_path(args.model_output_path, split_name)
        print("writing model to %s" % fname)
        with open(fname, 'wb') as f:
            pickle.dump(best, f)
    return best


def make_model_path(path, split_name):
    return path if path.endswith('.pkl') else os.path.join(path, '%s.pkl' % split_name)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    print(' '.join([sys.argv[0]] + list(argv)))            # decode*.sh recovers the command from the log
    args = build_parser().parse_args(argv)
    if not args.dataset.startswith('synthetic:'):
        raise SystemExit("dataset readers (%s) are outside this build; use --dataset synthetic:<%s>, or register "
                         "action_segmentation_amd.semimarkov.SemiMarkovModel in the reference's main.py (INTEGRATION.md)"
                         % (args.dataset, '|'.join(synth.CONFIGS)))
    if not args.cuda:
        raise SystemExit("--cuda is required: the semi-Markov path has no CPU back-end in this build")
    from . import distributed
    from ._lib import host_cores
    # the host side of this path is glue around kernels (tiny tensors): a pool of one thread per VISIBLE CPU only adds
    # wake-up latency, and in a container with a CPU quota it gets the process throttled
    torch.set_num_threads(max(1, min(8, host_cores())))
    # torchrun: one rank per GPU over RCCL, decode sharded by video, training data-parallel (SMM_DIST_BACKEND=gloo: a
    # rehearsal with several ranks on one GPU)
    distributed.init(backend=os.environ.get('SMM_DIST_BACKEND') or None)
    cfg_name = args.dataset.split(':', 1)[1]
    split_name = 'synthetic'
    torch.manual_seed(args.seed)
    dev = torch.device('cuda', torch.cuda.current_device())
    train_data = synth.SynthDatasplit(cfg_name, seed=args.seed, device=dev)
    test_data = synth.SynthDatasplit(cfg_name, seed=args.seed, video_seed=1, device=dev)     # same label space
    # the datasplits are thousands of long-lived tensors: out of the collector's sight, or every full collection a decode's
    # few hundred result arrays trigger walks all of them (60-90 ms beside a 3 ms predict: scripts/probe_predict_fused.py)
    import gc
    gc.collect()
    gc.freeze()
    if args.model_input_path:
        with open(make_model_path(args.model_input_path, split_name), 'rb') as f:
            model = pickle.load(f)
        model.args = args
        model.model.args = args
        model.model.cuda()
        model.model.eval()
    else:
        model = train(args, train_data, test_data, split_name)
    evaluate(model, train_data, 'train', args)
    preds, stats = evaluate(model, test_data, 'test', args)
    if args.prediction_output_path:
        os.makedirs(args.prediction_output_path, exist_ok=True)
        for video, pred in preds.items():
            np.save(os.path.join(args.prediction_output_path, video + '.npy'), pred)
    return stats


if __name__ == '__main__':
    main()
