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


def evaluate(model, data, name, args=None):
    """main.py:126-160 ``test``: decode, then the per-task statistics of ``accuracy_corpus`` summed over tasks."""
    preds = model.predict(data)
    by_task = accuracy_corpus(data, preds, optimal_assignment_for(args) if args is not None else False,
                              seed=getattr(args, 'seed', 0) if args is not None else 0)
    stats = summarise(by_task, STAT_KEYS, prefix=name + '_')
    print(', '.join(STAT_KEYS))
    print(', '.join('%.4f' % stats[name + '_' + k] for k in STAT_KEYS))
    return preds, stats


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
    cfg_name = args.dataset.split(':', 1)[1]
    torch.manual_seed(args.seed)
    dev = torch.device('cuda', torch.cuda.current_device())
    train = synth.SynthDatasplit(cfg_name, seed=args.seed, device=dev)
    test = synth.SynthDatasplit(cfg_name, seed=args.seed, video_seed=1, device=dev)          # same label space
    if args.model_input_path:
        with open(os.path.join(args.model_input_path, 'synthetic.pkl'), 'rb') as f:
            model = pickle.load(f)
        model.args = args
        model.model.args = args
        model.model.cuda()
        model.model.eval()
    else:
        model = CLASSIFIERS[args.classifier].from_args(args, train)
        history = []
        model.fit(train, use_labels=(args.training == 'supervised'),
                  callback_fn=lambda epoch, stats: history.append((epoch, stats)))
        for epoch, stats in history:
            print('epoch %d: %s' % (epoch, stats))
        if args.model_output_path:
            os.makedirs(args.model_output_path, exist_ok=True)
            with open(os.path.join(args.model_output_path, 'synthetic.pkl'), 'wb') as f:
                pickle.dump(model, f)
    evaluate(model, train, 'train', args)
    preds, stats = evaluate(model, test, 'test', args)
    if args.prediction_output_path:
        os.makedirs(args.prediction_output_path, exist_ok=True)
        for video, pred in preds.items():
            np.save(os.path.join(args.prediction_output_path, video + '.npy'), pred)
    return stats


if __name__ == '__main__':
    main()
