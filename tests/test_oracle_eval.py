"""The restated evaluation counters (oracle/eval_ref.py) against the outputs of the reference's own ``Accuracy``
(tests/golden/eval_vectors.json, made by tests/golden/make_golden_eval.py)."""
import json
import os

import numpy as np
import pytest

from oracle import eval_ref

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, 'golden', 'eval_vectors.json')) as f:
    GOLD = json.load(f)
CASES = sorted(GOLD['cases'])


def check_stat(got, want, random_keys, deterministic_draws):
    assert set(got) == set(want)
    for key, pair in want.items():
        if key in random_keys and not deterministic_draws:
            assert got[key][1] == pair[1], key                      # denominators never depend on the draw
            continue
        np.testing.assert_allclose(np.asarray(got[key], dtype=np.float64), np.asarray(pair), rtol=1e-12, atol=0,
                                   err_msg=key)


@pytest.mark.parametrize('name', CASES)
def test_counters_match_reference(name):
    case = GOLD['cases'][name]
    inp, exp = case['inputs'], case['expected']
    stat, extras = eval_ref.task_counters(inp['gt'], inp['pred'], inp['background'], inp['possible'], inp['optimal'])
    check_stat(stat, exp['stat'], GOLD['random_keys'], name in ('perfect', 'single_segment'))
    assert extras['frames'] == exp['frames']
    assert {int(k): v for k, v in exp['gt2cluster'].items()} == {k: v for k, v in extras['gt2cluster'].items() if v}
    for k, v in exp['classes_mof'].items():
        assert extras['classes_mof'][int(k)] == v
    for k, v in exp['classes_iou'].items():
        assert extras['classes_iou'][int(k)] == v


def test_edit_distance_known_answers():
    assert eval_ref.edit_distance([], []) == 0
    assert eval_ref.edit_distance([1, 2, 3], []) == 3
    assert eval_ref.edit_distance([1, 2, 3], [1, 3]) == 1
    assert eval_ref.edit_distance(list('kitten'), list('sitting')) == 3
    assert eval_ref.edit_distance([5, 5, 5], [6, 6, 6]) == 3


def test_draw_is_roughly_uniform():
    counts = np.zeros(8)
    for v in range(4000):
        counts[min(range(8), key=lambda t: (eval_ref.frame_hash(3, v, t), t))] += 1
    assert counts.min() > 400 and counts.max() < 600


@pytest.mark.parametrize('name', sorted(GOLD['datasplit_cases']))
def test_datasplit_level_counters_match_reference(name):
    """The restated ``Datasplit.accuracy_corpus`` (multi-label ground truth, --frame_subsample re-expansion, background
    canonicalisation) against the reference's own method run on a stand-in datasplit (make_golden_eval.py)."""
    case = GOLD['datasplit_cases'][name]
    inp, exp = case['inputs'], case['expected']
    got = eval_ref.datasplit_counters(inp['tasks'], inp['background'], inp['videos'], inp['subsample'],
                                      inp['annotate_background_with_previous'], inp['optimal'])
    assert set(got) == set(exp)
    for task in exp:
        check_stat(got[task], exp[task], GOLD['random_keys'], False)
