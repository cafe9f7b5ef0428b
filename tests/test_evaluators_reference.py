"""The evaluators against the REFERENCE's own outputs (tests/golden/evaluator.npz, produced by importing
/root/reference/evaluators.py in the build container: tests/golden/make_reference_fixtures.py).  Bit-exact (float64 ==):
the scores are ratios of integer counts, and the hot path's arg-max results feed them ("integer argmax bit-exact")."""
import os

import numpy as np
import pytest

from oracle import ops as O

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'evaluator.npz')


@pytest.fixture(scope='module')
def fix():
    return np.load(FIX)


def _cases(fix):
    return sorted({k.split('/')[0] for k in fix.files if k.endswith('/y_true')})


def test_fixture_is_reference_generated(fix):
    names = _cases(fix)
    assert len(names) >= 12 and any(n.startswith('seg_ids') for n in names) and any(n.startswith('cls_') for n in names)
    assert sum(1 for k in fix.files if k.endswith('/accuracy') and not k.startswith('is_better')) == len(names)


def test_oracle_accuracy_matches_reference(fix):
    for n in _cases(fix):
        assert O.accuracy_score(fix[n + '/y_true'], fix[n + '/y_pred']) == float(fix[n + '/accuracy']), n


def test_evaluators_match_reference_bit_exactly(fix):
    import myconvnet_amd.evaluators as E
    evs = {'accuracy': E.AccuracyEvaluator(), 'error': E.ErrorEvaluator(), 'top1': E.AccuracyTop1Evaluator(), 'top5': E.AccuracyTop5Evaluator(),
           'miou': E.MeanIoUEvaluator(), 'miou_bkgd0': E.MeanIoUEvaluator(bkgd_idx=0)}
    checked = 0
    for n in _cases(fix):
        yt, yp = fix[n + '/y_true'], fix[n + '/y_pred']
        for key, ev in evs.items():
            if n + '/' + key not in fix.files:
                continue
            got = ev.score(yt.copy(), yp.copy())
            assert float(got) == float(fix[n + '/' + key]), (n, key, got, float(fix[n + '/' + key]))
            checked += 1
    assert checked >= 50
    # an all-ignored image scores 1, ignored pixels do not count (the case the round-1 evaluator got wrong)
    assert float(fix['seg_n3_c5/accuracy']) != float(np.mean(np.where(np.isclose(fix['seg_n3_c5/y_true'].sum(-1), 1),
                                                                      fix['seg_n3_c5/y_true'].argmax(-1) == fix['seg_n3_c5/y_pred'].argmax(-1), True)))


def test_is_better_matches_reference(fix):
    import myconvnet_amd.evaluators as E
    acc, err = E.AccuracyEvaluator(), E.ErrorEvaluator()
    for (c, b), ra, re_ in zip(fix['is_better/pairs'], fix['is_better/accuracy'], fix['is_better/error']):
        assert bool(acc.is_better(c, b)) == bool(ra)
        assert bool(err.is_better(c, b)) == bool(re_)
    assert acc.worst_score == 0.0 and err.worst_score == 1.0 and acc.mode == 'max' and err.mode == 'min'
