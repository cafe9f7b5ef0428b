"""CPU test of the flip-aware comparison (tests/flip_util.py): a stand-in "device" = the float64 oracle with two ReLU decisions inverted
at its own near-ties.  The comparison must find exactly those decisions, follow them, and fail when the difference is not a decision."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, 'golden'))

import make_golden as MG  # noqa: E402
from flip_util import flip_aware_step, rel_l2  # noqa: E402
from oracle import net as ON  # noqa: E402


def _setup():
    spec = ON.ResNetSpec.resnet50(10, 8)
    params, stats = MG.net_params(spec)
    fresh = lambda: ON.TrainState({k: v.astype(np.float64) for k, v in params.items()}, {k: v.astype(np.float64) for k, v in stats.items()})  # noqa: E731
    rng = np.random.default_rng(0)
    towers = [(rng.random((4, 32, 32, 3)), rng.integers(0, 10, 4).astype(np.float64)) for _ in range(2)]
    return spec, fresh, towers


def test_flip_search_recovers_inverted_decisions_and_follows_them():
    spec, fresh, towers = _setup()
    tie_tol = 2e-4                                   # (wide, so that the tiny net has candidates; the GPU tests use 4e-6)
    probe = {'tie_tol': tie_tol}
    _, _, g0 = ON.train_step(spec, fresh(), None, None, batch_total=8, tower_batches=towers, probe=probe)
    ties = probe['near_ties']
    assert len(ties) >= 8
    # the two near-ties whose inversion moves the gradient most, in different towers if possible
    effect = []
    for t in ties:
        g = ON.train_step(spec, fresh(), None, None, batch_total=8, tower_batches=towers, probe={'relu_flips': {(t[0], t[1]): [t[2]]}})[2]
        effect.append(max(rel_l2(g[k], g0[k]) for k in g0))
    order = np.argsort(effect)[::-1]
    picked = [ties[order[0]], ties[order[1]]]
    assert effect[order[1]] > 1e-4                   # both are visible
    flips = {}
    for t in picked:
        flips.setdefault((t[0], t[1]), []).append(t[2])
    dev_state = fresh()
    _, _, gdev = ON.train_step(spec, dev_state, None, None, batch_total=8, tower_batches=towers, probe={'relu_flips': flips})
    assert max(rel_l2(gdev[k], g0[k]) for k in g0) > 1e-4
    st = fresh()
    loss, pred, g, found = flip_aware_step(spec, st, towers, gdev, batch_total=8, tol=1e-6, tie_tol=tie_tol, max_flips=6)
    assert sorted(found) == sorted(picked)
    assert max(rel_l2(g[k], gdev[k]) for k in gdev) <= 1e-6
    for k in st.params:                              # the oracle state advanced along the device's branch
        np.testing.assert_allclose(st.params[k], dev_state.params[k], rtol=0, atol=1e-12)


def test_flip_search_rejects_a_difference_that_is_no_decision():
    spec, fresh, towers = _setup()
    _, _, g0 = ON.train_step(spec, fresh(), None, None, batch_total=8, tower_batches=towers)
    bad = {k: v.copy() for k, v in g0.items()}
    k0 = sorted(bad)[3]
    bad[k0] = bad[k0] * 1.01                         # a 1 % scale error in one tensor
    with pytest.raises(AssertionError):
        flip_aware_step(spec, fresh(), towers, bad, batch_total=8, tol=1e-3, tie_tol=2e-4, max_flips=6)
    # and without any near-tie the message says so
    with pytest.raises(AssertionError, match='no ReLU near-tie'):
        flip_aware_step(spec, fresh(), towers, bad, batch_total=8, tol=1e-3, tie_tol=1e-12, max_flips=6)
