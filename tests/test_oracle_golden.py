"""The oracle against the committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py).
PARITY UNPINNED: the fixtures freeze the oracle, they do not come from TensorFlow."""
import os
import sys

import numpy as np
import pytest

from oracle import net as ON
from oracle import ops as O

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import make_golden as MG  # noqa: E402


@pytest.fixture(scope='module')
def gops():
    return np.load(os.path.join(HERE, 'golden', 'ops.npz'))


@pytest.fixture(scope='module')
def gnet():
    return np.load(os.path.join(HERE, 'golden', 'resnet50_w8.npz'))


def close(a, b, tol=2e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize('name', sorted(MG.CONV_CASES))
def test_conv_golden(gops, name):
    n, h, w, cin, cout, k, s, pad, dil = MG.CONV_CASES[name]
    x, wt, dy = [gops[name + '/' + t].astype(np.float64) for t in ('x', 'w', 'dy')]
    y = O.conv2d_fwd(x, wt, s, pad, dil)
    if 'biased' in name:
        y = y + gops[name + '/b']
        close(O.bias_add_bwd(dy), gops[name + '/db'])
    close(y, gops[name + '/y'])
    close(O.conv2d_dgrad(dy, wt, x.shape, s, pad, dil), gops[name + '/dx'])
    close(O.conv2d_wgrad(x, dy, wt.shape, s, pad, dil), gops[name + '/dw'])


def test_bn_pool_xent_sgd_golden(gops):
    g = gops
    y, bm, bv, sm, si = O.bn_fwd_train(g['bn/x'].astype(np.float64), g['bn/gamma'].astype(np.float64), g['bn/beta'].astype(np.float64), 1e-3)
    close(y, g['bn/y']); close(bm, g['bn/batch_mean']); close(bv, g['bn/batch_var']); close(si, g['bn/invstd'])
    dx, dg, db = O.bn_bwd(g['bn/dy'].astype(np.float64), g['bn/x'].astype(np.float64), g['bn/gamma'].astype(np.float64), sm, si)
    close(dx, g['bn/dx']); close(dg, g['bn/dgamma'], 1e-5); close(db, g['bn/dbeta'], 1e-5)
    for nm, (k, s) in {'mp3x3_s2': (3, 2), 'mp2x2_s2': (2, 2)}.items():
        y, arg = O.maxpool_fwd(g[nm + '/x'].astype(np.float64), k, s, 'SAME')
        np.testing.assert_array_equal(arg, g[nm + '/arg'])
        close(y, g[nm + '/y'])
        close(O.maxpool_bwd(g[nm + '/dy'].astype(np.float64), arg, g[nm + '/x'].shape, k, s, 'SAME'), g[nm + '/dx'])
    oh = O.one_hot_labels(g['xent/labels'], 10)
    np.testing.assert_array_equal(oh, g['xent/onehot'])
    pred, loss, ce, dl = O.softmax_xent_fwd_bwd(g['xent/logits'].astype(np.float64), oh.astype(np.float64))
    close(pred, g['xent/pred']); close(dl, g['xent/dlogits']); assert abs(loss - float(g['xent/loss'])) < 1e-12
    w, a, e = g['sgd/w0'].astype(np.float64), np.zeros(64), g['sgd/w0'].astype(np.float64)
    for t in range(3):
        w, a, e = O.sgd_nesterov_step(w, g['sgd/grads'][t].astype(np.float64), a, 0.1, 0.9, l2=1e-4, ema=e, ema_d=O.ema_decay(0.99, t))
        close(np.stack([w, a, e]), g['sgd/traj'][t], 1e-12)


def test_resnet_two_step_golden(gnet):
    spec = ON.ResNetSpec.resnet50(10, 8)
    params, stats = MG.net_params(spec)
    chk = np.array([sum(float(v.astype(np.float64).sum()) for v in params.values()), sum(float((v.astype(np.float64) ** 2).sum()) for v in params.values())])
    np.testing.assert_allclose(chk, gnet['param_checksum'], rtol=1e-12)          # the seeded parameter generator has not drifted
    state = ON.TrainState({k: v.astype(np.float64) for k, v in params.items()}, {k: v.astype(np.float64) for k, v in stats.items()})
    keys = [str(k) for k in gnet['keys']]
    for step in range(2):
        p = 'step{}/'.format(step)
        x = gnet[p + 'x_u8'].astype(np.float32) / np.float32(255)
        y = gnet[p + 'y']
        loss, pred, grads = ON.train_step(spec, state, x.astype(np.float64), y.astype(np.float64), batch_total=8)
        assert abs(loss - float(gnet[p + 'loss'])) <= 1e-9
        close(pred, gnet[p + 'pred'])
        np.testing.assert_array_equal(pred.argmax(-1), gnet[p + 'argmax'])
        np.testing.assert_allclose([np.linalg.norm(grads[k]) for k in keys], gnet[p + 'grad_norms'], rtol=1e-7)
        np.testing.assert_allclose([np.linalg.norm(state.params[k]) for k in keys], gnet[p + 'param_norms'], rtol=1e-9)
    close(state.stats['block_4/res_2/conv_2/bn/mu'], gnet['final/block_4_mu'])
    close(state.ema['block_None/logits/weights'], gnet['final/ema_logits_w'])
