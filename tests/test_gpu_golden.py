"""GPU parity against the committed golden fixtures (tests/golden/*.npz): fixed inputs and expected outputs that
travel to the GPU box; fp32 path, tolerance 1e-3 relative (asserted at 2e-5 rel-L2), integer results bit-exact."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import make_golden as MG  # noqa: E402


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.fixture(scope='module')
def gops():
    return np.load(os.path.join(HERE, 'golden', 'ops.npz'))


@pytest.mark.parametrize('name', sorted(MG.CONV_CASES))
def test_conv_golden(gops, name):
    import abi_util as u
    n, h, w, cin, cout, k, s, pad, dil = MG.CONV_CASES[name]
    x, wt, dy = gops[name + '/x'], gops[name + '/w'], gops[name + '/dy']
    bias = gops[name + '/b'] if 'biased' in name else None
    if cin % 4:                                       # stem: channels stored with a 16-byte stride
        xp = np.zeros(x.shape[:3] + (4,), np.float32)
        xp[..., :cin] = x
        y = u.conv_fwd(xp, wt, s, pad, dil, 'float32', bias=bias, x_cs=4)
        dw = u.conv_wgrad(xp, dy, wt.shape, s, pad, dil, 'float32', x_cs=4)
    else:
        y = u.conv_fwd(x, wt, s, pad, dil, 'float32', bias=bias)
        dw = u.conv_wgrad(x, dy, wt.shape, s, pad, dil, 'float32')
        assert rel_l2(u.conv_dgrad(dy, wt, x.shape, s, pad, dil, 'float32'), gops[name + '/dx']) <= 2e-5
    assert rel_l2(y, gops[name + '/y']) <= 2e-5
    assert rel_l2(dw, gops[name + '/dw']) <= 2e-5


def test_bn_maxpool_golden(gops):
    import abi_util as u
    from myconvnet_amd import _ffi
    g = gops
    out = u.bn_fwd_train(g['bn/x'], g['bn/gamma'], g['bn/beta'], 1e-3, 'float32', running=(np.zeros(8, np.float32), np.ones(8, np.float32)), momentum=0.99)
    for k, f in (('y', 'bn/y'), ('batch_mean', 'bn/batch_mean'), ('batch_var', 'bn/batch_var'), ('save_invstd', 'bn/invstd'),
                 ('running_mean', 'bn/running_mean'), ('running_var', 'bn/running_var')):
        assert rel_l2(out[k], g[f]) <= 1e-5, k
    dx, dg, db, _ = u.bn_bwd(g['bn/dy'], g['bn/x'], None, g['bn/gamma'], out['save_mean'], out['save_invstd'], 'float32')
    assert rel_l2(dx, g['bn/dx']) <= 2e-5 and rel_l2(dg, g['bn/dgamma']) <= 1e-4 and rel_l2(db, g['bn/dbeta']) <= 1e-4
    for nm, (k, s) in {'mp3x3_s2': (3, 2), 'mp2x2_s2': (2, 2)}.items():
        x = g[nm + '/x']
        h = x.shape[1]
        pt, _, pl, _ = u.O.resolve_pads(h, h, k, k, s, s, 'SAME')
        oh = g[nm + '/y'].shape[1]
        xd = u.dev(x)
        y = torch.zeros(g[nm + '/y'].shape, dtype=torch.float32, device=u.DEV)
        arg = torch.zeros(g[nm + '/y'].shape, dtype=torch.int8, device=u.DEV)
        _ffi.check(_ffi.lib.mcn_maxpool_fwd(xd.data_ptr(), y.data_ptr(), arg.data_ptr(), x.shape[0], h, h, x.shape[3], k, k, s, s, pt, pl, oh, oh, _ffi.F32, u.stream()))
        np.testing.assert_array_equal(u.host(y), g[nm + '/y'])
        np.testing.assert_array_equal(arg.cpu().numpy(), g[nm + '/arg'])          # integer arg-max bit-exact


def test_resnet_two_step_golden():
    import myconvnet_amd as M
    from oracle import net as ON
    gnet = np.load(os.path.join(HERE, 'golden', 'resnet50_w8.npz'))
    spec = ON.ResNetSpec.resnet50(10, 8)
    params, stats = MG.net_params(spec)
    model = M.ResNet50([64, 64, 3], 10, batch_size=8, width_div=8, num_gpus=1)
    model.set_variables(dict(params, **stats))
    opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=1, learning_warmup_epochs=0.0)
    keys = [str(k) for k in gnet['keys']]
    for step in range(2):
        p = 'step{}/'.format(step)
        x = gnet[p + 'x_u8'].astype(np.float32) / np.float32(255)
        model.feed(x, gnet[p + 'y'])
        loss, _, pred = opt._step(None)
        assert abs(loss - float(gnet[p + 'loss'])) <= 1e-4 * abs(float(gnet[p + 'loss']))
        assert rel_l2(model.fetch(model.logits), gnet[p + 'logits']) <= 1e-4
        assert rel_l2(pred, gnet[p + 'pred']) <= 1e-4
        np.testing.assert_array_equal(pred.argmax(-1), gnet[p + 'argmax'])
        grads, data = model.get_variables('grad'), model.get_variables('data')
        np.testing.assert_allclose([np.linalg.norm(grads[k]) for k in keys], gnet[p + 'grad_norms'], rtol=1e-3)
        np.testing.assert_allclose([np.linalg.norm(data[k]) for k in keys], gnet[p + 'param_norms'], rtol=1e-5)
        if step == 0:
            for k in ('block_0/conv_0/weights', 'block_2/res_0/conv_skip/weights', 'block_None/logits/weights'):
                assert rel_l2(grads[k], gnet[p + 'grad/' + k]) <= 1e-3, k
    assert rel_l2(data['block_4/res_2/conv_2/bn/mu'], gnet['final/block_4_mu']) <= 1e-4
    assert rel_l2(model.get_variables('ema')['block_None/logits/weights'], gnet['final/ema_logits_w']) <= 1e-5


# ---- EfficientNet row (tests/golden/mbconv.npz) ----------------------------------------------------------------------
@pytest.fixture(scope='module')
def gmb():
    return np.load(os.path.join(HERE, 'golden', 'mbconv.npz'))


@pytest.mark.parametrize('name', sorted(MG.DW_CASES))
def test_depthwise_golden(gmb, name):
    import abi_util as u
    n, h, w, c, k, s, pad = MG.DW_CASES[name]
    x, wt, dy = gmb[name + '/x'], gmb[name + '/w'], gmb[name + '/dy']
    assert rel_l2(u.dwconv_fwd(x, wt, s, pad), gmb[name + '/y']) <= 2e-5
    assert rel_l2(u.dwconv_dgrad(dy, wt, x.shape, s, pad), gmb[name + '/dx']) <= 2e-5
    assert rel_l2(u.dwconv_wgrad(x, dy, k, s, pad), gmb[name + '/dw']) <= 2e-5


def test_efficientnet_two_step_golden(gmb):
    """Fixed inputs, labels and stochastic-depth / dropout masks from the fixture; loss, predictions, arg-max and the
    norms of all 109 gradients / parameters after each step."""
    import myconvnet_amd as M
    from oracle import net as ON
    spec = ON.EfficientNetSpec.b0(10, width_div=2, depth_div=2)
    params, stats = MG.effnet_params(spec)
    model = M.EfficientNetB0([64, 64, 3], 10, batch_size=8, width_div=2, depth_div=2, final_drop_rate=0.3, dropout_rate=0.25, num_gpus=1)
    model.set_variables(dict(params, **stats))
    opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, steps_per_epoch=1, learning_warmup_epochs=0.0)
    keys = [str(k) for k in gmb['net/keys']]
    units = [str(k) for k in gmb['net/units']]
    for step in range(2):
        p = 'net/step{}/'.format(step)
        masks = {u + '/drop/survived': gmb[p + 'survival'][i] for i, u in enumerate(units)}
        masks['block_None/logits/dropout'] = gmb[p + 'dropout']
        assert set(masks) == set(n.scope for n in model._random_nodes)
        model.fixed_random_masks = masks
        model.feed(gmb[p + 'x_u8'].astype(np.float32) / np.float32(255), gmb[p + 'y'])
        loss, _, pred = opt._step(None)
        assert abs(loss - float(gmb[p + 'loss'])) <= 1e-4 * abs(float(gmb[p + 'loss']))
        assert rel_l2(pred, gmb[p + 'pred']) <= 1e-4
        np.testing.assert_array_equal(pred.argmax(-1), gmb[p + 'argmax'])
        grads, data = model.get_variables('grad'), model.get_variables('data')
        ref = gmb[p + 'grad_norms']
        got = np.array([np.linalg.norm(grads[k]) for k in keys])
        live = ref > 1e-6 * np.median(ref)                     # exact-zero gradients (see test_gpu_efficientnet.worst_grad)
        np.testing.assert_allclose(got[live], ref[live], rtol=1e-3)
        assert (got[~live] <= 1e-4 * np.median(ref)).all()
        np.testing.assert_allclose([np.linalg.norm(data[k]) for k in keys], gmb[p + 'param_norms'], rtol=1e-5)
