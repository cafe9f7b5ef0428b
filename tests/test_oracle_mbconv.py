"""EfficientNet row of the oracle on the CPU: against torch (independent implementation of the same
published op definitions, float64), against the committed fixture tests/golden/mbconv.npz, and the architecture's
published parameter count.  PARITY UNPINNED as everywhere."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import net as ON
from oracle import ops as O

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import make_golden as MG  # noqa: E402

RNG = np.random.default_rng(31)


def t(a, grad=False):
    x = torch.from_numpy(np.ascontiguousarray(a))
    x.requires_grad_(grad)
    return x


def torch_dw(x, w, stride, padding='SAME'):
    """NHWC / [kh,kw,C,1] -> torch grouped conv with explicit TF-SAME pads."""
    n, h, wd, c = x.shape
    k = w.shape[0]
    pt, pb, pl, pr = O.resolve_pads(h, wd, k, k, stride, stride, padding)
    xt = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    return F.conv2d(xt, w.permute(2, 3, 0, 1), stride=stride, groups=c).permute(0, 2, 3, 1)


@pytest.mark.parametrize('case', [(2, 9, 9, 6, 3, 1), (2, 10, 10, 5, 5, 2), (2, 9, 7, 4, 5, 2), (1, 8, 8, 3, 3, 2), (2, 5, 5, 8, 5, 1)])
def test_depthwise_vs_torch(case):
    n, h, w_, c, k, s = case
    x, w = RNG.standard_normal((n, h, w_, c)), RNG.standard_normal((k, k, c, 1))
    y = O.depthwise_conv2d_fwd(x, w, s, 'SAME')
    xt, wt = t(x, True), t(w, True)
    yt = torch_dw(xt, wt, s)
    np.testing.assert_allclose(y, yt.detach().numpy(), rtol=1e-10, atol=1e-10)
    dy = RNG.standard_normal(y.shape)
    yt.backward(t(dy))
    np.testing.assert_allclose(O.depthwise_conv2d_dgrad(dy, w, x.shape, s, 'SAME'), xt.grad.numpy(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(O.depthwise_conv2d_wgrad(x, dy, w.shape, s, 'SAME'), wt.grad.numpy(), rtol=1e-10, atol=1e-10)


def test_swish_sigmoid_se_vs_torch():
    x, dy = 3 * RNG.standard_normal((5, 7)), RNG.standard_normal((5, 7))
    xt = t(x, True)
    (xt * torch.sigmoid(xt)).backward(t(dy))
    np.testing.assert_allclose(O.swish_fwd(x), (xt * torch.sigmoid(xt)).detach().numpy(), rtol=1e-12)
    np.testing.assert_allclose(O.swish_bwd(dy, x), xt.grad.numpy(), rtol=1e-10, atol=1e-12)
    xt = t(x, True)
    torch.sigmoid(xt).backward(t(dy))
    np.testing.assert_allclose(O.sigmoid_bwd(dy, O.sigmoid_fwd(x)), xt.grad.numpy(), rtol=1e-10, atol=1e-12)
    a, m, g = RNG.standard_normal((3, 4, 4, 6)), RNG.random((3, 1, 1, 6)), RNG.standard_normal((3, 4, 4, 6))
    at, mt = t(a, True), t(m, True)
    (at * mt).backward(t(g))
    dx, dm = O.channel_scale_bwd(g, a, m)
    np.testing.assert_allclose(dx, at.grad.numpy(), rtol=1e-12)
    np.testing.assert_allclose(dm, mt.grad.numpy().reshape(3, 6), rtol=1e-10)


def _torch_effnet_forward(spec, P, x, survival=None, dropout=None):
    def bn(h, s):
        mean = h.mean(dim=(0, 1, 2))
        var = ((h - mean) ** 2).mean(dim=(0, 1, 2))
        return (h - mean) / torch.sqrt(var + 1e-3) * P[s + '/gamma'] + P[s + '/beta']

    def conv(h, s, stride=1, bias=False):
        w = P[s + '/weights']
        n, hh, ww, c = h.shape
        k = w.shape[0]
        pt, pb, pl, pr = O.resolve_pads(hh, ww, k, k, stride, stride, 'SAME')
        y = F.conv2d(F.pad(h.permute(0, 3, 1, 2), (pl, pr, pt, pb)), w.permute(3, 2, 0, 1), stride=stride).permute(0, 2, 3, 1)
        return y + P[s + '/biases'] if bias else y

    sw = lambda v: v * torch.sigmoid(v)
    h = sw(bn(conv(x, 'block_0/conv_0', spec.strides[0]), 'block_0/conv_0/norm'))
    for name, k, s, cin, cout, mult, dr in spec.units():
        skip = h if (s == 1 and cin == cout) else None
        y = h
        if mult > 1:
            y = sw(bn(conv(y, name + '/conv_0'), name + '/conv_0/norm'))
        y = sw(bn(torch_dw(y, P[name + '/conv_1/weights'], s), name + '/conv_1/norm'))
        m = y.mean(dim=(1, 2), keepdim=True)
        m = sw(conv(m, name + '/se_mask/conv_0', 1, True))
        m = torch.sigmoid(conv(m, name + '/se_mask/conv_1', 1, True))
        y = bn(conv(y * m, name + '/conv_2'), name + '/conv_2/norm')
        if skip is not None:
            if survival is not None and dr > 0:
                y = y * t(survival[name]).reshape(-1, 1, 1, 1)
            y = y + skip
        h = y
    last = 'block_{}'.format(len(spec.channels) - 1)
    h = sw(bn(conv(h, last + '/conv_0'), last + '/conv_0/norm')).mean(dim=(1, 2))
    if dropout is not None:
        h = h * t(dropout)
    return h @ P['block_None/logits/weights'] + P['block_None/logits/biases']


@pytest.mark.parametrize('drop', [False, True])
def test_efficientnet_step_vs_torch_autograd(drop):
    spec = ON.EfficientNetSpec.b0(10, width_div=4, depth_div=2)
    params, stats = ON.init_variables(spec.variables(), seed=5, dtype=np.float64)
    for k in params:
        if k.endswith('gamma'):
            params[k] = 0.5 + RNG.random(params[k].shape)
        if k.endswith('beta') or k.endswith('biases'):
            params[k] = 0.1 * RNG.standard_normal(params[k].shape)
    survival = dropout = None
    if drop:
        spec.final_drop_rate = 0.4
        survival = {name: (RNG.random(4) >= dr) / (1.0 - dr) for name, k, s, cin, cout, mult, dr in spec.units() if s == 1 and cin == cout}
        dropout = (RNG.random((4, spec.channels[-1])) >= 0.3) / 0.7
        spec.survival, spec.dropout_mask = survival, dropout
    state = ON.TrainState(params, stats)
    x = RNG.random((4, 32, 32, 3))
    y = np.array([1, 7, 3, 9], dtype=np.float64)
    tape, out, pred, loss, onehot = ON.forward_loss(spec, state, x, y)
    grads = tape.backward()
    P = {k: t(v, True) for k, v in params.items()}
    logits = _torch_effnet_forward(spec, P, t(O.input_prep(x)), survival, dropout)
    ce = -(t(onehot) * F.log_softmax(logits, -1)).sum(-1).mean()
    ce.backward()
    np.testing.assert_allclose(out.a, logits.detach().numpy(), rtol=1e-8, atol=1e-9)
    assert set(grads) == set(P)
    for k in P:
        np.testing.assert_allclose(grads[k], P[k].grad.numpy(), rtol=1e-6, atol=1e-9, err_msg=k)


def test_efficientnet_b0_known_answer():
    """5,288,548 trainable parameters: the published EfficientNet-B0 count (arXiv 1905.11946, '5.3M'), which the
    reference's stage table (models/efficientnet.py:12-16) reproduces."""
    spec = ON.EfficientNetSpec.b0(1000)
    assert sum(int(np.prod(s)) for _, s, k in spec.variables() if k not in ('mu', 'sigma')) == 5288548


@pytest.fixture(scope='module')
def gm():
    return np.load(os.path.join(HERE, 'golden', 'mbconv.npz'))


def close(a, b, tol=2e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize('name', sorted(MG.DW_CASES))
def test_depthwise_golden(gm, name):
    n, h, w, c, k, s, pad = MG.DW_CASES[name]
    x, wt, dy = [gm[name + '/' + q].astype(np.float64) for q in ('x', 'w', 'dy')]
    close(O.depthwise_conv2d_fwd(x, wt, s, pad), gm[name + '/y'])
    close(O.depthwise_conv2d_dgrad(dy, wt, x.shape, s, pad), gm[name + '/dx'])
    close(O.depthwise_conv2d_wgrad(x, dy, wt.shape, s, pad), gm[name + '/dw'])


def test_act_se_golden(gm):
    x, dy = gm['act/x'].astype(np.float64), gm['act/dy'].astype(np.float64)
    close(O.swish_fwd(x), gm['act/swish']); close(O.swish_bwd(dy, x), gm['act/swish_dx']); close(O.sigmoid_fwd(x), gm['act/sigmoid'])
    dx, dm = O.channel_scale_bwd(gm['se/dy'].astype(np.float64), gm['se/x'].astype(np.float64), gm['se/m'].astype(np.float64))
    close(O.channel_scale_fwd(gm['se/x'].astype(np.float64), gm['se/m'].astype(np.float64)), gm['se/y']); close(dx, gm['se/dx']); close(dm, gm['se/dm'])


def test_efficientnet_two_step_golden(gm):
    spec = ON.EfficientNetSpec.b0(10, width_div=2, depth_div=2)
    spec.final_drop_rate = 0.3
    params, stats = MG.effnet_params(spec)
    chk = np.array([sum(float(v.astype(np.float64).sum()) for v in params.values()), sum(float((v.astype(np.float64) ** 2).sum()) for v in params.values())])
    np.testing.assert_allclose(chk, gm['net/param_checksum'], rtol=1e-12)
    state = ON.TrainState({k: v.astype(np.float64) for k, v in params.items()}, {k: v.astype(np.float64) for k, v in stats.items()})
    keys = [str(k) for k in gm['net/keys']]
    units = [str(u) for u in gm['net/units']]
    for step in range(2):
        p = 'net/step{}/'.format(step)
        x = gm[p + 'x_u8'].astype(np.float32) / np.float32(255)
        spec.survival = {u: gm[p + 'survival'][i].astype(np.float64) for i, u in enumerate(units)}
        spec.dropout_mask = gm[p + 'dropout'].astype(np.float64)
        loss, pred, grads = ON.train_step(spec, state, x.astype(np.float64), gm[p + 'y'].astype(np.float64), batch_total=8)
        assert abs(loss - float(gm[p + 'loss'])) <= 1e-9
        close(pred, gm[p + 'pred'])
        np.testing.assert_array_equal(pred.argmax(-1), gm[p + 'argmax'])
        np.testing.assert_allclose([np.linalg.norm(grads[k]) for k in keys], gm[p + 'grad_norms'], rtol=1e-7, atol=1e-12)
        np.testing.assert_allclose([np.linalg.norm(state.params[k]) for k in keys], gm[p + 'param_norms'], rtol=1e-9)


# ---- segmentation row: DeepLabv3+ oracle vs torch autograd ------------------------------------------------------
def _torch_deeplab_forward(spec, P, x):
    def bn(h, s):
        mean = h.mean(dim=(0, 1, 2))
        var = ((h - mean) ** 2).mean(dim=(0, 1, 2))
        return (h - mean) / torch.sqrt(var + 1e-3) * P[s + '/gamma'] + P[s + '/beta']

    def conv(h, s, stride=1, dil=1, bias=False):
        w = P[s + '/weights']
        n, hh, ww, c = h.shape
        k = w.shape[0]
        pt, pb, pl, pr = O.resolve_pads(hh, ww, k, k, stride, stride, 'SAME', dil, dil)
        y = F.conv2d(F.pad(h.permute(0, 3, 1, 2), (pl, pr, pt, pb)), w.permute(3, 2, 0, 1), stride=stride, dilation=dil).permute(0, 2, 3, 1)
        return y + P[s + '/biases'] if bias else y

    def maxpool(h, k, s, padding):
        n, hh, ww, c = h.shape
        pt, pb, pl, pr = O.resolve_pads(hh, ww, k, k, s, s, padding)
        return F.max_pool2d(F.pad(h.permute(0, 3, 1, 2), (pl, pr, pt, pb), value=float('-inf')), k, s).permute(0, 2, 3, 1)

    def resize(h, hw):
        return F.interpolate(h.permute(0, 3, 1, 2), size=tuple(hw), mode='bilinear', align_corners=True).permute(0, 2, 3, 1)

    bb = spec.backbone
    ch = bb.channels
    h = maxpool(F.relu(bn(conv(x, 'block_0/conv_0', bb.strides[0]), 'block_0/conv_0/bn')), 3, 2, 'SAME')
    feats = {}
    cin = ch[0]
    for i in range(1, len(ch)):
        for j in range(bb.res_units[i]):
            s = bb.strides[i] if j == 0 else 1
            nm = 'block_{}/res_{}'.format(i, j)
            cout = ch[i]
            if cin == cout:
                skip = maxpool(h, s, s, 'VALID') if s > 1 else h
            else:
                skip = bn(conv(h, nm + '/conv_skip', s), nm + '/conv_skip/bn')
            y = F.relu(bn(conv(h, nm + '/conv_0'), nm + '/conv_0/bn'))
            y = F.relu(bn(conv(y, nm + '/conv_1', s, bb.unit_dilation(i, j)), nm + '/conv_1/bn'))
            y = bn(conv(y, nm + '/conv_2'), nm + '/conv_2/bn')
            h = F.relu(y + skip)
            cin = cout
        feats[i] = h
    f4, f1 = feats[4], feats[1]
    ys = [bn(conv(f4, 'block_5/aspp/conv_0'), 'block_5/aspp/conv_0/norm')]
    for i, dil in enumerate(spec.aspp_dilations):
        sc = 'block_5/aspp/conv_{}'.format(i + 1)
        ys.append(bn(conv(f4, sc, 1, dil), sc + '/norm'))
    h = bn(conv(torch.cat(ys, -1), 'block_5/aspp/conv_out'), 'block_5/aspp/conv_out/norm')
    feat = bn(conv(f1, 'block_6/features'), 'block_6/features/norm')
    h = resize(h, feat.shape[1:3])
    h = bn(conv(torch.cat([h, feat], -1), 'block_6/decoder/conv_0'), 'block_6/decoder/conv_0/norm')
    return resize(conv(h, 'block_None/logits', bias=True), x.shape[1:3])


def test_deeplab_step_vs_torch_autograd():
    spec = ON.DeepLabSpec(5, width_div=8, depth_div=3, aspp_dilations=(1, 2, 3))
    params, stats = ON.init_variables(spec.variables(), seed=7, dtype=np.float64)
    for k in params:
        if k.endswith('gamma'):
            params[k] = 0.5 + RNG.random(params[k].shape)
        if k.endswith('beta') or k.endswith('biases'):
            params[k] = 0.1 * RNG.standard_normal(params[k].shape)
    state = ON.TrainState(params, stats)
    x = RNG.random((2, 65, 65, 3))
    y = RNG.integers(0, 6, (2, 65, 65)).astype(np.float64)
    y[0, 0, :4] = np.nan
    tape, out, pred, loss, onehot = ON.forward_loss(spec, state, x, y)
    grads = tape.backward()
    P = {k: t(v, True) for k, v in params.items()}
    logits = _torch_deeplab_forward(spec, P, t(O.input_prep(x)))
    valid = t(onehot.sum(-1))
    ce = (-(t(onehot) * F.log_softmax(logits, -1)).sum(-1) * valid).mean()     # mean over ALL pixels, ignored ones weigh 0
    ce.backward()
    np.testing.assert_allclose(out.a, logits.detach().numpy(), rtol=1e-8, atol=1e-9)
    l2 = sum((p ** 2).sum() / 2 for k, p in P.items() if k.endswith('/weights')) * 1e-4
    np.testing.assert_allclose(loss, (ce + l2).item(), rtol=1e-10)
    assert set(grads) == set(P)
    for k in P:
        np.testing.assert_allclose(grads[k], P[k].grad.numpy(), rtol=1e-6, atol=1e-9, err_msg=k)


def test_deeplab_known_answers():
    """DeepLabv3+ on ResNet-50 (BASELINE configs[4]): 25.6M backbone - fc head + ASPP / decoder head; per-class logits."""
    spec = ON.DeepLabSpec(19)
    names = [n for n, _, _ in spec.variables()]
    assert 'block_5/aspp/conv_3/weights' in names and 'block_6/decoder/conv_0/norm/gamma' in names and 'block_None/logits/biases' in names
    n = sum(int(np.prod(s)) for _, s, k in spec.variables() if k not in ('mu', 'sigma'))
    backbone = 25557032 - (2048 * 1000 + 1000)
    head = 2048 * 256 + 3 * 9 * 2048 * 256 + 1024 * 256 + 256 * 48 + 9 * 304 * 256 + 256 * 19 + 19 + 2 * (5 * 256 + 48 + 256)
    assert n == backbone + head == 39170995
