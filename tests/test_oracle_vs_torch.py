"""Cross-check of the NumPy oracle against torch-CPU (an independent implementation of the same
published op definitions; NOT the oracle of record).  float64 so that agreement is ~1e-12."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from oracle import ops, net

RNG = np.random.default_rng(7)


def t(a, grad=False):
    x = torch.from_numpy(np.ascontiguousarray(a))
    x.requires_grad_(grad)
    return x


def torch_conv(x, w, stride, padding, dilation=1):
    """NHWC/HWIO -> torch NCHW/OIHW with explicit asymmetric TF-SAME pads."""
    n, h, wd, c = x.shape
    kh, kw = w.shape[0], w.shape[1]
    pt, pb, pl, pr = ops.resolve_pads(h, wd, kh, kw, stride, stride, padding, dilation, dilation)
    xt = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xt, w.permute(3, 2, 0, 1), stride=stride, dilation=dilation)
    return y.permute(0, 2, 3, 1)


CONV_CASES = [
    # n, h, w, cin, cout, k, stride, padding, dilation
    (2, 8, 8, 8, 16, 1, 1, 'SAME', 1),
    (2, 8, 8, 8, 16, 1, 2, 'SAME', 1),
    (2, 7, 7, 8, 8, 3, 1, 'SAME', 1),
    (2, 16, 16, 8, 8, 3, 2, 'SAME', 1),      # even H, pads (0,1)
    (2, 15, 15, 8, 8, 3, 2, 'SAME', 1),      # odd H, pads (1,1)
    (2, 16, 16, 3, 16, 7, 2, 'SAME', 1),     # stem, pads (2,3)
    (2, 9, 9, 4, 8, 7, 1, 'VALID', 1),       # VGG head style
    (1, 14, 14, 8, 8, 3, 1, 'SAME', 2),      # dilated
    (2, 8, 6, 5, 7, 3, 1, 'SAME', 1),        # ragged
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case):
    n, h, w_, cin, cout, k, s, pad, dil = case
    x = RNG.standard_normal((n, h, w_, cin))
    w = RNG.standard_normal((k, k, cin, cout))
    y = ops.conv2d_fwd(x, w, s, pad, dil)
    xt, wt = t(x, True), t(w, True)
    yt = torch_conv(xt, wt, s, pad, dil)
    assert y.shape == tuple(yt.shape)
    np.testing.assert_allclose(y, yt.detach().numpy(), rtol=1e-10, atol=1e-10)
    dy = RNG.standard_normal(y.shape)
    yt.backward(t(dy))
    np.testing.assert_allclose(ops.conv2d_dgrad(dy, w, x.shape, s, pad, dil), xt.grad.numpy(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(ops.conv2d_wgrad(x, dy, w.shape, s, pad, dil), wt.grad.numpy(), rtol=1e-10, atol=1e-10)


def test_same_pads_table():
    assert ops.same_pads(224, 7, 2) == (2, 3)
    assert ops.same_pads(56, 3, 1) == (1, 1)
    assert ops.same_pads(56, 3, 2) == (0, 1)
    assert ops.same_pads(56, 1, 2) == (0, 0)
    assert ops.same_pads(112, 3, 2) == (0, 1)
    assert ops.same_pads(513, 7, 2) == (3, 3)
    assert ops.same_pads(15, 3, 2) == (1, 1)


def test_bn_train_and_bwd():
    x = RNG.standard_normal((4, 5, 6, 8)) * 3 + 1.5
    g = RNG.standard_normal(8)
    b = RNG.standard_normal(8)
    y, bm, bv, sm, si = ops.bn_fwd_train(x, g, b, 1e-3)
    xt, gt, bt = t(x, True), t(g, True), t(b, True)
    rm, rv = torch.zeros(8, dtype=torch.float64), torch.ones(8, dtype=torch.float64)
    yt = F.batch_norm(xt.permute(0, 3, 1, 2), rm, rv, gt, bt, training=True, momentum=1.0, eps=1e-3).permute(0, 2, 3, 1)
    np.testing.assert_allclose(y, yt.detach().numpy(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(bm, rm.numpy(), rtol=1e-10)
    np.testing.assert_allclose(bv, rv.numpy(), rtol=1e-10)           # torch running_var is unbiased too
    dy = RNG.standard_normal(y.shape)
    yt.backward(t(dy))
    dx, dg, db = ops.bn_bwd(dy, x, g, sm, si)
    np.testing.assert_allclose(dx, xt.grad.numpy(), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(dg, gt.grad.numpy(), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(db, bt.grad.numpy(), rtol=1e-9, atol=1e-10)
    yi = ops.bn_fwd_infer(x, g, b, bm, bv, 1e-3)
    yit = F.batch_norm(t(x).permute(0, 3, 1, 2), t(bm), t(bv), t(g), t(b), training=False, eps=1e-3).permute(0, 2, 3, 1)
    np.testing.assert_allclose(yi, yit.numpy(), rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize('h,k,s,pad', [(16, 3, 2, 'SAME'), (15, 3, 2, 'SAME'), (8, 2, 2, 'SAME'), (7, 2, 2, 'SAME'),
                                        (8, 2, 2, 'VALID'), (1, 2, 2, 'SAME')])
def test_maxpool(h, k, s, pad):
    x = RNG.standard_normal((2, h, h, 4))
    x[0, :, :, 0] = 0.0                      # ties: first max in window scan order wins
    y, arg = ops.maxpool_fwd(x, k, s, pad)
    pt, pb, pl, pr = ops.resolve_pads(h, h, k, k, s, s, pad)
    xt = t(x, True)
    xp = F.pad(xt.permute(0, 3, 1, 2), (pl, pr, pt, pb), value=float('-inf'))
    yt = F.max_pool2d(xp, k, s).permute(0, 2, 3, 1)
    np.testing.assert_array_equal(y, yt.detach().numpy())
    dy = RNG.standard_normal(y.shape)
    yt.backward(t(dy))
    np.testing.assert_allclose(ops.maxpool_bwd(dy, arg, x.shape, k, s, pad), xt.grad.numpy(), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('h,k,s,pad', [(8, 3, 2, 'SAME'), (7, 3, 1, 'SAME'), (8, 2, 2, 'VALID')])
def test_avgpool(h, k, s, pad):
    x = RNG.standard_normal((2, h, h, 4))
    y = ops.avgpool_fwd(x, k, s, pad)
    pt, pb, pl, pr = ops.resolve_pads(h, h, k, k, s, s, pad)
    xt = t(x, True)
    xp = F.pad(xt.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    op = F.pad(torch.ones(1, 1, h, h, dtype=torch.float64), (pl, pr, pt, pb))
    yt = (F.avg_pool2d(xp, k, s) / F.avg_pool2d(op, k, s)).permute(0, 2, 3, 1)
    np.testing.assert_allclose(y, yt.detach().numpy(), rtol=1e-12, atol=1e-12)
    dy = RNG.standard_normal(y.shape)
    yt.backward(t(dy))
    np.testing.assert_allclose(ops.avgpool_bwd(dy, x.shape, k, s, pad), xt.grad.numpy(), rtol=1e-12, atol=1e-12)


def test_softmax_xent():
    logits = RNG.standard_normal((6, 10)) * 3
    labels = np.array([1, 3, np.nan, 9, 0, 12], dtype=np.float64)   # NaN and out-of-range -> invalid rows
    oh = ops.one_hot_labels(labels, 10, np.float64)
    assert oh[2].sum() == 0 and oh[5].sum() == 0
    pred, loss, ce, dl = ops.softmax_xent_fwd_bwd(logits, oh)
    lt = t(logits, True)
    logsm = F.log_softmax(lt, dim=-1)
    ce_t = -(t(oh) * logsm).sum(-1)
    valid = torch.tensor([1, 1, 0, 1, 1, 0], dtype=torch.float64)
    loss_t = (ce_t * valid).mean()
    loss_t.backward()
    np.testing.assert_allclose(loss, loss_t.item(), rtol=1e-12)
    np.testing.assert_allclose(dl, lt.grad.numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(pred, F.softmax(t(logits), -1).numpy(), rtol=1e-12)
    # label smoothing
    pred, loss, ce, dl = ops.softmax_xent_fwd_bwd(logits, oh, label_smoothing=0.1)
    lt = t(logits, True)
    lab = t(oh) * 0.9 + 0.1 / 10
    loss_t = ((-(lab * F.log_softmax(lt, -1)).sum(-1)) * valid).mean()
    loss_t.backward()
    np.testing.assert_allclose(loss, loss_t.item(), rtol=1e-12)
    np.testing.assert_allclose(dl, lt.grad.numpy(), rtol=1e-10, atol=1e-12)


def test_nesterov_matches_torch_sgd():
    """TF ApplyMomentum(use_nesterov) == torch SGD(nesterov=True, dampening=0) with lr folded the TF way
    when lr is constant: TF accum a = m*a + g ; w -= lr*(g + m*a)."""
    w = RNG.standard_normal(50)
    wt = t(w.copy(), True)
    opt = torch.optim.SGD([wt], lr=0.05, momentum=0.9, nesterov=True, weight_decay=1e-4)
    acc = np.zeros_like(w)
    for i in range(4):
        g = RNG.standard_normal(50)
        wt.grad = t(g.copy())
        opt.step()
        w, acc, _ = ops.sgd_nesterov_step(w, g, acc, 0.05, 0.9, l2=1e-4)
        np.testing.assert_allclose(w, wt.detach().numpy(), rtol=1e-12, atol=1e-12)


def test_ema_decay_and_lr_schedule():
    assert ops.ema_decay(0.99, 0) == pytest.approx(0.1)
    assert ops.ema_decay(0.99, 1000) == pytest.approx(0.99)
    assert ops.lr_multiplier(0, 100, 10) == pytest.approx(0.01)
    assert ops.lr_multiplier(99, 100, 10) == pytest.approx(1.0)
    assert ops.lr_multiplier(100, 100, 10, decay_method='cosine', decay_params=(0,)) == pytest.approx(1.0)
    assert ops.lr_multiplier(550, 100, 10, decay_method='cosine', decay_params=(0,)) == pytest.approx(0.5)
    assert ops.lr_multiplier(550, 100, 10, decay_method='poly', decay_params=(2,)) == pytest.approx(0.25)


def _torch_resnet_forward(spec, P, S, x, train=True):
    def conv(h, name, s):
        return torch_conv(h, P[name + '/weights'], s, 'SAME')

    def bn(h, name):
        return F.batch_norm(h.permute(0, 3, 1, 2), None if train else S[name + '/mu'], None if train else S[name + '/sigma'],
                            P[name + '/gamma'], P[name + '/beta'], training=train, eps=1e-3).permute(0, 2, 3, 1)

    def mp(h, k, s, pad):
        pt, pb, pl, pr = ops.resolve_pads(h.shape[1], h.shape[2], k, k, s, s, pad)
        return F.max_pool2d(F.pad(h.permute(0, 3, 1, 2), (pl, pr, pt, pb), value=float('-inf')), k, s).permute(0, 2, 3, 1)
    ch = spec.channels
    h = mp(F.relu(bn(conv(x, 'block_0/conv_0', 2), 'block_0/conv_0/bn')), 3, 2, 'SAME')
    cin = ch[0]
    for i in range(1, len(ch)):
        for j in range(spec.res_units[i]):
            s = spec.strides[i] if j == 0 else 1
            nm = 'block_{}/res_{}'.format(i, j)
            cout = ch[i]
            if cin == cout:
                skip = mp(h, s, s, 'VALID') if s > 1 else h
            else:
                skip = bn(conv(h, nm + '/conv_skip', s), nm + '/conv_skip/bn')
            if spec.bottleneck:
                y = F.relu(bn(conv(h, nm + '/conv_0', 1), nm + '/conv_0/bn'))
                y = F.relu(bn(conv(y, nm + '/conv_1', s), nm + '/conv_1/bn'))
                y = bn(conv(y, nm + '/conv_2', 1), nm + '/conv_2/bn')
            else:
                y = F.relu(bn(conv(h, nm + '/conv_0', s), nm + '/conv_0/bn'))
                y = bn(conv(y, nm + '/conv_1', 1), nm + '/conv_1/bn')
            h = F.relu(y + skip)
            cin = cout
    h = h.mean(dim=(1, 2))
    return h @ P['block_None/logits/weights'] + P['block_None/logits/biases']


@pytest.mark.parametrize('bottleneck', [True, False])
def test_resnet_step_vs_torch_autograd(bottleneck):
    """Whole-network gradient of the oracle (explicit backward) vs torch autograd, float64."""
    spec = net.ResNetSpec.resnet50(10, width_div=8) if bottleneck else net.ResNetSpec.resnet18(10, width_div=8)
    params, stats = net.init_variables(spec.variables(), seed=3, dtype=np.float64)
    for k in params:                         # make zero-init gammas non-trivial for the check
        if k.endswith('gamma'):
            params[k] = 0.5 + RNG.random(params[k].shape)
        if k.endswith('beta'):
            params[k] = 0.1 * RNG.standard_normal(params[k].shape)
    state = net.TrainState(params, stats)
    x = RNG.random((4, 32, 32, 3))
    y = np.array([1, 7, 3, 9], dtype=np.float64)
    tape, out, pred, loss, onehot = net.forward_loss(spec, state, x, y)
    grads = tape.backward()
    P = {k: t(v, True) for k, v in params.items()}
    logits = _torch_resnet_forward(spec, P, None, t(ops.input_prep(x)))
    ce = -(t(onehot) * F.log_softmax(logits, -1)).sum(-1).mean()
    l2 = sum((p ** 2).sum() / 2 for k, p in P.items() if k.endswith('/weights')) * 1e-4
    ce.backward()
    np.testing.assert_allclose(out.a, logits.detach().numpy(), rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(loss, (ce + l2).item(), rtol=1e-10)
    assert set(grads) == set(P)
    for k in P:
        np.testing.assert_allclose(grads[k], P[k].grad.numpy(), rtol=1e-6, atol=1e-9, err_msg=k)


def test_resnet50_known_answers():
    """The only 'known answers' the reference prints (convnet.py:210-212):
    25,557,032 trainable parameters and 4,087,136,256 conv MACs for ResNet-50 @224."""
    spec = net.ResNetSpec.resnet50(1000)
    n = sum(int(np.prod(s)) for _, s, k in spec.variables() if k not in ('mu', 'sigma'))
    assert n == 25557032
    assert len([1 for _, _, k in spec.variables() if k not in ('mu', 'sigma')]) == 161


def test_decoupled_decay_variants_formulae():
    """optimizers.py:163-170 restated in oracle.ops.decoupled_decay, against the formulae written out with torch."""
    rng = np.random.default_rng(3)
    w = rng.standard_normal(1000) * 0.2
    w[:3] = 0.0
    wt = torch.tensor(w)
    np.testing.assert_allclose(ops.decoupled_decay(w, 0.01), (wt - 0.01 * wt).numpy(), rtol=1e-14)
    np.testing.assert_allclose(ops.decoupled_decay(w, 0.01, l1=True), (wt - 0.01 * torch.sign(wt)).numpy(), rtol=1e-14)
    hub = wt - 0.01 * wt / torch.sqrt(1 + (wt / 0.1) ** 2)
    np.testing.assert_allclose(ops.decoupled_decay(w, 0.01, l1=True, huber_delta=0.1), hub.numpy(), rtol=1e-14)   # Huber wins over L1
    # small |w|: Huber ~ L2 decay; large |w|: ~ L1 decay with step wd*delta
    assert abs(ops.decoupled_decay(np.array([1e-4]), 0.01, huber_delta=0.1)[0] - (1e-4 - 0.01 * 1e-4)) < 1e-10
    assert abs(ops.decoupled_decay(np.array([50.0]), 0.01, huber_delta=0.1)[0] - (50.0 - 0.01 * 0.1)) < 1e-8


def test_bn_frozen_statistics_gradient_matches_torch_autograd():
    """oracle.ops.bn_bwd_frozen = autograd through F.batch_norm(training=False) (fused_batch_norm(is_training=False))."""
    rng = np.random.default_rng(8)
    x = rng.standard_normal((3, 5, 4, 6))
    dy = rng.standard_normal(x.shape)
    gamma, beta = 0.5 + rng.random(6), rng.standard_normal(6)
    mean, var = 0.3 * rng.standard_normal(6), 0.5 + rng.random(6)
    xt = torch.tensor(x.transpose(0, 3, 1, 2), requires_grad=True)
    gt, bt = torch.tensor(gamma, requires_grad=True), torch.tensor(beta, requires_grad=True)
    y = F.batch_norm(xt, torch.tensor(mean), torch.tensor(var), gt, bt, training=False, eps=1e-3)
    np.testing.assert_allclose(ops.bn_fwd_infer(x, gamma, beta, mean, var, 1e-3), y.detach().numpy().transpose(0, 2, 3, 1), rtol=1e-12, atol=1e-12)
    y.backward(torch.tensor(dy.transpose(0, 3, 1, 2)))
    dx, dg, db = ops.bn_bwd_frozen(dy, x, gamma, mean, var, 1e-3)
    np.testing.assert_allclose(dx, xt.grad.numpy().transpose(0, 2, 3, 1), rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(dg, gt.grad.numpy(), rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(db, bt.grad.numpy(), rtol=1e-11, atol=1e-12)


def test_blocks_to_train_freezes_variables_and_statistics_in_the_oracle():
    spec = net.ResNetSpec.resnet18(5, 8)
    params, stats = net.init_variables(spec.variables(), seed=1, dtype=np.float64)
    rng = np.random.default_rng(2)
    for k in params:
        if k.endswith('gamma'):
            params[k] = 0.5 + rng.random(params[k].shape)
    state = net.TrainState(params, stats)
    x, y = rng.random((4, 32, 32, 3)), np.array([0., 1., 2., 3.])
    hp = dict(blocks_to_train=[4, None], base_weight_decay=0.1)
    _, _, grads = net.train_step(spec, state, x, y, hp=hp, batch_total=4)
    assert set(grads) == {k for k in params if k.startswith('block_4/') or k.startswith('block_None/')}
    for k in params:
        moved = not np.array_equal(state.params[k], params[k])
        assert moved == (k in grads) or not np.any(grads.get(k, 1)), k
    for k in stats:                                       # statistics follow blocks_to_train when update_batch_norm is None
        assert np.array_equal(state.stats[k], stats[k]) == (not k.startswith('block_4/')), k
    # update_batch_norm=True overrides: every BN uses batch statistics and updates
    state2 = net.TrainState(params, stats)
    net.train_step(spec, state2, x, y, hp=dict(hp, update_batch_norm=True), batch_total=4)
    assert all(not np.array_equal(state2.stats[k], stats[k]) for k in stats)
    assert net.trainable_name('block_None/logits/weights', [None]) and not net.trainable_name('block_3/res_0/conv_0/weights', [None])


def test_torch_cpu_step_matches_oracle():
    """oracle/torch_cpu.py (the timed CPU stand-in of bench.py's cpu_baseline) performs the same training step as the NumPy
    oracle of record: loss, updated parameters, EMA shadows and BN running statistics after two steps, float64."""
    from oracle.torch_cpu import ResNetTorchCPU
    spec = net.ResNetSpec.resnet50(10, width_div=8)
    params, stats = net.init_variables(spec.variables(), seed=3, dtype=np.float64)
    rng = np.random.default_rng(21)
    for k in params:
        if k.endswith('gamma'):
            params[k] = 0.5 + rng.random(params[k].shape)
    state = net.TrainState({k: v.copy() for k, v in params.items()}, {k: v.copy() for k, v in stats.items()})
    tc = ResNetTorchCPU(spec, params, stats, channels_last=False)
    for step in range(2):
        x = rng.random((4, 32, 32, 3))
        y = rng.integers(0, 10, 4).astype(np.float64)
        rloss, rpred, _ = net.train_step(spec, state, x, y, batch_total=4)
        loss, logits = tc.train_step(x, y, batch_total=4)
        assert loss == pytest.approx(rloss, rel=1e-9)
        np.testing.assert_allclose(torch.softmax(logits, -1).numpy(), rpred, rtol=1e-7, atol=1e-10)
    got = tc.params_hwio()
    for k, v in state.params.items():
        np.testing.assert_allclose(got[k], v, rtol=1e-6, atol=1e-9, err_msg=k)
    for k, v in state.stats.items():
        np.testing.assert_allclose(tc.S[k].numpy(), v, rtol=1e-6, atol=1e-10, err_msg=k)
    for k, v in state.ema_stats.items():
        np.testing.assert_allclose(tc.ema_stats[k].numpy(), v, rtol=1e-6, atol=1e-10, err_msg=k)


def test_focal_gradient_matches_autograd():
    """focal_loss_factor / sigmoid_focal_loss_factor (convnet.py:581-592): the oracle's closed-form gradient against torch autograd of the reference's
    own expression — pow(1 - sum(Y * softmax), gamma) differentiated through the softmax, the sigmoid factor detached (tf.stop_gradient)."""
    rng = np.random.default_rng(31)
    B, C = 6, 9
    z = rng.standard_normal((B, C))
    y = np.eye(C)[rng.integers(0, C, B)]
    y[1] = 0.0
    w = 0.5 + rng.random(C)
    for fg, sa, ls in ((2.0, 0.0, 0.0), (0.0, 4.0, 0.0), (1.5, 3.0, 0.1)):
        from oracle import ops as O
        pred, loss, ce, dl = O.softmax_xent_fwd_bwd(z, y, w, ls, 1.0, focal_gamma=fg, sigmoid_focal_alpha=sa)
        zt = torch.tensor(z, dtype=torch.float64, requires_grad=True)
        yt, wt = torch.tensor(y), torch.tensor(w)
        p = torch.softmax(zt, -1)
        labels = yt * (1 - ls) + ls / C if ls > 0 else yt
        losses = -(labels * torch.log_softmax(zt, -1)).sum(-1)
        pt = (yt * p).sum(-1)
        if fg > 0:
            losses = losses * torch.pow(1.0 - pt, fg)
        if sa > 0:
            losses = losses * ((1.0 - torch.sigmoid(sa * (pt - 0.5))).detach() / (1.0 - torch.sigmoid(torch.tensor(-0.5 * sa, dtype=torch.float64))))
        sy = yt.sum(-1)
        valid = ((sy > 1 - 1e-5) & (sy < 1 + 1e-5)).double()
        tl = ((yt * wt).sum(-1) * valid * losses).mean()
        tl.backward()
        assert abs(float(tl) - float(loss)) <= 1e-12 * abs(float(loss))
        np.testing.assert_allclose(dl, zt.grad.numpy(), rtol=1e-10, atol=1e-13)
