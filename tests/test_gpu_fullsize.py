"""GPU tests at BASELINE.json's full sizes (configs[1] / configs[2]: ResNet-v1.5-50, 224x224, B = 256), where the NumPy
oracle would need hours: size-independent properties that any indexing / tiling / split error at these sizes breaks.

* every ResNet-50 conv geometry at B = 256: the adjoint identities <conv(x), dy> = <x, dgrad(dy)> = <w, wgrad(x, dy)>
  (forward, data gradient and weight gradient are the three faces of one bilinear form) and linearity of the forward;
* batch norm on the stem's 3.2 M-row tensor: statistics against an independent float64 reduction, and the two orthogonality
  relations of the backward pass (sum dx = 0, sum dx * xhat = 0 per channel);
* max-pool: the gradient routing conserves mass; softmax cross-entropy: rows of pred sum to 1, rows of dlogits to 0;
* the fused Nesterov / L2 / EMA update on the full 25.6 M-element parameter vector against the formula;
* the whole network, one training step at B = 256: finite loss near ln(1000) + L2 term, bit-identical when repeated.
The small-size parity of the same kernels against the oracle is in test_gpu_ops.py / test_gpu_net.py.
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B = 256
# (H, Cin, Cout, k, stride): the 23 distinct conv geometries of ResNet-v1.5-50
R50_CONVS = [(224, 3, 64, 7, 2), (56, 64, 64, 1, 1), (56, 64, 256, 1, 1), (56, 256, 64, 1, 1), (56, 64, 64, 3, 1), (56, 256, 128, 1, 1),
             (56, 256, 512, 1, 2), (56, 128, 128, 3, 2), (28, 128, 512, 1, 1), (28, 512, 128, 1, 1), (28, 128, 128, 3, 1), (28, 512, 256, 1, 1),
             (28, 512, 1024, 1, 2), (28, 256, 256, 3, 2), (14, 256, 1024, 1, 1), (14, 1024, 256, 1, 1), (14, 256, 256, 3, 1), (14, 1024, 512, 1, 1),
             (14, 1024, 2048, 1, 2), (14, 512, 512, 3, 2), (7, 512, 2048, 1, 1), (7, 2048, 512, 1, 1), (7, 512, 512, 3, 1)]


def _u():
    import abi_util
    return abi_util


def dot64(a, b):
    return float((a.double().flatten() * b.double().flatten()).sum().item())


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
@pytest.mark.parametrize('layer', R50_CONVS, ids=lambda l: 'h{}_{}to{}_k{}s{}'.format(*l))
def test_conv_adjoint_identities_at_b256(layer, dtype):
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    h, cin, cout, k, s = layer
    gen = torch.Generator(device=u.DEV).manual_seed(h * 1000 + cin + cout + k)
    td = u.TDT[dtype]
    ce = 4 if dtype == 'float32' else 8
    cs = cin if cin % ce == 0 else (cin + ce - 1) // ce * ce             # the stem's 3 channels live in one 16-byte chunk
    x = torch.zeros((B, h, h, cs), device=u.DEV, dtype=td)
    x[..., :cin] = torch.randn((B, h, h, cin), device=u.DEV, generator=gen).to(td)
    w = (torch.randn((k, k, cin, cout), device=u.DEV, generator=gen) / np.sqrt(k * k * cin)).float()
    g = u.geom((B, h, h, cin), (k, k, cin, cout), s, 'SAME', x_cs=cs if cs != cin else 0)
    oh = -(-h // s)
    y = torch.full((B, oh, oh, cout), float('nan'), device=u.DEV, dtype=td)
    dy = torch.randn((B, oh, oh, cout), device=u.DEV, generator=gen).to(td)
    ws = u.workspace(max(lib.mcn_conv2d_workspace_bytes(op, ctypes.byref(g), u.MDT[dtype]) for op in (_ffi.CONV_FWD, _ffi.CONV_DGRAD, _ffi.CONV_WGRAD)))
    st = u.stream()
    _ffi.check(lib.mcn_conv2d_fwd(x.data_ptr(), w.data_ptr(), 0, 0, y.data_ptr(), ctypes.byref(g), u.MDT[dtype], _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, st))
    assert torch.isfinite(y.float()).all()
    wq = w.to(td).float()                                                 # the weights as the kernels use them (rounded per use)
    lhs = dot64(y, dy)
    scale = float(y.double().norm().item() * dy.double().norm().item())
    tol = 2e-5 if dtype == 'float32' else 6e-3                            # bf16: y and dx are rounded to 8 bits once
    # <conv(x), dy> = <w, wgrad(x, dy)>
    dw = torch.full((k, k, cin, cout), float('nan'), device=u.DEV, dtype=torch.float32)
    _ffi.check(lib.mcn_conv2d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 0, ctypes.byref(g), 1.0, u.MDT[dtype], _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, st))
    assert torch.isfinite(dw).all()
    assert abs(lhs - dot64(wq, dw)) <= tol * scale, ('wgrad adjoint', lhs, dot64(wq, dw), scale)
    # <conv(x), dy> = <x, dgrad(dy)>   (the stem has no data gradient in the network: its dx is not on the MFMA path either)
    if cs == cin:
        dx = torch.full((B, h, h, cin), float('nan'), device=u.DEV, dtype=td)
        _ffi.check(lib.mcn_conv2d_dgrad(dy.data_ptr(), w.data_ptr(), 0, dx.data_ptr(), ctypes.byref(g), 0, u.MDT[dtype], _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, st))
        assert torch.isfinite(dx.float()).all()
        assert abs(lhs - dot64(x, dx)) <= tol * scale, ('dgrad adjoint', lhs, dot64(x, dx), scale)
    # linearity in x (fp32: exact products, fp32 accumulation — only summation-order noise)
    if dtype == 'float32':
        x2 = torch.zeros_like(x)
        x2[..., :cin] = torch.randn((B, h, h, cin), device=u.DEV, generator=gen)
        y2, y12 = torch.empty_like(y), torch.empty_like(y)
        _ffi.check(lib.mcn_conv2d_fwd(x2.data_ptr(), w.data_ptr(), 0, 0, y2.data_ptr(), ctypes.byref(g), u.MDT[dtype], _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, st))
        x12 = 0.5 * x - 2.0 * x2
        _ffi.check(lib.mcn_conv2d_fwd(x12.data_ptr(), w.data_ptr(), 0, 0, y12.data_ptr(), ctypes.byref(g), u.MDT[dtype], _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, st))
        err = float((y12 - (0.5 * y - 2.0 * y2)).double().norm().item() / y12.double().norm().item())
        assert err <= 2e-6, ('linearity', err)


SPOT = (0, 127, 255)


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
@pytest.mark.parametrize('layer', R50_CONVS, ids=lambda l: 'h{}_{}to{}_k{}s{}'.format(*l))
def test_conv_oracle_spot_parity_at_b256(layer, dtype):
    """The ORACLE at the bench's own launch sizes (the adjoint identities above cannot see an indexing error common to
    forward, dgrad and wgrad — shared tile / transform bookkeeping, 32-bit byte offsets, tile rounds beyond one pass of the chip — that only
    shows at B = 256).  A convolution is independent per image, so images 0, 127 and 255 of the full-size launch are compared with
    oracle.ops.conv2d_fwd / conv2d_dgrad on those images; the weight gradient of a dy that is zero outside them with conv2d_wgrad on them
    (every split / slab of the full-size wgrad launch still runs).  Same per-op bars as tests/test_gpu_ops.py."""
    from myconvnet_amd import _ffi
    from oracle import ops as O
    from test_gpu_ops import check
    u = _u()
    lib = _ffi.lib
    h, cin, cout, k, s = layer
    gen = torch.Generator(device=u.DEV).manual_seed(h * 977 + cin * 3 + cout + k)
    td = u.TDT[dtype]
    ce = 4 if dtype == 'float32' else 8
    cs = cin if cin % ce == 0 else (cin + ce - 1) // ce * ce
    x = torch.zeros((B, h, h, cs), device=u.DEV, dtype=td)
    x[..., :cin] = torch.randn((B, h, h, cin), device=u.DEV, generator=gen).to(td)
    w = (torch.randn((k, k, cin, cout), device=u.DEV, generator=gen) / np.sqrt(k * k * cin)).float()
    g = u.geom((B, h, h, cin), (k, k, cin, cout), s, 'SAME', x_cs=cs if cs != cin else 0)
    oh = -(-h // s)
    ws = u.workspace(max(lib.mcn_conv2d_workspace_bytes(op, ctypes.byref(g), u.MDT[dtype]) for op in (_ffi.CONV_FWD, _ffi.CONV_DGRAD, _ffi.CONV_WGRAD)))
    st = u.stream()
    sel = list(SPOT)
    xs = x[sel][..., :cin].double().cpu().numpy()                          # the stored (already rounded) values of the three images
    wq = w.to(td).double().cpu().numpy()                                  # the weights as the kernels use them
    # forward
    y = torch.full((B, oh, oh, cout), float('nan'), device=u.DEV, dtype=td)
    _ffi.check(lib.mcn_conv2d_fwd(x.data_ptr(), w.data_ptr(), 0, 0, y.data_ptr(), ctypes.byref(g), u.MDT[dtype], _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, st))
    assert torch.isfinite(y.float()).all()
    check(y[sel].float().cpu().numpy(), O.conv2d_fwd(xs, wq, s, 'SAME', 1), dtype, 'fwd images {}'.format(SPOT))
    # data gradient (the stem has none in the network)
    dy = torch.randn((B, oh, oh, cout), device=u.DEV, generator=gen).to(td)
    dys = dy[sel].double().cpu().numpy()
    if cs == cin:
        dx = torch.full((B, h, h, cin), float('nan'), device=u.DEV, dtype=td)
        _ffi.check(lib.mcn_conv2d_dgrad(dy.data_ptr(), w.data_ptr(), 0, dx.data_ptr(), ctypes.byref(g), 0, u.MDT[dtype], _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, st))
        assert torch.isfinite(dx.float()).all()
        check(dx[sel].float().cpu().numpy(), O.conv2d_dgrad(dys, wq, (3, h, h, cin), s, 'SAME', 1), dtype, 'dgrad images {}'.format(SPOT))
    # weight gradient: dy zero outside the three images
    dyz = torch.zeros_like(dy)
    dyz[sel] = dy[sel]
    dw = torch.full((k, k, cin, cout), float('nan'), device=u.DEV, dtype=torch.float32)
    _ffi.check(lib.mcn_conv2d_wgrad(x.data_ptr(), dyz.data_ptr(), dw.data_ptr(), 0, ctypes.byref(g), 1.0, u.MDT[dtype], _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, st))
    check(dw.cpu().numpy(), O.conv2d_wgrad(xs, dys, (k, k, cin, cout), s, 'SAME', 1), 'float32', 'wgrad images {}'.format(SPOT), rel=2e-5 if dtype == 'float32' else 2e-5)


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_batch_norm_at_stem_size(dtype):
    """[256, 112, 112, 64]: 3.2 M rows per channel."""
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    td = u.TDT[dtype]
    c, m = 64, B * 112 * 112
    gen = torch.Generator(device=u.DEV).manual_seed(7)
    x = (torch.randn((m, c), device=u.DEV, generator=gen) * (0.5 + torch.rand(c, device=u.DEV, generator=gen)) + 3.0 * torch.randn(c, device=u.DEV, generator=gen)).to(td)
    gamma = (0.5 + torch.rand(c, device=u.DEV, generator=gen)).float()
    beta = (0.3 * torch.randn(c, device=u.DEV, generator=gen)).float()
    y = torch.empty_like(x)
    sm, si, bm, bv = [torch.zeros(c, dtype=torch.float32, device=u.DEV) for _ in range(4)]
    ws = u.workspace(lib.mcn_bn_workspace_bytes(m, c))
    st = u.stream()
    _ffi.check(lib.mcn_bn_fwd_train(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 0, y.data_ptr(), 0, sm.data_ptr(), si.data_ptr(), bm.data_ptr(), bv.data_ptr(), 0, 0,
                                    0.99, m, c, 1e-3, 0, u.MDT[dtype], ws.data_ptr(), ws.numel() * 4, st))
    xd = x.double()
    mean, var = xd.mean(0), xd.var(0, unbiased=False)
    assert float((sm.double() - mean).abs().max() / mean.abs().max()) <= 1e-6
    assert float((si.double() - 1.0 / torch.sqrt(var + 1e-3)).abs().max() * torch.sqrt(var + 1e-3).max()) <= 1e-5
    assert float((bv.double() - var * m / (m - 1)).abs().max() / var.max()) <= 1e-5
    # the normalised output has mean beta and variance gamma^2 * var / (var + eps) per channel
    yd = y.double()
    tol = 1e-4 if dtype == 'float32' else 5e-3
    assert float((yd.mean(0) - beta.double()).abs().max()) <= tol
    assert float((yd.var(0, unbiased=False) - gamma.double() ** 2 * var / (var + 1e-3)).abs().max()) <= tol * 4
    # backward: dx is orthogonal to the all-ones vector and to xhat, channel by channel
    dy = torch.randn((m, c), device=u.DEV, generator=gen).to(td)
    dx = torch.empty_like(x)
    dg, db = torch.zeros(c, dtype=torch.float32, device=u.DEV), torch.zeros(c, dtype=torch.float32, device=u.DEV)
    _ffi.check(lib.mcn_bn_bwd(dy.data_ptr(), x.data_ptr(), 0, 0, gamma.data_ptr(), beta.data_ptr(), sm.data_ptr(), si.data_ptr(), dx.data_ptr(), 0, dg.data_ptr(), db.data_ptr(),
                              1.0, m, c, 0, u.MDT[dtype], ws.data_ptr(), ws.numel() * 4, st))
    xhat = (xd - mean) / torch.sqrt(var + 1e-3)
    dxd, dyd = dx.double(), dy.double()
    ref = float(dxd.abs().sum(0).max())
    btol = 1e-5 if dtype == 'float32' else 3e-3
    assert float(dxd.sum(0).abs().max()) <= btol * ref
    assert float((dxd * xhat).sum(0).abs().max()) <= btol * ref
    assert float((db.double() - dyd.sum(0)).abs().max() / dyd.sum(0).abs().max()) <= 1e-4
    assert float((dg.double() - (dyd * xhat).sum(0)).abs().max() / (dyd * xhat).sum(0).abs().max()) <= 1e-4


# (mcn_conv2d_dgrad_bnred / _addmasked_bnred / mcn_conv2d_fwd_bnstats at B = 256: tests/test_gpu_fullsize_fused.py)


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_maxpool_and_softmax_properties_at_b256(dtype):
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    td = u.TDT[dtype]
    gen = torch.Generator(device=u.DEV).manual_seed(11)
    x = torch.randn((B, 112, 112, 64), device=u.DEV, generator=gen).to(td)
    y = torch.empty((B, 56, 56, 64), device=u.DEV, dtype=td)
    arg = torch.empty((B, 56, 56, 64), device=u.DEV, dtype=torch.int8)
    st = u.stream()
    _ffi.check(lib.mcn_maxpool_fwd(x.data_ptr(), y.data_ptr(), arg.data_ptr(), B, 112, 112, 64, 3, 3, 2, 2, 0, 0, 56, 56, u.MDT[dtype], st))
    ref = torch.nn.functional.max_pool2d(torch.nn.functional.pad(x.float().permute(0, 3, 1, 2), (0, 1, 0, 1), value=float('-inf')), 3, 2)
    assert torch.equal(y.float(), ref.permute(0, 2, 3, 1))                # a maximum is exact in any precision
    dy = torch.rand((B, 56, 56, 64), device=u.DEV, generator=gen).to(td)
    dx = torch.empty_like(x)
    _ffi.check(lib.mcn_maxpool_bwd(dy.data_ptr(), arg.data_ptr(), dx.data_ptr(), B, 112, 112, 64, 3, 3, 2, 2, 0, 0, 56, 56, u.MDT[dtype], st))
    s_dx, s_dy = float(dx.double().sum()), float(dy.double().sum())
    assert abs(s_dx - s_dy) <= (1e-9 if dtype == 'float32' else 2e-3) * s_dy     # routing conserves the gradient mass
    assert int((dx != 0).sum()) <= dy.numel()
    if dtype == 'float32':
        logits = 3.0 * torch.randn((B, 1000), device=u.DEV, generator=gen)
        labels = torch.randint(0, 1000, (B,), device=u.DEV, generator=gen)
        onehot = torch.nn.functional.one_hot(labels, 1000).float()
        pred, dl = torch.empty_like(logits), torch.empty_like(logits)
        ce, coef, loss = torch.zeros(B, device=u.DEV), torch.zeros(B, device=u.DEV), torch.zeros(1, device=u.DEV)
        _ffi.check(lib.mcn_softmax_xent_fwd_bwd(logits.data_ptr(), onehot.data_ptr(), 0, pred.data_ptr(), ce.data_ptr(), coef.data_ptr(), dl.data_ptr(), loss.data_ptr(),
                                                B, 1000, 0.0, 1.0, st))
        assert float((pred.double().sum(1) - 1.0).abs().max()) <= 1e-5
        assert float(dl.double().sum(1).abs().max()) <= 1e-7
        ref_loss = torch.nn.functional.cross_entropy(logits.double(), labels)
        assert abs(float(loss.item()) - float(ref_loss)) <= 1e-5 * float(ref_loss)


def test_nesterov_l2_ema_update_on_the_full_parameter_vector():
    from myconvnet_amd import _ffi
    u = _u()
    n = 25557032 // 4 * 4
    gen = torch.Generator(device=u.DEV).manual_seed(3)
    w, g, a, e = [torch.randn(n, device=u.DEV, generator=gen) * s for s in (0.05, 0.01, 0.02, 0.05)]
    w0, a0, e0 = w.double(), a.double(), e.double()
    lr, mom, l2, wd, d, gs = 0.1, 0.9, 1e-4, 1e-3, 0.97, 0.125
    _ffi.check(_ffi.lib.mcn_sgd_nesterov_fused(w.data_ptr(), g.data_ptr(), a.data_ptr(), e.data_ptr(), n, lr, mom, l2, wd, d, gs, u.stream()))
    gt = gs * g.double() + l2 * w0
    a1 = mom * a0 + gt
    w1 = w0 - lr * gt - lr * mom * a1
    w1 = w1 - wd * w1
    e1 = d * e0 + (1 - d) * w0
    for got, ref in ((w, w1), (a, a1), (e, e1)):
        assert float((got.double() - ref).abs().max() / ref.abs().max()) <= 1e-6


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_resnet50_step_at_baseline_size_is_finite_and_deterministic(dtype):
    """BASELINE configs[1] / [2] arithmetic on one GPU: B = 256, 224x224, 1000 classes, He-normal init."""
    import myconvnet_amd as M
    rng = np.random.default_rng(1234)
    x = rng.random((B, 224, 224, 3), dtype=np.float32)
    y = rng.integers(0, 1000, B).astype(np.float32)
    results = []
    for rep in range(2):
        model = M.ResNet50([224, 224, 3], 1000, batch_size=B, num_gpus=1, half_precision=(dtype == 'bfloat16'), seed=0)
        opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, momentum=0.9, steps_per_epoch=5000, num_epochs=90)
        model.feed(x, y)
        losses = []
        for _ in range(2):
            opt._update_learning_rate()
            loss, y_true, y_pred = opt._step(None)
            losses.append(loss)
        assert np.isfinite(losses).all()
        # random init: the softmax is near uniform over 1000 classes; the L2 term of 25.5 M He-normal weights adds ~0.5-2
        assert np.log(1000.0) - 0.5 <= losses[0] <= np.log(1000.0) + 8.0, losses
        assert y_pred.shape == (B, 1000) and abs(float(y_pred.sum(1).mean()) - 1.0) <= 1e-3
        np.testing.assert_array_equal(y_true.argmax(1), y.astype(np.int64))
        grads = model.store.grad.detach().clone()
        assert torch.isfinite(grads).all() and float(grads.abs().max()) > 0
        results.append((losses, model.store.data.detach().clone(), grads, model.stats.data.detach().clone()))
        del model, opt
        torch.cuda.empty_cache()
    (l0, w0, g0, s0), (l1, w1, g1, s1) = results
    assert l0 == l1                                                       # same kernels, same order, no atomics: bit-identical
    assert torch.equal(w0, w1) and torch.equal(g0, g1) and torch.equal(s0, s1)


# (H, C, k, stride): the depthwise convs of EfficientNet-B0 (BASELINE configs[3], B = 512)
B0_DW = [(112, 32, 3, 1), (112, 96, 3, 2), (56, 144, 3, 1), (56, 144, 5, 2), (28, 240, 5, 1), (28, 240, 3, 2), (14, 480, 3, 1), (14, 480, 5, 1),
         (14, 672, 5, 1), (14, 672, 5, 2), (7, 1152, 5, 1), (7, 1152, 3, 1)]


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
@pytest.mark.parametrize('layer', B0_DW, ids=lambda l: 'h{}_c{}_k{}s{}'.format(*l))
def test_depthwise_adjoint_identities_at_b512(layer, dtype):
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    h, c, k, s = layer
    bsz = 512
    td = u.TDT[dtype]
    gen = torch.Generator(device=u.DEV).manual_seed(h * 100 + c + k)
    x = torch.randn((bsz, h, h, c), device=u.DEV, generator=gen).to(td)
    w = (torch.randn((k, k, c), device=u.DEV, generator=gen) / k).float()
    g = u.dw_geom((bsz, h, h, c), k, s, 'SAME')
    oh = -(-h // s)
    y = torch.full((bsz, oh, oh, c), float('nan'), device=u.DEV, dtype=td)
    dy = torch.randn((bsz, oh, oh, c), device=u.DEV, generator=gen).to(td)
    dx = torch.full((bsz, h, h, c), float('nan'), device=u.DEV, dtype=td)
    dw = torch.full((k, k, c), float('nan'), device=u.DEV, dtype=torch.float32)
    ws = u.workspace(lib.mcn_dwconv2d_workspace_bytes(ctypes.byref(g), u.MDT[dtype]))
    st = u.stream()
    _ffi.check(lib.mcn_dwconv2d_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), ctypes.byref(g), u.MDT[dtype], st))
    _ffi.check(lib.mcn_dwconv2d_dgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), ctypes.byref(g), 0, u.MDT[dtype], st))
    _ffi.check(lib.mcn_dwconv2d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ctypes.byref(g), 1.0, u.MDT[dtype], ws.data_ptr(), ws.numel() * 4, st))
    assert torch.isfinite(y.float()).all() and torch.isfinite(dx.float()).all() and torch.isfinite(dw).all()
    wq = w.to(td).float()
    lhs = dot64(y, dy)
    scale = float(y.double().norm().item() * dy.double().norm().item())
    tol = 2e-5 if dtype == 'float32' else 6e-3
    assert abs(lhs - dot64(x, dx)) <= tol * scale, ('dgrad adjoint', lhs, dot64(x, dx), scale)
    # per channel: <y_c, dy_c> = <w_c, dw_c> — a depthwise conv is C independent bilinear forms
    per_c = (y.double() * dy.double()).sum((0, 1, 2))
    per_w = (wq.double() * dw.double()).sum((0, 1))
    cs = torch.sqrt((y.double() ** 2).sum((0, 1, 2)) * (dy.double() ** 2).sum((0, 1, 2)))
    assert float(((per_c - per_w).abs() / cs).max()) <= tol * 4, ('wgrad adjoint per channel', float(((per_c - per_w).abs() / cs).max()))


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
@pytest.mark.parametrize('layer', B0_DW, ids=lambda l: 'h{}_c{}_k{}s{}'.format(*l))
def test_depthwise_oracle_spot_parity_at_b512(layer, dtype):
    """The oracle on images 0, 255 and 511 of the full-size depthwise launches of configs[3] (see test_conv_oracle_spot_parity_at_b256)."""
    from myconvnet_amd import _ffi
    from oracle import ops as O
    from test_gpu_ops import check
    u = _u()
    lib = _ffi.lib
    h, c, k, s = layer
    bsz = 512
    sel = [0, 255, 511]
    td = u.TDT[dtype]
    gen = torch.Generator(device=u.DEV).manual_seed(h * 131 + c + k)
    x = torch.randn((bsz, h, h, c), device=u.DEV, generator=gen).to(td)
    w = (torch.randn((k, k, c), device=u.DEV, generator=gen) / k).float()
    g = u.dw_geom((bsz, h, h, c), k, s, 'SAME')
    oh = -(-h // s)
    y = torch.full((bsz, oh, oh, c), float('nan'), device=u.DEV, dtype=td)
    dy = torch.randn((bsz, oh, oh, c), device=u.DEV, generator=gen).to(td)
    dyz = torch.zeros_like(dy)
    dyz[sel] = dy[sel]
    dx = torch.full((bsz, h, h, c), float('nan'), device=u.DEV, dtype=td)
    dw = torch.full((k, k, c), float('nan'), device=u.DEV, dtype=torch.float32)
    ws = u.workspace(lib.mcn_dwconv2d_workspace_bytes(ctypes.byref(g), u.MDT[dtype]))
    st = u.stream()
    _ffi.check(lib.mcn_dwconv2d_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), ctypes.byref(g), u.MDT[dtype], st))
    _ffi.check(lib.mcn_dwconv2d_dgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), ctypes.byref(g), 0, u.MDT[dtype], st))
    _ffi.check(lib.mcn_dwconv2d_wgrad(x.data_ptr(), dyz.data_ptr(), dw.data_ptr(), ctypes.byref(g), 1.0, u.MDT[dtype], ws.data_ptr(), ws.numel() * 4, st))
    xs, dys = x[sel].double().cpu().numpy(), dy[sel].double().cpu().numpy()
    wq = w.to(td).double().cpu().numpy().reshape(k, k, c, 1)
    check(y[sel].float().cpu().numpy(), O.depthwise_conv2d_fwd(xs, wq, s, 'SAME', 1), dtype, 'dw fwd images {}'.format(sel))
    check(dx[sel].float().cpu().numpy(), O.depthwise_conv2d_dgrad(dys, wq, xs.shape, s, 'SAME', 1), dtype, 'dw dgrad images {}'.format(sel))
    check(dw.cpu().numpy().reshape(k, k, c, 1), O.depthwise_conv2d_wgrad(xs, dys, (k, k, c, 1), s, 'SAME', 1), 'float32', 'dw wgrad images {}'.format(sel), rel=2e-5)


# the 1x1 convs that run OFF the MFMA path at their BASELINE sizes: the DeepLabv3+ class-logit conv (configs[4]: 16 x 129 x 129 pixels,
# 256 -> 21 with bias) and the squeeze-excite convs of EfficientNet-B0's widest blocks (configs[3]: M = batch = 512 pixels)
SKINNY_FULL = [(16, 129, 256, 21), (512, 1, 672, 28), (512, 1, 28, 672), (512, 1, 1152, 48), (512, 1, 48, 1152)]


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
@pytest.mark.parametrize('layer', SKINNY_FULL, ids=lambda l: 'n{}_h{}_{}to{}'.format(*l))
def test_skinny_and_se_convs_at_baseline_sizes(layer, dtype):
    """forward / dgrad / wgrad / bias gradient against float64 matmuls of the SAME (storage-rounded) operands: a 1x1 conv is a
    matrix product, so torch gives an independent reference at any size."""
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    n, h, cin, cout = layer
    td = u.TDT[dtype]
    gen = torch.Generator(device=u.DEV).manual_seed(n + h + cin)
    x = torch.randn((n, h, h, cin), device=u.DEV, generator=gen).to(td)
    w = (torch.randn((1, 1, cin, cout), device=u.DEV, generator=gen) / cin ** 0.5).float()
    b = torch.randn((cout,), device=u.DEV, generator=gen).float()
    dy = torch.randn((n, h, h, cout), device=u.DEV, generator=gen).to(td)
    g = u.geom((n, h, h, cin), (1, 1, cin, cout), 1, 'SAME')
    y = torch.full((n, h, h, cout), float('nan'), device=u.DEV, dtype=td)
    dx = torch.full((n, h, h, cin), float('nan'), device=u.DEV, dtype=td)
    dw = torch.full((1, 1, cin, cout), float('nan'), device=u.DEV, dtype=torch.float32)
    db = torch.full((cout,), float('nan'), device=u.DEV, dtype=torch.float32)
    ws = u.workspace(max(lib.mcn_conv2d_workspace_bytes(op, ctypes.byref(g), u.MDT[dtype]) for op in (_ffi.CONV_FWD, _ffi.CONV_DGRAD, _ffi.CONV_WGRAD)))
    st = u.stream()
    _ffi.check(lib.mcn_conv2d_fwd(x.data_ptr(), w.data_ptr(), 0, b.data_ptr(), y.data_ptr(), ctypes.byref(g), u.MDT[dtype], _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, st))
    _ffi.check(lib.mcn_conv2d_dgrad(dy.data_ptr(), w.data_ptr(), 0, dx.data_ptr(), ctypes.byref(g), 0, u.MDT[dtype], _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, st))
    _ffi.check(lib.mcn_conv2d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), ctypes.byref(g), 1.0, u.MDT[dtype], _ffi.NHWC, ws.data_ptr(),
                                    ws.numel() * 4, st))
    xm, dym = x.double().reshape(-1, cin), dy.double().reshape(-1, cout)
    wq = w.to(td).double().reshape(cin, cout)                       # weights are rounded to the storage type per use
    out_tol = 1e-4 if dtype == 'float32' else 1e-2                  # outputs are stored in T (bf16: 2^-8 rounding), parameter gradients in fp32
    par_tol = 1e-4                                                  # fp32 accumulation over up to 266 256 pixels against float64

    def close(got, ref, tol, what):
        err = float((got.double().reshape(ref.shape) - ref).abs().max() / ref.abs().max())
        assert err <= tol, (what, err)

    close(y, xm @ wq + b.double(), out_tol, 'forward + bias')
    close(dx, dym @ wq.t(), out_tol, 'dgrad')
    close(dw, xm.t() @ dym, par_tol, 'wgrad')
    close(db, dym.sum(0), par_tol, 'bias gradient')


def _step_twice_at_baseline(build, x, y, check):
    """Two fresh models, two steps each: finite, in the loss window of a random init, bit-identical when repeated."""
    import myconvnet_amd as M
    results = []
    for rep in range(2):
        model = build()
        opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, momentum=0.9, steps_per_epoch=5000, num_epochs=90)
        model.fixed_random_masks = None
        model._mask_rng = np.random.default_rng(99)                       # same stochastic-depth / dropout draws in both repeats
        model.feed(x, y)
        losses = []
        for _ in range(2):
            opt._update_learning_rate()
            loss, y_true, y_pred = opt._step(None)
            losses.append(loss)
        assert np.isfinite(losses).all(), losses
        check(losses, y_true, y_pred)
        grads = model.store.grad.detach().clone()
        assert torch.isfinite(grads).all() and float(grads.abs().max()) > 0
        assert torch.isfinite(model.store.data).all() and torch.isfinite(model.stats.data).all()
        results.append((losses, model.store.data.detach().clone(), grads, model.stats.data.detach().clone()))
        del model, opt
        torch.cuda.empty_cache()
    (l0, w0, g0, s0), (l1, w1, g1, s1) = results
    assert l0 == l1
    assert torch.equal(w0, w1) and torch.equal(g0, g1) and torch.equal(s0, s1)


def test_efficientnet_b0_step_at_baseline_size_is_finite_and_deterministic():
    """BASELINE configs[3]: EfficientNet-B0 bf16, 224x224, B = 512 on one GPU (depthwise + swish + squeeze-excite +
    stochastic depth + dropout path), whole network, He / variance-scaling init."""
    import myconvnet_amd as M
    Bn = 512
    rng = np.random.default_rng(1234)
    x = rng.random((Bn, 224, 224, 3), dtype=np.float32)
    y = rng.integers(0, 1000, Bn).astype(np.float32)

    def check(losses, y_true, y_pred):
        assert np.log(1000.0) - 0.5 <= losses[0] <= np.log(1000.0) + 8.0, losses
        assert y_pred.shape == (Bn, 1000) and abs(float(y_pred.sum(1).mean()) - 1.0) <= 1e-3
        np.testing.assert_array_equal(y_true.argmax(1), y.astype(np.int64))
    _step_twice_at_baseline(lambda: M.EfficientNetB0([224, 224, 3], 1000, batch_size=Bn, num_gpus=1, half_precision=True, seed=0), x, y, check)


def test_deeplabv3plus_step_at_baseline_size_is_finite_and_deterministic():
    """BASELINE configs[4] on one GPU: DeepLabv3+ (ResNet-50 OS16), 513x513 synthetic Cityscapes (19 classes, label 0 =
    ignored pixel), B = 16, bf16: dilated 3x3 convs, ASPP, align_corners resize, concat, per-pixel loss."""
    import myconvnet_amd as M
    Bn, S, C = 16, 513, 19
    rng = np.random.default_rng(1234)
    x = rng.random((Bn, S, S, 3), dtype=np.float32)
    y = rng.integers(0, C + 1, (Bn, S, S)).astype(np.float32)

    def check(losses, y_true, y_pred):
        # per-pixel CE averaged over ALL pixels (ignored ones contribute 0, convnet.py:594): ~ (C/(C+1)) ln C at a random init
        assert 0.5 * np.log(C) <= losses[0] <= np.log(C) + 8.0, losses
        assert y_pred.shape == (Bn, S, S, C)
        valid = y > 0
        np.testing.assert_array_equal(y_true.sum(-1) == 1, valid)
        np.testing.assert_array_equal(y_true.argmax(-1)[valid], (y[valid] - 1).astype(np.int64))
        assert abs(float(y_pred[0].sum(-1).mean()) - 1.0) <= 1e-3
    _step_twice_at_baseline(lambda: M.DeepLabV3PlusResNet50([S, S, 3], C, batch_size=Bn, num_gpus=1, half_precision=True, seed=0), x, y, check)
