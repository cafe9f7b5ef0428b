"""GPU parity, EfficientNet MBConv row: depthwise conv fwd / dgrad / wgrad, BN+swish, swish / sigmoid,
squeeze-excite channel scale — HIP kernels through the C-ABI vs the NumPy oracle.  Tolerances as in test_gpu_ops.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ops as O  # noqa: E402
from test_gpu_ops import check, q, DTYPES  # noqa: E402


def _u():
    import abi_util
    return abi_util


RNG = np.random.default_rng(23)

# n, h, w, c, k, stride, padding, dilation  (EfficientNet-B0 uses k in {3,5}, s in {1,2}, C in {32..1152})
DW_CASES = [
    (2, 12, 12, 32, 3, 1, 'SAME', 1),
    (2, 12, 12, 96, 3, 2, 'SAME', 1),            # even H stride 2: pads (0,1)
    (2, 9, 9, 144, 5, 2, 'SAME', 1),             # odd H stride 2 5x5: pads (2,2)
    (3, 14, 14, 240, 5, 1, 'SAME', 1),
    (2, 7, 7, 1152, 3, 1, 'SAME', 1),            # widest B0 layer
    (2, 7, 5, 672, 5, 1, 'SAME', 1),             # non-square
    (1, 10, 10, 40, 7, 1, 'SAME', 1),            # generic filter size (tap-per-block wgrad)
    (2, 9, 9, 16, 3, 1, 'VALID', 2),             # VALID + dilation
    (1, 1, 1, 8, 3, 1, 'SAME', 1),               # 1x1 map
    (2, 12, 14, 48, 5, 2, 'SAME', 1),            # even H stride 2 5x5: pads (1,2) — the EfficientNet 56 -> 28 / 14 -> 7 case
    (2, 9, 11, 24, 3, 2, 'SAME', 1),             # odd H stride 2 3x3: pads (1,1), width not a multiple of the strip
    (1, 13, 10, 16, 5, 2, 'SAME', 1),            # odd H / even W: pads (2,2) x (1,2)
    # larger maps: the LDS-band kernels cut the image into several row / strip tiles (halo rows shared between tiles, last tiles partly empty)
    (2, 40, 44, 48, 5, 1, 'SAME', 1),
    (3, 33, 31, 72, 3, 1, 'SAME', 1),            # odd sizes: strips and tiles with ragged ends
    (2, 28, 28, 240, 3, 2, 'SAME', 1),           # stride 2 on the band path (output 14x14)
    (40, 56, 56, 32, 3, 1, 'SAME', 1),           # more tiles than wgrad workgroups: accumulators carried across tiles
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', DW_CASES)
def test_depthwise_fwd_dgrad_wgrad(case, dtype):
    u = _u()
    n, h, w_, c, k, s, pad, dil = case
    x = RNG.standard_normal((n, h, w_, c)).astype(np.float32)
    w = (RNG.standard_normal((k, k, c, 1)) / k).astype(np.float32)
    xq, wq = q(x, dtype), q(w, dtype)
    y_ref = O.depthwise_conv2d_fwd(xq, wq, s, pad, dil)
    check(u.dwconv_fwd(x, w, s, pad, dil, dtype), y_ref, dtype, 'dw_fwd')
    dy = RNG.standard_normal(y_ref.shape).astype(np.float32)
    dyq = q(dy, dtype)
    check(u.dwconv_dgrad(dy, w, x.shape, s, pad, dil, dtype), O.depthwise_conv2d_dgrad(dyq, wq, x.shape, s, pad, dil), dtype, 'dw_dgrad')
    base = RNG.standard_normal(x.shape).astype(np.float32)
    check(u.dwconv_dgrad(dy, w, x.shape, s, pad, dil, dtype, accumulate_into=base),
          q(base, dtype) + O.depthwise_conv2d_dgrad(dyq, wq, x.shape, s, pad, dil), dtype, 'dw_dgrad accumulate')
    dw_ref = O.depthwise_conv2d_wgrad(xq, dyq, w.shape, s, pad, dil)
    check(u.dwconv_wgrad(x, dy, k, s, pad, dil, dtype, scale=0.5), 0.5 * dw_ref, 'float32', 'dw_wgrad', rel=2e-5)   # fp32 sums of exact products


def test_depthwise_rejects_unbuilt_options():
    import ctypes
    import torch
    from myconvnet_amd import _ffi
    g = _ffi.conv_geom(1, 4, 4, 8, 16, 3, 3, 1, 1, 1, 1, (1, 1, 1, 1), 0)      # channel multiplier 2
    t = torch.zeros(1024, device='cuda')
    assert _ffi.lib.mcn_dwconv2d_fwd(t.data_ptr(), t.data_ptr(), t.data_ptr(), ctypes.byref(g), _ffi.F32, 0) == _ffi.E_UNSUPPORTED
    g = _ffi.conv_geom(1, 4, 4, 6, 6, 3, 3, 1, 1, 1, 1, (1, 1, 1, 1), 0)       # C not a chunk multiple
    assert _ffi.lib.mcn_dwconv2d_fwd(t.data_ptr(), t.data_ptr(), t.data_ptr(), ctypes.byref(g), _ffi.F32, 0) == _ffi.E_UNSUPPORTED


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(4, 6, 6, 16), (8, 9, 9, 96), (2, 3, 3, 1152)])
def test_bn_swish_fwd_bwd(shape, dtype):
    """normalization + swish of every EfficientNet conv (models/efficientnet.py:62-67): fused apply and backward."""
    u = _u()
    c = shape[-1]
    x = (1.5 * RNG.standard_normal(shape) + 0.3).astype(np.float32)
    g = (0.5 + RNG.random(c)).astype(np.float32)
    b = (0.3 * RNG.standard_normal(c)).astype(np.float32)
    xq = q(x, dtype)
    z_ref, bm, bv, sm, si = O.bn_fwd_train(xq, g.astype(np.float64), b.astype(np.float64), 1e-3)
    out = u.bn_fwd_train(x, g, b, 1e-3, dtype, act=2)
    check(out['y'], O.swish_fwd(z_ref), dtype, 'bn+swish y')
    check(out['batch_mean'], bm, 'float32', 'batch_mean', rel=1e-5)
    dy = RNG.standard_normal(shape).astype(np.float32)
    dz = O.swish_bwd(q(dy, dtype), z_ref)
    dx_ref, dg_ref, db_ref = O.bn_bwd(dz, xq, g.astype(np.float64), sm, si)
    dx, dg, db, _ = u.bn_bwd(dy, x, None, g, out['save_mean'], out['save_invstd'], dtype, act=2, beta=b)
    check(dx, dx_ref, dtype, 'bn+swish dx', rel=None if dtype == 'float32' else 1.2e-2)
    check(dg, dg_ref, 'float32', 'dgamma', rel=1e-4 if dtype == 'float32' else 5e-3, mx=5e-2)
    check(db, db_ref, 'float32', 'dbeta', rel=1e-4 if dtype == 'float32' else 5e-3, mx=5e-2)
    # inference path with the same activation
    mean, var = RNG.standard_normal(c).astype(np.float32) * 0.1, (0.5 + RNG.random(c)).astype(np.float32)
    yi = u.bn_fwd_infer(x, g, b, mean, var, 1e-3, dtype, act=2)
    check(yi, O.swish_fwd(O.bn_fwd_infer(xq, g.astype(np.float64), b.astype(np.float64), mean.astype(np.float64), var.astype(np.float64), 1e-3)), dtype, 'bn infer + swish')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(3, 7, 7, 1152), (2, 14, 14, 672), (5, 28, 28, 40), (2, 56, 56, 96), (4, 1, 1, 16), (3, 9, 5, 24)])
def test_bn_swish_with_pooled_means_in_one_pass(shape, dtype):
    """mcn_bn_fwd_train_gap (round 4): BN + swish whose apply pass also leaves the squeeze-excite squeeze (per-image channel means of the
    STORED output, models/efficientnet.py:183).  y and every statistic bit-identical to mcn_bn_fwd_train; the means against the oracle's mean of
    the stored y and against mcn_global_avgpool_fwd on it (fp32 sums in another order, one rounding to the storage type)."""
    import torch
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    n, h, w_, c = shape
    x = (1.5 * RNG.standard_normal(shape) + 0.3).astype(np.float32)
    gamma = (0.5 + RNG.random(c)).astype(np.float32)
    beta = (0.3 * RNG.standard_normal(c)).astype(np.float32)
    ref = u.bn_fwd_train(x, gamma, beta, 1e-3, dtype, act=2, running=(np.zeros(c, np.float32), np.ones(c, np.float32)))
    md, td = u.MDT[dtype], u.TDT[dtype]
    xd, gd, bd = u.dev(x, dtype), u.dev(gamma), u.dev(beta)
    y = torch.full(shape, float('nan'), dtype=td, device=u.DEV)
    gap = torch.full((n, c), float('nan'), dtype=td, device=u.DEV)
    sm, si, bm, bv = [torch.zeros(c, dtype=torch.float32, device=u.DEV) for _ in range(4)]
    rm, rv = u.dev(np.zeros(c, np.float32)), u.dev(np.ones(c, np.float32))
    m = n * h * w_
    ws = u.workspace(lib.mcn_bn_workspace_bytes(m, c))
    _ffi.check(lib.mcn_bn_fwd_train_gap(xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), y.data_ptr(), gap.data_ptr(), sm.data_ptr(), si.data_ptr(), bm.data_ptr(), bv.data_ptr(),
                                        rm.data_ptr(), rv.data_ptr(), 0.99, n, h * w_, c, 1e-3, 2, md, ws.data_ptr(), ws.numel() * 4, u.stream()))
    yh = u.host(y)
    np.testing.assert_array_equal(yh, ref['y'])
    for got, key in ((sm, 'save_mean'), (si, 'save_invstd'), (bm, 'batch_mean'), (bv, 'batch_var'), (rm, 'running_mean'), (rv, 'running_var')):
        np.testing.assert_array_equal(u.host(got), ref[key])
    want = yh.astype(np.float64).reshape(n, h * w_, c).mean(1)
    check(u.host(gap), want, dtype, 'pooled means', rel={'float32': 2e-6, 'bfloat16': 4e-3, 'float16': 5e-4}[dtype])
    g2 = torch.full((n, c), float('nan'), dtype=td, device=u.DEV)
    _ffi.check(lib.mcn_global_avgpool_fwd(y.data_ptr(), g2.data_ptr(), n, h * w_, c, md, u.stream()))
    a, b = u.host(gap).astype(np.float64), u.host(g2).astype(np.float64)
    # two orders of the same fp32 sum (values of magnitude ~1: a few 1e-7 apart), then one rounding to the storage type: at most one storage ulp
    ulp = {'float32': 0.0, 'bfloat16': 2.0 ** -7, 'float16': 2.0 ** -10}[dtype]
    assert np.all(np.abs(a - b) <= 1e-6 + ulp * np.abs(b)), float(np.abs(a - b).max())
    # the squeeze-excite form without the stored output: y = NULL leaves the same means and statistics; mcn_bn_act_scale_fwd rebuilds y from x inside the channel scale
    if c % (4 if dtype == 'float32' else 8) == 0:
        gap3 = torch.full((n, c), float('nan'), dtype=td, device=u.DEV)
        sm3, si3 = torch.zeros(c, dtype=torch.float32, device=u.DEV), torch.zeros(c, dtype=torch.float32, device=u.DEV)
        _ffi.check(lib.mcn_bn_fwd_train_gap(xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), 0, gap3.data_ptr(), sm3.data_ptr(), si3.data_ptr(), 0, 0, 0, 0, 0.99, n, h * w_, c, 1e-3, 2, md,
                                            ws.data_ptr(), ws.numel() * 4, u.stream()))
        np.testing.assert_array_equal(u.host(gap3), u.host(gap))
        np.testing.assert_array_equal(u.host(sm3), ref['save_mean'])
        mk = u.dev(RNG.random((n, c)).astype(np.float32), dtype)
        ys_ref = torch.full(shape, float('nan'), dtype=td, device=u.DEV)
        _ffi.check(lib.mcn_channel_scale_fwd(y.data_ptr(), mk.data_ptr(), ys_ref.data_ptr(), n, h * w_, c, md, u.stream()))
        ys = torch.full(shape, float('nan'), dtype=td, device=u.DEV)
        _ffi.check(lib.mcn_bn_act_scale_fwd(xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), sm3.data_ptr(), si3.data_ptr(), mk.data_ptr(), ys.data_ptr(), n, h * w_, c, 2, md, u.stream()))
        a, b = u.host(ys).astype(np.float64), u.host(ys_ref).astype(np.float64)
        # (scale / shift rebuilt as gamma * invstd, beta - mean * that: the last bit may differ from the finalize kernel's -> rare one-ulp differences of the storage type)
        ulp1 = {'float32': 2.0 ** -21, 'bfloat16': 2.0 ** -7, 'float16': 2.0 ** -10}[dtype]
        assert np.all(np.abs(a - b) <= ulp1 * np.maximum(np.abs(b), 1e-3) + 1e-7) and (dtype == 'float32' or (a != b).mean() < 0.01), (float(np.abs(a - b).max()), float((a != b).mean()))


@pytest.mark.parametrize('dtype', DTYPES)
def test_swish_sigmoid_elementwise(dtype):
    u = _u()
    from myconvnet_amd import _ffi
    x = (3.0 * RNG.standard_normal((37, 24))).astype(np.float32)          # 888 elements: vector body + scalar tail
    dy = RNG.standard_normal(x.shape).astype(np.float32)
    xq, dyq = q(x, dtype), q(dy, dtype)
    y, dx = u.act(x, dy, _ffi.ACT_SWISH, dtype)
    check(y, O.swish_fwd(xq), dtype, 'swish')
    check(dx, O.swish_bwd(dyq, xq), dtype, 'swish bwd')
    y, dx = u.act(x, dy, _ffi.ACT_SIGMOID, dtype)
    check(y, O.sigmoid_fwd(xq), dtype, 'sigmoid')
    check(dx, O.sigmoid_bwd(dyq, q(y, dtype)), dtype, 'sigmoid bwd')        # differentiates through the stored (rounded) y
    y, dx = u.act(x, dy, _ffi.ACT_RELU, dtype)
    check(y, O.relu_fwd(xq), dtype, 'relu')


@pytest.mark.parametrize('dtype', DTYPES)
def test_relu6_lrelu_tanh_elementwise(dtype):
    """the remaining entries of the reference's activation dispatcher (convnet.py:2514-2547): tf.nn.relu6 / Relu6Grad,
    tf.nn.leaky_relu(alpha) / LeakyReluGrad (default alpha 0.2 and an explicit one), tf.nn.tanh / TanhGrad"""
    u = _u()
    from myconvnet_amd import _ffi
    x = (4.0 * RNG.standard_normal((37, 24))).astype(np.float32)          # values on both sides of 0 and of 6; vector body + scalar tail
    x[0, :6] = [0.0, 6.0, -0.0, 7.5, 5.9921875, -3.0]                     # the kinks themselves (bf16-representable)
    dy = RNG.standard_normal(x.shape).astype(np.float32)
    xq, dyq = q(x, dtype), q(dy, dtype)
    y, dx = u.act(x, dy, _ffi.ACT_RELU6, dtype)
    np.testing.assert_array_equal(y, O.relu6_fwd(xq))                     # clamps are exact in every storage type
    np.testing.assert_array_equal(dx, O.relu6_bwd(dyq, xq))
    for alpha in (None, 0.05):
        y, dx = u.act(x, dy, _ffi.ACT_LRELU, dtype, param=alpha)
        a = 0.2 if alpha is None else alpha
        check(y, O.lrelu_fwd(xq, np.float32(a)), dtype, 'lrelu')
        check(dx, O.lrelu_bwd(dyq, xq, np.float32(a)), dtype, 'lrelu bwd')
    y, dx = u.act(x, dy, _ffi.ACT_TANH, dtype)
    check(y, O.tanh_fwd(xq), dtype, 'tanh')
    check(dx, O.tanh_bwd(dyq, q(y, dtype)), dtype, 'tanh bwd')            # differentiates through the stored (rounded) y


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(3, 7, 7, 32), (4, 5, 9, 144), (2, 14, 14, 480), (2, 1, 1, 8)])
def test_se_channel_scale(shape, dtype):
    u = _u()
    n, h, w, c = shape
    x = RNG.standard_normal(shape).astype(np.float32)
    m = RNG.random((n, 1, 1, c)).astype(np.float32)
    dy = RNG.standard_normal(shape).astype(np.float32)
    xq, mq, dyq = q(x, dtype), q(m, dtype), q(dy, dtype)
    y, dx, dm = u.channel_scale(x, m, dy, dtype)
    check(y, O.channel_scale_fwd(xq, mq), dtype, 'se scale')
    dx_ref, dm_ref = O.channel_scale_bwd(dyq, xq, mq)
    check(dx, dx_ref, dtype, 'se dx')
    check(dm, dm_ref, dtype, 'se dm')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(3, 7, 7, 32), (4, 5, 9, 144), (2, 14, 14, 480)])
def test_bn_backward_composes_squeeze_excite_gradient(shape, dtype):
    """mcn_bn_bwd_se + mcn_channel_scale_bwd_dm against mcn_channel_scale_bwd + mcn_global_avgpool_bwd_acc + mcn_bn_bwd(swish): dx, dgamma,
    dbeta and dm bit for bit."""
    import ctypes
    import torch
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    n, h, w_, c = shape
    hw, m_ = h * w_, n * h * w_
    x = (1.2 * RNG.standard_normal(shape) + 0.1).astype(np.float32)
    gamma = (0.5 + RNG.random(c)).astype(np.float32)
    beta = (0.3 * RNG.standard_normal(c)).astype(np.float32)
    out = u.bn_fwd_train(x, gamma, beta, 1e-3, dtype, act=2)                 # BN + swish: x_se
    xd, gd, bd = u.dev(x, dtype), u.dev(gamma), u.dev(beta)
    xse = u.dev(out['y'], dtype)
    sm, si = u.dev(out['save_mean']), u.dev(out['save_invstd'])
    mk = u.dev(RNG.random((n, c)).astype(np.float32), dtype)
    dy = u.dev(RNG.standard_normal(shape).astype(np.float32), dtype)
    dgap = u.dev(RNG.standard_normal((n, c)).astype(np.float32), dtype)
    ws = u.workspace(lib.mcn_bn_workspace_bytes(m_, c))
    st = u.stream()
    # three kernels
    dxse = torch.full(shape, float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    dm_ref = torch.full((n, c), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    _ffi.check(lib.mcn_channel_scale_bwd(dy.data_ptr(), xse.data_ptr(), mk.data_ptr(), dxse.data_ptr(), dm_ref.data_ptr(), n, hw, c, u.MDT[dtype], st))
    _ffi.check(lib.mcn_global_avgpool_bwd_acc(dgap.data_ptr(), dxse.data_ptr(), n, hw, c, u.MDT[dtype], st))
    dx_ref = torch.full(shape, float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    dg_ref, db_ref = torch.zeros(c, device=u.DEV), torch.zeros(c, device=u.DEV)
    _ffi.check(lib.mcn_bn_bwd(dxse.data_ptr(), xd.data_ptr(), 0, 0, gd.data_ptr(), bd.data_ptr(), sm.data_ptr(), si.data_ptr(), dx_ref.data_ptr(), 0, dg_ref.data_ptr(), db_ref.data_ptr(),
                              0.5, m_, c, 2, u.MDT[dtype], ws.data_ptr(), ws.numel() * 4, st))
    # composed
    dm = torch.full((n, c), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    _ffi.check(lib.mcn_channel_scale_bwd_dm(dy.data_ptr(), xse.data_ptr(), dm.data_ptr(), n, hw, c, u.MDT[dtype], st))
    dx = torch.full(shape, float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    dg, db = torch.zeros(c, device=u.DEV), torch.zeros(c, device=u.DEV)
    _ffi.check(lib.mcn_bn_bwd_se(dy.data_ptr(), mk.data_ptr(), dgap.data_ptr(), xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), sm.data_ptr(), si.data_ptr(), dx.data_ptr(), dg.data_ptr(),
                                 db.data_ptr(), 0.5, n, hw, c, u.MDT[dtype], ws.data_ptr(), ws.numel() * 4, st))
    for a, b in ((dx, dx_ref), (dg, dg_ref), (db, db_ref), (dm, dm_ref)):
        np.testing.assert_array_equal(u.host(a), u.host(b))
    assert np.abs(u.host(dx)).max() > 0
    # round 4: the pair that skips the BN backward's reduction pass — dm from the BN's INPUT (x_se rebuilt on the fly), the BN-backward sums from
    # per-image sums of the unrounded gradient: dm identical where the rebuilt x_se is (it is the forward's expression), dgamma / dbeta to fp32
    # summation accuracy, dx to one rounding of the storage type around the two coefficients
    dm2 = torch.full((n, c), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    sums = torch.full((int(lib.mcn_se_bwd_sums_floats(n, hw, c, u.MDT[dtype])),), float('nan'), dtype=torch.float32, device=u.DEV)
    _ffi.check(lib.mcn_channel_scale_bwd_dm_bnsums(dy.data_ptr(), xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), sm.data_ptr(), si.data_ptr(), dm2.data_ptr(), sums.data_ptr(), n, hw, c,
                                                  u.MDT[dtype], st))
    dx2 = torch.full(shape, float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    dg2, db2 = torch.zeros(c, device=u.DEV), torch.zeros(c, device=u.DEV)
    _ffi.check(lib.mcn_bn_bwd_se_sums(dy.data_ptr(), mk.data_ptr(), dgap.data_ptr(), xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), sm.data_ptr(), si.data_ptr(), sums.data_ptr(), dx2.data_ptr(),
                                      dg2.data_ptr(), db2.data_ptr(), 0.5, n, hw, c, u.MDT[dtype], ws.data_ptr(), ws.numel() * 4, st))
    assert np.isfinite(u.host(sums)).all()
    dmh, dmr = u.host(dm2).astype(np.float64), u.host(dm_ref).astype(np.float64)
    ulp = {'float32': 2.0 ** -22, 'bfloat16': 2.0 ** -7, 'float16': 2.0 ** -10}[dtype]
    # (the pass slices an image's pixels over workgroups: fp32 sums in another order; for the 2-byte types the rounded result rarely moves)
    assert np.abs(dmh - dmr).max() <= ulp * np.abs(dmr).max() and (dtype == 'float32' or (dmh != dmr).mean() <= 0.02), (np.abs(dmh - dmr).max(), (dmh != dmr).mean())
    # dgamma / dbeta of the composition sum the ROUNDED g, the pair the unrounded one: |difference| <= rounding unit x sum |g s'| / sqrt(M) (random signs), with margin
    tol = {'float32': 2e-5, 'bfloat16': 4e-3, 'float16': 6e-4}[dtype]
    gq = np.abs(u.host(dxse).astype(np.float64))
    np.testing.assert_allclose(u.host(db2), u.host(db_ref), rtol=tol, atol=4 * tol * gq.sum((0, 1, 2)).max() / np.sqrt(m_))
    np.testing.assert_allclose(u.host(dg2), u.host(dg_ref), rtol=tol, atol=8 * tol * gq.sum((0, 1, 2)).max() / np.sqrt(m_))
    check(u.host(dx2), u.host(dx_ref), dtype, 'dx of the sums pair vs the composition', rel={'float32': 1e-5, 'bfloat16': 3e-3, 'float16': 4e-4}[dtype],
          mx={'float32': 1e-4, 'bfloat16': 2e-2, 'float16': 3e-3}[dtype])


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(3, 7, 7, 32), (4, 5, 9, 144), (2, 14, 14, 480), (6, 28, 28, 240)])
def test_squeeze_excite_fused_entry_points_against_the_oracle(shape, dtype):
    """The three squeeze-excite entry points of round 4 against oracle.ops DIRECTLY (not against other HIP kernels): BN(train) + swish + channel scale
    forward (mcn_bn_fwd_train_gap(y = NULL) + mcn_bn_act_scale_fwd), and the backward pair mcn_channel_scale_bwd_dm_bnsums + mcn_bn_bwd_se_sums,
    composed in the oracle from bn_fwd_train / swish / channel_scale / global_avgpool / bn_bwd (models/efficientnet.py:150-163, 179-197).
    Tolerances: storage rounding of the composed tensors (x_se, dy * m) sits inside the 2-byte bars of check(); the BN-backward sums of the
    fused pair are those of the UNROUNDED gradient (dx then no longer sums to exactly zero per channel in bf16 — the residual is
    bounded below)."""
    import torch
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    n, h, w_, c = shape
    hw, m_ = h * w_, n * h * w_
    md = u.MDT[dtype]
    x = (1.2 * RNG.standard_normal(shape) + 0.1).astype(np.float32)
    gamma = (0.5 + RNG.random(c)).astype(np.float32)
    beta = (0.3 * RNG.standard_normal(c)).astype(np.float32)
    mk = RNG.random((n, c)).astype(np.float32)
    dy = RNG.standard_normal(shape).astype(np.float32)
    dgap = RNG.standard_normal((n, c)).astype(np.float32)
    xq, mq, dyq, dgq = (q(a, dtype).astype(np.float64) for a in (x, mk, dy, dgap))
    # ---- oracle: forward ----
    bn_out, _, _, mean, invstd = O.bn_fwd_train(xq, gamma.astype(np.float64), beta.astype(np.float64), 1e-3)
    xse = q(O.swish_fwd(bn_out), dtype).astype(np.float64)                       # the tensor the unfused graph stores
    gap_ref = O.global_avgpool_fwd(xse)
    y_ref = O.channel_scale_fwd(xse, mq)
    # ---- device: forward ----
    xd, gd, bd, mkd = u.dev(x, dtype), u.dev(gamma), u.dev(beta), u.dev(mk, dtype)
    sm, si, bm, bv = [torch.zeros(c, dtype=torch.float32, device=u.DEV) for _ in range(4)]
    gap = torch.full((n, c), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    ws = u.workspace(lib.mcn_bn_workspace_bytes(m_, c))
    st = u.stream()
    _ffi.check(lib.mcn_bn_fwd_train_gap(xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), 0, gap.data_ptr(), sm.data_ptr(), si.data_ptr(), bm.data_ptr(), bv.data_ptr(), 0, 0, 0.99,
                                        n, hw, c, 1e-3, _ffi.ACT_SWISH, md, ws.data_ptr(), ws.numel() * 4, st))
    check(u.host(sm), mean, 'float32', 'save_mean', rel=1e-5, mx=1e-4)
    check(u.host(si), invstd, 'float32', 'save_invstd', rel=1e-5)
    check(u.host(gap), gap_ref, dtype, 'pooled means of the unstored BN + swish output')
    y = torch.full(shape, float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    _ffi.check(lib.mcn_bn_act_scale_fwd(xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), sm.data_ptr(), si.data_ptr(), mkd.data_ptr(), y.data_ptr(), n, hw, c, _ffi.ACT_SWISH, md, st))
    check(u.host(y), y_ref, dtype, 'bn + swish + channel scale')
    # an activation the fused passes do not implement is refused, not run as the identity
    assert lib.mcn_bn_fwd_train_gap(xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), 0, gap.data_ptr(), sm.data_ptr(), si.data_ptr(), bm.data_ptr(), bv.data_ptr(), 0, 0, 0.99,
                                    n, hw, c, 1e-3, _ffi.ACT_RELU6, md, ws.data_ptr(), ws.numel() * 4, st) == _ffi.E_UNSUPPORTED
    # ---- oracle: backward ----
    dxse, dm_ref = O.channel_scale_bwd(dyq, xse, mq)
    g = q(dxse, dtype).astype(np.float64) + O.global_avgpool_bwd(dgq, shape)     # the unfused graph stores dy * m, then accumulates the pooled branch
    g_store = q(g, dtype).astype(np.float64)
    dsw = O.swish_bwd(g_store, bn_out)
    dx_ref, dg_ref, db_ref = O.bn_bwd(dsw, xq, gamma.astype(np.float64), mean, invstd)
    # ---- device: backward ----
    dyd, dgd = u.dev(dy, dtype), u.dev(dgap, dtype)
    dm = torch.full((n, c), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    sums = torch.full((int(lib.mcn_se_bwd_sums_floats(n, hw, c, md)),), float('nan'), dtype=torch.float32, device=u.DEV)
    _ffi.check(lib.mcn_channel_scale_bwd_dm_bnsums(dyd.data_ptr(), xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), sm.data_ptr(), si.data_ptr(), dm.data_ptr(), sums.data_ptr(), n, hw, c, md, st))
    dx = torch.full(shape, float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    dg, db = torch.zeros(c, device=u.DEV), torch.zeros(c, device=u.DEV)
    _ffi.check(lib.mcn_bn_bwd_se_sums(dyd.data_ptr(), mkd.data_ptr(), dgd.data_ptr(), xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), sm.data_ptr(), si.data_ptr(), sums.data_ptr(), dx.data_ptr(),
                                      dg.data_ptr(), db.data_ptr(), 1.0, n, hw, c, md, ws.data_ptr(), ws.numel() * 4, st))
    check(u.host(dm), dm_ref, dtype, 'dm = sum dy * x_se (x_se rebuilt from the BN input)')
    # parameter gradients: the oracle sums the gradient g as the unfused graph STORES it (two roundings to the storage type), the fused pair sums the
    # unrounded g.  Both are sums of M random-sign terms, so the difference relative to the sum is a few storage ulps (bf16 2^-8, fp16 2^-11), measured
    # 6-8e-3 / 1.6e-3 at these shapes
    ptol = {'float32': 1e-4, 'bfloat16': 2e-2, 'float16': 4e-3}[dtype]
    check(u.host(dg), dg_ref, 'float32', 'dgamma', rel=ptol, mx=max(ptol * 4, 1e-3))
    check(u.host(db), db_ref, 'float32', 'dbeta', rel=ptol, mx=max(ptol * 4, 1e-3))
    check(u.host(dx), dx_ref, dtype, 'dx')
    # the residual the unrounded sums leave: per channel |sum dx| stays below one storage rounding of every element (sqrt(M) growth)
    dxh = u.host(dx).astype(np.float64).reshape(-1, c)
    xhat = (xq.reshape(-1, c) - mean) * invstd
    ulp = {'float32': 2.0 ** -23, 'bfloat16': 2.0 ** -8, 'float16': 2.0 ** -11}[dtype]
    bound = 8.0 * ulp * np.sqrt(m_) * np.abs(dxh).max() + 1e-6 * np.abs(dxh).sum(0).max()
    assert np.abs(dxh.sum(0)).max() <= bound, (np.abs(dxh.sum(0)).max(), bound)
    # (sum dx * xhat is not zero by construction: eps sits inside the square root, so sum xhat^2 = M * var / (var + eps); against the oracle's dx instead)
    ref_x = (np.asarray(dx_ref, np.float64).reshape(-1, c) * xhat).sum(0)
    assert np.abs((dxh * xhat).sum(0) - ref_x).max() <= 3.0 * bound + 1e-5 * np.abs(dxh * xhat).sum(0).max()


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
@pytest.mark.parametrize('case', __import__('dw_mult_case').CASES)
def test_depthwise_channel_multiplier_and_bias_through_the_layer_api(case, dtype):
    """conv_layer(depthwise=True) with out_channels = cin * mult (mult != 1) and / or biased=True — forward, data gradient (through the channel repeat),
    filter gradient [kh, kw, cin, mult] and bias gradient against oracle.ops.depthwise_conv2d_* / bias_add_* (tests/dw_mult_case.py)."""
    import myconvnet_amd as M
    import dw_mult_case
    dw_mult_case.run_case(M, case, dtype, check, q)
