"""GPU parity, segmentation row: bilinear resize (align_corners) fwd / bwd, channel concat, SegNet label
encoding, per-pixel softmax cross-entropy with ignored pixels — HIP kernels through the C-ABI vs the NumPy oracle."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ops as O  # noqa: E402
from test_gpu_ops import check, q, DTYPES  # noqa: E402

RNG = np.random.default_rng(41)


def _u():
    import abi_util
    return abi_util


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', [(2, 5, 7, 8, 9, 13), (2, 9, 9, 16, 33, 33), (1, 33, 33, 24, 129, 129), (2, 3, 5, 8, 3, 5), (2, 1, 1, 8, 4, 6),
                                  (1, 8, 8, 8, 4, 3), (2, 17, 17, 256, 33, 33), (2, 9, 9, 19, 33, 33), (1, 5, 6, 3, 10, 12)])   # last two: class maps, C not a chunk multiple
def test_resize_bilinear_align_corners(case, dtype):
    from myconvnet_amd import _ffi
    u = _u()
    n, h, w, c, oh, ow = case
    x = RNG.standard_normal((n, h, w, c)).astype(np.float32)
    dy = RNG.standard_normal((n, oh, ow, c)).astype(np.float32)
    xd, dyd = u.dev(x, dtype), u.dev(dy, dtype)
    y = torch.full((n, oh, ow, c), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    dx = torch.full((n, h, w, c), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    _ffi.check(_ffi.lib.mcn_resize_bilinear_fwd(xd.data_ptr(), y.data_ptr(), n, h, w, c, oh, ow, 1, u.MDT[dtype], u.stream()))
    _ffi.check(_ffi.lib.mcn_resize_bilinear_bwd(dyd.data_ptr(), dx.data_ptr(), n, h, w, c, oh, ow, 1, u.MDT[dtype], u.stream()))
    check(u.host(y), O.resize_bilinear_fwd(q(x, dtype), (oh, ow)), dtype, 'resize fwd')
    check(u.host(dx), O.resize_bilinear_bwd(q(dy, dtype), x.shape), dtype, 'resize bwd')
    if oh == h and ow == w:
        np.testing.assert_array_equal(u.host(y), u.host(xd))                  # identity size: exact copy


@pytest.mark.parametrize('dtype', DTYPES)
def test_concat_and_split(dtype):
    from myconvnet_amd import _ffi
    u = _u()
    parts = [RNG.standard_normal((2, 5, 5, c)).astype(np.float32) for c in (16, 8, 24, 16)]
    tot = sum(p.shape[-1] for p in parts)
    m = 2 * 5 * 5
    dst = torch.full((2, 5, 5, tot), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    off = 0
    keep = []
    for p in parts:
        pd = u.dev(p, dtype)
        keep.append(pd)
        _ffi.check(_ffi.lib.mcn_copy_channels(pd.data_ptr(), p.shape[-1], 0, dst.data_ptr(), tot, off, m, p.shape[-1], u.MDT[dtype], u.stream()))
        off += p.shape[-1]
    ref = O.concat_fwd([q(p, dtype) for p in parts])
    np.testing.assert_array_equal(u.host(dst), ref)
    off = 0
    for p, r in zip(parts, O.concat_bwd(ref, [p.shape[-1] for p in parts])):
        out = torch.full(p.shape, float('nan'), dtype=u.TDT[dtype], device=u.DEV)
        _ffi.check(_ffi.lib.mcn_copy_channels(dst.data_ptr(), tot, off, out.data_ptr(), p.shape[-1], 0, m, p.shape[-1], u.MDT[dtype], u.stream()))
        np.testing.assert_array_equal(u.host(out), r)
        off += p.shape[-1]


def test_seg_labels_and_per_pixel_loss():
    from myconvnet_amd import _ffi
    u = _u()
    n, h, w, c = 2, 37, 41, 19
    labels = RNG.integers(0, c + 2, (n, h, w)).astype(np.float32)           # 0 = ignore, c+1 = out of range
    labels[0, 0, :5] = np.nan
    labels[1, 3, 3] = 4.4999                                                 # round(3.4999) = 3
    labels[1, 3, 4] = 3.5                                                    # round(2.5) = 2 (half to even)
    ld = u.dev(labels)
    oh = torch.full((n, h, w, c), float('nan'), dtype=torch.float32, device=u.DEV)
    _ffi.check(_ffi.lib.mcn_one_hot_seg(ld.data_ptr(), oh.data_ptr(), n * h * w, c, u.stream()))
    ref_oh = O.seg_one_hot_labels(labels, c)
    np.testing.assert_array_equal(u.host(oh), ref_oh)
    assert ref_oh[1, 3, 3].argmax() == 3 and ref_oh[1, 3, 4].argmax() == 2 and ref_oh[0, 0, 0].sum() == 0
    logits = (2 * RNG.standard_normal((n, h, w, c))).astype(np.float32)
    cw = (0.5 + RNG.random(c)).astype(np.float32)
    B = n * h * w
    zd, cwd = u.dev(logits), u.dev(cw)
    pred = torch.zeros((B, c), dtype=torch.float32, device=u.DEV)
    ce, coef = torch.zeros(B, dtype=torch.float32, device=u.DEV), torch.zeros(B, dtype=torch.float32, device=u.DEV)
    dl = torch.zeros((B, c), dtype=torch.float32, device=u.DEV)
    loss = torch.zeros(4, dtype=torch.float32, device=u.DEV)
    ws = u.workspace(4096)
    _ffi.check(_ffi.lib.mcn_softmax_xent_rows_fwd_bwd(zd.data_ptr(), oh.data_ptr(), cwd.data_ptr(), pred.data_ptr(), ce.data_ptr(), coef.data_ptr(),
                                                      dl.data_ptr(), loss.data_ptr(), B, c, 0.0, 1.0, ws.data_ptr(), ws.numel() * 4, u.stream()))
    rp, rl, rce, rdl = O.softmax_xent_fwd_bwd(logits.reshape(B, c).astype(np.float64), ref_oh.reshape(B, c).astype(np.float64), cw.astype(np.float64))
    check(u.host(pred), rp, 'float32', 'pred', rel=1e-5)
    check(u.host(dl), rdl, 'float32', 'dlogits', rel=1e-5)
    assert abs(float(loss[0].item()) - rl) <= 1e-5 * abs(rl)
    # the block-per-row entry point computes the same thing
    pred2, dl2, loss2 = torch.zeros_like(pred), torch.zeros_like(dl), torch.zeros(4, dtype=torch.float32, device=u.DEV)
    _ffi.check(_ffi.lib.mcn_softmax_xent_fwd_bwd(zd.data_ptr(), oh.data_ptr(), cwd.data_ptr(), pred2.data_ptr(), ce.data_ptr(), coef.data_ptr(),
                                                 dl2.data_ptr(), loss2.data_ptr(), B, c, 0.0, 1.0, u.stream()))
    check(u.host(pred2), u.host(pred), 'float32', 'pred (two entry points)', rel=1e-6)
    check(u.host(dl2), u.host(dl), 'float32', 'dlogits (two entry points)', rel=1e-6)


def test_full_size_properties_seg_and_mbconv():
    """Size-independent properties at the full layer sizes of BASELINE configs[3] / configs[4] (reduced batch):
    adjointness <dy, op(x)> == <op^T(dy), x> of the depthwise conv (and == <wgrad, w>) and of the bilinear resize,
    and the row properties of the per-pixel loss on a 513x513 class map."""
    from myconvnet_amd import _ffi
    u = _u()
    rng = np.random.default_rng(77)
    # EfficientNet-B0 block_2 depthwise 3x3/2 on 112x112x96 and block_6 5x5/1 on 14x14x672
    for (n, h, c, k, s) in [(4, 112, 96, 3, 2), (8, 14, 672, 5, 1)]:
        x = rng.standard_normal((n, h, h, c)).astype(np.float32)
        w = (rng.standard_normal((k, k, c, 1)) / k).astype(np.float32)
        y = u.dwconv_fwd(x, w, s, 'SAME')
        dy = rng.standard_normal(y.shape).astype(np.float32)
        lhs = float((dy.astype(np.float64) * y).sum())
        dx = u.dwconv_dgrad(dy, w, x.shape, s, 'SAME')
        dw = u.dwconv_wgrad(x, dy, k, s, 'SAME')
        assert abs(lhs - float((dx.astype(np.float64) * x).sum())) <= 1e-4 * abs(lhs) + 1e-2
        assert abs(lhs - float((dw.astype(np.float64) * w).sum())) <= 1e-4 * abs(lhs) + 1e-2
    # DeepLab: ASPP output 33x33x256 -> 129x129, class map 129x129x19 -> 513x513
    for (n, h, c, oh) in [(2, 33, 256, 129), (2, 129, 19, 513)]:
        x = rng.standard_normal((n, h, h, c)).astype(np.float32)
        dy = rng.standard_normal((n, oh, oh, c)).astype(np.float32)
        xd, dyd = u.dev(x), u.dev(dy)
        y = torch.zeros((n, oh, oh, c), dtype=torch.float32, device=u.DEV)
        dx = torch.zeros((n, h, h, c), dtype=torch.float32, device=u.DEV)
        _ffi.check(_ffi.lib.mcn_resize_bilinear_fwd(xd.data_ptr(), y.data_ptr(), n, h, h, c, oh, oh, 1, _ffi.F32, u.stream()))
        _ffi.check(_ffi.lib.mcn_resize_bilinear_bwd(dyd.data_ptr(), dx.data_ptr(), n, h, h, c, oh, oh, 1, _ffi.F32, u.stream()))
        lhs = float((dy.astype(np.float64) * u.host(y)).sum())
        rhs = float((u.host(dx).astype(np.float64) * x).sum())
        assert abs(lhs - rhs) <= 1e-5 * abs(lhs) + 1e-2
        assert abs(u.host(dx).sum(dtype=np.float64) - dy.sum(dtype=np.float64)) <= 1e-4 * np.abs(dy).sum()   # interpolation weights sum to 1
    # per-pixel loss at 513x513, 19 classes
    n, hw, c = 2, 513, 19
    labels = rng.integers(0, c + 1, (n, hw, hw)).astype(np.float32)
    logits = rng.standard_normal((n, hw, hw, c)).astype(np.float32)
    B = n * hw * hw
    ld, zd = u.dev(labels), u.dev(logits)
    oh_ = torch.zeros((B, c), dtype=torch.float32, device=u.DEV)
    pred, dl = torch.zeros((B, c), dtype=torch.float32, device=u.DEV), torch.zeros((B, c), dtype=torch.float32, device=u.DEV)
    ce, coef = torch.zeros(B, dtype=torch.float32, device=u.DEV), torch.zeros(B, dtype=torch.float32, device=u.DEV)
    loss = torch.zeros(4, dtype=torch.float32, device=u.DEV)
    ws = u.workspace(4096)
    _ffi.check(_ffi.lib.mcn_one_hot_seg(ld.data_ptr(), oh_.data_ptr(), B, c, u.stream()))
    _ffi.check(_ffi.lib.mcn_softmax_xent_rows_fwd_bwd(zd.data_ptr(), oh_.data_ptr(), 0, pred.data_ptr(), ce.data_ptr(), coef.data_ptr(), dl.data_ptr(),
                                                      loss.data_ptr(), B, c, 0.0, 1.0, ws.data_ptr(), ws.numel() * 4, u.stream()))
    p, g, cf, cel = u.host(pred), u.host(dl), u.host(coef), u.host(ce)
    valid = labels.reshape(-1) > 0
    np.testing.assert_array_equal(cf, valid.astype(np.float32))
    np.testing.assert_allclose(p.sum(-1), 1.0, rtol=1e-5)
    assert np.abs(g.sum(-1)).max() <= 1e-9                                      # each row of dlogits sums to 0
    assert np.abs(g[~valid]).max() == 0.0                                       # ignored pixels carry no gradient
    assert abs(float(loss[0].item()) - float((cel.astype(np.float64) * cf).sum() / B)) <= 1e-5
