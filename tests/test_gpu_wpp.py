"""conv_gemm_nt_wpp (round 4): the window ping-pong kernel of the 2-byte types for 3x3 / stride-1 convolutions (tile hint 6, and the default for
layers with four or more 64-channel chunks per tap) through the C-ABI: forward (plain, bias, BN-statistics epilogue) and dgrad (plain, masked
residual add, BN-backward sums, both) against the NumPy oracle and against the two-buffer tile on the same inputs.  The kernel sums K in
(chunk, tap) order, the two-buffer kernels in (tap, chunk) order: equal up to fp32 summation order, i.e. rare one-ulp flips of the stored type."""
import ctypes
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import ops as O  # noqa: E402
from test_gpu_ops import check, q  # noqa: E402

RNG = np.random.default_rng(4646)
FLIP = {'bfloat16': dict(rel=1.5e-3, mx=1e-2), 'float16': dict(rel=2e-4, mx=2e-3)}


def _u():
    import abi_util
    return abi_util


def _name(lib, op, g, md):
    buf = ctypes.create_string_buffer(128)
    lib.mcn_conv2d_kernel_name(op, ctypes.byref(g), md, buf, 128)
    return buf.value.decode()


# n, h, w, cin, cout, dilation, taken without a hint
CASES = [(3, 14, 14, 256, 256, 1, True),        # the ResNet 14x14 layer: four chunks, two N tiles, ragged last M tile
         (2, 7, 7, 512, 136, 1, True),          # one ragged M tile (98 rows), ragged second N tile (8 columns), eight chunks
         (5, 28, 28, 128, 64, 1, False),        # BN = 64 instantiation, two chunks
         (1, 56, 56, 64, 128, 1, False),        # widest window the kernel takes at dilation 1 (span 114 rows), one chunk
         (2, 20, 20, 192, 128, 2, False)]       # dilation 2 (span 84), odd chunk count


@pytest.mark.parametrize('dtype', ['bfloat16', 'float16'])
@pytest.mark.parametrize('case', CASES)
def test_window_pingpong_forward(case, dtype):
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    n, h, w_, cin, cout, dil, by_default = case
    md = u.MDT[dtype]
    x = RNG.standard_normal((n, h, w_, cin)).astype(np.float32)
    w = (RNG.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    bias = RNG.standard_normal(cout).astype(np.float32)
    g6, g2 = u.geom(x.shape, w.shape, 1, 'SAME', dil), u.geom(x.shape, w.shape, 1, 'SAME', dil)
    g6.tile, g2.tile = 6, (1 if cout > 64 else 2)
    assert _name(lib, _ffi.CONV_FWD, g6, md).startswith('conv_gemm_nt_wpp<')
    assert _name(lib, _ffi.CONV_FWD, g6, u.MDT['float32']).find('wpp') < 0                    # 2-byte types only
    g0 = u.geom(x.shape, w.shape, 1, 'SAME', dil)
    assert _name(lib, _ffi.CONV_FWD, g0, md).startswith('conv_gemm_nt_wpp<') == by_default
    xd, wd, bd = u.dev(x, dtype), u.dev(w), u.dev(bias)
    ws = u.workspace(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_FWD, ctypes.byref(g6), md))

    def fwd(g, b=0):
        y = torch.full((n, h, w_, cout), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
        _ffi.check(lib.mcn_conv2d_fwd(xd.data_ptr(), wd.data_ptr(), 0, b, y.data_ptr(), ctypes.byref(g), md, _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, u.stream()))
        return u.host(y)
    y6 = fwd(g6)
    ref = O.conv2d_fwd(q(x, dtype), q(w, dtype), 1, 'SAME', dil)
    check(y6, ref, dtype, 'wpp fwd vs oracle')
    check(y6, fwd(g2), dtype, 'wpp fwd vs two-buffer tile', **FLIP[dtype])
    np.testing.assert_array_equal(fwd(g6), y6)                                               # deterministic
    check(fwd(g6, bd.data_ptr()), ref + bias, dtype, 'wpp fwd + bias vs oracle')
    # BN-statistics epilogue: the stored output is the plain launch's; the partial rows sum to the column sums of what was stored
    rpp = ctypes.c_int32(0)
    rows = lib.mcn_conv2d_bnstats_rows(ctypes.byref(g6), md, ctypes.byref(rpp))
    m = n * h * w_
    assert rows > 0 and rpp.value == 64 and rows * rpp.value >= m
    part = torch.full((rows, 3, cout), float('nan'), dtype=torch.float32, device=u.DEV)
    y = torch.full((n, h, w_, cout), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    _ffi.check(lib.mcn_conv2d_fwd_bnstats(xd.data_ptr(), wd.data_ptr(), 0, 0, y.data_ptr(), part.data_ptr(), ctypes.byref(g6), md, _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, u.stream()))
    np.testing.assert_array_equal(u.host(y), y6)
    gamma, beta = (0.5 + RNG.random(cout)).astype(np.float32), (0.3 * RNG.standard_normal(cout)).astype(np.float32)
    want = u.bn_fwd_train(y6, gamma, beta, 1e-3, dtype, act=1)
    out = torch.full((n, h, w_, cout), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    gd, bt = u.dev(gamma), u.dev(beta)
    sm, si, bm, bv = [torch.zeros(cout, dtype=torch.float32, device=u.DEV) for _ in range(4)]
    rm, rv = torch.zeros(cout, dtype=torch.float32, device=u.DEV), torch.ones(cout, dtype=torch.float32, device=u.DEV)
    bws = u.workspace(lib.mcn_bn_workspace_bytes(m, cout))
    _ffi.check(lib.mcn_bn_fwd_train_fused(y.data_ptr(), part.data_ptr(), rows, rpp.value, gd.data_ptr(), bt.data_ptr(), 0, out.data_ptr(), 0, sm.data_ptr(), si.data_ptr(),
                                          bm.data_ptr(), bv.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.99, m, cout, 1e-3, 1, md, bws.data_ptr(), bws.numel() * 4, u.stream()))
    check(u.host(sm), want['save_mean'], 'float32', 'save_mean from the wpp epilogue', rel=1e-5, mx=1e-4)
    check(u.host(si), want['save_invstd'], 'float32', 'save_invstd from the wpp epilogue', rel=1e-5)
    check(u.host(out), want['y'], dtype, 'fused BN output', rel=4e-3)


@pytest.mark.parametrize('dtype', ['bfloat16', 'float16'])
@pytest.mark.parametrize('case', CASES)
def test_window_pingpong_dgrad(case, dtype):
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    n, h, w_, cout, cin, dil, by_default = case               # (the dgrad GEMM's K side is Cout: the case's first channel count)
    md = u.MDT[dtype]
    w = (RNG.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cout)).astype(np.float32)
    dy = RNG.standard_normal((n, h, w_, cout)).astype(np.float32)
    xs = (n, h, w_, cin)
    g6, g2 = u.geom(xs, w.shape, 1, 'SAME', dil), u.geom(xs, w.shape, 1, 'SAME', dil)
    g6.tile, g2.tile = 6, (1 if cin > 64 else 2)
    assert _name(lib, _ffi.CONV_DGRAD, g6, md).startswith('conv_gemm_nt_wpp<')
    assert _name(lib, _ffi.CONV_DGRAD, u.geom(xs, w.shape, 1, 'SAME', dil), md).startswith('conv_gemm_nt_wpp<') == by_default
    lst = ctypes.create_string_buffer(512)
    assert lib.mcn_conv2d_launch_list(_ffi.CONV_DGRAD, ctypes.byref(g6), md, lst, 512) == 1 and lst.value.decode().startswith('conv_gemm_nt_wpp<') and lst.value.decode().strip().endswith(':9')
    dyd, wd = u.dev(dy, dtype), u.dev(w)
    ws = u.workspace(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_DGRAD, ctypes.byref(g6), md))

    def dgrad(g):
        dx = torch.full(xs, float('nan'), dtype=u.TDT[dtype], device=u.DEV)
        _ffi.check(lib.mcn_conv2d_dgrad(dyd.data_ptr(), wd.data_ptr(), 0, dx.data_ptr(), ctypes.byref(g), 0, md, _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, u.stream()))
        return u.host(dx)
    dx6 = dgrad(g6)
    check(dx6, O.conv2d_dgrad(q(dy, dtype), q(w, dtype), xs, 1, 'SAME', dil), dtype, 'wpp dgrad vs oracle')
    check(dx6, dgrad(g2), dtype, 'wpp dgrad vs two-buffer tile', **FLIP[dtype])
    # accumulate epilogue (dx += dgrad)
    base = RNG.standard_normal(xs).astype(np.float32)
    acc = u.dev(base, dtype)
    _ffi.check(lib.mcn_conv2d_dgrad(dyd.data_ptr(), wd.data_ptr(), 0, acc.data_ptr(), ctypes.byref(g6), 1, md, _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, u.stream()))
    acc2 = u.dev(base, dtype)
    _ffi.check(lib.mcn_conv2d_dgrad(dyd.data_ptr(), wd.data_ptr(), 0, acc2.data_ptr(), ctypes.byref(g2), 1, md, _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, u.stream()))
    check(u.host(acc), u.host(acc2), dtype, 'wpp dgrad accumulate vs two-buffer tile', **FLIP[dtype])
    # masked residual add, BN-backward sums, and both in one launch
    assert lib.mcn_conv2d_dgrad_addmasked_ok(ctypes.byref(g6), md) == 1
    src = RNG.standard_normal(xs).astype(np.float32)
    mb = lib.mcn_bn_relu_mask_bytes(n * h * w_, cin, md)
    amask = torch.from_numpy(RNG.integers(0, 256, mb, dtype=np.uint8)).to(u.DEV)
    rmask_h = RNG.integers(0, 256, mb, dtype=np.uint8)
    rmask = torch.from_numpy(rmask_h).to(u.DEV)
    xbn = (1.2 * RNG.standard_normal(xs) + 0.2).astype(np.float32)
    sd, xd = u.dev(src, dtype), u.dev(xbn, dtype)

    def addmasked(g):
        dx = torch.full(xs, float('nan'), dtype=u.TDT[dtype], device=u.DEV)
        _ffi.check(lib.mcn_conv2d_dgrad_addmasked(dyd.data_ptr(), wd.data_ptr(), 0, dx.data_ptr(), sd.data_ptr(), amask.data_ptr(), ctypes.byref(g), md, _ffi.NHWC,
                                                  ws.data_ptr(), ws.numel() * 4, u.stream()))
        return u.host(dx)
    am6 = addmasked(g6)
    check(am6, addmasked(g2), dtype, 'wpp dgrad + masked add vs two-buffer tile', **FLIP[dtype])
    rows = lib.mcn_conv2d_dgrad_bnred_rows(ctypes.byref(g6), md)
    assert rows > 0
    bits = ((rmask_h.reshape(-1, 1) >> np.arange(8)) & 1).reshape(-1, cin).astype(np.float64)
    xq = q(xbn, dtype).reshape(-1, cin).astype(np.float64)
    for both in (False, True):
        dx = torch.full(xs, float('nan'), dtype=u.TDT[dtype], device=u.DEV)
        part = torch.full((rows, 2, cin), float('nan'), dtype=torch.float32, device=u.DEV)
        if both:
            _ffi.check(lib.mcn_conv2d_dgrad_addmasked_bnred(dyd.data_ptr(), wd.data_ptr(), 0, dx.data_ptr(), sd.data_ptr(), amask.data_ptr(), xd.data_ptr(), rmask.data_ptr(),
                                                            part.data_ptr(), ctypes.byref(g6), md, _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, u.stream()))
        else:
            _ffi.check(lib.mcn_conv2d_dgrad_bnred(dyd.data_ptr(), wd.data_ptr(), 0, dx.data_ptr(), xd.data_ptr(), rmask.data_ptr(), part.data_ptr(), ctypes.byref(g6), md, _ffi.NHWC,
                                                  ws.data_ptr(), ws.numel() * 4, u.stream()))
        got = u.host(dx)
        np.testing.assert_array_equal(got, am6 if both else dx6)              # the stored gradient is the plain launch's
        p = u.host(part).astype(np.float64)
        assert not np.isnan(p).any()
        dxm = got.reshape(-1, cin).astype(np.float64) * bits
        np.testing.assert_allclose(p[:, 0].sum(0), dxm.sum(0), rtol=2e-5, atol=2e-5 * np.abs(dxm).sum(0).max())
        np.testing.assert_allclose(p[:, 1].sum(0), (dxm * xq).sum(0), rtol=2e-5, atol=2e-5 * np.abs(dxm * xq).sum(0).max())


def test_window_pingpong_hint_on_other_geometries_runs_the_plain_tile():
    """tile hint 6 on a geometry the kernel does not take (stride 2, 1x1, ragged channel chunks, a window wider than its LDS buffer) runs the plain
    8-wave / heuristic tile: same results as without the hint, bit for bit."""
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    for (n, h, w_, cin, cout, k, s, dil) in [(2, 16, 16, 64, 128, 3, 2, 1), (2, 14, 14, 128, 256, 1, 1, 1), (3, 9, 11, 72, 136, 3, 1, 1), (1, 64, 64, 64, 128, 3, 1, 1), (1, 24, 24, 64, 128, 3, 1, 3)]:
        x = RNG.standard_normal((n, h, w_, cin)).astype(np.float32)
        w = (RNG.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
        g6, g4 = u.geom(x.shape, w.shape, s, 'SAME', dil), u.geom(x.shape, w.shape, s, 'SAME', dil)
        g6.tile, g4.tile = 6, 4
        md = u.MDT['bfloat16']
        assert 'wpp' not in _name(lib, _ffi.CONV_FWD, g6, md) and _name(lib, _ffi.CONV_FWD, g6, md) == _name(lib, _ffi.CONV_FWD, g4, md)
        xd, wd = u.dev(x, 'bfloat16'), u.dev(w)
        ws = u.workspace(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_FWD, ctypes.byref(g6), md))
        outs = []
        for g in (g6, g4):
            y = torch.full((n, -(-h // s), -(-w_ // s), cout), float('nan'), dtype=torch.bfloat16, device=u.DEV)
            _ffi.check(lib.mcn_conv2d_fwd(xd.data_ptr(), wd.data_ptr(), 0, 0, y.data_ptr(), ctypes.byref(g), md, _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, u.stream()))
            outs.append(u.host(y))
        np.testing.assert_array_equal(outs[0], outs[1])
