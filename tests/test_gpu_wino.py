"""GPU parity tests of the Winograd F(2x2, 3x3) path (fp32, 3x3 / stride 1 / pad 1; csrc/wino_kernels.h) through the C-ABI.

Every case is checked three ways: against the float64 oracle (the op tests' fp32 bar: rel-L2 <= 2e-5), against the direct kernels on the
same inputs (MCN_TILE_NOWINO in mcn_conv_geom.tile), and — for the fused epilogues — against what the direct kernels' epilogues feed the BN
entry points.  Shapes cover odd maps (tiles hanging over the edge), ragged tile / channel blocks, one and several 32-channel super-steps,
a channel stride of the input larger than Cin, bias, accumulation into an existing gradient, and the pre-packed operand (U).
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ops as O  # noqa: E402

NOWINO = 0x200
RNG = np.random.default_rng(2024)

# n, h, w, cin, cout
CASES = [
    (2, 8, 8, 32, 64),          # one super-step, one block
    (3, 7, 7, 64, 128),         # odd map: the last tile row / column hangs over the edge
    (2, 14, 10, 64, 96),        # ragged cout block (96 = 64 + 32)
    (5, 9, 11, 96, 40),         # ragged everything; cout < one wave tile
    (1, 1, 1, 32, 32),          # a single pixel
    (70, 14, 14, 128, 128),     # several tile blocks per cout block, more blocks than one XCD round
    (4, 28, 28, 32, 36),        # cout not a multiple of 8
]


def _u():
    import abi_util
    return abi_util


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _kernel(g, op):
    from myconvnet_amd import _ffi
    buf = ctypes.create_string_buffer(128)
    _ffi.lib.mcn_conv2d_kernel_name(op, ctypes.byref(g), _ffi.F32, buf, 128)
    return buf.value.decode()


@pytest.mark.parametrize('case', CASES)
def test_wino_forward_and_dgrad_match_oracle_and_direct_kernels(case):
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    n, h, w_, cin, cout = case
    x = RNG.standard_normal((n, h, w_, cin)).astype(np.float32)
    w = (RNG.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    bias = RNG.standard_normal(cout).astype(np.float32)
    dy = RNG.standard_normal((n, h, w_, cout)).astype(np.float32)
    g = u.geom(x.shape, w.shape, 1, 'SAME')
    gd = u.geom(x.shape, w.shape, 1, 'SAME')
    gd.tile = NOWINO
    assert 'wino' in _kernel(g, _ffi.CONV_FWD) and 'wino' not in _kernel(gd, _ffi.CONV_FWD)
    xd, wd, bd, dyd = u.dev(x), u.dev(w), u.dev(bias), u.dev(dy)

    def fwd(gm, b=0, packed=0):
        y = torch.full((n, h, w_, cout), float('nan'), dtype=torch.float32, device=u.DEV)
        ws = u.workspace(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_FWD, ctypes.byref(gm), u.MDT['float32']))
        _ffi.check(lib.mcn_conv2d_fwd(xd.data_ptr(), wd.data_ptr(), packed, b, y.data_ptr(), ctypes.byref(gm), u.MDT['float32'], _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, u.stream()))
        return u.host(y)

    def dgrad(gm, into=None, packed=0):
        dx = torch.full(x.shape, float('nan'), dtype=torch.float32, device=u.DEV) if into is None else u.dev(into)
        ws = u.workspace(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_DGRAD, ctypes.byref(gm), u.MDT['float32']))
        _ffi.check(lib.mcn_conv2d_dgrad(dyd.data_ptr(), wd.data_ptr(), packed, dx.data_ptr(), ctypes.byref(gm), 0 if into is None else 1, u.MDT['float32'], _ffi.NHWC,
                                        ws.data_ptr(), ws.numel() * 4, u.stream()))
        return u.host(dx)

    y_ref = O.conv2d_fwd(x, w, 1, 'SAME', 1)
    y, y_dir = fwd(g), fwd(gd)
    assert np.isfinite(y).all()
    assert rel_l2(y, y_ref) <= 2e-5 and rel_l2(y_dir, y_ref) <= 2e-5, (rel_l2(y, y_ref), rel_l2(y_dir, y_ref))
    assert np.abs(y - y_ref).max() <= 1e-3 * np.abs(y_ref).max()
    np.testing.assert_array_equal(fwd(g), y)                                             # deterministic
    yb = fwd(g, bd.data_ptr())
    assert rel_l2(yb, y_ref + bias) <= 2e-5
    pk, keep = u.prepack(w, g, _ffi.CONV_FWD, 'float32')                                 # the batched pack launch writes U
    np.testing.assert_array_equal(fwd(g, packed=pk.data_ptr()), y)

    dx_ref = O.conv2d_dgrad(dy, w, x.shape, 1, 'SAME', 1)
    dx, dx_dir = dgrad(g), dgrad(gd)
    assert rel_l2(dx, dx_ref) <= 2e-5 and rel_l2(dx_dir, dx_ref) <= 2e-5, (rel_l2(dx, dx_ref), rel_l2(dx_dir, dx_ref))
    prev = RNG.standard_normal(x.shape).astype(np.float32)
    assert rel_l2(dgrad(g, into=prev), dx_ref + prev) <= 2e-5                            # a second gradient contribution: dx +=
    pk2, keep2 = u.prepack(w, g, _ffi.CONV_DGRAD, 'float32')
    np.testing.assert_array_equal(dgrad(g, packed=pk2.data_ptr()), dx)
    if 'wino' in _kernel(g, _ffi.CONV_DGRAD):                                            # (its input channels = cout must be a multiple of 32)
        assert lib.mcn_conv2d_dgrad_addmasked_ok(ctypes.byref(g), u.MDT['float32']) == 0


def test_wino_input_channel_stride():
    """x stored with a channel stride larger than Cin (a view into a concatenation): the extra channels are never read"""
    u = _u()
    n, h, w_, cin, cout, cs = 2, 10, 12, 32, 64, 48
    xs = RNG.standard_normal((n, h, w_, cs)).astype(np.float32)
    xs[..., cin:] = np.nan
    w = (RNG.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    y = u.conv_fwd(xs, w, 1, 'SAME', 1, 'float32', x_cs=cs)
    assert rel_l2(y, O.conv2d_fwd(xs[..., :cin], w, 1, 'SAME', 1)) <= 2e-5


@pytest.mark.parametrize('case', [(3, 7, 7, 64, 128), (9, 14, 14, 32, 96), (40, 28, 28, 64, 64)])
def test_wino_bn_statistics_epilogue(case):
    """mcn_conv2d_fwd_bnstats on a Winograd layer writes COUNTED rows (rows_per_partial == 0): the BN forward fed with them matches the BN
    forward that reads the output itself (odd maps: positions outside the image are not counted)."""
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    n, h, w_, cin, cout = case
    x = RNG.standard_normal((n, h, w_, cin)).astype(np.float32)
    w = (RNG.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    g = u.geom(x.shape, w.shape, 1, 'SAME')
    rpp = ctypes.c_int32(-7)
    rows = lib.mcn_conv2d_bnstats_rows(ctypes.byref(g), u.MDT['float32'], ctypes.byref(rpp))
    assert rows > 0 and rpp.value == 0
    xd, wd = u.dev(x), u.dev(w)
    y = torch.full((n, h, w_, cout), float('nan'), dtype=torch.float32, device=u.DEV)
    part = torch.full((rows, 4, cout), float('nan'), dtype=torch.float32, device=u.DEV)
    ws = u.workspace(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_FWD, ctypes.byref(g), u.MDT['float32']))
    _ffi.check(lib.mcn_conv2d_fwd_bnstats(xd.data_ptr(), wd.data_ptr(), 0, 0, y.data_ptr(), part.data_ptr(), ctypes.byref(g), u.MDT['float32'], _ffi.NHWC,
                                          ws.data_ptr(), ws.numel() * 4, u.stream()))
    yh, p = u.host(y), u.host(part).astype(np.float64)
    assert rel_l2(yh, O.conv2d_fwd(x, w, 1, 'SAME', 1)) <= 2e-5
    assert np.isfinite(p).all()
    m = n * h * w_
    np.testing.assert_array_equal(p[:, 3].sum(0), np.full(cout, m))                       # every pixel counted once per channel
    y64 = yh.astype(np.float64).reshape(-1, cout)
    tot = (p[:, 0] + p[:, 3] * p[:, 2]).sum(0)
    np.testing.assert_allclose(tot, y64.sum(0), rtol=1e-5, atol=1e-5 * np.abs(y64).sum(0).max())
    sq = (p[:, 1] + 2 * p[:, 2] * p[:, 0] + p[:, 3] * p[:, 2] ** 2).sum(0)
    np.testing.assert_allclose(sq, (y64 ** 2).sum(0), rtol=1e-5)
    # through the BN entry point
    gamma = (0.5 + RNG.random(cout)).astype(np.float32)
    beta = (0.3 * RNG.standard_normal(cout)).astype(np.float32)
    ref = u.bn_fwd_train(yh, gamma, beta, 1e-3, 'float32', act=1)
    out = torch.full(yh.shape, float('nan'), dtype=torch.float32, device=u.DEV)
    sm, si, bm, bv = (torch.zeros(cout, device=u.DEV) for _ in range(4))
    rm, rv = torch.zeros(cout, device=u.DEV), torch.ones(cout, device=u.DEV)
    bws = u.workspace(lib.mcn_bn_workspace_bytes(m, cout))
    gd_, bd_ = u.dev(gamma), u.dev(beta)
    _ffi.check(lib.mcn_bn_fwd_train_fused(y.data_ptr(), part.data_ptr(), rows, 0, gd_.data_ptr(), bd_.data_ptr(), 0, out.data_ptr(), 0, sm.data_ptr(), si.data_ptr(),
                                          bm.data_ptr(), bv.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.99, m, cout, 1e-3, 1, u.MDT['float32'], bws.data_ptr(), bws.numel() * 4, u.stream()))
    assert rel_l2(u.host(sm), ref['save_mean']) <= 1e-5 and rel_l2(u.host(si), ref['save_invstd']) <= 1e-5
    assert rel_l2(u.host(out), ref['y']) <= 1e-5


@pytest.mark.parametrize('case', [(3, 7, 7, 64, 64), (6, 14, 14, 96, 32), (40, 28, 28, 128, 128)])
def test_wino_dgrad_bn_backward_sums(case):
    """mcn_conv2d_dgrad_bnred on a Winograd layer: dx as mcn_conv2d_dgrad, the column sums of the partial rows are sum dy' and sum dy' x"""
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    n, h, w_, cin, cout = case
    xbn = (1.2 * RNG.standard_normal((n, h, w_, cin)) + 0.2).astype(np.float32)
    gamma = (0.5 + RNG.random(cin)).astype(np.float32)
    beta = (0.3 * RNG.standard_normal(cin)).astype(np.float32)
    fw = u.bn_fwd_train(xbn, gamma, beta, 1e-3, 'float32', act=1, want_mask=True)
    wt = (RNG.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    g = u.geom(xbn.shape, wt.shape, 1, 'SAME')
    dy = RNG.standard_normal((n, h, w_, cout)).astype(np.float32)
    rows = lib.mcn_conv2d_dgrad_bnred_rows(ctypes.byref(g), u.MDT['float32'])
    assert rows > 0
    dx_ref = O.conv2d_dgrad(dy, wt, xbn.shape, 1, 'SAME', 1)
    dyd, wd, xd = u.dev(dy), u.dev(wt), u.dev(xbn)
    mk = torch.as_tensor(fw['relu_mask']).to(u.DEV)
    dx = torch.full(xbn.shape, float('nan'), dtype=torch.float32, device=u.DEV)
    part = torch.full((rows, 2, cin), float('nan'), dtype=torch.float32, device=u.DEV)
    ws = u.workspace(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_DGRAD, ctypes.byref(g), u.MDT['float32']))
    _ffi.check(lib.mcn_conv2d_dgrad_bnred(dyd.data_ptr(), wd.data_ptr(), 0, dx.data_ptr(), xd.data_ptr(), mk.data_ptr(), part.data_ptr(), ctypes.byref(g), u.MDT['float32'], _ffi.NHWC,
                                          ws.data_ptr(), ws.numel() * 4, u.stream()))
    dxh = u.host(dx)
    assert rel_l2(dxh, dx_ref) <= 2e-5
    p = u.host(part).astype(np.float64)
    assert np.isfinite(p).all()
    bits = (fw['y'].reshape(-1, cin) > 0).astype(np.float64)
    dxm = dxh.reshape(-1, cin).astype(np.float64) * bits
    xq = xbn.reshape(-1, cin).astype(np.float64)
    np.testing.assert_allclose(p[:, 0].sum(0), dxm.sum(0), rtol=2e-5, atol=2e-5 * np.abs(dxm).sum(0).max())
    np.testing.assert_allclose(p[:, 1].sum(0), (dxm * xq).sum(0), rtol=2e-5, atol=2e-5 * np.abs(dxm * xq).sum(0).max())


@pytest.mark.parametrize('case', [(2, 8, 8, 64, 64), (3, 7, 7, 32, 96), (5, 14, 10, 72, 40), (1, 1, 1, 32, 32), (70, 14, 14, 128, 128), (16, 28, 28, 64, 36)])
def test_wino_wgrad_matches_oracle_and_direct_kernel(case):
    """Winograd F(3x3, 2x2) weight gradient (tiles = the GEMM's K dimension, split over workgroups, slabs reduced in a fixed order)"""
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    n, h, w_, cin, cout = case
    x = RNG.standard_normal((n, h, w_, cin)).astype(np.float32)
    dy = RNG.standard_normal((n, h, w_, cout)).astype(np.float32)
    g = u.geom(x.shape, (3, 3, cin, cout), 1, 'SAME')
    gd = u.geom(x.shape, (3, 3, cin, cout), 1, 'SAME')
    gd.tile = NOWINO
    assert 'wino' in _kernel(g, _ffi.CONV_WGRAD) and 'wino' not in _kernel(gd, _ffi.CONV_WGRAD)
    xd, dyd = u.dev(x), u.dev(dy)

    def wgrad(gm, scale=1.0, bias=False):
        dw = torch.full((3, 3, cin, cout), float('nan'), dtype=torch.float32, device=u.DEV)
        db = torch.full((cout,), float('nan'), dtype=torch.float32, device=u.DEV)
        ws = u.workspace(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_WGRAD, ctypes.byref(gm), u.MDT['float32']))
        _ffi.check(lib.mcn_conv2d_wgrad(xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), db.data_ptr() if bias else 0, ctypes.byref(gm), float(scale), u.MDT['float32'], _ffi.NHWC,
                                        ws.data_ptr(), ws.numel() * 4, u.stream()))
        return u.host(dw), u.host(db)

    ref = O.conv2d_wgrad(x, dy, (3, 3, cin, cout), 1, 'SAME', 1)
    dw, _ = wgrad(g)
    dwd, _ = wgrad(gd)
    assert np.isfinite(dw).all()
    assert rel_l2(dw, ref) <= 2e-5 and rel_l2(dwd, ref) <= 2e-5, (rel_l2(dw, ref), rel_l2(dwd, ref))
    np.testing.assert_array_equal(wgrad(g)[0], dw)                                       # fixed-order reduce: bit-reproducible
    dws, db = wgrad(g, scale=0.25, bias=True)
    assert rel_l2(dws, 0.25 * ref) <= 2e-5
    assert rel_l2(db, 0.25 * dy.reshape(-1, cout).astype(np.float64).sum(0)) <= 2e-5


def test_wino_k_sliced_tail_matches_unsplit_launch():
    """More blocks than CUs with a short last round (296 = 256 + 40 here): the tail blocks run K-sliced (parked accumulators + a reduce launch
    that runs the epilogue); MCN_TILE_NOSPLIT keeps every block whole.  Same results up to fp32 summation order, deterministic, also with the
    BN-statistics and BN-backward epilogues."""
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    n, h, w_, cin, cout = 24, 28, 28, 64, 256
    x = RNG.standard_normal((n, h, w_, cin)).astype(np.float32)
    w = (RNG.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    g = u.geom(x.shape, w.shape, 1, 'SAME')
    gn = u.geom(x.shape, w.shape, 1, 'SAME')
    gn.tile = 0x100                                                                      # MCN_TILE_NOSPLIT
    xd, wd = u.dev(x), u.dev(w)
    rows = lib.mcn_conv2d_bnstats_rows(ctypes.byref(g), u.MDT['float32'], None)

    def fwd(gm):
        y = torch.full((n, h, w_, cout), float('nan'), dtype=torch.float32, device=u.DEV)
        part = torch.full((rows, 4, cout), float('nan'), dtype=torch.float32, device=u.DEV)
        ws = u.workspace(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_FWD, ctypes.byref(gm), u.MDT['float32']))
        _ffi.check(lib.mcn_conv2d_fwd_bnstats(xd.data_ptr(), wd.data_ptr(), 0, 0, y.data_ptr(), part.data_ptr(), ctypes.byref(gm), u.MDT['float32'], _ffi.NHWC,
                                              ws.data_ptr(), ws.numel() * 4, u.stream()))
        return u.host(y), u.host(part)

    (y, p), (yn, pn) = fwd(g), fwd(gn)
    assert np.isfinite(y).all() and np.isfinite(p).all()
    assert rel_l2(y, yn) <= 1e-6 and rel_l2(y, O.conv2d_fwd(x, w, 1, 'SAME', 1)) <= 2e-5
    np.testing.assert_array_equal(p[:, 3], pn[:, 3])
    assert rel_l2(p[:, 0] + p[:, 3] * p[:, 2], pn[:, 0] + pn[:, 3] * pn[:, 2]) <= 1e-5
    y2, p2 = fwd(g)
    np.testing.assert_array_equal(y2, y)
    np.testing.assert_array_equal(p2, p)
    # dgrad of the transposed geometry (256 -> 64 as seen from dy): accumulate epilogue through the reduce launch
    dy = RNG.standard_normal((n, h, w_, cout)).astype(np.float32)
    prev = RNG.standard_normal(x.shape).astype(np.float32)
    a = u.conv_dgrad(dy, w, x.shape, 1, 'SAME', 1, 'float32', accumulate_into=prev)
    assert rel_l2(a, O.conv2d_dgrad(dy, w, x.shape, 1, 'SAME', 1) + prev) <= 2e-5
