"""GPU parity tests, op by op: HIP kernels (through the C-ABI) vs the NumPy oracle on the same seeded inputs.

Tolerances: fp32 path — north_star's 1e-3 relative (we assert much tighter: rel-L2 <= 2e-5 and max-abs <= 1e-3 of the
reference's max magnitude); bf16 path — oracle evaluated in float64 on the bf16-rounded inputs, rel-L2 <= 8e-3
(bf16 has 8 significant bits; outputs are rounded once).  Integer outputs (arg-max) are bit-exact.
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ops as O  # noqa: E402


def _u():
    import abi_util
    return abi_util


RNG = np.random.default_rng(11)
DTYPES = ['float32', 'bfloat16', 'float16']        # float16: the reference's own low precision (f-4), same kernels as bf16


def check(got, ref, dtype, what='', rel=None, mx=None):
    ref = np.asarray(ref, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    assert got.shape == ref.shape, '{}: shape {} vs {}'.format(what, got.shape, ref.shape)
    assert np.isfinite(got).all(), '{}: non-finite output'.format(what)
    scale = max(np.abs(ref).max(), 1e-30)
    rl2 = np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30)
    mabs = np.abs(got - ref).max() / scale
    # defaults per storage type (float16: 11 significant bits against bf16's 8); explicit bounds are the bf16 ones
    rel = rel if rel is not None else {'float32': 2e-5, 'bfloat16': 8e-3, 'float16': 1.5e-3}[dtype]
    mx = mx if mx is not None else {'float32': 1e-3, 'bfloat16': 3e-2, 'float16': 6e-3}[dtype]
    assert rl2 <= rel and mabs <= mx, '{} [{}]: rel-L2 {:.3e} (<= {:.1e}), max-abs/scale {:.3e} (<= {:.1e})'.format(what, dtype, rl2, rel, mabs, mx)


def q(a, dtype):
    return _u().lp_round(a, dtype)


# n, h, w, cin, cout, k, stride, padding, dilation
CONV_CASES = [
    (2, 8, 8, 16, 32, 1, 1, 'SAME', 1),          # 1x1 plain GEMM path
    (2, 8, 8, 16, 32, 1, 2, 'SAME', 1),          # 1x1 / 2 (projection shortcut)
    (2, 7, 7, 32, 16, 3, 1, 'SAME', 1),          # 3x3 / 1
    (2, 16, 16, 16, 16, 3, 2, 'SAME', 1),        # 3x3 / 2 even H: pads (0,1)
    (2, 15, 15, 16, 16, 3, 2, 'SAME', 1),        # 3x3 / 2 odd H: pads (1,1)
    (3, 14, 14, 64, 64, 3, 1, 'SAME', 1),        # one full 64-wide K step per tap
    (2, 14, 14, 128, 256, 1, 1, 'SAME', 1),      # multi K-step, BN=128 tiles
    (5, 9, 11, 72, 136, 3, 1, 'SAME', 1),        # ragged M / N / K tails
    (2, 9, 9, 8, 16, 7, 1, 'VALID', 1),          # VGG-head style VALID
    (1, 14, 14, 16, 16, 3, 1, 'SAME', 2),        # dilated (DeepLab row)
    (2, 8, 6, 5, 7, 3, 1, 'SAME', 1),            # odd channels: naive fallback kernels
    (1, 1, 1, 32, 16, 3, 1, 'SAME', 1),          # 1x1 map (VGG trunk at 8x8 ends here)
    (2, 9, 9, 64, 19, 1, 1, 'SAME', 1),          # "skinny" 1x1: 19 class logits on a chunked input (segmentation head)
    (3, 5, 7, 32, 5, 1, 1, 'SAME', 1),           # skinny, one accumulator bucket
    (1, 17, 13, 256, 27, 1, 1, 'SAME', 1),       # skinny fwd / dgrad (32-wide bucket), fallback wgrad
    (64, 1, 1, 6, 96, 1, 1, 'SAME', 1),          # few INPUT channels (SE expand conv): dgrad / wgrad run the skinny kernels mirrored
    (3, 5, 7, 20, 64, 1, 1, 'SAME', 1),          # the same on a spatial map
    (300, 1, 1, 1152, 48, 1, 1, 'SAME', 1),      # SE reduce conv of the widest B0 block (few pixels: chunk-parallel skinny dgrad)
    (130, 1, 1, 672, 28, 1, 1, 'SAME', 1),       # SE reduce conv, 28 outputs: wave-per-pixel skinny forward (ragged last block of 4 pixels)
    (67, 1, 1, 20, 480, 1, 1, 'SAME', 1),        # SE expand conv: its dgrad is the wave-per-pixel skinny forward over dy
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case, dtype):
    u = _u()
    n, h, w_, cin, cout, k, s, pad, dil = case
    x = RNG.standard_normal((n, h, w_, cin)).astype(np.float32)
    w = (RNG.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    xq, wq = q(x, dtype), q(w, dtype)
    y_ref = O.conv2d_fwd(xq, wq, s, pad, dil)
    check(u.conv_fwd(x, w, s, pad, dil, dtype), y_ref, dtype, 'conv_fwd')
    dy = RNG.standard_normal(y_ref.shape).astype(np.float32)
    dyq = q(dy, dtype)
    check(u.conv_dgrad(dy, w, x.shape, s, pad, dil, dtype), O.conv2d_dgrad(dyq, wq, x.shape, s, pad, dil), dtype, 'conv_dgrad')
    dw_ref = O.conv2d_wgrad(xq, dyq, w.shape, s, pad, dil)
    dw, db = u.conv_wgrad(x, dy, w.shape, s, pad, dil, dtype, with_bias=True)
    check(dw, dw_ref, dtype, 'conv_wgrad', rel=2e-5 if dtype == 'float32' else 2e-3)   # fp32 accumulation of exact products
    check(db, O.bias_add_bwd(dyq), dtype, 'bias_grad', rel=2e-5 if dtype == 'float32' else 2e-3)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', [CONV_CASES[1], CONV_CASES[3], CONV_CASES[4], CONV_CASES[7], CONV_CASES[10]])
def test_conv_prepacked_operands_equal_per_call_packing(case, dtype):
    """Weights packed once by the batched launch (mcn_conv2d_pack_*) give bit-identical convolutions."""
    u = _u()
    n, h, w_, cin, cout, k, s, pad, dil = case
    x = RNG.standard_normal((n, h, w_, cin)).astype(np.float32)
    w = (RNG.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    y0 = u.conv_fwd(x, w, s, pad, dil, dtype)
    np.testing.assert_array_equal(u.conv_fwd(x, w, s, pad, dil, dtype, use_prepack=True), y0)
    dy = RNG.standard_normal(y0.shape).astype(np.float32)
    np.testing.assert_array_equal(u.conv_dgrad(dy, w, x.shape, s, pad, dil, dtype, use_prepack=True), u.conv_dgrad(dy, w, x.shape, s, pad, dil, dtype))


@pytest.mark.parametrize('dtype', DTYPES)
def test_conv_stem_padded_channels(dtype):
    """Stem: 7x7/2 on 3 channels stored with a channel stride of one 16-byte chunk (pads (2,3))."""
    u = _u()
    cs = 4 if dtype == 'float32' else 8
    x = RNG.random((2, 32, 32, 3)).astype(np.float32)
    xp = np.zeros((2, 32, 32, cs), np.float32)
    xp[..., :3] = x
    w = (RNG.standard_normal((7, 7, 3, 16)) / 12).astype(np.float32)
    y_ref = O.conv2d_fwd(q(x, dtype), q(w, dtype), 2, 'SAME')
    check(u.conv_fwd(xp, w, 2, 'SAME', 1, dtype, x_cs=cs), y_ref, dtype, 'stem fwd')
    dy = RNG.standard_normal(y_ref.shape).astype(np.float32)
    check(u.conv_wgrad(xp, dy, w.shape, 2, 'SAME', 1, dtype, x_cs=cs), O.conv2d_wgrad(q(x, dtype), q(dy, dtype), w.shape, 2, 'SAME'), dtype,
          'stem wgrad', rel=2e-5 if dtype == 'float32' else 2e-3)


@pytest.mark.parametrize('dtype', DTYPES)
def test_conv_bias_and_dgrad_accumulate(dtype):
    u = _u()
    x = RNG.standard_normal((2, 8, 8, 16)).astype(np.float32)
    w = (RNG.standard_normal((3, 3, 16, 32)) / 12).astype(np.float32)
    b = RNG.standard_normal(32).astype(np.float32)
    check(u.conv_fwd(x, w, 1, 'SAME', 1, dtype, bias=b), O.conv2d_fwd(q(x, dtype), q(w, dtype), 1, 'SAME') + b, dtype, 'conv+bias')
    dy = RNG.standard_normal((2, 8, 8, 32)).astype(np.float32)
    base = RNG.standard_normal(x.shape).astype(np.float32)
    ref = q(base, dtype) + O.conv2d_dgrad(q(dy, dtype), q(w, dtype), x.shape, 1, 'SAME')
    check(u.conv_dgrad(dy, w, x.shape, 1, 'SAME', 1, dtype, accumulate_into=base), ref, dtype, 'dgrad accumulate')
    # strided 1x1 accumulate (projection shortcut joining conv_0's dgrad)
    w1 = (RNG.standard_normal((1, 1, 16, 32)) / 4).astype(np.float32)
    dy1 = RNG.standard_normal((2, 4, 4, 32)).astype(np.float32)
    ref = q(base, dtype) + O.conv2d_dgrad(q(dy1, dtype), q(w1, dtype), x.shape, 2, 'SAME')
    check(u.conv_dgrad(dy1, w1, x.shape, 2, 'SAME', 1, dtype, accumulate_into=base), ref, dtype, 'strided dgrad accumulate')
    # skinny 1x1 (19 outputs): bias in the forward, accumulate in the dgrad; the kernel the library reports is the skinny one
    from myconvnet_amd import _ffi
    xs = RNG.standard_normal((2, 6, 5, 64)).astype(np.float32)
    ws_ = (RNG.standard_normal((1, 1, 64, 19)) / 8).astype(np.float32)
    bs = RNG.standard_normal(19).astype(np.float32)
    buf = ctypes.create_string_buffer(128)
    g = u.geom(xs.shape, ws_.shape, 1, 'SAME')
    for op, nm in ((_ffi.CONV_FWD, 'skinny_conv_fwd'), (_ffi.CONV_DGRAD, 'skinny_conv_dgrad'), (_ffi.CONV_WGRAD, 'skinny_conv_wgrad')):
        assert _ffi.lib.mcn_conv2d_kernel_name(op, ctypes.byref(g), u.MDT[dtype], buf, 128) == 1 and buf.value.decode().startswith(nm), buf.value
    check(u.conv_fwd(xs, ws_, 1, 'SAME', 1, dtype, bias=bs), O.conv2d_fwd(q(xs, dtype), q(ws_, dtype), 1, 'SAME') + bs, dtype, 'skinny conv+bias')
    dys = RNG.standard_normal((2, 6, 5, 19)).astype(np.float32)
    bases = RNG.standard_normal(xs.shape).astype(np.float32)
    ref = q(bases, dtype) + O.conv2d_dgrad(q(dys, dtype), q(ws_, dtype), xs.shape, 1, 'SAME')
    check(u.conv_dgrad(dys, ws_, xs.shape, 1, 'SAME', 1, dtype, accumulate_into=bases), ref, dtype, 'skinny dgrad accumulate')
    # mirrored case (6 input channels): accumulate in the dgrad
    xm = RNG.standard_normal((5, 3, 4, 6)).astype(np.float32)
    wm = (RNG.standard_normal((1, 1, 6, 32)) / 3).astype(np.float32)
    dym = RNG.standard_normal((5, 3, 4, 32)).astype(np.float32)
    basem = RNG.standard_normal(xm.shape).astype(np.float32)
    refm = q(basem, dtype) + O.conv2d_dgrad(q(dym, dtype), q(wm, dtype), xm.shape, 1, 'SAME')
    check(u.conv_dgrad(dym, wm, xm.shape, 1, 'SAME', 1, dtype, accumulate_into=basem), refm, dtype, 'few-input dgrad accumulate')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(4, 6, 6, 16), (2, 5, 7, 24), (3, 4, 4, 7), (8, 14, 14, 256), (2, 3, 3, 2048)])
def test_bn_fwd_bwd(shape, dtype):
    u = _u()
    c = shape[-1]
    x = (RNG.standard_normal(shape) * 2.0 + 3.0).astype(np.float32)        # non-zero mean: exercises the shifted sums
    g = (0.5 + RNG.random(c)).astype(np.float32)
    b = RNG.standard_normal(c).astype(np.float32)
    xq = q(x, dtype)
    y, bm, bv, sm, si = O.bn_fwd_train(xq, g.astype(np.float64), b.astype(np.float64), 1e-3)
    run0 = (RNG.standard_normal(c).astype(np.float32), (0.5 + RNG.random(c)).astype(np.float32))
    out = u.bn_fwd_train(x, g, b, 1e-3, dtype, running=run0, momentum=0.9)
    check(out['y'], y, dtype, 'bn y')
    check(out['batch_mean'], bm, 'float32', 'bn batch_mean', rel=1e-5)
    check(out['batch_var'], bv, 'float32', 'bn batch_var (unbiased)', rel=1e-5)
    check(out['save_invstd'], si, 'float32', 'bn invstd', rel=1e-5)
    rm, rv = O.bn_running_update(run0[0].astype(np.float64), run0[1].astype(np.float64), bm, bv, 0.9)
    check(out['running_mean'], rm, 'float32', 'running mean', rel=1e-5)
    check(out['running_var'], rv, 'float32', 'running var', rel=1e-5)
    dy = RNG.standard_normal(shape).astype(np.float32)
    dx, dg, db = O.bn_bwd(q(dy, dtype), xq, g.astype(np.float64), sm, si)
    gdx, gdg, gdb, _ = u.bn_bwd(dy, x, None, g, sm, si, dtype)
    check(gdx, dx, dtype, 'bn dx')
    check(gdg, dg, 'float32', 'bn dgamma', rel=1e-4)
    check(gdb, db, 'float32', 'bn dbeta', rel=1e-4)


@pytest.mark.parametrize('dtype', DTYPES)
def test_bn_fused_add_relu(dtype):
    """y = relu(bn(x) + skip); backward masks with [y > 0] and fans the masked gradient out to the skip branch."""
    u = _u()
    shape = (4, 7, 7, 32)
    x = RNG.standard_normal(shape).astype(np.float32)
    skip = RNG.standard_normal(shape).astype(np.float32)
    g = (0.5 + RNG.random(32)).astype(np.float32)
    b = RNG.standard_normal(32).astype(np.float32)
    xq, sq = q(x, dtype), q(skip, dtype)
    ybn, _, _, sm, si = O.bn_fwd_train(xq, g.astype(np.float64), b.astype(np.float64), 1e-3)
    y_ref = np.maximum(ybn + sq, 0)
    out = u.bn_fwd_train(x, g, b, 1e-3, dtype, skip=skip, act=1)
    check(out['y'], y_ref, dtype, 'bn+add+relu')
    dy = RNG.standard_normal(shape).astype(np.float32)
    ydev = out['y']                                    # the mask must come from the values the device stored
    dmask = q(dy, dtype) * (ydev > 0)
    dx, dg, db = O.bn_bwd(dmask, xq, g.astype(np.float64), sm, si)
    gdx, gdg, gdb, gds = u.bn_bwd(dy, x, ydev, g, sm, si, dtype, act=1, want_dskip=True)
    check(gdx, dx, dtype, 'fused dx')
    check(gds, dmask, dtype, 'fused dskip', rel=1e-6 if dtype == 'float32' else 1e-6)
    check(gdg, dg, 'float32', 'fused dgamma', rel=1e-4)
    check(gdb, db, 'float32', 'fused dbeta', rel=1e-4)
    # the byte mask written by the forward ([y > 0], one byte per 16-byte chunk) replaces y in the backward, bit for bit
    outm = u.bn_fwd_train(x, g, b, 1e-3, dtype, skip=skip, act=1, want_mask=True)
    np.testing.assert_array_equal(outm['y'], ydev)
    vec = 4 if dtype == 'float32' else 8
    bits = (ydev.reshape(-1, vec) > 0).astype(np.uint32)
    ref_mask = (bits << np.arange(vec, dtype=np.uint32)).sum(-1).astype(np.uint8)
    np.testing.assert_array_equal(outm['relu_mask'][:ref_mask.size], ref_mask)
    mdx, mdg, mdb, mds = u.bn_bwd(dy, x, None, g, sm, si, dtype, act=1, want_dskip=True, relu_mask=outm['relu_mask'])
    for a_, b_ in ((mdx, gdx), (mdg, gdg), (mdb, gdb), (mds, gds)):
        np.testing.assert_array_equal(a_, b_)


@pytest.mark.parametrize('dtype', DTYPES)
def test_bn_relu_mask_recomputed_from_x(dtype):
    """y = relu(bn(x)) without a residual: the backward may drop the read of y and recompute the mask from x with the
    forward's own fp32 expression; it must agree with the stored-y mask element for element."""
    u = _u()
    shape = (8, 9, 9, 64)
    x = RNG.standard_normal(shape).astype(np.float32)
    g = (0.5 + RNG.random(64)).astype(np.float32)
    b = (0.3 * RNG.standard_normal(64)).astype(np.float32)
    out = u.bn_fwd_train(x, g, b, 1e-3, dtype, act=1)
    dy = RNG.standard_normal(shape).astype(np.float32)
    sm, si = out['save_mean'], out['save_invstd']
    ref = u.bn_bwd(dy, x, out['y'], g, sm, si, dtype, act=1)                 # mask from the stored forward output
    got = u.bn_bwd(dy, x, None, g, sm, si, dtype, act=1, beta=b)             # mask recomputed from x
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])
    np.testing.assert_array_equal(got[2], ref[2])
    xq = q(x, dtype)
    dx, dg, db = O.bn_bwd(q(dy, dtype) * (out['y'] > 0), xq, g.astype(np.float64), sm.astype(np.float64), si.astype(np.float64))
    check(got[0], dx, dtype, 'dx (recomputed mask)')
    check(got[1], dg, 'float32', 'dgamma', rel=1e-4)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(4, 6, 6, 16), (3, 4, 4, 7), (8, 14, 14, 256)])
@pytest.mark.parametrize('act', [0, 1, 2])
def test_bn_bwd_frozen(shape, act, dtype):
    """Gradient of fused_batch_norm(is_training=False) inside a training graph (update_batch_norm=False): affine in x."""
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    c = shape[-1]
    m = int(np.prod(shape[:-1]))
    x = RNG.standard_normal(shape).astype(np.float32)
    dy = RNG.standard_normal(shape).astype(np.float32)
    gamma = (0.5 + RNG.random(c)).astype(np.float32)
    beta = (0.3 * RNG.standard_normal(c)).astype(np.float32)
    mean = (0.2 * RNG.standard_normal(c)).astype(np.float32)
    var = (0.5 + RNG.random(c)).astype(np.float32)
    xq, dyq = q(x, dtype), q(dy, dtype)
    z = O.bn_fwd_infer(xq, gamma.astype(np.float64), beta.astype(np.float64), mean.astype(np.float64), var.astype(np.float64), 1e-3)
    if act == 1:
        yq = q(np.maximum(z, 0), dtype)
        dz = dyq * (yq > 0)
    elif act == 2:
        sg = 1.0 / (1.0 + np.exp(-z))
        yq = q(z * sg, dtype)
        dz = dyq * (sg + z * sg * (1 - sg))
    else:
        yq, dz = q(z, dtype), dyq
    rdx, rdg, rdb = O.bn_bwd_frozen(dz, xq, gamma.astype(np.float64), mean.astype(np.float64), var.astype(np.float64), 1e-3)
    xd, dyd, yd = u.dev(x, dtype), u.dev(dy, dtype), u.dev(yq.astype(np.float32), dtype)
    gd, bd, md, vd = u.dev(gamma), u.dev(beta), u.dev(mean), u.dev(var)
    dx = torch.full(shape, float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    dg, db = torch.zeros(c, dtype=torch.float32, device=u.DEV), torch.zeros(c, dtype=torch.float32, device=u.DEV)
    ws = u.workspace(lib.mcn_bn_workspace_bytes(m, c))
    _ffi.check(lib.mcn_bn_bwd_frozen(dyd.data_ptr(), xd.data_ptr(), yd.data_ptr(), gd.data_ptr(), bd.data_ptr(), md.data_ptr(), vd.data_ptr(), 1e-3,
                                     dx.data_ptr(), 0, dg.data_ptr(), db.data_ptr(), 0.5, m, c, act, u.MDT[dtype], ws.data_ptr(), ws.numel() * 4, u.stream()))
    check(u.host(dx), rdx, dtype, 'dx')
    check(u.host(dg), 0.5 * rdg, 'float32', 'dgamma', rel=2e-5 if dtype == 'float32' else 1e-2, mx=1e-3 if dtype == 'float32' else 3e-2)
    check(u.host(db), 0.5 * rdb, 'float32', 'dbeta', rel=2e-5 if dtype == 'float32' else 1e-2, mx=1e-3 if dtype == 'float32' else 3e-2)
    # frozen gamma / beta (blocks_to_train): null gradient pointers
    _ffi.check(lib.mcn_bn_bwd_frozen(dyd.data_ptr(), xd.data_ptr(), yd.data_ptr(), gd.data_ptr(), bd.data_ptr(), md.data_ptr(), vd.data_ptr(), 1e-3,
                                     dx.data_ptr(), 0, 0, 0, 1.0, m, c, act, u.MDT[dtype], ws.data_ptr(), ws.numel() * 4, u.stream()))
    check(u.host(dx), rdx, dtype, 'dx (no dgamma / dbeta)')
    if act == 1:
        assert lib.mcn_bn_bwd_frozen(dyd.data_ptr(), xd.data_ptr(), 0, gd.data_ptr(), bd.data_ptr(), md.data_ptr(), vd.data_ptr(), 1e-3, dx.data_ptr(), 0, 0, 0,
                                     1.0, m, c, act, u.MDT[dtype], ws.data_ptr(), ws.numel() * 4, u.stream()) == _ffi.E_BADARG


@pytest.mark.parametrize('dtype', DTYPES)
def test_bn_infer_and_affine(dtype):
    u = _u()
    from myconvnet_amd import _ffi
    lib = _ffi.lib
    shape = (2, 5, 5, 16)
    x = RNG.standard_normal(shape).astype(np.float32)
    g, b = (0.5 + RNG.random(16)).astype(np.float32), RNG.standard_normal(16).astype(np.float32)
    mu, var = RNG.standard_normal(16).astype(np.float32), (0.5 + RNG.random(16)).astype(np.float32)
    xd, y = u.dev(x, dtype), torch.zeros(shape, dtype=u.TDT[dtype], device=u.DEV)
    gd, bd, mud, vard = u.dev(g), u.dev(b), u.dev(mu), u.dev(var)      # keep the device buffers alive across the async launches
    _ffi.check(lib.mcn_bn_fwd_infer(xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), mud.data_ptr(), vard.data_ptr(), 0,
                                    y.data_ptr(), 50, 16, 1e-3, 0, u.MDT[dtype], u.stream()))
    check(u.host(y), O.bn_fwd_infer(q(x, dtype), g.astype(np.float64), b.astype(np.float64), mu.astype(np.float64), var.astype(np.float64)), dtype, 'bn infer')
    _ffi.check(lib.mcn_channel_affine(xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), y.data_ptr(), 50, 16, u.MDT[dtype], u.stream()))
    check(u.host(y), q(x, dtype) * g + b, dtype, 'channel affine')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('cfg', [(16, 3, 2, 'SAME', 16), (12, 3, 2, 'SAME', 24), (15, 3, 2, 'SAME', 8), (8, 2, 2, 'SAME', 32), (7, 2, 2, 'SAME', 5), (1, 2, 2, 'SAME', 8),
                                 (8, 2, 2, 'VALID', 16)])
def test_maxpool(cfg, dtype):
    u = _u()
    from myconvnet_amd import _ffi
    lib = _ffi.lib
    h, k, s, pad, c = cfg
    x = RNG.standard_normal((2, h, h, c)).astype(np.float32)
    x[0, :, :, 0] = 0.0                              # ties (post-ReLU zeros are the common case)
    xq = q(x, dtype)
    y_ref, arg_ref = O.maxpool_fwd(xq, k, s, pad)
    pt, _, pl, _ = O.resolve_pads(h, h, k, k, s, s, pad)
    oh = y_ref.shape[1]
    xd = u.dev(x, dtype)
    y = torch.zeros(y_ref.shape, dtype=u.TDT[dtype], device=u.DEV)
    arg = torch.zeros(y_ref.shape, dtype=torch.int8, device=u.DEV)
    _ffi.check(lib.mcn_maxpool_fwd(xd.data_ptr(), y.data_ptr(), arg.data_ptr(), 2, h, h, c, k, k, s, s, pt, pl, oh, oh, u.MDT[dtype], u.stream()))
    np.testing.assert_array_equal(u.host(y).astype(np.float64), y_ref)              # max is exact
    np.testing.assert_array_equal(arg.cpu().numpy(), arg_ref)                       # integer arg-max bit-exact
    dy = RNG.standard_normal(y_ref.shape).astype(np.float32)
    dx = torch.zeros(x.shape, dtype=u.TDT[dtype], device=u.DEV)
    dyd = u.dev(dy, dtype)
    _ffi.check(lib.mcn_maxpool_bwd(dyd.data_ptr(), arg.data_ptr(), dx.data_ptr(), 2, h, h, c, k, k, s, s, pt, pl, oh, oh, u.MDT[dtype],
                                   u.stream()))
    check(u.host(dx), O.maxpool_bwd(q(dy, dtype), arg_ref, x.shape, k, s, pad), dtype, 'maxpool bwd')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('cfg', [(8, 3, 2, 'SAME', 16), (7, 3, 1, 'SAME', 8), (8, 2, 2, 'VALID', 5)])
def test_avgpool(cfg, dtype):
    u = _u()
    from myconvnet_amd import _ffi
    lib = _ffi.lib
    h, k, s, pad, c = cfg
    x = RNG.standard_normal((2, h, h, c)).astype(np.float32)
    y_ref = O.avgpool_fwd(q(x, dtype), k, s, pad)
    pt, _, pl, _ = O.resolve_pads(h, h, k, k, s, s, pad)
    oh = y_ref.shape[1]
    y = torch.zeros(y_ref.shape, dtype=u.TDT[dtype], device=u.DEV)
    xd = u.dev(x, dtype)
    _ffi.check(lib.mcn_avgpool_fwd(xd.data_ptr(), y.data_ptr(), 2, h, h, c, k, k, s, s, pt, pl, oh, oh, u.MDT[dtype], u.stream()))
    check(u.host(y), y_ref, dtype, 'avgpool fwd')
    dy = RNG.standard_normal(y_ref.shape).astype(np.float32)
    dx = torch.zeros(x.shape, dtype=u.TDT[dtype], device=u.DEV)
    dyd = u.dev(dy, dtype)
    _ffi.check(lib.mcn_avgpool_bwd(dyd.data_ptr(), dx.data_ptr(), 2, h, h, c, k, k, s, s, pt, pl, oh, oh, u.MDT[dtype], u.stream()))
    check(u.host(dx), O.avgpool_bwd(q(dy, dtype), x.shape, k, s, pad), dtype, 'avgpool bwd')


@pytest.mark.parametrize('dtype', DTYPES)
def test_global_avgpool_and_eltwise(dtype):
    u = _u()
    from myconvnet_amd import _ffi
    lib = _ffi.lib
    md = u.MDT[dtype]
    x = RNG.standard_normal((3, 7, 7, 40)).astype(np.float32)
    y = torch.zeros((3, 40), dtype=u.TDT[dtype], device=u.DEV)
    xd = u.dev(x, dtype)
    _ffi.check(lib.mcn_global_avgpool_fwd(xd.data_ptr(), y.data_ptr(), 3, 49, 40, md, u.stream()))
    check(u.host(y), O.global_avgpool_fwd(q(x, dtype)), dtype, 'gap fwd')
    dy = RNG.standard_normal((3, 40)).astype(np.float32)
    dx = torch.zeros(x.shape, dtype=u.TDT[dtype], device=u.DEV)
    dyd = u.dev(dy, dtype)
    _ffi.check(lib.mcn_global_avgpool_bwd(dyd.data_ptr(), dx.data_ptr(), 3, 49, 40, md, u.stream()))
    check(u.host(dx), O.global_avgpool_bwd(q(dy, dtype), x.shape), dtype, 'gap bwd')
    # relu / relu_bwd / add_relu / accumulate on a ragged length (vector body + scalar tail)
    n = 1003
    a, b = RNG.standard_normal(n).astype(np.float32), RNG.standard_normal(n).astype(np.float32)
    ad, bd = u.dev(a, dtype), u.dev(b, dtype)
    o = torch.zeros(n, dtype=u.TDT[dtype], device=u.DEV)
    _ffi.check(lib.mcn_relu_fwd(ad.data_ptr(), o.data_ptr(), n, md, u.stream()))
    np.testing.assert_array_equal(u.host(o), np.maximum(q(a, dtype), 0))
    _ffi.check(lib.mcn_relu_bwd(ad.data_ptr(), bd.data_ptr(), o.data_ptr(), n, md, u.stream()))
    np.testing.assert_array_equal(u.host(o), q(a, dtype) * (q(b, dtype) > 0))
    _ffi.check(lib.mcn_add_relu_fwd(ad.data_ptr(), bd.data_ptr(), o.data_ptr(), n, 1, md, u.stream()))
    check(u.host(o), np.maximum(q(a, dtype) + q(b, dtype), 0), dtype, 'add_relu')
    _ffi.check(lib.mcn_accumulate(ad.data_ptr(), bd.data_ptr(), n, md, u.stream()))
    check(u.host(ad), q(a, dtype) + q(b, dtype), dtype, 'accumulate')
    f = torch.zeros(n, dtype=torch.float32, device=u.DEV)
    _ffi.check(lib.mcn_cast(bd.data_ptr(), md, f.data_ptr(), _ffi.F32, n, u.stream()))
    np.testing.assert_array_equal(u.host(f), q(b, dtype))


@pytest.mark.parametrize('layout', ['NHWC', 'NCHW'])
@pytest.mark.parametrize('dtype', DTYPES)
def test_input_prep_and_one_hot(dtype, layout):
    u = _u()
    from myconvnet_amd import _ffi
    lib = _ffi.lib
    x = RNG.random((2, 6, 5, 3)).astype(np.float32)
    src = x if layout == 'NHWC' else np.ascontiguousarray(x.transpose(0, 3, 1, 2))
    cs = 8
    y = torch.full((2, 6, 5, cs), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    srcd = u.dev(src)
    _ffi.check(lib.mcn_input_prep(srcd.data_ptr(), y.data_ptr(), 2, 6, 5, 3, cs, 0.5, 2.0, _ffi.NCHW if layout == 'NCHW' else _ffi.NHWC,
                                  u.MDT[dtype], u.stream()))
    got = u.host(y)
    check(got[..., :3], O.input_prep(x.astype(np.float64)), dtype, 'input_prep')
    assert (got[..., 3:] == 0).all()
    labels = np.array([1, 3, np.nan, 9, 0, 12, -1], dtype=np.float32)
    oh = torch.zeros((7, 10), dtype=torch.float32, device=u.DEV)
    labd = u.dev(labels)
    _ffi.check(lib.mcn_one_hot(labd.data_ptr(), oh.data_ptr(), 7, 10, u.stream()))
    np.testing.assert_array_equal(u.host(oh), O.one_hot_labels(labels, 10))


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('dims', [(8, 64, 40), (5, 2048 // 8, 1000), (3, 10, 7)])
def test_fc_fwd_bwd(dims, dtype):
    u = _u()
    from myconvnet_amd import _ffi
    lib = _ffi.lib
    B, In, Out = dims
    x = RNG.standard_normal((B, In)).astype(np.float32)
    w = (RNG.standard_normal((In, Out)) / np.sqrt(In)).astype(np.float32)
    b = RNG.standard_normal(Out).astype(np.float32)
    md = u.MDT[dtype]
    ws = u.workspace(lib.mcn_fc_workspace_bytes(B, In, Out, md))
    xd, wd, bd = u.dev(x, dtype), u.dev(w), u.dev(b)
    y = torch.zeros((B, Out), dtype=u.TDT[dtype], device=u.DEV)
    _ffi.check(lib.mcn_fc_fwd(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, In, Out, md, ws.data_ptr(), ws.numel() * 4, u.stream()))
    xq, wq = q(x, dtype), q(w, dtype)
    check(u.host(y), O.fc_fwd(xq, wq, b.astype(np.float64)), dtype, 'fc fwd')
    dy = RNG.standard_normal((B, Out)).astype(np.float32)
    dx = torch.zeros((B, In), dtype=u.TDT[dtype], device=u.DEV)
    dw = torch.zeros((In, Out), dtype=torch.float32, device=u.DEV)
    db = torch.zeros(Out, dtype=torch.float32, device=u.DEV)
    dyd = u.dev(dy, dtype)
    _ffi.check(lib.mcn_fc_bwd(dyd.data_ptr(), xd.data_ptr(), wd.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), 1.0, B, In, Out,
                              md, ws.data_ptr(), ws.numel() * 4, u.stream()))
    rdx, rdw, rdb = O.fc_bwd(q(dy, dtype), xq, wq)
    check(u.host(dx), rdx, dtype, 'fc dx')
    check(u.host(dw), rdw, dtype, 'fc dw', rel=2e-5 if dtype == 'float32' else 2e-3)
    check(u.host(db), rdb, dtype, 'fc db', rel=2e-5 if dtype == 'float32' else 2e-3)


@pytest.mark.parametrize('ls', [0.0, 0.1])
def test_softmax_xent(ls):
    u = _u()
    from myconvnet_amd import _ffi
    lib = _ffi.lib
    B, C = 7, 1000
    logits = (RNG.standard_normal((B, C)) * 3).astype(np.float32)
    labels = np.array([1, 3, np.nan, 999, 0, 1200, 5], dtype=np.float32)
    oh = O.one_hot_labels(labels, C)
    cw = (0.5 + RNG.random(C)).astype(np.float32)
    pred_r, loss_r, ce_r, dl_r = O.softmax_xent_fwd_bwd(logits.astype(np.float64), oh.astype(np.float64), cw.astype(np.float64), ls, 4.0)
    pred, dl = [torch.zeros((B, C), dtype=torch.float32, device=u.DEV) for _ in range(2)]
    ce, coef, loss = [torch.zeros(8, dtype=torch.float32, device=u.DEV) for _ in range(3)]
    ld, ohd, cwd = u.dev(logits), u.dev(oh), u.dev(cw)
    _ffi.check(lib.mcn_softmax_xent_fwd_bwd(ld.data_ptr(), ohd.data_ptr(), cwd.data_ptr(), pred.data_ptr(), ce.data_ptr(),
                                            coef.data_ptr(), dl.data_ptr(), loss.data_ptr(), B, C, ls, 4.0, u.stream()))
    check(u.host(pred), pred_r, 'float32', 'pred', rel=1e-5)
    check(u.host(ce)[:B], ce_r, 'float32', 'ce', rel=1e-5)
    check(u.host(dl), dl_r, 'float32', 'dlogits', rel=1e-5)
    assert abs(u.host(loss)[0] - loss_r) <= 1e-5 * abs(loss_r)
    np.testing.assert_array_equal(u.host(pred).argmax(-1), pred_r.argmax(-1))          # integer arg-max bit-exact


@pytest.mark.parametrize('focal', [dict(focal_gamma=2.0), dict(sigmoid_focal_alpha=4.0), dict(focal_gamma=1.5, sigmoid_focal_alpha=3.0)])
@pytest.mark.parametrize('rows', [False, True])
def test_softmax_xent_focal_factors(focal, rows):
    """focal_loss_factor / sigmoid_focal_loss_factor (convnet.py:581-592) in both loss kernels: (1 - p_t)^gamma differentiated THROUGH the softmax,
    the sigmoid factor under stop_gradient; with label smoothing, class weights, an ignored row and loss scaling.  (The oracle's gamma gradient is
    checked against finite differences in tests/test_oracle_vs_torch.py::test_focal_gradient_matches_autograd.)"""
    u = _u()
    from myconvnet_amd import _ffi
    lib = _ffi.lib
    B, C = (300, 19) if rows else (9, 257)
    fg, sa = focal.get('focal_gamma', 0.0), focal.get('sigmoid_focal_alpha', 0.0)
    logits = (RNG.standard_normal((B, C)) * 2).astype(np.float32)
    lab = RNG.integers(0, C, B)
    logits[np.arange(B), lab] += 5.0                                      # p_t of a sizeable fraction: the factors matter
    oh = np.eye(C, dtype=np.float32)[lab]
    oh[3] = 0.0                                                           # ignored row: p_t = 0, factor 1, weight 0
    cw = (0.5 + RNG.random(C)).astype(np.float32)
    ls = 0.0 if rows else 0.1
    pred_r, loss_r, ce_r, dl_r = O.softmax_xent_fwd_bwd(logits.astype(np.float64), oh.astype(np.float64), cw.astype(np.float64), ls, 2.0, focal_gamma=fg,
                                                        sigmoid_focal_alpha=sa)
    plain = O.softmax_xent_fwd_bwd(logits.astype(np.float64), oh.astype(np.float64), cw.astype(np.float64), ls, 2.0)
    assert abs(loss_r - plain[1]) > 1e-2 * abs(plain[1])                  # the factors are visible
    pred, dl = [torch.zeros((B, C), dtype=torch.float32, device=u.DEV) for _ in range(2)]
    ce, coef = [torch.zeros(B, dtype=torch.float32, device=u.DEV) for _ in range(2)]
    loss = torch.zeros(4, dtype=torch.float32, device=u.DEV)
    ld, ohd, cwd = u.dev(logits), u.dev(oh), u.dev(cw)
    if rows:
        ws = u.workspace(8192)
        _ffi.check(lib.mcn_softmax_xent_rows_focal_fwd_bwd(ld.data_ptr(), ohd.data_ptr(), 0, cwd.data_ptr(), pred.data_ptr(), ce.data_ptr(), coef.data_ptr(), dl.data_ptr(),
                                                           loss.data_ptr(), B, C, ls, 2.0, fg, sa, ws.data_ptr(), ws.numel() * 4, u.stream()))
    else:
        _ffi.check(lib.mcn_softmax_xent_focal_fwd_bwd(ld.data_ptr(), ohd.data_ptr(), cwd.data_ptr(), pred.data_ptr(), ce.data_ptr(), coef.data_ptr(), dl.data_ptr(),
                                                      loss.data_ptr(), B, C, ls, 2.0, fg, sa, u.stream()))
    check(u.host(pred), pred_r, 'float32', 'pred', rel=1e-5)
    check(u.host(ce), ce_r, 'float32', 'factored ce', rel=2e-5)
    check(u.host(dl), dl_r, 'float32', 'dlogits', rel=2e-5)
    assert abs(u.host(loss)[0] - loss_r) <= 1e-5 * abs(loss_r)
    assert lib.mcn_softmax_xent_focal_fwd_bwd(ld.data_ptr(), ohd.data_ptr(), 0, pred.data_ptr(), ce.data_ptr(), coef.data_ptr(), dl.data_ptr(), loss.data_ptr(), B, C, 0.0, 1.0,
                                              -1.0, 0.0, u.stream()) == _ffi.E_BADARG


def test_l1_regulariser_value_and_gradient():
    """l1_reg (convnet.py:553-557): out += factor * sum |w|; gradient l1 * sign(w) added to the flat gradient divided by the tower factor the update
    kernel multiplies back (hyper[3])."""
    u = _u()
    from myconvnet_amd import _ffi
    lib = _ffi.lib
    n = 100003
    w = RNG.standard_normal(n).astype(np.float32)
    w[::17] = 0.0
    g0 = RNG.standard_normal(n).astype(np.float32)
    wd, gd = u.dev(w), u.dev(g0)
    out = torch.full((4,), 1.5, dtype=torch.float32, device=u.DEV)
    ws = u.workspace(8192)
    _ffi.check(lib.mcn_l1_loss(wd.data_ptr(), n, 1e-3, out.data_ptr(), ws.data_ptr(), ws.numel() * 4, u.stream()))
    assert abs(u.host(out)[0] - (1.5 + O.l1_reg_loss([w], 1e-3))) <= 1e-5 * (1.5 + O.l1_reg_loss([w], 1e-3))
    hyper = u.dev(np.array([0.1, 0.0, 0.99, 0.25], np.float32))
    _ffi.check(lib.mcn_l1_grad_h(gd.data_ptr(), wd.data_ptr(), n, 1e-3, hyper.data_ptr(), u.stream()))
    np.testing.assert_allclose(u.host(gd), g0 + np.float32(1e-3 / 0.25) * np.sign(w), rtol=1e-6, atol=1e-7)


def test_sgd_nesterov_ema_l2_trajectory():
    """3-step trajectory of the fused update vs the oracle (EMA of the pre-update value, L2 folded into the gradient)."""
    u = _u()
    from myconvnet_amd import _ffi
    lib = _ffi.lib
    n = 1027
    w0 = RNG.standard_normal(n).astype(np.float32)
    w, a, e = w0.astype(np.float64), np.zeros(n), w0.astype(np.float64)
    pad = (n + 3) // 4 * 4
    wd_, ad_, ed_ = [torch.zeros(pad, dtype=torch.float32, device=u.DEV) for _ in range(3)]
    wd_[:n] = u.dev(w0)
    ed_[:n] = u.dev(w0)
    for step in range(3):
        g = RNG.standard_normal(n).astype(np.float32)
        gd = torch.zeros(pad, dtype=torch.float32, device=u.DEV)
        gd[:n] = u.dev(g)
        d = O.ema_decay(0.99, step)
        lr = 0.05 * (step + 1)
        w, a, e = O.sgd_nesterov_step(w, g.astype(np.float64), a, lr, 0.9, l2=1e-4, ema=e, ema_d=d, wd=1e-3, grad_scale=0.5)
        _ffi.check(lib.mcn_sgd_nesterov_fused(wd_.data_ptr(), gd.data_ptr(), ad_.data_ptr(), ed_.data_ptr(), n, lr, 0.9, 1e-4, 1e-3, d, 0.5, u.stream()))
        check(u.host(wd_)[:n], w, 'float32', 'w step {}'.format(step), rel=1e-6)
        check(u.host(ad_)[:n], a, 'float32', 'accum step {}'.format(step), rel=1e-6)
        check(u.host(ed_)[:n], e, 'float32', 'ema step {}'.format(step), rel=1e-6)


@pytest.mark.parametrize('mode', [0, 1, 2])
def test_decoupled_decay_variants(mode):
    from myconvnet_amd import _ffi
    u = _u()
    n = 100003
    w = (RNG.standard_normal(n) * 0.1).astype(np.float32)
    w[:7] = 0.0                                                              # sign(0) = 0
    wd_ = u.dev(w)
    _ffi.check(_ffi.lib.mcn_decoupled_decay(wd_.data_ptr(), n, 0.03, mode, 0.05, u.stream()))
    ref = O.decoupled_decay(w.astype(np.float64), 0.03, l1=(mode == 1), huber_delta=0.05 if mode == 2 else None)
    check(u.host(wd_), ref, 'float32', 'decay mode {}'.format(mode), rel=1e-6, mx=1e-6)
    assert _ffi.lib.mcn_decoupled_decay(wd_.data_ptr(), n, 0.03, 2, 0.0, u.stream()) == _ffi.E_BADARG
    assert _ffi.lib.mcn_decoupled_decay(wd_.data_ptr(), n, 0.03, 3, 1.0, u.stream()) == _ffi.E_BADARG


def test_l2_loss_ema_and_bn_chain():
    u = _u()
    from myconvnet_amd import _ffi
    lib = _ffi.lib
    n = 40000
    w = RNG.standard_normal(n).astype(np.float32)
    out = torch.full((4,), 2.5, dtype=torch.float32, device=u.DEV)
    ws = u.workspace(8192)
    wdv = u.dev(w)
    _ffi.check(lib.mcn_l2_loss(wdv.data_ptr(), n, 1e-4, out.data_ptr(), ws.data_ptr(), ws.numel() * 4, u.stream()))
    assert abs(u.host(out)[0] - (2.5 + O.l2_reg_loss([w], 1e-4))) < 1e-5
    run = RNG.standard_normal(300).astype(np.float32)
    batch = RNG.standard_normal((4, 300)).astype(np.float32)
    rd = u.dev(run)
    batchd = u.dev(batch)
    _ffi.check(lib.mcn_bn_running_chain(rd.data_ptr(), batchd.data_ptr(), 4, 300, 0.99, u.stream()))
    ref = run.astype(np.float64)
    for k in range(4):
        ref = 0.99 * ref + 0.01 * batch[k]
    check(u.host(rd), ref, 'float32', 'bn chain', rel=1e-6)
    sh = u.dev(run)
    b0 = u.dev(batch[0])
    _ffi.check(lib.mcn_ema_update(sh.data_ptr(), b0.data_ptr(), 300, 0.9, u.stream()))
    check(u.host(sh), 0.9 * run + 0.1 * batch[0], 'float32', 'ema', rel=1e-6)


def test_full_size_properties():
    """Size-independent properties at BASELINE's full layer sizes (B=256 is too slow for the oracle):
    linearity of conv in x, <dy, conv(x)> == <dgrad(dy), x> == <wgrad(x,dy), w> (adjointness), and BN output moments."""
    u = _u()
    # block_2/res_0/conv_1 at reduced batch: 3x3/2, 128->128, 56x56
    n, h, c = 8, 56, 128
    x = RNG.standard_normal((n, h, h, c)).astype(np.float32)
    x2 = RNG.standard_normal((n, h, h, c)).astype(np.float32)
    w = (RNG.standard_normal((3, 3, c, c)) / np.sqrt(9 * c)).astype(np.float32)
    y1, y2 = u.conv_fwd(x, w, 2, 'SAME'), u.conv_fwd(x2, w, 2, 'SAME')
    y12 = u.conv_fwd(x + 2 * x2, w, 2, 'SAME')
    check(y12, y1.astype(np.float64) + 2 * y2, 'float32', 'linearity', rel=1e-5)
    dy = RNG.standard_normal(y1.shape).astype(np.float32)
    lhs = float((dy.astype(np.float64) * y1).sum())
    dx = u.conv_dgrad(dy, w, x.shape, 2, 'SAME')
    dw = u.conv_wgrad(x, dy, w.shape, 2, 'SAME')
    assert abs(lhs - float((dx.astype(np.float64) * x).sum())) <= 1e-4 * abs(lhs) + 1e-2
    assert abs(lhs - float((dw.astype(np.float64) * w).sum())) <= 1e-4 * abs(lhs) + 1e-2
    out = u.bn_fwd_train(y1, np.ones(c, np.float32), np.zeros(c, np.float32), 1e-3)
    yb = out['y'].reshape(-1, c).astype(np.float64)
    assert np.abs(yb.mean(0)).max() < 1e-4
    v = y1.reshape(-1, c).astype(np.float64).var(0)
    np.testing.assert_allclose(yb.var(0), v / (v + 1e-3), rtol=1e-3)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', [(4, 14, 14, 64, 64, 3, 1), (2, 28, 28, 32, 128, 1, 1), (3, 9, 11, 72, 136, 3, 1), (2, 16, 16, 16, 32, 3, 2), (8, 14, 14, 256, 64, 1, 1),
                                  (16, 56, 56, 16, 32, 1, 1),       # > 512 partial rows -> the folded (two-stage) merge
                                  # more tiles than resident workgroups: the persistent 1x1 kernel walks several tiles per workgroup
                                  # (K = 64 in a 2-byte type is a one-step K loop: the LDS buffer parity alternates tile by tile)
                                  (600, 16, 8, 64, 256, 1, 1), (350, 16, 8, 256, 128, 1, 1), (700, 16, 8, 96, 136, 1, 1)])
def test_conv_epilogue_bn_statistics(case, dtype):
    """conv -> BN fusion: the conv writes per-(M tile, wave row) column sums / sums of squares of the values it stores;
    mcn_bn_fwd_train_fused must give what mcn_bn_fwd_train gives on the same conv output (same tf.nn.fused_batch_norm
    semantics), for every tile candidate, including ragged M / N tails."""
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    n, h, w_, cin, cout, k, s = case
    x = (RNG.standard_normal((n, h, w_, cin)) + 3.0).astype(np.float32)       # offset input: |mean| >> std in the conv output
    w = (RNG.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin) + 0.05).astype(np.float32)
    gamma = (0.5 + RNG.random(cout)).astype(np.float32)
    beta = (0.3 * RNG.standard_normal(cout)).astype(np.float32)
    y_plain = u.conv_fwd(x, w, s, 'SAME', 1, dtype)
    ref = u.bn_fwd_train(y_plain, gamma, beta, 1e-3, dtype, act=1, running=(np.zeros(cout, np.float32), np.ones(cout, np.float32)))
    for tile in range(lib.mcn_conv2d_tile_candidates(_ffi.CONV_FWD) + 1):
        g = u.geom(x.shape, w.shape, s, 'SAME')
        g.tile = tile
        rpp = ctypes.c_int32(0)
        rows = lib.mcn_conv2d_bnstats_rows(ctypes.byref(g), u.MDT[dtype], ctypes.byref(rpp))
        counted = rpp.value <= 0                   # one row per persistent workgroup, four planes: the count is explicit
        assert rows > 0 and (counted or rows * rpp.value >= y_plain.size // cout)
        xd, wd = u.dev(x, dtype), u.dev(w)
        y = torch.full(y_plain.shape, float('nan'), dtype=u.TDT[dtype], device=u.DEV)
        part = torch.full((rows, 4 if counted else 3, cout), float('nan'), dtype=torch.float32, device=u.DEV)
        ws = u.workspace(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_FWD, ctypes.byref(g), u.MDT[dtype]))
        _ffi.check(lib.mcn_conv2d_fwd_bnstats(xd.data_ptr(), wd.data_ptr(), 0, 0, y.data_ptr(), part.data_ptr(), ctypes.byref(g), u.MDT[dtype], _ffi.NHWC,
                                              ws.data_ptr(), ws.numel() * 4, u.stream()))
        ys = u.host(y)
        if dtype == 'float32' and n >= 300:
            # (fp32 layers with a thin last round of tiles run that round K-sliced — stream-K tail — so two tile shapes may sum a
            # K loop in different orders: equal up to fp32 rounding, not bit for bit)
            np.testing.assert_allclose(ys, y_plain, rtol=5e-6, atol=1e-5)
        else:
            np.testing.assert_array_equal(ys, y_plain)                          # the stored output is unchanged
        p = u.host(part).astype(np.float64)
        yq = ys.astype(np.float64).reshape(-1, cout)                            # the partials describe the values THIS launch stored
        m = yq.shape[0]
        if counted:
            if rpp.value < 0:
                # compact rows keyed by the channel block: [blocks][rows / blocks][4][BN] (include/mcn.h) -> per channel its block's rows
                bnb = -rpp.value
                nblk = -(-cout // bnb)
                assert rows % nblk == 0
                rpb = rows // nblk
                flat = p.reshape(-1)[:nblk * rpb * 4 * bnb].reshape(nblk, rpb, 4, bnb)
                p = np.stack([flat[c // bnb, :, :, c % bnb] for c in range(cout)], axis=-1)       # [rpb][4][cout]
            # per channel the rows partition the pixels and merge to the tensor's moments
            cnt = p[:, 3, :]
            assert not np.isnan(cnt).any() and np.all(cnt >= 0) and np.all(cnt.sum(0) == m)
            live = cnt > 0
            cs = np.where(live, cnt, 1.0)
            s1, s2, pv = (np.where(live, p[:, i, :], 0.0) for i in range(3))
            mean_k = pv + s1 / cs
            m2_k = s2 - s1 ** 2 / cs
            tot_mean = (cnt * mean_k).sum(0) / m
            tot_var = (np.where(live, m2_k + cnt * mean_k ** 2, 0.0)).sum(0) / m - tot_mean ** 2
            np.testing.assert_allclose(tot_mean, yq.mean(0), rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(tot_var, yq.var(0), rtol=2e-4)
        for kk in range(0 if counted else rows):                                 # every partial: shifted sums of its own rows
            blk = yq[kk * rpp.value:(kk + 1) * rpp.value]
            if len(blk) == 0:
                continue
            np.testing.assert_array_equal(p[kk, 2], blk[0])                     # the shift is the first stored row, exactly
            d = blk - blk[0]
            np.testing.assert_allclose(p[kk, 0], d.sum(0), rtol=1e-4, atol=1e-5 * np.abs(d).sum(0).max() + 1e-30)
            np.testing.assert_allclose(p[kk, 1], (d ** 2).sum(0), rtol=1e-4, atol=1e-30)
        out = torch.full(y_plain.shape, float('nan'), dtype=u.TDT[dtype], device=u.DEV)
        gd, bd = u.dev(gamma), u.dev(beta)
        sm, si, bm, bv = [torch.zeros(cout, dtype=torch.float32, device=u.DEV) for _ in range(4)]
        rm, rv = torch.zeros(cout, dtype=torch.float32, device=u.DEV), torch.ones(cout, dtype=torch.float32, device=u.DEV)
        bws = u.workspace(lib.mcn_bn_workspace_bytes(m, cout))
        _ffi.check(lib.mcn_bn_fwd_train_fused(y.data_ptr(), part.data_ptr(), rows, rpp.value, gd.data_ptr(), bd.data_ptr(), 0, out.data_ptr(), 0, sm.data_ptr(), si.data_ptr(),
                                              bm.data_ptr(), bv.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.99, m, cout, 1e-3, 1, u.MDT[dtype], bws.data_ptr(),
                                              bws.numel() * 4, u.stream()))
        check(u.host(out), ref['y'], dtype, 'fused bn y (tile {})'.format(tile), rel=1e-5 if dtype == 'float32' else 4e-3)
        check(u.host(sm), ref['save_mean'], 'float32', 'save_mean', rel=1e-5, mx=1e-4)
        check(u.host(si), ref['save_invstd'], 'float32', 'save_invstd', rel=1e-5)
        check(u.host(rv), ref['running_var'], 'float32', 'running_var', rel=1e-5)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', [(2, 14, 14, 128, 256, 1, 1), (3, 9, 11, 72, 136, 3, 1), (2, 16, 16, 64, 128, 3, 2), (2, 28, 28, 64, 64, 3, 1)])
def test_conv_tile_candidates_agree(case, dtype):
    """mcn_conv_geom.tile (the autotuning hook): every tile candidate of forward and dgrad — incl. the 8-wave 256x128
    bf16 tile — computes the same convolution (same products, same K order per output: bit-identical results)."""
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    n, h, w_, cin, cout, k, s = case
    x = RNG.standard_normal((n, h, w_, cin)).astype(np.float32)
    w = (RNG.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    y0 = u.conv_fwd(x, w, s, 'SAME', 1, dtype)
    dy = RNG.standard_normal(y0.shape).astype(np.float32)
    dx0 = u.conv_dgrad(dy, w, x.shape, s, 'SAME', 1, dtype)
    check(y0, O.conv2d_fwd(q(x, dtype), q(w, dtype), s, 'SAME'), dtype, 'fwd')
    names = set()
    for tile in range(1, lib.mcn_conv2d_tile_candidates(_ffi.CONV_FWD) + 1):
        g = u.geom(x.shape, w.shape, s, 'SAME')
        g.tile = tile
        buf = ctypes.create_string_buffer(128)
        lib.mcn_conv2d_kernel_name(_ffi.CONV_FWD, ctypes.byref(g), u.MDT[dtype], buf, 128)
        names.add(buf.value.decode())
        xd, wd, dyd = u.dev(x, dtype), u.dev(w), u.dev(dy, dtype)
        y = torch.full(y0.shape, float('nan'), dtype=u.TDT[dtype], device=u.DEV)
        dx = torch.full(x.shape, float('nan'), dtype=u.TDT[dtype], device=u.DEV)
        ws = u.workspace(max(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_FWD, ctypes.byref(g), u.MDT[dtype]),
                             lib.mcn_conv2d_workspace_bytes(_ffi.CONV_DGRAD, ctypes.byref(g), u.MDT[dtype])))
        _ffi.check(lib.mcn_conv2d_fwd(xd.data_ptr(), wd.data_ptr(), 0, 0, y.data_ptr(), ctypes.byref(g), u.MDT[dtype], _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, u.stream()))
        np.testing.assert_array_equal(u.host(y), y0, err_msg='fwd tile {}'.format(tile))
        _ffi.check(lib.mcn_conv2d_dgrad(dyd.data_ptr(), wd.data_ptr(), 0, dx.data_ptr(), ctypes.byref(g), 0, u.MDT[dtype], _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, u.stream()))
        np.testing.assert_array_equal(u.host(dx), dx0, err_msg='dgrad tile {}'.format(tile))
    if dtype != 'float32' and cout > 64:
        assert any(', 8, 0>' in nm for nm in names), names             # the 8-wave tile was exercised


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', [(8, 28, 28, 256, 1024, 1, 1),      # fwd splits (64x64 / 128x64 tiles)
                                  (64, 14, 14, 1024, 256, 1, 1),     # dgrad splits
                                  (96, 7, 7, 512, 2048, 1, 1),       # 128x128 (and the 8-wave bf16 tile) split
                                  (128, 14, 14, 256, 256, 3, 1),     # 3x3: taps inside the K-slices
                                  (200, 15, 15, 512, 512, 3, 2)])    # stride 2: dgrad parity classes, ragged M tail
def test_conv_streamk_tail(case, dtype):
    """fp32 layers whose tile count leaves the last round of workgroups mostly empty run that round K-sliced (partials in the
    workspace, fixed-order reduce): same convolution as the unsplit path up to fp32 summation order, deterministic, also
    with the BN-statistics epilogue and the masked residual add; too small a workspace falls back to the unsplit path."""
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    n, h, w_, cin, cout, k, s = case
    x = RNG.standard_normal((n, h, w_, cin)).astype(np.float32)
    w = (RNG.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    g0 = u.geom(x.shape, w.shape, s, 'SAME')
    oh, ow = -(-h // s), -(-w_ // s)
    dy = RNG.standard_normal((n, oh, ow, cout)).astype(np.float32)
    xd, wd, dyd = u.dev(x, dtype), u.dev(w), u.dev(dy, dtype)
    tol = dict(rel=2e-6, mx=2e-5) if dtype == 'float32' else dict(rel=1.5e-3, mx=1e-2)   # bf16: rare one-ulp rounding flips
    nsplit = 0

    def run_fwd(g, ws, stats=None):
        y = torch.full((n, oh, ow, cout), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
        if stats is None:
            _ffi.check(lib.mcn_conv2d_fwd(xd.data_ptr(), wd.data_ptr(), 0, 0, y.data_ptr(), ctypes.byref(g), u.MDT[dtype], _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, u.stream()))
        else:
            _ffi.check(lib.mcn_conv2d_fwd_bnstats(xd.data_ptr(), wd.data_ptr(), 0, 0, y.data_ptr(), stats.data_ptr(), ctypes.byref(g), u.MDT[dtype], _ffi.NHWC,
                                                  ws.data_ptr(), ws.numel() * 4, u.stream()))
        return u.host(y)

    def run_dgrad(g, ws):
        dx = torch.full(x.shape, float('nan'), dtype=u.TDT[dtype], device=u.DEV)
        _ffi.check(lib.mcn_conv2d_dgrad(dyd.data_ptr(), wd.data_ptr(), 0, dx.data_ptr(), ctypes.byref(g), 0, u.MDT[dtype], _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, u.stream()))
        return u.host(dx)

    for tile in range(lib.mcn_conv2d_tile_candidates(_ffi.CONV_FWD) + 1):
        g = u.geom(x.shape, w.shape, s, 'SAME')
        g.tile = tile | 0x200                                                            # MCN_TILE_NOWINO: this test is about the direct kernels' tail
        gn = u.geom(x.shape, w.shape, s, 'SAME')
        gn.tile = tile | 0x100 | 0x200                                           # MCN_TILE_NOSPLIT
        assert lib.mcn_conv2d_kslices(_ffi.CONV_FWD, ctypes.byref(gn), u.MDT[dtype]) == 1
        ws = u.workspace(max(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_FWD, ctypes.byref(g), u.MDT[dtype]),
                             lib.mcn_conv2d_workspace_bytes(_ffi.CONV_DGRAD, ctypes.byref(g), u.MDT[dtype])))
        if lib.mcn_conv2d_kslices(_ffi.CONV_FWD, ctypes.byref(g), u.MDT[dtype]) > 1:
            nsplit += 1
            y_ref = run_fwd(gn, ws)
            y = run_fwd(g, ws)
            check(y, y_ref, dtype, 'fwd split vs unsplit (tile {})'.format(tile), **tol)
            assert not np.array_equal(y, y_ref) or dtype != 'float32'            # the split path really ran
            np.testing.assert_array_equal(run_fwd(g, ws), y)                       # deterministic
            # workspace without room for the partials: the unsplit path, bit for bit
            pack = lib.mcn_conv2d_packed_bytes(_ffi.CONV_FWD, ctypes.byref(g), u.MDT[dtype])
            small = u.workspace(pack + 1024)
            np.testing.assert_array_equal(run_fwd(g, small), y_ref)
            # BN statistics ride the reduce pass: partial rows are indexed by absolute M tile
            rpp = ctypes.c_int32(0)
            rows = lib.mcn_conv2d_bnstats_rows(ctypes.byref(g), u.MDT[dtype], ctypes.byref(rpp))
            part = torch.full((rows, 3, cout), float('nan'), dtype=torch.float32, device=u.DEV)
            ys = run_fwd(g, ws, part)
            np.testing.assert_array_equal(ys, y)
            pr = u.host(part).astype(np.float64)
            yq = y.astype(np.float64).reshape(-1, cout)
            assert np.isfinite(pr[:-(-yq.shape[0] // rpp.value)]).all()
            for kk in (0, rows // 2, -(-yq.shape[0] // rpp.value) - 1):
                blk = yq[kk * rpp.value:(kk + 1) * rpp.value]
                np.testing.assert_array_equal(pr[kk, 2], blk[0])
                np.testing.assert_allclose(pr[kk, 0], (blk - blk[0]).sum(0), rtol=1e-4, atol=1e-5 * np.abs(blk - blk[0]).sum(0).max() + 1e-30)
        if lib.mcn_conv2d_kslices(_ffi.CONV_DGRAD, ctypes.byref(g), u.MDT[dtype]) > 1:
            nsplit += 1
            dx_ref = run_dgrad(gn, ws)
            dx = run_dgrad(g, ws)
            check(dx, dx_ref, dtype, 'dgrad split vs unsplit (tile {})'.format(tile), **tol)
            np.testing.assert_array_equal(run_dgrad(g, ws), dx)
            if lib.mcn_conv2d_dgrad_addmasked_ok(ctypes.byref(g), u.MDT[dtype]):
                src = RNG.standard_normal(x.shape).astype(np.float32)
                mb = lib.mcn_bn_relu_mask_bytes(n * h * w_, cin, u.MDT[dtype])
                mask = torch.from_numpy(RNG.integers(0, 256, mb, dtype=np.uint8)).to(u.DEV)
                sd = u.dev(src, dtype)
                outs = []
                for gg in (gn, g):
                    dxa = torch.full(x.shape, float('nan'), dtype=u.TDT[dtype], device=u.DEV)
                    _ffi.check(lib.mcn_conv2d_dgrad_addmasked(dyd.data_ptr(), wd.data_ptr(), 0, dxa.data_ptr(), sd.data_ptr(), mask.data_ptr(), ctypes.byref(gg),
                                                              u.MDT[dtype], _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, u.stream()))
                    outs.append(u.host(dxa))
                check(outs[1], outs[0], dtype, 'dgrad+masked add split vs unsplit (tile {})'.format(tile), **tol)
    # fp32 only: the bf16 layers are not MFMA bound and the split cost 4 % of the ResNet-50 step (conv.hip, sk_plan)
    assert nsplit > 0 if dtype == 'float32' else nsplit == 0, 'stream-K tail: {} split launches'.format(nsplit)
    # the unsplit path against the oracle at this size (1x1 cases: the oracle is one matmul)
    if k == 1:
        g = u.geom(x.shape, w.shape, s, 'SAME')
        ws = u.workspace(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_FWD, ctypes.byref(g), u.MDT[dtype]))
        check(run_fwd(g, ws), O.conv2d_fwd(q(x, dtype), q(w, dtype), s, 'SAME'), dtype, 'fwd (default plan) vs oracle')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', [(4, 14, 14, 64, 32, 1), (3, 9, 11, 72, 136, 3), (2, 28, 28, 256, 64, 1)])
def test_conv_dgrad_with_masked_residual_add(case, dtype):
    """dx = dgrad(dy) + add_src * [add_mask bit]: the fan-in of an identity shortcut fused into conv_0's dgrad epilogue must
    equal (bit for bit) the two-step path it replaces: dskip = masked dy_block written first, dgrad accumulated on top."""
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    n, h, w_, cin, cout, k = case
    wgt = (RNG.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    dy = RNG.standard_normal((n, h, w_, cout)).astype(np.float32)
    dyb = RNG.standard_normal((n, h, w_, cin)).astype(np.float32)            # gradient of the block output
    yb = RNG.standard_normal((n, h, w_, cin)).astype(np.float32)             # block output (pre-ReLU sign decides the mask)
    vec = 4 if dtype == 'float32' else 8
    bits = (yb.reshape(-1, vec) > 0).astype(np.uint32)
    mask = (bits << np.arange(vec, dtype=np.uint32)).sum(-1).astype(np.uint8)
    masked = np.where(yb > 0, q(dyb, dtype), 0.0)
    ref = u.conv_dgrad(dy, wgt, (n, h, w_, cin), 1, 'SAME', 1, dtype, accumulate_into=masked.astype(np.float32))
    g = u.geom((n, h, w_, cin), wgt.shape, 1, 'SAME')
    assert lib.mcn_conv2d_dgrad_addmasked_ok(ctypes.byref(g), u.MDT[dtype]) == 1
    dyd, wd, src, md = u.dev(dy, dtype), u.dev(wgt), u.dev(dyb, dtype), torch.as_tensor(mask).to(u.DEV)
    dx = torch.full((n, h, w_, cin), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    ws = u.workspace(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_DGRAD, ctypes.byref(g), u.MDT[dtype]))
    _ffi.check(lib.mcn_conv2d_dgrad_addmasked(dyd.data_ptr(), wd.data_ptr(), 0, dx.data_ptr(), src.data_ptr(), md.data_ptr(), ctypes.byref(g), u.MDT[dtype],
                                              _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, u.stream()))
    np.testing.assert_array_equal(u.host(dx), ref)
    g2 = u.geom((n, 2 * h, 2 * w_, cin), wgt.shape, 2, 'SAME')
    assert lib.mcn_conv2d_dgrad_addmasked_ok(ctypes.byref(g2), u.MDT[dtype]) == 0      # strided: not eligible


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', [(4, 14, 14, 64, 32, 1), (3, 9, 11, 72, 136, 3), (2, 28, 28, 256, 64, 1), (8, 16, 16, 8, 8, 1), (40, 28, 28, 512, 128, 1)])
def test_conv_dgrad_masked_residual_add_with_bn_backward_sums(case, dtype):
    """mcn_conv2d_dgrad_addmasked_bnred (round 4): ONE launch gives dx = dgrad(dy) + add_src * [add_mask bit] — bit for bit
    mcn_conv2d_dgrad_addmasked — AND the partial rows of the BN-backward sums over that dx (dy' = dx where the unit's OWN ReLU bit is set,
    x = the input of the unit's output BN): column sums against NumPy, then mcn_bn_bwd_from_partials against mcn_bn_bwd."""
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    n, h, w_, cin, cout, k = case
    md = u.MDT[dtype]
    wgt = (RNG.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    dy = RNG.standard_normal((n, h, w_, cout)).astype(np.float32)
    dyn = RNG.standard_normal((n, h, w_, cin)).astype(np.float32)            # gradient of the NEXT unit's output ...
    ynext = RNG.standard_normal((n, h, w_, cin)).astype(np.float32)          # ... and its sign pattern (add_mask)
    vec = 4 if dtype == 'float32' else 8
    nbits = (ynext.reshape(-1, vec) > 0).astype(np.uint32)
    add_mask = (nbits << np.arange(vec, dtype=np.uint32)).sum(-1).astype(np.uint8)
    # this unit's output BN: y_b = relu(bn(xbn) + skip), its byte mask from the forward kernel itself
    xbn = (1.2 * RNG.standard_normal((n, h, w_, cin)) + 0.2).astype(np.float32)
    skip = RNG.standard_normal((n, h, w_, cin)).astype(np.float32)
    gamma = (0.5 + RNG.random(cin)).astype(np.float32)
    beta = (0.3 * RNG.standard_normal(cin)).astype(np.float32)
    fw = u.bn_fwd_train(xbn, gamma, beta, 1e-3, dtype, skip=skip, act=1, want_mask=True)
    g = u.geom((n, h, w_, cin), wgt.shape, 1, 'SAME')
    assert lib.mcn_conv2d_dgrad_addmasked_ok(ctypes.byref(g), md) == 1
    rows = lib.mcn_conv2d_dgrad_bnred_rows(ctypes.byref(g), md)
    assert rows > 0
    dyd, wd, src, amd = u.dev(dy, dtype), u.dev(wgt), u.dev(dyn, dtype), torch.as_tensor(add_mask).to(u.DEV)
    xd, mk = u.dev(xbn, dtype), torch.as_tensor(fw['relu_mask']).to(u.DEV)
    ws = u.workspace(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_DGRAD, ctypes.byref(g), md))
    ref = torch.full((n, h, w_, cin), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    _ffi.check(lib.mcn_conv2d_dgrad_addmasked(dyd.data_ptr(), wd.data_ptr(), 0, ref.data_ptr(), src.data_ptr(), amd.data_ptr(), ctypes.byref(g), md, _ffi.NHWC,
                                              ws.data_ptr(), ws.numel() * 4, u.stream()))
    dx = torch.full((n, h, w_, cin), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    part = torch.full((rows, 2, cin), float('nan'), dtype=torch.float32, device=u.DEV)
    _ffi.check(lib.mcn_conv2d_dgrad_addmasked_bnred(dyd.data_ptr(), wd.data_ptr(), 0, dx.data_ptr(), src.data_ptr(), amd.data_ptr(), xd.data_ptr(), mk.data_ptr(),
                                                    part.data_ptr(), ctypes.byref(g), md, _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, u.stream()))
    dx_ref = u.host(ref)
    np.testing.assert_array_equal(u.host(dx), dx_ref)
    p = u.host(part).astype(np.float64)
    assert not np.isnan(p).any()
    # (the unit's OWN mask bytes, not y > 0: a positive pre-activation below the storage type's smallest subnormal stores 0 with its bit set)
    bits = ((np.asarray(fw['relu_mask']).reshape(-1, 1) >> np.arange(vec)) & 1).reshape(-1, cin).astype(np.float64)
    assert np.abs(bits - (fw['y'].reshape(-1, cin) > 0)).sum() <= 2
    dxm = dx_ref.reshape(-1, cin).astype(np.float64) * bits
    xq = q(xbn, dtype).reshape(-1, cin)
    np.testing.assert_allclose(p[:, 0].sum(0), dxm.sum(0), rtol=2e-5, atol=2e-5 * np.abs(dxm).sum(0).max())
    np.testing.assert_allclose(p[:, 1].sum(0), (dxm * xq).sum(0), rtol=2e-5, atol=2e-5 * np.abs(dxm * xq).sum(0).max())
    m = n * h * w_
    gd, bd, sm, si = u.dev(gamma), u.dev(beta), u.dev(fw['save_mean']), u.dev(fw['save_invstd'])
    bws = u.workspace(lib.mcn_bn_workspace_bytes(m, cin))

    def run(fn):
        o = torch.full(xbn.shape, float('nan'), dtype=u.TDT[dtype], device=u.DEV)
        dg, db = torch.zeros(cin, device=u.DEV), torch.zeros(cin, device=u.DEV)
        fn(o, dg, db)
        return u.host(o), u.host(dg), u.host(db)
    a = run(lambda o, dg, db: _ffi.check(lib.mcn_bn_bwd(dx.data_ptr(), xd.data_ptr(), 0, mk.data_ptr(), gd.data_ptr(), bd.data_ptr(), sm.data_ptr(), si.data_ptr(), o.data_ptr(), 0,
                                                        dg.data_ptr(), db.data_ptr(), 0.5, m, cin, 1, md, bws.data_ptr(), bws.numel() * 4, u.stream())))
    b = run(lambda o, dg, db: _ffi.check(lib.mcn_bn_bwd_from_partials(dx.data_ptr(), xd.data_ptr(), mk.data_ptr(), gd.data_ptr(), bd.data_ptr(), sm.data_ptr(), si.data_ptr(),
                                                                      part.data_ptr(), rows, o.data_ptr(), dg.data_ptr(), db.data_ptr(), 0.5, m, cin, md,
                                                                      bws.data_ptr(), bws.numel() * 4, u.stream())))
    check(b[0], a[0], dtype, 'dx from partials', rel=2e-5 if dtype == 'float32' else 4e-3)
    check(b[1], a[1], 'float32', 'dgamma', rel=1e-4)
    check(b[2], a[2], 'float32', 'dbeta', rel=1e-4)


@pytest.mark.parametrize('dtype', ['bfloat16', 'float16'])
@pytest.mark.parametrize('case', [(2, 32, 32, 3, 64, 7, 'SAME'),      # the ResNet stem: TF SAME pads (2,3), even left pad
                                  (3, 18, 20, 3, 32, 3, 'SAME'),      # the EfficientNet stem: pads (0,1)
                                  (2, 16, 16, 4, 16, 5, 'SAME'),      # 4 input channels, pads (1,2): odd left pad
                                  (1, 12, 14, 1, 8, 7, 'VALID'),      # no padding, 1 channel
                                  (2, 9, 8, 2, 24, 3, 'SAME')])       # odd H, pads (1,0)... W even is what the form needs
def test_conv_pixel_pair_form(case, dtype):
    """Pixel-pair form of a stride-2 conv on <= 4 input channels (mcn_conv2d_pair_geom / _pair_weights / _pair_wgrad_fold): the
    image stored 4 channels per pixel, run through the ORDINARY conv entry points with the paired geometry and filter, must give
    the plain convolution and (after the fold) its weight gradient."""
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    n, h, w_, cin, cout, k, pad = case
    x = RNG.standard_normal((n, h, w_, cin)).astype(np.float32)
    w = (RNG.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    xq, wq = q(x, dtype), q(w, dtype)
    y_ref = O.conv2d_fwd(xq, wq, 2, pad)
    g = u.geom(x.shape, w.shape, 2, pad, 1, 4)
    pg = _ffi.ConvGeom()
    assert lib.mcn_conv2d_pair_geom(ctypes.byref(g), u.MDT[dtype], ctypes.byref(pg)) == 1
    assert (pg.W, pg.Cin, pg.SW, pg.x_cs, pg.SH, pg.KH) == (w_ // 2, 8, 1, 8, 2, k) and pg.KW <= (k + 2) // 2 + 1
    assert lib.mcn_conv2d_pair_geom(ctypes.byref(g), _ffi.F32, ctypes.byref(_ffi.ConvGeom())) == 0          # 2-byte storage types only
    x4 = np.zeros((n, h, w_, 4), np.float32)
    x4[..., :cin] = x
    xd, wd = u.dev(x4, dtype), u.dev(w)
    wp = torch.full((pg.KH * pg.KW * 8 * cout,), float('nan'), dtype=torch.float32, device=u.DEV)
    _ffi.check(lib.mcn_conv2d_pair_weights(wd.data_ptr(), wp.data_ptr(), ctypes.byref(g), u.MDT[dtype], u.stream()))
    y = torch.full(y_ref.shape, float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    ws = u.workspace(max(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_FWD, ctypes.byref(pg), u.MDT[dtype]), lib.mcn_conv2d_workspace_bytes(_ffi.CONV_WGRAD, ctypes.byref(pg), u.MDT[dtype])))
    _ffi.check(lib.mcn_conv2d_fwd(xd.data_ptr(), wp.data_ptr(), 0, 0, y.data_ptr(), ctypes.byref(pg), u.MDT[dtype], _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, u.stream()))
    check(u.host(y), y_ref, dtype, 'paired fwd')
    # the same products as the 8-channel-padded form, summed in another order: equal up to one rounding of the stored type
    np.testing.assert_allclose(u.host(y), u.conv_fwd(x, w, 2, pad, 1, dtype), rtol=2e-2 if dtype == 'bfloat16' else 3e-3, atol=2e-2 if dtype == 'bfloat16' else 3e-3)
    dy = RNG.standard_normal(y_ref.shape).astype(np.float32)
    dyd = u.dev(dy, dtype)
    dwp = torch.full_like(wp, float('nan'))
    _ffi.check(lib.mcn_conv2d_wgrad(xd.data_ptr(), dyd.data_ptr(), dwp.data_ptr(), 0, ctypes.byref(pg), 0.5, u.MDT[dtype], _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, u.stream()))
    dw = torch.full(w.shape, float('nan'), dtype=torch.float32, device=u.DEV)
    _ffi.check(lib.mcn_conv2d_pair_wgrad_fold(dwp.data_ptr(), dw.data_ptr(), ctypes.byref(g), u.MDT[dtype], u.stream()))
    check(u.host(dw), 0.5 * O.conv2d_wgrad(xq, q(dy, dtype), w.shape, 2, pad), 'float32', 'paired wgrad', rel=2e-5)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', [(2, 16, 16, 64, 3, 2, 'SAME'), (3, 9, 11, 16, 3, 2, 'SAME'), (2, 12, 12, 8, 2, 2, 'VALID'), (1, 7, 7, 24, 3, 1, 'SAME')])
def test_bn_relu_maxpool_in_one_pass(case, dtype):
    """conv -> BN(train) -> ReLU -> max-pool with the statistics from the conv epilogue (the ResNet stem): the pass that normalises,
    rectifies and pools must give, bit for bit, the pooled values and the arg-max of mcn_bn_fwd_train_fused followed by
    mcn_maxpool_fwd — and the same saved / running statistics."""
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    n, h, w_, c, k, s, pad = case
    cin = 8
    x = (RNG.standard_normal((n, h, w_, cin)) + 0.5).astype(np.float32)
    wt = (RNG.standard_normal((1, 1, cin, c)) / np.sqrt(cin)).astype(np.float32)
    gamma = (0.5 + RNG.random(c)).astype(np.float32)
    beta = (0.3 * RNG.standard_normal(c)).astype(np.float32)
    g = u.geom(x.shape, wt.shape, 1, 'SAME')
    rpp = ctypes.c_int32(0)
    rows = lib.mcn_conv2d_bnstats_rows(ctypes.byref(g), u.MDT[dtype], ctypes.byref(rpp))
    assert rows > 0
    xd, wd, gd, bd = u.dev(x, dtype), u.dev(wt), u.dev(gamma), u.dev(beta)
    y = torch.full((n, h, w_, c), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    part = torch.full((rows, 4, c), float('nan'), dtype=torch.float32, device=u.DEV)
    ws = u.workspace(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_FWD, ctypes.byref(g), u.MDT[dtype]))
    _ffi.check(lib.mcn_conv2d_fwd_bnstats(xd.data_ptr(), wd.data_ptr(), 0, 0, y.data_ptr(), part.data_ptr(), ctypes.byref(g), u.MDT[dtype], _ffi.NHWC,
                                          ws.data_ptr(), ws.numel() * 4, u.stream()))
    m = n * h * w_
    pads = O.resolve_pads(h, w_, k, k, s, s, pad, 1, 1)
    oh, ow = O.out_size(h, k, s, pad, 1), O.out_size(w_, k, s, pad, 1)
    bws = u.workspace(lib.mcn_bn_workspace_bytes(m, c))

    def stats():
        return [torch.zeros(c, dtype=torch.float32, device=u.DEV) for _ in range(4)] + [torch.zeros(c, dtype=torch.float32, device=u.DEV), torch.ones(c, dtype=torch.float32, device=u.DEV)]
    # reference: the two separate calls
    s_ref = stats()
    yb = torch.full((n, h, w_, c), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    _ffi.check(lib.mcn_bn_fwd_train_fused(y.data_ptr(), part.data_ptr(), rows, rpp.value, gd.data_ptr(), bd.data_ptr(), 0, yb.data_ptr(), 0, *[t.data_ptr() for t in s_ref],
                                          0.99, m, c, 1e-3, 1, u.MDT[dtype], bws.data_ptr(), bws.numel() * 4, u.stream()))
    p_ref = torch.full((n, oh, ow, c), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    a_ref = torch.full((n, oh, ow, c), -7, dtype=torch.int8, device=u.DEV)
    _ffi.check(lib.mcn_maxpool_fwd(yb.data_ptr(), p_ref.data_ptr(), a_ref.data_ptr(), n, h, w_, c, k, k, s, s, pads[0], pads[2], oh, ow, u.MDT[dtype], u.stream()))
    # one pass
    s_one = stats()
    p_one = torch.full((n, oh, ow, c), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    a_one = torch.full((n, oh, ow, c), -7, dtype=torch.int8, device=u.DEV)
    _ffi.check(lib.mcn_bn_fwd_train_fused_maxpool(y.data_ptr(), part.data_ptr(), rows, rpp.value, gd.data_ptr(), bd.data_ptr(), p_one.data_ptr(), a_one.data_ptr(),
                                                  *[t.data_ptr() for t in s_one], 0.99, n, h, w_, c, 1e-3, k, k, s, s, pads[0], pads[2], oh, ow, u.MDT[dtype],
                                                  bws.data_ptr(), bws.numel() * 4, u.stream()))
    np.testing.assert_array_equal(u.host(p_one), u.host(p_ref))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(a_one.cpu().numpy(), a_ref.cpu().numpy())
    for a, b in zip(s_one, s_ref):
        np.testing.assert_array_equal(u.host(a), u.host(b))


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', [(2, 16, 16, 64, 'SAME'), (3, 9, 11, 16, 'SAME'), (2, 13, 12, 8, 'VALID'), (4, 32, 32, 32, 'SAME')])
def test_bn_backward_routes_maxpool_gradient(case, dtype):
    """mcn_bn_bwd_maxpool (BN + ReLU in front of a 3x3 / 2 max-pool: the pooled gradient is routed by arg-max inside the two BN
    passes) against mcn_maxpool_bwd followed by mcn_bn_bwd: dx, dgamma and dbeta bit for bit in the per-pixel form; the 2 x 2 block
    form (even maps without a leading pad: the stem) sums the pixels in another order — dgamma / dbeta to fp32 summation accuracy, dx
    to one rounding of the storage type around them."""
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    n, h, w_, c, pad = case
    x = (1.3 * RNG.standard_normal((n, h, w_, c)) + 0.2).astype(np.float32)
    gamma = (0.5 + RNG.random(c)).astype(np.float32)
    beta = (0.3 * RNG.standard_normal(c)).astype(np.float32)
    m = n * h * w_
    out = u.bn_fwd_train(x, gamma, beta, 1e-3, dtype, act=1)
    pads = O.resolve_pads(h, w_, 3, 3, 2, 2, pad, 1, 1)
    oh, ow = O.out_size(h, 3, 2, pad, 1), O.out_size(w_, 3, 2, pad, 1)
    xd, gd, bd = u.dev(x, dtype), u.dev(gamma), u.dev(beta)
    yd = u.dev(out['y'], dtype)
    sm, si = u.dev(out['save_mean']), u.dev(out['save_invstd'])
    pooled = torch.empty((n, oh, ow, c), dtype=u.TDT[dtype], device=u.DEV)
    arg = torch.empty((n, oh, ow, c), dtype=torch.int8, device=u.DEV)
    _ffi.check(lib.mcn_maxpool_fwd(yd.data_ptr(), pooled.data_ptr(), arg.data_ptr(), n, h, w_, c, 3, 3, 2, 2, pads[0], pads[2], oh, ow, u.MDT[dtype], u.stream()))
    dp = u.dev(RNG.standard_normal((n, oh, ow, c)).astype(np.float32), dtype)
    ws = u.workspace(lib.mcn_bn_workspace_bytes(m, c))
    # two steps
    dy = torch.full((n, h, w_, c), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    _ffi.check(lib.mcn_maxpool_bwd(dp.data_ptr(), arg.data_ptr(), dy.data_ptr(), n, h, w_, c, 3, 3, 2, 2, pads[0], pads[2], oh, ow, u.MDT[dtype], u.stream()))
    dx_ref = torch.full((n, h, w_, c), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    dg_ref, db_ref = torch.zeros(c, device=u.DEV), torch.zeros(c, device=u.DEV)
    _ffi.check(lib.mcn_bn_bwd(dy.data_ptr(), xd.data_ptr(), 0, 0, gd.data_ptr(), bd.data_ptr(), sm.data_ptr(), si.data_ptr(), dx_ref.data_ptr(), 0, dg_ref.data_ptr(), db_ref.data_ptr(),
                              0.5, m, c, 1, u.MDT[dtype], ws.data_ptr(), ws.numel() * 4, u.stream()))
    # routed
    dx = torch.full((n, h, w_, c), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    dg, db = torch.zeros(c, device=u.DEV), torch.zeros(c, device=u.DEV)
    _ffi.check(lib.mcn_bn_bwd_maxpool(dp.data_ptr(), arg.data_ptr(), xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), sm.data_ptr(), si.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                      0.5, n, h, w_, c, 3, 3, 2, 2, pads[0], pads[2], oh, ow, u.MDT[dtype], ws.data_ptr(), ws.numel() * 4, u.stream()))
    block_form = pads[0] == 0 and pads[2] == 0 and h % 2 == 0 and w_ % 2 == 0
    if not block_form:
        np.testing.assert_array_equal(u.host(dx), u.host(dx_ref))
        np.testing.assert_array_equal(u.host(dg), u.host(dg_ref))
        np.testing.assert_array_equal(u.host(db), u.host(db_ref))
    else:
        # (two orders of the same fp32 sums: the absolute noise follows the size of the sums — a channel whose total nearly cancels sits next
        # to channels of magnitude 60 — not the size of the result)
        np.testing.assert_allclose(u.host(dg), u.host(dg_ref), rtol=2e-5, atol=5e-7 * float(np.abs(u.host(dg_ref)).max()) + 1e-6)
        np.testing.assert_allclose(u.host(db), u.host(db_ref), rtol=2e-5, atol=5e-7 * float(np.abs(u.host(db_ref)).max()) + 1e-6)
        a, b = u.host(dx).astype(np.float64), u.host(dx_ref).astype(np.float64)
        ulp = {'float32': 2.0 ** -23, 'bfloat16': 2.0 ** -8, 'float16': 2.0 ** -11}[dtype]
        assert np.abs(a - b).max() <= 2 * ulp * np.abs(b).max() + 1e-7            # the coefficients moved by ~1e-7: at most a rounding step
        assert np.mean(a != b) <= (0.02 if dtype != 'float32' else 1.0)         # ... and in a 2-byte type for few elements
    assert np.abs(u.host(dx)).max() > 0


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(2, 8, 8, 64), (3, 7, 5, 24), (4, 14, 14, 256)])
def test_projection_shortcut_bn_folded_into_consumer(shape, dtype):
    """y = relu(bn(x) + bn_s(xs)) with the shortcut BN run statistics-only (mcn_bn_fwd_train_fused_stats) and applied inside the main
    BN's pass (mcn_bn_fwd_train_fused_affskip) against the two full calls: y, ReLU mask and all statistics bit for bit."""
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    n, h, w_, c = shape
    m = n * h * w_
    cin = 16
    xin = (RNG.standard_normal((n, h, w_, cin)) + 0.3).astype(np.float32)
    w1 = (RNG.standard_normal((1, 1, cin, c)) / 4).astype(np.float32)
    w2 = (RNG.standard_normal((1, 1, cin, c)) / 4 + 0.05).astype(np.float32)
    g = u.geom(xin.shape, w1.shape, 1, 'SAME')
    rpp = ctypes.c_int32(0)
    rows = lib.mcn_conv2d_bnstats_rows(ctypes.byref(g), u.MDT[dtype], ctypes.byref(rpp))
    assert rows > 0
    xd = u.dev(xin, dtype)
    ws = u.workspace(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_FWD, ctypes.byref(g), u.MDT[dtype]))

    def conv(wt):
        y = torch.full((n, h, w_, c), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
        part = torch.full((rows, 4, c), float('nan'), dtype=torch.float32, device=u.DEV)
        wd = u.dev(wt)
        _ffi.check(lib.mcn_conv2d_fwd_bnstats(xd.data_ptr(), wd.data_ptr(), 0, 0, y.data_ptr(), part.data_ptr(), ctypes.byref(g), u.MDT[dtype], _ffi.NHWC,
                                              ws.data_ptr(), ws.numel() * 4, u.stream()))
        return y, part
    x, px = conv(w1)
    xs, ps = conv(w2)
    gm, bm = u.dev((0.5 + RNG.random(c)).astype(np.float32)), u.dev((0.3 * RNG.standard_normal(c)).astype(np.float32))
    gs, bs = u.dev((0.5 + RNG.random(c)).astype(np.float32)), u.dev((0.3 * RNG.standard_normal(c)).astype(np.float32))
    bws = u.workspace(lib.mcn_bn_workspace_bytes(m, c))
    nb = max(int(lib.mcn_bn_relu_mask_bytes(m, c, u.MDT[dtype])), 1)

    def stats():
        return [torch.zeros(c, dtype=torch.float32, device=u.DEV) for _ in range(4)] + [torch.zeros(c, dtype=torch.float32, device=u.DEV), torch.ones(c, dtype=torch.float32, device=u.DEV)]
    # reference: shortcut BN to a tensor, then the main BN with that tensor as residual
    st_s, st_m = stats(), stats()
    s_t = torch.full((n, h, w_, c), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    _ffi.check(lib.mcn_bn_fwd_train_fused(xs.data_ptr(), ps.data_ptr(), rows, rpp.value, gs.data_ptr(), bs.data_ptr(), 0, s_t.data_ptr(), 0, *[t.data_ptr() for t in st_s],
                                          0.99, m, c, 1e-3, 0, u.MDT[dtype], bws.data_ptr(), bws.numel() * 4, u.stream()))
    y_ref = torch.full((n, h, w_, c), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    k_ref = torch.full((nb,), 0xAA, dtype=torch.uint8, device=u.DEV)
    _ffi.check(lib.mcn_bn_fwd_train_fused(x.data_ptr(), px.data_ptr(), rows, rpp.value, gm.data_ptr(), bm.data_ptr(), s_t.data_ptr(), y_ref.data_ptr(), k_ref.data_ptr(),
                                          *[t.data_ptr() for t in st_m], 0.99, m, c, 1e-3, 1, u.MDT[dtype], bws.data_ptr(), bws.numel() * 4, u.stream()))
    # folded
    st_s2, st_m2 = stats(), stats()
    aff = torch.full((2 * c,), float('nan'), dtype=torch.float32, device=u.DEV)
    _ffi.check(lib.mcn_bn_fwd_train_fused_stats(ps.data_ptr(), rows, rpp.value, gs.data_ptr(), bs.data_ptr(), *[t.data_ptr() for t in st_s2], 0.99, m, c, 1e-3, aff.data_ptr(),
                                                bws.data_ptr(), bws.numel() * 4, u.stream()))
    y = torch.full((n, h, w_, c), float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    k = torch.full((nb,), 0xAA, dtype=torch.uint8, device=u.DEV)
    _ffi.check(lib.mcn_bn_fwd_train_fused_affskip(x.data_ptr(), px.data_ptr(), rows, rpp.value, gm.data_ptr(), bm.data_ptr(), xs.data_ptr(), aff.data_ptr(), y.data_ptr(), k.data_ptr(),
                                                  *[t.data_ptr() for t in st_m2], 0.99, m, c, 1e-3, u.MDT[dtype], bws.data_ptr(), bws.numel() * 4, u.stream()))
    np.testing.assert_array_equal(u.host(y), u.host(y_ref))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(k.cpu().numpy(), k_ref.cpu().numpy())
    for a, b in zip(st_s2 + st_m2, st_s + st_m):
        np.testing.assert_array_equal(u.host(a), u.host(b))


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', [(2, 16, 16, 64, 256, 1, 1),        # bottleneck conv_2 (1x1): its dgrad produces BN1's dy
                                  (3, 14, 14, 64, 64, 3, 1),         # conv_1 (3x3): BN0's dy (fp32: the window kernel)
                                  (2, 16, 16, 128, 128, 3, 2),       # stride 2: four parity launches share the partial rows
                                  (2, 9, 11, 72, 40, 3, 1),          # ragged tiles
                                  (8, 16, 16, 8, 8, 3, 1),           # one 16-byte chunk of channels (the width-reduced test networks)
                                  (8, 16, 16, 8, 32, 1, 1),
                                  (40, 28, 28, 128, 128, 3, 1)])     # fp32: stream-K tail in the same launch
def test_dgrad_accumulates_bn_backward_sums(case, dtype):
    """mcn_conv2d_dgrad_bnred: dx as mcn_conv2d_dgrad, plus partial rows whose column sums are sum dy' and sum dy' * x (dy' = the stored dx
    where the forward's ReLU bit is set); mcn_bn_bwd_from_partials then gives the BN backward of mcn_bn_bwd (reduction order aside)."""
    from myconvnet_amd import _ffi
    u = _u()
    lib = _ffi.lib
    n, h, w_, cin, cout, k, s = case
    xbn = (1.2 * RNG.standard_normal((n, h, w_, cin)) + 0.2).astype(np.float32)      # the BN's input (= output of the conv in front)
    gamma = (0.5 + RNG.random(cin)).astype(np.float32)
    beta = (0.3 * RNG.standard_normal(cin)).astype(np.float32)
    fw = u.bn_fwd_train(xbn, gamma, beta, 1e-3, dtype, act=1, want_mask=True)
    wt = (RNG.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    g = u.geom(xbn.shape, wt.shape, s, 'SAME')
    oh, ow = O.out_size(h, k, s, 'SAME', 1), O.out_size(w_, k, s, 'SAME', 1)
    dy = RNG.standard_normal((n, oh, ow, cout)).astype(np.float32)
    rows = lib.mcn_conv2d_dgrad_bnred_rows(ctypes.byref(g), u.MDT[dtype])
    assert rows > 0
    dx_ref = u.conv_dgrad(dy, wt, xbn.shape, s, 'SAME', 1, dtype)
    dyd, wd, xd = u.dev(dy, dtype), u.dev(wt), u.dev(xbn, dtype)
    mk = torch.as_tensor(fw['relu_mask']).to(u.DEV)
    dx = torch.full(xbn.shape, float('nan'), dtype=u.TDT[dtype], device=u.DEV)
    part = torch.full((rows, 2, cin), float('nan'), dtype=torch.float32, device=u.DEV)
    ws = u.workspace(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_DGRAD, ctypes.byref(g), u.MDT[dtype]))
    _ffi.check(lib.mcn_conv2d_dgrad_bnred(dyd.data_ptr(), wd.data_ptr(), 0, dx.data_ptr(), xd.data_ptr(), mk.data_ptr(), part.data_ptr(), ctypes.byref(g), u.MDT[dtype], _ffi.NHWC,
                                          ws.data_ptr(), ws.numel() * 4, u.stream()))
    np.testing.assert_array_equal(u.host(dx), dx_ref)
    p = u.host(part).astype(np.float64)
    assert not np.isnan(p).any()
    ce = 4 if dtype == 'float32' else 8
    bits = np.unpackbits(fw['relu_mask'][:n * h * w_ * (cin // ce)].reshape(-1, cin // ce, 1), axis=2, bitorder='little')[:, :, :ce].reshape(-1, cin).astype(np.float64)
    np.testing.assert_array_equal(bits, (fw['y'].reshape(-1, cin) > 0).astype(np.float64))
    dxm = dx_ref.reshape(-1, cin).astype(np.float64) * bits
    xq = q(xbn, dtype).reshape(-1, cin)
    np.testing.assert_allclose(p[:, 0].sum(0), dxm.sum(0), rtol=2e-5, atol=2e-5 * np.abs(dxm).sum(0).max())
    np.testing.assert_allclose(p[:, 1].sum(0), (dxm * xq).sum(0), rtol=2e-5, atol=2e-5 * np.abs(dxm * xq).sum(0).max())
    # the BN backward from those rows against the three-pass kernel
    m = n * h * w_
    gd, bd, sm, si = u.dev(gamma), u.dev(beta), u.dev(fw['save_mean']), u.dev(fw['save_invstd'])
    bws = u.workspace(lib.mcn_bn_workspace_bytes(m, cin))

    def run(fn, *extra):
        o = torch.full(xbn.shape, float('nan'), dtype=u.TDT[dtype], device=u.DEV)
        dg, db = torch.zeros(cin, device=u.DEV), torch.zeros(cin, device=u.DEV)
        fn(o, dg, db)
        return u.host(o), u.host(dg), u.host(db)
    a = run(lambda o, dg, db: _ffi.check(lib.mcn_bn_bwd(dx.data_ptr(), xd.data_ptr(), 0, mk.data_ptr(), gd.data_ptr(), bd.data_ptr(), sm.data_ptr(), si.data_ptr(), o.data_ptr(), 0,
                                                        dg.data_ptr(), db.data_ptr(), 0.5, m, cin, 1, u.MDT[dtype], bws.data_ptr(), bws.numel() * 4, u.stream())))
    b = run(lambda o, dg, db: _ffi.check(lib.mcn_bn_bwd_from_partials(dx.data_ptr(), xd.data_ptr(), mk.data_ptr(), gd.data_ptr(), bd.data_ptr(), sm.data_ptr(), si.data_ptr(),
                                                                      part.data_ptr(), rows, o.data_ptr(), dg.data_ptr(), db.data_ptr(), 0.5, m, cin, u.MDT[dtype],
                                                                      bws.data_ptr(), bws.numel() * 4, u.stream())))
    check(b[0], a[0], dtype, 'dx from partials', rel=2e-5 if dtype == 'float32' else 4e-3)
    check(b[1], a[1], 'float32', 'dgamma', rel=1e-4)
    check(b[2], a[2], 'float32', 'dbeta', rel=1e-4)
