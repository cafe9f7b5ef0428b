"""The ORACLE at the bench's own sizes for the launches the training step REALLY makes.

tests/test_gpu_fullsize.py::test_conv_oracle_spot_parity_at_b256 calls the plain entry points; the step of BASELINE configs[1] / [2] launches the FUSED
ones with pre-packed filters: mcn_conv2d_fwd_bnstats (-> the persistent 1x1 kernel with carried statistics, the Winograd kernels with the statistics
epilogue and their K-sliced tails, the window ping-pong kernel of the 2-byte types), mcn_conv2d_dgrad_bnred and mcn_conv2d_dgrad_addmasked_bnred
(-> the dgrad epilogues that carry a BN's backward sums, with or without the masked residual fan-in).  Here each of them runs at B = 256 on every
geometry the step uses it on:
  * images 0, 127, 255 of the stored output against oracle.ops (a convolution is independent per image);
  * the statistics / BN-backward sums of the partial rows against float64 reductions of the stored tensor over the WHOLE batch;
  * the kernel symbol behind the call, derived exactly as bench.py derives it for its line (bench.conv_call_launches), against the symbol table of the
    step recorded below — the launch checked here is by name the launch the bench line reports.
Bars: the per-op bars of tests/test_gpu_ops.py (fp32 rel-L2 2e-5; bf16 8e-3: one rounding of the stored value).
"""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

B = 256
SPOT = (0, 127, 255)
from test_gpu_fullsize import R50_CONVS  # noqa: E402

# kernel symbols of the ResNet-50 B = 256 step as `bench.py --layers` prints them (gpurun_out/r5_layers, round 5): (entry point, H, Cin, Cout, k, s)
STEP_SYMBOLS = {
    'fp32': {
        ('fwd_bnstats', 56, 64, 256, 1, 1): 'conv_gemm_nt_pers<float, 64, 64, 4, 3>',
        ('fwd_bnstats', 56, 64, 64, 1, 1): 'conv_gemm_nt_pers<float, 64, 64, 4, 3>',
        ('fwd_bnstats', 56, 256, 64, 1, 1): 'conv_gemm_nt_pers<float, 64, 64, 4, 3>',
        ('fwd_bnstats', 56, 256, 128, 1, 1): 'conv_gemm_nt_pers<float, 64, 64, 4, 3>',
        ('fwd_bnstats', 28, 128, 512, 1, 1): 'conv_gemm_nt_pers<float, 64, 64, 4, 3>',
        ('fwd_bnstats', 28, 512, 128, 1, 1): 'conv_gemm_nt_pers<float, 64, 64, 4, 3>',
        ('fwd_bnstats', 28, 512, 256, 1, 1): 'conv_gemm_nt_pers<float, 64, 64, 4, 3>',
        ('fwd_bnstats', 14, 256, 1024, 1, 1): 'conv_gemm_nt_pers<float, 64, 64, 4, 3>',
        ('fwd_bnstats', 14, 1024, 512, 1, 1): 'conv_gemm_nt_pers<float, 64, 64, 4, 3>',
        ('fwd_bnstats', 7, 512, 2048, 1, 1): 'conv_gemm_nt_pers<float, 64, 64, 4, 3>',
        ('fwd_bnstats', 14, 1024, 256, 1, 1): 'conv_gemm_nt<float, 64, 64, 0, 4, 1>',          # (stream-K tail: not the persistent kernel)
        ('fwd_bnstats', 7, 2048, 512, 1, 1): 'conv_gemm_nt<float, 64, 64, 0, 4, 1>',
        ('fwd_bnstats', 56, 64, 64, 3, 1): 'conv_wino_f2k3_w8<0, 1>',
        ('fwd_bnstats', 28, 128, 128, 3, 1): 'conv_wino_f2k3_w8<0, 1>',
        ('fwd_bnstats', 14, 256, 256, 3, 1): 'conv_wino_f2k3_w8<0, 1>',
        ('fwd_bnstats', 7, 512, 512, 3, 1): 'conv_wino_f2k3_w8<0, 1>',
        ('fwd_bnstats', 224, 3, 64, 7, 2): 'conv_gemm_nt<float, 64, 64, 2, 4, 1>',
        ('fwd_bnstats', 56, 128, 128, 3, 2): 'conv_gemm_nt<float, 64, 64, 1, 4, 1>',
        ('fwd_bnstats', 56, 256, 512, 1, 2): 'conv_gemm_nt<float, 64, 64, 1, 4, 1>',
        ('fwd_bnstats', 28, 256, 256, 3, 2): 'conv_gemm_nt<float, 64, 64, 1, 4, 1>',
        ('fwd_bnstats', 28, 512, 1024, 1, 2): 'conv_gemm_nt<float, 64, 64, 1, 4, 1>',
        ('fwd_bnstats', 14, 512, 512, 3, 2): 'conv_gemm_nt<float, 64, 64, 1, 4, 1>',
        ('fwd_bnstats', 14, 1024, 2048, 1, 2): 'conv_gemm_nt<float, 64, 64, 1, 4, 1>',
        ('dgrad_bnred', 56, 64, 64, 3, 1): 'conv_wino_f2k3_w8<0, 4>',
        ('dgrad_bnred', 28, 128, 128, 3, 1): 'conv_wino_f2k3_w8<0, 4>',
        ('dgrad_bnred', 14, 256, 256, 3, 1): 'conv_wino_f2k3_w8<0, 4>',
        ('dgrad_bnred', 7, 512, 512, 3, 1): 'conv_wino_f2k3_w8<0, 4>',
        ('dgrad_bnred', 56, 64, 256, 1, 1): 'conv_gemm_nt<float, 64, 64, 0, 4, 4>',
        ('dgrad_bnred', 28, 128, 512, 1, 1): 'conv_gemm_nt<float, 64, 64, 0, 4, 4>',
        ('dgrad_bnred', 14, 256, 1024, 1, 1): 'conv_gemm_nt<float, 64, 64, 0, 4, 4>',
        ('dgrad_bnred', 7, 512, 2048, 1, 1): 'conv_gemm_nt<float, 64, 64, 0, 4, 4>',
        ('dgrad_bnred', 56, 128, 128, 3, 2): 'conv_gemm_nt<float, 64, 64, 1, 4, 4>',
        ('dgrad_bnred', 28, 256, 256, 3, 2): 'conv_gemm_nt<float, 64, 64, 1, 4, 4>',
        ('dgrad_bnred', 14, 512, 512, 3, 2): 'conv_gemm_nt<float, 64, 64, 1, 4, 4>',
        ('dgrad_addmasked_bnred', 56, 256, 64, 1, 1): 'conv_gemm_nt<float, 64, 64, 0, 4, 5>',
        ('dgrad_addmasked_bnred', 28, 512, 128, 1, 1): 'conv_gemm_nt<float, 64, 64, 0, 4, 5>',
        ('dgrad_addmasked_bnred', 14, 1024, 256, 1, 1): 'conv_gemm_nt<float, 64, 64, 0, 4, 5>',
        ('dgrad_addmasked_bnred', 7, 2048, 512, 1, 1): 'conv_gemm_nt<float, 64, 64, 0, 4, 5>',
    },
    'bf16': {
        ('fwd_bnstats', 56, 64, 256, 1, 1): 'conv_gemm_nt_pers<bf16, 128, 128, 4, 3>',
        ('fwd_bnstats', 28, 128, 512, 1, 1): 'conv_gemm_nt_pers<bf16, 128, 128, 4, 3>',
        ('fwd_bnstats', 14, 256, 1024, 1, 1): 'conv_gemm_nt_pers<bf16, 128, 128, 4, 3>',
        ('fwd_bnstats', 14, 1024, 256, 1, 1): 'conv_gemm_nt_pers<bf16, 128, 128, 4, 3>',
        ('fwd_bnstats', 28, 512, 128, 1, 1): 'conv_gemm_nt_pers<bf16, 128, 128, 4, 3>',
        ('fwd_bnstats', 56, 256, 128, 1, 1): 'conv_gemm_nt_pers<bf16, 128, 128, 4, 3>',
        ('fwd_bnstats', 7, 512, 2048, 1, 1): 'conv_gemm_nt_pers<bf16, 128, 128, 4, 3>',
        ('fwd_bnstats', 7, 2048, 512, 1, 1): 'conv_gemm_nt_pers<bf16, 128, 128, 4, 3>',
        ('fwd_bnstats', 28, 512, 256, 1, 1): 'conv_gemm_nt_pers<bf16, 128, 128, 4, 3>',
        ('fwd_bnstats', 14, 1024, 512, 1, 1): 'conv_gemm_nt_pers<bf16, 128, 128, 4, 3>',
        ('fwd_bnstats', 56, 256, 64, 1, 1): 'conv_gemm_nt<bf16, 128, 64, 0, 4, 1>',
        ('fwd_bnstats', 56, 64, 64, 1, 1): 'conv_gemm_nt<bf16, 128, 64, 0, 4, 1>',
        ('fwd_bnstats', 14, 256, 256, 3, 1): 'conv_gemm_nt_wpp<bf16, 128, 1>',
        ('fwd_bnstats', 7, 512, 512, 3, 1): 'conv_gemm_nt_wpp<bf16, 128, 1>',
        ('fwd_bnstats', 56, 64, 64, 3, 1): 'conv_gemm_nt<bf16, 128, 64, 1, 4, 1>',
        ('fwd_bnstats', 28, 128, 128, 3, 1): 'conv_gemm_nt<bf16, 128, 128, 1, 4, 1>',
        ('fwd_bnstats', 56, 128, 128, 3, 2): 'conv_gemm_nt<bf16, 128, 128, 1, 4, 1>',
        ('fwd_bnstats', 56, 256, 512, 1, 2): 'conv_gemm_nt<bf16, 128, 128, 1, 4, 1>',
        ('fwd_bnstats', 28, 256, 256, 3, 2): 'conv_gemm_nt<bf16, 128, 128, 1, 4, 1>',
        ('fwd_bnstats', 28, 512, 1024, 1, 2): 'conv_gemm_nt<bf16, 128, 128, 1, 4, 1>',
        ('fwd_bnstats', 14, 512, 512, 3, 2): 'conv_gemm_nt<bf16, 128, 128, 1, 4, 1>',
        ('fwd_bnstats', 14, 1024, 2048, 1, 2): 'conv_gemm_nt<bf16, 128, 128, 1, 4, 1>',
        ('dgrad_addmasked_bnred', 56, 256, 64, 1, 1): 'conv_gemm_nt<bf16, 128, 128, 0, 4, 5>',
        ('dgrad_addmasked_bnred', 28, 512, 128, 1, 1): 'conv_gemm_nt<bf16, 128, 128, 0, 4, 5>',
        ('dgrad_addmasked_bnred', 14, 1024, 256, 1, 1): 'conv_gemm_nt<bf16, 128, 128, 0, 4, 5>',
        ('dgrad_addmasked_bnred', 7, 2048, 512, 1, 1): 'conv_gemm_nt<bf16, 128, 128, 0, 4, 5>',
    },
}
BDT = {'float32': 'fp32', 'bfloat16': 'bf16'}


def _u():
    import abi_util
    return abi_util


def _symbol_of(name, args, dtype):
    """the symbol bench.py books this call under (same function, same argument layout as the launch lists)"""
    import bench
    return bench.conv_call_launches(name, args, BDT[dtype])


def _check_symbol(kind, layer, dtype, name, args):
    gm, launches, key, _, _ = _symbol_of(name, args, dtype)
    want = STEP_SYMBOLS[BDT[dtype]].get((kind,) + tuple(layer))
    if want is not None:
        assert key == want, 'the step launches {} for {} {}; this call would launch {}'.format(want, kind, layer, key)
    return key, launches


def _ids(l):
    return 'h{}_{}to{}_k{}s{}'.format(*l)


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
@pytest.mark.parametrize('layer', R50_CONVS, ids=_ids)
def test_fwd_bnstats_oracle_spot_parity_at_b256(layer, dtype):
    """mcn_conv2d_fwd_bnstats with the pre-packed filter (what fwd_conv emits for all 53 convs of the step): stored y of images {0, 127, 255} against
    oracle.ops.conv2d_fwd; mean / biased variance that mcn_bn_fwd_train_fused_stats forms from the partial rows against float64 moments of the
    stored y over all B x OH x OW pixels."""
    from myconvnet_amd import _ffi
    from oracle import ops as O
    from test_gpu_ops import check
    u = _u()
    lib = _ffi.lib
    h, cin, cout, k, s = layer
    md = u.MDT[dtype]
    td = u.TDT[dtype]
    gen = torch.Generator(device=u.DEV).manual_seed(h * 31 + cin * 7 + cout + k)
    ce = 4 if dtype == 'float32' else 8
    cs = cin if cin % ce == 0 else (cin + ce - 1) // ce * ce
    x = torch.zeros((B, h, h, cs), device=u.DEV, dtype=td)
    x[..., :cin] = (torch.randn((B, h, h, cin), device=u.DEV, generator=gen) + 0.5).to(td)          # (offset input: |mean| comparable to std in y)
    w = (torch.randn((k, k, cin, cout), device=u.DEV, generator=gen) / np.sqrt(k * k * cin)).float()
    g = u.geom((B, h, h, cin), (k, k, cin, cout), s, 'SAME', x_cs=cs if cs != cin else 0)
    oh = -(-h // s)
    m = B * oh * oh
    rpp = ctypes.c_int32(0)
    rows = int(lib.mcn_conv2d_bnstats_rows(ctypes.byref(g), md, ctypes.byref(rpp)))
    assert rows > 0, 'the step takes this BN\'s statistics from the conv epilogue'
    ws = u.workspace(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_FWD, ctypes.byref(g), md))
    wp, keep = u.prepack(w.cpu().numpy(), g, _ffi.CONV_FWD, dtype)
    wpp = wp.data_ptr() if wp is not None else 0
    y = torch.full((B, oh, oh, cout), float('nan'), device=u.DEV, dtype=td)
    part = torch.full((rows, 4, cout), float('nan'), device=u.DEV, dtype=torch.float32)
    args = [x.data_ptr(), w.data_ptr(), wpp, 0, y.data_ptr(), part.data_ptr(), ctypes.byref(g), md, _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, u.stream()]
    key, launches = _check_symbol('fwd_bnstats', layer, dtype, 'mcn_conv2d_fwd_bnstats', args)
    _ffi.check(lib.mcn_conv2d_fwd_bnstats(*args))
    assert torch.isfinite(y.float()).all()
    sel = list(SPOT)
    xs = x[sel][..., :cin].double().cpu().numpy()
    wq = w.to(td).double().cpu().numpy()
    check(y[sel].float().cpu().numpy(), O.conv2d_fwd(xs, wq, s, 'SAME', 1), dtype, 'fwd + statistics, images {} ({})'.format(SPOT, key))
    # statistics: finalize from the partial rows (no apply pass) against float64 moments of the STORED tensor
    sm, si, bm, bv = [torch.zeros(cout, dtype=torch.float32, device=u.DEV) for _ in range(4)]
    aff = torch.zeros(2 * cout, dtype=torch.float32, device=u.DEV)
    ones = torch.ones(cout, dtype=torch.float32, device=u.DEV)
    bws = u.workspace(lib.mcn_bn_workspace_bytes(m, cout))
    _ffi.check(lib.mcn_bn_fwd_train_fused_stats(part.data_ptr(), rows, rpp.value, ones.data_ptr(), 0, sm.data_ptr(), si.data_ptr(), bm.data_ptr(), bv.data_ptr(), 0, 0, 0.99,
                                                m, cout, 1e-3, aff.data_ptr(), bws.data_ptr(), bws.numel() * 4, u.stream()))
    yd = y.reshape(m, cout).double()
    mean = yd.mean(0)
    var = ((yd - mean) ** 2).mean(0)
    sd = torch.sqrt(var)
    assert float(((sm.double() - mean).abs() / (sd + mean.abs())).max()) <= 1e-6, (key, 'mean')
    assert float(((1.0 / si.double() ** 2 - 1e-3 - var).abs() / var).max()) <= 2e-5, (key, 'variance')
    assert float(((bv.double() - var * m / (m - 1)).abs() / var).max()) <= 2e-5, (key, 'unbiased variance')


BNRED_LAYERS = [(56, 64, 64, 3, 1), (56, 128, 128, 3, 2), (28, 128, 128, 3, 1), (28, 256, 256, 3, 2), (14, 256, 256, 3, 1), (14, 512, 512, 3, 2), (7, 512, 512, 3, 1),
                (56, 64, 256, 1, 1), (28, 128, 512, 1, 1), (14, 256, 1024, 1, 1), (7, 512, 2048, 1, 1)]


def _bn_with_mask(lib, u, _ffi, xbn, gamma, beta, skip, dtype):
    """training-mode BN + [residual] + ReLU through the product kernel: its stored output and the byte mask the backward epilogues read"""
    m, c = xbn.numel() // xbn.shape[-1], xbn.shape[-1]
    md = u.MDT[dtype]
    y = torch.empty_like(xbn)
    mask = torch.zeros(int(lib.mcn_bn_relu_mask_bytes(m, c, md)), dtype=torch.uint8, device=u.DEV)
    sm, si, bm, bv = [torch.zeros(c, dtype=torch.float32, device=u.DEV) for _ in range(4)]
    bws = u.workspace(lib.mcn_bn_workspace_bytes(m, c))
    _ffi.check(lib.mcn_bn_fwd_train(xbn.data_ptr(), gamma.data_ptr(), beta.data_ptr(), skip.data_ptr() if skip is not None else 0, y.data_ptr(), mask.data_ptr(), sm.data_ptr(),
                                    si.data_ptr(), bm.data_ptr(), bv.data_ptr(), 0, 0, 0.99, m, c, 1e-3, 1, md, bws.data_ptr(), bws.numel() * 4, u.stream()))
    return y, mask, sm, si, bws


def _mask_bits(mask, m, c, vec):
    """[m][c] 0/1 from the byte mask (one byte per 16-byte chunk, bit i = element i), on the device"""
    b = mask.reshape(m, c // vec, 1).to(torch.int32)
    sh = torch.arange(vec, device=mask.device, dtype=torch.int32).reshape(1, 1, vec)
    return ((b >> sh) & 1).reshape(m, c).double()


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
@pytest.mark.parametrize('layer', BNRED_LAYERS, ids=_ids)
def test_dgrad_bnred_oracle_spot_parity_at_b256(layer, dtype):
    """mcn_conv2d_dgrad_bnred on every geometry the step uses it on (the dgrad of conv_1 and conv_2 of all 16 bottlenecks: Winograd <0, 4> and its K-sliced
    tails in fp32, the window ping-pong / two-buffer kernels in bf16, four parity launches for stride 2, 25 088 partial rows + the fold stage at 56 x 56):
    dx of images {0, 127, 255} against oracle.ops.conv2d_dgrad, dx bit-identical to mcn_conv2d_dgrad, the column sums of the partial rows against
    float64 reductions of (stored dx) x (the BN's own ReLU bits) [x the BN's input], and mcn_bn_bwd_from_partials against mcn_bn_bwd."""
    from myconvnet_amd import _ffi
    from oracle import ops as O
    from test_gpu_ops import check
    u = _u()
    lib = _ffi.lib
    h, cin, cout, k, s = layer
    td, md = u.TDT[dtype], u.MDT[dtype]
    vec = 4 if dtype == 'float32' else 8
    gen = torch.Generator(device=u.DEV).manual_seed(11 + h + cin)
    m = B * h * h
    xbn = (torch.randn((B, h, h, cin), device=u.DEV, generator=gen) * 1.3 + 0.4).to(td)               # the BN's input
    gamma = (0.5 + torch.rand(cin, device=u.DEV, generator=gen)).float()
    beta = (0.3 * torch.randn(cin, device=u.DEV, generator=gen)).float()
    y, mask, sm, si, bws = _bn_with_mask(lib, u, _ffi, xbn, gamma, beta, None, dtype)
    g = u.geom((B, h, h, cin), (k, k, cin, cout), s, 'SAME')
    oh = (h + s - 1) // s
    w = (torch.randn((k, k, cin, cout), device=u.DEV, generator=gen) / np.sqrt(k * k * cin)).float()
    dy = torch.randn((B, oh, oh, cout), device=u.DEV, generator=gen).to(td)
    rows = int(lib.mcn_conv2d_dgrad_bnred_rows(ctypes.byref(g), md))
    assert rows > 0
    ws = u.workspace(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_DGRAD, ctypes.byref(g), md))
    wp, keep = u.prepack(w.cpu().numpy(), g, _ffi.CONV_DGRAD, dtype)
    wpp = wp.data_ptr() if wp is not None else 0
    dx0, dx1 = torch.empty_like(xbn), torch.full_like(xbn, float('nan'))
    part = torch.full((rows, 2, cin), float('nan'), dtype=torch.float32, device=u.DEV)
    st = u.stream()
    _ffi.check(lib.mcn_conv2d_dgrad(dy.data_ptr(), w.data_ptr(), wpp, dx0.data_ptr(), ctypes.byref(g), 0, md, _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, st))
    args = [dy.data_ptr(), w.data_ptr(), wpp, dx1.data_ptr(), xbn.data_ptr(), mask.data_ptr(), part.data_ptr(), ctypes.byref(g), md, _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, st]
    key, launches = _check_symbol('dgrad_bnred', layer, dtype, 'mcn_conv2d_dgrad_bnred', args)
    _ffi.check(lib.mcn_conv2d_dgrad_bnred(*args))
    assert torch.equal(dx0, dx1)
    assert not bool(torch.isnan(part).any())
    sel = list(SPOT)
    wq = w.to(td).double().cpu().numpy()
    check(dx1[sel].float().cpu().numpy(), O.conv2d_dgrad(dy[sel].double().cpu().numpy(), wq, (3, h, h, cin), s, 'SAME', 1), dtype, 'dgrad + BN sums, images {} ({})'.format(SPOT, key))
    on = _mask_bits(mask, m, cin, vec)
    assert float((on - (y.reshape(m, cin) > 0).double()).abs().sum()) <= 4             # (a positive pre-activation below the smallest subnormal stores 0 with its bit set)
    dxm = dx1.reshape(m, cin).double() * on
    xd = xbn.reshape(m, cin).double()
    s1, s2 = dxm.sum(0), (dxm * xd).sum(0)
    p = part.double().sum(0)
    assert float((p[0] - s1).abs().max()) <= 1e-5 * float(dxm.abs().sum(0).max())
    assert float((p[1] - s2).abs().max()) <= 1e-5 * float((dxm * xd).abs().sum(0).max())

    def run(fn):
        o = torch.full_like(xbn, float('nan'))
        dg, db = torch.zeros(cin, device=u.DEV), torch.zeros(cin, device=u.DEV)
        fn(o, dg, db)
        return o, dg, db
    a = run(lambda o, dg, db: _ffi.check(lib.mcn_bn_bwd(dx1.data_ptr(), xbn.data_ptr(), 0, mask.data_ptr(), gamma.data_ptr(), beta.data_ptr(), sm.data_ptr(), si.data_ptr(), o.data_ptr(), 0,
                                                        dg.data_ptr(), db.data_ptr(), 1.0, m, cin, 1, md, bws.data_ptr(), bws.numel() * 4, st)))
    b = run(lambda o, dg, db: _ffi.check(lib.mcn_bn_bwd_from_partials(dx1.data_ptr(), xbn.data_ptr(), mask.data_ptr(), gamma.data_ptr(), beta.data_ptr(), sm.data_ptr(), si.data_ptr(),
                                                                      part.data_ptr(), rows, o.data_ptr(), dg.data_ptr(), db.data_ptr(), 1.0, m, cin, md,
                                                                      bws.data_ptr(), bws.numel() * 4, st)))
    ref = float(a[0].double().abs().max())
    assert float((a[0].double() - b[0].double()).abs().max()) <= (1e-5 if dtype == 'float32' else 8e-3) * ref        # (bf16: one ulp of the largest element)
    for i in (1, 2):
        assert float((a[i].double() - b[i].double()).abs().max()) <= 2e-5 * float(dxm.abs().sum(0).max())


ACCRED_LAYERS = [(56, 256, 64, 1, 1), (28, 512, 128, 1, 1), (14, 1024, 256, 1, 1), (7, 2048, 512, 1, 1)]


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
@pytest.mark.parametrize('layer', ACCRED_LAYERS, ids=_ids)
def test_dgrad_addmasked_bnred_oracle_spot_parity_at_b256(layer, dtype):
    """mcn_conv2d_dgrad_addmasked_bnred (NT_EPI_ACCRED: the 12 conv_0 dgrads of the step that complete a unit output's gradient and carry that unit's
    output BN's backward sums) and mcn_conv2d_dgrad_addmasked: dx of images {0, 127, 255} = oracle dgrad + [next unit's ReLU bits] x next unit's
    gradient; both entry points bit-identical in dx; sums against float64 reductions of (stored dx) x (this unit's own ReLU bits) [x BN input]."""
    from myconvnet_amd import _ffi
    from oracle import ops as O
    from test_gpu_ops import check
    u = _u()
    lib = _ffi.lib
    h, cin, cout, k, s = layer
    td, md = u.TDT[dtype], u.MDT[dtype]
    vec = 4 if dtype == 'float32' else 8
    gen = torch.Generator(device=u.DEV).manual_seed(13 + h + cin)
    m = B * h * h
    # this unit's output BN: y_b = relu(bn(xbn) + skip) -> its byte mask; the next unit's output sign pattern -> add_mask
    xbn = (torch.randn((B, h, h, cin), device=u.DEV, generator=gen) * 1.2 + 0.2).to(td)
    skip = torch.randn((B, h, h, cin), device=u.DEV, generator=gen).to(td)
    gamma = (0.5 + torch.rand(cin, device=u.DEV, generator=gen)).float()
    beta = (0.3 * torch.randn(cin, device=u.DEV, generator=gen)).float()
    yb, mask, sm, si, bws = _bn_with_mask(lib, u, _ffi, xbn, gamma, beta, skip, dtype)
    del skip
    ynext = torch.randn((B, h, h, cin), device=u.DEV, generator=gen)
    bits = (ynext.reshape(-1, vec) > 0).to(torch.int32)
    add_mask = (bits << torch.arange(vec, device=u.DEV, dtype=torch.int32)).sum(-1).to(torch.uint8)
    del bits
    src = torch.randn((B, h, h, cin), device=u.DEV, generator=gen).to(td)                             # gradient of the next unit's output
    g = u.geom((B, h, h, cin), (k, k, cin, cout), s, 'SAME')
    assert lib.mcn_conv2d_dgrad_addmasked_ok(ctypes.byref(g), md) == 1
    rows = int(lib.mcn_conv2d_dgrad_bnred_rows(ctypes.byref(g), md))
    assert rows > 0
    w = (torch.randn((k, k, cin, cout), device=u.DEV, generator=gen) / np.sqrt(k * k * cin)).float()
    dy = torch.randn((B, h, h, cout), device=u.DEV, generator=gen).to(td)
    ws = u.workspace(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_DGRAD, ctypes.byref(g), md))
    wp, keep = u.prepack(w.cpu().numpy(), g, _ffi.CONV_DGRAD, dtype)
    wpp = wp.data_ptr() if wp is not None else 0
    st = u.stream()
    dx0, dx1 = torch.full_like(xbn, float('nan')), torch.full_like(xbn, float('nan'))
    part = torch.full((rows, 2, cin), float('nan'), dtype=torch.float32, device=u.DEV)
    a0 = [dy.data_ptr(), w.data_ptr(), wpp, dx0.data_ptr(), src.data_ptr(), add_mask.data_ptr(), ctypes.byref(g), md, _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, st]
    _ffi.check(lib.mcn_conv2d_dgrad_addmasked(*a0))
    args = [dy.data_ptr(), w.data_ptr(), wpp, dx1.data_ptr(), src.data_ptr(), add_mask.data_ptr(), xbn.data_ptr(), mask.data_ptr(), part.data_ptr(), ctypes.byref(g), md, _ffi.NHWC,
            ws.data_ptr(), ws.numel() * 4, st]
    key, launches = _check_symbol('dgrad_addmasked_bnred', layer, dtype, 'mcn_conv2d_dgrad_addmasked_bnred', args)
    _ffi.check(lib.mcn_conv2d_dgrad_addmasked_bnred(*args))
    assert torch.equal(dx0, dx1)
    assert not bool(torch.isnan(part).any())
    sel = list(SPOT)
    wq = w.to(td).double().cpu().numpy()
    ref = O.conv2d_dgrad(dy[sel].double().cpu().numpy(), wq, (3, h, h, cin), s, 'SAME', 1)
    ref = ref + np.where(ynext[sel].cpu().numpy() > 0, src[sel].double().cpu().numpy(), 0.0)
    check(dx1[sel].float().cpu().numpy(), ref, dtype, 'dgrad + masked fan-in + BN sums, images {} ({})'.format(SPOT, key))
    on = _mask_bits(mask, m, cin, vec)
    assert float((on - (yb.reshape(m, cin) > 0).double()).abs().sum()) <= 4
    dxm = dx1.reshape(m, cin).double() * on
    xd = xbn.reshape(m, cin).double()
    p = part.double().sum(0)
    assert float((p[0] - dxm.sum(0)).abs().max()) <= 1e-5 * float(dxm.abs().sum(0).max())
    assert float((p[1] - (dxm * xd).sum(0)).abs().max()) <= 1e-5 * float((dxm * xd).abs().sum(0).max())
