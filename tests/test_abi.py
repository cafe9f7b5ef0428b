"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/mcn.h declares
(no compute without a GPU), and argument validation returns status codes + messages instead of crashing."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'mcn.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(mcn_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_are_exported_and_bound():
    from myconvnet_amd import _ffi
    syms = declared_symbols()
    assert len(syms) >= 30
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), 'libmcn_hip.so does not export {}'.format(s)
    assert set(syms) == set(_ffi.SIGNATURES), set(syms) ^ set(_ffi.SIGNATURES)
    assert _ffi.lib.mcn_version() == 100


def test_header_compiles_as_c():
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, 't.c')
        open(c, 'w').write('#include "mcn.h"\nint main(void){ mcn_conv_geom g; (void)g; return MCN_OK; }\n')
        subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), '-c', c, '-o', os.path.join(d, 't.o')])


def test_argument_validation_without_gpu():
    from myconvnet_amd import _ffi
    lib = _ffi.lib
    g = _ffi.conv_geom(1, 8, 8, 0, 16, 3, 3, 1, 1, 1, 1, (1, 1, 1, 1))       # Cin = 0
    assert lib.mcn_conv2d_fwd(1, 1, 0, 0, 1, ctypes.byref(g), _ffi.F32, _ffi.NHWC, 0, 0, 0) == _ffi.E_BADARG
    assert 'bad geometry' in _ffi.last_error()
    g = _ffi.conv_geom(1, 8, 8, 16, 16, 3, 3, 1, 1, 1, 1, (1, 1, 1, 1))
    assert lib.mcn_conv2d_fwd(0, 0, 0, 0, 0, ctypes.byref(g), _ffi.F32, _ffi.NHWC, 0, 0, 0) == _ffi.E_BADARG
    assert lib.mcn_conv2d_fwd(1, 1, 0, 0, 1, ctypes.byref(g), _ffi.F32, _ffi.NCHW, 0, 0, 0) == _ffi.E_UNSUPPORTED
    assert lib.mcn_conv2d_fwd(1, 1, 0, 0, 1, ctypes.byref(g), 7, _ffi.NHWC, 0, 0, 0) == _ffi.E_UNSUPPORTED      # no such dtype
    assert 'dtype 7' in _ffi.last_error()
    assert lib.mcn_conv2d_fwd(1, 1, 0, 0, 1, ctypes.byref(g), _ffi.F16, _ffi.NHWC, 0, 0, 0) == _ffi.E_WORKSPACE        # fp16 is a storage type (f-4)
    assert lib.mcn_conv2d_kernel_name(_ffi.CONV_FWD, ctypes.byref(g), _ffi.F16, ctypes.create_string_buffer(96), 96) == 1
    assert lib.mcn_conv2d_fwd(1, 1, 0, 0, 1, ctypes.byref(g), _ffi.F32, _ffi.NHWC, 0, 0, 0) == _ffi.E_WORKSPACE
    assert lib.mcn_conv2d_packed_bytes(_ffi.CONV_FWD, ctypes.byref(g), _ffi.F32) >= 16 * 9 * 16 * 4
    buf = ctypes.create_string_buffer(96)
    assert lib.mcn_conv2d_kernel_name(_ffi.CONV_FWD, ctypes.byref(g), _ffi.BF16, buf, 96) == 1 and buf.value.startswith(b'conv_gemm_nt<bf16')
    g2 = _ffi.conv_geom(4, 16, 16, 16, 16, 3, 3, 2, 2, 1, 1, (0, 1, 0, 1))
    assert lib.mcn_conv2d_kernel_name(_ffi.CONV_DGRAD, ctypes.byref(g2), _ffi.F32, buf, 96) == 4        # one launch per stride-parity class
    assert lib.mcn_conv2d_kernel_name(_ffi.CONV_WGRAD, ctypes.byref(g2), _ffi.F32, buf, 96) == 1 and buf.value.startswith(b'conv_gemm_tn<float')
    assert lib.mcn_conv2d_workspace_bytes(_ffi.CONV_FWD, ctypes.byref(g), _ffi.F32) >= 16 * 9 * 16 * 4
    # per-launch list: the stride-parity classes of a strided dgrad differ in taps (and may differ in kernel symbol)
    lbuf = ctypes.create_string_buffer(512)
    assert lib.mcn_conv2d_launch_list(_ffi.CONV_DGRAD, ctypes.byref(g2), _ffi.F32, lbuf, 512) == 4
    lines = lbuf.value.decode().splitlines()
    assert len(lines) == 4 and sorted(int(ln.rsplit(':', 1)[1]) for ln in lines) == [1, 2, 2, 4] and all(ln.startswith(('conv_gemm_nt<float', 'conv_gemm_nt_pers<float')) for ln in lines)
    assert lib.mcn_conv2d_launch_list(_ffi.CONV_FWD, ctypes.byref(g2), _ffi.F32, lbuf, 512) == 1 and lbuf.value.decode().endswith(':9\n')
    assert lib.mcn_conv2d_launch_list(_ffi.CONV_FWD, ctypes.byref(g2), _ffi.F32, lbuf, 8) == _ffi.E_BADARG
    # stream-K plan: small layers run unsplit; MCN_TILE_NOSPLIT turns the split off for a layer that has one
    assert lib.mcn_conv2d_kslices(_ffi.CONV_FWD, ctypes.byref(g), _ffi.F32) == 1
    gb = _ffi.conv_geom(256, 7, 7, 512, 512, 3, 3, 1, 1, 1, 1, (1, 1, 1, 1))
    assert lib.mcn_conv2d_kslices(_ffi.CONV_FWD, ctypes.byref(gb), _ffi.F32) == 1                     # fp32 3x3 / stride 1: the Winograd kernel (its own tail plan)
    assert lib.mcn_conv2d_kernel_name(_ffi.CONV_FWD, ctypes.byref(gb), _ffi.F32, buf, 96) == 1 and buf.value.startswith(b'conv_wino_f2k3_w8<')
    assert lib.mcn_conv2d_kernel_name(_ffi.CONV_WGRAD, ctypes.byref(gb), _ffi.F32, buf, 96) == 1 and buf.value == b'conv_wino_wgrad_f3k2_w8'
    # ... which introspection now reports: 14 x 14 / 256 -> 256 at B = 256 is 784 blocks = 3 rounds + 16 blocks -> the tail blocks are K-sliced,
    # and the launch list names the slice and the reduce launch beside the body
    gw = _ffi.conv_geom(256, 14, 14, 256, 256, 3, 3, 1, 1, 1, 1, (1, 1, 1, 1))
    assert lib.mcn_conv2d_kslices(_ffi.CONV_FWD, ctypes.byref(gw), _ffi.F32) == 8 and lib.mcn_conv2d_kslices(_ffi.CONV_DGRAD, ctypes.byref(gw), _ffi.F32) == 8
    assert lib.mcn_conv2d_launch_list(_ffi.CONV_FWD, ctypes.byref(gw), _ffi.F32, lbuf, 512) == 3
    assert [ln.rsplit(':', 1)[0] for ln in lbuf.value.decode().splitlines()] == ['conv_wino_f2k3_w8<0, 0>', 'conv_wino_f2k3_w8<128, 0>', 'conv_wino_f2k3_w8<64, 0>']
    gw.tile = 0x100                                                                                      # MCN_TILE_NOSPLIT
    assert lib.mcn_conv2d_kslices(_ffi.CONV_FWD, ctypes.byref(gw), _ffi.F32) == 1 and lib.mcn_conv2d_launch_list(_ffi.CONV_FWD, ctypes.byref(gw), _ffi.F32, lbuf, 512) == 1
    gb.tile = 0x200                                                                                      # MCN_TILE_NOWINO: the direct kernels
    assert lib.mcn_conv2d_kslices(_ffi.CONV_FWD, ctypes.byref(gb), _ffi.F32) > 1 and lib.mcn_conv2d_kslices(_ffi.CONV_FWD, ctypes.byref(gb), _ffi.BF16) == 1
    gb.tile = 0x300
    assert lib.mcn_conv2d_kslices(_ffi.CONV_FWD, ctypes.byref(gb), _ffi.F32) == 1
    assert lib.mcn_decoupled_decay(0, 10, 0.1, 0, 0.0, 0) == _ffi.E_BADARG
    assert lib.mcn_bn_bwd_frozen(0, 0, 0, 0, 0, 0, 0, 1e-3, 0, 0, 0, 0, 1.0, 10, 4, 0, _ffi.F32, 0, 0, 0) == _ffi.E_BADARG
    assert lib.mcn_bn_fwd_train(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.9, 10, 4, 1e-3, 0, _ffi.F32, 0, 0, 0) == _ffi.E_BADARG
    assert lib.mcn_bn_workspace_bytes(1000, 64) > 0
    assert lib.mcn_relu_fwd(0, 0, 10, _ffi.F32, 0) == _ffi.E_BADARG
    assert lib.mcn_sgd_nesterov_fused(0, 0, 0, 0, 10, 0.1, 0.9, 0.0, 0.0, 0.99, 1.0, 0) == _ffi.E_BADARG
    with pytest.raises(_ffi.McnError):
        _ffi.check(lib.mcn_softmax_xent_fwd_bwd(0, 0, 0, 0, 0, 0, 0, 0, 4, 10, 0.0, 1.0, 0))


def test_missing_library_fails_loudly(tmp_path):
    from myconvnet_amd import _ffi
    with pytest.raises(ImportError, match='no CPU fallback'):
        _ffi.load(str(tmp_path / 'libmcn_hip.so'))


def test_pixel_pair_geometry_without_gpu():
    """mcn_conv2d_pair_geom (no launch): the stride-2 stem on <= 4 channels in a 2-byte type becomes a stride-(SH,1) conv of the
    [N,H,W/2,8] view; fp32, odd widths, wide inputs and stride-1 convs have no such form."""
    import ctypes
    from myconvnet_amd import _ffi
    lib = _ffi.lib
    g = _ffi.conv_geom(256, 224, 224, 3, 64, 7, 7, 2, 2, 1, 1, (2, 3, 2, 3), 4)              # ResNet stem, TF SAME pads
    pg = _ffi.ConvGeom()
    assert lib.mcn_conv2d_pair_geom(ctypes.byref(g), _ffi.BF16, ctypes.byref(pg)) == 1
    assert (pg.N, pg.H, pg.W, pg.Cin, pg.Cout, pg.KH, pg.KW, pg.SH, pg.SW, pg.padT, pg.padB, pg.padL, pg.padR, pg.x_cs) == (256, 224, 112, 8, 64, 7, 4, 2, 1, 2, 3, 1, 2, 8)
    g3 = _ffi.conv_geom(8, 224, 224, 3, 32, 3, 3, 2, 2, 1, 1, (0, 1, 0, 1), 4)               # EfficientNet stem
    assert lib.mcn_conv2d_pair_geom(ctypes.byref(g3), _ffi.F16, ctypes.byref(pg)) == 1 and (pg.KW, pg.padL, pg.padR, pg.W) == (2, 0, 1, 112)
    assert lib.mcn_conv2d_pair_geom(ctypes.byref(g), _ffi.F32, ctypes.byref(pg)) == 0        # 4-element chunks: nothing to gain
    for bad in (_ffi.conv_geom(8, 224, 223, 3, 64, 7, 7, 2, 2, 1, 1, (2, 3, 2, 2), 4),        # odd width
                _ffi.conv_geom(8, 224, 224, 8, 64, 7, 7, 2, 2, 1, 1, (2, 3, 2, 3), 8),        # more than 4 channels
                _ffi.conv_geom(8, 224, 224, 3, 64, 3, 3, 1, 1, 1, 1, (1, 1, 1, 1), 4),        # stride 1
                _ffi.conv_geom(8, 224, 224, 3, 64, 7, 7, 2, 2, 1, 1, (2, 3, 2, 3), 8)):       # image stored 8 channels per pixel
        assert lib.mcn_conv2d_pair_geom(ctypes.byref(bad), _ffi.BF16, ctypes.byref(pg)) == 0
    assert lib.mcn_conv2d_pair_weights(0, 0, ctypes.byref(g), _ffi.BF16, 0) == _ffi.E_BADARG
