"""Thin numpy <-> C-ABI helpers for the GPU parity tests (calls go through ctypes into libmcn_hip.so exactly as
the product does; torch is only the device-memory container)."""
import ctypes

import numpy as np
import torch

from myconvnet_amd import _ffi
from myconvnet_amd._ffi import lib
from oracle import ops as O

DEV = 'cuda:0'
TDT = {'float32': torch.float32, 'bfloat16': torch.bfloat16, 'float16': torch.float16}
MDT = {'float32': _ffi.F32, 'bfloat16': _ffi.BF16, 'float16': _ffi.F16}


def dev(a, dtype='float32'):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV).to(TDT[dtype]).contiguous()


def host(t):
    torch.cuda.synchronize()
    return t.float().cpu().numpy()


def stream():
    return torch.cuda.current_stream().cuda_stream


def bf16_round(a):
    """Round an fp32/fp64 array to bf16 precision (what the device stores), returned as float64."""
    return torch.as_tensor(np.asarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy().astype(np.float64)


def lp_round(a, dtype):
    """Round to the storage precision of `dtype` (what the device stores), returned as float64."""
    if dtype == 'float32':
        return np.asarray(a, dtype=np.float32).astype(np.float64)
    return torch.as_tensor(np.asarray(a, dtype=np.float32)).to(TDT[dtype]).float().numpy().astype(np.float64)


def geom(x_shape, w_shape, stride, padding, dilation=1, x_cs=0):
    n, h, w, c = x_shape
    kh, kw, cin, cout = w_shape
    sh, sw = O._pair(stride)
    dh, dw = O._pair(dilation)
    pads = O.resolve_pads(h, w, kh, kw, sh, sw, padding, dh, dw)
    return _ffi.conv_geom(n, h, w, cin, cout, kh, kw, sh, sw, dh, dw, pads, x_cs)


def workspace(nbytes):
    return torch.zeros(max(int(nbytes), 256) // 4 + 64, dtype=torch.float32, device=DEV)


def prepack(w, g, op, dtype):
    """Packed operand through the batch API (one job); returns the device tensor or None when the op takes none."""
    nb = int(lib.mcn_conv2d_packed_bytes(op, ctypes.byref(g), MDT[dtype]))
    if nb == 0:
        return None, None
    wd = dev(w)
    buf = torch.zeros(nb, dtype=torch.uint8, device=DEV)
    job = (_ffi.PackJob * 1)(_ffi.PackJob(wd.data_ptr(), buf.data_ptr(), g, op, 0))
    tb = int(lib.mcn_conv2d_pack_table_bytes(job, 1))
    host = (ctypes.c_char * tb)()
    nd = ctypes.c_int32(0)
    _ffi.check(lib.mcn_conv2d_pack_table_build(job, 1, MDT[dtype], ctypes.cast(host, ctypes.c_void_p), tb, ctypes.byref(nd)))
    table = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(DEV)
    _ffi.check(lib.mcn_conv2d_pack_run(table.data_ptr(), nd.value, MDT[dtype], stream()))
    return buf, (wd, table)


def conv_fwd(x, w, stride, padding, dilation=1, dtype='float32', bias=None, x_cs=0, use_prepack=False):
    g = geom(x.shape[:3] + (w.shape[2],), w.shape, stride, padding, dilation, x_cs)
    oh = O.out_size(x.shape[1], w.shape[0], O._pair(stride)[0], padding, O._pair(dilation)[0])
    ow = O.out_size(x.shape[2], w.shape[1], O._pair(stride)[1], padding, O._pair(dilation)[1])
    xd, wd = dev(x, dtype), dev(w)
    bd = dev(bias) if bias is not None else None
    y = torch.full((x.shape[0], oh, ow, w.shape[3]), float('nan'), dtype=TDT[dtype], device=DEV)
    ws = workspace(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_FWD, ctypes.byref(g), MDT[dtype]))
    pk, keep = prepack(w, g, _ffi.CONV_FWD, dtype) if use_prepack else (None, None)
    packed = pk.data_ptr() if pk is not None else 0
    _ffi.check(lib.mcn_conv2d_fwd(xd.data_ptr(), wd.data_ptr(), packed, bd.data_ptr() if bd is not None else 0, y.data_ptr(), ctypes.byref(g),
                                  MDT[dtype], _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, stream()))
    return host(y)


def conv_dgrad(dy, w, x_shape, stride, padding, dilation=1, dtype='float32', accumulate_into=None, use_prepack=False):
    g = geom(x_shape, w.shape, stride, padding, dilation)
    dyd, wd = dev(dy, dtype), dev(w)
    if accumulate_into is not None:
        dx = dev(accumulate_into, dtype)
    else:
        dx = torch.full(tuple(x_shape), float('nan'), dtype=TDT[dtype], device=DEV)
    ws = workspace(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_DGRAD, ctypes.byref(g), MDT[dtype]))
    pk, keep = prepack(w, g, _ffi.CONV_DGRAD, dtype) if use_prepack else (None, None)
    _ffi.check(lib.mcn_conv2d_dgrad(dyd.data_ptr(), wd.data_ptr(), pk.data_ptr() if pk is not None else 0, dx.data_ptr(), ctypes.byref(g), 1 if accumulate_into is not None else 0,
                                    MDT[dtype], _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, stream()))
    return host(dx)


def conv_wgrad(x, dy, w_shape, stride, padding, dilation=1, dtype='float32', with_bias=False, scale=1.0, x_cs=0):
    g = geom(x.shape[:3] + (w_shape[2],), w_shape, stride, padding, dilation, x_cs)
    xd, dyd = dev(x, dtype), dev(dy, dtype)
    dw = torch.full(tuple(w_shape), float('nan'), dtype=torch.float32, device=DEV)
    db = torch.full((w_shape[3],), float('nan'), dtype=torch.float32, device=DEV) if with_bias else None
    ws = workspace(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_WGRAD, ctypes.byref(g), MDT[dtype]))
    _ffi.check(lib.mcn_conv2d_wgrad(xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), db.data_ptr() if db is not None else 0, ctypes.byref(g),
                                    float(scale), MDT[dtype], _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, stream()))
    return (host(dw), host(db)) if with_bias else host(dw)


def bn_fwd_train(x, gamma, beta, eps=1e-3, dtype='float32', skip=None, act=0, running=None, momentum=0.99, want_mask=False):
    c = x.shape[-1]
    m = x.size // c
    xd = dev(x, dtype)
    gd, bd = dev(gamma), dev(beta)
    sd = dev(skip, dtype) if skip is not None else None
    y = torch.full(x.shape, float('nan'), dtype=TDT[dtype], device=DEV)
    sm, si, bm, bv = [torch.zeros(c, dtype=torch.float32, device=DEV) for _ in range(4)]
    rm = dev(running[0]) if running is not None else None
    rv = dev(running[1]) if running is not None else None
    ws = workspace(lib.mcn_bn_workspace_bytes(m, c))
    mask = torch.full((max(int(lib.mcn_bn_relu_mask_bytes(m, c, MDT[dtype])), 1),), 0xAA, dtype=torch.uint8, device=DEV) if want_mask else None
    _ffi.check(lib.mcn_bn_fwd_train(xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), sd.data_ptr() if sd is not None else 0, y.data_ptr(),
                                    mask.data_ptr() if mask is not None else 0, sm.data_ptr(), si.data_ptr(), bm.data_ptr(), bv.data_ptr(), rm.data_ptr() if rm is not None else 0,
                                    rv.data_ptr() if rv is not None else 0, momentum, m, c, eps, act, MDT[dtype], ws.data_ptr(),
                                    ws.numel() * 4, stream()))
    out = dict(y=host(y), save_mean=host(sm), save_invstd=host(si), batch_mean=host(bm), batch_var=host(bv))
    if running is not None:
        out['running_mean'], out['running_var'] = host(rm), host(rv)
    if mask is not None:
        torch.cuda.synchronize()
        out['relu_mask'] = mask.cpu().numpy()
    return out


def bn_bwd(dy, x, y, gamma, save_mean, save_invstd, dtype='float32', act=0, want_dskip=False, scale=1.0, beta=None, relu_mask=None):
    c = x.shape[-1]
    m = x.size // c
    dyd, xd = dev(dy, dtype), dev(x, dtype)
    yd = dev(y, dtype) if y is not None else None
    gd, smd, sid = dev(gamma), dev(save_mean), dev(save_invstd)
    bd = dev(beta) if beta is not None else None
    dx = torch.full(x.shape, float('nan'), dtype=TDT[dtype], device=DEV)
    dsk = torch.full(x.shape, float('nan'), dtype=TDT[dtype], device=DEV) if want_dskip else None
    dg, db = torch.zeros(c, dtype=torch.float32, device=DEV), torch.zeros(c, dtype=torch.float32, device=DEV)
    ws = workspace(lib.mcn_bn_workspace_bytes(m, c))
    mk = torch.as_tensor(relu_mask).to(DEV) if relu_mask is not None else None
    _ffi.check(lib.mcn_bn_bwd(dyd.data_ptr(), xd.data_ptr(), yd.data_ptr() if yd is not None else 0, mk.data_ptr() if mk is not None else 0, gd.data_ptr(), bd.data_ptr() if bd is not None else 0, smd.data_ptr(), sid.data_ptr(),
                              dx.data_ptr(), dsk.data_ptr() if dsk is not None else 0, dg.data_ptr(), db.data_ptr(), float(scale), m, c, act,
                              MDT[dtype], ws.data_ptr(), ws.numel() * 4, stream()))
    return host(dx), host(dg), host(db), (host(dsk) if dsk is not None else None)


# ---- EfficientNet row ---------------------------------------------------------------------------------
def dw_geom(x_shape, k, stride, padding, dilation=1):
    n, h, w, c = x_shape
    sh, sw = O._pair(stride)
    dh, dw = O._pair(dilation)
    pads = O.resolve_pads(h, w, k, k, sh, sw, padding, dh, dw)
    return _ffi.conv_geom(n, h, w, c, c, k, k, sh, sw, dh, dw, pads, 0)


def dwconv_fwd(x, w, stride, padding, dilation=1, dtype='float32'):
    k = w.shape[0]
    g = dw_geom(x.shape, k, stride, padding, dilation)
    oh = O.out_size(x.shape[1], k, O._pair(stride)[0], padding, O._pair(dilation)[0])
    ow = O.out_size(x.shape[2], k, O._pair(stride)[1], padding, O._pair(dilation)[1])
    xd, wd = dev(x, dtype), dev(w.reshape(k, k, -1))
    y = torch.full((x.shape[0], oh, ow, x.shape[3]), float('nan'), dtype=TDT[dtype], device=DEV)
    _ffi.check(lib.mcn_dwconv2d_fwd(xd.data_ptr(), wd.data_ptr(), y.data_ptr(), ctypes.byref(g), MDT[dtype], stream()))
    return host(y)


def dwconv_dgrad(dy, w, x_shape, stride, padding, dilation=1, dtype='float32', accumulate_into=None):
    k = w.shape[0]
    g = dw_geom(x_shape, k, stride, padding, dilation)
    dyd, wd = dev(dy, dtype), dev(w.reshape(k, k, -1))
    dx = dev(accumulate_into, dtype) if accumulate_into is not None else torch.full(tuple(x_shape), float('nan'), dtype=TDT[dtype], device=DEV)
    _ffi.check(lib.mcn_dwconv2d_dgrad(dyd.data_ptr(), wd.data_ptr(), dx.data_ptr(), ctypes.byref(g), 1 if accumulate_into is not None else 0,
                                      MDT[dtype], stream()))
    return host(dx)


def dwconv_wgrad(x, dy, k, stride, padding, dilation=1, dtype='float32', scale=1.0):
    g = dw_geom(x.shape, k, stride, padding, dilation)
    xd, dyd = dev(x, dtype), dev(dy, dtype)
    dw = torch.full((k, k, x.shape[3], 1), float('nan'), dtype=torch.float32, device=DEV)
    ws = workspace(lib.mcn_dwconv2d_workspace_bytes(ctypes.byref(g), MDT[dtype]))
    _ffi.check(lib.mcn_dwconv2d_wgrad(xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), ctypes.byref(g), float(scale), MDT[dtype], ws.data_ptr(),
                                      ws.numel() * 4, stream()))
    return host(dw)


def channel_scale(x, m, dy, dtype='float32'):
    n, h, w, c = x.shape
    xd, md, dyd = dev(x, dtype), dev(m.reshape(n, c), dtype), dev(dy, dtype)
    y = torch.full(x.shape, float('nan'), dtype=TDT[dtype], device=DEV)
    dx = torch.full(x.shape, float('nan'), dtype=TDT[dtype], device=DEV)
    dm = torch.full((n, c), float('nan'), dtype=TDT[dtype], device=DEV)
    _ffi.check(lib.mcn_channel_scale_fwd(xd.data_ptr(), md.data_ptr(), y.data_ptr(), n, h * w, c, MDT[dtype], stream()))
    _ffi.check(lib.mcn_channel_scale_bwd(dyd.data_ptr(), xd.data_ptr(), md.data_ptr(), dx.data_ptr(), dm.data_ptr(), n, h * w, c, MDT[dtype], stream()))
    return host(y), host(dx), host(dm)


def act(x, dy, kind, dtype='float32', param=None):
    xd, dyd = dev(x, dtype), dev(dy, dtype)
    y = torch.full(x.shape, float('nan'), dtype=TDT[dtype], device=DEV)
    dx = torch.full(x.shape, float('nan'), dtype=TDT[dtype], device=DEV)
    if param is None:
        _ffi.check(lib.mcn_act_fwd(xd.data_ptr(), y.data_ptr(), x.size, kind, MDT[dtype], stream()))
        _ffi.check(lib.mcn_act_bwd(dyd.data_ptr(), xd.data_ptr(), y.data_ptr(), dx.data_ptr(), x.size, kind, MDT[dtype], stream()))
    else:
        _ffi.check(lib.mcn_act_fwd_p(xd.data_ptr(), y.data_ptr(), x.size, kind, float(param), MDT[dtype], stream()))
        _ffi.check(lib.mcn_act_bwd_p(dyd.data_ptr(), xd.data_ptr(), y.data_ptr(), dx.data_ptr(), x.size, kind, float(param), MDT[dtype], stream()))
    return host(y), host(dx)


def bn_fwd_infer(x, gamma, beta, mean, var, eps=1e-3, dtype='float32', act=0):
    c = x.shape[-1]
    xd, gd, bd, md, vd = dev(x, dtype), dev(gamma), dev(beta), dev(mean), dev(var)
    y = torch.full(x.shape, float('nan'), dtype=TDT[dtype], device=DEV)
    _ffi.check(lib.mcn_bn_fwd_infer(xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), md.data_ptr(), vd.data_ptr(), 0, y.data_ptr(), x.size // c, c,
                                    eps, act, MDT[dtype], stream()))
    return host(y)
