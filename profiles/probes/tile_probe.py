"""Per-layer timing of every NT tile candidate (forward, forward + BN statistics, dgrad, dgrad + masked residual) through the C-ABI with
pre-packed filters, B = 256.  usage: python profiles/probes/tile_probe.py [dtype] [layer-set: all|k3|k1] [first_tile last_tile]"""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, 'tests')
sys.path.insert(0, '.')
from myconvnet_amd import _ffi  # noqa: E402
import abi_util as u  # noqa: E402

lib = _ffi.lib
B = 256
K3 = [(56, 64, 64, 3, 1), (28, 128, 128, 3, 1), (14, 256, 256, 3, 1), (7, 512, 512, 3, 1), (56, 128, 128, 3, 2), (28, 256, 256, 3, 2), (14, 512, 512, 3, 2)]
K1 = [(56, 64, 256, 1, 1), (56, 256, 64, 1, 1), (28, 128, 512, 1, 1), (28, 512, 128, 1, 1), (14, 256, 1024, 1, 1), (14, 1024, 256, 1, 1), (7, 512, 2048, 1, 1),
      (7, 2048, 512, 1, 1), (56, 256, 128, 1, 1), (28, 512, 256, 1, 1), (14, 1024, 512, 1, 1)]


def timeit(fn, reps=20):
    for _ in range(3):
        rc = fn()
    assert rc == 0, _ffi.last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else 'bfloat16'
    which = sys.argv[2] if len(sys.argv) > 2 else 'all'
    t0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    t1 = int(sys.argv[4]) if len(sys.argv) > 4 else lib.mcn_conv2d_tile_candidates(_ffi.CONV_FWD)
    layers = {'all': K3 + K1, 'k3': K3, 'k1': K1}[which]
    md = u.MDT[dtype]
    for (h, ci, co, k, s) in layers:
        x = torch.randn((B, h, h, ci), device='cuda').to(u.TDT[dtype])
        w = (torch.randn((k, k, ci, co), device='cuda') / np.sqrt(k * k * ci)).float()
        oh = -(-h // s)
        y = torch.empty((B, oh, oh, co), device='cuda', dtype=u.TDT[dtype])
        dy = torch.randn((B, oh, oh, co), device='cuda').to(u.TDT[dtype])
        dx = torch.empty_like(x)
        src = torch.randn_like(x)
        mask = torch.randint(0, 256, (int(lib.mcn_bn_relu_mask_bytes(B * h * h, ci, md)),), device='cuda', dtype=torch.uint8)
        flop = 2.0 * B * oh * oh * k * k * ci * co
        print('%s h%d %d->%d k%d s%d  (%.1f GFLOP)' % (dtype, h, ci, co, k, s, flop / 1e9), flush=True)
        for tile in range(t0, t1 + 1):                     # 0 = the library's own choice
            g = u.geom((B, h, h, ci), (k, k, ci, co), s, 'SAME')
            g.tile = tile
            wsb = max(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_FWD, ctypes.byref(g), md), lib.mcn_conv2d_workspace_bytes(_ffi.CONV_DGRAD, ctypes.byref(g), md))
            ws = u.workspace(wsb)
            pf, keep_f = u.prepack(w.cpu().numpy(), g, _ffi.CONV_FWD, dtype)
            pd, keep_d = u.prepack(w.cpu().numpy(), g, _ffi.CONV_DGRAD, dtype)
            pfp, pdp = (pf.data_ptr() if pf is not None else 0), (pd.data_ptr() if pd is not None else 0)
            rpp = ctypes.c_int32(0)
            rows = lib.mcn_conv2d_bnstats_rows(ctypes.byref(g), md, ctypes.byref(rpp))
            part = torch.empty((max(rows, 1) * 4, co), device='cuda', dtype=torch.float32)
            st = u.stream()
            name = ctypes.create_string_buffer(160)
            lib.mcn_conv2d_kernel_name(_ffi.CONV_FWD, ctypes.byref(g), md, name, 160)
            t_f = timeit(lambda: lib.mcn_conv2d_fwd(x.data_ptr(), w.data_ptr(), pfp, 0, y.data_ptr(), ctypes.byref(g), md, _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, st))
            t_s = timeit(lambda: lib.mcn_conv2d_fwd_bnstats(x.data_ptr(), w.data_ptr(), pfp, 0, y.data_ptr(), part.data_ptr(), ctypes.byref(g), md, _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, st)) if rows > 0 else -1.0
            t_d = timeit(lambda: lib.mcn_conv2d_dgrad(dy.data_ptr(), w.data_ptr(), pdp, dx.data_ptr(), ctypes.byref(g), 0, md, _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, st))
            t_a = -1.0
            if lib.mcn_conv2d_dgrad_addmasked_ok(ctypes.byref(g), md):
                t_a = timeit(lambda: lib.mcn_conv2d_dgrad_addmasked(dy.data_ptr(), w.data_ptr(), pdp, dx.data_ptr(), src.data_ptr(), mask.data_ptr(), ctypes.byref(g), md, _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, st))
            print('   tile %d: fwd %6.1f us (%4.0f TF)  +stats %6.1f  dgrad %6.1f (%4.0f TF)  +addmasked %6.1f   %s' % (tile, t_f, flop / t_f / 1e6, t_s, t_d, flop / t_d / 1e6, t_a, name.value.decode()), flush=True)


if __name__ == '__main__':
    main()
