"""Per-layer timing of the depthwise forward / dgrad / wgrad of EfficientNet-B0 (B = 512, bf16) through the C-ABI: us and TB/s of the
tensors each pass must move once.  usage: python profiles/probes/dw_probe.py [dtype]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, ROOT)
from myconvnet_amd import _ffi  # noqa: E402
import abi_util as u  # noqa: E402

lib = _ffi.lib
B = 512
# (H, C, K, stride) of the 16 depthwise layers of EfficientNet-B0 at 224 x 224 (models/efficientnet.py:126-197)
LAYERS = [(112, 32, 3, 1), (112, 96, 3, 2), (56, 144, 3, 1), (56, 144, 5, 2), (28, 240, 5, 1), (28, 240, 3, 2), (14, 480, 3, 1), (14, 480, 5, 1), (14, 672, 5, 1),
          (14, 672, 5, 2), (7, 1152, 5, 1), (7, 1152, 3, 1)]


def timeit(fn, reps=10):
    for _ in range(2):
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
    md, es = u.MDT[dtype], (4 if dtype == 'float32' else 2)
    for (h, c, k, s) in LAYERS:
        x = torch.randn((B, h, h, c), device='cuda').to(u.TDT[dtype])
        w = torch.randn((k, k, c, 1), device='cuda').float()
        oh = -(-h // s)
        y = torch.empty((B, oh, oh, c), device='cuda', dtype=u.TDT[dtype])
        dy = torch.randn_like(y)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        g = u.geom((B, h, h, c), (k, k, c, c), s, 'SAME')                   # (depthwise: Cout = Cin, channel multiplier 1)
        ws = u.workspace(lib.mcn_dwconv2d_workspace_bytes(ctypes.byref(g), md))
        byt = (x.numel() + y.numel()) * es
        tf = timeit(lambda: lib.mcn_dwconv2d_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), ctypes.byref(g), md, u.stream()))
        td = timeit(lambda: lib.mcn_dwconv2d_dgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), ctypes.byref(g), 0, md, u.stream()))
        tw = timeit(lambda: lib.mcn_dwconv2d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ctypes.byref(g), 1.0, md, ws.data_ptr(), ws.numel() * 4, u.stream()))
        print('%s %3dx%-3d C %4d k%d s%d  %6.1f MB  fwd %6.1f us %5.2f TB/s  dgrad %6.1f us %5.2f  wgrad %6.1f us %5.2f' % (
            dtype, h, h, c, k, s, byt / 1e6, tf, byt / tf / 1e6, td, byt / td / 1e6, tw, byt / tw / 1e6), flush=True)


if __name__ == '__main__':
    main()
