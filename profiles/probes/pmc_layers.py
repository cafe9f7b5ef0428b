"""A handful of stand-alone conv launches for a counter pass (rocprofv3 --pmc ... -- python3 profiles/probes/pmc_layers.py): the 14x14 3x3 layer on the window ping-pong
kernel (tile 0 = the library's choice) and on the two-buffer kernel (tile 1), the expanding / contracting 1x1 layers forward with statistics, and the dgrad with the masked fan-in.
Each call is launched REPS times; profiles/probes/pmc_layers_summary.py averages the counters per kernel symbol."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, 'tests')
sys.path.insert(0, '.')
from myconvnet_amd import _ffi  # noqa: E402
import abi_util as u  # noqa: E402

lib = _ffi.lib
B, REPS = 256, 6
LAYERS = [(14, 256, 256, 3, 0), (14, 256, 256, 3, 1), (7, 512, 512, 3, 0), (7, 512, 512, 3, 1), (28, 128, 128, 3, 0),
          (14, 256, 1024, 1, 0), (28, 128, 512, 1, 0), (56, 64, 256, 1, 0), (14, 1024, 256, 1, 0)]


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else 'bfloat16'
    md = u.MDT[dtype]
    for (h, ci, co, k, tile) in LAYERS:
        x = torch.randn((B, h, h, ci), device='cuda').to(u.TDT[dtype])
        w = (torch.randn((k, k, ci, co), device='cuda') / np.sqrt(k * k * ci)).float()
        y = torch.empty((B, h, h, co), device='cuda', dtype=u.TDT[dtype])
        dy = torch.randn((B, h, h, co), device='cuda').to(u.TDT[dtype])
        dx = torch.empty_like(x)
        src = torch.randn_like(x)
        mask = torch.randint(0, 256, (int(lib.mcn_bn_relu_mask_bytes(B * h * h, ci, md)),), device='cuda', dtype=torch.uint8)
        g = u.geom((B, h, h, ci), (k, k, ci, co), 1, 'SAME')
        g.tile = tile
        ws = u.workspace(max(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_FWD, ctypes.byref(g), md), lib.mcn_conv2d_workspace_bytes(_ffi.CONV_DGRAD, ctypes.byref(g), md)))
        pf, keep_f = u.prepack(w.cpu().numpy(), g, _ffi.CONV_FWD, dtype)
        pd, keep_d = u.prepack(w.cpu().numpy(), g, _ffi.CONV_DGRAD, dtype)
        pfp, pdp = (pf.data_ptr() if pf is not None else 0), (pd.data_ptr() if pd is not None else 0)
        rpp = ctypes.c_int32(0)
        rows = lib.mcn_conv2d_bnstats_rows(ctypes.byref(g), md, ctypes.byref(rpp))
        part = torch.empty((max(rows, 1) * 4, co), device='cuda', dtype=torch.float32)
        st = u.stream()
        for _ in range(REPS):
            _ffi.check(lib.mcn_conv2d_fwd_bnstats(x.data_ptr(), w.data_ptr(), pfp, 0, y.data_ptr(), part.data_ptr(), ctypes.byref(g), md, _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, st))
        if k == 1 and lib.mcn_conv2d_dgrad_addmasked_ok(ctypes.byref(g), md):
            for _ in range(REPS):
                _ffi.check(lib.mcn_conv2d_dgrad_addmasked(dy.data_ptr(), w.data_ptr(), pdp, dx.data_ptr(), src.data_ptr(), mask.data_ptr(), ctypes.byref(g), md, _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, st))
        torch.cuda.synchronize()
        print('done', h, ci, co, k, tile, flush=True)


if __name__ == '__main__':
    main()
