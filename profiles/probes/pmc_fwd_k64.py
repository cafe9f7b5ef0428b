"""Counter passes over ONE kind of launch at a time: the plain fp32 1x1 forward (conv_gemm_nt<float,64,64,0,4,0>) on the layer that takes twice its MFMA time
(56 x 56 64 -> 256) next to one that does not (14 x 14 256 -> 1024, same FLOP), B = 256.  Run under rocprofv3 --pmc ... (profiles/collect_pmc_k64.sh);
profiles/probes/pmc_layers_summary.py averages per (kernel symbol, grid)."""
import ctypes
import sys

import numpy as np
import torch

import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, ROOT)
from myconvnet_amd import _ffi  # noqa: E402
import abi_util as u  # noqa: E402

lib = _ffi.lib
B, REPS = 256, 5
for (h, ci, co) in [(56, 64, 256), (14, 256, 1024), (28, 128, 512)]:
    md = u.MDT['float32']
    x = torch.randn((B, h, h, ci), device='cuda')
    w = (torch.randn((1, 1, ci, co), device='cuda') / np.sqrt(ci)).float()
    y = torch.empty((B, h, h, co), device='cuda')
    g = u.geom((B, h, h, ci), (1, 1, ci, co), 1, 'SAME')
    ws = u.workspace(lib.mcn_conv2d_workspace_bytes(_ffi.CONV_FWD, ctypes.byref(g), md))
    pf, keep = u.prepack(w.cpu().numpy(), g, _ffi.CONV_FWD, 'float32')
    for _ in range(REPS):
        _ffi.check(lib.mcn_conv2d_fwd(x.data_ptr(), w.data_ptr(), pf.data_ptr() if pf is not None else 0, 0, y.data_ptr(), ctypes.byref(g), md, _ffi.NHWC, ws.data_ptr(), ws.numel() * 4, u.stream()))
    torch.cuda.synchronize()
