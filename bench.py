#!/usr/bin/env python
"""bench.py — images/sec of the ResNet-v1.5-50 training step (forward + backward + fused Nesterov/L2/EMA update
[+ RCCL gradient all-reduce]) on synthetic 224x224x3 batches, B = 256 per GPU, on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W [--dtype fp32|bf16|fp16] [--batch 256]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over one synthetic batch already resident in HBM.  Rank 0 prints ONE JSON line.
Default workload = BASELINE.json configs[1] (fp32, B=256, 1 GPU); `--dtype bf16` runs configs[2]'s arithmetic.
At N=1 the line also carries:
  roofline     — the dominant kernel's algorithmic FLOP/s (HIP events around every launch of that kernel in an extra,
                 instrumented pass over the same launch lists; torch's current stream IS the launch stream) against the
                 dense MFMA peak of the dtype (MI355X_MICROARCH.md: fp32 157.3 TFLOP/s, bf16 2500 TFLOP/s)
  cpu_baseline — a torch-CPU (oneDNN) restatement of the same step (oracle/torch_cpu.py, "port": the reference's TF-CPU path
                 cannot run here) timed on the host cores on a bounded sample (same network, B=32, 5 steps); the NumPy oracle
                 of record rides along as `numpy_port` (B=8)
  config.fetch = false: the timed step skips the reference's per-step device->host copy of Y_all / pred and the numpy
                 score (optimizers.py:590-594, :410); everything else of `_step` is inside the timed region
  bf16, fp16   — secondary measurements of the same step with bf16 storage (the north-star arithmetic) and with the reference's own
                 half precision (fp16 storage, loss scaling 128), only in the default fp32 run.
roofline.traffic = HBM bytes per launch of that kernel from the committed PMC passes (profiles/collect.sh; null if absent or if the
file's build id is not the loaded library's).  Winograd launches (fp32 3x3 / stride 1) are booked with their EXECUTED flop (16/36 of the
direct count x tile cover) in roofline / roofline_by_kernel; `direct_equivalent_tflops` carries the direct-convolution rate of the same calls.
--model efficientnet_b0 | deeplabv3plus: BASELINE configs[3] / configs[4] on one GPU (secondary workloads; roofline = the dominant
streaming C-ABI call against the HBM peak).
--no-overlap: wgrad on the main stream (profiling: per-kernel averages are then not stretched by co-running kernels).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAIN_FLOP_PER_IMAGE = 2 * (3 * (4087136256 + 2048000) - 118013952)      # 24,299,077,632
PEAK_TFLOPS = {'fp32': 157.3, 'bf16': 2500.0, 'fp16': 2500.0}                             # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0                                                     # HBM3E, same guide


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--dtype', choices=['fp32', 'bf16', 'fp16'], default='fp32',
                    help="fp16 = the reference's own half precision (convnet.py:63 half_precision + optimizers.py:102-111 loss scaling, factor 128)")
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default 256; 512 for --model efficientnet_b0)')
    ap.add_argument('--model', default='resnet50', choices=['resnet50', 'efficientnet_b0', 'deeplabv3plus'],
                    help='resnet50 = the headline workload (BASELINE configs[1]/[2]); efficientnet_b0 = configs[3], deeplabv3plus = configs[4] on one GPU (secondary, SURVEY 8f-2 / 8f-3)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true')
    ap.add_argument('--no-roofline', action='store_true', help='A/B runs: timed steps only (no instrumented pass, no roofline object)')
    ap.add_argument('--no-ema', action='store_true', help='disable the EMA shadows (on by default in the reference)')
    ap.add_argument('--autotune', action='store_true', help='time the tile candidates per layer at start-up (untimed) instead of the library heuristic')
    ap.add_argument('--no-overlap', action='store_true', help='run wgrad on the main stream (serial kernels: the rocprofv3 per-kernel averages then equal the roofline object)')
    ap.add_argument('--all-kernels', action='store_true', help='kernel_ms_per_step lists every C-ABI entry point / kernel, not the top 14')
    ap.add_argument('--layers', action='store_true', help='print a per-launch table (stderr) from the instrumented pass')
    args = ap.parse_args()
    if args.batch is None:
        args.batch = {'efficientnet_b0': 512, 'deeplabv3plus': 16}.get(args.model, 256)
    return args


def build_model(args, dtype, world):
    import myconvnet_amd as M
    cls = {'efficientnet_b0': M.EfficientNetB0, 'deeplabv3plus': M.DeepLabV3PlusResNet50}.get(args.model, M.ResNet50)
    size, classes = (513, 19) if args.model == 'deeplabv3plus' else (224, 1000)          # configs[4]: 513x513 synthetic Cityscapes
    model = cls([size, size, 3], classes, batch_size=args.batch * world, num_gpus=world, half_precision=(dtype != 'fp32'),
                       half_precision_dtype=('float16' if dtype == 'fp16' else 'bfloat16'), seed=0, overlap_wgrad=not args.no_overlap, device='cuda:{}'.format(local_device()))
    opt = M.MomentumOptimizer(model, None, None, base_learning_rate=0.1, momentum=0.9, steps_per_epoch=5000, num_epochs=90,
                              update_ema=not args.no_ema, loss_scaling_factor=(128.0 if dtype == 'fp16' else 1.0))
    rank = int(os.environ.get('RANK', 0))
    rng = np.random.default_rng(1234 + rank)                               # SURVEY §8d synthetic inputs
    x = rng.random((args.batch, size, size, 3), dtype=np.float32)
    if args.model == 'deeplabv3plus':
        y = rng.integers(0, classes + 1, (args.batch, size, size)).astype(np.float32)   # 0 = ignored pixel
    else:
        y = rng.integers(0, classes, args.batch).astype(np.float32)
    model.feed(x, y)                                                       # resident in HBM before the timed region
    torch.cuda.synchronize()
    return model, opt


def run_steps(opt, n, fetch=False):
    for _ in range(n):
        opt._update_learning_rate()
        opt._step(None, fetch=fetch)
        opt.curr_step += 1


# kernel symbols (prefixes) behind the streaming C-ABI calls whose PMC bytes are summed per call (only calls that own their symbols in the
# profiled model: EfficientNet's BN + swish passes, squeeze-excite BN backward, depthwise convolutions)
CALL_KERNELS = {
    'mcn_bn_bwd': ['bn_bwd_reduce_kernel', 'bn_bwd_apply_kernel', 'bn_bwd_finalize_kernel'],
    'mcn_bn_bwd_se': ['bn_bwd_reduce_se_kernel', 'bn_bwd_apply_se_kernel'],
    'mcn_bn_bwd_se_sums': ['bn_bwd_apply_se_kernel', 'se_bwd_sums_kernel'],
    'mcn_channel_scale_bwd_dm_bnsums': ['se_bwd_pre_kernel', 'se_bwd_dm_fold_kernel'],
    'mcn_bn_act_scale_fwd': ['bn_act_scale_kernel'],
    'mcn_bn_fwd_train_fused': ['bn_apply_kernel', 'bn_fwd_finalize_fused_kernel', 'bn_fold_partials_kernel'],
    'mcn_bn_fwd_train_gap': ['bn_apply_gap_kernel'],
    'mcn_dwconv2d_fwd': ['dw_band_kernel', 'dw_strip_kernel'],
    'mcn_dwconv2d_dgrad': ['dw_dgrad', 'dw_band_kernel', 'dw_strip_kernel'],
    'mcn_dwconv2d_wgrad': ['dw_wgrad'],
}


def pmc_traffic(kernel, dtype, batch, model='resnet50'):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/collect.sh: separate FETCH_SIZE and
    WRITE_SIZE runs of this same command, corrected as MI355X_MICROARCH.md prescribes: KiB units, FETCH_SIZE doubled on
    gfx950).  PMC counters cannot be collected from inside the process, so this reads the summary; null when absent or
    when it was taken for another kernel / batch."""
    tag = dtype if model == 'resnet50' else '%s_%s' % (model, dtype)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'pmc_traffic_%s.json' % tag)
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return {}
    k = d.get('kernels', {}).get(kernel)
    if not k and kernel in CALL_KERNELS and d.get('steps_profiled'):
        # a streaming C-ABI call = several kernel symbols (reduce + finalize + apply ...): HBM bytes of all of them per call
        mine = [v for name, v in d.get('kernels', {}).items() if any(name.startswith(pfx) for pfx in CALL_KERNELS[kernel])]
        calls = d.get('calls_per_step', {}).get(kernel)
        if mine and calls:
            k = {'bytes_per_launch': int(sum(v['bytes_per_launch'] * v['launches_profiled'] for v in mine) / d['steps_profiled'] / calls)}
    if not k or d.get('batch') != batch:
        return {}
    # the summary must come from THIS build of the kernels: a csrc/ change without a re-run of profiles/collect.sh would otherwise
    # report the previous build's traffic without a word
    from myconvnet_amd._ffi import lib
    have = lib.mcn_build_id().decode()
    if d.get('build_id') != have:
        return {'traffic': None, 'traffic_note': 'profiles/pmc_traffic_%s.json was collected from build %s, this library is %s: re-run profiles/collect.sh'
                % (dtype, str(d.get('build_id'))[:12], have[:12])}
    return {'traffic': k['bytes_per_launch'], 'traffic_unit': 'bytes/launch', 'traffic_source': 'profiles/pmc_traffic_%s.json' % dtype}


def pmc_group_traffic(prefixes, dtype, batch, model='resnet50'):
    """HBM bytes PER STEP of every kernel symbol starting with one of `prefixes` (same committed PMC summary, same build-id rule as pmc_traffic);
    None when the summary is absent / stale."""
    tag = dtype if model == 'resnet50' else '%s_%s' % (model, dtype)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'pmc_traffic_%s.json' % tag)
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    from myconvnet_amd._ffi import lib
    if d.get('batch') != batch or d.get('build_id') != lib.mcn_build_id().decode() or not d.get('steps_profiled'):
        return None
    mine = [v for name, v in d.get('kernels', {}).items() if any(name.startswith(p) for p in prefixes)]
    if not mine:
        return None
    return sum(v['bytes_per_launch'] * v['launches_profiled'] for v in mine) / d['steps_profiled']


def timed(opt, steps, warmup, world, autotune=False):
    import torch.distributed as dist
    if autotune:                                       # untimed start-up: two steps to fill the buffers, then time the tile candidates
        run_steps(opt, 2)
        opt.model.autotune()
    main_cus = int(os.environ.get('MCN_MAIN_CUS', '0'))       # experiment (LABNOTES.md section 3, "CU masks"): the main stream on a subset of the CUs
    if main_cus > 0:
        from myconvnet_amd.graph import masked_stream
        ctx = torch.cuda.stream(masked_stream(torch.device('cuda', local_device()), main_cus, int(os.environ.get('MCN_MAIN_CU0', '0'))))
    else:
        import contextlib
        ctx = contextlib.nullcontext()
    with ctx:
        run_steps(opt, warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with ctx:
        run_steps(opt, steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


# ---- per-kernel timing (instrumented pass) ---------------------------------------------------------------------
def _not_persistent(sym):
    """conv_gemm_nt_pers<T, BM, BN, NW, E> -> the conv_gemm_nt symbol of the same tile (mcn_conv2d_kernel_name answers from the
    geometry alone; launches with a bias, and fp32 launches with the accumulate epilogue, stay on conv_gemm_nt)."""
    import re
    m = re.match(r'conv_gemm_nt_pers<(.+), (\d+), (\d+), (\d+), (\d+)>$', sym)
    if not m:
        return sym
    t, bm, bn, nw, e = m.groups()
    return 'conv_gemm_nt<{}, {}, {}, 0, {}, {}>'.format(t, bm, bn, nw, 0 if e == '3' else e)      # (3 = counted statistics: persistent only)


def _hbm_call_bytes(name, a, es):
    """Algorithmic HBM bytes of the streaming (non-GEMM) C-ABI calls: every tensor the call must read or write, once per pass that
    needs it (DESIGN.md section 3, table of kernels).  a = the call's argument list (executor.py), es = bytes per element.
    Returns 0 for calls without a figure (they still show up with their time)."""
    def mc(i):
        return float(a[i]) * float(a[i + 1])
    if name == 'mcn_bn_fwd_train_fused':                 # apply pass: x -> y (+ residual)
        return es * mc(16) * (2 + (1 if a[6] else 0))
    if name == 'mcn_bn_fwd_train_fused_affskip':         # x, shortcut conv output -> y
        return es * mc(17) * 3
    if name == 'mcn_bn_fwd_train':                       # statistics pass + apply pass
        return es * mc(13) * (3 + (1 if a[3] else 0))
    if name == 'mcn_bn_fwd_train_gap':                   # statistics pass + apply pass (the pooled means ride in the apply pass; y = NULL: means only, no write)
        return es * float(a[12]) * float(a[13]) * float(a[14]) * (3 if a[3] else 2)
    if name == 'mcn_bn_act_scale_fwd':                   # BN input -> scaled output (the BN + swish output is rebuilt on the fly)
        return es * float(a[7]) * float(a[8]) * float(a[9]) * 2
    if name == 'mcn_bn_fwd_train_fused_maxpool':         # x -> pooled + arg-max
        n, h, w, c, oh, ow = a[15], a[16], a[17], a[18], a[-6], a[-5]
        return es * n * h * w * c + (es + 1) * n * oh * ow * c
    if name == 'mcn_bn_bwd':                             # reduce (dy, x [, y | byte mask]) + apply (dy, x [, y | byte mask] -> dx [, dskip])
        sign = (2.0 / 16.0) if a[3] else (2.0 if a[2] else 0.0)          # the ReLU sign pattern: one byte per 16-byte chunk when the forward left a mask, else y itself
        return es * mc(13) * (5 + sign + (1 if a[9] else 0))
    if name == 'mcn_bn_bwd_from_partials':               # apply pass only (dy, x, byte mask -> dx)
        return es * mc(13) * (3 + 1.0 / 16.0)
    if name == 'mcn_bn_bwd_se':                          # both passes read dy and x, the apply pass writes dx
        return es * float(a[12]) * float(a[13]) * float(a[14]) * 5
    if name == 'mcn_bn_bwd_se_sums':                     # the apply pass only (dy, x -> dx): the sums come from mcn_channel_scale_bwd_dm_bnsums
        return es * float(a[13]) * float(a[14]) * float(a[15]) * 3
    if name == 'mcn_channel_scale_bwd_dm_bnsums':        # dy and the BN's input, once
        return es * float(a[8]) * float(a[9]) * float(a[10]) * 2
    if name == 'mcn_bn_bwd_maxpool':                     # both passes read x and the pooled gradient + arg-max, one writes dx
        n, h, w, c, oh, ow = a[11], a[12], a[13], a[14], a[-6], a[-5]
        return es * n * h * w * c * 3 + 2 * (es + 1) * n * oh * ow * c
    if name == 'mcn_bn_bwd_frozen':
        return es * mc(13) * 3
    if name.startswith('mcn_dwconv2d_'):
        gm = [x for x in a if hasattr(x, '_obj')][0]._obj
        oh = (gm.H + gm.padT + gm.padB - (gm.KH - 1) * gm.DH - 1) // gm.SH + 1
        ow = (gm.W + gm.padL + gm.padR - (gm.KW - 1) * gm.DW - 1) // gm.SW + 1
        return es * gm.N * gm.Cin * (gm.H * gm.W + oh * ow)
    if name in ('mcn_channel_scale_fwd', 'mcn_channel_scale_bwd', 'mcn_channel_scale_bwd_dm'):
        n, hw, c = a[-5], a[-4], a[-3]
        return es * float(n) * hw * c * (2 if name != 'mcn_channel_scale_bwd' else 4)
    return 0.0


WINO_DIRECT_FLOP = {}

CONV_OPS = ('mcn_conv2d_fwd', 'mcn_conv2d_fwd_bnstats', 'mcn_conv2d_dgrad', 'mcn_conv2d_dgrad_addmasked', 'mcn_conv2d_dgrad_bnred', 'mcn_conv2d_dgrad_addmasked_bnred',
            'mcn_conv2d_wgrad')
WINO_TAIL_LAUNCHES = {}
WINO_TAIL_SYMBOLS = {}


def conv_call_launches(name, a, dtype):
    """The kernel launches behind ONE conv C-ABI call of the launch lists, named as rocprofv3 names them.  name = the entry point, a = its
    argument list (executor.py), dtype = 'fp32' | 'bf16' | 'fp16'.  Returns (geom, [(symbol, weight)], dominant symbol, direct-convolution
    FLOP, algorithmic HBM bytes).  A call that launches several kernels lists them all: a strided dgrad one per stride-parity class (weight =
    its taps), a Winograd call with a K-sliced tail its body, slice and reduce symbols (body first).  tests/test_gpu_fullsize_fused.py calls
    this with its own argument lists, so that the launches it checks against the oracle are BY NAME the ones the bench line reports."""
    import ctypes
    from myconvnet_amd import _ffi
    from myconvnet_amd._ffi import lib
    op = {'mcn_conv2d_wgrad': _ffi.CONV_WGRAD}.get(name, _ffi.CONV_FWD if name.startswith('mcn_conv2d_fwd') else _ffi.CONV_DGRAD)
    mdt = {'fp32': _ffi.F32, 'bf16': _ffi.BF16, 'fp16': _ffi.F16}[dtype]
    lbuf = ctypes.create_string_buffer(1024)
    gm = [x for x in a if hasattr(x, '_obj')][0]._obj          # ctypes.byref(geom)
    oh = (gm.H + gm.padT + gm.padB - (gm.KH - 1) * gm.DH - 1) // gm.SH + 1
    ow = (gm.W + gm.padL + gm.padR - (gm.KW - 1) * gm.DW - 1) // gm.SW + 1
    flop = 2.0 * gm.N * oh * ow * gm.KH * gm.KW * gm.Cin * gm.Cout * getattr(gm, '_flop_scale', 1.0)   # (pixel-pair stem: count the 7x7x3 MACs)
    # one entry per GEMM launch of the call (a strided dgrad: one per stride-parity class, not all the same symbol)
    lib.mcn_conv2d_launch_list(op, ctypes.byref(gm), mdt, lbuf, 1024)
    launches = [ln.rsplit(':', 1) for ln in lbuf.value.decode().splitlines()]
    epi = None
    if name == 'mcn_conv2d_fwd_bnstats':                       # the instantiation with the BN-statistics epilogue
        epi = ', 1>'
    elif (name == 'mcn_conv2d_dgrad' and a[5]) or name == 'mcn_conv2d_dgrad_addmasked':
        epi = ', 2>'                                           # ... with the accumulate epilogue
    elif name == 'mcn_conv2d_dgrad_bnred':
        epi = ', 4>'                                           # ... with the BN-backward sums
    elif name == 'mcn_conv2d_dgrad_addmasked_bnred':
        epi = ', 5>'                                           # ... with the masked residual fan-in AND the BN-backward sums (round 4)
    wino = bool(launches) and launches[0][0].startswith('conv_wino')
    if wino:                                                   # (the slices store plain accumulators: <128, 0> whatever the call's epilogue)
        launches = [(k.replace(', 0>', epi) if (epi and not k.startswith('conv_wino_f2k3_w8<128')) else k, int(t)) for k, t in launches]
    else:
        launches = [(k.replace(', 0>', epi) if epi else k, int(t)) for k, t in launches]
    if (epi == ', 2>' and dtype == 'fp32') or epi in (', 4>', ', 5>') or (name == 'mcn_conv2d_fwd' and int(os.environ.get('MCN_NT_PERS', '1')) < 2):
        # fp32 keeps conv_gemm_nt for the accumulate epilogue; by default only the statistics forward is persistent
        launches = [(_not_persistent(k), t) for k, t in launches]
    key = launches[0][0] if wino else max(launches, key=lambda kt: kt[1])[0]               # (per-layer table: the launch with the most taps)
    es = 4 if dtype == 'fp32' else 2
    # algorithmic HBM bytes: each activation tensor once + the filter once (a stride-s 1x1 reads 1/s^2 of x)
    xe = gm.N * gm.H * gm.W * gm.Cin if (gm.KH > 1 or gm.SH == 1) else gm.N * oh * ow * gm.Cin
    byt = es * (xe + gm.N * oh * ow * gm.Cout) + (4 if name == 'mcn_conv2d_wgrad' else es) * gm.KH * gm.KW * gm.Cin * gm.Cout
    if (name == 'mcn_conv2d_dgrad' and a[5]) or name == 'mcn_conv2d_dgrad_addmasked':
        byt += es * xe                                         # accumulate / fused residual fan-in: one more read of a dx-sized tensor
    if name == 'mcn_conv2d_dgrad_bnred':
        byt += es * xe                                         # the BN's input, read beside the dx tile for sum dy' * x
    if name == 'mcn_conv2d_dgrad_addmasked_bnred':
        byt += 2 * es * xe                                     # the next unit's gradient (masked fan-in) and the BN's input
    return gm, launches, key, flop, float(byt)




def instrumented_pass(model, dtype, reps=3, layers=False):
    """Time every C-ABI launch of forward+backward with HIP events on the launch stream (torch's current stream IS the
    launch stream).  Conv calls are keyed by the exact kernel symbol rocprofv3 reports (mcn_conv2d_kernel_name); a call
    that launches the kernel several times (stride-2 dgrad: one launch per parity class) is counted as that many
    launches.  Returns {key: [launches, ms, flop]} per step."""
    import ctypes
    from myconvnet_amd import _ffi
    from myconvnet_amd._ffi import lib, check
    low = model._train_low
    sp = model.stream_ptr()
    ops = {'mcn_conv2d_fwd': _ffi.CONV_FWD, 'mcn_conv2d_fwd_bnstats': _ffi.CONV_FWD, 'mcn_conv2d_dgrad': _ffi.CONV_DGRAD, 'mcn_conv2d_dgrad_addmasked': _ffi.CONV_DGRAD, 'mcn_conv2d_dgrad_bnred': _ffi.CONV_DGRAD, 'mcn_conv2d_dgrad_addmasked_bnred': _ffi.CONV_DGRAD, 'mcn_conv2d_wgrad': _ffi.CONV_WGRAD}
    mdt = {'fp32': _ffi.F32, 'bf16': _ffi.BF16, 'fp16': _ffi.F16}[dtype]
    buf = ctypes.create_string_buffer(128)
    lbuf = ctypes.create_string_buffer(1024)
    table, rows = {}, {}
    calls = low.fwd.calls + low.bwd.calls
    WINO_TAIL_LAUNCHES.clear()
    WINO_TAIL_SYMBOLS.clear()
    WINO_DIRECT_FLOP.clear()
    low.prepack.run(sp)
    # cost of an empty event bracket on this stream (two records back to back), subtracted from every bracket below
    empt = []
    for _ in range(64):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e1.record()
        empt.append((e0, e1))
    torch.cuda.synchronize()
    bracket_ms = float(np.median([a.elapsed_time(b) for a, b in empt]))
    table['_bracket_us'] = [0, bracket_ms, 0.0, 0.0]
    for rep in range(reps):
        evs = []
        for fn, a in calls:
            a[-1] = sp
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(fn(*a))
            e1.record()
            evs.append((fn, a, e0, e1))
        torch.cuda.synchronize()
        if rep == 0:
            continue                                                       # first rep warms caches / clocks
        for fn, a, e0, e1 in evs:
            ms = max(e0.elapsed_time(e1) - bracket_ms, 0.0)
            name = getattr(fn, '__name__', 'other')
            if name in ('mcn_fc_fwd', 'mcn_fc_bwd'):
                # the fully connected layer is the 1x1-conv kernel on a [B,1,1,In] tensor: count it under the kernel symbol
                # rocprofv3 reports, so that launches and average durations agree with the --stats table
                B_, In_, Out_ = (a[4], a[5], a[6]) if name == 'mcn_fc_fwd' else (a[7], a[8], a[9])
                gm = _ffi.conv_geom(B_, 1, 1, In_, Out_, 1, 1, 1, 1, 1, 1, (0, 0, 0, 0), 0)
                op = _ffi.CONV_FWD if name == 'mcn_fc_fwd' else (_ffi.CONV_DGRAD if a[3] else _ffi.CONV_WGRAD)
                nl = lib.mcn_conv2d_kernel_name(op, ctypes.byref(gm), mdt, buf, 128)
                key = buf.value.decode()
                if name == 'mcn_fc_fwd':                                   # biased: not the persistent kernel
                    key = _not_persistent(key)
                flop = 2.0 * B_ * In_ * Out_
                es = 4 if dtype == 'fp32' else 2
                byt = es * B_ * (In_ + Out_) + (4 if op == _ffi.CONV_WGRAD else es) * In_ * Out_
            elif name not in ops:
                key, flop, nl, byt = name, 0.0, 1, 0.0
                try:
                    byt = float(_hbm_call_bytes(name, a, 4 if dtype == 'fp32' else 2))
                except (IndexError, TypeError, ValueError):
                    byt = 0.0
                if layers and name.startswith('mcn_dwconv2d_'):           # depthwise rows of the per-layer table (HBM-bound: GB/s is the figure)
                    gm = [x for x in a if hasattr(x, '_obj')][0]._obj
                    oh = (gm.H + gm.padT + gm.padB - (gm.KH - 1) * gm.DH - 1) // gm.SH + 1
                    ow = (gm.W + gm.padL + gm.padR - (gm.KW - 1) * gm.DW - 1) // gm.SW + 1
                    es = 4 if dtype == 'fp32' else 2
                    dflop = 2.0 * gm.N * oh * ow * gm.KH * gm.KW * gm.Cin
                    dbyt = es * gm.N * gm.Cin * (gm.H * gm.W + oh * ow)   # x and y (fwd), dy and dx (dgrad), x and dy (wgrad) once each
                    r = rows.setdefault(('dw_' + name[13:], gm.H, gm.Cin, gm.Cin, gm.KH, gm.SH, name), [0, 0.0, dflop, dbyt])
                    r[0] += 1
                    r[1] += ms
            else:
                gm, launches, key, flop, byt = conv_call_launches(name, a, dtype)
                nl = len(launches)
                if key.startswith('conv_wino'):
                    # a K-sliced Winograd tail adds a slice and a reduce launch to the call: the bracket covers all three, the entry is booked per
                    # CALL under the body's symbol (launches = calls, so that bytes per launch and PMC traffic share one denominator) and the
                    # tail launches ride along as `tail_launches_per_step`
                    WINO_TAIL_LAUNCHES[key] = WINO_TAIL_LAUNCHES.get(key, 0) + (len(launches) - 1)
                    WINO_TAIL_SYMBOLS.setdefault(key, set()).update(k for k, _ in launches[1:])
                    launches = launches[:1]
                    nl = 1
                if layers:
                    r = rows.setdefault((name[11:], gm.H, gm.Cin, gm.Cout, gm.KH, gm.SH, key), [0, 0.0, flop, byt])
                    r[0] += 1
                    r[1] += ms
                if key.startswith('conv_wino'):
                    # (the per-layer table above shows direct-equivalent TFLOP/s)  Winograd F(2x2, 3x3): the MFMAs execute 16 multiplications per 2x2 outputs instead of 36 (tiles hanging over an odd map's
                    # edge included).  The table carries the EXECUTED flop, so that `frac` stays a statement about the MFMA pipe; the
                    # direct-convolution flop of the same calls are kept beside it (roofline_by_kernel: direct_equivalent_tflops)
                    cov = (2.0 * ((gm.H + 1) // 2) / gm.H) * (2.0 * ((gm.W + 1) // 2) / gm.W)
                    WINO_DIRECT_FLOP[key] = WINO_DIRECT_FLOP.get(key, 0.0) + flop / (reps - 1)
                    flop *= 16.0 / 36.0 * cov
            if name in ops and len(launches) > 1:
                # the bracket covers all launches of the call: split time, flops and bytes by filter taps (= by flops)
                taps = float(sum(t for _, t in launches))
                for k, tp in launches:
                    t = table.setdefault(k, [0, 0.0, 0.0, 0.0])
                    t[0] += 1
                    t[1] += ms * tp / taps
                    t[2] += flop * tp / taps
                    t[3] += byt * tp / taps
                continue
            t = table.setdefault(key, [0, 0.0, 0.0, 0.0])
            t[0] += nl
            t[1] += ms
            t[2] += flop
            t[3] += byt
    if layers:
        print('kind   H   Cin  Cout k s  n    us/call   TFLOP/s   GB/s(min traffic)  kernel', file=sys.stderr)
        for key, (n, ms, flop, byt) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            us = ms / n * 1e3
            print('{:5s} {:3d} {:5d} {:5d} {} {} {:2d} {:10.1f} {:8.1f} {:8.0f}  {}'.format(key[0], key[1], key[2], key[3], key[4], key[5],
                  n // (reps - 1), us, flop / us / 1e6, byt / us / 1e3, key[6]), file=sys.stderr)
    for k in WINO_TAIL_LAUNCHES:
        WINO_TAIL_LAUNCHES[k] //= (reps - 1)
    for k, t in table.items():
        if k == '_bracket_us':
            continue
        t[0] //= (reps - 1)
        t[1] /= (reps - 1)
        t[2] /= (reps - 1)
        t[3] /= (reps - 1)
    return table


def streaming_roofline(model, dtype, batch, model_name, layers=False, all_kernels=False):
    """roofline object of an HBM-bound workload (EfficientNet-B0, DeepLabv3+): the entry — conv kernel symbol or streaming C-ABI call — with the
    most time per step among those with a byte / FLOP figure.  These networks are HBM-bound (2.3 GFLOP per image
    against ResNet-50's activation volume), so the figure is algorithmic bytes / time against the HBM peak unless the entry's arithmetic
    intensity says otherwise."""
    out = {}
    table = instrumented_pass(model, dtype, layers=layers)
    bracket_us = table.pop('_bracket_us')[1] * 1e3
    out['kernel_ms_per_step'] = {k: round(v[1], 3) for k, v in sorted(table.items(), key=lambda kv: -kv[1][1])[:(None if all_kernels else 14)]}
    out['kernel_ms_total'] = round(sum(v[1] for v in table.values()), 3)
    ridge = PEAK_TFLOPS[dtype] * 1e12 / (PEAK_HBM_GBS * 1e9)

    def entry(k, v):
        cnt, ms_k, flop, byt = v
        e = {'launches_per_step': cnt, 'ms_per_step': round(ms_k, 3), 'gbs': round(byt / (ms_k * 1e-3) / 1e9, 0) if ms_k > 0 else 0.0,
             'hbm_frac': round(byt / (ms_k * 1e-3) / 1e9 / PEAK_HBM_GBS, 3) if ms_k > 0 else 0.0, 'algorithmic_bytes_per_launch': int(byt / max(cnt, 1))}
        if flop > 0:
            e['tflops'] = round(flop / (ms_k * 1e-3) / 1e12, 1)
            e['bound'] = 'hbm' if flop / max(byt, 1.0) < ridge else 'mfma'
        else:
            e['bound'] = 'hbm'
        t = pmc_traffic(k, dtype, batch, model_name).get('traffic')
        if t is not None:
            e['traffic'] = t
        return e
    rated = {k: v for k, v in table.items() if v[3] > 0 and v[1] > 0}
    if rated:
        name, v = max(rated.items(), key=lambda kv: kv[1][1])
        e = entry(name, v)
        hbm = e['bound'] == 'hbm'
        ach = e['gbs'] if hbm else e.get('tflops', 0.0)
        peak = PEAK_HBM_GBS if hbm else PEAK_TFLOPS[dtype]
        out['roofline'] = {'bound': e['bound'], 'kernel': name, 'launches_per_step': v[0], 'avg_launch_us': round(v[1] / max(v[0], 1) * 1e3, 2), 'achieved': ach,
                           'peak': peak, 'unit': 'GB/s' if hbm else 'TFLOP/s', 'frac': round(ach / peak, 4), 'traffic': e.get('traffic'),
                           'algorithmic_bytes_per_launch': e['algorithmic_bytes_per_launch'], 'event_bracket_overhead_us': round(bracket_us, 2),
                           'note': 'entry = one C-ABI call (a BN call is its reduce / finalize / apply launches together); bytes = every tensor once per pass that needs it'}
        out['roofline_by_kernel'] = {k: entry(k, vv) for k, vv in sorted(rated.items(), key=lambda kv: -kv[1][1])[:10]}
    out['calls_per_step'] = {k: int(v[0]) for k, v in table.items() if v[0] > 0}
    return out


def host_threads():
    """CPU threads this process may really use: the cgroup CPU quota if there is one (a GPU box exposes 256 CPUs but gives a
    one-GPU job a share of 16: 128 oneDNN threads on that share ran the step 2.6x slower than 8 threads on 8 cores), else
    the scheduler affinity."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f, open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as g:
                q, p = int(f.read()), int(g.read())
            if q > 0:
                n = min(n, max(1, (q + p // 2) // p))
        except (OSError, ValueError):
            pass
    return int(os.environ.get('MCN_CPU_THREADS', n))


def cpu_baseline():
    """CPU stand-ins for the reference's `num_gpus=0` TensorFlow path (which cannot run in this image), timed on the host
    cores in the same run — BASELINE.md §4: the same ResNet-v1.5-50 fp32 training step at B=32, 2 warm-up + 5 timed steps,
    median.  Primary: oracle/torch_cpu.py (torch-CPU ops = oneDNN, autograd), pinned to the NumPy oracle by
    tests/test_oracle_vs_torch.py.  Secondary key `numpy_port`: the NumPy oracle of record itself (B=8, mostly
    single-threaded).  Neither is TensorFlow; both are labelled stand-ins."""
    from oracle import net as ON
    from oracle.torch_cpu import ResNetTorchCPU
    spec = ON.ResNetSpec.resnet50(1000)
    params, stats = ON.init_variables(spec.variables(), seed=0, dtype=np.float32)
    rng = np.random.default_rng(1234)
    B = 32
    x = rng.random((B, 224, 224, 3), dtype=np.float32)
    y = rng.integers(0, 1000, B).astype(np.float32)
    threads = host_threads()
    torch.set_num_threads(threads)
    tc = ResNetTorchCPU(spec, params, stats, channels_last=True)
    for _ in range(2):
        tc.train_step(x, y, batch_total=256)
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        tc.train_step(x, y, batch_total=256)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    out = dict(value=round(B / med, 3), unit='images/sec', cores=threads, kind='port', ms_per_step=round(med * 1e3, 1),
               impl='torch-CPU (oneDNN) stand-in, not TensorFlow', host_cpus=os.cpu_count(), omp_num_threads=os.environ.get('OMP_NUM_THREADS'),
               sample='ResNet-v1.5-50 fp32 224x224 training step (fwd + autograd bwd + Nesterov/L2/EMA), B=32, median of 5 steps after '
                      '2 warm-up, torch intra-op threads = `cores`; the reference TF-1.x CPU path cannot run in this image')
    del tc
    st = ON.TrainState(params, stats)
    xs, ys = x[:8], y[:8]
    ON.train_step(spec, st, xs, ys, batch_total=256)                      # warm-up (BLAS threads, page faults)
    t2 = []
    for _ in range(2):
        t0 = time.perf_counter()
        ON.train_step(spec, st, xs, ys, batch_total=256)
        t2.append(time.perf_counter() - t0)
    out['numpy_port'] = dict(value=round(8 / float(np.median(t2)), 3), unit='images/sec',
                             sample='NumPy(OpenBLAS) oracle of record, same step, B=8, median of 2 steps after 1 warm-up')
    # third stand-in: the product's own host code (graph, launch lists, optimizer) on libmcn_cpu.so — the C-ABI as plain C++ / OpenMP loops
    #  — in a child process (this one holds libmcn_hip.so)
    cpu_lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'myconvnet_amd', 'libmcn_cpu.so')
    if os.path.exists(cpu_lib):
        import subprocess
        code = ('import sys, time, numpy as np; sys.path.insert(0, %r); import myconvnet_amd as M\n'
                'B = 8\n'
                'm = M.ResNet50([224, 224, 3], 1000, batch_size=B, num_gpus=1, device="cpu", seed=0)\n'
                'o = M.MomentumOptimizer(m, None, None, base_learning_rate=0.1, momentum=0.9, steps_per_epoch=5000, num_epochs=90)\n'
                'r = np.random.default_rng(1234); m.feed(r.random((B, 224, 224, 3), dtype=np.float32), r.integers(0, 1000, B).astype(np.float32))\n'
                'o._step(None, fetch=False)\n'
                't = []\n'
                'for _ in range(2):\n'
                '    t0 = time.perf_counter(); o._step(None, fetch=False); t.append(time.perf_counter() - t0)\n'
                'print("IPS", B / float(np.median(t)))\n') % os.path.dirname(os.path.abspath(__file__))
        try:
            r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, MCN_LIB_PATH=cpu_lib, OMP_NUM_THREADS=str(threads)), stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, text=True, timeout=240)
            ips = [float(ln.split()[1]) for ln in r.stdout.splitlines() if ln.startswith('IPS')]
            if ips:
                out['loops_port'] = dict(value=round(ips[0], 3), unit='images/sec', cores=threads,
                                         sample='the product\'s host code on libmcn_cpu.so (naive C++ / OpenMP loops behind the same C-ABI), same step, B=8, median of 2 steps after 1 warm-up')
        except (subprocess.TimeoutExpired, OSError):
            pass
    return out


def executed_flop_per_image(dtype):
    """FLOP the MFMA pipes EXECUTE per image and training step.  fp32: the 13 stride-1 3x3 layers (forward, dgrad and wgrad) run Winograd
    F(2x2, 3x3) / F(3x3, 2x2) — 16 multiplications per 2x2 outputs instead of 36, times the tile cover of an odd map (7x7 is covered by 4x4
    tiles of 2x2: 64/49) — unless MCN_WINOGRAD=0.  The 2-byte types run every layer as a direct convolution.  (the headline
    utilisation used the direct count for both.)"""
    if dtype != 'fp32' or os.environ.get('MCN_WINOGRAD', '1') == '0':
        return float(TRAIN_FLOP_PER_IMAGE)
    saved = 0.0
    for h, c, n in ((56, 64, 3), (28, 128, 3), (14, 256, 5), (7, 512, 2)):
        direct = 2.0 * h * h * 9 * c * c * 3                              # fwd + dgrad + wgrad of one layer
        cover = float(((h + 1) // 2 * 2) ** 2) / (h * h)
        saved += n * direct * (1.0 - 16.0 / 36.0 * cover)
    return float(TRAIN_FLOP_PER_IMAGE) - saved


def verify_line(out):
    """Recompute every derived figure of a bench line from the line's OWN fields (a line must be self-consistent) and
    return the list of disagreements (empty = consistent).  main() asserts it on the line it is about to print; tests/test_bench_line.py runs
    it over the committed lines under profiles/."""
    bad = []

    def close(what, got, want, rel=2e-3, absol=0.0):
        if got is None or want is None:
            return
        if abs(float(got) - float(want)) > max(rel * abs(float(want)), absol):
            bad.append('{}: stored {} but its own fields give {:.6g}'.format(what, got, want))
    dt = {'f32': 'fp32', 'bf16': 'bf16', 'f16': 'fp16'}.get(out.get('dtype'))
    gb = out.get('config', {}).get('global_batch')
    if gb and out.get('ms_per_step'):
        close('value', out['value'], gb / out['ms_per_step'] * 1e3, rel=1e-3)
    n = max(int(out.get('n_gpus', 1)), 1)
    if dt and 'executed_flop_per_image' in out:
        close('e2e_mfma_frac', out.get('e2e_mfma_frac'), out['value'] / n * out['executed_flop_per_image'] / (PEAK_TFLOPS[dt] * 1e12), absol=1e-4)
        close('direct_equivalent_frac', out.get('direct_equivalent_frac'), out['value'] / n * out['train_flop_per_image'] / (PEAK_TFLOPS[dt] * 1e12), absol=1e-4)
    rf = out.get('roofline')
    if rf:
        close('roofline.frac', rf['frac'], rf['achieved'] / rf['peak'], absol=2e-4)
        if dt:
            close('roofline.peak', rf['peak'], PEAK_HBM_GBS if rf['unit'] == 'GB/s' else PEAK_TFLOPS[dt], rel=1e-9)
        kms = out.get('kernel_ms_per_step', {}).get(rf.get('kernel'))
        if kms:
            close('roofline.avg_launch_us x launches_per_step', rf['avg_launch_us'] * rf['launches_per_step'] * 1e-3, kms, rel=5e-3)
        e = out.get('roofline_by_kernel', {}).get(rf.get('kernel'))
        if e and rf['unit'] == 'TFLOP/s':
            close('roofline.achieved vs roofline_by_kernel.tflops', rf['achieved'], e['tflops'], absol=0.06)
        if e:
            close('roofline.algorithmic_bytes_per_launch', rf.get('algorithmic_bytes_per_launch'), e['algorithmic_bytes_per_launch'], rel=1e-6)
    for k, e in out.get('roofline_by_kernel', {}).items():
        if dt and 'frac' in e:
            close('roofline_by_kernel[{}].frac'.format(k), e['frac'], e['tflops'] / PEAK_TFLOPS[dt], absol=1e-3)
        if e.get('ms_per_step') and 'gbs' in e:
            close('roofline_by_kernel[{}].gbs'.format(k), e['gbs'], e['algorithmic_bytes_per_launch'] * e['launches_per_step'] / (e['ms_per_step'] * 1e-3) / 1e9,
                  rel=max(1e-2, 6e-4 / max(e['ms_per_step'], 1e-6)), absol=1.0)
        if 'hbm_frac' in e:
            close('roofline_by_kernel[{}].hbm_frac'.format(k), e['hbm_frac'], e['gbs'] / PEAK_HBM_GBS, absol=1e-3)
        if k.startswith('conv_wino') and 'tail_launches_per_step' not in e:
            bad.append('roofline_by_kernel[{}]: a Winograd entry must say how many tail launches its brackets include'.format(k))
    for k, e in out.get('hbm_by_call', {}).items():
        if 'gbs' in e:
            # (ms_per_step is stored to a microsecond: a 30 us call carries +-1.7 % of rounding)
            close('hbm_by_call[{}].gbs'.format(k), e['gbs'], e['algorithmic_bytes_per_call'] * e['calls_per_step'] / (e['ms_per_step'] * 1e-3) / 1e9,
                  rel=max(1e-2, 6e-4 / max(e['ms_per_step'], 1e-6)), absol=1.0)
            close('hbm_by_call[{}].hbm_frac'.format(k), e['hbm_frac'], e['gbs'] / PEAK_HBM_GBS, absol=1e-3)
        if 'traffic_per_step' in e:
            close('hbm_by_call[{}].traffic_gbs'.format(k), e['traffic_gbs'], e['traffic_per_step'] / (e['ms_per_step'] * 1e-3) / 1e9, rel=max(1e-2, 6e-4 / max(e['ms_per_step'], 1e-6)), absol=1.0)
            close('hbm_by_call[{}].traffic_hbm_frac'.format(k), e['traffic_hbm_frac'], e['traffic_gbs'] / PEAK_HBM_GBS, absol=1e-3)
    for key in ('bf16', 'fp16'):
        e = out.get(key)
        if e and 'e2e_mfma_frac' in e and gb:
            close(key + '.value', e['value'], gb / e['ms_per_step'] * 1e3, rel=1e-3)
            close(key + '.e2e_mfma_frac', e['e2e_mfma_frac'], e['value'] * TRAIN_FLOP_PER_IMAGE / (PEAK_TFLOPS[key] * 1e12), absol=1e-4)
    for key in ('efficientnet_b0_bf16', 'deeplabv3plus_bf16'):
        e = out.get(key)
        if e:
            close(key + '.value', e['value'], e['batch'] / e['ms_per_step'] * 1e3, rel=1e-3)
            if 'roofline' in e:
                close(key + '.roofline.frac', e['roofline']['frac'], e['roofline']['achieved'] / e['roofline']['peak'], absol=2e-4)
    cb = out.get('cpu_baseline')
    if cb is not None:
        for f in ('value', 'unit', 'cores', 'kind', 'sample'):
            if f not in cb:
                bad.append('cpu_baseline.{} missing'.format(f))
    return bad


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: N fresh child processes, one per GPU, started BEFORE this process makes any GPU call
    (a process that has initialised the GPU must never exec or fork into another one; this parent only ever waits).  Rank 0's stdout is
    this process's stdout, so exactly one JSON line comes out; the exit code is the worst child's."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                c = p.poll()
                if c is None:
                    continue
                pending.remove(p)
                if c != 0:
                    rc = rc or c
                    for q in pending:                   # one rank died: the others would wait for it in a collective forever
                        q.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def local_device():
    """GPU index of this rank: LOCAL_RANK (one process per GPU); MCN_BENCH_DEVICE overrides it (rehearsing several ranks on a
    one-GPU box with MCN_DIST_BACKEND=gloo)."""
    return int(os.environ.get('MCN_BENCH_DEVICE', os.environ.get('LOCAL_RANK', 0)))


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))          # no launcher: this process only starts and reaps the ranks (no GPU call above this line)
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    assert world == args.gpus, '--gpus {} but WORLD_SIZE={}'.format(args.gpus, world)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
    torch.cuda.set_device(local_device())
    if world > 1:
        from myconvnet_amd.dist import init_process_group
        init_process_group('cuda:{}'.format(local_device()))

    model, opt = build_model(args, args.dtype, world)
    dt = timed(opt, args.steps, args.warmup, world, args.autotune)
    ms = dt / args.steps * 1e3
    ips = args.batch * world * args.steps / dt
    if args.model != 'resnet50':
        # secondary workloads: no single-kernel roofline claim; per-call times from the instrumented pass
        title = {'efficientnet_b0': ('EfficientNet-B0', 224, 'configs[3]'), 'deeplabv3plus': ('DeepLabv3+ (ResNet-50 OS16 backbone)', 513, 'configs[4]')}[args.model]
        out = {'metric': 'images/sec {} {}x{} synthetic training step'.format(title[0], title[1], title[1]), 'value': round(ips, 2), 'unit': 'images/sec', 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': {'fp32': 'f32', 'bf16': 'bf16', 'fp16': 'f16'}[args.dtype], 'data': 'synthetic',
               'config': {'workload': '{} {} {}x{} synthetic training step (BASELINE {}), batch={}/GPU'.format(title[0], args.dtype, title[1], title[1], title[2], args.batch),
                          'global_batch': args.batch * world, 'parallelism': 'dp{}'.format(world), 'ema': not args.no_ema, 'fetch': False},
               'conv_macs_per_image': int(model.conv_macs), 'params': int(model.params)}
        if world == 1 and not args.no_roofline:
            out.update(streaming_roofline(model, args.dtype, args.batch, args.model, layers=args.layers, all_kernels=args.all_kernels))
        if rank == 0:
            print(json.dumps(out))
        return
    out = {
        'metric': 'images/sec ResNet-v1.5-50 224x224 synthetic training step',
        'value': round(ips, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': {'fp32': 'f32', 'bf16': 'bf16', 'fp16': 'f16'}[args.dtype], 'data': 'synthetic',
        'config': {'workload': 'ResNet-v1.5-50 {} 224x224 synthetic ImageNet-1k training step (fwd+bwd+Nesterov/L2/EMA), batch={}/GPU'
                   .format(args.dtype, args.batch), 'global_batch': args.batch * world, 'parallelism': 'dp{}'.format(world),
                   'ema': not args.no_ema, 'fetch': False},
        'train_flop_per_image': TRAIN_FLOP_PER_IMAGE,
        # MFMA-pipe utilisation of the whole step from the flop the kernels EXECUTE (Winograd layers: 16/36 x tile cover); the direct-convolution
        # equivalent of the same images/s rides beside it (the two used to be one number)
        'executed_flop_per_image': int(executed_flop_per_image(args.dtype)),
        'e2e_mfma_frac': round(ips / world * executed_flop_per_image(args.dtype) / (PEAK_TFLOPS[args.dtype] * 1e12), 4),
        'direct_equivalent_frac': round(ips / world * TRAIN_FLOP_PER_IMAGE / (PEAK_TFLOPS[args.dtype] * 1e12), 4),
    }
    if world == 1 and not args.no_roofline:
        # the reference's own _step also copies loss, Y_all and pred to the host every step (optimizers.py:590-594: a device sync per step);
        # `value` skips that (config.fetch = false) — this is the same step WITH it, over min(steps, 10) steps
        nf = min(args.steps, 10)
        if nf >= 5:                                       # (not in the one-step runs of the PMC passes: profiles/summarize.py counts their steps)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_steps(opt, nf, fetch=True)
            torch.cuda.synchronize()
            dtf = time.perf_counter() - t0
            out['fetch_true'] = {'value': round(args.batch * nf / dtf, 2), 'unit': 'images/sec', 'ms_per_step': round(dtf / nf * 1e3, 3), 'steps': nf,
                                 'what': 'the same step with the per-step device->host copy of loss / Y_all / pred (optimizers.py:590-594)'}
        table = instrumented_pass(model, args.dtype, layers=args.layers)
        bracket_us = table.pop('_bracket_us')[1] * 1e3
        convs = {k: v for k, v in table.items() if k.startswith('conv_gemm') or k.startswith('conv_wino')}
        dom = max(convs.items(), key=lambda kv: kv[1][1])                 # the kernel symbol with the most time per step
        name, (cnt, ms_k, flop, alg_bytes) = dom
        ach = flop / (ms_k * 1e-3) / 1e12
        # which roof bounds this symbol: its arithmetic intensity (FLOP per algorithmic byte) against the ridge point peak FLOP/s / peak B/s
        # (fp32 19.7 FLOP/B, bf16 312 FLOP/B: the 1x1 layers of the 2-byte types sit under the HBM roof, every fp32 conv under the MFMA roof)
        hbm_bound = alg_bytes > 0 and flop / alg_bytes < PEAK_TFLOPS[args.dtype] * 1e12 / (PEAK_HBM_GBS * 1e9)
        if hbm_bound:
            gbs = alg_bytes / (ms_k * 1e-3) / 1e9
            out['roofline'] = {'bound': 'hbm', 'kernel': name, 'launches_per_step': cnt, 'avg_launch_us': round(ms_k / cnt * 1e3, 2),
                               'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(gbs / PEAK_HBM_GBS, 4), 'traffic': None,
                               'mfma_tflops': round(ach, 2), 'event_bracket_overhead_us': round(bracket_us, 2)}
        else:
            out['roofline'] = {'bound': 'mfma', 'kernel': name, 'launches_per_step': cnt, 'avg_launch_us': round(ms_k / cnt * 1e3, 2),
                               'achieved': round(ach, 2), 'peak': PEAK_TFLOPS[args.dtype], 'unit': 'TFLOP/s',
                               'frac': round(ach / PEAK_TFLOPS[args.dtype], 4), 'traffic': None,
                               'event_bracket_overhead_us': round(bracket_us, 2)}
        out['roofline']['algorithmic_bytes_per_launch'] = int(alg_bytes / cnt)
        out['roofline'].update(pmc_traffic(name, args.dtype, args.batch))
        # the same figure for every conv kernel symbol (round 1's dominant symbol, the 3x3 forward AND dgrad, is several symbols now)
        out['roofline_by_kernel'] = {k: {'launches_per_step': v[0], 'ms_per_step': round(v[1], 3), 'tflops': round(v[2] / (v[1] * 1e-3) / 1e12, 1),
                                         'frac': round(v[2] / (v[1] * 1e-3) / 1e12 / PEAK_TFLOPS[args.dtype], 3),
                                         'gbs': round(v[3] / (v[1] * 1e-3) / 1e9, 0), 'algorithmic_bytes_per_launch': int(v[3] / max(v[0], 1)),
                                         'traffic': pmc_traffic(k, args.dtype, args.batch).get('traffic')}
                                     for k, v in sorted(convs.items(), key=lambda kv: -kv[1][1])[:8] if v[1] > 0}
        for k, d in WINO_DIRECT_FLOP.items():                            # (tflops / frac above: executed Winograd flop)
            if k in out['roofline_by_kernel']:
                e = out['roofline_by_kernel'][k]
                e['direct_equivalent_tflops'] = round(d / (table[k][1] * 1e-3) / 1e12, 1)
                # a Winograd entry is per C-ABI CALL: `launches_per_step` = launches of the body symbol = calls; the calls with a K-sliced tail add one
                # slice and one reduce launch each, inside the same event bracket (`tail_launches_per_step`, `tail_symbols`).  `traffic` = PMC bytes of
                # the body launch + the tail symbols' launches of an average call, so that it shares its denominator with `algorithmic_bytes_per_launch`
                tail = WINO_TAIL_LAUNCHES.get(k, 0)
                e['tail_launches_per_step'] = tail
                e['tail_symbols'] = sorted(WINO_TAIL_SYMBOLS.get(k, ()))
                if e.get('traffic') is not None and tail:
                    tb = [pmc_traffic(sym, args.dtype, args.batch).get('traffic') for sym in e['tail_symbols']]
                    if all(t is not None for t in tb):
                        e['traffic_body_launch'] = e['traffic']
                        e['traffic'] = int(e['traffic'] + sum(tb) * (tail / max(len(tb), 1)) / max(e['launches_per_step'], 1))
        # SURVEY 8(d) asks for BOTH roofs: the batch-norm calls of the step against the HBM roof (algorithmic bytes = every tensor once per pass that
        # needs it, _hbm_call_bytes; PMC traffic of the calls' kernel symbols where a call owns its symbols, else of the group)
        hb = {}
        for k in ('mcn_bn_fwd_train_fused', 'mcn_bn_fwd_train_fused_affskip', 'mcn_bn_fwd_train_fused_maxpool', 'mcn_bn_bwd', 'mcn_bn_bwd_from_partials', 'mcn_bn_bwd_maxpool'):
            v = table.get(k)
            if v and v[1] > 0 and v[3] > 0:
                hb[k] = {'calls_per_step': v[0], 'ms_per_step': round(v[1], 3), 'algorithmic_bytes_per_call': int(v[3] / max(v[0], 1)),
                         'gbs': round(v[3] / (v[1] * 1e-3) / 1e9, 0), 'hbm_frac': round(v[3] / (v[1] * 1e-3) / 1e9 / PEAK_HBM_GBS, 3)}
        grp = pmc_group_traffic(['bn_bwd_reduce_kernel', 'bn_bwd_apply_kernel', 'bn_bwd_finalize_kernel', 'bn_bwd_fold_partials_kernel'], args.dtype, args.batch)
        bw = [hb[k] for k in ('mcn_bn_bwd', 'mcn_bn_bwd_from_partials') if k in hb]
        if grp is not None and bw:
            ms_g = sum(e['ms_per_step'] for e in bw)
            hb['bn_backward_calls'] = {'what': 'mcn_bn_bwd + mcn_bn_bwd_from_partials (they share kernel symbols: PMC traffic is per group)', 'ms_per_step': round(ms_g, 3),
                                       'algorithmic_bytes_per_step': int(sum(e['algorithmic_bytes_per_call'] * e['calls_per_step'] for e in bw)), 'traffic_per_step': int(grp),
                                       'traffic_gbs': round(grp / (ms_g * 1e-3) / 1e9, 0), 'traffic_hbm_frac': round(grp / (ms_g * 1e-3) / 1e9 / PEAK_HBM_GBS, 3)}
        t_f = pmc_traffic('mcn_bn_fwd_train_fused', args.dtype, args.batch).get('traffic')
        if t_f is not None and 'mcn_bn_fwd_train_fused' in hb:
            e = hb['mcn_bn_fwd_train_fused']
            e['traffic'] = t_f
            e['traffic_gbs'] = round(t_f * e['calls_per_step'] / (e['ms_per_step'] * 1e-3) / 1e9, 0)
        if hb:
            out['hbm_by_call'] = hb
        tot = sum(v[1] for v in table.values())
        out['kernel_ms_per_step'] = {k: round(v[1], 3) for k, v in sorted(table.items(), key=lambda kv: -kv[1][1])[:(None if args.all_kernels else 12)]}
        out['kernel_ms_total'] = round(tot, 3)
        conv_ms = sum(v[1] for v in convs.values())
        conv_flop = sum(v[2] for v in convs.values())
        out['conv_tflops_all'] = round(conv_flop / (conv_ms * 1e-3) / 1e12, 2)
        if not args.no_secondary and args.dtype == 'fp32':
            del model, opt
            torch.cuda.empty_cache()
            import argparse as _a
            a2 = _a.Namespace(**vars(args))
            m2, o2 = build_model(a2, 'bf16', 1)
            dt2 = timed(o2, args.steps, args.warmup, 1, args.autotune)
            ips2 = args.batch * args.steps / dt2
            out['bf16'] = {'value': round(ips2, 2), 'unit': 'images/sec', 'ms_per_step': round(dt2 / args.steps * 1e3, 3),
                           'e2e_mfma_frac': round(ips2 * executed_flop_per_image('bf16') / (PEAK_TFLOPS['bf16'] * 1e12), 4)}
            del m2, o2
            torch.cuda.empty_cache()
            # the reference's own half precision: fp16 storage + loss scaling (same kernels, the f16 MFMA instead of the bf16 one)
            m2, o2 = build_model(a2, 'fp16', 1)
            dt2 = timed(o2, args.steps, args.warmup, 1, args.autotune)
            out['fp16'] = {'value': round(args.batch * args.steps / dt2, 2), 'unit': 'images/sec', 'ms_per_step': round(dt2 / args.steps * 1e3, 3), 'loss_scaling_factor': 128.0}
            del m2, o2
            torch.cuda.empty_cache()
            # the other single-GPU configurations of BASELINE.json as secondary keys (short runs; `--model ...` gives the full line)
            for key, mdl, bsz in (('efficientnet_b0_bf16', 'efficientnet_b0', 512), ('deeplabv3plus_bf16', 'deeplabv3plus', 16)):
                a3 = _a.Namespace(**vars(args))
                a3.model, a3.batch = mdl, bsz
                m3, o3 = build_model(a3, 'bf16', 1)
                dt3 = timed(o3, args.steps, args.warmup, 1, False)
                out[key] = {'value': round(bsz * args.steps / dt3, 2), 'unit': 'images/sec', 'ms_per_step': round(dt3 / args.steps * 1e3, 3), 'batch': bsz,
                            'config': 'BASELINE configs[3] (EfficientNet-B0, 224x224)' if mdl == 'efficientnet_b0' else 'BASELINE configs[4] on one GPU (DeepLabv3+, 513x513)'}
                sr = streaming_roofline(m3, 'bf16', bsz, mdl)                # (`bench.py --model ...` prints the full table)
                if 'roofline' in sr:
                    out[key]['roofline'] = sr['roofline']
                del m3, o3
                torch.cuda.empty_cache()
        if not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
    if rank == 0:
        problems = verify_line(out)
        assert not problems, 'bench line is not self-consistent: ' + '; '.join(problems)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        # hold every rank until rank 0 has printed; the measurement is over, so a peer that has already left this last barrier and closed its
        # sockets (gloo: "Connection closed by peer", seen once with 4 ranks on one GPU) must not turn a finished run into a failed one
        try:
            dist.barrier()
        except RuntimeError as e:
            print('rank {}: final barrier: {}'.format(rank, str(e).splitlines()[0]), file=sys.stderr)
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
