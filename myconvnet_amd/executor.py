"""Lowering of graph Nodes to C-ABI launch lists (forward, backward) — the only place that calls
libmcn_hip.  Every emit_* function cites the reference op it stands for in include/mcn.h."""
import ctypes
import os

import torch

from . import _ffi
from ._ffi import lib
from .graph import MCN_DT, TORCH_DT, Program, ptr


def _env_flag(name, default):
    """Experiment switch: MCN_<NAME>=0/1 overrides a lowering default (an explicit model kwarg still wins)."""
    e = os.environ.get(name)
    return default if e is None else bool(int(e))


class Lowering(object):
    """Emits the launch lists of one graph for one mode ('train' or 'eval')."""

    def __init__(self, graph, model, mode, loss_scale=1.0):
        self.g = graph
        self.model = model
        self.mode = mode
        self.train = mode == 'train'
        self.dt = MCN_DT[graph.dtype]
        self.loss_scale = float(loss_scale)
        self.fwd = Program()
        self.bwd = Program()
        self.prepack = Program()       # one launch: cast / re-pack every conv weight from its fp32 master (or EMA shadow)
        self.ws = None
        self.keep = []                 # objects that must outlive the programs (ctypes structs, scratch)
        # BN statistics in the conv epilogue: +2-3 % end to end for the 2-byte types; in fp32 the round-1 epilogue cost the
        # output-bound 1x1 convs what the skipped statistics pass saved, with the branch-free packed-math epilogue of round 2 it
        # nets +0.7 % (71.2 vs 71.8 ms per step, two A/B pairs on one box) and is on for every dtype
        self.fuse_bn_stats = bool(model._parameters.get('fuse_bn_stats', _env_flag('MCN_FUSE_BN_STATS', True)))
        # (+1 % in bf16, +0.6 % in fp32 since the accumulate epilogue issues its loads in one batch)
        self.defer_dskip = bool(model._parameters.get('defer_dskip', _env_flag('MCN_DEFER_DSKIP', True)))
        self.lazy_grad = {}            # tensor id -> (dy_block ptr, mask ptr): a gradient contribution that is applied by the consumer
        self.written = set()           # tensor ids whose .grad already holds a contribution
        self.fused_pools = set()       # ids of max-pool nodes whose forward runs inside the BN apply pass in front of them
        self.fused_gaps = set()        # ids of global-average nodes whose forward runs inside the BN apply pass in front of them
        self.dw_wgrad_side = bool(model._parameters.get('dw_wgrad_side', _env_flag('MCN_DW_WGRAD_SIDE', True)))     # depthwise wgrad on the wgrad stream
        self.fuse_bn_gap = bool(model._parameters.get('fuse_bn_gap', _env_flag('MCN_FUSE_BN_GAP', True)))
        self.fuse_se_fwd = bool(model._parameters.get('fuse_se_fwd', _env_flag('MCN_FUSE_SE_FWD', True)))          # ... and its BN + swish output never stored (needs fuse_se_sums)
        self.se_elided = set()          # ids of BN-output tensors that are not materialised (rebuilt inside mcn_bn_act_scale_fwd)
        self.fuse_se = os.environ.get('MCN_FUSE_SE', '1') != '0'        # (read once: the forward decision and the backward route must agree)
        self.fuse_se_sums = bool(model._parameters.get('fuse_se_sums', _env_flag('MCN_FUSE_SE_SUMS', True)))        # squeeze-excite: BN-backward sums from the channel scale's reduction pass
        self.pool_routes = {}          # BN-output tensor id -> max-pool node whose gradient that BN's backward routes itself
        self.se_routes = {}            # BN-output tensor id -> {dy, m, dgap, gap}: squeeze-excite gradient composed inside that BN's backward
        self.aff_skips = {}            # shortcut-BN output tensor id -> (its input tensor, its affine [2][C]): applied by the consumer BN
        # BN-backward sums (sum dy', sum dy' * x) in the epilogue of the dgrad that produces that BN's output gradient (the two inner BNs
        # of a bottleneck).  Measured on one box, B=256: the BN backward's kernels lose 0.6 ms (bf16) / 1.0 ms (fp32) and the dgrads gain
        # nothing, yet with the wgrad on its side stream the step is 68.83 vs 68.92 ms in fp32 and 22.35 vs 22.25 ms in bf16 — the reduction
        # pass is HBM-bound and was already running under the MFMA-bound wgrad (serial launch order: 71.28 vs 71.79 ms fp32, 22.44 vs
        # 22.55 ms bf16).  Re-measured on the round's final tree (streaming BN loads, XCD-aware wgrad order): fp32 68.42-68.46 vs 68.97-69.28 ms,
        # bf16 21.41-21.48 vs 21.32-21.35 ms.  On for fp32, off for the 2-byte types.
        self.fuse_bn_bwd_red = bool(model._parameters.get('fuse_bn_bwd_red', _env_flag('MCN_FUSE_BN_BWD_RED', graph.dtype == 'float32')))
        # ... and of a residual unit's OUTPUT BN in the epilogue of the launch that completes its gradient (round 4: the next unit's first dgrad with
        # the masked fan-in, mcn_conv2d_dgrad_addmasked_bnred): 12 of ResNet-50's 16 unit-output BNs (36 % of all BN elements) lose their
        # reduction pass over (dy, x).  MCN_FUSE_BN_OUT_RED=0 / fuse_bn_out_red=False: off.
        self.fuse_bn_out_red = bool(model._parameters.get('fuse_bn_out_red', _env_flag('MCN_FUSE_BN_OUT_RED', True)))
        self.scratch = {}

    # ---- helpers ----------------------------------------------------------------------------------
    def vptr(self, var):
        """Pointer to a variable's value: master in training, EMA shadow in evaluation
        (reference convnet.py:1406-1408, 1456-1458, 1872-1876)."""
        if var is None:
            return 0
        return var.data.data_ptr() if self.train else var.ema.data_ptr()

    def workspace_bytes(self):
        need = 4096
        for n in self.g.nodes:
            if n.op == 'conv':
                for op in (_ffi.CONV_FWD, _ffi.CONV_DGRAD, _ffi.CONV_WGRAD):
                    gm = self.op_geom(n, op)
                    keep = gm.tile
                    for t in range(lib.mcn_conv2d_tile_candidates(op) + 1):       # any tile the autotuner may pick
                        gm.tile = t
                        need = max(need, lib.mcn_conv2d_workspace_bytes(op, ctypes.byref(gm), self.dt))
                    gm.tile = keep
            elif n.op == 'dwconv':
                need = max(need, lib.mcn_dwconv2d_workspace_bytes(ctypes.byref(n.attrs['geom']), self.dt))
            elif n.op == 'bn':
                x = n.inputs[0]
                need = max(need, lib.mcn_bn_workspace_bytes(x.numel // x.shape[-1], x.shape[-1]))
            elif n.op == 'fc':
                x = n.inputs[0]
                need = max(need, lib.mcn_fc_workspace_bytes(x.shape[0], x.shape[1], n.outputs[0].shape[1], self.dt))
            elif n.op == 'biasadd':
                x = n.inputs[0]
                need = max(need, lib.mcn_bias_grad_workspace_bytes(x.numel // x.shape[-1], x.shape[-1]))
        return int(need)

    def op_geom(self, n, op):
        """Each of fwd / dgrad / wgrad owns a copy of the node's geometry: the tile hint is tuned per op."""
        key = 'geom_op'
        if key not in n.attrs:
            n.attrs[key] = {}
        d = n.attrs[key]
        if op not in d:
            src = n.attrs['geom']
            d[op] = _ffi.ConvGeom(*[getattr(src, f) for f, _ in _ffi.ConvGeom._fields_])
            d[op]._flop_scale = n.attrs.get('flop_scale', 1.0)       # (pixel-pair form: real MACs per MAC of the launched geometry; bench.py)
        return d[op]

    def se_sums_buffer(self, n):
        """fp32 [N * HS][5][C] per-image-slice sums of mcn_channel_scale_bwd_dm_bnsums (mcn_se_bwd_sums_floats() of them), shared by the squeeze-excite blocks of one size (written and consumed
        inside one block's backward, on the main stream)"""
        k = ('se_sums', n)                                     # (one buffer per size: the launch lists hold raw pointers)
        if k not in self.scratch:
            self.scratch[k] = torch.zeros(n, dtype=torch.float32, device=self.g.device)
        return self.scratch[k]

    def scratch_like(self, t, key):
        k = (key, t.shape, t.dtype)
        if k not in self.scratch:
            self.scratch[k] = torch.zeros(t.shape, dtype=TORCH_DT[t.dtype], device=self.g.device)
        return self.scratch[k]

    def contribute(self, t, produce):
        """Route a gradient contribution into t.grad.  `produce(dst_ptr, accumulate)` emits the launches;
        ops that cannot accumulate natively call contribute_via_scratch instead."""
        if not t.needs_grad:
            return
        first = t.id not in self.written
        self.written.add(t.id)
        produce(t.grad.data_ptr(), 0 if first else 1)

    def contribute_via_scratch(self, t, produce_write):
        """For ops that can only overwrite: first contribution writes t.grad directly, later ones go through
        a scratch buffer + mcn_accumulate."""
        if not t.needs_grad:
            return
        if t.id not in self.written:
            self.written.add(t.id)
            produce_write(t.grad.data_ptr())
        else:
            s = self.scratch_like(t, 'acc')
            produce_write(s.data_ptr())
            self.bwd.add(lib.mcn_accumulate, t.grad.data_ptr(), s.data_ptr(), t.grad.numel(), MCN_DT[t.dtype])

    # ---- driver ---------------------------------------------------------------------------------------
    def lower(self):
        ws_bytes = self.workspace_bytes()
        self.ws = torch.zeros(ws_bytes // 4 + 64, dtype=torch.float32, device=self.g.device)
        self.ws_ptr, self.ws_bytes = self.ws.data_ptr(), self.ws.numel() * 4
        # wgrad runs on a side stream (own workspace): it is MFMA-bound and independent of the dgrad -> BN-backward chain,
        # whose BN kernels are HBM-bound, so the two overlap on the chip; the optimizer joins the side stream
        self.overlap_wgrad = self.train and bool(self.model._parameters.get('overlap_wgrad', True)) and self.g.device.type == 'cuda'
        if self.overlap_wgrad:
            self.ws2 = torch.zeros(ws_bytes // 4 + 64, dtype=torch.float32, device=self.g.device)
            side_cus = int(os.environ.get('MCN_SIDE_CUS', '0'))                 # experiment: the wgrad stream on a subset of the CUs
            if side_cus > 0:
                from .graph import masked_stream
                self.bwd.side_stream = masked_stream(self.g.device, side_cus, int(os.environ.get('MCN_SIDE_CU0', '0')))
            else:
                self.bwd.side_stream = torch.cuda.Stream(device=self.g.device)      # (stream priorities were measured: no effect)
        self.plan_packed_weights()
        for n in self.g.nodes:
            getattr(self, 'fwd_' + n.op)(n)
        if self.train:
            for n in reversed(self.g.nodes):
                f = getattr(self, 'bwd_' + n.op, None)
                # (no gradient reaches a node all of whose variables upstream are frozen: blocks_to_train)
                if f is not None and any(t.needs_grad for t in n.outputs):
                    f(n)
            assert not self.lazy_grad, 'deferred residual gradients were not consumed: {}'.format(list(self.lazy_grad))
        if self.train and os.environ.get('MCN_PROBE_FWD_BN_SIDE') == '1' and self.g.device.type == 'cuda':
            # TIMING PROBE (wrong results): every forward BN call on a second stream and nobody waits for it — the upper bound of what hiding
            # the forward's bandwidth-bound apply passes under the next conv could buy
            for i, (fn, _) in enumerate(self.fwd.calls):
                if getattr(fn, '__name__', '').startswith('mcn_bn_fwd_train'):
                    self.fwd.side[i] = None
            self.fwd.side_stream = torch.cuda.Stream(device=self.g.device)
        return self

    def plan_packed_weights(self):
        """One packed operand per (conv, op) kept across the step, refreshed by ONE batched launch (mcn_conv2d_pack_run)
        instead of a pack kernel in front of every conv launch: the reference's per-use cast of the fp32 master
        (convnet.py:1421-1422) happens once per step here.  The buffers are shared by the train and eval lowerings
        (each runs its own table — masters vs EMA shadows — before its forward pass)."""
        jobs, offs, total = [], {}, 0
        for n in self.g.nodes:
            if n.op != 'conv':
                continue
            gm = n.attrs['geom']
            for op in ((_ffi.CONV_FWD, _ffi.CONV_DGRAD) if (self.train and n.inputs[0].needs_grad) else (_ffi.CONV_FWD,)):
                nb = int(lib.mcn_conv2d_packed_bytes(op, ctypes.byref(gm), self.dt))
                if nb:
                    offs[(id(n), op)] = total
                    total += (nb + 255) // 256 * 256
        shared = getattr(self.model, '_packed_store', None)
        if shared is None or shared.numel() < total:
            shared = torch.zeros(max(total, 256), dtype=torch.uint8, device=self.g.device)
            self.model._packed_store = shared
        self.packed_ptr = {k: shared.data_ptr() + o for k, o in offs.items()}
        for n in self.g.nodes:
            if n.op != 'conv':
                continue
            if 'geom_orig' in n.attrs:
                # pixel-pair form (convnet._plan_pixel_pairs): the paired filter is rebuilt from the master (or its EMA shadow) in front
                # of the batched pack
                self.prepack.add(lib.mcn_conv2d_pair_weights, self.vptr(n.attrs['w']), self.pair_buffers(n)[0].data_ptr(), ctypes.byref(n.attrs['geom_orig']), self.dt)
            for op in (_ffi.CONV_FWD, _ffi.CONV_DGRAD):
                if (id(n), op) in self.packed_ptr:
                    jobs.append(_ffi.PackJob(self.wsrc(n), self.packed_ptr[(id(n), op)], n.attrs['geom'], op, 0))
        if not jobs:
            return
        arr = (_ffi.PackJob * len(jobs))(*jobs)
        nbytes = int(lib.mcn_conv2d_pack_table_bytes(arr, len(jobs)))
        host = (ctypes.c_char * nbytes)()
        ndesc = ctypes.c_int32(0)
        _ffi.check(lib.mcn_conv2d_pack_table_build(arr, len(jobs), self.dt, ctypes.cast(host, ctypes.c_void_p), nbytes, ctypes.byref(ndesc)))
        self.pack_table = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(self.g.device)
        self.prepack.add(lib.mcn_conv2d_pack_run, self.pack_table.data_ptr(), ndesc.value, self.dt)

    def wp(self, n, op):
        return self.packed_ptr.get((id(n), op), 0)

    def pair_buffers(self, n):
        """(paired filter, paired filter gradient) of a conv in pixel-pair form: fp32 [KH][KW'][8][Cout], shared by the lowerings"""
        if 'pair_buf' not in n.attrs:
            pg = n.attrs['geom']
            size = pg.KH * pg.KW * pg.Cin * pg.Cout
            n.attrs['pair_buf'] = (torch.zeros(size, dtype=torch.float32, device=self.g.device), torch.zeros(size, dtype=torch.float32, device=self.g.device))
        return n.attrs['pair_buf']

    def wsrc(self, n):
        """fp32 HWIO filter the conv entry points read for node n: the variable (master / EMA shadow), or its paired form"""
        if 'geom_orig' in n.attrs:
            return self.pair_buffers(n)[0].data_ptr()
        return self.vptr(n.attrs['w'])

    def autotune(self, reps=3):
        """Measure, don't guess: time every tile candidate of every conv launch of this lowering on the GPU (HIP events
        on the launch stream, data already in the buffers) and pin the fastest through mcn_conv_geom.tile.  The result
        of a conv does not depend on the tile except for the fp32 summation order of the split wgrad."""
        names = {'mcn_conv2d_fwd': _ffi.CONV_FWD, 'mcn_conv2d_fwd_bnstats': _ffi.CONV_FWD, 'mcn_conv2d_dgrad': _ffi.CONV_DGRAD, 'mcn_conv2d_dgrad_addmasked': _ffi.CONV_DGRAD,
                 'mcn_conv2d_dgrad_bnred': _ffi.CONV_DGRAD, 'mcn_conv2d_dgrad_addmasked_bnred': _ffi.CONV_DGRAD, 'mcn_conv2d_wgrad': _ffi.CONV_WGRAD}
        sp = torch.cuda.current_stream(self.g.device).cuda_stream
        self.prepack.run(sp)
        chosen = {}
        for fn, args in self.fwd.calls + self.bwd.calls:
            op = names.get(getattr(fn, '__name__', ''))
            if op is None:
                continue
            gm = [a for a in args if hasattr(a, '_obj')][0]._obj
            if id(gm) in chosen:
                continue
            args[-1] = sp
            best, best_t = 0, None
            for cand in range(lib.mcn_conv2d_tile_candidates(op) + 1):
                gm.tile = cand
                _ffi.check(fn(*args))                       # warm
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    _ffi.check(fn(*args))
                e1.record()
                e1.synchronize()
                t = e0.elapsed_time(e1)
                if best_t is None or t < best_t * 0.985:    # keep the heuristic unless a candidate is clearly faster
                    best, best_t = cand, t
            gm.tile = best
            chosen[id(gm)] = best
        self.sync_partial_rows()
        return chosen

    def sync_partial_rows(self):
        """After a tile was pinned (autotune, or mcn_conv_geom.tile set by hand): the partial-row counts that the BN calls behind a fused conv / dgrad
        were emitted with follow the tile."""
        for fn, args in self.fwd.calls:                     # the partial-row count of a fused conv -> BN pair follows the tile
            name = getattr(fn, '__name__', '')
            if name in ('mcn_bn_fwd_train_fused', 'mcn_bn_fwd_train_fused_maxpool', 'mcn_bn_fwd_train_fused_affskip', 'mcn_bn_fwd_train_fused_stats'):
                ip = 0 if name == 'mcn_bn_fwd_train_fused_stats' else 1          # position of (partials, rows, rows_per_partial) in the call
                for nd in self.g.nodes:
                    fs = nd.attrs.get('fused_stats') if nd.op == 'bn' else None
                    if fs is not None and fs[0].data_ptr() == args[ip]:
                        rpp = ctypes.c_int32(0)
                        args[ip + 1] = int(lib.mcn_conv2d_bnstats_rows(ctypes.byref(fs[1]), self.dt, ctypes.byref(rpp)))
                        args[ip + 2] = rpp.value
        for fn, args in self.bwd.calls:                     # ... and so does the row count of a dgrad -> BN-backward pair
            if getattr(fn, '__name__', '') == 'mcn_bn_bwd_from_partials':
                for nd in self.g.nodes:
                    br = nd.attrs.get('bwd_red') if nd.op == 'bn' else None
                    if br is not None and br[0].data_ptr() == args[7]:
                        args[8] = int(lib.mcn_conv2d_dgrad_bnred_rows(ctypes.byref(br[1]), self.dt))

    # ---- input / labels ---------------------------------------------------------------------------------
    def fwd_input(self, n):
        y = n.outputs[0]
        N, H, W, C = y.shape
        m = self.model
        src_layout = _ffi.NCHW if n.attrs['src_nchw'] else _ffi.NHWC
        self.fwd.add(lib.mcn_input_prep, m.X_in.data_ptr(), y.buf.data_ptr(), N, H, W, C, y.cs, float(n.attrs['image_mean']),
                     float(n.attrs['scale_factor']), src_layout, MCN_DT[y.dtype])

    def fwd_labels(self, n):
        y = n.outputs[0]
        if n.attrs.get('seg'):
            self.fwd.add(lib.mcn_one_hot_seg, self.model.Y_in.data_ptr(), y.buf.data_ptr(), y.numel // y.shape[-1], y.shape[-1])
        else:
            self.fwd.add(lib.mcn_one_hot, self.model.Y_in.data_ptr(), y.buf.data_ptr(), y.shape[0], y.shape[1])

    # ---- segmentation path: bilinear resize, channel concat (models/deeplabv3plus.py) -----------------------------------
    def _resize_args(self, n):
        x, y = n.inputs[0], n.outputs[0]
        N, H, W, C = x.shape
        return [N, H, W, C, y.shape[1], y.shape[2], 1 if n.attrs['align'] else 0, MCN_DT[x.dtype]]

    def fwd_resize(self, n):
        x, y = n.inputs[0], n.outputs[0]
        self.fwd.add(lib.mcn_resize_bilinear_fwd, x.buf.data_ptr(), y.buf.data_ptr(), *self._resize_args(n))

    def bwd_resize(self, n):
        x, y = n.inputs[0], n.outputs[0]
        self.contribute_via_scratch(x, lambda dst: self.bwd.add(lib.mcn_resize_bilinear_bwd, y.grad.data_ptr(), dst, *self._resize_args(n)))

    def fwd_concat(self, n):
        y = n.outputs[0]
        M, tot, off = y.numel // y.shape[-1], y.shape[-1], 0
        for t in n.inputs:
            self.fwd.add(lib.mcn_copy_channels, t.buf.data_ptr(), t.shape[-1], 0, y.buf.data_ptr(), tot, off, M, t.shape[-1], MCN_DT[y.dtype])
            off += t.shape[-1]

    def bwd_concat(self, n):
        y = n.outputs[0]
        M, tot, off = y.numel // y.shape[-1], y.shape[-1], 0
        for t in n.inputs:
            c = t.shape[-1]
            self.contribute_via_scratch(t, lambda dst, o=off, cc=c: self.bwd.add(lib.mcn_copy_channels, y.grad.data_ptr(), tot, o, dst, cc, 0, M, cc,
                                                                                MCN_DT[y.dtype]))
            off += c

    # ---- conv -----------------------------------------------------------------------------------------------
    def _bn_consumer(self, n):
        """The training-mode batch norm that is the only reader of this conv's output (its statistics can then be
        accumulated in the conv epilogue), else None."""
        y = n.outputs[0]
        if not (self.train and self.fuse_bn_stats and len(y.consumers) == 1):
            return None
        c = y.consumers[0]
        if c.op != 'bn' or not c.attrs['update'] or c.inputs[0] is not y or c not in self.g.nodes:
            return None
        return c

    def _red_conv(self, bn):
        """The conv that is the only reader of this training-mode BN + ReLU's output and whose dgrad can accumulate the BN's backward
        sums in its epilogue (mcn_conv2d_dgrad_bnred), else None."""
        a = bn.attrs
        x, y = bn.inputs[0], bn.outputs[0]
        if not (self.train and self.fuse_bn_bwd_red and a['update'] and a.get('act', 0) == _ffi.ACT_RELU and a.get('skip') is None
                and x.needs_grad and len(y.consumers) == 1):
            return None
        c = y.consumers[0]
        if c.op != 'conv' or c.inputs[0] is not y or c not in self.g.nodes or 'geom_orig' in c.attrs or x.shape[-1] % (4 if self.g.dtype == 'float32' else 8):
            return None
        if self._pool_consumer(bn) is not None or lib.mcn_conv2d_dgrad_bnred_rows(ctypes.byref(self.op_geom(c, _ffi.CONV_DGRAD)), self.dt) <= 0:
            return None
        return c

    def _out_red_bn(self, conv, x):
        """`conv` is about to write the complete gradient of x with mcn_conv2d_dgrad_addmasked (x's readers: this conv and the next residual
        add).  If x is itself the output of a training-mode BN + residual + ReLU whose own skip gradient will be deferred (so that its backward
        writes nothing but dx), return that BN: its backward sums can ride in this launch's epilogue (mcn_conv2d_dgrad_addmasked_bnred).
        fuse_bn_out_red / MCN_FUSE_BN_OUT_RED=0 switch it off."""
        bn = x.producer
        if not (self.train and self.fuse_bn_out_red and bn is not None and bn.op == 'bn' and bn in self.g.nodes and bn.outputs[0] is x):
            return None
        a = bn.attrs
        skip = a.get('skip')
        if not (a['update'] and a.get('act', 0) == _ffi.ACT_RELU and skip is not None and 'relu_mask' in a and bn.inputs[0].needs_grad
                and bn.inputs[0].shape[-1] % (4 if self.g.dtype == 'float32' else 8) == 0):
            return None
        if x.id in self.pool_routes or x.id in self.se_routes:
            return None
        if skip.needs_grad and not self._can_defer_dskip(bn, skip):
            return None                                     # its backward would have to materialise dskip: keep the two-pass form
        if int(lib.mcn_conv2d_dgrad_bnred_rows(ctypes.byref(self.op_geom(conv, _ffi.CONV_DGRAD)), self.dt)) <= 0:
            return None
        return bn

    def fwd_conv(self, n):
        x, y = n.inputs[0], n.outputs[0]
        gm = self.op_geom(n, _ffi.CONV_FWD)
        self.keep.append(gm)
        bn = self._bn_consumer(n)
        rows = int(lib.mcn_conv2d_bnstats_rows(ctypes.byref(gm), self.dt, None)) if bn is not None else 0
        if rows > 0 and y.shape[-1] % (4 if self.g.dtype == 'float32' else 8) == 0:
            cand = []
            keep = gm.tile
            for t in range(lib.mcn_conv2d_tile_candidates(_ffi.CONV_FWD) + 1):     # room for any tile the autotuner may pin
                gm.tile = t
                cand.append(int(lib.mcn_conv2d_bnstats_rows(ctypes.byref(gm), self.dt, None)))
            gm.tile = keep
            part = torch.zeros((max(cand), 4, y.shape[-1]), dtype=torch.float32, device=self.g.device)      # (4 planes: room for counted rows)
            bn.attrs['fused_stats'] = (part, gm)
            self.fwd.add(lib.mcn_conv2d_fwd_bnstats, x.buf.data_ptr(), self.wsrc(n), self.wp(n, _ffi.CONV_FWD), self.vptr(n.attrs.get('b')),
                         y.buf.data_ptr(), part.data_ptr(), ctypes.byref(gm), self.dt, _ffi.NHWC, self.ws_ptr, self.ws_bytes)
            return
        self.fwd.add(lib.mcn_conv2d_fwd, x.buf.data_ptr(), self.wsrc(n), self.wp(n, _ffi.CONV_FWD), self.vptr(n.attrs.get('b')), y.buf.data_ptr(),
                     ctypes.byref(gm), self.dt, _ffi.NHWC, self.ws_ptr, self.ws_bytes)

    def bwd_conv(self, n):
        x, y = n.inputs[0], n.outputs[0]
        gm = self.op_geom(n, _ffi.CONV_DGRAD)
        gw = self.op_geom(n, _ffi.CONV_WGRAD)
        w, b = n.attrs['w'], n.attrs.get('b')
        gs = 1.0 / self.loss_scale
        def emit_wgrad():
            if not w.trainable:
                return
            paired = 'geom_orig' in n.attrs
            dw_ptr = self.pair_buffers(n)[1].data_ptr() if paired else w.grad.data_ptr()     # pixel-pair form: gradient of the paired filter, gathered back below
            if self.overlap_wgrad:
                self.bwd.add_side(lib.mcn_conv2d_wgrad, x.buf.data_ptr(), y.grad.data_ptr(), dw_ptr,
                                  b.grad.data_ptr() if b is not None else 0, ctypes.byref(gw), gs, self.dt, _ffi.NHWC, self.ws2.data_ptr(),
                                  self.ws2.numel() * 4)
                if paired:
                    self.bwd.add_side(lib.mcn_conv2d_pair_wgrad_fold, dw_ptr, w.grad.data_ptr(), ctypes.byref(n.attrs['geom_orig']), self.dt)
            else:
                self.bwd.add(lib.mcn_conv2d_wgrad, x.buf.data_ptr(), y.grad.data_ptr(), dw_ptr,
                             b.grad.data_ptr() if b is not None else 0, ctypes.byref(gw), gs, self.dt, _ffi.NHWC, self.ws_ptr, self.ws_bytes)
                if paired:
                    self.bwd.add(lib.mcn_conv2d_pair_wgrad_fold, dw_ptr, w.grad.data_ptr(), ctypes.byref(n.attrs['geom_orig']), self.dt)
            self.bwd.mark(('grad_ready', tuple(v.name for v in (w, b) if v is not None)))

        # With the side stream the wgrad is enqueued AFTER the dgrad: its start event then sits behind the dgrad, so the
        # two MFMA-bound kernels do not split the machine and the wgrad runs beside the HBM-bound BN backward that follows (measured: +0.5 %,
        # within noise; halving the wgrad's occupancy to leave registers for the BN waves costs 5 %).
        late = self.overlap_wgrad
        if not late:
            emit_wgrad()
        lazy = self.lazy_grad.pop(x.id, None)
        if lazy is not None:
            # identity shortcut: dx = dgrad + [y_block > 0] * dy_block in the epilogue (first and only writer of x.grad so far)
            assert x.needs_grad and x.id not in self.written
            self.written.add(x.id)
            red = self._out_red_bn(n, x)
            if red is not None:
                # x = y_b = relu(bn(u_b) + skip_b) and this launch writes its COMPLETE gradient (this dgrad + the next unit's masked fan-in): the
                # backward sums of that unit's output BN ride in the same epilogue (bwd_bn then runs mcn_bn_bwd_from_partials: no reduction pass)
                pa = red.attrs
                keep, cand = gm.tile, []
                for t in range(lib.mcn_conv2d_tile_candidates(_ffi.CONV_DGRAD) + 1):        # room for any tile the autotuner may pin
                    gm.tile = t
                    cand.append(int(lib.mcn_conv2d_dgrad_bnred_rows(ctypes.byref(gm), self.dt)))
                gm.tile = keep
                pa['bwd_red'] = (torch.zeros((max(cand), 2, x.shape[-1]), dtype=torch.float32, device=self.g.device), gm, n)
                self.bwd.add(lib.mcn_conv2d_dgrad_addmasked_bnred, y.grad.data_ptr(), w.data.data_ptr(), self.wp(n, _ffi.CONV_DGRAD), x.grad.data_ptr(), lazy[0], lazy[1],
                             red.inputs[0].buf.data_ptr(), pa['relu_mask'].data_ptr(), pa['bwd_red'][0].data_ptr(), ctypes.byref(gm), self.dt, _ffi.NHWC,
                             self.ws_ptr, self.ws_bytes)
                pa['bwd_red_done'] = True
            else:
                self.bwd.add(lib.mcn_conv2d_dgrad_addmasked, y.grad.data_ptr(), w.data.data_ptr(), self.wp(n, _ffi.CONV_DGRAD), x.grad.data_ptr(), lazy[0], lazy[1],
                             ctypes.byref(gm), self.dt, _ffi.NHWC, self.ws_ptr, self.ws_bytes)
        elif x.needs_grad and x.producer is not None and x.producer.op == 'bn' and x.producer.attrs.get('bwd_red', (None, None, None))[2] is n:
            # x = relu(bn(u)), read by this conv only: the epilogue also leaves the BN backward's sums (bwd_bn then runs its apply pass only)
            pa = x.producer.attrs
            assert x.id not in self.written and gm is pa['bwd_red'][1]
            self.written.add(x.id)
            self.bwd.add(lib.mcn_conv2d_dgrad_bnred, y.grad.data_ptr(), w.data.data_ptr(), self.wp(n, _ffi.CONV_DGRAD), x.grad.data_ptr(), x.producer.inputs[0].buf.data_ptr(),
                         pa['relu_mask'].data_ptr(), pa['bwd_red'][0].data_ptr(), ctypes.byref(gm), self.dt, _ffi.NHWC, self.ws_ptr, self.ws_bytes)
            pa['bwd_red_done'] = True
        elif x.needs_grad:
            self.contribute(x, lambda dst, acc: self.bwd.add(lib.mcn_conv2d_dgrad, y.grad.data_ptr(), w.data.data_ptr(), self.wp(n, _ffi.CONV_DGRAD), dst,
                                                             ctypes.byref(gm), acc, self.dt, _ffi.NHWC, self.ws_ptr, self.ws_bytes))
        if late:
            emit_wgrad()

    # ---- depthwise conv / squeeze-excite (EfficientNet MBConv, models/efficientnet.py:126-197) ----------------------
    def fwd_dwconv(self, n):
        x, y = n.inputs[0], n.outputs[0]
        self.fwd.add(lib.mcn_dwconv2d_fwd, x.buf.data_ptr(), self.vptr(n.attrs['w']), y.buf.data_ptr(), ctypes.byref(n.attrs['geom']), self.dt)

    def bwd_dwconv(self, n):
        x, y = n.inputs[0], n.outputs[0]
        w = n.attrs['w']
        gm = n.attrs['geom']
        def emit_wgrad():
            if not w.trainable:
                return
            if self.overlap_wgrad and self.dw_wgrad_side:
                # (round 4) like the conv wgrads: on the side stream, behind the event at this point of the main list — its two reads (x, dy) and the
                # slab fold then run beside the dgrad -> BN-backward chain instead of in front of it
                self.bwd.add_side(lib.mcn_dwconv2d_wgrad, x.buf.data_ptr(), y.grad.data_ptr(), w.grad.data_ptr(), ctypes.byref(gm), 1.0 / self.loss_scale,
                                  self.dt, self.ws2.data_ptr(), self.ws2.numel() * 4)
            else:
                self.bwd.add(lib.mcn_dwconv2d_wgrad, x.buf.data_ptr(), y.grad.data_ptr(), w.grad.data_ptr(), ctypes.byref(gm), 1.0 / self.loss_scale,
                             self.dt, self.ws_ptr, self.ws_bytes)
            self.bwd.mark(('grad_ready', (w.name,)))
        late = self.overlap_wgrad and self.dw_wgrad_side
        if not late:
            emit_wgrad()
        if x.needs_grad:
            self.contribute(x, lambda dst, acc: self.bwd.add(lib.mcn_dwconv2d_dgrad, y.grad.data_ptr(), w.data.data_ptr(), dst, ctypes.byref(gm), acc,
                                                             self.dt))
        if late:
            emit_wgrad()

    # ---- depthwise channel multiplier / bias (convnet.py:1634-1650, 1678-1694) --------------------------------------
    def fwd_chrepeat(self, n):
        x, y = n.inputs[0], n.outputs[0]
        C = x.shape[-1]
        self.fwd.add(lib.mcn_channel_repeat_fwd, x.buf.data_ptr(), y.buf.data_ptr(), x.numel // C, C, n.attrs['mult'], MCN_DT[x.dtype])

    def bwd_chrepeat(self, n):
        x, y = n.inputs[0], n.outputs[0]
        C = x.shape[-1]
        self.contribute_via_scratch(x, lambda dst: self.bwd.add(lib.mcn_channel_repeat_bwd, y.grad.data_ptr(), dst, x.numel // C, C, n.attrs['mult'], MCN_DT[x.dtype]))

    def fwd_biasadd(self, n):
        x, y = n.inputs[0], n.outputs[0]
        C = x.shape[-1]
        if 'ones' not in n.attrs:
            n.attrs['ones'] = torch.ones(C, dtype=torch.float32, device=self.g.device)
        self.fwd.add(lib.mcn_channel_affine, x.buf.data_ptr(), n.attrs['ones'].data_ptr(), self.vptr(n.attrs['b']), y.buf.data_ptr(), x.numel // C, C, MCN_DT[x.dtype])

    def bwd_biasadd(self, n):
        x, y = n.inputs[0], n.outputs[0]
        C = x.shape[-1]
        b = n.attrs['b']
        dt = MCN_DT[x.dtype]
        if b.trainable:
            self.bwd.add(lib.mcn_bias_grad, y.grad.data_ptr(), b.grad.data_ptr(), x.numel // C, C, 1.0 / self.loss_scale, dt, self.ws_ptr, self.ws_bytes)
            self.bwd.mark(('grad_ready', (b.name,)))
        if x.needs_grad:                                     # the add passes the gradient through
            if x.id not in self.written:
                self.written.add(x.id)
                self.bwd.add(lib.mcn_cast, y.grad.data_ptr(), dt, x.grad.data_ptr(), dt, y.buf.numel())
            else:
                self.bwd.add(lib.mcn_accumulate, x.grad.data_ptr(), y.grad.data_ptr(), y.buf.numel(), dt)

    def fwd_chscale(self, n):
        x, m, y = n.inputs[0], n.inputs[1], n.outputs[0]
        N, H, W, C = x.shape
        if x.id in self.se_elided:
            # x = the BN + swish output of a squeeze-excite block that was never stored (_se_elide): rebuilt from the BN's input inside the scale pass
            bn = x.producer
            a, st = bn.attrs, bn.attrs['saved']
            self.fwd.add(lib.mcn_bn_act_scale_fwd, bn.inputs[0].buf.data_ptr(), self.vptr(a.get('gamma')), self.vptr(a.get('beta')), st['mean'].data_ptr(), st['invstd'].data_ptr(),
                         m.buf.data_ptr(), y.buf.data_ptr(), N, H * W, C, a.get('act', 0), MCN_DT[x.dtype])
            return
        self.fwd.add(lib.mcn_channel_scale_fwd, x.buf.data_ptr(), m.buf.data_ptr(), y.buf.data_ptr(), N, H * W, C, MCN_DT[x.dtype])

    def _se_elide(self, bn, gap):
        """(round 4) training-mode BN + swish whose output x_se is read only by the squeeze-excite squeeze — produced by this BN's own apply pass — and by the
        channel scale, with the backward taking dm and the BN's sums from the BN's INPUT (fuse_se_sums): x_se is never needed in memory.  The apply pass then
        leaves the pooled means only and the scale pass rebuilds x_se on the fly (mcn_bn_act_scale_fwd): one write of the expanded activations per MBConv
        block less.  Returns the channel-scale node or None.  MCN_FUSE_SE_FWD=0 switches it off."""
        y = bn.outputs[0]
        if not (self.train and self.fuse_se_sums and self.fuse_se_fwd and self.fuse_se):
            return None
        if bn.attrs.get('act', 0) != _ffi.ACT_SWISH or not bn.attrs.get('update') or bn.attrs.get('skip') is not None:
            return None
        others = [c for c in y.consumers if c is not gap]
        if len(others) != 1 or others[0].op != 'chscale' or others[0].inputs[0] is not y or len(y.shape) != 4 or y.shape[-1] % (4 if y.dtype == 'float32' else 8):
            return None
        n = others[0]
        if not y.needs_grad or not n.inputs[1].needs_grad or self.g.nodes.index(n) < self.g.nodes.index(bn):
            return None
        return n

    def _se_route(self, n):
        """squeeze-excite pattern around this channel scale: its input x is the output of a training-mode BN + swish and is read only by
        the SE branch's global average pool and by this node -> x's gradient is composed inside that BN's backward passes
        (mcn_bn_bwd_se) and never written.  Returns the pool node or None.  MCN_FUSE_SE=0 switches it off."""
        x, m = n.inputs[0], n.inputs[1]
        if 'se_route' in n.attrs:                           # decided once, at forward lowering (the BN output was elided on the strength of it)
            assert x.id not in self.written
            return n.attrs['se_route']
        if not self.fuse_se or not self.train or not x.needs_grad or not m.needs_grad or x.id in self.written:
            return None
        bn = x.producer
        if bn is None or bn.op != 'bn' or bn.attrs.get('act', 0) != _ffi.ACT_SWISH or not bn.attrs.get('update') or bn.attrs.get('skip') is not None:
            return None
        others = [c for c in x.consumers if c is not n]
        if len(others) != 1 or others[0].op != 'gap' or len(x.shape) != 4 or x.shape[-1] % (4 if x.dtype == 'float32' else 8):
            return None
        return others[0]

    def bwd_chscale(self, n):
        x, m, y = n.inputs[0], n.inputs[1], n.outputs[0]
        N, H, W, C = x.shape
        dt = MCN_DT[x.dtype]
        post = []
        gap = self._se_route(n)
        if gap is None and x.id in self.se_elided:
            raise RuntimeError('channel scale: the BN output was not stored in the forward pass but its backward route is not the squeeze-excite one')
        if gap is not None:
            # reduction half only; the BN in front composes round(round(dy * m) + dgap / HW) itself (bwd_gap adds its part of the route)
            if m.id not in self.written:
                self.written.add(m.id)
                dm = m.grad.data_ptr()
            else:
                sc = self.scratch_like(m, 'chscale_dm')
                dm = sc.data_ptr()
                post.append((m.grad.data_ptr(), sc.data_ptr(), m.grad.numel(), MCN_DT[m.dtype]))
            route = {'dy': y.grad.data_ptr(), 'm': m.buf.data_ptr(), 'gap': gap}
            if self.fuse_se_sums:
                # (round 4) the same reduction reads the BN's INPUT and also leaves the per-image sums from which the BN backward forms its own sums:
                # mcn_bn_bwd_se_sums then runs without its reduction pass over the expanded activations (MCN_FUSE_SE_SUMS=0: the two-pass form)
                bn = x.producer
                sa, st = bn.attrs, bn.attrs['saved']
                sums = self.se_sums_buffer(int(lib.mcn_se_bwd_sums_floats(N, H * W, C, dt)))
                self.bwd.add(lib.mcn_channel_scale_bwd_dm_bnsums, y.grad.data_ptr(), bn.inputs[0].buf.data_ptr(), self.vptr(sa.get('gamma')), self.vptr(sa.get('beta')),
                             st['mean'].data_ptr(), st['invstd'].data_ptr(), dm, sums.data_ptr(), N, H * W, C, dt)
                route['sums'] = sums.data_ptr()
            else:
                self.bwd.add(lib.mcn_channel_scale_bwd_dm, y.grad.data_ptr(), x.buf.data_ptr(), dm, N, H * W, C, dt)
            for a in post:
                self.bwd.add(lib.mcn_accumulate, *a)
            self.se_routes[x.id] = route
            return

        def target(t, key):
            if t.id not in self.written:
                self.written.add(t.id)
                return t.grad.data_ptr()
            s = self.scratch_like(t, key)
            post.append((t.grad.data_ptr(), s.data_ptr(), t.grad.numel(), MCN_DT[t.dtype]))
            return s.data_ptr()
        dx = target(x, 'chscale_dx') if x.needs_grad else self.scratch_like(x, 'chscale_dx').data_ptr()
        dm = target(m, 'chscale_dm') if m.needs_grad else self.scratch_like(m, 'chscale_dm').data_ptr()
        self.bwd.add(lib.mcn_channel_scale_bwd, y.grad.data_ptr(), x.buf.data_ptr(), m.buf.data_ptr(), dx, dm, N, H * W, C, dt)
        for a in post:
            self.bwd.add(lib.mcn_accumulate, *a)

    def fwd_act(self, n):
        x, y = n.inputs[0], n.outputs[0]
        self.fwd.add(lib.mcn_act_fwd_p, x.buf.data_ptr(), y.buf.data_ptr(), y.buf.numel(), n.attrs['kind'], float(n.attrs.get('param', 0.2)), MCN_DT[x.dtype])

    def bwd_act(self, n):
        x, y = n.inputs[0], n.outputs[0]
        self.contribute_via_scratch(x, lambda dst: self.bwd.add(lib.mcn_act_bwd_p, y.grad.data_ptr(), x.buf.data_ptr(), y.buf.data_ptr(), dst,
                                                                y.buf.numel(), n.attrs['kind'], float(n.attrs.get('param', 0.2)), MCN_DT[x.dtype]))

    def _mulmask_args(self, n):
        """dropout / stochastic depth through the channel-scale kernel: [N, HW, C] * mask[N, C] with, for the per-sample
        factor, the tensor viewed as [N, numel/N/ce, ce] and the factor replicated over one 16-byte chunk."""
        x = n.inputs[0]
        mask = n.attrs['mask']
        N = x.shape[0]
        cols = mask.shape[1]
        return mask.data_ptr(), N, x.numel // N // cols, cols

    def fwd_mulmask(self, n):
        x, y = n.inputs[0], n.outputs[0]
        dt = MCN_DT[x.dtype]
        if not self.train:                                           # evaluation: rate 0 (convnet.py:155-158, 2505)
            self.fwd.add(lib.mcn_cast, x.buf.data_ptr(), dt, y.buf.data_ptr(), dt, y.buf.numel())
            return
        mp, N, HW, C = self._mulmask_args(n)
        self.fwd.add(lib.mcn_channel_scale_fwd, x.buf.data_ptr(), mp, y.buf.data_ptr(), N, HW, C, dt)

    def bwd_mulmask(self, n):
        x, y = n.inputs[0], n.outputs[0]
        mp, N, HW, C = self._mulmask_args(n)
        self.contribute_via_scratch(x, lambda dst: self.bwd.add(lib.mcn_channel_scale_fwd, y.grad.data_ptr(), mp, dst, N, HW, C, MCN_DT[x.dtype]))

    # ---- batch norm --------------------------------------------------------------------------------------------
    def fwd_bn(self, n):
        x, y = n.inputs[0], n.outputs[0]
        a = n.attrs
        C = x.shape[-1]
        M = x.numel // C
        skip = a.get('skip')
        if self.train and a['update']:
            st = a['saved']
            single = self.model.world_size == 1
            a.pop('bwd_red', None)
            a.pop('bwd_red_done', None)
            # a BN with a fused residual cannot recompute its ReLU mask from x: keep [y > 0] as one byte per 16-byte chunk
            # (the backward then reads 1/16 of the bytes of y, twice)
            mask_ptr = 0
            red = self._red_conv(n)                   # (the dgrad epilogue that sums this BN's backward terms reads the same byte mask)
            if (skip is not None or red is not None) and a.get('act', 0) == _ffi.ACT_RELU and x.needs_grad:
                nb = int(lib.mcn_bn_relu_mask_bytes(M, C, MCN_DT[x.dtype]))
                if nb:
                    a['relu_mask'] = torch.zeros(nb, dtype=torch.uint8, device=self.g.device)
                    mask_ptr = a['relu_mask'].data_ptr()
                    if red is not None:
                        gd = self.op_geom(red, _ffi.CONV_DGRAD)
                        keep, cand = gd.tile, []
                        for t in range(lib.mcn_conv2d_tile_candidates(_ffi.CONV_DGRAD) + 1):     # room for any tile the autotuner may pin
                            gd.tile = t
                            cand.append(int(lib.mcn_conv2d_dgrad_bnred_rows(ctypes.byref(gd), self.dt)))
                        gd.tile = keep
                        a['bwd_red'] = (torch.zeros((max(cand), 2, C), dtype=torch.float32, device=self.g.device), gd, red)
            if 'fused_stats' in a:
                part, gm = a['fused_stats']
                rpp = ctypes.c_int32(0)
                rows = int(lib.mcn_conv2d_bnstats_rows(ctypes.byref(gm), self.dt, ctypes.byref(rpp)))
                aff = self.aff_skips.get(skip.id) if skip is not None else None
                if aff is not None:
                    # residual input = the output of a shortcut BN that ran statistics-only: fold its apply pass into this one
                    assert a.get('act', 0) == _ffi.ACT_RELU
                    self.fwd.add(lib.mcn_bn_fwd_train_fused_affskip, x.buf.data_ptr(), part.data_ptr(), rows, rpp.value, self.vptr(a['gamma']), self.vptr(a['beta']),
                                 aff[0].buf.data_ptr(), aff[1].data_ptr(), y.buf.data_ptr(), mask_ptr, st['mean'].data_ptr(), st['invstd'].data_ptr(), st['bmean'].data_ptr(),
                                 st['bvar'].data_ptr(), a['mu'].data.data_ptr() if single else 0, a['sigma'].data.data_ptr() if single else 0,
                                 float(a['momentum']), M, C, float(a['eps']), MCN_DT[x.dtype], self.ws_ptr, self.ws_bytes)
                    return
                if self._affine_consumer(n) is not None:
                    if 'aff_out' not in a:
                        a['aff_out'] = torch.zeros(2 * C, dtype=torch.float32, device=self.g.device)
                    self.aff_skips[y.id] = (x, a['aff_out'])
                    self.fwd.add(lib.mcn_bn_fwd_train_fused_stats, part.data_ptr(), rows, rpp.value, self.vptr(a['gamma']), self.vptr(a['beta']), st['mean'].data_ptr(),
                                 st['invstd'].data_ptr(), st['bmean'].data_ptr(), st['bvar'].data_ptr(), a['mu'].data.data_ptr() if single else 0,
                                 a['sigma'].data.data_ptr() if single else 0, float(a['momentum']), M, C, float(a['eps']), a['aff_out'].data_ptr(), self.ws_ptr, self.ws_bytes)
                    return
                pool = self._pool_consumer(n)
                if pool is not None:
                    # conv -> BN -> ReLU -> max-pool (the stem): finalize, then one pass that normalises, rectifies and pools; y is never
                    # written (its readers: the pool, here; the BN backward recomputes the ReLU mask from x)
                    if 'argmax' not in pool.attrs:
                        pool.attrs['argmax'] = torch.zeros(pool.outputs[0].shape, dtype=torch.int8, device=self.g.device)
                    self.fused_pools.add(id(pool))
                    N_, H_, W_ = x.shape[0], x.shape[1], x.shape[2]
                    self.fwd.add(lib.mcn_bn_fwd_train_fused_maxpool, x.buf.data_ptr(), part.data_ptr(), rows, rpp.value, self.vptr(a['gamma']), self.vptr(a['beta']),
                                 pool.outputs[0].buf.data_ptr(), pool.attrs['argmax'].data_ptr(), st['mean'].data_ptr(), st['invstd'].data_ptr(),
                                 st['bmean'].data_ptr(), st['bvar'].data_ptr(), a['mu'].data.data_ptr() if single else 0, a['sigma'].data.data_ptr() if single else 0,
                                 float(a['momentum']), *(self._pool_args(pool)[:4] + [float(a['eps'])] + self._pool_args(pool)[4:] + [MCN_DT[x.dtype], self.ws_ptr, self.ws_bytes]))
                    return
                self.fwd.add(lib.mcn_bn_fwd_train_fused, x.buf.data_ptr(), part.data_ptr(), rows, rpp.value,
                             self.vptr(a['gamma']), self.vptr(a['beta']), ptr(skip.buf) if skip else 0,
                             y.buf.data_ptr(), mask_ptr, st['mean'].data_ptr(), st['invstd'].data_ptr(), st['bmean'].data_ptr(), st['bvar'].data_ptr(),
                             a['mu'].data.data_ptr() if single else 0, a['sigma'].data.data_ptr() if single else 0,
                             float(a['momentum']), M, C, float(a['eps']), a.get('act', 0), MCN_DT[x.dtype], self.ws_ptr, self.ws_bytes)
                return
            gap = self._gap_consumer(n) if (skip is None and not mask_ptr) else None
            if gap is not None:
                # BN + swish in front of a squeeze-excite block: the apply pass also leaves the per-image channel means (the SE branch's
                # tf.reduce_mean, models/efficientnet.py:183) — fwd_gap then emits nothing
                self.fused_gaps.add(id(gap))
                y_ptr = y.buf.data_ptr()
                for c_ in y.consumers:                     # (a re-lowering decides afresh)
                    c_.attrs.pop('se_route', None)
                chs = self._se_elide(n, gap)
                if chs is not None:                        # the channel scale rebuilds this output from x: means only
                    self.se_elided.add(y.id)
                    chs.attrs['se_route'] = gap            # ... and its backward takes the squeeze-excite route: ONE decision for both halves
                    y_ptr = 0
                    if y.buf is not None and y.buf.is_floating_point():
                        y.buf.fill_(float('nan'))           # never written by the training passes: a stray reader must not see plausible zeros
                self.fwd.add(lib.mcn_bn_fwd_train_gap, x.buf.data_ptr(), self.vptr(a['gamma']), self.vptr(a['beta']), y_ptr, gap.outputs[0].buf.data_ptr(),
                             st['mean'].data_ptr(), st['invstd'].data_ptr(), st['bmean'].data_ptr(), st['bvar'].data_ptr(),
                             a['mu'].data.data_ptr() if single else 0, a['sigma'].data.data_ptr() if single else 0,
                             float(a['momentum']), x.shape[0], M // x.shape[0], C, float(a['eps']), a.get('act', 0), MCN_DT[x.dtype], self.ws_ptr, self.ws_bytes)
                return
            self.fwd.add(lib.mcn_bn_fwd_train, x.buf.data_ptr(), self.vptr(a['gamma']), self.vptr(a['beta']), ptr(skip.buf) if skip else 0,
                         y.buf.data_ptr(), mask_ptr, st['mean'].data_ptr(), st['invstd'].data_ptr(), st['bmean'].data_ptr(), st['bvar'].data_ptr(),
                         a['mu'].data.data_ptr() if single else 0, a['sigma'].data.data_ptr() if single else 0,
                         float(a['momentum']), M, C, float(a['eps']), a.get('act', 0), MCN_DT[x.dtype], self.ws_ptr, self.ws_bytes)
        else:
            mu, sg = a['mu'], a['sigma']
            mp = mu.data.data_ptr() if self.train else mu.ema.data_ptr()
            sp = sg.data.data_ptr() if self.train else sg.ema.data_ptr()
            self.fwd.add(lib.mcn_bn_fwd_infer, x.buf.data_ptr(), self.vptr(a['gamma']), self.vptr(a['beta']), mp, sp,
                         ptr(skip.buf) if skip else 0, y.buf.data_ptr(), M, C, float(a['eps']), a.get('act', 0), MCN_DT[x.dtype])

    def _can_defer_dskip(self, n, skip):
        """Could _defer_dskip hand this residual BN's skip gradient to the skip branch's other consumer?  1 = projection shortcut's BN backward,
        2 = the dgrad of the block's first conv, 0 = no.  No side effects (also asked ahead of time by _out_red_bn)."""
        a = n.attrs
        if 'relu_mask' not in a or not self.defer_dskip or skip.id in self.written or skip.id in self.lazy_grad:
            return 0
        others = [c for c in skip.consumers if c is not n]
        prod = skip.producer
        if not others and prod is not None and prod.op == 'bn' and prod.attrs.get('update') and not prod.attrs.get('act', 0) \
                and prod.attrs.get('skip') is None and self.train:
            return 1                                                        # projection shortcut: conv_skip -> bn -> (add)
        if len(others) == 1 and others[0].op == 'conv' and others[0].inputs[0] is skip and others[0] in self.g.nodes \
                and self.g.nodes.index(others[0]) < self.g.nodes.index(n) \
                and lib.mcn_conv2d_dgrad_addmasked_ok(ctypes.byref(self.op_geom(others[0], _ffi.CONV_DGRAD)), self.dt):
            return 2                                                        # identity shortcut: the block's conv_0 reads it
        return 0

    def _defer_dskip(self, n, skip, y):
        """Residual BN (y = relu(bn(x) + skip)) with a ReLU byte mask: instead of materialising dskip = [y > 0] * dy, let
        the skip branch's only other gradient consumer apply it — the dgrad of the block's first conv (identity shortcut:
        mcn_conv2d_dgrad_addmasked) or the projection shortcut's BN backward.  Returns True when deferred."""
        if not self._can_defer_dskip(n, skip):
            return False
        self.lazy_grad[skip.id] = (y.grad.data_ptr(), n.attrs['relu_mask'].data_ptr())
        return True

    def bwd_bn(self, n):
        x, y = n.inputs[0], n.outputs[0]
        a = n.attrs
        frozen = not a['update']                         # statistics are constants (convnet.py:1915-1923): affine gradient
        C = x.shape[-1]
        M = x.numel // C
        skip = a.get('skip')
        st = a['saved']
        gs = 1.0 / self.loss_scale
        dskip_ptr, post = 0, None
        if skip is not None and skip.needs_grad and self._defer_dskip(n, skip, y):
            pass                                         # the skip branch's consumer applies [y > 0] * dy itself (no dskip tensor)
        elif skip is not None and skip.needs_grad:
            if skip.id not in self.written:
                self.written.add(skip.id)
                dskip_ptr = skip.grad.data_ptr()
            else:
                s = self.scratch_like(skip, 'bn_dskip')
                dskip_ptr = s.data_ptr()
                post = (skip.grad.data_ptr(), s.data_ptr(), skip.grad.numel(), MCN_DT[skip.dtype])
        g, b = a['gamma'], a['beta']

        # ReLU without a fused residual: the mask is recomputed from x inside the kernel (y pointer = 0);
        # with a fused residual: from the byte mask of the forward pass
        yptr = y.buf.data_ptr() if (skip is not None or not a.get('act', 0)) else 0
        mptr = a['relu_mask'].data_ptr() if 'relu_mask' in a else 0
        dy_ptr, act = y.grad.data_ptr(), a.get('act', 0)
        lazy = self.lazy_grad.pop(y.id, None)
        if lazy is not None:
            # this BN produced the skip branch of a residual block: its output gradient is [y_block > 0] * dy_block, read
            # straight from the block's gradient and byte mask (mathematically a ReLU in front of this BN's output)
            assert y.id not in self.written and not act and skip is None
            dy_ptr, mptr, yptr, act = lazy[0], lazy[1], 0, _ffi.ACT_RELU

        def emit_frozen(dst):
            mu, sg = a['mu'], a['sigma']
            self.bwd.add(lib.mcn_bn_bwd_frozen, dy_ptr, x.buf.data_ptr(), y.buf.data_ptr(), self.vptr(g), self.vptr(b), mu.data.data_ptr(),
                         sg.data.data_ptr(), float(a['eps']), dst, dskip_ptr, g.grad.data_ptr() if g is not None and g.trainable else 0,
                         b.grad.data_ptr() if b is not None and b.trainable else 0, gs, M, C, act, MCN_DT[x.dtype],
                         self.ws_ptr, self.ws_bytes)

        route = self.pool_routes.pop(y.id, None)             # this BN's output gradient is a 3x3 / 2 max-pool's: routed inside the passes
        se = self.se_routes.pop(y.id, None)                  # ... or a squeeze-excite block's (channel scale + pooled branch)

        def emit(dst):
            if frozen:
                return emit_frozen(dst)
            if se is not None:
                assert act == _ffi.ACT_SWISH and skip is None and lazy is None and 'dgap' in se
                N_, HW_ = x.shape[0], M // x.shape[0]
                if 'sums' in se:
                    self.bwd.add(lib.mcn_bn_bwd_se_sums, se['dy'], se['m'], se['dgap'], x.buf.data_ptr(), self.vptr(g), self.vptr(b), st['mean'].data_ptr(), st['invstd'].data_ptr(),
                                 se['sums'], dst, g.grad.data_ptr() if g is not None and g.trainable else 0, b.grad.data_ptr() if b is not None and b.trainable else 0, gs,
                                 N_, HW_, C, MCN_DT[x.dtype], self.ws_ptr, self.ws_bytes)
                    return
                self.bwd.add(lib.mcn_bn_bwd_se, se['dy'], se['m'], se['dgap'], x.buf.data_ptr(), self.vptr(g), self.vptr(b), st['mean'].data_ptr(), st['invstd'].data_ptr(),
                             dst, g.grad.data_ptr() if g is not None and g.trainable else 0, b.grad.data_ptr() if b is not None and b.trainable else 0, gs,
                             N_, HW_, C, MCN_DT[x.dtype], self.ws_ptr, self.ws_bytes)
                return
            if route is not None:
                assert act == _ffi.ACT_RELU and skip is None and yptr == 0 and lazy is None
                self.bwd.add(lib.mcn_bn_bwd_maxpool, route.outputs[0].grad.data_ptr(), route.attrs['argmax'].data_ptr(), x.buf.data_ptr(), self.vptr(g), self.vptr(b),
                             st['mean'].data_ptr(), st['invstd'].data_ptr(), dst, g.grad.data_ptr() if g is not None and g.trainable else 0,
                             b.grad.data_ptr() if b is not None and b.trainable else 0, gs, *(self._pool_args(route) + [MCN_DT[x.dtype], self.ws_ptr, self.ws_bytes]))
                return
            if a.get('bwd_red_done'):
                # (internal BN + ReLU read by one conv: mcn_conv2d_dgrad_bnred; or a residual unit's output BN whose complete gradient came out of
                # mcn_conv2d_dgrad_addmasked_bnred — its own skip gradient is deferred or not needed, so only dx is written)
                assert act == _ffi.ACT_RELU and lazy is None and mptr and not dskip_ptr and post is None
                part, gd = a['bwd_red'][0], a['bwd_red'][1]
                self.bwd.add(lib.mcn_bn_bwd_from_partials, dy_ptr, x.buf.data_ptr(), mptr, self.vptr(g), self.vptr(b), st['mean'].data_ptr(), st['invstd'].data_ptr(),
                             part.data_ptr(), int(lib.mcn_conv2d_dgrad_bnred_rows(ctypes.byref(gd), self.dt)), dst,
                             g.grad.data_ptr() if g is not None and g.trainable else 0, b.grad.data_ptr() if b is not None and b.trainable else 0, gs, M, C,
                             MCN_DT[x.dtype], self.ws_ptr, self.ws_bytes)
                return
            self.bwd.add(lib.mcn_bn_bwd, dy_ptr, x.buf.data_ptr(), yptr, mptr, self.vptr(g), self.vptr(b), st['mean'].data_ptr(),
                         st['invstd'].data_ptr(), dst, dskip_ptr, g.grad.data_ptr() if g is not None and g.trainable else 0,
                         b.grad.data_ptr() if b is not None and b.trainable else 0, gs, M, C, act, MCN_DT[x.dtype],
                         self.ws_ptr, self.ws_bytes)
        self.contribute_via_scratch(x, emit)
        if post:
            self.bwd.add(lib.mcn_accumulate, *post)
        self.bwd.mark(('grad_ready', tuple(v.name for v in (g, b) if v is not None and v.trainable)))

    # ---- element-wise ----------------------------------------------------------------------------------------------
    def fwd_affine(self, n):
        x, y = n.inputs[0], n.outputs[0]
        if 'dev' not in n.attrs:
            n.attrs['dev'] = (torch.tensor(n.attrs['scale'], dtype=torch.float32, device=self.g.device),
                              torch.tensor(n.attrs['shift'], dtype=torch.float32, device=self.g.device))
        sc, sh = n.attrs['dev']
        self.fwd.add(lib.mcn_channel_affine, x.buf.data_ptr(), sc.data_ptr(), sh.data_ptr(), y.buf.data_ptr(), x.numel // x.cs, x.cs,
                     MCN_DT[x.dtype])

    def fwd_relu(self, n):
        x, y = n.inputs[0], n.outputs[0]
        self.fwd.add(lib.mcn_relu_fwd, x.buf.data_ptr(), y.buf.data_ptr(), y.buf.numel(), MCN_DT[x.dtype])

    def bwd_relu(self, n):
        x, y = n.inputs[0], n.outputs[0]
        self.contribute_via_scratch(x, lambda dst: self.bwd.add(lib.mcn_relu_bwd, y.grad.data_ptr(), y.buf.data_ptr(), dst, y.buf.numel(),
                                                                MCN_DT[x.dtype]))

    def fwd_add(self, n):
        a, b, y = n.inputs[0], n.inputs[1], n.outputs[0]
        self.fwd.add(lib.mcn_add_relu_fwd, a.buf.data_ptr(), b.buf.data_ptr(), y.buf.data_ptr(), y.buf.numel(), n.attrs.get('act', 0),
                     MCN_DT[y.dtype])

    def bwd_add(self, n):
        y = n.outputs[0]
        act = n.attrs.get('act', 0)
        dt = MCN_DT[y.dtype]
        if act:
            g = self.scratch_like(y, 'add_mask')
            self.bwd.add(lib.mcn_add_relu_bwd, y.grad.data_ptr(), y.buf.data_ptr(), g.data_ptr(), y.buf.numel(), act, dt)
            gp = g.data_ptr()
        else:
            gp = y.grad.data_ptr()
        for t in n.inputs[:2]:
            if not t.needs_grad:
                continue
            if t.id not in self.written:
                self.written.add(t.id)
                self.bwd.add(lib.mcn_cast, gp, dt, t.grad.data_ptr(), dt, y.buf.numel())
            else:
                self.bwd.add(lib.mcn_accumulate, t.grad.data_ptr(), gp, y.buf.numel(), dt)

    def fwd_cast(self, n):
        x, y = n.inputs[0], n.outputs[0]
        self.fwd.add(lib.mcn_cast, x.buf.data_ptr(), MCN_DT[x.dtype], y.buf.data_ptr(), MCN_DT[y.dtype], y.buf.numel())

    def bwd_cast(self, n):
        x, y = n.inputs[0], n.outputs[0]
        if not x.needs_grad:
            return
        if x.id in self.written:
            raise NotImplementedError('cast backward into an already-written gradient')
        self.written.add(x.id)
        self.bwd.add(lib.mcn_cast, y.grad.data_ptr(), MCN_DT[y.dtype], x.grad.data_ptr(), MCN_DT[x.dtype], y.buf.numel())

    # ---- pooling ---------------------------------------------------------------------------------------------------------
    def _pool_args(self, n):
        x, y = n.inputs[0], n.outputs[0]
        a = n.attrs
        N, H, W, C = x.shape
        return [N, H, W, C, a['kh'], a['kw'], a['sh'], a['sw'], a['pt'], a['pl'], y.shape[1], y.shape[2]]

    def _stats_fusable(self, bn):
        """will fwd_conv accumulate this training-mode BN's statistics in its producer's epilogue?"""
        x = bn.inputs[0]
        prod = x.producer
        if prod is None or prod.op != 'conv' or self._bn_consumer(prod) is not bn or x.shape[-1] % (4 if self.g.dtype == 'float32' else 8):
            return False
        return int(lib.mcn_conv2d_bnstats_rows(ctypes.byref(self.op_geom(prod, _ffi.CONV_FWD)), self.dt, None)) > 0

    def _affine_consumer(self, n):
        """projection shortcut: this training-mode BN (no activation, no residual) feeds ONLY the residual input of another
        training-mode BN + ReLU whose statistics also come from a conv epilogue -> that BN folds this one's apply pass into its own
        (mcn_bn_fwd_train_fused_stats / _affskip); returns it, or None.  MCN_FUSE_BN_SKIP=0 switches it off."""
        if os.environ.get('MCN_FUSE_BN_SKIP', '1') == '0' or not self.train:
            return None
        a, y = n.attrs, n.outputs[0]
        if a.get('act', 0) or a.get('skip') is not None or len(y.consumers) != 1:
            return None
        m = y.consumers[0]
        if m.op != 'bn' or m.attrs.get('skip') is not y or m.inputs[0] is y or m.attrs.get('act', 0) != _ffi.ACT_RELU or not m.attrs.get('update'):
            return None
        if y.shape != m.inputs[0].shape or not self._stats_fusable(m) or m not in self.g.nodes or self.g.nodes.index(m) < self.g.nodes.index(n):
            return None
        return m

    def _pool_consumer(self, n):
        """the max-pool node that is the only reader of this training-mode BN + ReLU's output (folded into the BN apply pass), or None"""
        if os.environ.get('MCN_FUSE_BN_POOL', '1') == '0' or not self.train:
            return None
        a, y = n.attrs, n.outputs[0]
        if a.get('act', 0) != _ffi.ACT_RELU or a.get('skip') is not None or len(y.consumers) != 1 or y.consumers[0].op != 'maxpool':
            return None
        if len(y.shape) != 4 or y is self.model.d.get('logits'):
            return None
        return y.consumers[0]

    def fwd_maxpool(self, n):
        x, y = n.inputs[0], n.outputs[0]
        if id(n) in self.fused_pools:                       # computed by the BN apply pass in front (mcn_bn_fwd_train_fused_maxpool)
            return
        if 'argmax' not in n.attrs:
            n.attrs['argmax'] = torch.zeros(y.shape, dtype=torch.int8, device=self.g.device)
        self.fwd.add(lib.mcn_maxpool_fwd, x.buf.data_ptr(), y.buf.data_ptr(), n.attrs['argmax'].data_ptr(), *(self._pool_args(n) + [MCN_DT[x.dtype]]))

    def bwd_maxpool(self, n):
        x, y = n.inputs[0], n.outputs[0]
        a = n.attrs
        if (id(n) in self.fused_pools and os.environ.get('MCN_FUSE_BN_POOL', '1') != '2' and x.needs_grad and (a['kh'], a['kw'], a['sh'], a['sw']) == (3, 3, 2, 2)
                and x.shape[-1] % (4 if x.dtype == 'float32' else 8) == 0 and x.id not in self.written and x.numel // x.shape[-1] < 0xffffffff):
            # the BN in front (its forward already ran the pool) routes the pooled gradient inside its two backward passes
            # (mcn_bn_bwd_maxpool): the full-resolution gradient is not written.  MCN_FUSE_BN_POOL=2: forward fusion only.
            self.written.add(x.id)
            self.pool_routes[x.id] = n
            return
        self.contribute_via_scratch(x, lambda dst: self.bwd.add(lib.mcn_maxpool_bwd, y.grad.data_ptr(), n.attrs['argmax'].data_ptr(), dst,
                                                                *(self._pool_args(n) + [MCN_DT[x.dtype]])))

    def fwd_avgpool(self, n):
        x, y = n.inputs[0], n.outputs[0]
        self.fwd.add(lib.mcn_avgpool_fwd, x.buf.data_ptr(), y.buf.data_ptr(), *(self._pool_args(n) + [MCN_DT[x.dtype]]))

    def bwd_avgpool(self, n):
        x, y = n.inputs[0], n.outputs[0]
        self.contribute_via_scratch(x, lambda dst: self.bwd.add(lib.mcn_avgpool_bwd, y.grad.data_ptr(), dst, *(self._pool_args(n) + [MCN_DT[x.dtype]])))

    def _gap_consumer(self, bn):
        """the global-average node that reads this training-mode BN's (4-D, dense) output, when the BN's apply pass can produce the pooled
        means itself (mcn_bn_fwd_train_gap: the squeeze-excite squeeze, round 4).  MCN_FUSE_BN_GAP=0 switches it off."""
        y = bn.outputs[0]
        if not self.train or not self.fuse_bn_gap or len(y.shape) != 4 or y.shape[0] > 65535:
            return None
        gaps = [c for c in y.consumers if c.op == 'gap' and c.inputs[0] is y and c in self.g.nodes]
        if len(gaps) != 1 or self.g.nodes.index(gaps[0]) < self.g.nodes.index(bn):
            return None
        return gaps[0]

    def fwd_stopgrad(self, n):
        pass                                                # the output shares its input's storage (graph.allocate): no launch

    def bwd_stopgrad(self, n):
        pass                                                # ... and no gradient flows back (tf.stop_gradient, models/deeplabv3plus.py:53)

    def fwd_gap(self, n):
        if id(n) in self.fused_gaps:                        # produced by the BN apply pass in front of it
            return
        x, y = n.inputs[0], n.outputs[0]
        N, H, W, C = x.shape
        self.fwd.add(lib.mcn_global_avgpool_fwd, x.buf.data_ptr(), y.buf.data_ptr(), N, H * W, C, MCN_DT[x.dtype])

    def bwd_gap(self, n):
        x, y = n.inputs[0], n.outputs[0]
        N, H, W, C = x.shape
        r = self.se_routes.get(x.id)
        if r is not None and r['gap'] is n:                 # squeeze-excite route: the BN backward reads this pooled gradient directly
            r['dgap'] = y.grad.data_ptr()
            self.written.add(x.id)
            return
        # (accumulating form when x.grad already holds the channel-scale branch's contribution: no scratch tensor + add pass)
        self.contribute(x, lambda dst, acc: self.bwd.add(lib.mcn_global_avgpool_bwd_acc if acc else lib.mcn_global_avgpool_bwd, y.grad.data_ptr(), dst,
                                                         N, H * W, C, MCN_DT[x.dtype]))

    # ---- fc -------------------------------------------------------------------------------------------------------------------
    def fwd_fc(self, n):
        x, y = n.inputs[0], n.outputs[0]
        B, In = x.shape
        Out = y.shape[1]
        self.fwd.add(lib.mcn_fc_fwd, x.buf.data_ptr(), self.vptr(n.attrs['w']), self.vptr(n.attrs.get('b')), y.buf.data_ptr(), B, In, Out,
                     MCN_DT[x.dtype], self.ws_ptr, self.ws_bytes)

    def bwd_fc(self, n):
        x, y = n.inputs[0], n.outputs[0]
        B, In = x.shape
        Out = y.shape[1]
        w, b = n.attrs['w'], n.attrs.get('b')
        gs = 1.0 / self.loss_scale
        dt = MCN_DT[x.dtype]
        if w.trainable:
            self.bwd.add(lib.mcn_fc_bwd, y.grad.data_ptr(), x.buf.data_ptr(), w.data.data_ptr(), 0, w.grad.data_ptr(),
                         b.grad.data_ptr() if b is not None else 0, gs, B, In, Out, dt, self.ws_ptr, self.ws_bytes)
            self.bwd.mark(('grad_ready', tuple(v.name for v in (w, b) if v is not None)))
        if x.needs_grad:
            self.contribute_via_scratch(x, lambda dst: self.bwd.add(lib.mcn_fc_bwd, y.grad.data_ptr(), x.buf.data_ptr(), w.data.data_ptr(), dst,
                                                                    0, 0, gs, B, In, Out, dt, self.ws_ptr, self.ws_bytes))

    # ---- softmax / loss ------------------------------------------------------------------------------------------------------------
    def fwd_loss(self, n):
        logits, onehot = n.inputs[0], n.inputs[1]
        a = n.attrs
        B, C = a.get('rows', logits.shape[0]), logits.shape[-1]
        m = self.model
        dl = logits.grad.data_ptr() if (self.train and logits.needs_grad) else 0
        if dl:
            self.written.add(logits.id)
        fg, sa = float(a.get('focal_gamma', 0.0)), float(a.get('sigmoid_focal_alpha', 0.0))
        if len(n.inputs) == 3 or ((fg > 0.0 or sa > 0.0) and a.get('per_pixel')):
            # SegNet label smoothing: raw one-hot map + its 5x5 average (segnet.py:117-122); focal factors on the per-pixel loss (convnet.py:581-592)
            assert a.get('per_pixel')
            self.fwd.add(lib.mcn_softmax_xent_rows_focal_fwd_bwd, logits.buf.data_ptr(), onehot.buf.data_ptr(), n.inputs[2].buf.data_ptr() if len(n.inputs) == 3 else 0,
                         ptr(a.get('class_w')), a['pred'].buf.data_ptr(), a['ce'].data_ptr(), a['coef'].data_ptr(), dl, a['loss'].data_ptr(), B, C,
                         float(a['label_smoothing']), self.loss_scale, fg, sa, self.ws_ptr, self.ws_bytes)
        elif fg > 0.0 or sa > 0.0:
            self.fwd.add(lib.mcn_softmax_xent_focal_fwd_bwd, logits.buf.data_ptr(), onehot.buf.data_ptr(), ptr(a.get('class_w')), a['pred'].buf.data_ptr(),
                         a['ce'].data_ptr(), a['coef'].data_ptr(), dl, a['loss'].data_ptr(), B, C, float(a['label_smoothing']), self.loss_scale, fg, sa)
        elif a.get('per_pixel'):
            self.fwd.add(lib.mcn_softmax_xent_rows_fwd_bwd, logits.buf.data_ptr(), onehot.buf.data_ptr(), ptr(a.get('class_w')), a['pred'].buf.data_ptr(),
                         a['ce'].data_ptr(), a['coef'].data_ptr(), dl, a['loss'].data_ptr(), B, C, float(a['label_smoothing']), self.loss_scale,
                         self.ws_ptr, self.ws_bytes)
        else:
            self.fwd.add(lib.mcn_softmax_xent_fwd_bwd, logits.buf.data_ptr(), onehot.buf.data_ptr(), ptr(a.get('class_w')), a['pred'].buf.data_ptr(),
                         a['ce'].data_ptr(), a['coef'].data_ptr(), dl, a['loss'].data_ptr(), B, C, float(a['label_smoothing']), self.loss_scale)
        nw = m.n_l2_elems
        if a['l2_reg'] > 0.0 and nw > 0:
            # the regulariser always reads the master variables (collection 'weight_variables', convnet.py:535)
            self.fwd.add(lib.mcn_l2_loss, m.store.data.data_ptr(), nw, float(a['l2_reg']), a['loss'].data_ptr(), self.ws_ptr, self.ws_bytes)
        if a.get('l1_reg', 0.0) > 0.0 and nw > 0:              # L1 term over the same variables (convnet.py:553-557); its gradient: optimizers.py
            self.fwd.add(lib.mcn_l1_loss, m.store.data.data_ptr(), nw, float(a['l1_reg']), a['loss'].data_ptr(), self.ws_ptr, self.ws_bytes)
