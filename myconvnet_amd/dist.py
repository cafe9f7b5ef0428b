"""Data parallelism: one process per GPU, RCCL over xGMI (torch.distributed 'nccl' backend IS RCCL on ROCm).

Replaces the reference's in-graph tower replication with parameter-server averaging (convnet.py:431-436,
optimizers.py:121-147):
  * gradients: all-reduce(sum) of contiguous ranges of the flat fp32 gradient buffer, bucketed in reverse layer
    order and enqueued as soon as the last wgrad of a bucket has been launched, so the collective runs beside the
    remaining backward kernels; the 1/N of the tower mean is folded into the optimizer kernel's grad_scale;
  * BN running statistics: one all-gather of every rank's (batch_mean, batch_var) and the reference's chained
    update running <- m*running + (1-m)*batch[k], k = 0..N-1 (convnet.py:1899-1909), identical on every rank;
  * loss: mean of the per-rank losses (convnet.py:510), for reporting only.
The bucket planner works on plain tensors so it is covered by gloo tests on CPU.
"""
import os

import torch
import torch.distributed as dist


def init_process_group(device=None):
    """Idempotent init from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun contract)."""
    if dist.is_initialized():
        return
    backend = 'nccl' if (device is not None and torch.device(device).type == 'cuda') else 'gloo'
    backend = os.environ.get('MCN_DIST_BACKEND', backend)        # 'gloo' lets several ranks share one GPU in tests
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    kw = {}
    if backend == 'nccl':
        kw['device_id'] = torch.device(device)
    dist.init_process_group(backend=backend, rank=int(os.environ.get('RANK', 0)), world_size=int(os.environ.get('WORLD_SIZE', 1)), **kw)


def all_gather_rows(t, group=None):
    """Concatenate every rank's `t` along axis 0 in rank order (the reference's tf.concat over towers, convnet.py:503-509)."""
    world = dist.get_world_size(group)
    t = t.contiguous()
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    if dist.get_backend(group) == 'nccl':
        dist.all_gather_into_tensor(out, t, group=group)
    else:
        dist.all_gather(list(out.chunk(world, dim=0)), t, group=group)
    return out


def plan_buckets(variables, ready_index, bucket_bytes):
    """variables: [(name, offset, padded_size)] of ONE flat buffer; ready_index: name -> backward call index after
    which the gradient is final.  Returns [(launch_index, [(start, end), ...])]: element ranges to all-reduce once
    the call at launch_index-1 has been enqueued.  Variables are grouped in completion order; each bucket's
    variables are merged into maximal contiguous ranges."""
    order = sorted(variables, key=lambda v: ready_index[v[0]])
    buckets, cur, cur_bytes = [], [], 0
    for v in order:
        cur.append(v)
        cur_bytes += v[2] * 4
        if cur_bytes >= bucket_bytes:
            buckets.append(cur)
            cur, cur_bytes = [], 0
    if cur:
        buckets.append(cur)
    out = []
    for b in buckets:
        idx = max(ready_index[v[0]] for v in b)
        spans = sorted((v[1], v[1] + v[2]) for v in b)
        merged = []
        for s, e in spans:
            if merged and merged[-1][1] == s:
                merged[-1][1] = e
            else:
                merged.append([s, e])
        out.append((idx, [tuple(m) for m in merged]))
    return out


class GradientReducer(object):
    """Bucketed, overlapped all-reduce of a flat gradient tensor."""

    def __init__(self, flat_grad, variables, ready_index, bucket_mb=25.0, group=None, side_stream=None):
        self.flat = flat_grad
        self.group = group
        self.side_stream = side_stream    # the stream the wgrad kernels run on (None: everything is on the current stream)
        self.plan = plan_buckets(variables, ready_index, int(bucket_mb * 1024 * 1024))
        self.works = []
        self._hooks = {}
        for idx, spans in self.plan:
            self._hooks.setdefault(idx, []).extend(spans)

    def hooks(self):
        def make(spans):
            def fire():
                if self.side_stream is not None:
                    # the bucket's gradients come from both streams (wgrad: side, BN / fc: main): order the collective
                    # after both without stalling the main stream's backward chain
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream())
                    self.side_stream.wait_event(ev)
                    with torch.cuda.stream(self.side_stream):
                        for s, e in spans:
                            self.works.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                    return
                for s, e in spans:
                    self.works.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return fire
        return {idx: make(spans) for idx, spans in self._hooks.items()}

    def finish(self):
        for w in self.works:
            w.wait()
        self.works = []

    def covered_elements(self):
        return sum(e - s for _, spans in self.plan for s, e in spans)


class DataParallel(object):
    def __init__(self, model, bucket_mb=25.0):
        self.model = model
        init_process_group(model.device)
        self.world = dist.get_world_size()
        assert self.world == model.world_size, 'WORLD_SIZE {} != num_gpus {}'.format(self.world, model.world_size)
        marks = model._train_low.bwd.marks
        ready = {}
        for label, idx in marks.items():
            if isinstance(label, tuple) and label[0] == 'grad_ready':
                for name in label[1]:
                    ready[name] = idx
        st = model.store
        # frozen variables (blocks_to_train) have no gradient: they stay out of the exchange
        variables = [(v.name, v.offset, (v.size + 3) // 4 * 4) for v in st.variables if v.trainable]
        missing = [v[0] for v in variables if v[0] not in ready]
        assert not missing, 'no backward completion point for {}'.format(missing[:3])
        self.reducer = GradientReducer(st.grad, variables, ready, bucket_mb, side_stream=model._train_low.bwd.side_stream)
        assert self.reducer.covered_elements() == sum(v[2] for v in variables)
        self.gathered_stats = torch.zeros((self.world, model.batch_stats.numel()), dtype=torch.float32, device=model.device)
        self._loss_tmp = torch.zeros(1, dtype=torch.float32, device=model.device)
        # identical initial state on every rank
        for t in (st.data, st.ema, st.accum, model.stats.data, model.stats.ema):
            dist.broadcast(t, src=0)

    def hooks(self):
        return self.reducer.hooks()

    def finish(self):
        self.reducer.finish()

    def reduce_all(self):
        """One blocking all-reduce of the whole flat gradient (used when the gradients must be complete before the
        exchange: per-tower clipping by global norm, optimizers.py:112-113)."""
        dist.all_reduce(self.model.store.grad, op=dist.ReduceOp.SUM)

    def gather_bn_stats(self):
        if dist.get_backend() == 'nccl':
            dist.all_gather_into_tensor(self.gathered_stats.view(-1), self.model.batch_stats)
        else:                                                    # gloo has no all_gather_into_tensor for device tensors
            dist.all_gather([self.gathered_stats[r] for r in range(self.world)], self.model.batch_stats)

    def gather_bn_stats_async(self, chain):
        """The all-gather of the ranks' batch statistics and the chained running update (`chain`: a Program) on a stream of their
        own, behind the forward pass: nothing in the backward pass reads the running statistics, so the collective's latency (a
        small message: one round trip over xGMI) stays off the main stream's critical path.  join_bn_stats() orders the main
        stream behind it — the optimizer step does that before its update, i.e. before anything of the next step."""
        if self.model.device.type != 'cuda':
            self.gather_bn_stats()
            chain.run(0)
            return
        if getattr(self, 'stats_stream', None) is None:
            self.stats_stream = torch.cuda.Stream(device=self.model.device)
            self._stats_fwd, self._stats_done = torch.cuda.Event(), torch.cuda.Event()
        self._stats_fwd.record(torch.cuda.current_stream())
        self.stats_stream.wait_event(self._stats_fwd)
        with torch.cuda.stream(self.stats_stream):
            self.gather_bn_stats()
            chain.run(self.stats_stream.cuda_stream)
            self._stats_done.record(self.stats_stream)
        self._stats_pending = True

    def join_bn_stats(self):
        if getattr(self, '_stats_pending', False):
            torch.cuda.current_stream().wait_event(self._stats_done)
            self._stats_pending = False

    def mean_scalar(self, t):
        self._loss_tmp.copy_(t.reshape(1))
        dist.all_reduce(self._loss_tmp, op=dist.ReduceOp.SUM)
        return float(self._loss_tmp.item()) / self.world
