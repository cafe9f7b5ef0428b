"""Static graph + launch-list executor (the stand-in for TF-1.x graph mode on this path).

The reference builds a TensorFlow graph once (ConvNet.build -> _build_model, convnet.py:134,474;
Optimizer._optimize_and_update, optimizers.py:89) and then replays it with session.run
(optimizers.py:590).  Here the block methods of ConvNet record Nodes on symbolic Tensors with
static shapes; compile() fuses BN+ReLU(+residual add), lays every activation / gradient out in
HBM once (288 GB makes re-use unnecessary), and emits flat lists of C-ABI launches for forward,
backward and update.  Running a step = walking those lists on one HIP stream (optionally captured
into a hipGraph), with no per-step allocation and no host synchronisation.
"""
import numpy as np
import torch

from . import _ffi

TORCH_DT = {'float32': torch.float32, 'bfloat16': torch.bfloat16, 'float16': torch.float16}
MCN_DT = {'float32': _ffi.F32, 'bfloat16': _ffi.BF16, 'float16': _ffi.F16}


def same_pads(in_size, k, s, d=1):
    """TF SAME: out = ceil(in/s); total = max((out-1)*s + (k-1)*d + 1 - in, 0); before = total//2."""
    out = -(-in_size // s)
    total = max((out - 1) * s + (k - 1) * d + 1 - in_size, 0)
    return total // 2, total - total // 2


def out_size(in_size, k, s, padding, d=1):
    if padding.upper() == 'SAME':
        return -(-in_size // s)
    return -(-(in_size - ((k - 1) * d + 1) + 1) // s)


class Tensor(object):
    """Symbolic activation.  `shape` is always NHWC (or [N, C]); `cs` is the physical channel stride."""
    _count = 0

    def __init__(self, shape, dtype, name='', channel_first=False):
        self.shape = tuple(int(s) for s in shape)
        self.dtype = dtype
        self.name = name
        self.cs = self.shape[-1]
        self.channel_first = channel_first
        self.producer = None
        self.consumers = []
        self.buf = None
        self.grad = None
        self.needs_grad = False
        self.alias_of = None          # set when a fusion removes this tensor
        self.id = Tensor._count
        Tensor._count += 1

    @property
    def numel(self):
        return int(np.prod(self.shape[:-1])) * self.cs

    def get_shape(self):
        """Shape in the API layout (NCHW when channel_first), like tf.Tensor.get_shape().as_list()."""
        if len(self.shape) == 4 and self.channel_first:
            n, h, w, c = self.shape
            return [n, c, h, w]
        return list(self.shape)

    def __repr__(self):
        return 'Tensor({}, {}, {})'.format(self.name, self.shape, self.dtype)


class Variable(object):
    """A trainable parameter or a BN statistic; storage is a view into one of the model's flat buffers."""

    def __init__(self, name, shape, kind, trainable, init, block):
        self.name = name
        self.shape = tuple(int(s) for s in shape)
        self.kind = kind              # 'weight' | 'bias' | 'gamma' | 'beta' | 'mu' | 'sigma'
        self.trainable = trainable
        self.init = init              # callable(shape, generator) -> torch tensor (cpu, fp32)
        self.block = block
        self.offset = None
        self.size = int(np.prod(self.shape))
        self.store = None             # the FlatStore holding it

    @property
    def data(self):
        return self.store.data[self.offset:self.offset + self.size].view(self.shape)

    @property
    def grad(self):
        return self.store.grad[self.offset:self.offset + self.size].view(self.shape)

    @property
    def ema(self):
        return self.store.ema[self.offset:self.offset + self.size].view(self.shape)

    @property
    def accum(self):
        return self.store.accum[self.offset:self.offset + self.size].view(self.shape)


class FlatStore(object):
    """Contiguous fp32 storage for a list of variables (each padded to 16 bytes)."""

    def __init__(self, variables, device, with_grad):
        off = 0
        for v in variables:
            v.offset = off
            v.store = self
            off += (v.size + 3) // 4 * 4
        self.size = off
        self.variables = list(variables)
        n = max(off, 4)
        self.data = torch.zeros(n, dtype=torch.float32, device=device)
        self.ema = torch.zeros(n, dtype=torch.float32, device=device)
        self.grad = torch.zeros(n, dtype=torch.float32, device=device) if with_grad else None
        self.accum = torch.zeros(n, dtype=torch.float32, device=device) if with_grad else None


class Node(object):
    def __init__(self, op, inputs, outputs, **attrs):
        self.op = op
        self.inputs = list(inputs)
        self.outputs = list(outputs)
        self.attrs = attrs
        self.scope = attrs.get('scope', '')
        for t in self.inputs:
            t.consumers.append(self)
        for t in self.outputs:
            t.producer = self

    def __repr__(self):
        return 'Node({}, {})'.format(self.op, self.scope)


class Program(object):
    """A flat list of (fn, args) C-ABI launches; args[-1] is the stream slot.  Calls added with add_side() go to a second
    HIP stream: each waits (event) for everything enqueued on the main stream before it and the main stream joins the
    side stream at the end of run() — used to run the MFMA-bound wgrad kernels beside the HBM-bound BN-backward chain."""

    def __init__(self):
        self.calls = []
        self.marks = {}               # label -> index (used to place all-reduce hooks)
        self.side = {}                # call index -> torch.cuda.Event (lazily created)
        self.side_stream = None
        self._join = None

    def add(self, fn, *args):
        self.calls.append((fn, list(args) + [None]))

    def add_side(self, fn, *args):
        self.side[len(self.calls)] = None
        self.calls.append((fn, list(args) + [None]))

    def mark(self, label):
        self.marks.setdefault(label, len(self.calls))

    def run(self, stream_ptr, hooks=None):
        check = _ffi.check
        side = self.side if (self.side and self.side_stream is not None) else None
        if side is None and not hooks:
            for fn, args in self.calls:
                args[-1] = stream_ptr
                rc = fn(*args)
                if rc:
                    check(rc)
            return
        main = torch.cuda.current_stream() if side is not None else None
        side_ptr = self.side_stream.cuda_stream if side is not None else 0
        for i, (fn, args) in enumerate(self.calls):
            if hooks:
                h = hooks.get(i)
                if h is not None:
                    h()
            if side is not None and i in side:
                ev = side[i]
                if ev is None:
                    ev = side[i] = torch.cuda.Event()
                ev.record(main)
                self.side_stream.wait_event(ev)
                args[-1] = side_ptr
            else:
                args[-1] = stream_ptr
            rc = fn(*args)
            if rc:
                check(rc)
        if hooks:
            h = hooks.get(len(self.calls))
            if h is not None:
                h()
        if side is not None:
            if self._join is None:
                self._join = torch.cuda.Event()
            self._join.record(self.side_stream)
            main.wait_event(self._join)

    def __len__(self):
        return len(self.calls)


def ptr(t):
    return 0 if t is None else t.data_ptr()


def masked_stream(device, ncu, first=0, total=256):
    """A HIP stream whose kernels may only use `ncu` of the chip's CUs (hipExtStreamCreateWithCUMask; mask bits first .. first + ncu - 1,
    which the driver deals round-robin over the XCDs), wrapped as a torch stream.  Experiment switch of the two-stream backward
    (MCN_SIDE_CUS / MCN_MAIN_CUS, LABNOTES.md section 3 "CU masks"): the default streams are unmasked."""
    import ctypes
    import os
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
    words = (total + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for b in range(first, min(total, first + ncu)):
        mask[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError('hipExtStreamCreateWithCUMask failed: {}'.format(rc))
    return torch.cuda.ExternalStream(st.value, device=device)


class Graph(object):
    def __init__(self, device, dtype):
        self.device = device
        self.dtype = dtype            # compute dtype of activations
        self.nodes = []
        self.tensors = []
        self.feeds = {}

    def tensor(self, shape, dtype=None, name='', channel_first=False):
        t = Tensor(shape, dtype or self.dtype, name, channel_first)
        self.tensors.append(t)
        return t

    def node(self, op, inputs, outputs, **attrs):
        n = Node(op, inputs, outputs, **attrs)
        self.nodes.append(n)
        return n

    # ---- fusion -----------------------------------------------------------------------------------
    def fuse(self):
        """bn -> relu, bn -> swish, bn -> add(skip) [-> relu]: the activation and the residual add run inside the
        BN apply pass (one read + one write instead of three / five)."""
        alive = list(self.nodes)

        def sole_consumer(t, op):
            return len(t.consumers) == 1 and t.consumers[0].op == op

        changed = True
        while changed:
            changed = False
            for n in alive:
                if n.op != 'bn' or n.attrs.get('act', 0) or not n.attrs.get('training_graph', True):
                    continue
                y = n.outputs[0]
                if n.attrs.get('skip') is None and sole_consumer(y, 'add') and y.consumers[0].inputs[0] is y \
                        and y.consumers[0].inputs[1] is not y:
                    add = y.consumers[0]
                    skip = add.inputs[1]
                    # the skip must be produced before this bn runs
                    if alive.index(skip.producer) < alive.index(n) if skip.producer in alive else True:
                        n.attrs['skip'] = skip
                        skip.consumers.remove(add)
                        skip.consumers.append(n)
                        n.inputs.append(skip)
                        y.alias_of = add.outputs[0]
                        n.outputs[0] = add.outputs[0]
                        add.outputs[0].producer = n
                        alive.remove(add)
                        changed = True
                        break
                if n.attrs.get('skip') is None and sole_consumer(y, 'act') and y.consumers[0].attrs['kind'] == _ffi.ACT_SWISH:
                    sw = y.consumers[0]                     # bn -> swish (every EfficientNet conv): one apply pass
                    n.attrs['act'] = _ffi.ACT_SWISH
                    y.alias_of = sw.outputs[0]
                    n.outputs[0] = sw.outputs[0]
                    sw.outputs[0].producer = n
                    alive.remove(sw)
                    changed = True
                    break
                if sole_consumer(y, 'relu'):
                    relu = y.consumers[0]
                    n.attrs['act'] = _ffi.ACT_RELU
                    y.alias_of = relu.outputs[0]
                    n.outputs[0] = relu.outputs[0]
                    relu.outputs[0].producer = n
                    alive.remove(relu)
                    changed = True
                    break
            if not changed:
                # add -> relu (identity-shortcut units whose bn could not absorb the add)
                for n in alive:
                    if n.op == 'add' and not n.attrs.get('act', 0) and sole_consumer(n.outputs[0], 'relu'):
                        relu = n.outputs[0].consumers[0]
                        n.attrs['act'] = _ffi.ACT_RELU
                        n.outputs[0].alias_of = relu.outputs[0]
                        n.outputs[0] = relu.outputs[0]
                        relu.outputs[0].producer = n
                        alive.remove(relu)
                        changed = True
                        break
        self.nodes = alive

    # ---- memory plan ----------------------------------------------------------------------------------
    def live_tensors(self):
        seen, out = set(), []
        for n in self.nodes:
            for t in n.inputs + n.outputs:
                if t.id not in seen:
                    seen.add(t.id)
                    out.append(t)
        return out

    def allocate(self, training):
        for t in self.live_tensors():
            if t.buf is None and getattr(t, 'share_of', None) is None:
                shape = t.shape[:-1] + (t.cs,)
                t.buf = torch.zeros(shape, dtype=TORCH_DT[t.dtype], device=self.device)
        for t in self.live_tensors():                       # stop_gradient outputs: the producer's storage under another name
            src = getattr(t, 'share_of', None)
            while src is not None and src.buf is None and getattr(src, 'share_of', None) is not None:
                src = src.share_of
            if t.buf is None and src is not None:
                t.buf = src.buf
        if training:
            # which tensors need a gradient: anything downstream of a trainable variable
            for n in self.nodes:
                if n.op in ('input', 'labels', 'stopgrad'):   # (tf.stop_gradient: nothing downstream of x through this node needs x's gradient)
                    continue
                # (frozen variables, blocks_to_train, do not start a gradient path)
                trains = n.attrs.get('has_params') and any(getattr(n.attrs.get(k), 'trainable', False) for k in ('w', 'b', 'gamma', 'beta'))
                if trains or any(t.needs_grad for t in n.inputs):
                    for t in n.outputs:
                        if t.dtype in TORCH_DT and n.op not in ('softmax', 'loss'):
                            t.needs_grad = True
            for t in self.live_tensors():
                if t.needs_grad and t.grad is None:
                    t.grad = torch.zeros(t.shape, dtype=TORCH_DT[t.dtype], device=self.device)

    def activation_bytes(self):
        tot = 0
        for t in self.live_tensors():
            for b in (t.buf, t.grad):
                if b is not None:
                    tot += b.numel() * b.element_size()
        return tot
