"""ConvNet — host-side mirror of the reference's building-block surface (reference convnet.py:14-2577)
for the conv / batch-norm / ReLU / pooling training path, executing on MI355X through libmcn_hip.

Same constructor kwargs, hooks (`_init_params`, `_build_model` returning a dict with 'logits' and
'pred') and block-method signatures as the reference, so a model file written against the
reference's ConvNet needs only its `tf.*` calls swapped for the methods here.
What differs by design: there is no TF session — `_build_model` records a static graph once, and
`compile()` turns it into flat lists of HIP launches (graph.py / executor.py).

Not built (raise NotImplementedError): depthwise conv, weight standardisation, group norms,
augmentation, dropout rate > 0, transposed conv / upsampling (SURVEY.md §8f "next" rows).
"""
import math
import os
from abc import abstractmethod
from contextlib import contextmanager, nullcontext

import numpy as np
import torch

from . import _ffi
from .graph import MCN_DT, TORCH_DT, FlatStore, Graph, Variable, out_size, same_pads


# ---- initializers (reference passes tf.initializers.* objects) -----------------------------------------
def he_normal():
    """tf.initializers.he_normal(): truncated normal (+-2 sigma), std = sqrt(2/fan_in)/0.87962566103423978."""
    def init(shape, gen):
        fan_in = int(np.prod(shape[:-1])) if len(shape) > 1 else int(shape[0])
        std = math.sqrt(2.0 / max(fan_in, 1)) / 0.87962566103423978
        t = torch.empty(shape, dtype=torch.float32)
        torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=gen)
        return t
    return init


def variance_scaling(scale=1.0, mode='fan_in', distribution='truncated_normal'):
    """tf.initializers.variance_scaling (models/efficientnet.py:20-23): n = fan_in | fan_out | their mean with
    fan_out = shape[-1] * receptive field; truncated normal std = sqrt(scale/n)/0.8796, uniform limit = sqrt(3*scale/n)."""
    def init(shape, gen):
        rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
        fan_in = (shape[-2] if len(shape) > 1 else shape[0]) * rf
        fan_out = shape[-1] * rf
        n = {'fan_in': fan_in, 'fan_out': fan_out, 'fan_avg': (fan_in + fan_out) / 2.0}[mode]
        t = torch.empty(shape, dtype=torch.float32)
        if distribution == 'uniform':
            lim = math.sqrt(3.0 * scale / max(n, 1))
            t.uniform_(-lim, lim, generator=gen)
        else:
            std = math.sqrt(scale / max(n, 1)) / 0.87962566103423978
            torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=gen)
        return t
    return init


def zeros():
    return lambda shape, gen: torch.zeros(shape, dtype=torch.float32)


def ones():
    return lambda shape, gen: torch.ones(shape, dtype=torch.float32)


def _pair(v):
    if isinstance(v, (list, tuple)):
        return [v[0], v[0]] if len(v) == 1 else [v[0], v[1]]
    return [v, v]


class ConvNet(object):
    def __init__(self, input_shape, num_classes, loss_weights=None, session=None, model_scope=None,
                 companion_networks=None, next_elements=None, backbone_only=False, auto_build=True, **kwargs):
        """Arguments as reference convnet.py:15-28.  `session`, `next_elements` and `companion_networks` are TF
        plumbing with no meaning here and are accepted and ignored."""
        self._block_list = []
        self.__dict__['_curr_block'] = None
        self._custom_feed_dict = dict()
        assert len(input_shape) == 3, 'input_size must contain 3D size'
        self._input_size = list(input_shape)
        self._num_classes = num_classes
        self._loss_weights = loss_weights
        self._model_scope = model_scope
        self._backbone_only = backbone_only
        self._parameters = kwargs

        # reference: tf.float16 when half_precision (convnet.py:63).  The MI355X build defaults its low precision to bf16
        # (fp32 exponent range: no loss scaling needed); half_precision_dtype='float16' selects the reference's own type,
        # to be used with the optimizer's loss_scaling_factor (optimizers.py:102-111) exactly as in the reference
        lp = str(kwargs.get('half_precision_dtype', 'bfloat16'))
        assert lp in ('bfloat16', 'float16'), 'half_precision_dtype must be bfloat16 or float16'
        self._dtype = lp if kwargs.get('half_precision', False) else 'float32'
        self._channel_first = kwargs.get('channel_first', False)
        self._argmax_output = kwargs.get('argmax_output', False)

        self.rank = int(os.environ.get('RANK', 0))
        self.local_rank = int(os.environ.get('LOCAL_RANK', 0))
        num_gpus = kwargs.get('num_gpus', None)
        if num_gpus is None:
            num_gpus = int(os.environ.get('WORLD_SIZE', 1))
        # one process per GPU replaces the reference's in-graph towers (convnet.py:431-436)
        self.world_size = max(int(num_gpus), 1)
        self._num_devices = self.world_size
        self._compute_device = 'gpu'
        self._device_offset = 0
        dev = kwargs.get('device', None)
        if dev is None:
            dev = 'cuda:{}'.format(self.local_rank) if torch.cuda.is_available() else 'cpu'
        self.device = torch.device(dev)

        total_batch = int(kwargs.get('batch_size', 16))            # reference default, dataset.py:101
        assert total_batch % self.world_size == 0, 'batch_size must be divisible by the number of GPUs'
        self.batch_size = total_batch
        self.device_batch = total_batch // self.world_size          # dataset.py:113

        self._dropout_weights = kwargs.get('dropout_weights', False)
        self._dropout_features = kwargs.get('dropout_features', True)
        self._blocks_to_train = kwargs.get('blocks_to_train', None)
        self._update_batch_norm = kwargs.get('update_batch_norm', None)
        self._moving_average_decay = kwargs.get('moving_average_momentum', kwargs.get('moving_average_decay', 0.99))
        self._batch_norm_decay = kwargs.get('batch_norm_momentum', kwargs.get('batch_norm_decay', 0.99))

        self._flops = 0
        self._params = 0
        self._nodes = 0
        self._conv_macs = 0
        self._layer_info = []
        self._scope = []
        self._collections = {}
        self.variables = {}             # name -> Variable (trainables and BN statistics)
        self._var_order = []
        self._random_nodes = []         # dropout / stochastic-depth nodes whose masks are drawn on the host per step
        self._mask_rng = np.random.default_rng([int(kwargs.get('seed', 0)), self.rank, 0x5eed])
        self.graph = None
        self.compiled = False
        self.seed = int(kwargs.get('seed', 0))
        self.fuse = bool(kwargs.get('fuse', True))
        if auto_build:
            self.build()

    # ---- properties the reference exposes ------------------------------------------------------------------
    name = 'ConvNet'
    input_size = property(lambda self: self._input_size)
    num_classes = property(lambda self: self._num_classes)
    feature_reduction = property(lambda self: self._parameters.get('feature_reduction_factor', 0))   # convnet.py:109
    loss_weights = property(lambda self: self._loss_weights)
    model_scope = property(lambda self: self._model_scope)
    backbone_only = property(lambda self: self._backbone_only)
    dtype = property(lambda self: self._dtype)
    channel_first = property(lambda self: self._channel_first)
    argmax_output = property(lambda self: self._argmax_output)
    num_devices = property(lambda self: self._num_devices)
    compute_device = property(lambda self: self._compute_device)
    device_offset = property(lambda self: self._device_offset)
    dropout_weights = property(lambda self: self._dropout_weights)
    dropout_features = property(lambda self: self._dropout_features)
    blocks_to_train = property(lambda self: self._blocks_to_train)
    update_batch_norm = property(lambda self: self._update_batch_norm)
    moving_average_decay = property(lambda self: self._moving_average_decay)
    batch_norm_decay = property(lambda self: self._batch_norm_decay)
    block_list = property(lambda self: self._block_list)
    num_blocks = property(lambda self: self.__dict__.get('_num_blocks', len(self._block_list)))
    flops = property(lambda self: self._flops)
    params = property(lambda self: self._params)
    nodes = property(lambda self: self._nodes)
    conv_macs = property(lambda self: self._conv_macs)
    layer_info = property(lambda self: self._layer_info)
    custom_feed_dict = property(lambda self: self._custom_feed_dict)

    def __setattr__(self, key, value):
        if key == '_curr_block':                       # convnet.py:239-244
            self.__dict__[key] = value
            if value not in self._block_list:
                self._block_list.append(value)
        else:
            super(ConvNet, self).__setattr__(key, value)

    # ---- scopes / collections / variables -----------------------------------------------------------------------
    @contextmanager
    def variable_scope(self, name):
        """Stand-in for tf.variable_scope: nests names as 'block_1/res_0/conv_0/...'."""
        self._scope.append(str(name))
        try:
            yield
        finally:
            self._scope.pop()

    def scope_name(self, leaf=None):
        parts = list(self._scope) + ([leaf] if leaf else [])
        return '/'.join(parts)

    def add_to_collection(self, name, tensor):
        """reference convnet.py (add_to_collection(name, tensor)): named lists of variables."""
        self._collections.setdefault(name, []).append(tensor)

    def get_collection(self, key):
        return list(self._collections.get(key, []))

    def _trainable_here(self):
        return self._blocks_to_train is None or self._curr_block in self._blocks_to_train

    def _new_variable(self, name, shape, kind, init, trainable):
        full = self.scope_name(name)
        if full in self.variables:
            raise ValueError('variable {} already exists'.format(full))
        if not isinstance(shape, (list, tuple)):
            shape = [shape]
        v = Variable(full, shape, kind, trainable, init, self._curr_block)
        self.variables[full] = v
        self._var_order.append(v)
        self.add_to_collection('block_{}/variables'.format(self._curr_block), v)
        return v

    def weight_variable(self, shape, initializer=None, weight_standardization=False, paddings=((0, 0), (0, 0)), name='weights'):
        """reference convnet.py:1382-1431 (fp32 master on the parameter device, EMA shadow, per-use cast)."""
        if weight_standardization:
            raise NotImplementedError('weight standardisation is outside the built path')
        if any(p > 0 for pp in paddings for p in pp):
            raise NotImplementedError('kernel_paddings are outside the built path')
        if self._dropout_weights and self._parameters.get('dropout_rate', 0.0) > 0.0:
            raise NotImplementedError('weight dropout is outside the built path')
        v = self._new_variable(name, shape, 'weight', initializer or he_normal(), self._trainable_here())
        self.add_to_collection('weight_variables', v)
        self.add_to_collection('block_{}/weight_variables'.format(self._curr_block), v)
        return v

    def bias_variable(self, shape, initializer=None, name='biases'):
        """reference convnet.py:1433-1462."""
        v = self._new_variable(name, shape, 'bias', initializer or zeros(), self._trainable_here())
        self.add_to_collection('bias_variables', v)
        self.add_to_collection('block_{}/bias_variables'.format(self._curr_block), v)
        return v

    # ---- build -------------------------------------------------------------------------------------------------------
    def build(self):
        """reference convnet.py:134-237 (conditions, global step, EMA, _init_params, _init_model, report)."""
        kwargs = self._parameters
        self.global_step = 0
        self.is_train = True
        self.dropout_rate = kwargs.get('dropout_rate', 0.0)
        self.dropout_rate_weights = self.dropout_rate if self._dropout_weights else 0.0
        self.dropout_rate_features = self.dropout_rate if self._dropout_features else 0.0
        self.image_mean = kwargs.get('image_mean', 0.5) if kwargs.get('zero_center', True) else 0.0
        self.scale_factor = kwargs.get('scale_factor', 2.0)
        self.graph = Graph(self.device, self._dtype)
        with self.variable_scope(self._model_scope) if self._model_scope is not None else nullcontext():
            self._init_params(**kwargs)
            self._init_model(**kwargs)
        self._flops, self._params, self._nodes = int(self._flops), int(self._params), int(self._nodes)
        for blk in list(self._block_list):
            if not self.get_collection('block_{}/variables'.format(blk)):
                self._block_list.remove(blk)
        if kwargs.get('verbose', False) and self.rank == 0:
            print('\n# computing devices : {} {}(s)'.format(self.num_devices, self.compute_device))
            print('# variable blocks : {} {}'.format(self.num_blocks, self.block_list))
            print('\n# FLOPs : {:-15,}\n# Params: {:-15,}\n# Nodes : {:-15,}\n'.format(self.flops, self.params, self.nodes))
        from . import _ffi
        if kwargs.get('auto_compile', True) and (self.device.type == 'cuda' or _ffi.IS_CPU_LIB):
            self.compile()

    def _init_params(self, **kwargs):
        pass

    @abstractmethod
    def _build_model(self):
        """Must return a dict of tensors including 'logits' and 'pred' (reference convnet.py:257-264)."""

    def _init_model(self, **kwargs):
        """reference convnet.py:425-513 for one tower: labels, input preparation, model, loss."""
        B = self.device_batch
        H, W, C = self._input_size
        g = self.graph
        # X: prepared input; channels zero-padded to one 16-byte chunk so the stem conv takes the MFMA path
        chunk = 4 if self._dtype == 'float32' else 8
        self.X = g.tensor((B, H, W, C), self._dtype, 'X', self._channel_first)
        self.X.cs = (C + chunk - 1) // chunk * chunk
        g.node('input', [], [self.X], image_mean=self.image_mean, scale_factor=self.scale_factor, src_nchw=self._channel_first)
        if kwargs.get('zero_pad_ratio', 0.0) > 0.0 or kwargs.get('cutmix', False):
            raise NotImplementedError('augmentation / zero padding are outside the built path')
        self._curr_block = None
        self.d = self._build_model()
        if self._backbone_only:
            self.logits = self.pred = self.loss_tensor = None
            return
        logits = self.d['logits']
        if logits.dtype != 'float32':                                  # convnet.py:477-480
            logits32 = g.tensor(logits.shape, 'float32', 'logits_fp32')
            g.node('cast', [logits], [logits32])
            logits = logits32
            self.d['logits'] = logits
        self.logits = logits
        self._build_loss(**kwargs)

    def _build_loss(self, **kwargs):
        """reference convnet.py:528-597."""
        # bias_norm_decay (convnet.py:536-537, optimizers.py:150-151): biases, gammas and betas join the regularised set
        self.bias_norm_decay = bool(kwargs.get('bias_norm_decay', False))
        g = self.graph
        shape = tuple(self.logits.shape)                    # [B, C] (classification) or [B, H, W, C] (SegNet: per-pixel loss)
        seg = len(shape) == 4
        self._label_shape = shape[:-1]
        self.Y = g.tensor(shape, 'float32', 'Y_onehot')
        g.node('labels', [], [self.Y], seg=seg)
        labels = self.Y
        ls_factor = float(kwargs.get('label_smoothing', 0.0))
        if ls_factor > 0.0:
            labels = self._label_smoothing(labels, ls_factor)
        self._loss_fn(labels, self.logits, **kwargs)

    def _loss_fn(self, labels, logits, **kwargs):
        """reference convnet.py:599-601 (softmax cross-entropy over the last axis; the hook a model overrides for another loss).
        Here it records ONE fused node: softmax, cross-entropy against the (smoothed) labels, valid mask and the L2 term of
        _build_loss (mcn_softmax_xent_fwd_bwd / _rows_fwd_bwd + mcn_l2_loss)."""
        g = self.graph
        shape = tuple(logits.shape)
        self.pred = g.tensor(shape, 'float32', 'pred')
        self.d['pred'] = self.pred
        raw = getattr(labels, 'soft_avg_of', None)          # SegNet smoothing: `labels` is the 5x5 average of the raw one-hot map `raw`
        # l1_reg / focal_loss_factor / sigmoid_focal_loss_factor (convnet.py:530-533, 553-557, 581-592): 0 = off, as in the reference's defaults
        self._loss_node = g.node('loss', [logits, labels] if raw is None else [logits, raw, labels], [self.pred], l2_reg=float(kwargs.get('l2_reg', 1e-4)),
                                 l1_reg=float(kwargs.get('l1_reg', 0.0)), focal_gamma=float(kwargs.get('focal_loss_factor', 0.0)),
                                 sigmoid_focal_alpha=float(kwargs.get('sigmoid_focal_loss_factor', 0.0)),
                                 label_smoothing=float(getattr(labels, 'ls_factor', 0.0)), rows=int(np.prod(shape[:-1])), per_pixel=len(shape) == 4)
        return self._loss_node

    def _label_smoothing(self, labels, ls_factor, name='label_smoothing'):
        """reference convnet.py:603-607: labels * (1 - f) + f / num_classes — applied inside the loss kernel, so the factor rides
        on the label tensor."""
        labels.ls_factor = float(ls_factor)
        return labels

    def _set_num_blocks(self, num_blocks):
        """reference convnet.py:350-351 (models set the number of variable blocks explicitly)."""
        self.__dict__['_num_blocks'] = num_blocks

    # ---- compile: storage + launch lists -----------------------------------------------------------------------------------
    def compile(self, loss_scale=1.0):
        """Allocate storage and initialise the variables (once), then lower the graph to launch lists.  Calling it again
        (e.g. with another loss scale) only re-lowers: variables, EMA shadows, momentum and running statistics are kept."""
        from .executor import Lowering
        from . import _ffi
        if self.device.type != 'cuda' and not _ffi.IS_CPU_LIB:
            raise RuntimeError('ConvNet.compile() needs an MI355X (cuda device); the HIP path has no CPU fallback')
        if self.device.type == 'cuda' and _ffi.IS_CPU_LIB:
            raise RuntimeError('MCN_LIB_PATH names libmcn_cpu.so (host tensors only): build the model with device=\'cpu\'')
        g = self.graph
        if not getattr(self, '_allocated', False):
            self._allocate()
            self._allocated = True
        self.loss_scale = float(loss_scale)
        self._train_low = Lowering(g, self, 'train', loss_scale).lower()
        self._eval_low = Lowering(g, self, 'eval', 1.0).lower()
        self.compiled = True
        return self

    def _plan_pixel_pairs(self):
        """2-byte storage types: a stride-2 conv that is the only reader of the (<= 4-channel) network input runs in its pixel-pair
        form (mcn_conv2d_pair_geom: the image is stored 4 channels per pixel, two adjacent pixels are one 16-byte chunk, the
        7x7/2 stem becomes a 7x4 stride-(2,1) conv on 8 channels: K 392 -> 224).  The node keeps the original geometry as
        `geom_orig`; its weights / weight gradient pass through mcn_conv2d_pair_weights / _pair_wgrad_fold in the executor.
        MCN_PAIR_STEM=0 switches it off."""
        import ctypes
        if self._dtype == 'float32' or os.environ.get('MCN_PAIR_STEM', '1') == '0' or getattr(self, '_pairs_planned', False):
            return
        self._pairs_planned = True
        X = self.X
        readers = [n for n in self.graph.nodes if any(t is X for t in n.inputs)]
        if len(readers) != 1 or readers[0].op != 'conv' or X.needs_grad:
            return
        n = readers[0]
        src = n.attrs['geom']
        g4 = _ffi.ConvGeom(*[getattr(src, f) for f, _ in _ffi.ConvGeom._fields_])
        g4.x_cs = 4
        pg = _ffi.ConvGeom()
        if _ffi.lib.mcn_conv2d_pair_geom(ctypes.byref(g4), MCN_DT[self._dtype], ctypes.byref(pg)) != 1:
            return
        X.cs = 4
        n.attrs['geom_orig'] = g4
        n.attrs['geom'] = pg
        n.attrs['flop_scale'] = float(g4.KH * g4.KW * g4.Cin) / float(pg.KH * pg.KW * pg.Cin)     # real MACs per MAC of the paired form

    def _allocate(self):
        g = self.graph
        if self.fuse and not getattr(self, '_fused', False):
            g.fuse()
            self._fused = True
        self._plan_pixel_pairs()
        B = self.device_batch
        H, W, C = self._input_size
        dev = self.device
        self.X_in = torch.zeros((B, C, H, W) if self._channel_first else (B, H, W, C), dtype=torch.float32, device=dev)
        self.Y_in = torch.zeros(tuple(getattr(self, '_label_shape', (B,))), dtype=torch.float32, device=dev)
        # flat storage: [conv/fc weights | biases, gammas, betas] so the L2 term covers one contiguous range
        trainables = [v for v in self._var_order if v.kind in ('weight', 'bias', 'gamma', 'beta')]
        ordered = [v for v in trainables if v.kind == 'weight'] + [v for v in trainables if v.kind != 'weight']
        self.store = FlatStore(ordered, dev, with_grad=True)
        nw = [v for v in ordered if v.kind == 'weight']
        self.n_l2_elems = (nw[-1].offset + (nw[-1].size + 3) // 4 * 4) if nw else 0
        if getattr(self, 'bias_norm_decay', False):
            self.n_l2_elems = self.store.size            # L2 term and decoupled decay cover every trainable variable
        self.stats = FlatStore([v for v in self._var_order if v.kind in ('mu', 'sigma')], dev, with_grad=False)
        # per-step batch statistics in the same order as `stats` (for the cross-rank chain, convnet.py:1899-1909)
        self.batch_stats = torch.zeros(max(self.stats.size, 4), dtype=torch.float32, device=dev)
        self.initialize_variables()
        for n in g.nodes:
            if n.op == 'bn':
                a = n.attrs
                c = n.inputs[0].shape[-1]
                mu, sg = a['mu'], a['sigma']
                a['saved'] = dict(mean=torch.zeros(c, dtype=torch.float32, device=dev), invstd=torch.zeros(c, dtype=torch.float32, device=dev),
                                  bmean=self.batch_stats[mu.offset:mu.offset + c], bvar=self.batch_stats[sg.offset:sg.offset + c])
            elif n.op == 'mulmask':
                pass                                       # one flat buffer for all masks, below
            elif n.op == 'loss':
                a = n.attrs
                a['pred'] = self.pred
                a['ce'] = torch.zeros(a['rows'], dtype=torch.float32, device=dev)
                a['coef'] = torch.zeros(a['rows'], dtype=torch.float32, device=dev)
                a['loss'] = torch.zeros(4, dtype=torch.float32, device=dev)
                a['class_w'] = None if self._loss_weights is None else torch.tensor(np.asarray(self._loss_weights, dtype=np.float32), device=dev)
                self.loss_buf = a['loss']
                self.valid_coef = a['coef']
        self._allocate_masks()
        g.allocate(training=True)

    def initialize_variables(self, seed=None):
        """Fill every variable from its initializer (torch.Generator seeded per SURVEY §8d), EMA shadows start at the
        initial value (tf.train.ExponentialMovingAverage semantics)."""
        gen = torch.Generator().manual_seed(self.seed if seed is None else seed)
        for v in self._var_order:
            v.data.copy_(v.init(v.shape, gen).to(self.device))
        self.store.ema.copy_(self.store.data)
        self.stats.ema.copy_(self.stats.data)
        self.store.accum.zero_()
        self.store.grad.zero_()
        self.global_step = 0

    def set_variables(self, values, reset_state=True):
        """Inject explicit values {name: ndarray} (parity tests always do: the TF RNG stream is not reproducible)."""
        for k, a in values.items():
            v = self.variables[k]
            v.data.copy_(torch.as_tensor(np.asarray(a, dtype=np.float32)).view(v.shape).to(self.device))
        if reset_state:
            self.store.ema.copy_(self.store.data)
            self.stats.ema.copy_(self.stats.data)
            self.store.accum.zero_()
            self.global_step = 0

    def get_variables(self, which='data'):
        return {k: getattr(v, which).detach().float().cpu().numpy().copy() for k, v in self.variables.items()
                if which in ('data', 'ema') or v.kind in ('weight', 'bias', 'gamma', 'beta')}

    # ---- running ----------------------------------------------------------------------------------------------------------------
    def feed(self, X, Y=None):
        """Copy one device batch into the static input buffers (non-blocking for pinned / device sources)."""
        self.X_in.copy_(torch.as_tensor(X).to(self.X_in.dtype), non_blocking=True)
        if Y is not None:
            self.Y_in.copy_(torch.as_tensor(Y).to(self.Y_in.dtype), non_blocking=True)

    def stream_ptr(self):
        if self.device.type != 'cuda':                       # libmcn_cpu.so (MCN_LIB_PATH): host tensors, no stream
            return 0
        return torch.cuda.current_stream(self.device).cuda_stream

    def autotune(self):
        """Pin the fastest measured tile shape for every conv launch (call after a few steps, buffers hold real data);
        the eval lowering inherits the forward choices."""
        return self._train_low.autotune()     # both lowerings share the per-op geometry objects of each conv node

    def forward(self, train=True):
        low = self._train_low if train else self._eval_low
        sp = self.stream_ptr()
        low.prepack.run(sp)            # weights changed since the last pass (optimizer step / EMA): refresh the packed operands
        low.fwd.run(sp)

    def backward(self, hooks=None):
        self._train_low.bwd.run(self.stream_ptr(), hooks)

    def fetch(self, tensor):
        """Materialised value of a graph tensor as numpy (API layout)."""
        t = tensor
        while t.alias_of is not None and t.buf is None:
            raise KeyError('{} was fused away; build the model with fuse=False to fetch it'.format(tensor.name))
        a = t.buf[..., :t.shape[-1]].float().cpu().numpy()
        if len(t.shape) == 4 and self._channel_first:
            a = a.transpose(0, 3, 1, 2)
        return a

    def predict(self, dataset, verbose=False, return_images=True, max_examples=None, run_init_ops=True, **kwargs):
        """reference convnet.py:609-665: forward-only loop with is_train=False (EMA weights + EMA running statistics) over
        GLOBAL batches of dataset.batch_size; rank r evaluates shard r of each global batch (dataset.py:113-129) and the
        per-rank labels / predictions are all-gathered into the reference's Y_all / pred order, so every rank returns the
        whole set.  The last, short batch is delivered as the reference's tf.data pipeline delivers it: rows past the end
        are fed as ignored samples (NaN label) and dropped from the outputs and from that batch's loss mean."""
        b, world = self.device_batch, self.world_size
        assert getattr(dataset, 'num_shards', world) == world, 'Number of devices mismatch between the model and dataset'
        gb = b * world
        pred_size = dataset.num_examples if max_examples is None else min(max_examples, dataset.num_examples)
        num_steps = int(np.ceil(pred_size / gb))
        _X = np.zeros([pred_size] + list(self._input_size), dtype=np.float32) if return_images else np.zeros([pred_size, 4, 4, 3], np.float32)
        _Y_true = np.zeros([pred_size] + list(self.Y.shape[1:]), dtype=np.float32)
        _Y_pred = np.zeros([pred_size] + list(self.pred.shape[1:]), dtype=np.float32)
        _loss = np.zeros(num_steps, dtype=np.float32)
        gather = None
        if world > 1:
            from .dist import all_gather_rows as gather, init_process_group
            init_process_group(self.device)
        dataset.initialize()
        for i in range(num_steps):
            X, Y = dataset.next_batch(b, shard=self.rank)
            s = i * gb
            num_left = min(pred_size - s, gb)
            mine = int(np.clip(num_left - self.rank * b, 0, b))        # rows of this rank's shard that exist
            if mine < b:
                Y = np.array(Y, dtype=np.float32, copy=True)
                Y[mine:] = np.nan                                        # NaN label = ignored sample (convnet.py:441-449)
            self.feed(X, Y)
            self.forward(train=False)
            y_dev, p_dev = self.Y.buf, self.pred.buf
            if gather is not None:
                y_dev, p_dev = gather(y_dev), gather(p_dev)
            if return_images:
                xin = self.X_in if gather is None else gather(self.X_in)
                xin = xin.cpu().numpy()
                _X[s:s + num_left] = (xin.transpose(0, 2, 3, 1) if self._channel_first else xin)[:num_left]
            _Y_true[s:s + num_left] = y_dev.cpu().numpy()[:num_left]
            _Y_pred[s:s + num_left] = p_dev.cpu().numpy()[:num_left]
            _loss[i] = self._eval_batch_loss(mine, num_left, gather)
        return _X, _Y_true, _Y_pred, float(np.mean(_loss))

    def _eval_batch_loss(self, mine, num_left, gather):
        """Loss of the batch just evaluated = mean over towers (convnet.py:510); a short last batch averages its
        cross-entropy over the rows that exist (the reference's tower sees a smaller batch), not over the padded buffer."""
        b = self.device_batch
        full = self.loss_buf[0]
        if num_left < b * self.world_size:
            n = self._loss_node.attrs
            rows_per = n['rows'] // b                                     # 1, or H*W for per-pixel losses
            ce_sum = (n['ce'] * n['coef']).sum()
            l2 = full - ce_sum / n['rows']
            full = ce_sum / max(mine * rows_per, 1) + l2 if mine > 0 else l2 * 0.0
            if gather is not None:
                per_rank = gather(full.reshape(1)).cpu().numpy()
                towers = int(np.ceil(num_left / b))                       # ranks that received any row
                return float(per_rank[:towers].mean())
            return float(full.item())
        if gather is not None:
            return float(gather(full.reshape(1)).mean().item())
        return float(full.item())

    # ---- layers -----------------------------------------------------------------------------------------------------------------------
    def _log_layer(self, name, shape, flops, params, nodes):
        self._flops += flops
        self._nodes += nodes
        self._params += params
        self._layer_info.append({'name': name, 'shape': shape, 'flops': int(flops), 'params': int(params), 'nodes': int(nodes)})

    def conv_layer(self, x, kernel, stride, out_channels=None, padding='SAME', biased=True, depthwise=False, scope=None,
                   dilation=(1, 1), ws=False, kernel_paddings=((0, 0), (0, 0)), weight_initializer=None, bias_initializer=None,
                   verbose=False):
        """reference convnet.py:1597-1706 -> tf.nn.conv2d (:1659) [+ tf.nn.bias_add (:1694)]."""
        kernel, stride, dilation = _pair(kernel), _pair(stride), _pair(dilation)
        n, h, w, cin = x.shape
        if out_channels is None:
            out_channels = cin
        if depthwise:
            return self._depthwise_conv_layer(x, kernel, stride, out_channels, padding, biased, scope, dilation, ws, kernel_paddings,
                                              weight_initializer, bias_initializer)
        oh = out_size(h, kernel[0], stride[0], padding, dilation[0])
        ow = out_size(w, kernel[1], stride[1], padding, dilation[1])
        if padding.upper() == 'SAME':
            pt, pb = same_pads(h, kernel[0], stride[0], dilation[0])
            pl, pr = same_pads(w, kernel[1], stride[1], dilation[1])
        else:
            pt = pb = pl = pr = 0
        with self.variable_scope(scope) if scope is not None else nullcontext():
            wv = self.weight_variable([kernel[0], kernel[1], cin, out_channels], initializer=weight_initializer,
                                      weight_standardization=ws, paddings=kernel_paddings)
            bv = self.bias_variable(out_channels, initializer=bias_initializer) if biased else None
            name = self.scope_name()
        y = self.graph.tensor((n, oh, ow, out_channels), x.dtype, name + '/conv', self._channel_first)
        geom = _ffi.conv_geom(n, h, w, cin, out_channels, kernel[0], kernel[1], stride[0], stride[1], dilation[0], dilation[1],
                              (pt, pb, pl, pr), x.cs)
        self.graph.node('conv', [x], [y], scope=name, geom=geom, w=wv, b=bv, has_params=True)
        # the reference's report uses np.ceil on floats for out_size; identical for integer inputs (convnet.py:1626-1664)
        flops = oh * ow * kernel[0] * kernel[1] * cin * out_channels
        params = kernel[0] * kernel[1] * cin * out_channels
        self._conv_macs += flops
        if biased:
            flops += oh * ow * out_channels
            params += out_channels
        self._log_layer(name, [None, oh, ow, out_channels], flops, params, oh * ow * out_channels)
        return y

    def _depthwise_conv_layer(self, x, kernel, stride, out_channels, padding, biased, scope, dilation, ws, kernel_paddings,
                              weight_initializer, bias_initializer):
        """reference convnet.py:1634-1650 -> tf.nn.depthwise_conv2d (:1645), filter [kh, kw, cin, multiplier]."""
        n, h, w, cin = x.shape
        mult = max(out_channels // cin, 1)
        if biased and out_channels != cin * mult:             # (tf.nn.bias_add would refuse the shapes: convnet.py:1679,1694)
            raise ValueError('biased depthwise convolution: {} biases for {} output channels'.format(out_channels, cin * mult))
        oh = out_size(h, kernel[0], stride[0], padding, dilation[0])
        ow = out_size(w, kernel[1], stride[1], padding, dilation[1])
        if padding.upper() == 'SAME':
            pt, pb = same_pads(h, kernel[0], stride[0], dilation[0])
            pl, pr = same_pads(w, kernel[1], stride[1], dilation[1])
        else:
            pt = pb = pl = pr = 0
        with self.variable_scope(scope) if scope is not None else nullcontext():
            wv = self.weight_variable([kernel[0], kernel[1], cin, mult], initializer=weight_initializer, weight_standardization=ws,
                                      paddings=kernel_paddings)
            bv = self.bias_variable(out_channels, initializer=bias_initializer) if biased else None
            name = self.scope_name()
        cm = cin * mult
        if mult != 1:
            # channel multiplier (convnet.py:1635-1645, output channel c * mult + q): the multiplier-1 kernels on the input with every channel
            # repeated `mult` times — the [kh, kw, cin, mult] filter read as [kh, kw, cin * mult] is already in that order (include/mcn.h)
            xr = self.graph.tensor((n, h, w, cm), x.dtype, name + '/repeat', self._channel_first)
            self.graph.node('chrepeat', [x], [xr], scope=name, mult=mult)
            x = xr
        y = self.graph.tensor((n, oh, ow, cm), x.dtype, name + '/dwconv', self._channel_first)
        geom = _ffi.conv_geom(n, h, w, cm, cm, kernel[0], kernel[1], stride[0], stride[1], dilation[0], dilation[1], (pt, pb, pl, pr), 0)
        self.graph.node('dwconv', [x], [y], scope=name, geom=geom, w=wv, has_params=True)
        flops = oh * ow * kernel[0] * kernel[1] * cm
        params = kernel[0] * kernel[1] * cm
        if biased:
            # tf.nn.bias_add behind the depthwise convolution (convnet.py:1678-1694): its own pass (the stored conv output is rounded to the
            # storage type before the add, as TF's two ops do)
            yb = self.graph.tensor((n, oh, ow, cm), x.dtype, name + '/bias_add', self._channel_first)
            self.graph.node('biasadd', [y], [yb], scope=name, b=bv, has_params=True)
            y = yb
            flops += oh * ow * cm
            params += cm
        self._log_layer(name, [None, oh, ow, cm], flops, params, oh * ow * cm)
        return y

    def conv_bn_act(self, x, kernel, stride, out_channels=None, padding='SAME', biased=False, depthwise=False, scope=None,
                    dilation=(1, 1), ws=False, kernel_paddings=((0, 0), (0, 0)), weight_initializer=None, bias_initializer=None,
                    scale=True, shift=True, zero_scale_init=False, epsilon=1e-3, act_type='relu', act_params=None, verbose=False):
        """reference convnet.py:1550-1595."""
        with self.variable_scope(scope) if scope is not None else nullcontext():
            x = self.conv_layer(x, kernel, stride, out_channels, padding=padding, biased=biased, depthwise=depthwise, dilation=dilation,
                                ws=ws, kernel_paddings=kernel_paddings, weight_initializer=weight_initializer,
                                bias_initializer=bias_initializer)
            x = self.batch_norm(x, scale=scale, shift=shift, zero_scale_init=zero_scale_init, epsilon=epsilon)
            x = self.activation(x, activation_type=act_type, params=act_params)
        return x

    def fc_layer(self, x, out_dim, biased=True, scope=None, ws=False, weight_initializer=None, bias_initializer=None, verbose=False):
        """reference convnet.py:1708-1755 -> tf.matmul(x, W) + b."""
        in_dim = int(x.shape[-1])
        with self.variable_scope(scope) if scope is not None else nullcontext():
            wv = self.weight_variable([in_dim, out_dim], initializer=weight_initializer, weight_standardization=ws)
            bv = self.bias_variable(out_dim, initializer=bias_initializer) if biased else None
            name = self.scope_name()
        y = self.graph.tensor((x.shape[0], out_dim), x.dtype, name + '/fc')
        self.graph.node('fc', [x], [y], scope=name, w=wv, b=bv, has_params=True)
        flops = in_dim * out_dim + (out_dim if biased else 0)
        self._log_layer(name, [None, out_dim], flops, flops, out_dim)
        return y

    def normalization(self, x, norm_type='batch', norm_param=None, scale=True, shift=True, zero_scale_init=False, epsilon=1e-3,
                      scope='norm'):
        """reference convnet.py:1757-1778."""
        if norm_type is None:
            return x
        if norm_type.lower() == 'batch':
            return self.batch_norm(x, scale=scale, shift=shift, zero_scale_init=zero_scale_init, epsilon=epsilon, scope=scope)
        raise NotImplementedError('normalization type {} is outside the built path (supported: batch)'.format(norm_type))

    def batch_norm(self, x, scale=True, shift=True, zero_scale_init=False, epsilon=1e-3, scope='bn'):
        """reference convnet.py:1780-1926 -> tf.nn.fused_batch_norm + running-statistics update."""
        if isinstance(self._update_batch_norm, bool):
            update = self._update_batch_norm
        else:
            update = self._trainable_here()
        trainable = self._trainable_here()
        c = x.shape[-1]
        with self.variable_scope(scope):
            mu = self._new_variable('mu', c, 'mu', zeros(), False)
            sigma = self._new_variable('sigma', c, 'sigma', ones(), False)       # running VARIANCE despite the name
            for v in (mu, sigma):
                self.add_to_collection('norm_statistics', v)
                self.add_to_collection('block_{}/norm_statistics'.format(self._curr_block), v)
            gamma = beta = None
            if scale:
                gamma = self._new_variable('gamma', c, 'gamma', zeros() if zero_scale_init else ones(), trainable)
                self._params += c
            if shift:
                beta = self._new_variable('beta', c, 'beta', zeros(), trainable)
                self._params += c
                self._flops += x.shape[1] * x.shape[2] * c if len(x.shape) == 4 else c
            for v in (gamma, beta):
                if v is not None:
                    self.add_to_collection('norm_variables', v)
                    self.add_to_collection('block_{}/norm_variables'.format(self._curr_block), v)
            name = self.scope_name()
        y = self.graph.tensor(x.shape, x.dtype, name, self._channel_first)
        self.graph.node('bn', [x], [y], scope=name, gamma=gamma, beta=beta, mu=mu, sigma=sigma, eps=epsilon,
                        momentum=self._batch_norm_decay, update=update, has_params=True)
        return y

    def _pool(self, op, x, side_l, stride, padding):
        side_l, stride = _pair(side_l), _pair(stride)
        n, h, w, c = x.shape
        oh = out_size(h, side_l[0], stride[0], padding)
        ow = out_size(w, side_l[1], stride[1], padding)
        if padding.upper() == 'SAME':
            pt, _ = same_pads(h, side_l[0], stride[0])
            pl, _ = same_pads(w, side_l[1], stride[1])
        else:
            pt = pl = 0
        name = self.scope_name(op)
        y = self.graph.tensor((n, oh, ow, c), x.dtype, name, self._channel_first)
        self.graph.node(op, [x], [y], scope=name, kh=side_l[0], kw=side_l[1], sh=stride[0], sw=stride[1], pt=pt, pl=pl)
        self._flops += side_l[0] * side_l[1] * oh * ow * c
        self._nodes += oh * ow * c
        self._layer_info.append({'name': name, 'shape': [None, oh, ow, c], 'flops': int(side_l[0] * side_l[1] * oh * ow * c), 'params': 0,
                                 'nodes': int(oh * ow * c)})
        return y

    def max_pool(self, x, side_l, stride, padding='SAME'):
        """reference convnet.py:1472-1509 -> tf.nn.max_pool."""
        return self._pool('maxpool', x, side_l, stride, padding)

    def avg_pool(self, x, side_l, stride, padding='SAME'):
        """reference convnet.py:1511-1548 -> tf.nn.avg_pool."""
        return self._pool('avgpool', x, side_l, stride, padding)

    def pooling_layer(self, x, kernel, stride, padding='SAME', pooling_type='AVG'):
        """reference convnet.py:1464-1470."""
        if pooling_type.lower() == 'avg':
            return self.avg_pool(x, kernel, stride, padding=padding)
        if pooling_type.lower() == 'max':
            return self.max_pool(x, kernel, stride, padding=padding)
        raise ValueError('Pooling type of {} is not supported'.format(pooling_type))

    def global_avg_pool(self, x, keepdims=False):
        """Stand-in for tf.reduce_mean(x, axis=[1, 2][, keepdims=True]) at models/resnet_v1_5.py:72-73 and
        models/efficientnet.py:183."""
        n, h, w, c = x.shape
        y = self.graph.tensor((n, 1, 1, c) if keepdims else (n, c), x.dtype, self.scope_name('avgpool'))
        self.graph.node('gap', [x], [y], scope=self.scope_name())
        return y

    def upsampling_2d_layer(self, x, scale=2, out_shape=None, align_corners=False, force_unaligned=False, upsampling_method='bilinear',
                            name='upsampling'):
        """reference convnet.py:2378-2406 -> tf.image.resize_bilinear (interpolation in fp32, result in the compute dtype)."""
        if upsampling_method.lower() != 'bilinear':
            raise NotImplementedError('upsampling method {} is outside the built path (supported: bilinear)'.format(upsampling_method))
        if force_unaligned:
            raise NotImplementedError('force_unaligned (legacy asymmetric resize) is outside the built path')
        n, h, w, c = x.shape
        oh, ow = (h * scale, w * scale) if out_shape is None else (int(out_shape[0]), int(out_shape[1]))
        y = self.graph.tensor((n, oh, ow, c), x.dtype, self.scope_name(name), self._channel_first)
        self.graph.node('resize', [x], [y], scope=self.scope_name(name), align=bool(align_corners))
        return y

    def concat(self, xs, name='concat'):
        """Stand-in for tf.concat(xs, axis=channel axis) (models/deeplabv3plus.py:101,110)."""
        xs = list(xs)
        assert all(t.shape[:-1] == xs[0].shape[:-1] and t.dtype == xs[0].dtype for t in xs), 'concat: shapes differ'
        y = self.graph.tensor(tuple(xs[0].shape[:-1]) + (sum(t.shape[-1] for t in xs),), xs[0].dtype, self.scope_name(name), self._channel_first)
        self.graph.node('concat', xs, [y], scope=self.scope_name(name))
        return y

    def stop_gradient(self, x):
        """Stand-in for tf.stop_gradient (models/deeplabv3plus.py:50-53): the value of x, no gradient path into it.  A graph node whose output
        SHARES x's storage (nothing is copied or launched) and never needs a gradient, so every consumer's backward skips its data gradient."""
        y = self.graph.tensor(x.shape, x.dtype, self.scope_name('stop_gradient'), self._channel_first)
        y.cs = x.cs
        y.share_of = x
        self.graph.node('stopgrad', [x], [y], scope=self.scope_name())
        return y

    def channel_scale(self, x, mask):
        """Stand-in for `x = x*se_mask` (models/efficientnet.py:161): mask [N,1,1,C] broadcast over H, W."""
        assert mask.shape[0] == x.shape[0] and mask.shape[-1] == x.shape[-1] and mask.numel == x.shape[0] * x.shape[-1]
        y = self.graph.tensor(x.shape, x.dtype, self.scope_name('se_scale'), self._channel_first)
        self.graph.node('chscale', [x, mask], [y], scope=self.scope_name())
        return y

    def _random_mask(self, x, kind, rate, name):
        """y = x * mask, mask drawn on the host each training step (sample_random_masks); identity in evaluation.
        kind 'sample': survived[n]/(1-rate) (stochastic depth); 'element': keep[n,c]/(1-rate) (dropout on [N,C])."""
        y = self.graph.tensor(x.shape, x.dtype, self.scope_name(name), self._channel_first)
        nd = self.graph.node('mulmask', [x], [y], scope=self.scope_name(name), kind=kind, rate=float(rate))
        self._random_nodes.append(nd)
        return y

    def _allocate_masks(self):
        """All dropout / stochastic-depth masks of the model live in ONE device buffer (views per node) with two pinned host
        staging buffers: a step uploads them with a single asynchronous copy instead of one pageable, host-blocking copy per
        node (EfficientNet-B0: 17 per step, which kept the host from running ahead of the GPU)."""
        dev = self.device
        if not self._random_nodes:
            return
        tdt = TORCH_DT[self._random_nodes[0].inputs[0].dtype]
        offs, total = [], 0
        for nd in self._random_nodes:
            x = nd.inputs[0]
            ce = 4 if x.dtype == 'float32' else 8
            cols = ce if nd.attrs['kind'] == 'sample' else x.shape[-1]
            offs.append((total, x.shape[0], cols))
            total += (x.shape[0] * cols + 7) // 8 * 8                     # 16-byte aligned views
        self._mask_dev = torch.ones(total, dtype=tdt, device=dev)
        pin = dev.type == 'cuda'
        self._mask_host = [torch.ones(total, dtype=tdt).pin_memory() if pin else torch.ones(total, dtype=tdt) for _ in range(2)]
        self._mask_events = [None, None]
        self._mask_turn = 0
        for nd, (o, rows, cols) in zip(self._random_nodes, offs):
            nd.attrs['mask'] = self._mask_dev[o:o + rows * cols].view(rows, cols)
            nd.attrs['mask_off'] = o

    def sample_random_masks(self, masks=None):
        """Draw the dropout / stochastic-depth masks of the next training step (tf.random.uniform at convnet.py:2507,
        tf.nn.dropout) and upload them (one stream-ordered copy from a pinned buffer).  `masks` ({node scope: array})
        overrides the draw (parity tests feed the oracle's masks)."""
        if masks is None:
            masks = getattr(self, 'fixed_random_masks', None)      # tests pin the draw
        if not self._random_nodes or getattr(self, '_mask_dev', None) is None:
            return                                                  # not compiled yet
        turn = self._mask_turn
        self._mask_turn ^= 1
        ev = self._mask_events[turn]
        if ev is not None:
            ev.synchronize()                                        # the copy that last read this staging buffer (two steps ago)
        host = self._mask_host[turn]
        for nd in self._random_nodes:
            x = nd.inputs[0]
            rate = nd.attrs['rate']
            dev = nd.attrs['mask']
            if masks is not None and nd.scope in masks:
                keep = np.asarray(masks[nd.scope], dtype=np.float32)
            elif nd.attrs['kind'] == 'sample':
                keep = (self._mask_rng.random(x.shape[0]) >= rate).astype(np.float32) / (1.0 - rate)
            else:
                keep = (self._mask_rng.random((x.shape[0], x.shape[-1])) >= rate).astype(np.float32) / (1.0 - rate)
            if nd.attrs['kind'] == 'sample':
                keep = np.repeat(keep.reshape(-1, 1), dev.shape[1], axis=1)
            o = nd.attrs['mask_off']
            host[o:o + dev.numel()].copy_(torch.from_numpy(np.ascontiguousarray(keep).reshape(-1)))
        self._mask_dev.copy_(host, non_blocking=True)
        if self.device.type == 'cuda':
            if ev is None:
                ev = self._mask_events[turn] = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))

    def dropout(self, x, rate):
        """Stand-in for tf.nn.dropout at models/resnet_v1_5.py:75 / models/efficientnet.py:121 (on the pooled [N,C]
        features); rate 0 (the default) is the identity."""
        if rate and rate > 0.0:
            if len(x.shape) != 2:
                raise NotImplementedError('dropout is built for the pooled [N, C] features only')
            return self._random_mask(x, 'element', rate, 'dropout')
        return x

    def softmax(self, x):
        """Stand-in for tf.nn.softmax at models/resnet_v1_5.py:78: `pred` is produced by the fused loss kernel."""
        return x

    def stochastic_depth(self, x, skip, drop_rate=0.0, name='drop'):
        """reference convnet.py:2500-2512; drop_rate 0 -> x + skip."""
        if drop_rate > 0.0:
            with self.variable_scope(name):
                x = self._random_mask(x, 'sample', drop_rate, 'survived')
        assert x.shape == skip.shape, 'residual shapes differ: {} vs {}'.format(x.shape, skip.shape)
        y = self.graph.tensor(x.shape, x.dtype, self.scope_name('add'), self._channel_first)
        self.graph.node('add', [x, skip], [y], scope=self.scope_name())
        return y

    def activation(self, x, activation_type='relu', params=None):
        """reference convnet.py:2514-2534."""
        if activation_type is None:
            return x
        act = activation_type.lower()
        if act == 'relu':
            return self.relu(x, name=activation_type)
        if act == 'swish':
            return self.swish(x, name=activation_type)
        if act == 'sigmoid':
            return self.sigmoid(x, name=activation_type)
        if act == 'relu6':
            return self.relu6(x, name=activation_type)
        if act == 'lrelu' or act == 'leaky_relu':
            return self.lrelu(x, alpha=params, name=activation_type)
        if act == 'tanh':
            return self.tanh(x, name=activation_type)
        raise ValueError('Activation type of {} is not supported. Supported types: {}'
                         .format(activation_type, ['relu', 'relu6', 'lrelu', 'tanh', 'sigmoid', 'swish']))

    def _act(self, x, kind, name, param=None):
        y = self.graph.tensor(x.shape, x.dtype, self.scope_name(name), self._channel_first)
        attrs = dict(kind=kind) if param is None else dict(kind=kind, param=float(param))
        self.graph.node('act', [x], [y], scope=self.scope_name(), **attrs)
        return y

    def relu6(self, x, name='relu6'):
        """reference convnet.py:2539-2540 -> tf.nn.relu6."""
        return self._act(x, _ffi.ACT_RELU6, name)

    def lrelu(self, x, alpha=None, name='lrelu'):
        """reference convnet.py:2542-2545 -> tf.nn.leaky_relu(x, alpha), alpha defaults to 0.2."""
        return self._act(x, _ffi.ACT_LRELU, name, param=0.2 if alpha is None else alpha)

    def tanh(self, x, name='tanh'):
        """reference convnet.py:2547 -> tf.nn.tanh."""
        return self._act(x, _ffi.ACT_TANH, name)

    def sigmoid(self, x, name=None):
        """reference convnet.py:2549-2550 -> tf.nn.sigmoid."""
        return self._act(x, _ffi.ACT_SIGMOID, name or 'sigmoid')

    def swish(self, x, name='swish'):
        """reference convnet.py:2552-2556: x*sigmoid(x)."""
        return self._act(x, _ffi.ACT_SWISH, name)

    def relu(self, x, name='relu'):
        """reference convnet.py:2536-2537 -> tf.nn.relu."""
        y = self.graph.tensor(x.shape, x.dtype, self.scope_name(name), self._channel_first)
        self.graph.node('relu', [x], [y], scope=self.scope_name())
        return y
