"""
ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/ops.py header).

Stand-alone NumPy restatement of the networks and of one training step, written directly from
the reference's model files so that the topology is stated independently of the product's
graph builder:

  * ResNet (basic and bottleneck, v1.5)   — models/resnet_v1_5.py:9-203
  * VGG-16/19 trunk (+ head at 224x224)   — models/vggnet.py:11-139
  * loss                                  — convnet.py:528-601
  * optimizer step, EMA, BN running stats — optimizers.py:89-177, 668-677; convnet.py:183-184,
                                            1898-1914

Variables are addressed by the reference's scope names, e.g. 'block_1/res_0/conv_1/weights',
'block_1/res_0/conv_1/bn/gamma', 'block_None/logits/biases'.
"""
import numpy as np
from . import ops


class V(object):
    """A value on the oracle's tape: array + accumulated gradient.  `q` (optional) rounds activations and their
    gradients to the storage precision of the low-precision mode (see Tape.quant)."""
    __slots__ = ('a', 'g', 'q', 'lazy', 'fused')

    def __init__(self, a, q=None, lazy=None, fused=False):
        """lazy (fused rounding mode): the rounding function of a value that is NOT stored yet — the output of a BN (or residual add)
        that the device keeps in registers until the activation behind it: relu / swish / add consume it unrounded, any other reader
        rounds it first (Tape._mat).  Such a value has no stored gradient either."""
        self.q = q
        self.lazy = lazy
        self.fused = fused
        self.a = a if q is None else q(a)
        self.g = None

    def acc(self, g):
        if self.q is None:
            self.g = g if self.g is None else self.g + g
        elif self.fused:
            # the device adds a contribution to the stored gradient inside the producing kernel's epilogue: the fp32 result is added to
            # the stored value and the SUM is rounded (mcn_conv2d_dgrad accumulate / _addmasked) — one rounding per contribution
            self.g = self.q(g if self.g is None else self.g + g)
        else:
            g = self.q(g)
            self.g = g if self.g is None else self.q(self.g + g)


def trainable_name(name, blocks_to_train):
    """convnet.py:1384-1389: a variable trains iff its block (the `block_<id>` prefix of its name; the logits live in
    block None) is in blocks_to_train, or blocks_to_train is None."""
    if blocks_to_train is None:
        return True
    tok = name.split('/')[0]
    assert tok.startswith('block_'), name
    blk = None if tok == 'block_None' else int(tok[len('block_'):])
    return blk in blocks_to_train


class Tape(object):
    def __init__(self, params, train=True, bn_stats=None, eps=1e-3, quant=None, blocks_to_train=None, update_batch_norm=None, fused_rounding=False):
        """quant: optional rounding function emulating low-precision STORAGE of activations / activation gradients and
        the per-use cast of the weights (reference half_precision structure, convnet.py:63,1421-1422,1878-1879: fp32
        master weights, BN statistics and all accumulation in fp32).  Used to compare the bf16 device path like with like.
        fused_rounding: round where the DEVICE rounds: BN + activation, BN + residual add + ReLU and the
        shortcut-affine + add form are one pass with ONE rounding of the result (the op-by-op mode rounds after BN, after the add and
        after the activation), and a gradient contribution is added to the stored gradient unrounded (see V.acc)."""
        self.quant = quant
        self.fused = bool(fused_rounding) and quant is not None
        self.params = params          # name -> ndarray
        self.pv = {}                  # name -> V (created on first use)
        self.train = train
        self.bn_stats = bn_stats      # name -> ndarray (mu / sigma), used when train=False (or by a frozen BN)
        self.blocks_to_train = blocks_to_train        # None = all; list of block ids (None = the logits block)
        self.update_batch_norm = update_batch_norm    # True / False, or None = follow blocks_to_train (convnet.py:1781-1789)
        self.eps = eps
        self.bw = []                  # backward closures
        self.batch_stats = {}         # bn scope -> (batch_mean, batch_var_unbiased)
        self.d = {}
        # ReLU decisions at near-ties (test infrastructure for the fp32 device comparison, tests/flip_util.py): a pre-activation within
        # rounding of zero may fall on the other side of the ReLU on the device (fp32 sums) than here (float64).  `tie_tol` > 0 records the
        # elements with |z| <= tie_tol * rms(z) as (ReLU ordinal, flat index) in `near_ties`; `relu_flips` {ordinal: flat indices} inverts
        # the decision of those elements (forward value and gradient mask).
        # teacher forcing at residual-unit boundaries (test infrastructure for the low-precision device comparison): `force_act[name]` replaces
        # the activation at barrier `name`, `force_grad[name]` the gradient that flows back through it — with the device's stored tensors
        # there, rounding differences cannot cascade beyond one residual unit (the tiny test nets amplify one bf16 ulp 20-50x over their depth)
        self.force_act = {}
        self.force_grad = {}
        self.tie_tol = 0.0
        self.near_ties = []
        self.relu_flips = {}
        self.relu_count = 0

    def p(self, name):
        if name not in self.pv:
            self.pv[name] = V(self.params[name])
        return self.pv[name]

    def _v(self, a):
        """a stored activation: rounded to the storage type"""
        return V(a, self.quant, fused=self.fused)

    def _mat(self, x):
        """fused rounding mode: a value the device has not stored yet is stored (rounded) for this reader"""
        if getattr(x, 'lazy', None) is not None:
            x.a = x.lazy(x.a)
            x.q, x.lazy, x.fused = self.quant, None, self.fused
            if x.g is not None:
                x.g = x.q(x.g)
        return x

    # ---- ops ---------------------------------------------------------------------------
    def conv(self, x, scope, stride, padding='SAME', dilation=1, biased=False):
        self._mat(x)
        w = self.p(scope + '/weights')
        q = self.quant
        wq = w.a if q is None else q(w.a)
        y = V(ops.conv2d_fwd(x.a, wq, stride, padding, dilation), q, fused=self.fused)

        def bw():
            w.acc(ops.conv2d_wgrad(x.a, y.g, w.a.shape, stride, padding, dilation))
            if x.g is not False:
                x.acc(ops.conv2d_dgrad(y.g, wq, x.a.shape, stride, padding, dilation))
        self.bw.append(bw)
        if biased:
            b = self.p(scope + '/biases')
            y2 = V(ops.bias_add_fwd(y.a, b.a), q, fused=self.fused)

            def bwb():
                b.acc(ops.bias_add_bwd(y2.g))
                y.acc(y2.g)
            # bias backward must run before conv backward -> append after (reverse order)
            self.bw.append(bwb)
            return y2
        return y

    def bn_updates(self, scope):
        if isinstance(self.update_batch_norm, bool):
            return self.update_batch_norm
        return trainable_name(scope, self.blocks_to_train)

    def bn(self, x, scope):
        self._mat(x)
        gamma = self.p(scope + '/gamma')
        beta = self.p(scope + '/beta')
        if self.train and not self.bn_updates(scope):
            # frozen statistics (convnet.py:1915-1923): fused_batch_norm(is_training=False) on the running mean / variance
            # also while training; its gradient is the plain affine one
            mu, sigma = self.bn_stats[scope + '/mu'], self.bn_stats[scope + '/sigma']
            y = V(ops.bn_fwd_infer(x.a, gamma.a, beta.a, mu, sigma, self.eps), self.quant, fused=self.fused)

            def bwf():
                dx, dg, db = ops.bn_bwd_frozen(y.g, x.a, gamma.a, mu, sigma, self.eps)
                x.acc(dx)
                gamma.acc(dg)
                beta.acc(db)
            self.bw.append(bwf)
            return y
        if self.train:
            ya, bm, bv, sm, si = ops.bn_fwd_train(x.a, gamma.a, beta.a, self.eps)
            self.batch_stats[scope] = (bm, bv)
            y = V(ya, None, lazy=self.quant) if self.fused else V(ya, self.quant)

            def bw():
                dx, dg, db = ops.bn_bwd(y.g, x.a, gamma.a, sm, si)
                x.acc(dx)
                gamma.acc(dg)
                beta.acc(db)
            self.bw.append(bw)
            return y
        mu = self.bn_stats[scope + '/mu']
        sigma = self.bn_stats[scope + '/sigma']
        return V(ops.bn_fwd_infer(x.a, gamma.a, beta.a, mu, sigma, self.eps), self.quant, fused=self.fused)

    def relu(self, x):
        k = self.relu_count
        self.relu_count += 1
        if self.tie_tol > 0.0:
            z = np.abs(np.asarray(x.a, np.float64)).ravel()
            rms = float(np.sqrt(np.mean(z * z))) or 1.0
            self.near_ties += [(k, int(i)) for i in np.flatnonzero(z <= self.tie_tol * rms)]
        flips = self.relu_flips.get(k)
        if flips is None:
            y = V(ops.relu_fwd(x.a), self.quant, fused=self.fused)
            self.bw.append(lambda: x.acc(ops.relu_bwd(y.g, y.a)))
            return y
        mask = (x.a > 0)
        mask.reshape(-1)[np.asarray(flips, dtype=np.int64)] ^= True
        y = V(x.a * mask, self.quant, fused=self.fused)
        self.bw.append(lambda: x.acc(y.g * mask))
        return y

    def barrier(self, x, name):
        """identity on the value and its gradient unless the tape holds a forced activation / gradient for `name` (see __init__)"""
        if name not in self.force_act and name not in self.force_grad:
            return x
        self._mat(x)
        y = V(np.asarray(self.force_act.get(name, x.a), dtype=x.a.dtype), None)
        y.q, y.fused = x.q, x.fused                    # (a forced value is a stored tensor: already in the storage type)
        self.bw.append(lambda: x.acc(np.asarray(self.force_grad[name], dtype=x.a.dtype) if name in self.force_grad else y.g))
        return y

    def add(self, x, skip):
        if self.fused:
            # device: y = act(bn(x) + skip) in one pass — the skip operand is a stored tensor (a shortcut BN's output is rounded to the
            # storage type where the stored tensor would have been), the sum stays in registers
            self._mat(skip)
            y = V(ops.add_fwd(x.a, skip.a), None, lazy=self.quant)
        else:
            y = V(ops.add_fwd(x.a, skip.a), self.quant)

        def bw():
            x.acc(y.g)
            skip.acc(y.g)
        self.bw.append(bw)
        return y

    def dwconv(self, x, scope, stride, padding='SAME', dilation=1):
        self._mat(x)
        w = self.p(scope + '/weights')
        q = self.quant
        wq = w.a if q is None else q(w.a)
        y = V(ops.depthwise_conv2d_fwd(x.a, wq, stride, padding, dilation), q, fused=self.fused)

        def bw():
            w.acc(ops.depthwise_conv2d_wgrad(x.a, y.g, w.a.shape, stride, padding, dilation))
            if x.g is not False:
                x.acc(ops.depthwise_conv2d_dgrad(y.g, wq, x.a.shape, stride, padding, dilation))
        self.bw.append(bw)
        return y

    def swish(self, x):
        y = V(ops.swish_fwd(x.a), self.quant, fused=self.fused)
        self.bw.append(lambda: x.acc(ops.swish_bwd(y.g, x.a)))
        return y

    def sigmoid(self, x):
        self._mat(x)
        y = V(ops.sigmoid_fwd(x.a), self.quant, fused=self.fused)
        self.bw.append(lambda: x.acc(ops.sigmoid_bwd(y.g, y.a)))
        return y

    def scale_channels(self, x, m):
        """x * m, m: [N,1,1,C] (SE mask)."""
        self._mat(x)
        self._mat(m)
        y = V(ops.channel_scale_fwd(x.a, m.a), self.quant, fused=self.fused)

        def bw():
            dx, dm = ops.channel_scale_bwd(y.g, x.a, m.a)
            x.acc(dx)
            m.acc(dm.reshape(m.a.shape))
        self.bw.append(bw)
        return y

    def scale_samples(self, x, s):
        self._mat(x)
        """x * s[n] with a constant per-sample factor (stochastic depth survival / (1 - rate))."""
        y = V(ops.sample_scale_fwd(x.a, s.astype(x.a.dtype)), self.quant, fused=self.fused)
        self.bw.append(lambda: x.acc(ops.sample_scale_fwd(y.g, s.astype(x.a.dtype))))
        return y

    def mul_const(self, x, m):
        self._mat(x)
        """x * m with a constant mask of x's shape (tf.nn.dropout with the keep mask already scaled by 1/(1-rate))."""
        m = m.astype(x.a.dtype)
        y = V(x.a * m, self.quant, fused=self.fused)
        self.bw.append(lambda: x.acc(y.g * m))
        return y

    def mean_keepdims(self, x):
        self._mat(x)
        """tf.reduce_mean(x, [1,2], keepdims=True) (models/efficientnet.py:183)."""
        n, c = x.a.shape[0], x.a.shape[-1]
        y = V(ops.global_avgpool_fwd(x.a).reshape(n, 1, 1, c), self.quant, fused=self.fused)
        self.bw.append(lambda: x.acc(ops.global_avgpool_bwd(y.g.reshape(n, c), x.a.shape)))
        return y

    def stop_gradient(self, x):
        """tf.stop_gradient (models/deeplabv3plus.py:53): the stored value of x, nothing flows back (g = False: consumers skip their data gradient)."""
        self._mat(x)
        y = V(x.a, None, fused=self.fused)
        y.q = x.q
        y.g = False
        return y

    def resize(self, x, out_hw, align_corners=True):
        self._mat(x)
        y = V(ops.resize_bilinear_fwd(x.a, out_hw, align_corners), self.quant, fused=self.fused)
        self.bw.append(lambda: x.acc(ops.resize_bilinear_bwd(y.g, x.a.shape, align_corners)))
        return y

    def concat(self, xs):
        for v_ in xs:
            self._mat(v_)
        y = V(ops.concat_fwd([v.a for v in xs]), self.quant, fused=self.fused)

        def bw():
            for v, g in zip(xs, ops.concat_bwd(y.g, [v.a.shape[-1] for v in xs])):
                v.acc(g)
        self.bw.append(bw)
        return y

    def max_pool(self, x, k, s, padding='SAME'):
        self._mat(x)
        ya, arg = ops.maxpool_fwd(x.a, k, s, padding)
        y = V(ya, self.quant, fused=self.fused)
        self.bw.append(lambda: x.acc(ops.maxpool_bwd(y.g, arg, x.a.shape, k, s, padding)))
        return y

    def avg_pool(self, x, k, s, padding='SAME'):
        self._mat(x)
        y = V(ops.avgpool_fwd(x.a, k, s, padding), self.quant, fused=self.fused)
        self.bw.append(lambda: x.acc(ops.avgpool_bwd(y.g, x.a.shape, k, s, padding)))
        return y

    def global_avgpool(self, x):
        self._mat(x)
        y = V(ops.global_avgpool_fwd(x.a), self.quant, fused=self.fused)
        self.bw.append(lambda: x.acc(ops.global_avgpool_bwd(y.g, x.a.shape)))
        return y

    def fc(self, x, scope):
        self._mat(x)
        w = self.p(scope + '/weights')
        b = self.p(scope + '/biases')
        q = self.quant
        wq = w.a if q is None else q(w.a)
        y = V(ops.fc_fwd(x.a, wq, b.a), q, fused=self.fused)

        def bw():
            dx, dw, db = ops.fc_bwd(y.g, x.a, wq)
            x.acc(dx)
            w.acc(dw)
            b.acc(db)
        self.bw.append(bw)
        return y

    def backward(self):
        for f in reversed(self.bw):
            f()
        return {name: v.g for name, v in self.pv.items()}


# ------------------------------------------------------------------------------------------------
# ResNet v1.5  (models/resnet_v1_5.py)
# ------------------------------------------------------------------------------------------------
class ResNetSpec(object):
    def __init__(self, channels=(64, 256, 512, 1024, 2048), kernels=(7, 3, 3, 3, 3), strides=(2, 1, 2, 2, 2),
                 res_units=(None, 3, 4, 6, 3), bottleneck=True, num_classes=1000, in_channels=3,
                 backbone_only=False, dilations=None, multi_grid=(1, 2, 4)):
        self.channels = list(channels)
        self.kernels = list(kernels)
        self.strides = list(strides)
        self.res_units = list(res_units)
        self.bottleneck = bottleneck
        self.num_classes = num_classes
        self.in_channels = in_channels
        self.backbone_only = backbone_only
        self.dilations = list(dilations) if dilations is not None else None    # models/resnet_v1_5_dilated.py:11-12,63-67
        self.multi_grid = list(multi_grid)

    def unit_dilation(self, i, j):
        if self.dilations is None or self.dilations[i] == 1:
            return 1
        return self.dilations[i] * self.multi_grid[j % len(self.multi_grid)]

    @staticmethod
    def resnet50(num_classes=1000, width_div=1):
        return ResNetSpec(channels=[64 // width_div] + [c // width_div for c in (256, 512, 1024, 2048)],
                          num_classes=num_classes)

    @staticmethod
    def resnet18(num_classes=1000, width_div=1):
        return ResNetSpec(channels=[c // width_div for c in (64, 64, 128, 256, 512)],
                          res_units=(None, 2, 2, 2, 2), bottleneck=False, num_classes=num_classes)

    # -- variable inventory in creation order (the order tf.trainable_variables() would list) ------
    def variables(self):
        """[(name, shape, kind)], kind in {'weight','bias','gamma','gamma0','beta','mu','sigma'}."""
        out = []

        def conv(scope, k, cin, cout):
            out.append((scope + '/weights', (k, k, cin, cout), 'weight'))

        def bn(scope, c, zero=False):
            out.append((scope + '/mu', (c,), 'mu'))
            out.append((scope + '/sigma', (c,), 'sigma'))
            out.append((scope + '/gamma', (c,), 'gamma0' if zero else 'gamma'))
            out.append((scope + '/beta', (c,), 'beta'))

        ch = self.channels
        conv('block_0/conv_0', self.kernels[0], self.in_channels, ch[0])
        bn('block_0/conv_0/bn', ch[0])
        cin = ch[0]
        for i in range(1, len(ch)):
            for j in range(self.res_units[i]):
                name = 'block_{}/res_{}'.format(i, j)
                cout = ch[i]
                if cin != cout:
                    conv(name + '/conv_skip', 1, cin, cout)
                    bn(name + '/conv_skip/bn', cout)
                if self.bottleneck:
                    conv(name + '/conv_0', 1, cin, cout // 4)
                    bn(name + '/conv_0/bn', cout // 4)
                    conv(name + '/conv_1', self.kernels[i], cout // 4, cout // 4)
                    bn(name + '/conv_1/bn', cout // 4)
                    conv(name + '/conv_2', 1, cout // 4, cout)
                    bn(name + '/conv_2/bn', cout, zero=True)
                else:
                    conv(name + '/conv_0', self.kernels[i], cin, cout)
                    bn(name + '/conv_0/bn', cout)
                    conv(name + '/conv_1', 3, cout, cout)
                    bn(name + '/conv_1/bn', cout)
                cin = cout
        if not self.backbone_only:
            out.append(('block_None/logits/weights', (cin, self.num_classes), 'weight'))
            out.append(('block_None/logits/biases', (self.num_classes,), 'bias'))
        return out

    def forward(self, t, x):
        """x: V holding the prepared NHWC input.  Mirrors ResNet._build_model / _res_unit."""
        d = t.d
        ch = self.channels
        h = t.conv(x, 'block_0/conv_0', self.strides[0])
        d['block_0/conv_0'] = h
        h = t.bn(h, 'block_0/conv_0/bn')
        d['block_0/conv_0/bn'] = h
        h = t.relu(h)
        h = t.max_pool(h, 3, 2, 'SAME')
        h = t.barrier(h, 'block_0')
        d['block_0'] = h
        cin = ch[0]
        for i in range(1, len(ch)):
            for j in range(self.res_units[i]):
                s = self.strides[i] if j == 0 else 1
                name = 'block_{}/res_{}'.format(i, j)
                cout = ch[i]
                if cin == cout:
                    skip = t.max_pool(h, s, s, 'VALID') if s > 1 else h
                else:
                    skip = t.conv(h, name + '/conv_skip', s)
                    skip = t.bn(skip, name + '/conv_skip/bn')
                if self.bottleneck:
                    y = t.conv(h, name + '/conv_0', 1)
                    y = t.relu(t.bn(y, name + '/conv_0/bn'))
                    y = t.conv(y, name + '/conv_1', s, 'SAME', self.unit_dilation(i, j))   # v1.5: stride on the 3x3; dilated variants
                    d[name + '/conv_1'] = y
                    y = t.relu(t.bn(y, name + '/conv_1/bn'))
                    y = t.conv(y, name + '/conv_2', 1)
                    y = t.bn(y, name + '/conv_2/bn')
                else:
                    y = t.conv(h, name + '/conv_0', s)
                    y = t.relu(t.bn(y, name + '/conv_0/bn'))
                    y = t.conv(y, name + '/conv_1', 1)
                    y = t.bn(y, name + '/conv_1/bn')
                h = t.relu(t.add(y, skip))
                h = t.barrier(h, name)
                d[name] = h
                cin = cout
            d['block_{}'.format(i)] = h
        if self.backbone_only:
            return h
        h = t.global_avgpool(h)
        d['logits/avgpool'] = h
        logits = t.fc(h, 'block_None/logits')
        d['logits'] = logits
        return logits


# ------------------------------------------------------------------------------------------------
# EfficientNet  (models/efficientnet.py) — SURVEY §8f-2
# ------------------------------------------------------------------------------------------------
class EfficientNetSpec(object):
    def __init__(self, channels=(32, 16, 24, 40, 80, 112, 192, 320, 1280), kernels=(3, 3, 3, 5, 3, 5, 5, 3, None),
                 strides=(2, 1, 2, 2, 2, 1, 2, 1, None), conv_units=(None, 1, 2, 2, 3, 3, 4, 1, None),
                 multipliers=(None, 1, 6, 6, 6, 6, 6, 6, None), se_reduction=4, num_classes=1000, in_channels=3,
                 backbone_only=False, initial_drop_rate=0.0, final_drop_rate=0.0):
        self.channels, self.kernels, self.strides = list(channels), list(kernels), list(strides)
        self.conv_units, self.multipliers = list(conv_units), list(multipliers)
        self.se_reduction = se_reduction
        self.num_classes = num_classes
        self.in_channels = in_channels
        self.backbone_only = backbone_only
        self.initial_drop_rate, self.final_drop_rate = initial_drop_rate, final_drop_rate
        self.survival = {}            # unit name -> per-sample factor array (set by the test when drop rates > 0)
        self.dropout_mask = None      # [N, C] keep/(1-rate) mask on the pooled features (models/efficientnet.py:121)

    @staticmethod
    def b0(num_classes=1000, width_div=1, depth_div=1):
        """width_div / depth_div shrink the net for tests (not reference options)."""
        ch = [max(8, c // width_div) if c is not None else None for c in (32, 16, 24, 40, 80, 112, 192, 320, 1280)]
        units = [None if u is None else max(1, u // depth_div) for u in (None, 1, 2, 2, 3, 3, 4, 1, None)]
        return EfficientNetSpec(channels=ch, conv_units=units, num_classes=num_classes)

    def units(self):
        """[(name, kernel, stride, cin, cout, multiplier, drop_rate)] in build order (efficientnet.py:72-88)."""
        out = []
        nb = len(self.channels)
        cin = self.channels[0]
        for i in range(1, nb - 1):
            dr = self.initial_drop_rate + (self.final_drop_rate - self.initial_drop_rate) * i / (nb - 2)
            for j in range(self.conv_units[i]):
                s = self.strides[i] if j == 0 else 1
                out.append(('block_{}/mbconv_{}'.format(i, j), self.kernels[i], s, cin, self.channels[i], self.multipliers[i], dr))
                cin = self.channels[i]
        return out

    def variables(self):
        out = []

        def conv(scope, k, cin, cout, biased=False):
            out.append((scope + '/weights', (k, k, cin, cout), 'weight_fanout'))
            if biased:
                out.append((scope + '/biases', (cout,), 'bias'))

        def bn(scope, c, zero=False):
            out.append((scope + '/mu', (c,), 'mu'))
            out.append((scope + '/sigma', (c,), 'sigma'))
            out.append((scope + '/gamma', (c,), 'gamma0' if zero else 'gamma'))
            out.append((scope + '/beta', (c,), 'beta'))

        conv('block_0/conv_0', self.kernels[0], self.in_channels, self.channels[0])
        bn('block_0/conv_0/norm', self.channels[0])
        for name, k, s, cin, cout, mult, _ in self.units():
            mid = cin * mult
            has_skip = s == 1 and cin == cout
            if mult > 1:
                conv(name + '/conv_0', 1, cin, mid)
                bn(name + '/conv_0/norm', mid)
            out.append((name + '/conv_1/weights', (k, k, mid, 1), 'weight_fanout'))
            bn(name + '/conv_1/norm', mid)
            red = mid // (mult * self.se_reduction)
            conv(name + '/se_mask/conv_0', 1, mid, red, biased=True)
            conv(name + '/se_mask/conv_1', 1, red, mid, biased=True)
            conv(name + '/conv_2', 1, mid, cout)
            bn(name + '/conv_2/norm', cout, zero=has_skip)
        last = 'block_{}'.format(len(self.channels) - 1)
        conv(last + '/conv_0', 1, self.channels[-2], self.channels[-1])
        bn(last + '/conv_0/norm', self.channels[-1])
        if not self.backbone_only:
            out.append(('block_None/logits/weights', (self.channels[-1], self.num_classes), 'weight_fc_uniform'))
            out.append(('block_None/logits/biases', (self.num_classes,), 'bias'))
        return out

    def forward(self, t, x):
        d = t.d
        h = t.conv(x, 'block_0/conv_0', self.strides[0])
        h = t.swish(t.bn(h, 'block_0/conv_0/norm'))
        d['block_0'] = h
        for name, k, s, cin, cout, mult, dr in self.units():
            skip = h if (s == 1 and cin == cout) else None
            y = h
            if mult > 1:
                y = t.conv(y, name + '/conv_0', 1)
                y = t.swish(t.bn(y, name + '/conv_0/norm'))
            y = t.dwconv(y, name + '/conv_1', s)
            d[name + '/conv_1'] = y
            y = t.swish(t.bn(y, name + '/conv_1/norm'))
            m = t.mean_keepdims(y)
            m = t.swish(t.conv(m, name + '/se_mask/conv_0', 1, biased=True))
            m = t.sigmoid(t.conv(m, name + '/se_mask/conv_1', 1, biased=True))
            d[name + '/se_mask'] = m
            y = t.scale_channels(y, m)
            y = t.conv(y, name + '/conv_2', 1)
            y = t.bn(y, name + '/conv_2/norm')
            if skip is not None:
                if dr > 0.0 and t.train:
                    y = t.scale_samples(y, self.survival[name])
                y = t.add(y, skip)
            d[name] = y
            h = y
        last = 'block_{}'.format(len(self.channels) - 1)
        h = t.conv(h, last + '/conv_0', 1)
        h = t.swish(t.bn(h, last + '/conv_0/norm'))
        d[last] = h
        if self.backbone_only:
            return h
        h = t.global_avgpool(h)
        d['logits/avgpool'] = h
        if self.dropout_mask is not None and t.train:
            h = t.mul_const(h, self.dropout_mask)
        logits = t.fc(h, 'block_None/logits')
        d['logits'] = logits
        return logits


# ------------------------------------------------------------------------------------------------
# DeepLabv3+ on a dilated ResNet  (models/deeplabv3plus.py, models/resnet_v1_5_dilated.py, segmentation/segnet.py) — §8f-3
# ------------------------------------------------------------------------------------------------
class DeepLabSpec(object):
    segmentation = True

    def __init__(self, num_classes=19, width_div=1, depth_div=1, strides=(2, 1, 2, 2, 2), res_units=(None, 3, 4, 6, 3),
                 dilations=(None, 1, 1, 1, 2), aspp_dilations=(6, 12, 18), aspp_level_feature=False, feature_gradients=(None, True)):
        """Defaults = ResNet50OS16 (resnet_v1_5_dilated.py:7-12,145); ResNet101OS16: strides (2,1,2,2,1), units (None,3,4,23,3)."""
        units = [None if u is None else max(1, u // depth_div) for u in res_units]
        self.backbone = ResNetSpec(channels=[64 // width_div] + [c // width_div for c in (256, 512, 1024, 2048)], strides=strides,
                                   res_units=units, bottleneck=True, num_classes=num_classes, backbone_only=True, dilations=dilations)
        self.num_classes = num_classes
        self.backbone_only = False
        self.feature_channels = [max(8, 256 // width_div), max(8, 48 // width_div)] if width_div > 1 else [256, 48]
        self.aspp_dilations = list(aspp_dilations)
        self.aspp_level_feature = bool(aspp_level_feature)        # models/deeplabv3plus.py:90-99 (the reference's default: off)
        self.feature_gradients = list(feature_gradients)          # models/deeplabv3plus.py:50-53: False = tf.stop_gradient on that backbone feature

    def variables(self):
        out = self.backbone.variables()

        def conv_bn(scope, k, cin, cout):
            out.append((scope + '/weights', (k, k, cin, cout), 'weight'))
            for nm, kind in (('mu', 'mu'), ('sigma', 'sigma'), ('gamma', 'gamma'), ('beta', 'beta')):
                out.append((scope + '/norm/' + nm, (cout,), kind))
        c4, c1 = self.backbone.channels[4], self.backbone.channels[1]
        fa, fd = self.feature_channels
        conv_bn('block_5/aspp/conv_0', 1, c4, fa)
        for i in range(len(self.aspp_dilations)):
            conv_bn('block_5/aspp/conv_{}'.format(i + 1), 3, c4, fa)
        if self.aspp_level_feature:
            conv_bn('block_5/aspp/conv_pool', 1, c4, fa)
        conv_bn('block_5/aspp/conv_out', 1, fa * (1 + len(self.aspp_dilations) + (1 if self.aspp_level_feature else 0)), fa)
        conv_bn('block_6/features', 1, c1, fd)
        conv_bn('block_6/decoder/conv_0', 3, fa + fd, fa)
        out.append(('block_None/logits/weights', (1, 1, fa, self.num_classes), 'weight'))
        out.append(('block_None/logits/biases', (self.num_classes,), 'bias'))
        return out

    def forward(self, t, x):
        self.backbone.forward(t, x)
        d = t.d
        f4, f1 = d['block_4'], d['block_1']
        ys = [t.bn(t.conv(f4, 'block_5/aspp/conv_0', 1), 'block_5/aspp/conv_0/norm')]
        for i, dil in enumerate(self.aspp_dilations):
            sc = 'block_5/aspp/conv_{}'.format(i + 1)
            ys.append(t.bn(t.conv(f4, sc, 1, 'SAME', dil), sc + '/norm'))
        if self.aspp_level_feature:
            y = t.bn(t.conv(t.mean_keepdims(f4), 'block_5/aspp/conv_pool', 1), 'block_5/aspp/conv_pool/norm')
            ys.append(t.resize(y, f4.a.shape[1:3], False))
        h = t.bn(t.conv(t.concat(ys), 'block_5/aspp/conv_out', 1), 'block_5/aspp/conv_out/norm')
        d['block_5'] = h
        if not self.feature_gradients[1]:
            f1 = t.stop_gradient(f1)
        feat = t.bn(t.conv(f1, 'block_6/features', 1), 'block_6/features/norm')
        h = t.resize(h, feat.a.shape[1:3], True)
        h = t.bn(t.conv(t.concat([h, feat]), 'block_6/decoder/conv_0', 1), 'block_6/decoder/conv_0/norm')
        d['block_6'] = h
        h = t.conv(h, 'block_None/logits', 1, biased=True)
        logits = t.resize(h, x.a.shape[1:3], True)
        d['logits'] = logits
        return logits


# ------------------------------------------------------------------------------------------------
# VGG  (models/vggnet.py)
# ------------------------------------------------------------------------------------------------
VGG_MEAN = np.array([123.68, 116.78, 103.94])


class VGGSpec(object):
    def __init__(self, num_layers=16, num_classes=10, backbone_only=True, width_div=1):
        self.num_layers = num_layers
        self.num_classes = num_classes
        self.backbone_only = backbone_only
        self.plan = [[64, 64], [128, 128], [256] * (3 if num_layers == 16 else 4),
                     [512] * (3 if num_layers == 16 else 4), [512] * (3 if num_layers == 16 else 4)]
        self.plan = [[c // width_div for c in blk] for blk in self.plan]

    def variables(self):
        out = []
        cin = 3
        for b, blk in enumerate(self.plan):
            for j, c in enumerate(blk):
                out.append(('block_{}/conv_{}/weights'.format(b, j), (3, 3, cin, c), 'weight'))
                out.append(('block_{}/conv_{}/biases'.format(b, j), (c,), 'bias'))
                cin = c
        return out

    def forward(self, t, x, image_mean=0.5, scale_factor=2.0):
        """vggnet.py:23-25 re-scales the prepared input: (X/scale + mean)*255 - VGG_MEAN."""
        d = t.d
        xin = V(((x.a / scale_factor + image_mean) * 255.0 - VGG_MEAN.astype(x.a.dtype)).astype(x.a.dtype), t.quant)
        xin.g = False
        h = xin
        for b, blk in enumerate(self.plan):
            for j, _ in enumerate(blk):
                h = t.relu(t.conv(h, 'block_{}/conv_{}'.format(b, j), 1, 'SAME', biased=True))
            h = t.max_pool(h, 2, 2, 'SAME')
            d['block_{}'.format(b)] = h
        return h


# ------------------------------------------------------------------------------------------------
# initialisation (deterministic stand-in; the TF RNG stream is not reproducible — SURVEY §8c (10))
# ------------------------------------------------------------------------------------------------
def init_variables(var_list, seed=0, dtype=np.float32):
    """He-normal (truncated at 2 sigma, std = sqrt(2/fan_in)/0.8796) for weights; zeros for biases,
    beta, mu; ones for gamma, sigma; zeros for zero_scale_init gammas (convnet.py:1382,1805-1854)."""
    rng = np.random.default_rng(seed)
    params, stats = {}, {}
    for name, shape, kind in var_list:
        if kind == 'weight':
            fan_in = int(np.prod(shape[:-1]))
            std = np.sqrt(2.0 / fan_in) / 0.87962566103423978
            w = rng.standard_normal(shape)
            bad = np.abs(w) > 2
            while bad.any():
                w[bad] = rng.standard_normal(int(bad.sum()))
                bad = np.abs(w) > 2
            params[name] = (w * std).astype(dtype)
        elif kind == 'weight_fanout':
            # tf.initializers.variance_scaling(mode='fan_out'): truncated normal, std = sqrt(1/fan_out)/0.8796
            # (models/efficientnet.py:20); fan_out = shape[-1] * receptive field
            fan_out = int(np.prod(shape[:-2])) * shape[-1]
            std = np.sqrt(1.0 / fan_out) / 0.87962566103423978
            w = rng.standard_normal(shape)
            bad = np.abs(w) > 2
            while bad.any():
                w[bad] = rng.standard_normal(int(bad.sum()))
                bad = np.abs(w) > 2
            params[name] = (w * std).astype(dtype)
        elif kind == 'weight_fc_uniform':
            # variance_scaling(scale=1/3, mode='fan_out', distribution='uniform') (models/efficientnet.py:21-23)
            limit = np.sqrt(3.0 * (1.0 / 3.0) / shape[-1])
            params[name] = rng.uniform(-limit, limit, shape).astype(dtype)
        elif kind in ('bias', 'beta'):
            params[name] = np.zeros(shape, dtype)
        elif kind == 'gamma':
            params[name] = np.ones(shape, dtype)
        elif kind == 'gamma0':
            params[name] = np.zeros(shape, dtype)
        elif kind == 'mu':
            stats[name] = np.zeros(shape, dtype)
        elif kind == 'sigma':
            stats[name] = np.ones(shape, dtype)
    return params, stats


# ------------------------------------------------------------------------------------------------
# one training step  (Optimizer._step -> session.run of optimization_operation, optimizers.py:565-606)
# ------------------------------------------------------------------------------------------------
DEFAULT_HP = dict(image_mean=0.5, scale_factor=2.0, l2_reg=1e-4, momentum=0.9, base_learning_rate=0.1,
                  moving_average_decay=0.99, batch_norm_decay=0.99, label_smoothing=0.0,
                  base_weight_decay=0.0, loss_scaling_factor=1.0, eps=1e-3)


def _regularised(name, hp):
    """Member of the set the L2 term and the decoupled decay run over: collection 'weight_variables', plus biases and BN
    gamma / beta with bias_norm_decay (convnet.py:535-537, optimizers.py:149-151)."""
    return name.endswith('/weights') or bool(hp.get('bias_norm_decay', False))


class TrainState(object):
    def __init__(self, params, stats):
        self.params = {k: v.copy() for k, v in params.items()}
        self.stats = {k: v.copy() for k, v in stats.items()}
        self.accum = {k: np.zeros_like(v) for k, v in params.items()}
        self.ema = {k: v.copy() for k, v in params.items()}          # shadow starts at the initial value
        self.ema_stats = {k: v.copy() for k, v in stats.items()}
        self.step = 0


def forward_loss(spec, state, x_raw, y_float, hp=None, train=True, use_ema=False, quant=None, tie_tol=0.0, relu_flips=None, fused_rounding=False,
                 force_act=None, force_grad=None):
    hp = dict(DEFAULT_HP, **(hp or {}))
    params = state.ema if use_ema else state.params
    stats = state.ema_stats if use_ema else state.stats
    t = Tape(params, train=train, bn_stats=stats, eps=hp['eps'], quant=quant, blocks_to_train=hp.get('blocks_to_train'),
             update_batch_norm=hp.get('update_batch_norm'), fused_rounding=fused_rounding)
    t.tie_tol = float(tie_tol)
    t.relu_flips = dict(relu_flips or {})
    t.force_act, t.force_grad = dict(force_act or {}), dict(force_grad or {})
    dt = next(iter(params.values())).dtype
    x = V(ops.input_prep(x_raw.astype(dt), hp['image_mean'], hp['scale_factor']), quant, fused=t.fused)
    x.g = False
    if isinstance(spec, VGGSpec):
        out = spec.forward(t, x, hp['image_mean'], hp['scale_factor'])
    else:
        out = spec.forward(t, x)
    if spec.backbone_only:
        return t, out, None, None, None
    if getattr(spec, 'segmentation', False):
        # SegNet: labels [N,H,W], per-pixel CE averaged over all pixels, ignored pixels weigh 0 (segnet.py:31-50, convnet.py:528-597)
        onehot = ops.seg_one_hot_labels(y_float, spec.num_classes, dtype=dt)
        c = spec.num_classes
        ls_f = float(hp.get('label_smoothing', 0.0))
        avg = ops.avgpool_fwd(onehot, 5, 1, 'SAME').reshape(-1, c) if ls_f > 0.0 else None       # segnet.py:117-122
        pred, sm_loss, ce, dlogits = ops.softmax_xent_fwd_bwd(out.a.reshape(-1, c).astype(dt), onehot.reshape(-1, c), None, ls_f, avg_labels=avg,
                                                              focal_gamma=float(hp.get('focal_loss_factor', 0.0)),
                                                              sigmoid_focal_alpha=float(hp.get('sigmoid_focal_loss_factor', 0.0)))
        pred, dlogits = pred.reshape(out.a.shape), dlogits.reshape(out.a.shape)
    else:
        onehot = ops.one_hot_labels(y_float, spec.num_classes, dtype=dt)
        pred, sm_loss, ce, dlogits = ops.softmax_xent_fwd_bwd(out.a, onehot, None, hp['label_smoothing'], focal_gamma=float(hp.get('focal_loss_factor', 0.0)),
                                                              sigmoid_focal_alpha=float(hp.get('sigmoid_focal_loss_factor', 0.0)))
    weights = [v for k, v in params.items() if _regularised(k, hp)]
    loss = float(sm_loss) + ops.l2_reg_loss(weights, hp['l2_reg'])
    if hp.get('l1_reg', 0.0) > 0.0:                                            # convnet.py:553-557
        loss += ops.l1_reg_loss(weights, hp['l1_reg'])
    # loss scaling (optimizers.py:102-111): the loss is multiplied by the factor (only when > 1) before differentiation, so every
    # activation gradient is stored scaled — what matters under fp16 storage — and the parameter gradients are divided again
    ls = float(hp.get('loss_scaling_factor', 1.0))
    t.loss_scale = ls if ls > 1.0 else 1.0
    dlogits = dlogits * t.loss_scale
    t._mat(out)
    out.g = dlogits if quant is None else quant(dlogits)
    return t, out, pred, loss, onehot


def copy_state(state):
    """independent copy of a TrainState (to replay a step from the same starting point)"""
    c = TrainState(state.params, state.stats)
    c.accum = {k: v.copy() for k, v in state.accum.items()}
    c.ema = {k: v.copy() for k, v in state.ema.items()}
    c.ema_stats = {k: v.copy() for k, v in state.ema_stats.items()}
    c.step = state.step
    return c


def train_step(spec, state, x_raw, y_float, hp=None, lr_mult=1.0, batch_total=None,
               tower_batches=None, quant=None, probe=None, fused_rounding=False):
    """One optimisation step.  `tower_batches` (list of (x,y)) restates the multi-tower path:
    gradients averaged over towers (optimizers.py:125-142), BN running stats chained
    (convnet.py:1899-1909), loss = mean of tower losses (convnet.py:510).
    `probe` (test infrastructure, tests/flip_util.py): dict with 'tie_tol' and / or 'relu_flips' {(tower, ReLU ordinal): flat indices}, 'force_act' /
    'force_grad' {barrier name: array} (Tape.barrier: teacher forcing at the residual-unit boundaries, single tower);
    on return probe['near_ties'] lists the (tower, ordinal, flat index) of the ReLU inputs within tie_tol * rms of zero."""
    hp = dict(DEFAULT_HP, **(hp or {}))
    towers = tower_batches if tower_batches is not None else [(x_raw, y_float)]
    btot = batch_total if batch_total is not None else sum(len(t_[0]) for t_ in towers)
    lr = hp['base_learning_rate'] * btot / 256.0 * lr_mult          # optimizers.py:46,57
    grads_sum, losses, preds, bstats = None, [], [], []
    if probe is not None:
        probe['near_ties'] = []
    for ti, (xr, yf) in enumerate(towers):
        flips = {k[1]: v for k, v in (probe or {}).get('relu_flips', {}).items() if k[0] == ti}
        t, out, pred, loss, _ = forward_loss(spec, state, xr, yf, hp, train=True, quant=quant, tie_tol=(probe or {}).get('tie_tol', 0.0), relu_flips=flips,
                                             fused_rounding=fused_rounding, force_act=(probe or {}).get('force_act'), force_grad=(probe or {}).get('force_grad'))
        if probe is not None:
            probe['near_ties'] += [(ti, k, i) for k, i in t.near_ties]
        g = t.backward()
        if t.loss_scale != 1.0:
            g = {k: v / t.loss_scale for k, v in g.items()}
        g = {k: v for k, v in g.items() if trainable_name(k, hp.get('blocks_to_train'))}      # update_vars = tf.trainable_variables(), optimizers.py:53,106
        if hp.get('gradient_threshold') is not None:
            # the reference differentiates the full loss (CE + L2) and clips per tower (optimizers.py:106-113)
            # (with l1_reg the L1 term l1 * sum |w| is part of that loss too, convnet.py:553-557: its gradient l1 * sign(w) is clipped with the rest)
            l1 = hp.get('l1_reg', 0.0)
            g = {k: (v + hp['l2_reg'] * state.params[k] + (l1 * np.sign(state.params[k]) if l1 > 0.0 else 0.0) if _regularised(k, hp) else v) for k, v in g.items()}
            g, _ = ops.clip_by_global_norm(g, hp['gradient_threshold'])
        grads_sum = g if grads_sum is None else {k: grads_sum[k] + g[k] for k in g}
        losses.append(loss)
        preds.append(pred)
        bstats.append(t.batch_stats)
    ntow = len(towers)
    grads = {k: v / ntow for k, v in grads_sum.items()}
    d = ops.ema_decay(hp['moving_average_decay'], state.step)
    m = hp['batch_norm_decay']
    # EMA of running stats (pre-assign value), then the chained running-stat update
    for key in state.stats:                                    # ema.apply covers every statistic, also of frozen BNs (convnet.py:1812,1826)
        state.ema_stats[key] = d * state.ema_stats[key] + (1.0 - d) * state.stats[key]
    for scope in bstats[0]:
        mu, sg = ops.bn_running_update_chain(state.stats[scope + '/mu'], state.stats[scope + '/sigma'],
                                             [b[scope][0] for b in bstats], [b[scope][1] for b in bstats], m)
        state.stats[scope + '/mu'] = mu.astype(state.stats[scope + '/mu'].dtype)
        state.stats[scope + '/sigma'] = sg.astype(state.stats[scope + '/sigma'].dtype)
    wd = hp['base_weight_decay'] * btot / 256.0 * (lr_mult if hp.get('weight_decay_scheduling', True) else 1.0)   # optimizers.py:91,155-156
    for k in state.params:
        if k not in grads:                                     # frozen: only its EMA shadow moves (towards the constant value)
            state.ema[k] = (d * state.ema[k] + (1.0 - d) * state.params[k]).astype(state.params[k].dtype)
            continue
        is_w = _regularised(k, hp)
        if is_w and hp.get('l1_reg', 0.0) > 0.0 and hp.get('gradient_threshold') is None:   # d/dw l1 * |w| = l1 * sign(w): part of the gradient the update sees (and of the returned dict,
            grads[k] = grads[k] + hp['l1_reg'] * np.sign(state.params[k])      # like the device's flat buffer after mcn_l1_grad_h; one tower)
        w, a, e = ops.sgd_nesterov_step(state.params[k], grads[k], state.accum[k], lr, hp['momentum'],
                                        l2=hp['l2_reg'] if (is_w and hp.get('gradient_threshold') is None) else 0.0, ema=state.ema[k], ema_d=d,
                                        wd=wd if is_w else 0.0, l1_decay=bool(hp.get('l1_weight_decay', False)),
                                        huber_delta=hp.get('huber_decay_delta'))
        dt = state.params[k].dtype
        state.params[k], state.accum[k], state.ema[k] = w.astype(dt), a.astype(dt), e.astype(dt)
    state.step += 1
    return float(np.mean(losses)), np.concatenate(preds, axis=0), grads
