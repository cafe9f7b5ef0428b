"""ResNet v1.5 family on the MI355X building blocks.

Topology, scope names ('block_i/res_j/conv_k', '.../bn', 'block_None/logits') and block-method calls follow
reference models/resnet_v1_5.py:9-209 (stride on the 3x3 of a bottleneck = "v1.5", zero-initialised gamma on the
last BN of a bottleneck, 1x1/stride projection shortcut when the width changes, max-pool identity shortcut
otherwise), so parameters are addressable by the reference's variable names.  The unit bodies are described
by small tables instead of hand-unrolled code.
"""
from .convnet import ConvNet


class ResNet(ConvNet):
    """Basic-block network (ResNet-18 layout).  Subclasses change the stage table."""
    bottleneck = False

    def _init_params(self, **kwargs):
        self.channels = [64, 64, 128, 256, 512]
        self.kernels = [7, 3, 3, 3, 3]
        self.strides = [2, 1, 2, 2, 2]
        self.res_units = [None, 2, 2, 2, 2]
        self.norm_type = kwargs.get('norm_type', 'batch')
        self.norm_param = kwargs.get('norm_param', None)
        self.erase_relu = kwargs.get('erase_relu', False)
        self.initial_drop_rate = kwargs.get('initial_drop_rate', 0.0)
        self.final_drop_rate = kwargs.get('final_drop_rate', 0.0)
        self._width_div = int(kwargs.get('width_div', 1))      # test-only channel reduction (not in the reference)

    def _apply_width(self):
        if self._width_div > 1:
            self.channels = [c // self._width_div for c in self.channels]

    # -- helpers -----------------------------------------------------------------------------------------
    def _norm(self, x, **kw):
        return self.normalization(x, shift=True, scale=True, scope='bn', norm_type=self.norm_type,
                                  norm_param=self.norm_param, **kw)

    def _unit_plan(self, kernel, stride, out_channels):
        """[(scope, kernel, stride, channels, relu_after, zero_gamma)] for the residual branch."""
        if self.bottleneck:
            mid = out_channels // 4
            return [('conv_0', 1, 1, mid, True, False),
                    ('conv_1', kernel, stride, mid, True, False),       # v1.5: the 3x3 carries the stride
                    ('conv_2', 1, 1, out_channels, False, True)]
        return [('conv_0', kernel, stride, out_channels, True, False),
                ('conv_1', 3, 1, out_channels, False, False)]

    def _res_unit(self, x, kernel, stride, out_channels, d, drop_rate=0.0, name='res_unit'):
        in_channels = x.shape[-1]
        sh, sw = (stride if isinstance(stride, (list, tuple)) else (stride, stride))[:2] if not isinstance(stride, int) else (stride, stride)
        with self.variable_scope(name):
            if in_channels == out_channels:
                skip = self.max_pool(x, [sh, sw], [sh, sw], padding='VALID') if (sh > 1 or sw > 1) else x
            else:
                with self.variable_scope('conv_skip'):
                    skip = self.conv_layer(x, 1, [sh, sw], out_channels, padding='SAME', biased=False)
                    skip = self._norm(skip)
            d[name + '/branch'] = skip
            for scope, k, s, c, relu_after, zero_gamma in self._unit_plan(kernel, [sh, sw], out_channels):
                with self.variable_scope(scope):
                    x = self.conv_layer(x, k, s, c, padding='SAME', biased=False)
                    d[name + '/' + scope] = x
                    x = self._norm(x, zero_scale_init=True) if zero_gamma else self._norm(x)
                    d[name + '/' + scope + '/bn'] = x
                    if relu_after:
                        x = self.relu(x, name='relu')
                        d[name + '/' + scope + '/relu'] = x
            x = self.stochastic_depth(x, skip, drop_rate=drop_rate)
            if not self.erase_relu:
                x = self.relu(x, name='relu')
            d[name] = x
        return x

    def _build_model(self):
        d = dict()
        self._apply_width()
        num_blocks = min(len(self.channels), len(self.kernels), len(self.strides), len(self.res_units))
        self._curr_block = 0
        with self.variable_scope('block_0'):
            with self.variable_scope('conv_0'):
                x = self.conv_layer(self.X, self.kernels[0], self.strides[0], self.channels[0], padding='SAME', biased=False)
                d['block_0/conv_0'] = x
                x = self._norm(x)
                d['block_0/conv_0/bn'] = x
                x = self.relu(x, name='relu')
                d['block_0/conv_0/relu'] = x
                x = self.max_pool(x, 3, 2, padding='SAME')
                d['block_0/conv_0/maxpool'] = x
            d['block_0'] = x
        for i in range(1, num_blocks):
            self._curr_block = i
            dr = self.initial_drop_rate + (self.final_drop_rate - self.initial_drop_rate) * i / (num_blocks - 1)
            for j in range(self.res_units[i]):
                x = self._res_unit(x, self.kernels[i], self.strides[i] if j == 0 else 1, self.channels[i], d,
                                   drop_rate=dr, name='block_{}/res_{}'.format(i, j))
            d['block_{}'.format(i)] = x
        if self.backbone_only is False:
            self._curr_block = None
            with self.variable_scope('block_None'):
                with self.variable_scope('logits'):
                    if self.erase_relu:
                        x = self.relu(x, name='relu')
                    x = self.global_avg_pool(x)
                    d['logits/avgpool'] = x
                    x = self.dropout(x, rate=self.dropout_rate_features)
                    x = self.fc_layer(x, self.num_classes)
                    d['logits'] = x
                    d['pred'] = self.softmax(x)
        return d


class ResNetBot(ResNet):
    """Bottleneck network (ResNet-50 layout)."""
    bottleneck = True

    def _init_params(self, **kwargs):
        super()._init_params(**kwargs)
        self.channels = [64, 256, 512, 1024, 2048]
        self.res_units = [None, 3, 4, 6, 3]


class ResNet18(ResNet):
    pass


class ResNet34(ResNet):
    def _init_params(self, **kwargs):
        super()._init_params(**kwargs)
        self.res_units = [None, 3, 4, 6, 3]


class ResNet50(ResNetBot):
    pass


class ResNet101(ResNetBot):
    def _init_params(self, **kwargs):
        super()._init_params(**kwargs)
        self.res_units = [None, 3, 4, 23, 3]
