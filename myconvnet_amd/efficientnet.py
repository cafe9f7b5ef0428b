"""EfficientNet family on the MI355X building blocks.

Topology, scope names ('block_i/mbconv_j/conv_k', '.../norm', '.../se_mask/conv_k', 'block_None/logits'), stage
tables, compound-scaling rules and initialisers follow reference models/efficientnet.py:10-290, so parameters are
addressable by the reference's variable names.  The MBConv body is table-driven here.  Raw TF calls of the reference
model file have stand-ins on ConvNet: tf.reduce_mean(keepdims=True) -> global_avg_pool(keepdims=True),
`x*se_mask` -> channel_scale, tf.nn.dropout -> dropout.
"""
import math

from .convnet import ConvNet, variance_scaling


class EfficientNet(ConvNet):
    width_coefficient = 1.0
    depth_coefficient = 1.0

    def _init_params(self, **kwargs):
        self.channels = [32, 16, 24, 40, 80, 112, 192, 320, 1280]
        self.kernels = [3, 3, 3, 5, 3, 5, 5, 3, None]
        self.strides = [2, 1, 2, 2, 2, 1, 2, 1, None]
        self.conv_units = [None, 1, 2, 2, 3, 3, 4, 1, None]
        self.multipliers = [None, 1, 6, 6, 6, 6, 6, 6, None]
        self.se_reduction = 4
        self.conv_initializer = variance_scaling(mode='fan_out')
        self.fc_initializer = variance_scaling(scale=1.0 / 3.0, mode='fan_out', distribution='uniform')
        self.norm_type = kwargs.get('norm_type', 'batch')
        self.norm_param = kwargs.get('norm_param', None)
        self.striding_kernel_offset = kwargs.get('striding_kernel_offset', 0)
        self.striding_kernel_size = kwargs.get('striding_kernel_size', None)
        self.initial_drop_rate = kwargs.get('initial_drop_rate', 0.0)
        self.final_drop_rate = kwargs.get('final_drop_rate', 0.0)
        if self.width_coefficient != 1.0 or self.depth_coefficient != 1.0 or type(self) is not EfficientNetB0:
            self.channels = self._calc_widths(self.channels, self.width_coefficient)
            self.conv_units = self._calc_depths(self.conv_units, self.depth_coefficient)
        wd, dd = int(kwargs.get('width_div', 1)), int(kwargs.get('depth_div', 1))     # test-only reductions (not in the reference)
        if wd > 1:
            self.channels = [None if c is None else max(8, c // wd) for c in self.channels]
        if dd > 1:
            self.conv_units = [None if u is None else max(1, u // dd) for u in self.conv_units]

    # reference models/efficientnet.py:199-219
    def _calc_widths(self, widths, coefficient):
        divisor = 8
        out = []
        for w in widths:
            if w is None:
                out.append(None)
                continue
            w = coefficient * w
            new_w = max(divisor, (int(w + divisor / 2) // divisor) * divisor)
            if new_w < 0.9 * w:
                new_w += divisor
            out.append(new_w)
        return out

    def _calc_depths(self, depths, coefficient):
        return [None if d is None else int(math.ceil(coefficient * d)) for d in depths]

    def _strided_kernel(self, k, s):
        if s > 1:
            return k + self.striding_kernel_offset if self.striding_kernel_size is None else self.striding_kernel_size
        return k

    def _conv_norm_swish(self, x, kernel, stride, channels, d, key, depthwise=False, act=True, zero_scale_init=False):
        x = self.conv_layer(x, kernel, stride, channels, padding='SAME', biased=False, depthwise=depthwise,
                            weight_initializer=self.conv_initializer, verbose=True)
        d[key] = x
        x = self.normalization(x, shift=True, scale=True, zero_scale_init=zero_scale_init, scope='norm', norm_type=self.norm_type,
                               norm_param=self.norm_param)
        d[key + '/norm'] = x
        if act:
            x = self.swish(x, name='swish')
            d[key + '/swish'] = x
        return x

    def _build_model(self):
        d = dict()
        num_blocks = min(len(self.channels), len(self.kernels), len(self.strides), len(self.conv_units), len(self.multipliers))
        self._curr_block = 0
        with self.variable_scope('block_0'):
            with self.variable_scope('conv_0'):
                x = self._conv_norm_swish(self.X, self._strided_kernel(self.kernels[0], self.strides[0]), self.strides[0], self.channels[0],
                                          d, 'block_0/conv_0')
            d['block_0'] = x
        for i in range(1, num_blocks - 1):
            self._curr_block = i
            dr = self.initial_drop_rate + (self.final_drop_rate - self.initial_drop_rate) * i / (num_blocks - 2)
            for j in range(self.conv_units[i]):
                s = self.strides[i] if j == 0 else 1
                k = self._strided_kernel(self.kernels[i], s)
                x = self._mb_conv_unit(x, k, s, self.channels[i], self.multipliers[i], d, drop_rate=dr, name='block_{}/mbconv_{}'.format(i, j))
            d['block_{}'.format(self._curr_block)] = x
        self._curr_block += 1
        with self.variable_scope('block_{}'.format(self._curr_block)):
            with self.variable_scope('conv_0'):
                x = self._conv_norm_swish(x, 1, 1, self.channels[-1], d, 'logits/conv_0')
        d['block_{}'.format(self._curr_block)] = x
        if self.backbone_only is False:
            self._curr_block = None
            with self.variable_scope('block_{}'.format(self._curr_block)):
                with self.variable_scope('logits'):
                    x = self.global_avg_pool(x)
                    d['logits/avgpool'] = x
                    if self.feature_reduction > 1:
                        raise NotImplementedError('feature_reduction (experimental in the reference) is not built')
                    x = self.dropout(x, rate=self.dropout_rate_features)
                    x = self.fc_layer(x, self.num_classes, weight_initializer=self.fc_initializer)
                    d['logits'] = x
                    d['pred'] = self.softmax(x)
        return d

    def _mb_conv_unit(self, x, kernel, stride, out_channels, multiplier, d, drop_rate=0.0, name='mbconv'):
        """reference models/efficientnet.py:126-177."""
        in_channels = x.shape[-1]
        stride = list(stride) if isinstance(stride, (list, tuple)) else [stride, stride]
        if len(stride) == 1:
            stride = [stride[0], stride[0]]
        mid = in_channels * multiplier
        with self.variable_scope(name):
            skip = x if (stride[0] == 1 and stride[1] == 1 and in_channels == out_channels) else None
            d[name + '/branch'] = skip
            with self.variable_scope('conv_0'):
                if multiplier > 1:
                    x = self._conv_norm_swish(x, 1, 1, mid, d, name + '/conv_0')
            with self.variable_scope('conv_1'):
                x = self._conv_norm_swish(x, kernel, stride, mid, d, name + '/conv_1', depthwise=True)
            se_mask = self._se_mask(x, multiplier * self.se_reduction, name='se_mask')
            d[name + '/se_mask'] = se_mask
            x = self.channel_scale(x, se_mask)
            with self.variable_scope('conv_2'):
                x = self._conv_norm_swish(x, 1, 1, out_channels, d, name + '/conv_2', act=False, zero_scale_init=skip is not None)
            if skip is not None:
                x = self.stochastic_depth(x, skip, drop_rate=drop_rate)
            d[name] = x
        return x

    def _se_mask(self, x, reduction, name='se_mask'):
        """reference models/efficientnet.py:179-197: global mean -> biased 1x1 conv -> swish -> biased 1x1 conv -> sigmoid."""
        in_channels = x.shape[-1]
        with self.variable_scope(name):
            x = self.global_avg_pool(x, keepdims=True)
            with self.variable_scope('conv_0'):
                x = self.conv_layer(x, 1, 1, in_channels // reduction, weight_initializer=self.conv_initializer)
            x = self.swish(x, name='swish')
            with self.variable_scope('conv_1'):
                x = self.conv_layer(x, 1, 1, in_channels, weight_initializer=self.conv_initializer)
            x = self.sigmoid(x)
        return x


class EfficientNetB0(EfficientNet):   # 224
    pass


class EfficientNetB1(EfficientNet):   # 240
    width_coefficient, depth_coefficient = 1.0, 1.1


class EfficientNetB2(EfficientNet):   # 260
    width_coefficient, depth_coefficient = 1.1, 1.2


class EfficientNetB3(EfficientNet):   # 300
    width_coefficient, depth_coefficient = 1.2, 1.4


class EfficientNetB4(EfficientNet):   # 380
    width_coefficient, depth_coefficient = 1.4, 1.8


class EfficientNetB5(EfficientNet):   # 456
    width_coefficient, depth_coefficient = 1.6, 2.2


class EfficientNetB6(EfficientNet):   # 528
    width_coefficient, depth_coefficient = 1.8, 2.6


class EfficientNetB7(EfficientNet):   # 600
    width_coefficient, depth_coefficient = 2.0, 3.1
