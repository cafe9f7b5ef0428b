"""Dilated ResNet v1.5 backbones, reference models/resnet_v1_5_dilated.py:5-170: the bottleneck network of
resnet_v1_5.py with a per-stage dilation of the 3x3 conv and the multi-grid factors (1, 2, 4) inside dilated stages; same
scope names.  OS16 / OS8 variants as in the reference (note: ResNet50OS16 keeps the reference's stage table, stride 2 AND
dilation 2 in block 4, :145-146; the DeepLab file itself wires ResNet101OS16, stride 1 there)."""
from .resnet_v1_5 import ResNetBot


class ResNetDilated(ResNetBot):
    def _init_params(self, **kwargs):
        super()._init_params(**kwargs)
        self.channels = [64, 256, 512, 1024, 2048]
        self.kernels = [7, 3, 3, 3, 3]
        self.strides = [2, 1, 2, 2, 2]
        self.res_units = [None, 3, 4, 6, 3]
        self.dilations = [None, 1, 1, 1, 2]
        self.multi_grid = [1, 2, 4]
        dd = int(kwargs.get('depth_div', 1))                 # test-only reduction (not in the reference)
        self._depth_div = dd

    def _apply_width(self):
        super()._apply_width()
        if self._depth_div > 1:
            self.res_units = [None if u is None else max(1, u // self._depth_div) for u in self.res_units]
            self._depth_div = 1

    def _res_unit(self, x, kernel, stride, out_channels, dilation, d, drop_rate=0.0, name='res_unit'):
        """reference models/resnet_v1_5_dilated.py:88-141."""
        in_channels = x.shape[-1]
        stride = list(stride) if isinstance(stride, (list, tuple)) else [stride, stride]
        if len(stride) == 1:
            stride = [stride[0], stride[0]]
        with self.variable_scope(name):
            if in_channels == out_channels:
                skip = self.max_pool(x, stride, stride, padding='VALID') if (stride[0] > 1 or stride[1] > 1) else x
            else:
                with self.variable_scope('conv_skip'):
                    skip = self.conv_layer(x, 1, stride, out_channels, padding='SAME', biased=False)
                    skip = self._norm(skip)
            d[name + '/branch'] = skip
            plan = [('conv_0', 1, 1, out_channels // 4, 1, True, False),
                    ('conv_1', kernel, stride, out_channels // 4, dilation, True, False),
                    ('conv_2', 1, 1, out_channels, 1, False, True)]
            for scope, k, s, c, dil, relu_after, zero_gamma in plan:
                with self.variable_scope(scope):
                    x = self.conv_layer(x, k, s, c, padding='SAME', biased=False, dilation=dil)
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
        num_blocks = min(len(self.channels), len(self.kernels), len(self.strides), len(self.res_units), len(self.dilations))
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
                dil = 1 if self.dilations[i] == 1 else self.dilations[i] * self.multi_grid[j % len(self.multi_grid)]
                x = self._res_unit(x, self.kernels[i], self.strides[i] if j == 0 else 1, self.channels[i], dil, d,
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


class ResNet50OS16(ResNetDilated):
    pass


class ResNet50OS8(ResNetDilated):
    def _init_params(self, **kwargs):
        super()._init_params(**kwargs)
        self.strides = [2, 1, 2, 1, 1]
        self.dilations = [None, 1, 1, 2, 4]


class ResNet101OS16(ResNetDilated):
    def _init_params(self, **kwargs):
        super()._init_params(**kwargs)
        self.strides = [2, 1, 2, 2, 1]
        self.res_units = [None, 3, 4, 23, 3]
        self.dilations = [None, 1, 1, 1, 2]


class ResNet101OS8(ResNetDilated):
    def _init_params(self, **kwargs):
        super()._init_params(**kwargs)
        self.strides = [2, 1, 2, 1, 1]
        self.res_units = [None, 3, 4, 23, 3]
        self.dilations = [None, 1, 1, 2, 4]
