"""DeepLabv3+ on a dilated ResNet backbone (SURVEY §8f-3 / BASELINE configs[4]), reference models/deeplabv3plus.py:10-120:
ASPP (1x1 + three dilated 3x3 branches, each conv + norm, concat, 1x1 + norm), decoder (block_1 features -> 1x1 -> 48
channels + norm, bilinear upsample of the ASPP output with align_corners, concat, 3x3 + norm), biased 1x1 classifier and
a final align_corners upsample to the input size.  Scope names as in the reference ('block_5/aspp/conv_k/norm/gamma',
'block_6/features/...', 'block_6/decoder/conv_0/...', 'block_None/logits/...').  tf.concat -> self.concat, tf.nn.dropout ->
self.dropout, tf.stop_gradient -> self.stop_gradient."""
from .resnet_v1_5_dilated import ResNet50OS16, ResNet101OS16
from .segnet import SegNet


class _DeepLabV3Plus(SegNet):
    def _init_deeplab_params(self):
        self.feature_blocks = [4, 1]
        self.feature_channels = [256, 48]
        # (reference: fixed attributes [None, True] / False — the kwargs exist so that the reference's other branch of each switch,
        # tf.stop_gradient on the low-level feature and the ASPP image-level feature, can be exercised: models/deeplabv3plus.py:50-53, 90-99)
        self.feature_gradients = list(self._parameters.get('feature_gradients', [None, True]))
        self.drop_rate_multipliers = [1.0, 0.0]
        self.conv_kernels = [None, 3]
        self.aspp_dilations = list(self._parameters.get('aspp_dilations', [6, 12, 18]))   # kwarg: test-only (small maps); reference: fixed
        self.aspp_level_feature = bool(self._parameters.get('aspp_level_feature', False))
        wd = getattr(self, '_width_div', 1)                        # test-only reduction of the head widths (not in the reference)
        if wd > 1:
            self.feature_channels = [max(8, c // wd) for c in self.feature_channels]

    def _build_model_seg(self, d_backbone):
        d = dict()
        blocks, feature_channels, gradients = self.feature_blocks, self.feature_channels, self.feature_gradients
        self._num_decoder_blocks = min(len(blocks), len(feature_channels), len(gradients), len(self.drop_rate_multipliers), len(self.conv_kernels))
        self._curr_block += 1
        feat = d_backbone['block_{}'.format(blocks[0])]
        with self.variable_scope('block_{}'.format(self._curr_block)):
            feat = self.aspp_unit(feat, feature_channels[0], self.aspp_dilations, level_feature=self.aspp_level_feature)
            x = feat
        d['block_{}'.format(self._curr_block)] = x
        for i in range(1, self._num_decoder_blocks):
            self._curr_block += 1
            feat = d_backbone['block_{}'.format(blocks[i])]
            if not gradients[i]:
                feat = self.stop_gradient(feat)
            with self.variable_scope('block_{}'.format(self._curr_block)):
                with self.variable_scope('features'):
                    feat = self.dropout(feat, rate=self.dropout_rate_features * self.drop_rate_multipliers[i])
                    feat = self.conv_layer(feat, 1, 1, feature_channels[i], biased=False)
                    feat = self.normalization(feat, norm_type=self.norm_type, norm_param=self.norm_param)
                x = self.upsampling_2d_layer(x, out_shape=feat.shape[1:3], align_corners=True)
            x = self.decoder_unit(x, feat, self.conv_kernels[i], d, name='block_{}/decoder'.format(self._curr_block))
            d['block_{}'.format(self._curr_block)] = x
        self._curr_block = None
        with self.variable_scope('block_{}'.format(self._curr_block)):
            with self.variable_scope('logits'):
                x = self.conv_layer(x, 1, 1, self.num_classes)
                x = self.upsampling_2d_layer(x, out_shape=self.input_size[0:2], align_corners=True)
                d['logits'] = x
                d['pred'] = self.softmax(x)
        return d

    def aspp_unit(self, x, channels, dilations, level_feature=False, name='aspp'):
        with self.variable_scope(name):
            ys = []
            with self.variable_scope('conv_0'):
                y = self.conv_layer(x, 1, 1, channels, padding='SAME', biased=False, depthwise=False)
                ys.append(self.normalization(y, norm_type=self.norm_type, norm_param=self.norm_param))
            for i, dil in enumerate(dilations):
                with self.variable_scope('conv_{}'.format(i + 1)):
                    y = self.conv_layer(x, 3, 1, channels, padding='SAME', biased=False, depthwise=False, dilation=dil)
                    ys.append(self.normalization(y, norm_type=self.norm_type, norm_param=self.norm_param))
            if level_feature:
                # image-level feature (models/deeplabv3plus.py:90-99): global mean -> 1x1 conv -> norm -> resized back to the feature map (from a
                # 1x1 source: a broadcast; the reference's call leaves align_corners at its default False)
                with self.variable_scope('conv_pool'):
                    y = self.global_avg_pool(x, keepdims=True)
                    y = self.conv_layer(y, 1, 1, channels, padding='SAME', biased=False, depthwise=False, dilation=dilations[-1])
                    y = self.normalization(y, norm_type=self.norm_type, norm_param=self.norm_param)
                    ys.append(self.upsampling_2d_layer(y, out_shape=x.shape[1:3]))
            with self.variable_scope('conv_out'):
                x = self.concat(ys)
                x = self.conv_layer(x, 1, 1, channels, padding='SAME', biased=False, depthwise=False)
                x = self.normalization(x, norm_type=self.norm_type, norm_param=self.norm_param)
        return x

    def decoder_unit(self, x, feature, kernel, d, name='decoder'):
        with self.variable_scope(name):
            out_channels = x.shape[-1]
            x = self.concat([x, feature])
            with self.variable_scope('conv_0'):
                x = self.conv_layer(x, kernel, 1, out_channels, padding='SAME', biased=False, depthwise=False)
                x = self.normalization(x, norm_type=self.norm_type, norm_param=self.norm_param)
                d[name + '/conv_0'] = x
            d[name] = x
        return x


class DeepLabV3PlusResNet(_DeepLabV3Plus, ResNet101OS16):
    """The class the reference defines (ResNet-101, output stride 16)."""

    def _init_params(self, **kwargs):
        ResNet101OS16._init_params(self, **kwargs)
        self._init_deeplab_params()

    def _build_model(self):
        return ResNet101OS16._build_model(self)


class DeepLabV3PlusResNet50(_DeepLabV3Plus, ResNet50OS16):
    """BASELINE configs[4]: the same head on the ResNet-50 dilated backbone (models/resnet_v1_5_dilated.py:145)."""

    def _init_params(self, **kwargs):
        ResNet50OS16._init_params(self, **kwargs)
        self._init_deeplab_params()

    def _build_model(self):
        return ResNet50OS16._build_model(self)
