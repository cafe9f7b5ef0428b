"""VGG-16/19 on the MI355X building blocks (reference models/vggnet.py:11-143): biased 3x3 conv + ReLU stacks,
2x2/2 SAME max-pools, conv-as-FC head that needs a 224x224 input.  BASELINE config #1 runs the trunk
(backbone_only=True) at 8x8, see SURVEY.md §8f-0."""
from .convnet import ConvNet

VGG_MEAN = [123.68, 116.78, 103.94]  # RGB mean, reference models/vggnet.py:8


class VGGNet(ConvNet):
    def _init_params(self, **kwargs):
        self.num_layers = 16
        self.width_div = kwargs.get('width_div', 1)   # test-only channel reduction

    def _stage_plan(self):
        reps = 3 if self.num_layers == 16 else 4
        plan = [[64] * 2, [128] * 2, [256] * reps, [512] * reps, [512] * reps]
        return [[c // self.width_div for c in blk] for blk in plan]

    def _build_model(self):
        assert self.num_layers in (16, 19), 'Number of layers must be either 16 or 19.'
        d = dict()
        # reference vggnet.py:23-25: x = (X/scale_factor + image_mean)*255 - VGG_MEAN.  With the same input this is an
        # affine map per channel of the raw image, which the input-prep kernel applies directly.
        x = self.vgg_rescale(self.X)
        names = {0: 'conv1', 1: 'conv2', 2: 'conv3', 3: 'conv4', 4: 'conv5'}
        for b, blk in enumerate(self._stage_plan()):
            self._curr_block = b
            with self.variable_scope('block_{}'.format(b)):
                for j, c in enumerate(blk):
                    x = self.conv_relu(x, c, name='conv_{}'.format(j))
                    d['{}_{}'.format(names[b], j + 1)] = x
                x = self.max_pool(x, 2, 2)
            d['block_{}'.format(b)] = x
        if not self.backbone_only:
            self._curr_block = None
            with self.variable_scope('block_None'):
                assert self.input_size[0] == 224 and self.input_size[1] == 224, 'Input shape must be (224, 224, 3)'
                x = self.relu(self.conv_layer(x, 7, stride=1, out_channels=4096 // self.width_div, padding='VALID', biased=True, scope='fc_0'))
                d['fc6'] = x
                x = self.relu(self.conv_layer(x, 1, stride=1, out_channels=4096 // self.width_div, padding='SAME', biased=True, scope='fc_1'))
                d['fc7'] = x
                x = self.conv_layer(x, 1, stride=1, out_channels=self.num_classes, padding='SAME', biased=True, scope='fc_2')
                d['fc8'] = x
                x = self.flatten(x)
                d['logits'] = x
                d['pred'] = self.softmax(x)
        return d

    def conv_relu(self, x, channels, name, verbose=True):
        return self.relu(self.conv_layer(x, 3, stride=1, out_channels=channels, padding='SAME', biased=True, scope=name))

    def vgg_rescale(self, x):
        """reference vggnet.py:23-25: (X/scale_factor + image_mean)*255 - VGG_MEAN as one per-channel affine pass
        (scale 255/scale_factor, shift image_mean*255 - VGG_MEAN[c]; padded channels stay zero)."""
        y = self.graph.tensor(x.shape, x.dtype, 'vgg_input', self.channel_first)
        y.cs = x.cs
        c = x.shape[-1]
        scale = [255.0 / self.scale_factor] * c + [1.0] * (x.cs - c)
        shift = [self.image_mean * 255.0 - VGG_MEAN[i % 3] for i in range(c)] + [0.0] * (x.cs - c)
        self.graph.node('affine', [x], [y], scale=scale, shift=shift)
        return y

    def flatten(self, x):
        """tf.reshape(x, [-1, num_classes]) (vggnet.py:123) as a copy of the [N,1,1,C] tensor."""
        n, h, w, c = x.shape
        assert h == 1 and w == 1
        y = self.graph.tensor((n, c), x.dtype, 'flatten')
        self.graph.node('cast', [x], [y])
        return y


class VGG16(VGGNet):
    def _init_params(self, **kwargs):
        super()._init_params(**kwargs)
        self.num_layers = 16


class VGG19(VGGNet):
    def _init_params(self, **kwargs):
        super()._init_params(**kwargs)
        self.num_layers = 19
