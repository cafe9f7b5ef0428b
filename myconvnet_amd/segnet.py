"""SegNet: the segmentation flavour of ConvNet, reference segmentation/segnet.py:11-107.

Per-pixel labels [N, H, W] (0 = ignore, k = class k-1; NaN -> 0), one-hot of depth num_classes with all-zero rows for
ignored pixels, backbone (`_build_model` with backbone_only) + segmentation head (`_build_model_seg`), per-pixel softmax
cross-entropy averaged over ALL pixels (ignored ones contribute 0), L2 as in ConvNet; label smoothing = the 5x5 SAME average of
the one-hot label map (segnet.py:117-122).  Augmentation, cutmix and the debug colour images are outside the built path."""
from abc import abstractmethod

from .convnet import ConvNet


class SegNet(ConvNet):
    def _init_model(self, **kwargs):
        B = self.device_batch
        H, W, C = self._input_size
        g = self.graph
        chunk = 4 if self._dtype == 'float32' else 8
        self.X = g.tensor((B, H, W, C), self._dtype, 'X', self._channel_first)
        self.X.cs = (C + chunk - 1) // chunk * chunk
        g.node('input', [], [self.X], image_mean=self.image_mean, scale_factor=self.scale_factor, src_nchw=self._channel_first)
        if kwargs.get('zero_pad_ratio', 0.0) > 0.0 or kwargs.get('cutmix', False):
            raise NotImplementedError('augmentation / zero padding / cutmix are outside the built path')
        self._curr_block = None
        self._backbone_only = True                                   # segnet.py:64-67
        d_backbone = self._build_model()
        self._backbone_only = False
        self.d = self._build_model_seg(d_backbone)
        logits = self.d['logits']
        assert tuple(logits.shape) == (B, H, W, self.num_classes), 'segmentation logits must be [N, H, W, classes]: {}'.format(logits.shape)
        if logits.dtype != 'float32':                                # segnet.py:68-71
            logits32 = g.tensor(logits.shape, 'float32', 'logits_fp32')
            g.node('cast', [logits], [logits32])
            logits = logits32
            self.d['logits'] = logits
        for k, v in d_backbone.items():
            self.d.setdefault(k, v)
        self.logits = logits
        self._build_loss(**kwargs)

    @abstractmethod
    def _build_model_seg(self, d_backbone):
        """Must return a dict of tensors including 'logits' [N, H, W, classes] (segnet.py:99-106)."""

    def _label_smoothing(self, labels, ls_factor, name='label_smoothing'):
        """reference segnet.py:117-122: labels <- (1 - f) * labels + f * avg_pool2d(labels, 5x5, stride 1, SAME).  The average is one
        mcn_avgpool_fwd over the fp32 one-hot map (SAME: divided by the number of in-image cells); the mix itself happens inside the loss
        kernel (mcn_softmax_xent_rows_soft_fwd_bwd), so the factor and the raw map ride on the averaged tensor."""
        with self.variable_scope(name):
            avg = self.avg_pool(labels, (5, 5), (1, 1), padding='SAME')
        avg.ls_factor = float(ls_factor)
        avg.soft_avg_of = labels
        return avg
