"""myconvnet_amd — MI355X-native (gfx950) implementation of MyConvNet's conv / batch-norm / ReLU / pooling
training hot path behind the reference's own Python surface.  See DESIGN.md and include/mcn.h."""
from . import _ffi  # noqa: F401  (raises if libmcn_hip.so is missing: there is no CPU fallback)
from .convnet import ConvNet, he_normal, ones, variance_scaling, zeros  # noqa: F401
from .dataset import DataSet, synthetic  # noqa: F401
from .efficientnet import (EfficientNet, EfficientNetB0, EfficientNetB1, EfficientNetB2, EfficientNetB3, EfficientNetB4,  # noqa: F401
                           EfficientNetB5, EfficientNetB6, EfficientNetB7)
from .deeplabv3plus import DeepLabV3PlusResNet, DeepLabV3PlusResNet50  # noqa: F401
from .evaluators import AccuracyEvaluator  # noqa: F401
from .optimizers import MomentumOptimizer, Optimizer  # noqa: F401
from .resnet_v1_5 import ResNet18, ResNet34, ResNet50, ResNet101  # noqa: F401
from .resnet_v1_5_dilated import ResNet50OS8, ResNet50OS16, ResNet101OS8, ResNet101OS16, ResNetDilated  # noqa: F401
from .segnet import SegNet  # noqa: F401
from .vggnet import VGG16, VGG19  # noqa: F401

__version__ = '0.1.0'
