"""picklebot_amd - MI355X-native (gfx950) forward/backward hot path for Picklebot's video models.

    from picklebot_amd import MobileNetLarge3D, MobileNetSmall3D, MoViNetA2, MobileViT

mirrors the reference's `from mobilenet import ...` / `from movinet import ...` /
`from mobilevit import ...` (train.py:19-21).  Importing the package does not need a GPU; running
a module does (hand-written HIP kernels behind a C ABI, no eager fallback).
"""
from .mobilenet import Bottleneck3D, MobileNetLarge3D, MobileNetSmall3D, SEBlock3D  # noqa: F401
from .movinet import CausalConv3d, MoviNetBottleneck, MoViNetA2  # noqa: F401
from .mobilevit import Attention, FeedForward, MobileViT, MobileViTBlock, Transformer  # noqa: F401
from .mobilevitv2 import LinearSelfAttention  # noqa: F401
from .optim import FlatAdamW  # noqa: F401
from .functional import fp8_attention  # noqa: F401
from .kernels import weight_grad_overlap  # noqa: F401
from .harness import (DevicePrefetcher, calculate_accuracy, calculate_accuracy_bce, custom_collate, estimate_loss,  # noqa: F401
                      extract_features_labels, strip_compile_prefix)

__all__ = ["SEBlock3D", "Bottleneck3D", "MobileNetLarge3D", "MobileNetSmall3D", "CausalConv3d", "MoviNetBottleneck",
           "MoViNetA2", "FeedForward", "Attention", "Transformer", "MobileViTBlock", "MobileViT", "LinearSelfAttention"]
