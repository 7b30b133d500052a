"""Architecture tables for the Picklebot video models (pure data, no torch).

Every row restates a layer the reference builds literally; the citation gives the
reference file:line the numbers were read from.  Both the HIP-backed modules
(``picklebot_amd.mobilenet`` ...) and the CPU oracle (``oracle/torch_ref.py``) are
built from these tables, so the state_dict key names/shapes line up with the
reference's by construction; the golden vectors under
``tests/golden`` (generated from the real reference) pin the tables themselves.
"""
from collections import namedtuple

# One Bottleneck3D row: reference mobilenet.py:47-93 constructor arguments.
#   cin, cout, exp : in / out / expanded channels
#   stride         : scalar stride (applies to T, H and W - mobilenet.py:71)
#   se             : squeeze-excite on the expanded tensor
#   k              : depthwise kernel is (1, k, k), padding k//2 on T, H and W
#   act            : 'relu' | 'leaky_relu' | 'hardswish'
#   drop           : Dropout3d probability
BNeck = namedtuple("BNeck", "cin cout exp stride se k act drop")

# mobilenet.py:147-175 (MobileNetLarge3D block2..block5)
MOBILENET_LARGE_3D = {
    "stem_out": 16,                       # mobilenet.py:141  Conv3d(3,16,k3,s2,p1)+bias
    "blocks": {
        "block2": [BNeck(16, 16, 16, 1, False, 3, "relu", 0.2),
                   BNeck(16, 24, 64, 2, False, 3, "relu", 0.2),
                   BNeck(24, 24, 72, 1, False, 3, "relu", 0.2)],
        "block3": [BNeck(24, 40, 72, 2, True, 5, "relu", 0.2),
                   BNeck(40, 40, 120, 1, True, 5, "relu", 0.2),
                   BNeck(40, 40, 120, 1, True, 5, "relu", 0.2)],
        "block4": [BNeck(40, 80, 240, 2, False, 3, "hardswish", 0.2),
                   BNeck(80, 80, 240, 1, False, 3, "hardswish", 0.2),
                   BNeck(80, 80, 184, 1, False, 3, "hardswish", 0.2),
                   BNeck(80, 80, 184, 1, False, 3, "hardswish", 0.2),
                   BNeck(80, 112, 480, 1, True, 3, "hardswish", 0.2),
                   BNeck(112, 112, 672, 1, True, 3, "hardswish", 0.2)],
        "block5": [BNeck(112, 160, 672, 2, True, 5, "hardswish", 0.2),
                   BNeck(160, 160, 960, 1, True, 5, "hardswish", 0.2),
                   BNeck(160, 160, 960, 1, True, 5, "hardswish", 0.2)],
    },
    "last_conv": (160, 960),              # mobilenet.py:179  block6 conv 1x1x1 + bias
    "last_se": False,
    "last_name": "block6",
    "head_hidden": 1280,                  # mobilenet.py:187
}

# mobilenet.py:227-258 (MobileNetSmall3D)
MOBILENET_SMALL_3D = {
    "stem_out": 16,                       # mobilenet.py:221
    "blocks": {
        "block2": [BNeck(16, 16, 16, 2, True, 3, "leaky_relu", 0.2),
                   BNeck(16, 24, 72, 2, False, 3, "leaky_relu", 0.2),
                   BNeck(24, 24, 88, 1, False, 3, "leaky_relu", 0.2)],
        "block3": [BNeck(24, 40, 96, 2, True, 5, "hardswish", 0.2),
                   BNeck(40, 40, 240, 1, True, 5, "hardswish", 0.2),
                   BNeck(40, 40, 240, 1, True, 5, "hardswish", 0.2),
                   BNeck(40, 48, 120, 1, True, 5, "hardswish", 0.2),
                   BNeck(48, 48, 144, 1, True, 5, "hardswish", 0.2),
                   BNeck(48, 96, 288, 2, True, 5, "hardswish", 0.2),
                   BNeck(96, 96, 576, 1, True, 5, "hardswish", 0.2),
                   BNeck(96, 96, 576, 1, True, 5, "hardswish", 0.2)],
    },
    "last_conv": (96, 576),               # mobilenet.py:245  block4 conv 1x1x1 + bias
    "last_se": True,                      # mobilenet.py:246  SEBlock3D(576) before the BN
    "last_name": "block4",
    "head_hidden": 1024,                  # mobilenet.py:253
}

# One MoviNetBottleneck row: movinet.py:43-77 / table movinet.py:98-137.
#   kernel (kt,kh,kw), stride (st,sh,sw), padding (pt,ph,pw); SE always on (movinet.py:44).
MVBlock = namedtuple("MVBlock", "cin cout exp kernel stride padding")


def _mv(cin, cout, exp, kernel, spatial_stride=1):
    kt, kh, kw = kernel
    return MVBlock(cin, cout, exp, kernel, (1, spatial_stride, spatial_stride),
                   (kt // 2, kh // 2, kw // 2))


# movinet.py:98-137 (explicit paddings there equal (kt//2, kh//2, kw//2) on every row)
MOVINET_A2 = {
    "stem_out": 16,                       # movinet.py:92  Conv3d(3,16,(1,3,3),s(1,2,2),p(0,1,1)) no bias
    "blocks": {
        "block2": [_mv(16, 16, 40, (1, 5, 5), 2), _mv(16, 16, 40, (3, 3, 3)), _mv(16, 16, 64, (3, 3, 3))],
        "block3": [_mv(16, 40, 96, (3, 3, 3), 2), _mv(40, 40, 120, (3, 3, 3)), _mv(40, 40, 96, (3, 3, 3)),
                   _mv(40, 40, 96, (3, 3, 3)), _mv(40, 40, 120, (3, 3, 3))],
        "block4": [_mv(40, 72, 240, (5, 3, 3), 2), _mv(72, 72, 160, (3, 3, 3)), _mv(72, 72, 240, (3, 3, 3)),
                   _mv(72, 72, 192, (3, 3, 3)), _mv(72, 72, 240, (3, 3, 3))],
        "block5": [_mv(72, 72, 240, (5, 3, 3)), _mv(72, 72, 240, (3, 3, 3)), _mv(72, 72, 240, (3, 3, 3)),
                   _mv(72, 72, 240, (3, 3, 3)), _mv(72, 72, 144, (1, 5, 5)), _mv(72, 72, 240, (3, 3, 3))],
        "block6": [_mv(72, 144, 480, (5, 3, 3), 2), _mv(144, 144, 384, (1, 5, 5)), _mv(144, 144, 384, (1, 5, 5)),
                   _mv(144, 144, 480, (1, 5, 5)), _mv(144, 144, 480, (1, 5, 5)), _mv(144, 144, 480, (3, 3, 3)),
                   _mv(144, 144, 576, (1, 3, 3))],
    },
    "last_conv": (144, 640),              # movinet.py:140
    "head_hidden": 2048,                  # movinet.py:149
}

def movinet_a2_depthwise_layers(h=172, w=172):
    """(expanded channels, kernel, stride, (H_in, W_in)) of the 26 depthwise convs of MoViNetA2 at an h x w clip
    (movinet.py:92, 98-137): the stem halves H and W, then every bottleneck's depthwise conv strides them by its row."""
    hh, ww = conv_out(h, 3, 2, 1), conv_out(w, 3, 2, 1)
    out = []
    for rows in MOVINET_A2["blocks"].values():
        for r in rows:
            out.append((r.exp, tuple(r.kernel), tuple(r.stride), (hh, ww)))
            hh, ww = conv_out(hh, r.kernel[1], r.stride[1], r.padding[1]), conv_out(ww, r.kernel[2], r.stride[2], r.padding[2])
    return out


# train.py:156-161 registry names
MODEL_NAMES = ("MoViNetA2", "MobileNetLarge3D", "MobileNetSmall3D", "MobileViT")


def conv_out(size, k, stride, pad, dilation=1):
    """Output length of a torch Conv along one axis."""
    return (size + 2 * pad - dilation * (k - 1) - 1) // stride + 1


def bottleneck3d_out_shape(t, h, w, k, stride):
    """(T,H,W) after Bottleneck3D's depthwise conv: kernel (1,k,k), scalar stride and
    scalar padding k//2 on all three axes (mobilenet.py:67-75): T GROWS at stride 1."""
    p = k // 2
    return conv_out(t, 1, stride, p), conv_out(h, k, stride, p), conv_out(w, k, stride, p)
