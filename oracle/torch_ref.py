"""CPU ORACLE (test infrastructure, never shipped, never measured as the product).

A from-scratch ``torch.nn`` restatement of the reference's video models - same class
names, constructor signatures, attribute names (=> identical ``state_dict`` keys) and
quirks - built from the tables in ``picklebot_amd/arch.py``.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Parity pin: ``tests/test_oracle_golden.py`` checks every class here against golden
vectors produced by importing the real reference in the build container
(``oracle/make_golden.py``; fixtures in ``tests/golden/*.npz``).  The reference has no
tests/fixtures of its own, so those vectors are the pin.

Reference citations (``/root/reference``):
  SEBlock3D            mobilenet.py:11-26
  Bottleneck3D         mobilenet.py:47-93
  MobileNetLarge3D     mobilenet.py:133-210
  MobileNetSmall3D     mobilenet.py:213-278
  CausalConv3d         movinet.py:7-39
  MoviNetBottleneck    movinet.py:43-77
  MoViNetA2            movinet.py:80-179
  conv_1x1_bn/nxn_bn   mobilevit.py:12-24
  FeedForward          mobilevit.py:28-40
  Attention            mobilevit.py:43-86   (head merge restored, see class docstring)
  Transformer          mobilevit.py:89-104
  MobileViTBlock       mobilevit.py:107-143
  MobileViT            mobilevit.py:146-215
"""
import math
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from picklebot_amd import arch  # noqa: E402  (pure data tables)

_ACTS = {"relu": nn.ReLU, "leaky_relu": nn.LeakyReLU, "hardswish": nn.Hardswish}


class SEBlock3D(nn.Module):
    """x * hardsigmoid(W2 relu(W1 mean_THW(x) + b1) + b2)  (mobilenet.py:11-26)."""

    def __init__(self, channels):
        super().__init__()
        mid = channels // 4
        # indices 1 and 3 carry parameters -> keys se.1.*, se.3.*
        self.se = nn.Sequential(nn.AdaptiveAvgPool3d(1), nn.Conv3d(channels, mid, 1), nn.ReLU(),
                                nn.Conv3d(mid, channels, 1), nn.Hardsigmoid())

    def forward(self, x):
        return x * self.se(x)


class Bottleneck3D(nn.Module):
    """pw1 -> depthwise (1,k,k) with scalar stride/padding on T,H,W -> [SE] -> pw2 -> BN
    -> act -> Dropout3d; no residual, nothing between the three convs (mobilenet.py:84-93)."""

    def __init__(self, in_channels, out_channels, expanded_channels, stride=1, use_se=False,
                 kernel_size=3, nonlinearity=None, batchnorm=True, dropout=0, bias=False):
        super().__init__()
        e = expanded_channels
        self.pointwise_conv1 = nn.Conv3d(in_channels, e, 1, bias=bias)
        self.depthwise_conv = nn.Conv3d(e, e, (1, kernel_size, kernel_size), stride=stride,
                                        padding=kernel_size // 2, groups=e, bias=bias)
        self.squeeze_excite = SEBlock3D(e) if use_se else None
        self.pointwise_conv2 = nn.Conv3d(e, out_channels, 1, bias=bias)
        self.batchnorm = nn.BatchNorm3d(out_channels) if batchnorm else None
        self.nonlinearity = nn.Hardswish() if nonlinearity is None else nonlinearity
        self.dropout = nn.Dropout3d(p=dropout)

    def forward(self, x):
        x = self.depthwise_conv(self.pointwise_conv1(x))
        if self.squeeze_excite is not None:
            x = self.squeeze_excite(x)
        x = self.batchnorm(self.pointwise_conv2(x))
        return self.dropout(self.nonlinearity(x))


def _bneck_from_row(r):
    return Bottleneck3D(r.cin, r.cout, r.exp, stride=r.stride, use_se=r.se, kernel_size=r.k,
                        nonlinearity=_ACTS[r.act](), dropout=r.drop)


class _MobileNet3D(nn.Module):
    TABLE = None

    def __init__(self, num_classes=2):
        super().__init__()
        tb = self.TABLE
        self.num_classes = num_classes
        self.block1 = nn.Sequential(nn.Conv3d(3, tb["stem_out"], 3, stride=2, padding=1),
                                    nn.BatchNorm3d(tb["stem_out"]), nn.Hardswish())
        self._order = ["block1"]
        for name, rows in tb["blocks"].items():
            setattr(self, name, nn.Sequential(*[_bneck_from_row(r) for r in rows]))
            self._order.append(name)
        cin, cout = tb["last_conv"]
        tail = [nn.Conv3d(cin, cout, 1)]
        if tb["last_se"]:
            tail.append(SEBlock3D(cout))
        tail += [nn.BatchNorm3d(cout), nn.Hardswish()]
        setattr(self, tb["last_name"], nn.Sequential(*tail))
        self._order.append(tb["last_name"])
        self.classifier = nn.Sequential(nn.AdaptiveAvgPool3d((1, 1, 1)), nn.Conv3d(cout, tb["head_hidden"], 1),
                                        nn.Hardswish(), nn.Conv3d(tb["head_hidden"], num_classes, 1))

    def forward(self, x):
        for name in self._order:
            x = getattr(self, name)(x)
        return self.classifier(x).view(x.shape[0], self.num_classes)


class MobileNetLarge3D(_MobileNet3D):
    TABLE = arch.MOBILENET_LARGE_3D

    def initialize_weights(self):
        # mobilenet.py:203-210: the hasattr(module,"nonlinearity") test is never true for a
        # Conv3d/Linear, so no tensor changes.
        return None


class MobileNetSmall3D(_MobileNet3D):
    TABLE = arch.MOBILENET_SMALL_3D

    def initialize_weights(self):
        # mobilenet.py:268-278: only the BatchNorm3d branch has an effect (gamma=1, beta=0).
        for m in self.modules():
            if isinstance(m, nn.BatchNorm3d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)


# ----------------------------------------------------------------------------- MoViNet
class CausalConv3d(nn.Module):
    """Left-pad T by (k_t - 1) with `stream_buffer` (a scalar) then a plain Conv3d
    (movinet.py:7-39).  Not instantiated by any reference model."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1, stream_buffer=None, **kwargs):
        super().__init__()
        ks = kernel_size if isinstance(kernel_size, tuple) else (kernel_size,) * 3
        dl = dilation if isinstance(dilation, tuple) else (dilation,) * 3
        self.stream_buffer = 0 if stream_buffer is None else stream_buffer
        kt = ks[0]
        left, right = ((kt - 2) // 2, kt // 2) if kt % 2 == 0 else ((kt - 1) // 2, (kt - 1) // 2)
        self.p_left_causal, self.p_right_causal = left + right, 0
        self.conv3d = nn.Conv3d(in_channels, out_channels, ks, stride=stride, dilation=dl, **kwargs)

    def forward(self, x):
        x = F.pad(x, (0, 0, 0, 0, self.p_left_causal, 0), "constant", self.stream_buffer)
        return self.conv3d(x)


class MoviNetBottleneck(nn.Module):
    """expand -> depthwise (kt,k,k) -> SE -> project -> BN -> act; `dropout` exists but is
    not applied in forward (movinet.py:69-77)."""

    def __init__(self, in_channels, out_channels, expanded_channels, kernel_size, stride=1, use_se=True,
                 batchnorm=True, nonlinearity=None, bias=False, dropout=0, padding=None, dilation=1):
        super().__init__()
        e = expanded_channels
        self.expand = nn.Conv3d(in_channels, e, 1, bias=bias)
        if padding is None:  # movinet.py:49
            padding = ((kernel_size[0] - 1, kernel_size[1] // 2, kernel_size[2] // 2)
                       if isinstance(kernel_size, tuple) else kernel_size // 2)
        self.conv = nn.Conv3d(e, e, kernel_size, stride=stride, padding=padding, groups=e, bias=bias,
                              dilation=dilation)
        self.squeeze_excite = SEBlock3D(e) if use_se else None
        self.project = nn.Conv3d(e, out_channels, 1, bias=bias)
        self.batchnorm = nn.BatchNorm3d(out_channels) if batchnorm else None
        self.nonlinearity = nn.Hardswish() if nonlinearity is None else nonlinearity
        self.dropout = nn.Dropout3d(p=dropout)

    def forward(self, x):
        x = self.conv(self.expand(x))
        if self.squeeze_excite is not None:
            x = self.squeeze_excite(x)
        return self.nonlinearity(self.batchnorm(self.project(x)))


class MoViNetA2(nn.Module):
    def __init__(self, num_classes=2, buffer_size=2):
        super().__init__()
        tb = arch.MOVINET_A2
        self.num_classes, self.buffer_size = num_classes, buffer_size
        self.block1 = nn.Sequential(nn.Conv3d(3, tb["stem_out"], (1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1),
                                              bias=False), nn.BatchNorm3d(tb["stem_out"]), nn.Hardswish())
        self._order = ["block1"]
        for name, rows in tb["blocks"].items():
            setattr(self, name, nn.Sequential(*[
                MoviNetBottleneck(r.cin, r.cout, r.exp, kernel_size=r.kernel, stride=r.stride, padding=r.padding)
                for r in rows]))
            self._order.append(name)
        cin, cout = tb["last_conv"]
        self.conv = nn.Sequential(nn.Conv3d(cin, cout, 1, bias=False), nn.BatchNorm3d(cout), nn.Hardswish(),
                                  nn.Dropout3d(0.2))
        hid = tb["head_hidden"]
        self.classifier = nn.Sequential(nn.AdaptiveAvgPool3d((1, 1, 1)), nn.Flatten(), nn.Linear(cout, hid),
                                        nn.BatchNorm1d(hid), nn.Hardswish(), nn.Dropout(0.2),
                                        nn.Linear(hid, num_classes))

    def forward(self, x):
        for name in self._order:
            x = getattr(self, name)(x)
        return self.classifier(self.conv(x))

    def initialize_weights(self):  # movinet.py:167-179
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm3d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)


# ----------------------------------------------------------------------------- MobileViT
def conv_1x1_bn(in_channels, out_channels):
    return nn.Sequential(nn.Conv3d(in_channels, out_channels, 1, 1, 0, bias=False),
                         nn.BatchNorm3d(out_channels, affine=False), nn.SiLU())


def conv_nxn_bn(in_channels, out_channels, kernel_size=3, stride=1):
    return nn.Sequential(nn.Conv3d(in_channels, out_channels, kernel_size, stride, 1, bias=False),
                         nn.BatchNorm3d(out_channels, affine=False), nn.SiLU())


class FeedForward(nn.Module):
    def __init__(self, embed_dim, hidden_dim, dropout=0.):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(embed_dim, hidden_dim, bias=False), nn.SiLU(), nn.Dropout(dropout),
                                 nn.Linear(hidden_dim, embed_dim, bias=False), nn.Dropout(dropout))

    def forward(self, x):
        return self.net(x)


class Attention(nn.Module):
    """LayerNorm(no affine) -> to_qkv -> per-(b,p,head) softmax(q k^T / sqrt(d)) v -> merge heads
    -> to_out.  The reference at HEAD never merges the heads (mobilevit.py:71-74) and raises in
    to_out; the merge 'b p h n d -> b p n (h d)' that survives as a comment at mobilevit.py:82
    (and in profiling_mobilevit.py:102) is restored here, as SURVEY.md 8c prescribes.  The
    attention core is written as explicit matmul+softmax (equal to SDPA's math path)."""

    def __init__(self, embed_dim, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        self.inner_dim, self.heads, self.scale = dim_head * heads, heads, dim_head ** -0.5
        self.norm = nn.LayerNorm(embed_dim, elementwise_affine=False)
        self.dropout_p = dropout
        self.to_qkv = nn.Linear(embed_dim, self.inner_dim * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(self.inner_dim, embed_dim, bias=False), nn.Dropout(dropout))

    def forward(self, x):
        b, p, n, _ = x.shape
        q, k, v = self.to_qkv(self.norm(x)).chunk(3, dim=-1)
        q, k, v = (t.reshape(b, p, n, self.heads, -1).permute(0, 1, 3, 2, 4) for t in (q, k, v))
        att = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * self.scale, dim=-1)
        att = F.dropout(att, self.dropout_p, self.training)
        out = torch.matmul(att, v).permute(0, 1, 3, 2, 4).reshape(b, p, n, self.inner_dim)
        return self.to_out(out)


class Transformer(nn.Module):
    def __init__(self, embed_dim, depth, heads, dim_head, ffw_dim, dropout=0.):
        super().__init__()
        self.layers = nn.ModuleList([nn.ModuleList([Attention(embed_dim, heads, dim_head, dropout),
                                                    FeedForward(embed_dim, ffw_dim, dropout)])
                                     for _ in range(depth)])

    def forward(self, x):
        for attn, ff in self.layers:
            x = attn(x) + x
            x = ff(x) + x
        return x


def unfold_patches(x, pt, ph, pw):
    """'b d (t pt) (h ph) (w pw) -> b (pt ph pw) (t h w) d' (mobilevit.py:134) as reshape+permute."""
    b, d, T, H, W = x.shape
    t, h, w = T // pt, H // ph, W // pw
    x = x.reshape(b, d, t, pt, h, ph, w, pw).permute(0, 3, 5, 7, 2, 4, 6, 1)
    return x.reshape(b, pt * ph * pw, t * h * w, d)


def fold_patches(x, pt, ph, pw, t, h, w):
    """Inverse of unfold_patches (mobilevit.py:136)."""
    b, _, _, d = x.shape
    x = x.reshape(b, pt, ph, pw, t, h, w, d).permute(0, 7, 4, 1, 5, 2, 6, 3)
    return x.reshape(b, d, t * pt, h * ph, w * pw)


class MobileViTBlock(nn.Module):
    def __init__(self, embed_dim, depth, channel, kernel_size, patch_size, ffw_dim, dropout=0.,
                 use_linear_attention=False):
        super().__init__()
        self.pt, self.ph, self.pw = patch_size
        self.conv1 = conv_nxn_bn(channel, channel, kernel_size)
        self.conv2 = conv_1x1_bn(channel, embed_dim)
        self.transformer = Transformer(embed_dim, depth, 8, 16, ffw_dim, dropout)  # heads=8, d=16 fixed (:116)
        self.conv3 = conv_1x1_bn(embed_dim, channel)
        self.conv4 = conv_nxn_bn(2 * channel, channel, kernel_size)

    def forward(self, x):
        y = x
        x = self.conv2(self.conv1(x))
        _, _, t, h, w = x.shape
        tp, hp, wp = (-t) % self.pt, (-h) % self.ph, (-w) % self.pw
        x = F.pad(x, (0, wp, 0, hp, 0, tp))
        T, H, W = t + tp, h + hp, w + wp
        x = unfold_patches(x, self.pt, self.ph, self.pw)
        x = self.transformer(x)
        x = fold_patches(x, self.pt, self.ph, self.pw, T // self.pt, H // self.ph, W // self.pw)
        x = x[:, :, :t, :h, :w]
        x = self.conv3(x)
        return self.conv4(torch.cat((x, y), 1))


class MobileViT(nn.Module):
    def __init__(self, dims, channels, num_classes, expansion=4, kernel_size=3, patch_size=(2, 2, 2),
                 depths=(2, 4, 3), use_linear_attention=False):
        super().__init__()
        assert len(dims) == 3, 'dims must be a tuple of 3'
        assert len(depths) == 3, 'depths must be a tuple of 3'
        ch = channels
        self.conv1 = conv_nxn_bn(3, ch[0], stride=2)
        # mobilevit.py:167-171: stem.3 repeats channels[2]->channels[3]
        stem_rows = [(ch[0], ch[1], ch[0] * expansion, 1), (ch[1], ch[2], ch[1] * expansion, 2),
                     (ch[2], ch[3], ch[2] * expansion, 1), (ch[2], ch[3], ch[2] * expansion, 1)]
        self.stem = nn.ModuleList([Bottleneck3D(a, b, expanded_channels=e, stride=s) for a, b, e, s in stem_rows])
        # mobilevit.py:173-187: note the expansions channels[7]*expansion and channels[7]*4
        trunk_rows = [(ch[3], ch[4], ch[3] * expansion, dims[0], depths[0], ch[5], int(dims[0] * 2)),
                      (ch[5], ch[6], ch[7] * expansion, dims[1], depths[1], ch[7], int(dims[1] * 4)),
                      (ch[7], ch[8], ch[7] * 4, dims[2], depths[2], ch[9], int(dims[2] * 4))]
        self.trunk = nn.ModuleList([nn.ModuleList([
            Bottleneck3D(a, b, expanded_channels=e, stride=2),
            MobileViTBlock(d, dep, c, kernel_size, patch_size, f, use_linear_attention=use_linear_attention)])
            for a, b, e, d, dep, c, f in trunk_rows])
        self.to_logits = nn.Sequential(conv_1x1_bn(ch[-2], ch[-1]), _MeanTHW(),
                                       nn.Linear(ch[-1], num_classes, bias=False))

    def forward(self, x):
        x = self.conv1(x)
        for blk in self.stem:
            x = blk(x)
        for blk, vit in self.trunk:
            x = vit(blk(x))
        return self.to_logits(x)

    def initialize_weights(self):  # mobilevit.py:207-215
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Conv2d, nn.Conv3d)):
                m.weight.data.normal_(mean=0.0, std=0.02)
                if m.bias is not None:
                    m.bias.data.zero_()


class LinearSelfAttention(nn.Module):
    """mobilevitv2.py:9-55 (the shape prints of :30-38 left out).  Input (batch, embed_dim, pixels-in-patch, patches)."""

    def __init__(self, embed_dim, dropout=0.0):
        super().__init__()
        self.qkv_proj = nn.Conv2d(embed_dim, 1 + 2 * embed_dim, kernel_size=1, bias=True)     # :13-18
        self.attn_dropout = nn.Dropout(p=dropout)
        self.out_proj = nn.Conv2d(embed_dim, embed_dim, kernel_size=1, bias=True)             # :20-25
        self.embed_dim = embed_dim

    def forward(self, hidden_states):
        qkv = self.qkv_proj(hidden_states)
        query, key, value = torch.split(qkv, [1, self.embed_dim, self.embed_dim], dim=1)      # :37
        scores = self.attn_dropout(F.softmax(query, dim=-1))                                  # over the patches, :41-42
        context = (key * scores).sum(dim=-1, keepdim=True)                                    # :46-48
        return self.out_proj(F.relu(value) * context.expand_as(value))                        # :52-53


class _MeanTHW(nn.Module):
    """einops Reduce('b c t h w -> b c', 'mean') (mobilevit.py:191); parameter-free."""

    def forward(self, x):
        return x.mean(dim=(2, 3, 4))


REGISTRY = {"MoViNetA2": MoViNetA2, "MobileNetLarge3D": MobileNetLarge3D,
            "MobileNetSmall3D": MobileNetSmall3D, "MobileViT": MobileViT}
