"""Drop-in MI355X modules for the reference's mobilevit.py (conv_1x1_bn/conv_nxn_bn :12-24, FeedForward
:28-40, Attention :43-86, Transformer :89-104, MobileViTBlock :107-143, MobileViT :146-215): same names,
constructor signatures, attribute names and state_dict keys.

Attention: the reference at HEAD never merges the heads before `to_out` and raises (mobilevit.py:71-74);
as SURVEY.md 8c prescribes, the merge 'b p h n d -> b p n (h d)' that survives as a comment at
mobilevit.py:82 is what runs here (inside the attention kernel's output layout).  The three debug
prints of the reference constructor (mobilevit.py:46-48) are not reproduced.
"""
from typing import List, Tuple, Union

import torch
import torch.nn as nn

from . import functional as PF
from .mobilenet import Bottleneck3D, act_of, compute_dtype, from_ndhwc, to_ndhwc


def conv_1x1_bn(in_channels: int, out_channels: int) -> nn.Sequential:
    return nn.Sequential(nn.Conv3d(in_channels, out_channels, 1, 1, 0, bias=False),
                         nn.BatchNorm3d(out_channels, affine=False), nn.SiLU())


def conv_nxn_bn(in_channels: int, out_channels: int, kernel_size: Union[int, Tuple[int, ...]] = 3, stride: int = 1) -> nn.Sequential:
    return nn.Sequential(nn.Conv3d(in_channels, out_channels, kernel_size, stride, 1, bias=False),
                         nn.BatchNorm3d(out_channels, affine=False), nn.SiLU())


def run_conv_1x1_bn(seq, x5):
    conv, bn, act_m = seq
    act, slope = act_of(act_m)
    return PF.pw_bn_act(x5, conv.weight, conv.bias, bn, act=act, slope=slope, training=bn.training)


def run_conv_nxn_bn(seq, x5):
    conv, bn, act_m = seq
    act, slope = act_of(act_m)
    return PF.conv_bn_act(x5, conv.weight, conv.bias, bn, stride=conv.stride, pad_front=conv.padding, act=act,
                          slope=slope, training=bn.training)


def _no_dropout(p, who):
    if p:
        raise NotImplementedError(f"picklebot_amd.{who}: dropout > 0 (the reference builds these blocks with dropout=0)")


class FeedForward(nn.Module):
    def __init__(self, embed_dim: int, hidden_dim: int, dropout: float = 0.):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(embed_dim, hidden_dim, bias=False), nn.SiLU(), nn.Dropout(dropout),
                                 nn.Linear(hidden_dim, embed_dim, bias=False), nn.Dropout(dropout))

    def rows(self, rows, residual=True):
        """x + net(x) on token rows [R, D] (the residual of Transformer.forward is fused in); residual=False: net(x)."""
        _no_dropout(self.net[2].p if self.training else 0, "FeedForward")
        return PF.ff_block(rows, self.net[0].weight, self.net[3].weight, residual)

    @torch.compiler.disable
    @PF.scoped
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        rows = x.reshape(-1, x.shape[-1]).to(compute_dtype(x))
        return self.rows(rows, residual=False).view(x.shape)    # module contract is net(x) without the residual


class Attention(nn.Module):
    def __init__(self, embed_dim: int, heads: int = 8, dim_head: int = 64, dropout: float = 0.):
        super().__init__()
        self.inner_dim = dim_head * heads
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.norm = nn.LayerNorm(embed_dim, elementwise_affine=False)
        self.attend = nn.Softmax(dim=-1)
        self.dropout_p = dropout
        self.dropout = nn.Dropout(dropout)
        self.to_qkv = nn.Linear(embed_dim, self.inner_dim * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(self.inner_dim, embed_dim, bias=False), nn.Dropout(dropout))
        self.flash = True

    def rows(self, rows, G, n, residual=True):
        """x + to_out(attention(...)) on token rows [G*n, D]; residual=False: without the x."""
        if self.heads != 8 or self.inner_dim != 128:
            raise NotImplementedError("picklebot_amd.Attention: kernels are built for 8 heads of 16 (mobilevit.py:116)")
        _no_dropout(self.dropout_p if self.training else 0, "Attention")
        return PF.attn_block(rows, self.to_qkv.weight, self.to_out[0].weight, G, n, self.scale, self.norm.eps, residual)

    @torch.compiler.disable
    @PF.scoped
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        b, p, n, d = x.shape
        rows = x.reshape(-1, d).to(compute_dtype(x))
        return self.rows(rows, b * p, n, residual=False).view(x.shape)


class Transformer(nn.Module):
    def __init__(self, embed_dim: int, depth: int, heads: int, dim_head: int, ffw_dim: int, dropout: float = 0.) -> None:
        super().__init__()
        self.layers = nn.ModuleList([nn.ModuleList([Attention(embed_dim, heads, dim_head, dropout),
                                                    FeedForward(embed_dim, ffw_dim, dropout)]) for _ in range(depth)])

    @torch.compiler.disable
    @PF.scoped
    def forward(self, x):
        b, p, n, d = x.shape
        rows = x.reshape(-1, d)
        if rows.dtype != compute_dtype(x):
            rows = rows.to(compute_dtype(x))
        if PF.transformer_wide_ok(rows) and all(a.heads == 8 and a.inner_dim == 128 for a, _ in self.layers):
            # bf16 activations (autocast): one fused stack with the residual stream in fp32 between the blocks
            for attn, ff in self.layers:
                _no_dropout(attn.dropout_p if self.training else 0, "Attention")
                _no_dropout(ff.net[2].p if self.training else 0, "FeedForward")
            a0 = self.layers[0][0]
            ws = [(a.to_qkv.weight, a.to_out[0].weight, f.net[0].weight, f.net[3].weight) for a, f in self.layers]
            return PF.transformer_wide(rows, ws, b * p, n, a0.scale, a0.norm.eps).view(b, p, n, d)
        for attn, ff in self.layers:
            rows = attn.rows(rows, b * p, n)       # attn(x) + x
            rows = ff.rows(rows)                   # ff(x) + x
        return rows.view(b, p, n, d)


class MobileViTBlock(nn.Module):
    def __init__(self, embed_dim: int, depth: int, channel: int, kernel_size: Union[int, Tuple[int, ...]], patch_size: int,
                 ffw_dim: int, dropout: float = 0., use_linear_attention: bool = False) -> None:
        super().__init__()
        if use_linear_attention:
            # the reference accepts this flag and never reads it (mobilevit.py:106-117 always builds the softmax
            # Transformer; mobilevitv2.LinearSelfAttention is 2-D and unregistered): same here, loudly
            import warnings
            warnings.warn("picklebot_amd.MobileViTBlock: use_linear_attention=True is accepted and IGNORED, exactly as in the "
                          "reference (mobilevit.py:106-117 never reads it); the block runs softmax attention", stacklevel=2)
        self.pt, self.ph, self.pw = patch_size
        self.conv1 = conv_nxn_bn(channel, channel, kernel_size)
        self.conv2 = conv_1x1_bn(channel, embed_dim)
        self.transformer = Transformer(embed_dim=embed_dim, depth=depth, heads=8, dim_head=16, ffw_dim=ffw_dim, dropout=dropout)
        self.conv3 = conv_1x1_bn(embed_dim, channel)
        self.conv4 = conv_nxn_bn(2 * channel, channel, kernel_size)

    @torch.compiler.disable
    @PF.scoped
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y5 = to_ndhwc(x, compute_dtype(x))
        a5 = run_conv_1x1_bn(self.conv2, run_conv_nxn_bn(self.conv1, y5))
        thw = tuple(a5.shape[1:4])
        patch = (self.pt, self.ph, self.pw)
        tok = PF.unfold(a5, patch)                      # zero-pads up to patch multiples (mobilevit.py:129-134)
        tok = self.transformer(tok)
        v5 = PF.fold(tok, thw, patch)                   # and crops (mobilevit.py:136-137)
        c5 = run_conv_1x1_bn(self.conv3, v5)
        out5 = run_conv_nxn_bn(self.conv4, PF.concat_channels(c5, y5))
        return from_ndhwc(out5)


class _MeanTHW(nn.Module):
    """Parameter-free stand-in for einops Reduce('b c t h w -> b c', 'mean') at to_logits[1] (mobilevit.py:191)."""

    def forward(self, x):
        return x.mean(dim=(2, 3, 4))


class MobileViT(nn.Module):
    def __init__(self, dims: List[int], channels: List[int], num_classes: int, expansion: int = 4,
                 kernel_size: Union[int, Tuple[int, ...]] = 3, patch_size: Tuple[int, ...] = (2, 2, 2),
                 depths: Tuple[int, ...] = (2, 4, 3), use_linear_attention: bool = False):
        super().__init__()
        assert len(dims) == 3, 'dims must be a tuple of 3'
        assert len(depths) == 3, 'depths must be a tuple of 3'
        ch = channels
        self.conv1 = conv_nxn_bn(3, ch[0], stride=2)
        stem_rows = [(ch[0], ch[1], ch[0] * expansion, 1), (ch[1], ch[2], ch[1] * expansion, 2),
                     (ch[2], ch[3], ch[2] * expansion, 1), (ch[2], ch[3], ch[2] * expansion, 1)]
        self.stem = nn.ModuleList([Bottleneck3D(a, b, expanded_channels=e, stride=s) for a, b, e, s in stem_rows])
        trunk_rows = [(ch[3], ch[4], ch[3] * expansion, dims[0], depths[0], ch[5], int(dims[0] * 2)),
                      (ch[5], ch[6], ch[7] * expansion, dims[1], depths[1], ch[7], int(dims[1] * 4)),
                      (ch[7], ch[8], ch[7] * 4, dims[2], depths[2], ch[9], int(dims[2] * 4))]
        self.trunk = nn.ModuleList([nn.ModuleList([
            Bottleneck3D(a, b, expanded_channels=e, stride=2),
            MobileViTBlock(d, dep, c, kernel_size, patch_size, f, use_linear_attention=use_linear_attention)])
            for a, b, e, d, dep, c, f in trunk_rows])
        self.to_logits = nn.Sequential(conv_1x1_bn(ch[-2], ch[-1]), _MeanTHW(), nn.Linear(ch[-1], num_classes, bias=False))

    @torch.compiler.disable
    @PF.scoped
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        dt = compute_dtype(x)
        PF.prepare_model_weights(self, x, dt)
        conv, bn, act_m = self.conv1
        act, slope = act_of(act_m)
        y5 = PF.stem_bn_act(x, conv, bn, act=act, slope=slope, training=bn.training, dtype=dt,
                            x_scale=(1.0 / 255.0) if x.dtype == torch.uint8 else 1.0)
        y = from_ndhwc(y5)
        for blk in self.stem:
            y = blk(y)
        for blk, vit in self.trunk:
            y = vit(blk(y))
        y5 = run_conv_1x1_bn(self.to_logits[0], to_ndhwc(y))
        pooled = PF.pool_mean(y5)
        fc = self.to_logits[2]
        return PF.pw_bn_act(pooled, fc.weight, fc.bias, None, act="none", slope=0.0, training=self.training)

    def initialize_weights(self) -> None:      # mobilevit.py:207-215
        for module in self.modules():
            if isinstance(module, (nn.Linear, nn.Conv2d, nn.Conv3d)):
                module.weight.data.normal_(mean=0.0, std=0.02)
                if module.bias is not None:
                    module.bias.data.zero_()
