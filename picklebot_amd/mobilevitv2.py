"""MobileViT-V2 separable self-attention on 3-D patches.

The reference's mobilevitv2.py is a 2-D, ImageNet-shaped draft that train.py never registers; its README names
"MobileViT V1 & V2" as the roadmap.  What carries over to video is `LinearSelfAttention` (mobilevitv2.py:9-55):
the same module here (constructor signature, attribute names and state_dict keys `qkv_proj.*` / `out_proj.*`), with
"pixels in patch" = the pt*ph*pw voxels of a 3-D patch and "patches" = the t*h*w patch grid - i.e. exactly the token
tensor [B, P, n, D] that the 3-D unfold of MobileViTBlock produces (mobilevit.py:131-135).  The debug prints of the
reference forward (:30-38) are not reproduced.
"""
import torch
import torch.nn as nn

from . import functional as PF
from .mobilenet import compute_dtype


class LinearSelfAttention(nn.Module):
    def __init__(self, embed_dim: int, dropout: float = 0.0) -> None:
        super().__init__()
        self.qkv_proj = nn.Conv2d(in_channels=embed_dim, out_channels=1 + (2 * embed_dim), bias=True, kernel_size=1)
        self.attn_dropout = nn.Dropout(p=dropout)
        self.out_proj = nn.Conv2d(in_channels=embed_dim, out_channels=embed_dim, bias=True, kernel_size=1)
        self.embed_dim = embed_dim

    def rows(self, rows, G, n):
        """Token rows [G*n, D] (G = batch * pixels-in-patch, softmax over the n patches of a group) -> [G*n, D]."""
        if self.training and self.attn_dropout.p:
            raise NotImplementedError("picklebot_amd.LinearSelfAttention: dropout > 0 is not built")
        D = self.embed_dim
        if D > 256:
            raise NotImplementedError("picklebot_amd.LinearSelfAttention: embed_dim <= 256")
        return PF.linattn_block(rows, self.qkv_proj.weight.view(1 + 2 * D, D), self.qkv_proj.bias,
                                self.out_proj.weight.view(D, D), self.out_proj.bias, G, n)

    @torch.compiler.disable
    @PF.scoped
    def forward_tokens(self, tokens: torch.Tensor) -> torch.Tensor:
        """[B, P, n, D] channels-last tokens of the 3-D unfold -> same shape."""
        b, p, n, d = tokens.shape
        rows = tokens.reshape(-1, d)
        if rows.dtype != compute_dtype(tokens):
            rows = rows.to(compute_dtype(tokens))
        return self.rows(rows, b * p, n).view(b, p, n, d)

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        """The reference layout (batch, embed_dim, pixels-in-patch, patches) (mobilevitv2.py:29); the result has the
        same logical shape (channels-last in memory)."""
        return self.forward_tokens(hidden_states.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
