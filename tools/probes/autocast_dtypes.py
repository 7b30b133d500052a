"""What dtype does the reference's Transformer residual stream have under `torch.autocast('cuda', bfloat16)` (train.py:264)?
Plain torch on the GPU, no picklebot_amd: the op sequence of mobilevit.py:53-61,68-86,100-104 (LayerNorm -> Linear -> SDPA ->
Linear, `attn(x) + x`, Linear -> SiLU -> Linear, `ff(x) + x`) with the dtype of every intermediate printed.
    python tools/probes/autocast_dtypes.py"""
import torch

dev = "cuda:0"
D, n = 144, 768
norm = torch.nn.LayerNorm(D, elementwise_affine=False).to(dev)
to_qkv = torch.nn.Linear(D, 384, bias=False).to(dev)
to_out = torch.nn.Linear(128, D, bias=False).to(dev)
ff1, ff2 = torch.nn.Linear(D, 288, bias=False).to(dev), torch.nn.Linear(288, D, bias=False).to(dev)
conv = torch.nn.Conv3d(96, D, 1, bias=False).to(dev)
bn = torch.nn.BatchNorm3d(D, affine=False).to(dev)
x0 = torch.randn(1, 96, 6, 16, 16, device=dev).to(torch.bfloat16)        # train.py:106 hands the model bf16 clips
with torch.autocast("cuda", dtype=torch.bfloat16):
    c = conv(x0)
    b = bn(c)
    a = torch.nn.functional.silu(b)
    print("conv2 -> BatchNorm3d -> SiLU (the tokens' source):", c.dtype, b.dtype, a.dtype)
    x = a.permute(0, 2, 3, 4, 1).reshape(1, 2, n, D)
    xn = norm(x)
    qkv = to_qkv(xn)
    q, k, v = (t.view(1, 2, n, 8, 16).transpose(2, 3) for t in qkv.chunk(3, dim=-1))
    o = torch.nn.functional.scaled_dot_product_attention(q, k, v)
    y = to_out(o.transpose(2, 3).reshape(1, 2, n, 128))
    s1 = y + x
    h = ff2(torch.nn.functional.silu(ff1(s1)))
    s2 = h + s1
    print("stream in            :", x.dtype)
    print("LayerNorm out        :", xn.dtype)
    print("to_qkv out           :", qkv.dtype)
    print("SDPA out             :", o.dtype)
    print("to_out out           :", y.dtype)
    print("attn(x) + x          :", s1.dtype)
    print("ff(x)                :", h.dtype)
    print("ff(x) + x            :", s2.dtype)
    print("LayerNorm(next block):", norm(s2).dtype)
