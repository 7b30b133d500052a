"""Accuracy of the bf16 attention kernels against an fp64 softmax of the same bf16 inputs: forward output, log-sum-exp
(max abs) and the gradient computed from the kernel's own log-sum-exp.    python tools/probes/attn_accuracy.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from picklebot_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")


def ref(qkv, G, n, dout):
    x = qkv.double().reshape(G, n, 3, 8, 16).permute(2, 0, 3, 1, 4).clone().requires_grad_(True)   # [3][G][h][n][d]
    q, k, v = x[0], x[1], x[2]
    s = (q @ k.transpose(-1, -2)) * 0.25
    lse = torch.logsumexp(s, -1)
    o = torch.softmax(s, -1) @ v
    out = o.permute(0, 2, 1, 3).reshape(G * n, 128)
    out.backward(dout.double())
    dq = x.grad.permute(1, 3, 0, 2, 4).reshape(G * n, 384)
    return out.detach(), lse.permute(0, 2, 1).reshape(G * n, 8).detach(), dq


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


for name, amp, n in [("flat 0.3", 0.3, 784), ("unit 1.0", 1.0, 784), ("peaked 2.5", 2.5, 784), ("peaked 2.5 n=147", 2.5, 147), ("unit n=32", 1.0, 32)]:
    G = 8
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn(G * n, 384, generator=g) * amp).to(torch.bfloat16)
    dout = torch.randn(G * n, 128, generator=g).to(torch.bfloat16)
    out_r, lse_r, dq_r = ref(qkv, G, n, dout)
    line = "%-18s" % name
    for flag in ("1",):
        out, lse = K.attention_fwd(qkv.to(dev), G, n, 0.25)
        dq = K.attention_bwd(qkv.to(dev), out, dout.to(dev), lse, G, n, 0.25)
        line += "  | out %.2e lse %.2e dqkv %.2e" % (rel(out.cpu(), out_r), float((lse.cpu().double() - lse_r).abs().max()), rel(dq.cpu(), dq_r))
    print(line)
