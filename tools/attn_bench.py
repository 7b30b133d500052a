"""MobileViT attention kernels alone: us per launch and the fraction of the dense bf16 MFMA peak (2.5 PFLOP/s,
MI355X_MICROARCH.md), forward and backward, bf16.    python tools/attn_bench.py [G n] [--iters N] [--fwd-only]
Algorithmic flops: 4 n^2 d per (group, head) forward (QK^T and PV), 2.5x that backward (five products)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from picklebot_amd import kernels as K  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 20
if "--iters" in sys.argv:
    args.remove(str(iters))
shapes = [(int(args[0]), int(args[1]))] if len(args) >= 2 else [(64, 768), (128, 784), (64, 128), (128, 147), (64, 32)]
dev = torch.device("cuda:0")


def timed(fn):
    """us per launch with `iters` launches replayed from ONE HIP graph: an eager loop of these launches is host-bound (output
    allocation + ctypes + the launch itself ~ 90 us per iteration in round 3's numbers, whatever the kernel did)."""
    fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for G, n in shapes:
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(G * n, 384, generator=g).to(torch.bfloat16).to(dev)
    dout = torch.randn(G * n, 128, generator=g).to(torch.bfloat16).to(dev)
    out, lse = K.attention_fwd(qkv, G, n, 0.25)
    flops = 4.0 * n * n * 16 * 8 * G
    us = timed(lambda: K.attention_fwd(qkv, G, n, 0.25))
    line = "G%-4d n%-4d fwd %7.1f us = %.3f of peak" % (G, n, us, flops / (us * 1e-6) / 2.5e15)
    if "--fwd-only" not in sys.argv:
        us = timed(lambda: K.attention_bwd(qkv, out, dout, lse, G, n, 0.25))
        line += "   bwd %7.1f us = %.3f of peak" % (us, 2.5 * flops / (us * 1e-6) / 2.5e15)
    print(line, flush=True)
