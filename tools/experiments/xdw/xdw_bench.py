"""Time the fused expand+depthwise kernels on the MobileNetLarge3D shapes (batch 32), cold-ish (a 1 GB buffer is touched
between launches so operands come from HBM): python tools/xdw_bench.py [layer indices...]   env: PB_XDW_TILE / _EC / _ESPLIT / _DBG"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from picklebot_amd import kernels as K
L = [(16, 16, (8, 112, 112), 3, 1), (16, 64, (10, 112, 112), 3, 2), (24, 72, (6, 56, 56), 3, 1), (24, 72, (8, 56, 56), 5, 2),
     (40, 120, (6, 28, 28), 5, 1), (40, 120, (10, 28, 28), 5, 1), (40, 240, (14, 28, 28), 3, 2), (80, 240, (8, 14, 14), 3, 1),
     (80, 184, (10, 14, 14), 3, 1), (80, 184, (12, 14, 14), 3, 1), (80, 480, (14, 14, 14), 3, 1), (112, 672, (16, 14, 14), 3, 1),
     (112, 672, (18, 14, 14), 5, 2), (160, 960, (11, 7, 7), 5, 1), (160, 960, (15, 7, 7), 5, 1)]
which = [int(a) for a in sys.argv[1:]] or range(len(L))
dev = "cuda:0"
flush = torch.empty(256 << 20, dtype=torch.float32, device=dev)
for li in which:
    Cin, E, thw, k, s = L[li]
    kernel, stride, pad = (1, k, k), (s, s, s), (k // 2,) * 3
    x = torch.randn((32,) + thw + (Cin,), device=dev).bfloat16()
    w1 = torch.randn(E, Cin, device=dev) * Cin ** -0.5
    wdw = torch.randn(E, k * k, device=dev) / k
    with K.weight_scope():
        d, _ = K.xdw_fwd(x, w1, wdw, kernel, stride, pad, pool=True)
        dy = torch.randn_like(d)
        dw = torch.zeros(E, k * k, device=dev)
        res = []
        for name, fn in (("fwd", lambda: K.xdw_fwd(x, w1, wdw, kernel, stride, pad, pool=True)),
                         ("wgrad", lambda: K.xdw_bwd_weight(dy, x, w1, dw, kernel, stride, pad))):
            ts = []
            for it in range(6):
                flush.add_(1.0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            res.append(f"{name} {min(ts[1:]):7.1f} us")
        nb_f = x.numel() * 2 + d.numel() * 2
        print(f"L{li + 1:2d} Cin{Cin:3d} E{E:3d} {thw} k{k} s{s}: " + "  ".join(res) + f"   (fwd alg bytes {nb_f / 1e6:.0f} MB -> {nb_f / 1e6 / 6.0:.0f} us at 6 TB/s)", flush=True)
