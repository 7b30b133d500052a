"""Depthwise kernels of five MobileNetLarge3D layers ALONE, the same buffers launch after launch: input + output (50-260 MB)
then sit in the 256 MB memory-side cache, so this is what the kernels do when they are NOT waiting for HBM - the number to
hold against their in-step rate (bench.py's family table), where 20 GB per step stream through that cache.
    python tools/probes/dw_standalone.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from picklebot_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, it=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for (N, C, T, H, W, k, s) in [(32, 16, 8, 112, 112, 3, 1), (32, 64, 5, 112, 112, 3, 2), (32, 72, 6, 56, 56, 3, 1),
                              (32, 120, 6, 28, 28, 5, 1), (32, 672, 16, 14, 14, 3, 1)]:
    x = torch.randn(N, T, H, W, C, device=dev).to(torch.bfloat16)
    w = torch.randn(C, k * k, device=dev)
    kern, st, pad = (1, k, k), (1, s, s), (0, k // 2, k // 2)
    y = K.dw_fwd(x, w, kern, st, pad)
    nb = (x.numel() + y.numel()) * 2
    t = timed(lambda: K.dw_fwd(x, w, kern, st, pad))
    dy = torch.randn_like(y)
    tb = timed(lambda: K.dw_bwd_data(dy, w, (T, H, W), kern, st, pad))
    print(f"C{C} {T}x{H}x{W} k{k} s{s} ({nb / 1e6:.0f} MB): forward {t:6.1f} us = {nb / t / 1e6:4.2f} TB/s = {nb / t / 8e6:.2f} of the HBM peak   "
          f"backward-data {tb:6.1f} us = {nb / tb / 1e6:4.2f} TB/s", flush=True)
