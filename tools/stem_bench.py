"""Stand-alone timing of the stem forward / weight gradient at the bench shape (HIP events, median of 10).
    python tools/stem_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from picklebot_amd import kernels as K  # noqa: E402

dev = "cuda:0"
N = int(os.environ.get("N", 32))
x = (torch.rand(N, 16, 224, 224, 3, device=dev) * 255).to(torch.bfloat16).permute(0, 4, 1, 2, 3)      # NCDHW view of an NDHWC clip
w = torch.randn(16, 3, 3, 3, 3, device=dev) * 0.1
b = torch.randn(16, device=dev) * 0.1
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def timeit(fn, iters=10):
    ts = []
    for i in range(iters + 2):
        flush.fill_(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[2:])
    return ts[len(ts) // 2]


st = torch.zeros(K.STEM_REPL, 2, 16, device=dev)
y = K.stem_fwd(x, w, b, (3, 3, 3), (2, 2, 2), (1, 1, 1), torch.bfloat16, 1.0, st)
print("y", tuple(y.shape), "fwd %.1f us" % timeit(lambda: K.stem_fwd(x, w, b, (3, 3, 3), (2, 2, 2), (1, 1, 1), torch.bfloat16, 1.0, st)))
dy = torch.randn_like(y)
dw, db = torch.zeros_like(w), torch.zeros_like(b)
print("wgrad %.1f us" % timeit(lambda: K.stem_bwd_weight(dy, x, dw, db, (3, 3, 3), (2, 2, 2), (1, 1, 1))))
