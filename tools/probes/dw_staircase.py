"""Time of ONE depthwise forward launch against the number of clips (= work units), operands rotated over > 512 MB of
buffers so that nothing is served by the 256 MB memory-side cache: is the launch time a staircase in the number of
resident-wave rounds (tail effect), a line through the origin (bandwidth) or a line with an offset (latency floor)?
    python tools/probes/dw_staircase.py [C T H W k s]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from picklebot_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
args = [int(v) for v in sys.argv[1:7]] if len(sys.argv) >= 7 else [672, 16, 14, 14, 3, 1]
C, T, H, W, k, s = args
kern, st, pad = (1, k, k), (1, s, s), (0, k // 2, k // 2)
w = torch.randn(C, k * k, device=dev)
print(f"C{C} {T}x{H}x{W} k{k} s{s}   kernel {K.dw_kernel_name(0, 1, C, (T, H, W), kern, st, pad, torch.bfloat16) if hasattr(K, 'dw_kernel_name') else ''}")
for N in (4, 8, 12, 16, 20, 24, 28, 32, 40, 48, 64, 96):
    per = N * T * H * W * C * 2
    nbuf = max(2, int(600e6 // per) + 1)
    xs = [torch.randn(N, T, H, W, C, device=dev).to(torch.bfloat16) for _ in range(nbuf)]
    y = K.dw_fwd(xs[0], w, kern, st, pad)
    nb = (xs[0].numel() + y.numel()) * 2
    torch.cuda.synchronize()
    it = 3 * nbuf
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(it):
        K.dw_fwd(xs[i % nbuf], w, kern, st, pad)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / it * 1e3
    print(f"  N {N:3d}  frames {N * T:5d}  {nb / 1e6:6.1f} MB  {t:7.1f} us  {nb / t / 1e6:5.2f} TB/s", flush=True)
    del xs, y
