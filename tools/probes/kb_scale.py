"""Time the K = 672 -> N = 112 projection GEMM and a plain elementwise pass over the same X for several M (tiles per CU)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kbench
from picklebot_amd import kernels as K
for M in (32768, 65536, 131072, 196608, 262144, 112896):
    kbench.pw(M, 672, 112)
    x = torch.randn(M, 672, device="cuda:0").to(torch.bfloat16)
    kbench.report(f"affine_act_fwd M{M} C672 (read + write)", kbench.timeit(lambda: K.affine_act_fwd(x, act="relu")), 2 * M * 672 * 2)
    st = torch.zeros(K.REPL, 2, 672, device="cuda:0")
    kbench.report(f"chan_stats M{M} C672 (read only)", kbench.timeit(lambda: K.chan_stats(x, st)), M * 672 * 2)
