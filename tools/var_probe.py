"""How accurate is the single-pass variance of the BatchNorm statistics (fp32 slot sums of x and x^2, fp64 finalize) when a
channel's |mean| >> std?  Prints the relative error of var and rstd against fp64 for several mean/std ratios."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from picklebot_amd import kernels as K
dev = "cuda:0"
torch.manual_seed(0)
for M in (4096, 200000, 4000000):
    for ratio in (0, 3, 10, 30, 100):
        C = 32
        x = (torch.randn(M, C, dtype=torch.float64) + ratio).float().to(dev)
        st = torch.zeros(K.REPL, 2, C, device=dev)
        K.chan_stats(x, st)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        scale, shift, mean, rstd = K.bn_finalize(st, M, None, None, 1e-5, 0.1, rm, rv, False, C, dev)
        xd = x.double()
        var = xd.var(0, unbiased=False)
        ref = 1.0 / torch.sqrt(var + 1e-5)
        print(f"M {M:8d} mean/std {ratio:4d}: rstd rel err max {float(((rstd.double() - ref) / ref).abs().max()):.2e}  mean rel err {float(((mean.double() - xd.mean(0)) / xd.mean(0).abs().clamp_min(1e-9)).abs().max()):.2e}")
