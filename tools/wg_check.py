import os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from picklebot_amd import kernels as K
torch.manual_seed(0)
junk = torch.randn(64 << 20, device="cuda")          # dirty the caches / LDS users between cases
for rep in range(2):
    for (M, Kd, N) in [(48, 16, 384), (48, 128, 16), (64, 128, 16), (32, 128, 16), (96, 128, 16), (48, 128, 128), (48, 16, 128), (40, 72, 24),
                       (200, 16, 384), (5000, 24, 72), (150, 160, 960), (3000, 672, 112), (7000, 112, 672)]:
        x = torch.randn(M, Kd, device="cuda").bfloat16(); dy = torch.randn(M, N, device="cuda").bfloat16()
        dw = torch.zeros(N, Kd, device="cuda")
        junk.mul_(1.0001)
        K.pw_bwd_weight(dy, x, dw)
        ref = dy.float().t() @ x.float()
        err = ((dw - ref).norm() / ref.norm()).item()
        G = 4 if M % 4 == 0 and M // 4 >= 64 else 0
        msg = ""
        if G:
            rs = torch.rand(G, Kd, device="cuda") + 0.5
            dw2 = torch.zeros(N, Kd, device="cuda")
            K.pw_bwd_weight(dy, x, dw2, row_scale=rs, rows_per_group=M // G)
            xs = x.float() * rs.repeat_interleave(M // G, 0)
            ref2 = dy.float().t() @ xs
            msg = "  se err %.3g" % ((dw2 - ref2).norm() / ref2.norm()).item()
        print(M, Kd, N, "rel err %.3g" % err, msg)
