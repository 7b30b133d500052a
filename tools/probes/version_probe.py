import torch
p = torch.nn.Parameter(torch.randn(64, 64, device="cuda"))
q = torch.nn.Parameter(torch.randn(64, 64, device="cuda"))
for fused in (True, False):
    opt = torch.optim.AdamW([p, q], lr=1e-3, fused=fused)
    v0 = p.view(32, 128)._version
    p.grad = torch.randn_like(p); q.grad = torch.randn_like(q)
    opt.step()
    print("fused" if fused else "foreach", "version before", v0, "after", p.view(32, 128)._version, p._version)
