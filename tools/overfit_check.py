"""Sanity beyond one step: train on ONE fixed synthetic batch and watch the loss fall (catches a wrong-signed or
mis-scaled gradient that a per-step parity tolerance could hide).  python tools/overfit_check.py [model] [steps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import picklebot_amd as pb

name = sys.argv[1] if len(sys.argv) > 1 else "large"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda:0")
torch.manual_seed(0)
if name == "large":
    model, shape = pb.MobileNetLarge3D(num_classes=13), (8, 16, 112, 112, 3)
elif name == "movinet":
    model, shape = pb.MoViNetA2(num_classes=13), (4, 16, 96, 96, 3)
else:
    model = pb.MobileViT(num_classes=13, dims=[64, 80, 96], channels=[16, 16, 24, 24, 48, 48, 64, 64, 80, 80, 320])
    shape = (4, 16, 128, 128, 3)
model = model.to(dev).train()
for m in model.modules():
    if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout3d)):
        m.p = 0.0
x = torch.rand(shape, device=dev).permute(0, 4, 1, 2, 3).to(torch.bfloat16)
y = torch.randint(0, 13, (shape[0],), device=dev)
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)
first = last = None
for i in range(steps):
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = torch.nn.functional.cross_entropy(model(x).float(), y)
    loss.backward()
    opt.step()
    v = float(loss)
    first = v if first is None else first
    last = v
    if i % 10 == 0 or i == steps - 1:
        print(f"step {i:3d} loss {v:.4f}", flush=True)
assert last == last, "loss is NaN"
assert last < 0.5 * first, f"loss did not fall: {first:.4f} -> {last:.4f}"
print(f"OK: {first:.4f} -> {last:.4f}")
