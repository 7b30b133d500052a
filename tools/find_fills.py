"""Which Python lines launch aten::fill_ / zero_ / copy_ kernels during one training step (torch.profiler with stacks)."""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import picklebot_amd as pb
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
m = pb.MobileNetLarge3D(num_classes=13).to(dev).train()
opt = torch.optim.AdamW(m.parameters(), lr=3e-4, fused=True)
x = (torch.rand(8, 16, 224, 224, 3, device=dev).permute(0, 4, 1, 2, 3)).to(torch.bfloat16)
y = torch.randint(0, 13, (8,), device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = torch.nn.functional.cross_entropy(m(x).float(), y)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::zeros_like", "aten::ones_like", "aten::sum", "aten::mul", "aten::add", "aten::div"):
        st = [f for f in (e.stack or []) if "picklebot_amd" in f or "bench" in f or "find_fills" in f]
        cnt[(e.name, st[0] if st else (e.stack[0] if e.stack else "?"))] += 1
for (n, s), c in cnt.most_common(40):
    print(c, n, s)
