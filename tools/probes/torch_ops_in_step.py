import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import picklebot_amd
from torch.profiler import profile, ProfilerActivity
torch.manual_seed(0)
dev = torch.device("cuda:0")
m = picklebot_amd.MobileNetLarge3D(13).to(dev).train()
opt = picklebot_amd.FlatAdamW(m.parameters(), lr=3e-4, weight_decay=5e-4)
x = (torch.randint(0, 256, (8, 16, 224, 224, 3), dtype=torch.uint8, device=dev).permute(0, 4, 1, 2, 3).to(torch.bfloat16) / 255)
y = torch.randint(0, 13, (8,), device=dev)
def step():
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = torch.nn.functional.cross_entropy(m(x).float(), y)
    loss.backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA: continue
    if e.name.startswith("aten::") and e.cpu_parent is not None and not e.cpu_parent.name.startswith("aten::") or (e.name.startswith("aten::") and e.cpu_parent is None):
        rows.append(e.name)
import collections
c = collections.Counter(rows)
for k, v in c.most_common(40): print(v, k)
print("--- with stacks for copy_/contiguous/to")
seen = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::contiguous", "aten::to", "aten::clone", "aten::add_", "aten::mul", "aten::sum", "aten::fill_", "aten::zero_") and e.device_type != torch.autograd.DeviceType.CUDA:
        st = [s for s in (e.stack or []) if "picklebot_amd" in s or "bench" in s][:2]
        seen[(e.name, tuple(st))] += 1
for (n, st), v in seen.most_common(40): print(v, n, " <- ".join(s.split("/")[-1] for s in st))
