"""How many zero-filled fp32 elements does one MobileNetLarge3D training step take from the arena, by shape?"""
import collections, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import picklebot_amd
from picklebot_amd import kernels as K
dev = torch.device("cuda:0")
m = picklebot_amd.MobileNetLarge3D(13).to(dev).train()
opt = picklebot_amd.FlatAdamW(m.parameters(), lr=3e-4, weight_decay=5e-4)
x = (torch.randint(0, 256, (32, 16, 224, 224, 3), dtype=torch.uint8, device=dev).permute(0, 4, 1, 2, 3).to(torch.bfloat16))
y = torch.randint(0, 13, (32,), device=dev)
def step():
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = torch.nn.functional.cross_entropy(m(x).float(), y)
    loss.backward(); opt.step()
for _ in range(2): step()
cnt = collections.Counter(); calls = collections.Counter()
orig = K.zeros
def zeros(shape, device):
    shape = tuple(shape) if not isinstance(shape, int) else (shape,)
    n = 1
    for d in shape: n *= int(d)
    key = tuple(shape) if len(shape) != 3 else (shape[0], shape[1], "C")
    cnt[key] += n; calls[key] += 1
    return orig(shape, device)
K.zeros = zeros
import picklebot_amd.functional as PF
step(); torch.cuda.synchronize()
tot = sum(cnt.values())
print("total %.1f MB of zero-filled fp32 per step in %d requests" % (tot * 4 / 1e6, sum(calls.values())))
for k, v in cnt.most_common(12): print("  %-28s %8.2f MB in %3d requests" % (k, v * 4 / 1e6, calls[k]))
